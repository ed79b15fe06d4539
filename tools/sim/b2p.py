"""stride-1 two-plane (HSGPU_F_WIDE) variants: lo bit = (a + b3) & 31 or b3 & 31; hi bit = (prod >> hsh) & 31.
The shipped combination (a + b3, hsh = 8) against cheaper / other ones: candidate lanes per GiB."""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/sim'); sys.path.insert(0, '/root/repo')
from cur import *
def two_plane(k=14, hsh=8, offs=True, lo_from="b3"):
    LO = np.zeros(1 << k, np.uint32); HI = np.zeros(1 << k, np.uint32)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(5)]
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff
        e = prod >> (32 - k)
        a = prod >> 15
        HI[e] |= np.uint32(1 << ((prod >> hsh) & 31))
        if li.len >= 4: LO[e] |= np.uint32(1 << (((a if offs else 0) + v[3]) & 31))
        else: LO[e] = 0xffffffff
    pos = np.arange(n)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - k); a = prod >> np.uint32(15)
    hit = ((LO[e] >> (((a if offs else 0) + b3) & 31)) & (HI[e] >> ((prod >> hsh) & 31)) & 1).astype(bool)
    assert hit[true_e].all()
    report(f"two-plane S1 k={k} hsh={hsh} offs={offs}", hit, pos, 1)
if __name__ == "__main__":
    for hsh in (0, 3, 8, 11, 16, 24):
        for offs in (True, False):
            two_plane(14, hsh, offs)
