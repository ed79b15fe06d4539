"""Stride 2 with 128-bit entries {loA, hiA, loB, hiB} (2^13 of them in 128 KiB): ONE hash of the three bytes c[q-2..q] two end
offsets share serves the literals ending at q (plane pair A: first bit by c[q-3], second by the product) and those ending at
q + 1 (plane pair B: hashed bytes = the literal's bytes 3, 2, 1 from its end, first bit by its last byte c[q+1], second by other
product bits). A 3-byte literal owns loA of its entry (c[q-3] is outside it) and is enumerated over c[q-2] in the B planes.
Candidate lanes per GiB against HSGPU_F_WIDE (8.5 M); the loop is 11.5 vector instructions per PAIR of positions against 17.5."""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/sim'); sys.path.insert(0, '/root/repo')
from cur import *

def blind_vals():
    return sorted({c & 0xdf for c in range(256)})

def run(k=13, shared_hi=False, hb_shift=8, third=None):
    E = 1 << k
    LOA = np.zeros(E, np.uint32); HIA = np.zeros(E, np.uint32); LOB = np.zeros(E, np.uint32); HIB = np.zeros(E, np.uint32)
    nA = nB = 0
    for li in L:
        v = [li.val[p] & 0xdf for p in range(6)]
        # A: ends at q
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff; e = prod >> (32 - k)
        HIA[e] |= np.uint32(1 << (prod & 31))
        LOA[e] |= np.uint32(1 << (v[3] & 31)) if li.len >= 4 else np.uint32(0xffffffff)
        nA += 1
        # B: ends at q + 1: hashed bytes are v[3], v[2], v[1]; first bit by v[0]
        c2s = [v[3]] if li.len >= 4 else blind_vals()
        for c2 in c2s:
            x = c2 | v[2] << 8 | v[1] << 16
            prod = (x * MUL) & 0xffffffff; e = prod >> (32 - k)
            LOB[e] |= np.uint32(1 << (v[0] & 31))
            if shared_hi: HIA[e] |= np.uint32(1 << (prod & 31))
            else: HIB[e] |= np.uint32(1 << ((prod >> hb_shift) & 31))
            nB += 1
    pos = np.arange(0, n, 2)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    nx = B(-1, pos) & 0xdf
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - k)
    hitA = ((LOA[e] >> (b3 & 31)) & (HIA[e] >> (prod & 31)) & 1).astype(bool)
    if shared_hi: hitB = ((LOB[e] >> (nx & 31)) & (HIA[e] >> (prod & 31)) & 1).astype(bool)
    else: hitB = ((LOB[e] >> (nx & 31)) & (HIB[e] >> ((prod >> hb_shift) & 31)) & 1).astype(bool)
    hit = np.zeros(n + 1, bool)
    hit[pos[hitA]] = True
    hit[pos[hitB] + 1] = True
    hit = hit[:n]
    assert hit[true_e].all(), "recall"
    d = lambda P: np.unpackbits(P.view(np.uint8)).mean()
    allpos = np.arange(n)
    print(f"k={k} shared_hi={shared_hi} hb_shift={hb_shift} keys A {nA} B {nB} dens loA {d(LOA):.4f} hiA {d(HIA):.4f} loB {d(LOB):.4f} hiB {d(HIB):.4f}; hits A {hitA.sum()*16/1e6:.2f}M B {hitB.sum()*16/1e6:.2f}M per GiB")
    report(f"dual128 k={k} shared_hi={shared_hi}", hit, allpos, 1)

if __name__ == "__main__":
    run(13, False)
    run(13, True)
    run(13, False, hb_shift=5)
