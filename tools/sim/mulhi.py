"""HSGPU_F_WIDE with the entry index taken by ONE and from the high half of a 24 x 32-bit product (v_mul_hi_u32): h = (x * M) >> 32,
entry = (h >> 3) & (2^14 - 1) (byte address = h & 0x1fff8), second bit index = 5 bits of h picked by a byte/word select.
One vector instruction less per lookup than mul_u24 + shift + and. Which index bits keep the candidate rate?"""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/sim'); sys.path.insert(0, '/root/repo')
from cur import *
def run(M, hb_lo, k=14, a_lo=3, name=""):
    LO = np.zeros(1 << k, np.uint32); HI = np.zeros(1 << k, np.uint32)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(5)]
        x = v[2] | v[1] << 8 | v[0] << 16
        h = (x * M) >> 32
        e = (h >> a_lo) & ((1 << k) - 1)
        HI[e] |= np.uint32(1 << ((h >> hb_lo) & 31))
        LO[e] |= np.uint32(1 << (v[3] & 31)) if li.len >= 4 else np.uint32(0xffffffff)
    pos = np.arange(n)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    x = (b2 | b1 << 8 | b0 << 16).astype(np.uint64)
    h = ((x * np.uint64(M)) >> np.uint64(32)).astype(np.uint32)
    e = (h >> np.uint32(a_lo)) & np.uint32((1 << k) - 1)
    hit = ((LO[e] >> (b3 & 31)) & (HI[e] >> ((h >> np.uint32(hb_lo)) & 31)) & 1).astype(bool)
    assert hit[true_e].all()
    d = lambda P: np.unpackbits(P.view(np.uint8)).mean()
    report(f"mulhi M={M:#x} addr bits {a_lo}..{a_lo+k-1} hi idx bits {hb_lo}..{hb_lo+4} dens lo {d(LO):.4f} hi {d(HI):.4f} {name}", hit, pos, 1)
def cur_wide(k=14):
    LO = np.zeros(1 << k, np.uint32); HI = np.zeros(1 << k, np.uint32)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(5)]
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff
        e = prod >> (32 - k)
        HI[e] |= np.uint32(1 << (prod & 31))
        LO[e] |= np.uint32(1 << (v[3] & 31)) if li.len >= 4 else np.uint32(0xffffffff)
    pos = np.arange(n)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - k)
    hit = ((LO[e] >> (b3 & 31)) & (HI[e] >> (prod & 31)) & 1).astype(bool)
    report("current WIDE", hit, pos, 1)
if __name__ == "__main__":
    cur_wide()
    for M in (0x9E3779B1, 0x85EBCA6B, 0xC2B2AE35):
        run(M, 16); run(M, 17); run(M, 19, name="(bits 19..23: no overlap; needs a bfe)")
        run(M, 0, name="(low bits of h: overlap 3,4)")

def run15(M15, k=14, hi_from="b2"):
    """h = (x * (M15 << 17)) >> 32 = (x * M15) >> 15 (one v_mul_hi_u32); byte address = h & 0x1fff8, i.e. entry = bits 18..31 of x * M15;
    the second bit index straight from a corpus byte (an SDWA select, like b3's): b2's low five bits -- what prod & 31 is a permutation of"""
    LO = np.zeros(1 << k, np.uint32); HI = np.zeros(1 << k, np.uint32)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(5)]
        x = v[2] | v[1] << 8 | v[0] << 16
        h = (x * (M15 << 17)) >> 32
        e = (h >> 3) & ((1 << k) - 1)
        idx = v[2] & 31 if hi_from == "b2" else v[1] & 31 if hi_from == "b1" else v[0] & 31
        HI[e] |= np.uint32(1 << idx)
        LO[e] |= np.uint32(1 << (v[3] & 31)) if li.len >= 4 else np.uint32(0xffffffff)
    pos = np.arange(n)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    x = (b2 | b1 << 8 | b0 << 16).astype(np.uint64)
    h = ((x * np.uint64(M15 << 17)) >> np.uint64(32)).astype(np.uint32)
    e = (h >> np.uint32(3)) & np.uint32((1 << k) - 1)
    src = b2 if hi_from == "b2" else b1 if hi_from == "b1" else b0
    hit = ((LO[e] >> (b3 & 31)) & (HI[e] >> (src & 31)) & 1).astype(bool)
    assert hit[true_e].all()
    report(f"mulhi15 M15={M15:#x} hi idx from {hi_from}", hit, pos, 1)
if __name__ == "__main__":
    for M15 in (0x4F1B, 0x6A09, 0x79B9, 0x5BD1, 0x7FFF, 0x9E37 >> 1 | 1):
        run15(M15)
    run15(0x4F1B, hi_from="b1"); run15(0x4F1B, hi_from="b0")
