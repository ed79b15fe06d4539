"""Stride 2, 128-bit entries {loA, hiA, loB, hiB}, index = hash of the TWO bytes c[q-1], c[q] that the literals ending at q and at
q + 1 both contain whatever their length (>= 3): nothing is enumerated.
  A (ends at q):     loA bit c[q-2] & 31, hiA bit c[q-3] & 31 (3-byte literals: all of hiA)
  B (ends at q + 1): loB bit c[q+1] & 31, hiB bit c[q-2] & 31 (3-byte literals: all of hiB)"""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/sim'); sys.path.insert(0, '/root/repo')
from cur import *

def run(k=13, mode="lo5", mix=False):
    E = 1 << k
    LOA = np.zeros(E, np.uint32); HIA = np.zeros(E, np.uint32); LOB = np.zeros(E, np.uint32); HIB = np.zeros(E, np.uint32)
    def idx(c): return (c & 31) if mode == "lo5" else ((c ^ (c >> 3)) & 31) if mode == "x3" else ((c * 5) >> 2) & 31
    ALLB = np.uint32(0xffffffff)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(6)]
        x = v[1] | v[0] << 8
        prod = (x * MUL) & 0xffffffff; e = prod >> (32 - k)
        LOA[e] |= np.uint32(1 << idx(v[2]))
        HIA[e] |= np.uint32(1 << idx(v[3])) if li.len >= 4 else ALLB
        x = v[2] | v[1] << 8
        prod = (x * MUL) & 0xffffffff; e = prod >> (32 - k)
        LOB[e] |= np.uint32(1 << idx(v[0]))
        HIB[e] |= np.uint32(1 << idx(v[3])) if li.len >= 4 else ALLB
    pos = np.arange(0, n, 2)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    nx = B(-1, pos) & 0xdf
    x = b1 | b0 << 8
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - k)
    hitA = ((LOA[e] >> idx(b2)) & (HIA[e] >> idx(b3)) & 1).astype(bool)
    hitB = ((LOB[e] >> idx(nx)) & (HIB[e] >> idx(b2)) & 1).astype(bool)
    hit = np.zeros(n + 1, bool)
    hit[pos[hitA]] = True
    hit[pos[hitB] + 1] = True
    hit = hit[:n]
    assert hit[true_e].all(), "recall"
    d = lambda P: np.unpackbits(P.view(np.uint8)).mean()
    print(f"k={k} mode={mode} dens loA {d(LOA):.4f} hiA {d(HIA):.4f} loB {d(LOB):.4f} hiB {d(HIB):.4f}; hits A {hitA.sum()*16/1e6:.2f}M B {hitB.sum()*16/1e6:.2f}M per GiB")
    report(f"dual2 k={k} {mode}", hit, np.arange(n), 1)

if __name__ == "__main__":
    run(13, "lo5")
    run(13, "x3")
