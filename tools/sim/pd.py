import sys, time, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
from quad4 import *

def run_pd(L, log2e, b2mask, sh=(8, 13), name="", check=True, use5=False):
    """per-delta planes: lo = delta0 keys (bit b3&31 + pure-hash bit), hi = delta+1 keys (bit nx&31 + pure-hash bit)"""
    E = 1 << log2e
    LO = np.zeros(E, np.uint32); HI = np.zeros(E, np.uint32)
    nk = 0
    for li in L:
        v = li.val; m = li.msk
        known = [p < li.len for p in range(9)]
        for delta in (0, 1):
            hb = [v[delta + i] & BL for i in range(3)]
            kn = [known[delta + i] for i in range(3)]
            vals2 = [hb[2] & b2mask] if kn[2] else sorted({c & b2mask & BL for c in range(256)})
            for c2 in vals2:
                x = (c2 & b2mask) | hb[1] << 8 | hb[0] << 16
                prod = (x * MUL) & 0xffffffff; e = prod >> (32 - log2e)
                nk += 1
                if delta == 0:
                    i3 = idx_bits(v[3], m[3], "lo5") if known[3] else 0xffffffff
                    LO[e] |= i3 | (1 << ((prod >> sh[0]) & 31))
                else:
                    HI[e] |= idx_bits(v[0], m[0], "lo5") | (1 << ((prod >> sh[1]) & 31))
    x = (r2 & BL & b2mask) | (r1 & BL) << 8 | (r0 & BL) << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - log2e)
    lo, hi = LO[e], HI[e]
    h0 = (lo >> (r3 & 31)) & (lo >> ((prod >> sh[0]) & 31))
    h1 = (hi >> (rn & 31)) & (hi >> ((prod >> sh[1]) & 31))
    hit = ((h0 | h1) & 1).astype(bool)
    miss = -1
    if check:
        te = np.nonzero(true_e)[0]
        qi = np.where(te % 2 == 0, te, te - 1) // 2
        miss = int((~hit[qi]).sum())
    lanes = np.zeros(n // 16 + 1, bool); lanes[pos[hit] >> 4] = True
    d = lambda P: np.unpackbits(P.view(np.uint8)).mean()
    print(f"{name} pd log2e={log2e} b2mask={b2mask:x} sh={sh} keys={nk} dens LO {d(LO):.3f} HI {d(HI):.3f}"
          f" | cand pos {hit.sum()*16/1e6:.2f}M/GiB (d0 {(h0&1).sum()*16/1e6:.2f} d1 {(h1&1).sum()*16/1e6:.2f}) lanes {lanes.sum()*16/1e6:.2f}M/GiB; missed {miss}", flush=True)
if __name__ == "__main__":
    run_pd(ALL, 14, 0x1f, name="all")
    run_pd(ALL, 14, 0xdf, name="all")
    run_pd(ALL, 14, 0x1f, sh=(13, 13), name="all")
    run_pd([l for l in ALL if l.len >= 4], 14, 0xdf, name="len>=4", check=False)
    run_pd([l for l in ALL if l.len >= 5], 14, 0xdf, name="long", check=False)
    run_pd([l for l in ALL if l.len == 3], 14, 0x1f, name="len3", check=False)
