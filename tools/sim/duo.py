import sys, time, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
from quad4 import *

def run_duo(L, log2e, b2mask, ph_in="lo", ph_k=1, name="", check=True, skipF=False, th_shift=(8, 13)):
    E = 1 << log2e
    LO = np.zeros(E, np.uint32); HI = np.zeros(E, np.uint32); PHX = np.zeros(E, np.uint32)
    nk = 0
    for li in L:
        v = li.val; m = li.msk
        known = [p < li.len for p in range(9)]
        for delta in (0, 1):
            hb = [v[delta + i] & BL for i in range(3)]
            kn = [known[delta + i] for i in range(3)]
            if not kn[2] and skipF: continue
            vals2 = [hb[2] & b2mask] if kn[2] else sorted({c & b2mask & BL for c in range(256)})
            for c2 in vals2:
                x = (c2 & b2mask) | hb[1] << 8 | hb[0] << 16
                prod = (x * MUL) & 0xffffffff; e = prod >> (32 - log2e)
                nk += 1
                i3 = idx_bits(v[delta + 3], m[delta + 3], "lo5") if known[delta + 3] else 0xffffffff
                if delta == 0: i5 = idx_bits(v[4], m[4], "lo5") if known[4] else 0xffffffff
                else: i5 = idx_bits(v[0], m[0], "lo5")
                LO[e] |= i3; HI[e] |= i5
                hb_ = 0
                for s in th_shift[:ph_k]: hb_ |= 1 << ((prod >> s) & 31)
                if ph_in == "lo": LO[e] |= hb_
                elif ph_in == "hi": HI[e] |= hb_
                else: PHX[e] |= hb_
    x = (r2 & BL & b2mask) | (r1 & BL) << 8 | (r0 & BL) << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - log2e)
    lo, hi = LO[e], HI[e]
    php = lo if ph_in == "lo" else hi if ph_in == "hi" else PHX[e]
    hit = (lo >> (r3 & 31)) & ((hi >> (r4 & 31)) | (hi >> (rn & 31)))
    for s in th_shift[:ph_k]: hit = hit & (php >> ((prod >> s) & 31))
    hit = (hit & 1).astype(bool)
    miss = -1
    if check:
        te = np.nonzero(true_e)[0]
        qi = np.where(te % 2 == 0, te, te - 1) // 2
        miss = int((~hit[qi]).sum())
    lanes = np.zeros(n // 16 + 1, bool); lanes[pos[hit] >> 4] = True
    d = lambda P: np.unpackbits(P.view(np.uint8)).mean()
    print(f"{name} duo log2e={log2e} b2mask={b2mask:x} ph_in={ph_in} k={ph_k} skipF={skipF} keys={nk} dens LO {d(LO):.3f} HI {d(HI):.3f} PHX {d(PHX):.3f}"
          f" | cand pos {hit.sum()*16/1e6:.2f}M/GiB lanes {lanes.sum()*16/1e6:.2f}M/GiB; missed {miss}", flush=True)
if __name__ == "__main__":
    run_duo(ALL, 14, 0x1f, "lo", 1, "all")
    run_duo(ALL, 14, 0x1f, "lo", 2, "all")
    run_duo(ALL, 14, 0x1f, "x", 2, "all(3 planes)")
    run_duo(ALL, 14, 0x1f, "lo", 1, "all", check=False, skipF=True)
    run_duo(ALL, 14, 0xff, "lo", 1, "all-b2full", check=False, skipF=True)
    run_duo([l for l in ALL if l.len >= 5], 14, 0xff, "lo", 1, "long", check=False)
    run_duo([l for l in ALL if l.len >= 4], 14, 0xff, "lo", 1, "len>=4", check=False)
