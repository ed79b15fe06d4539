import sys, time, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
from quad4 import *
MUL2 = 0x85EBCA6B
def h2(x, variant):
    if variant == "ideal":  # independent multiplier, top bits
        return ((x * MUL2) & 0xffffffff) >> 27
    if variant == "hi48":   # bits 32..36 of the 48-bit product x*MUL (v_mul_hi_u32_u24)
        return ((x * MUL) >> 32) & 31
    if variant == "mid":    # bits 13..17 of prod
        return ((x * MUL) >> 13) & 31
    raise SystemExit
def run_pd(L, log2e, b2mask, variant, name="", check=True, addr_from="lo32"):
    E = 1 << log2e
    LO = np.zeros(E, np.uint32); HI = np.zeros(E, np.uint32)
    nk = 0
    def addr(x):
        if addr_from == "lo32": return ((x * MUL) & 0xffffffff) >> (32 - log2e)
        if addr_from == "hi48": return ((x * MUL) >> (48 - log2e)) & (E - 1)   # top bits of the 48-bit product
    for li in L:
        v = li.val; m = li.msk
        known = [p < li.len for p in range(9)]
        for delta in (0, 1):
            hb = [v[delta + i] & BL for i in range(3)]
            kn = [known[delta + i] for i in range(3)]
            vals2 = [hb[2] & b2mask] if kn[2] else sorted({c & b2mask & BL for c in range(256)})
            for c2 in vals2:
                x = (c2 & b2mask) | hb[1] << 8 | hb[0] << 16
                e = addr(x); nk += 1
                if delta == 0:
                    i3 = idx_bits(v[3], m[3], "lo5") if known[3] else 0xffffffff
                    LO[e] |= i3 | (1 << int(h2(x, variant)))
                else:
                    HI[e] |= idx_bits(v[0], m[0], "lo5") | (1 << int(h2(x, variant)))
    x = ((r2 & BL & b2mask) | (r1 & BL) << 8 | (r0 & BL) << 16).astype(np.uint64)
    e = addr(x).astype(np.int64)
    jh = h2(x, variant).astype(np.uint32)
    lo, hi = LO[e], HI[e]
    h0 = (lo >> (r3 & 31)) & (lo >> jh)
    h1 = (hi >> (rn & 31)) & (hi >> jh)
    hit = ((h0 | h1) & 1).astype(bool)
    miss = -1
    if check:
        te = np.nonzero(true_e)[0]
        qi = np.where(te % 2 == 0, te, te - 1) // 2
        miss = int((~hit[qi]).sum())
    lanes = np.zeros(n // 16 + 1, bool); lanes[pos[hit] >> 4] = True
    print(f"{name} pd log2e={log2e} b2mask={b2mask:x} h2={variant} addr={addr_from} keys={nk}"
          f" | cand pos {hit.sum()*16/1e6:.2f}M/GiB (d0 {(h0&1).sum()*16/1e6:.2f} d1 {(h1&1).sum()*16/1e6:.2f}) lanes {lanes.sum()*16/1e6:.2f}M/GiB; missed {miss}", flush=True)
if __name__ == "__main__":
    long = [l for l in ALL if l.len >= 5]
    run_pd(long, 14, 0xdf, "ideal", name="long", check=False)
    run_pd(long, 14, 0xdf, "hi48", name="long", check=False)
    run_pd(long, 14, 0xdf, "mid", name="long", check=False, addr_from="hi48")
    run_pd(ALL, 14, 0x1f, "ideal", name="all")
    run_pd(ALL, 14, 0x1f, "hi48", name="all")
