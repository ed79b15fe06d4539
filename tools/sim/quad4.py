import sys, time, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
import base
from base import Lit, MUL
N = 64 << 20
lits, corpus, off = base.load(N)
true_e = base.true_ends(lits, corpus, off)
ALL = [Lit(l) for l in lits]
pad = base.padded(corpus); n = corpus.size
pos = np.arange(0, n, 2)
def Rq(k): return pad[8 + pos - k]
r0, r1, r2, r3, r4, rn = Rq(0), Rq(1), Rq(2), Rq(3), Rq(4), Rq(-1)
BL = 0xdf

def idx_of(byte, mode):
    if mode == "lo5": return byte & 31
    if mode == "sh1": return (byte >> 1) & 31
    raise SystemExit
def idx_bits(v, m, mode):
    """bit mask of indices for a literal byte (value v, msk m): all bytes c with c&m == v"""
    out = 0
    if m == 0xff: cs = [v]
    elif m == 0xdf: cs = [v, v | 0x20]
    elif m == 0: return 0xffffffff
    else: cs = [c for c in range(256) if (c & m) == v]
    for c in cs: out |= 1 << int(idx_of(np.uint32(c), mode))
    return out

def run(L, log2e, b2mask, mode, ph_k=2, name="", check=True, e128=True):
    E = 1 << log2e
    P3 = np.zeros(E, np.uint32); P5 = np.zeros(E, np.uint32); PS = np.zeros(E, np.uint32); PH = np.zeros(E, np.uint32)
    def h(x):
        prod = (x * MUL) & 0xffffffff
        return prod >> (32 - log2e), prod
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
                e, prod = h(x)
                nk += 1
                k3 = known[delta + 3]
                i3 = idx_bits(v[delta + 3], m[delta + 3], mode)
                if delta == 0: k5 = known[4]; i5 = idx_bits(v[4], m[4], mode)
                else: k5 = True; i5 = idx_bits(v[0], m[0], mode)
                if k3 and k5 and kn[2]:
                    PH[e] |= 1 << ((prod >> 8) & 31)
                    P3[e] |= i3; P5[e] |= i5
                else:
                    PH[e] |= 1 << ((prod >> 13) & 31)
                    if ph_k >= 2: PH[e] |= 1 << ((prod >> 3) & 31)
                    if delta == 0 and k3: PS[e] |= i3
                    elif delta == 1: PS[e] |= i5
                    else: PS[e] = 0xffffffff
    x = (r2 & BL & b2mask) | (r1 & BL) << 8 | (r0 & BL) << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - log2e)
    p3, p5, ps, ph = P3[e], P5[e], PS[e], PH[e]
    j3, j4, jn = idx_of(r3, mode), idx_of(r4, mode), idx_of(rn, mode)
    t3 = p3 >> j3; t4 = p5 >> j4; tn = p5 >> jn
    ts = (ps >> j3) | (ps >> jn)
    th = ph >> ((prod >> 8) & 31)
    ths = ph >> ((prod >> 13) & 31)
    if ph_k >= 2: ths = ths & (ph >> ((prod >> 3) & 31))
    long_hit = (t3 & (t4 | tn) & th & 1).astype(bool)
    short_hit = (ts & ths & 1).astype(bool)
    hit = long_hit | short_hit
    miss = -1
    if check:
        te = np.nonzero(true_e)[0]
        qi = np.where(te % 2 == 0, te, te - 1) // 2
        miss = int((~hit[qi]).sum())
    lanes = np.zeros(n // 16 + 1, bool); lanes[pos[hit] >> 4] = True
    d = lambda P: np.unpackbits(P.view(np.uint8)).mean()
    print(f"{name} log2e={log2e} b2mask={b2mask:x} idx={mode} keys={nk} dens P3 {d(P3):.3f} P5 {d(P5):.3f} PS {d(PS):.3f} PH {d(PH):.3f}\n"
          f"   cand pos {hit.sum()*16/1e6:.2f}M/GiB (long {long_hit.sum()*16/1e6:.2f} short {short_hit.sum()*16/1e6:.2f}) lanes {lanes.sum()*16/1e6:.2f}M/GiB; missed true {miss}", flush=True)
if __name__ == "__main__":
    for mode in ("lo5", "sh1"):
        run(ALL, 13, 0x1f, mode, name="all")
        run([l for l in ALL if l.len == 3], 13, 0x1f, mode, name="len3", check=False)
        run([l for l in ALL if l.len == 4], 13, 0x1f, mode, name="len4", check=False)
        run([l for l in ALL if l.len >= 5], 13, 0x1f, mode, name="long", check=False)
