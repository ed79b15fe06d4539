import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from hyperscan_amd import corpus as cp

MUL = 0x9E3779
def is_alpha(c): return (65 <= c <= 90) or (97 <= c <= 122)

def load(nbytes=64 << 20, cache="/tmp/sim/fdr10k_%d.npz"):
    path = cache % nbytes
    lits, _ = cp.snort_like_literals(10000, seed=4)
    if os.path.exists(path):
        z = np.load(path)
        return lits, z["corpus"], z["off"]
    corpus, off = cp.packet_corpus(nbytes, lits, seed=10)
    np.savez(path, corpus=corpus, off=off)
    return lits, corpus, off

class Lit:
    """bytes at distance p from the end: val[p], msk[p] (p=0 last byte); msk 0 = outside/unknown"""
    def __init__(self, l):
        s = l.s
        self.len = len(s)
        self.val = [0] * 9; self.msk = [0] * 9
        for p in range(min(8, len(s))):
            c = s[len(s) - 1 - p]
            if l.nocase and is_alpha(c):
                self.val[p] = c & 0xdf; self.msk[p] = 0xdf
            else:
                self.val[p] = c; self.msk[p] = 0xff

def padded(corpus, front=8, back=8):
    pad = np.zeros(corpus.size + front + back, dtype=np.uint32)
    pad[front:front + corpus.size] = corpus
    return pad

def true_match_ends(lits, corpus):
    """set of end positions with at least one literal matching (ignoring block boundaries): via 8-byte window compare
    grouped by (msk) -- slow path: use hashing of last 3 bytes blind to prefilter"""
    L = [Lit(l) for l in lits]
    n = corpus.size
    pad = padded(corpus)
    ends = np.zeros(n, dtype=bool)
    # group literals by length (<=8) and case pattern is too many; do per-literal on candidate positions of its last 2 bytes
    b0 = pad[8:8 + n]; b1 = pad[7:7 + n]
    key2 = (b0 & 0xdf) | ((b1 & 0xdf) << 8)
    order = np.argsort(key2, kind="stable")
    sk = key2[order]
    for li in L:
        k = (li.val[0] & 0xdf) | ((li.val[1] & 0xdf) << 8)
        lo, hi = np.searchsorted(sk, k), np.searchsorted(sk, k, side="right")
        pos = order[lo:hi]
        ok = np.ones(pos.size, dtype=bool)
        for p in range(min(li.len, 8)):
            ok &= (pad[8 + pos - p] & li.msk[p]) == li.val[p]
        ends[pos[ok]] = True
    return ends
