import sys, time, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
import base
from base import Lit
N = 64 << 20
lits, corpus, off = base.load(N)
L = [Lit(l) for l in lits]
pad = base.padded(corpus); n = corpus.size
pos = np.arange(0, n, 2)
def Bq(k): return pad[8 + pos - k] & 0xdf
b = {k: Bq(k) for k in range(-1, 6)}
def exact(nbytes, minlen):
    # keys: blind bytes; delta0 uses p=0..nbytes-1 ; delta+1 uses p=1..nbytes (window q-nbytes+1..q = literal p=1..nbytes)
    k0, k1 = set(), set()
    for li in L:
        if li.len < minlen: continue
        if li.len >= nbytes: k0.add(tuple(li.val[p] & 0xdf for p in range(nbytes)))
        if li.len >= nbytes + 1: k1.add(tuple(li.val[p] & 0xdf for p in range(1, nbytes + 1)))
    def pack(t): 
        v = 0
        for i, c in enumerate(t): v |= c << (8 * i)
        return v
    keys = np.array(sorted({pack(t) for t in k0 | k1}), dtype=np.uint64)
    w = np.zeros(pos.size, dtype=np.uint64)
    for i in range(nbytes): w |= b[i].astype(np.uint64) << np.uint64(8 * i)
    hit = np.isin(w, keys)
    print(f"exact {nbytes}-byte window keys (len>={minlen}): {keys.size} keys, candidates {hit.sum()*16/1e6:.2f}M/GiB")
exact(4, 5); exact(5, 6); exact(3, 4); exact(4, 4)
