import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from hyperscan_amd import corpus as cp

MUL = 0x9E3779
def is_alpha(c): return (65 <= c <= 90) or (97 <= c <= 122)

def true_ends(lits, corpus, off, cache="/tmp/sim/true_ends_%d.npy"):
    """bool per corpus byte: some literal ends there (the compiled reference, oracle/_ref)"""
    path = cache % corpus.size
    if os.path.exists(path):
        return np.load(path)
    from tests import oracle_binding as ob
    ref = ob.Reference(lits, variant=ob.ref_variants()[-1])
    r = ref.collect_blocks(corpus, off)
    e = np.zeros(corpus.size, dtype=bool)
    e[off[r["block"]].astype(np.int64) + r["end"].astype(np.int64)] = True
    np.save(path, e)
    return e


def load(nbytes=64 << 20, cache="/tmp/sim/fdr10k_%d.npz"):
    os.makedirs("/tmp/sim", exist_ok=True)
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
