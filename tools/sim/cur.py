import sys, time, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
import base
from base import Lit, MUL
N = 64 << 20
lits, corpus, off = base.load(N)
true_e = base.true_ends(lits, corpus, off)
L = [Lit(l) for l in lits]
pad = base.padded(corpus)
n = corpus.size
def B(k, pos):  # byte at q-k
    return pad[8 + pos - k]
def report(name, hit, pos, stride):
    npos = hit.sum()
    lanes = np.zeros(n // 16 + 1, dtype=bool); lanes[pos[hit] >> 4] = True
    # recall check: every true end must be covered by a hit at lookup position e (or e-1/e+1 for stride 2)
    print(f"{name}: cand positions {npos} ({npos/ n*100:.3f}% of bytes) -> {npos*16/1e6:.2f}M/GiB ; cand lanes {lanes.sum()*16/1e6:.2f}M/GiB")
    return lanes

def variants(val, unk):
    sub = 0
    while True:
        yield val | sub
        sub = (sub - unk) & unk
        if sub == 0: break

def cur_design():
    k = 15
    filt = np.zeros(1 << k, dtype=np.uint32)
    nA = nB = 0
    for li in L:
        known4 = li.len >= 4
        # blind key bytes
        v = [li.val[p] & 0xdf for p in range(4)]; m = [li.msk[p] & 0xdf for p in range(4)]
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff
        a = prod >> (30 - k)
        if known4:
            nA += 1
            filt[a >> 2] |= np.uint32(1 << ((a + v[3]) & 31)) | np.uint32(1 << (((prod >> 8) + v[3]) & 31))
        else:
            nB += 1
            filt[a >> 2] = 0xffffffff
    print("keys A", nA, "B", nB, "bits set", sum(bin(int(w)).count('1') for w in filt))
    pos = np.arange(n)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    a = prod >> np.uint32(30 - k)
    w = filt[a >> 2]
    hit = ((w >> ((a + b3) & 31)) & (w >> (((prod >> 8) + b3) & 31)) & 1).astype(bool)
    assert hit[true_e].all(), "recall"
    report("current S1 K2 BFOLD", hit, pos, 1)
    return hit
if __name__ == "__main__":
    t = time.time(); cur_design(); print(time.time() - t)
