"""stride-1 K1 / K2 filter (blind, folded 3-byte keys) followed by the in-filter key gate on each lane's first candidate:
candidate lanes before and after (tools/sim; what profiles/r03_infilter_gate_ab.txt measured for K2 on the GPU)"""
import sys, numpy as np
sys.path.insert(0,'/root/repo/tools/sim'); sys.path.insert(0,'/root/repo')
from cur import *
HT_MUL = 0x9E3779B1
SALT = 0xB5000000
def gate_bit(key): return ((key.astype(np.uint64) * HT_MUL) & 0xffffffff) >> 16
def run(k2):
    k = 15
    filt = np.zeros(1 << k, dtype=np.uint32)
    gate = np.zeros(65536, dtype=bool)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(4)]
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff
        a = prod >> (30 - k)
        if li.len >= 4:
            filt[a >> 2] |= np.uint32(1 << ((a + v[3]) & 31))
            if k2: filt[a >> 2] |= np.uint32(1 << (((prod >> 8) + v[3]) & 31))
            key = v[3] | v[2] << 8 | v[1] << 16 | v[0] << 24
            gate[int(gate_bit(np.array([key], dtype=np.uint64))[0])] = True
        else:
            filt[a >> 2] = 0xffffffff
            key = (v[2] | v[1] << 8 | v[0] << 16) | SALT
            gate[int(gate_bit(np.array([key], dtype=np.uint64))[0])] = True
    pos = np.arange(n)
    b0, b1, b2, b3 = (B(i, pos) & 0xdf for i in range(4))
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    a = prod >> np.uint32(30 - k)
    w = filt[a >> 2]
    hit = (w >> ((a + b3) & 31))
    if k2: hit = hit & (w >> (((prod >> 8) + b3) & 31))
    hit = (hit & 1).astype(bool)
    assert hit[true_e].all()
    w4 = (b3 | b2 << 8 | b1 << 16 | b0 << 24).astype(np.uint64)
    gpass = gate[gate_bit(w4)] | gate[gate_bit(((w4 >> 8) | SALT))]
    lanes = np.zeros(n // 16 + 1, dtype=np.int32)
    np.add.at(lanes, pos[hit] >> 4, 1)
    # first candidate of each lane
    hp = pos[hit]
    first = np.ones(hp.size, dtype=bool); first[1:] = (hp[1:] >> 4) != (hp[:-1] >> 4)
    surv_first = gpass[hp] & first
    drop = np.zeros(n // 16 + 1, dtype=np.int32)
    np.add.at(drop, hp[first & ~gpass[hp]] >> 4, 1)
    after = lanes - drop
    print(f"k2={k2}: cand lanes {np.count_nonzero(lanes)*16/1e6:.2f}M/GiB positions {hit.sum()*16/1e6:.2f}M; after first-bit gate: lanes {np.count_nonzero(after)*16/1e6:.2f}M/GiB; gate density {gate.mean():.3f}; full gate (every bit): {(hit & gpass).sum()*16/1e6:.2f}M positions")
run(True); run(False)
