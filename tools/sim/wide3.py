"""HSGPU_F_WIDE plus a third test on the byte before b3 (b4), in the hi word beside the pure-hash bit: keys shorter than 5
bytes cannot know b4 and set every bit of their hi word. Candidate lanes per GiB against the two-test layout."""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/sim'); sys.path.insert(0, '/root/repo')
from cur import *
def run(third, k=14):
    LO = np.zeros(1 << k, np.uint32); HI = np.zeros(1 << k, np.uint32); H2 = np.zeros(1 << k, np.uint32)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(5)]
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff
        e = prod >> (32 - k)
        HI[e] |= np.uint32(1 << (prod & 31))
        LO[e] |= np.uint32(1 << (v[3] & 31)) if li.len >= 4 else np.uint32(0xffffffff)
        if third == "same":   HI[e] |= np.uint32(1 << (v[4] & 31)) if li.len >= 5 else np.uint32(0xffffffff)
        if third == "own":    H2[e] |= np.uint32(1 << (v[4] & 31)) if li.len >= 5 else np.uint32(0xffffffff)
    pos = np.arange(n)
    b0, b1, b2, b3, b4 = (B(i, pos) & 0xdf for i in range(5))
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - k)
    hit = (LO[e] >> (b3 & 31)) & (HI[e] >> (prod & 31))
    if third == "same": hit = hit & (HI[e] >> (b4 & 31))
    if third == "own": hit = hit & (H2[e] >> (b4 & 31))
    hit = (hit & 1).astype(bool)
    assert hit[true_e].all()
    report(f"wide third={third}", hit, pos, 1)
for t in (None, "same", "own"): run(t)
def run4(k=13, fourth=None):
    """128-bit entries {lo: b3, hi: pure hash, w2: b4, w3: fourth}"""
    LO = np.zeros(1 << k, np.uint32); HI = np.zeros(1 << k, np.uint32); W2 = np.zeros(1 << k, np.uint32); W3 = np.zeros(1 << k, np.uint32)
    for li in L:
        v = [li.val[p] & 0xdf for p in range(6)]
        x = v[2] | v[1] << 8 | v[0] << 16
        prod = (x * MUL) & 0xffffffff
        e = prod >> (32 - k)
        HI[e] |= np.uint32(1 << (prod & 31))
        LO[e] |= np.uint32(1 << (v[3] & 31)) if li.len >= 4 else np.uint32(0xffffffff)
        W2[e] |= np.uint32(1 << (v[4] & 31)) if li.len >= 5 else np.uint32(0xffffffff)
        if fourth == "b5": W3[e] |= np.uint32(1 << (v[5] & 31)) if li.len >= 6 else np.uint32(0xffffffff)
        if fourth == "h2": W3[e] |= np.uint32(1 << ((prod >> 5) & 31))
    pos = np.arange(n)
    b0, b1, b2, b3, b4, b5 = (B(i, pos) & 0xdf for i in range(6))
    x = b2 | b1 << 8 | b0 << 16
    prod = ((x.astype(np.uint64) * MUL) & 0xffffffff).astype(np.uint32)
    e = prod >> np.uint32(32 - k)
    hit = (LO[e] >> (b3 & 31)) & (HI[e] >> (prod & 31)) & (W2[e] >> (b4 & 31))
    if fourth == "b5": hit = hit & (W3[e] >> (b5 & 31))
    if fourth == "h2": hit = hit & (W3[e] >> ((prod >> 5) & 31))
    hit = (hit & 1).astype(bool)
    assert hit[true_e].all()
    report(f"quad-entry k={k} fourth={fourth}", hit, pos, 1)
for f in (None, "b5", "h2"): run4(13, f)
