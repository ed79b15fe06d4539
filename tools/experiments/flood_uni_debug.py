import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import hyperscan_amd as H
from hyperscan_amd import hwlm as hw
from tests import oracle_binding as ob
from tests.util import random_corpus, random_literals
short, start = True, 0
rng = np.random.default_rng(7 + short)
lits = []
for c in b"abz":
    lits += [H.HwlmLiteral(bytes([c]) * 4, id=len(lits)), H.HwlmLiteral(bytes([c]) * 8, id=len(lits) + 1),
             H.HwlmLiteral(b"x" + bytes([c]) * 4, id=len(lits) + 2), H.HwlmLiteral(bytes([c]) * 5, nocase=True, id=len(lits) + 3)]
if short:
    lits += [H.HwlmLiteral(b"aaa", id=len(lits)), H.HwlmLiteral(b"zz", id=len(lits) + 1)]
lits += [H.HwlmLiteral(l.s, nocase=l.nocase, id=len(lits) + i) for i, l in enumerate(random_literals(rng, 60, 4, 8))]
total = 3 << 20
corpus = random_corpus(rng, total, lits, plant_every=3000)
cuts = {0, total}
pos = 1000
for k in range(40):
    ln = int(rng.integers(3 << 10, 200 << 10))
    if pos + ln + 5000 > total:
        break
    v = b"abzAq"[k % 5]
    corpus[pos:pos + ln] = v
    if k % 3 == 0:
        corpus[pos - 1] = ord("x")
    for _ in range(int(rng.integers(0, 4))):
        cuts.add(pos + int(rng.integers(1, ln)))
    cuts.add(pos + int(rng.integers(-20, 20)))
    pos += ln + int(rng.integers(100, 30000))
off = np.array(sorted(cuts), dtype=np.uint64)
t = H.hwlm_build(lits)
print("table flags", hex(t.info()["flags"]))
s = H.Scratch(0)
want = ob.Oracle(lits).collect_blocks(corpus, off, start=start)
W = set(zip(want["block"].tolist(), want["end"].tolist(), want["id"].tolist()))
for rep in range(3):
    got = hw.hwlm_exec_batch(t, s, corpus, off, start=start)
    G = list(zip(got["block"].tolist(), got["end"].tolist(), got["id"].tolist()))
    Gs = set(G)
    extra = sorted(Gs - W); missing = sorted(W - Gs)
    print("rep", rep, "got", len(G), "distinct", len(Gs), "want", len(want), "extra", len(extra), "missing", len(missing))
    print("  extra head", extra[:8], "tail", extra[-4:])
    print("  missing head", missing[:8], "tail", missing[-4:])
    if extra:
        eb = np.array([e[0] for e in extra]); print("  extra by block", np.unique(eb, return_counts=True))
    if missing:
        mb = np.array([e[0] for e in missing]); print("  missing by block", np.unique(mb, return_counts=True))
        b = missing[0][0]; print("  block", b, "spans", int(off[b]), int(off[b+1]), "first missing abs pos", int(off[b]) + missing[0][1], "bytes", bytes(corpus[int(off[b]) + missing[0][1] - 9: int(off[b]) + missing[0][1] + 2]))
