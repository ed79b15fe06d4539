"""The bench's config-5 gate model (tests/rose_model.py) pinned to Python `re` by brute force, and -- through the HWLM
oracle's hits -- to the facade's own host confirm (hs_confirm_batch: no device needed)."""
import ctypes as C
import re

import numpy as np

from hyperscan_amd import hs
from hyperscan_amd.hwlm import HwlmLiteral
from tests import oracle_binding as ob
from tests import rose_model as RM


def _corpus(rng, lits, nblocks=60):
    words = [b"abc7", b"  key=", b"....END", b"END", b"\n", b" ", b"x", b"q9", b"word_12=", b"ENDEND", b"zz", b"\t\tab=", b"a1"] + lits
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(0, 40)))) for _ in range(nblocks)]
    off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
    return np.frombuffer(b"".join(blocks), dtype=np.uint8).copy(), off


def test_tail_ends_equal_re_bruteforce():
    rng = np.random.default_rng(3)
    alpha = b"ab z9_=\n\tENDx "
    for _ in range(300):
        blk = bytes(rng.choice(np.frombuffer(alpha, np.uint8), int(rng.integers(0, 30))))
        for kind, tail in enumerate(RM.TAILS):
            tre = re.compile(tail.encode())
            for s in range(len(blk) + 1):
                want = [to for to in range(s, len(blk) + 1) if tre.fullmatch(blk, s, to)]
                assert RM.tail_ends(blk, s, kind) == want, (blk, s, tail)


def test_model_equals_hs_confirm_batch_over_oracle_hits():
    rng = np.random.default_rng(4)
    lits = sorted({bytes(rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8), int(rng.integers(6, 13))))
                   for _ in range(40)})
    pats = [l.decode() + RM.TAILS[i % 3] for i, l in enumerate(lits)]
    db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))
    corpus, off = _corpus(rng, lits)
    keyed = db.literals()
    assert [k[0] for k in keyed] == [l[-8:] for l in lits]
    hl = [HwlmLiteral(k[0], k[1], i) for i, k in enumerate(keyed)]
    hits = ob.Oracle(hl).collect_blocks(corpus, off)
    want = RM.expected_events(corpus, off, lits, hits)
    assert len(want) > 20
    order = np.lexsort((hits["id"], hits["end"], hits["block"]))
    recs = np.zeros((len(order), 4), dtype=np.uint32)
    recs[:, 0], recs[:, 1], recs[:, 2], recs[:, 3] = hits["block"][order], hits["end"][order], hits["id"][order], hits["id"][order]
    got = []
    cb = hs.BATCH_CB(lambda b, i, f, t, _fl, _c: (got.append((int(b), int(i), int(t))), 0)[1])
    lib = hs._lib()
    lib.hs_confirm_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, hs.BATCH_CB, C.c_void_p]
    rv = lib.hs_confirm_batch(db._h, corpus.ctypes.data, off.ctypes.data, off.size - 1, recs.ctypes.data, len(recs), cb, None)
    assert rv == 0
    assert sorted(got) == want
