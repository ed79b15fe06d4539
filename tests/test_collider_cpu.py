"""The reference's hscollider regression corpus (tools/hscollider/test_cases) for the patterns the
hs_* facade accepts: tests/golden/collider_subset.json, made by tools/make_collider_fixture.py.
9734 of its corpus lines carry the end offsets the reference itself must report (the `id="corpus":
to, ...` lines); the rest come from the Python model that agrees with all of those.

CPU form: literal hits from the HWLM oracle for the literals each database is keyed on, then the
facade's host confirm. The GPU form of the same check is in test_zz_gpu_late_additions.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import hs
from hyperscan_amd.hwlm import MATCH_DTYPE
from tests import oracle_binding as ob

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "collider_subset.json")
FLAGS = {"i": hs.HS_FLAG_CASELESS, "s": hs.HS_FLAG_DOTALL, "m": hs.HS_FLAG_MULTILINE, "H": hs.HS_FLAG_SINGLEMATCH,
         "L": hs.HS_FLAG_SOM_LEFTMOST, "V": hs.HS_FLAG_ALLOWEMPTY, "8": hs.HS_FLAG_UTF8, "P": hs.HS_FLAG_PREFILTER}


def load_cases():
    return json.load(open(FIXTURE))["cases"]


def compile_case(c):
    flags = 0
    for ch in c["flags"]:
        flags |= FLAGS[ch]
    ext = hs.ExprExt.make(**c["ext"]) if c["ext"] else None
    return hs.Database.compile_ext([c["pattern"]], [flags], [c["id"]], [ext]), flags


def check_ends(c, flags, got_by_corpus):
    """the hscollider verdicts: identical end-offset sets (main.cpp:572-590); under SINGLEMATCH
    exactly one of the expected matches, none if none are expected (main.cpp:522-534) -- and here
    the earliest one, which is what this implementation promises"""
    for k, (ends, got) in enumerate(zip(c["ends"], got_by_corpus)):
        tos = sorted(t for t, _f in got)
        if flags & hs.HS_FLAG_SINGLEMATCH:
            assert tos == ends[:1], (c["id"], c["pattern"], c["flags"], bytes.fromhex(c["corpora"][k]), ends, tos)
        else:
            assert tos == ends, (c["id"], c["pattern"], c["flags"], bytes.fromhex(c["corpora"][k]), ends, tos)
        assert tos == [t for t, _f in got], "delivery must be in offset order"
        if flags & hs.HS_FLAG_SOM_LEFTMOST:
            assert all(f < t for t, f in got)
        else:
            assert all(f == 0 for _t, f in got)


def cpu_events(db, blocks):
    """every block through oracle literal hits + hs_confirm_batch -> [[(to, from)]] per block"""
    lits = [H.HwlmLiteral(b, nocase=nc, id=i) for i, (b, nc, _rid) in enumerate(db.literals())]
    corpus = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy() if any(blocks) else np.zeros(1, np.uint8)
    off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
    got = ob.Oracle(lits).collect_blocks(corpus, off)
    recs = np.zeros(len(got), dtype=MATCH_DTYPE)
    recs["block"], recs["end"], recs["id"], recs["lit"] = got["block"], got["end"], got["id"], got["id"]
    recs = np.ascontiguousarray(recs[np.lexsort((recs["id"], recs["end"], recs["block"]))])
    lib = hs._lib()
    lib.hs_confirm_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, hs.BATCH_CB,
                                     C.c_void_p]
    out = [[] for _ in blocks]

    def on(b, _i, f, t, _fl, _c):
        out[b].append((t, f))
        return 0

    cb = hs.BATCH_CB(on)
    rv = lib.hs_confirm_batch(db._h, corpus.ctypes.data, off.ctypes.data, len(blocks), recs.ctypes.data, recs.size, cb, None)
    assert rv == hs.HS_SUCCESS
    return out


def test_fixture_shape():
    cases = load_cases()
    kinds = [k for c in cases for k in c["kind"]]
    assert len(cases) >= 1150 and kinds.count("reference") >= 9700
    assert len({c["file"] for c in cases}) >= 15  # spread over the corpus's files


def test_collider_subset_on_cpu():
    cases = load_cases()
    n = 0
    for c in cases:
        db, flags = compile_case(c)
        blocks = [bytes.fromhex(h) for h in c["corpora"]]
        check_ends(c, flags, cpu_events(db, blocks))
        n += len(blocks)
    assert n >= 9800
