"""GPU tests written after round 1's GPU minutes were spent: they have been checked on the CPU
as far as that goes (oracle and compiled reference agree with every expectation used here)
but have not yet run on an MI355X, so they live in a file that sorts last: under `-x` a
surprise here cannot hide the suites that have."""
import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import accel
from hyperscan_amd import corpus as cp
from hyperscan_amd import hwlm as hw
from tests import golden_cases as gc
from tests import oracle_binding as ob
from tests.test_gpu_golden import batch_cases
from tests.util import as_set

pytestmark = pytest.mark.gpu

def test_scratch_regrowth_between_scans():
    """A scratch's buffers grow when a later scan is bigger. The control words rely on being
    left zeroed by the previous scan, so a regrown control block must be cleared even when
    hipMalloc hands the freed range back at the same address (it does, now and then: that case
    once produced garbage match counts). Small scan, then a much bigger one, on fresh
    scratches, several times over, with allocations of other sizes in between."""
    import torch

    lits = cp.teddy_literals()
    big, big_off = cp.packet_corpus(4 << 20, lits, seed=3, match_every=2048)
    small, small_off = cp.packet_corpus(4096, lits, seed=4, match_every=512)
    t = H.hwlm_build(lits)
    orc = ob.Oracle(lits)
    want_big = as_set(orc.collect_blocks(big, big_off))
    want_small = as_set(orc.collect_blocks(small, small_off))
    keep = []
    for trial in range(8):
        s = H.Scratch(0)
        try:
            assert as_set(hw.hwlm_exec_batch(t, s, small, small_off)) == want_small
            keep.append(torch.empty((trial + 1) * 12_345, dtype=torch.uint8, device="cuda:0"))  # perturb the allocator
            assert as_set(hw.hwlm_exec_batch(t, s, big, big_off)) == want_big
            assert as_set(hw.hwlm_exec_batch(t, s, small, small_off)) == want_small
        finally:
            s.close()


def test_flood_with_mask_counts(scratch):
    # unit/internal/fdr_flood.cpp:242-403 FDRFloodp.WithMask, all 256 byte values
    for c in range(256):
        lits, c_alt, _ = gc.flood_mask_literals(c)
        t = H.hwlm_build(lits)
        first, second = gc.flood_mask_expected_counts(c)
        cases = [dict(buf=bytes([c]) * 1024), dict(buf=bytes([c_alt]) * 1024)]
        for got, want in zip(batch_cases(t, scratch, cases), (first, second)):
            cnt = {}
            for _e, i in got:
                cnt[i] = cnt.get(i, 0) + 1
            for i, n in want.items():
                assert cnt.get(i, 0) == n, (c, i)


def test_class_scan_reference_unit_test_vectors():
    """The single-byte golden vectors (tests/golden_accel.py: Vermicelli / RVermicelli / Shufti /
    ReverseShufti / Truffle / ReverseTruffle unit tests) through hsgpu_class_scan_dev: every
    scanned slice is one block of a batch, all slices sharing a class go in one launch. The
    shufti and truffle classes take the round trip through the product's own mask builders and
    decoders (to_shufti -> from_shufti, to_truffle -> from_truffle)."""
    import torch

    from tests import golden_accel as ga
    from tests.test_oracle_accel import expected

    def class_for(kind, params):
        if kind in ("verm", "rverm"):
            return accel.CharClass.from_verm(params[0], params[1], False)
        cls = accel.CharClass(params)
        if kind in ("shufti", "rshufti"):
            lo, hi, _nb = cls.to_shufti()
            return accel.CharClass.from_shufti(lo, hi)
        return accel.CharClass.from_truffle(*cls.to_truffle())

    groups = {}
    for case in ga.cases():
        if case[1] in ("verm", "rverm", "shufti", "rshufti", "truffle", "rtruffle"):
            groups.setdefault((case[1], case[2]), []).append(case)
    assert len(groups) >= 20
    checked = 0
    for (kind, params), cs in groups.items():
        blocks = [c[3][c[4]: len(c[3]) - c[5]] for c in cs]
        corpus = np.frombuffer(b"".join(blocks), dtype=np.uint8)
        off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
        d = torch.from_numpy(corpus.copy()).to("cuda:0")
        d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
        _bm, first, last = accel.class_scan([class_for(kind, params)], d, corpus.size, d_off, len(blocks), True, True)
        got = (last if kind.startswith("r") else first).cpu().numpy().view(np.uint32)[0]
        for b, case in enumerate(cs):
            assert int(got[b]) == (expected(case) & 0xFFFFFFFF), case[0]
            checked += 1
    assert checked > 600


def test_hs_scan_grouped_tails():
    """tails with groups / alternation through the whole public path (GPU literal scan, then the
    host position automaton), against Python's re on the same buffer"""
    import re

    from hyperscan_amd import hs

    pats = [r"GET /(index|home|a+b)\.html?", r"key(=|: ?)(true|false|[0-9]+)", r"BEEF((ab|c)*d){1,3}", r"END(a|)(b|)c"]
    lits = [b"GET /", b"key", b"BEEF", b"END"]
    tails = [p[len(l):] for p, l in zip(pats, lits)]  # (no prefix character needs escaping)
    words = [b"GET /", b"index", b"home", b"aab", b".htm", b"l", b"key", b"=", b": ", b"true", b"77", b"BEEF", b"abd", b"cd", b"d",
             b"END", b"a", b"b", b"c", b" ", b"\n"]
    rng = np.random.default_rng(21)
    data = b"".join(words[int(i)] for i in rng.integers(0, len(words), 20_000))
    db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))
    sc = hs.HsScratch(db)
    got = []
    assert hs.scan(db, data, sc, lambda i, f, t: got.append((i, t)) and False) == hs.HS_SUCCESS
    want = set()
    for i, (tl, lit) in enumerate(zip(tails, lits)):
        tail = re.compile(tl.encode())
        k = data.find(lit)
        while k >= 0:
            s = k + len(lit)
            want |= {(i, to) for to in range(s, min(len(data), s + 64) + 1) if tail.fullmatch(data, s, to)}
            k = data.find(lit, k + 1)
    assert set(got) == want and len(got) == len(want) and len(want) > 200
    assert [t for _i, t in got] == sorted(t for _i, t in got)


def test_collider_subset_on_gpu():
    """the reference's hscollider corpus for the accepted patterns (tests/golden/collider_subset.json,
    9734 corpus lines with the reference's own expected end offsets) through hs_scan_batch: one
    database per pattern, its corpora as the blocks of one batch. CPU form: test_collider_cpu.py."""
    from hyperscan_amd import hs
    from tests.test_collider_cpu import check_ends, compile_case, load_cases

    n, sc = 0, None
    for c in load_cases():
        db, flags = compile_case(c)
        sc = sc or hs.HsScratch(db)  # a scratch serves every database (it grows on demand)
        blocks = [bytes.fromhex(h) for h in c["corpora"]]
        data = np.frombuffer(b"".join(blocks) or b"\0", dtype=np.uint8).copy()
        off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
        got = [[] for _ in blocks]
        assert hs.scan_batch(db, data, off, sc, lambda b, _i, f, t: got[b].append((t, f)) and False) == hs.HS_SUCCESS
        check_ends(c, flags, got)
        n += len(blocks)
    assert n >= 9800


def test_scan_vector_behaviour_cpp():
    """unit/hyperscan/behaviour.cpp:972-1115 Vectored1-5 (6 and 7 need HS_FLAG_ALLOWEMPTY): the
    segments are one logical buffer wherever the empty ones sit; count 0 scans nothing. Plus
    offsets that run through the segments and the mode errors."""
    from hyperscan_amd import hs

    db = hs.Database.compile(["^foo.*bar"], [hs.HS_FLAG_DOTALL], [0], mode=hs.HS_MODE_VECTORED)
    sc = hs.HsScratch(db)
    for segs in ([b"foo", b"   ", b"bar"], [b"", b"foo", b"   ", b"bar"], [b"foo", b"   ", b"", b"bar"], [b"foo", b"   ", b"bar", b""]):
        got = []
        assert hs.scan_vector(db, segs, sc, lambda i, f, t: got.append((i, f, t)) and False) == hs.HS_SUCCESS
        assert got == [(0, 0, 9)]
    got = []
    assert hs.scan_vector(db, [], sc, lambda i, f, t: got.append(t) and False) == hs.HS_SUCCESS and got == []
    db2 = hs.Database.compile(["needle", r"ab\d+"], [0, hs.HS_FLAG_SOM_LEFTMOST], [1, 2], mode=hs.HS_MODE_VECTORED)
    segs = [b"xxnee", b"dle ab", b"1", b"2 need", b"", b"le"]
    got = []
    assert hs.scan_vector(db2, segs, sc, lambda i, f, t: got.append((i, f, t)) and False) == hs.HS_SUCCESS
    assert got == [(1, 0, 8), (2, 9, 12), (2, 9, 13), (1, 0, 20)]
    assert hs.scan(db2, b"needle", sc) == hs.HS_DB_MODE_ERROR
    block_db = hs.Database.compile(["needle"])
    assert hs.scan_vector(block_db, [b"needle"], sc) == hs.HS_DB_MODE_ERROR


def test_hwlm_exec_argument_order_and_callback_context(scratch):
    """hsgpu_hwlm_exec takes hwlmExec's arguments in hwlmExec's order (src/hwlm/hwlm.h:116-118) and
    its callback's third argument is the scratch itself, as the reference passes its hs_scratch
    (HWLMCallback, src/hwlm/hwlm.h:80-99), or the pointer hung on the scratch."""
    import ctypes as C

    from hyperscan_amd import _native

    lib = _native.load_library()
    t = H.hwlm_build([H.HwlmLiteral("needle", False, 7)])
    buf = np.frombuffer(b"..needle..needle", dtype=np.uint8).copy()
    seen = []

    def cb(end, lit_id, ctx):
        seen.append((end, lit_id, ctx))
        return hw.HWLM_ALL_GROUPS

    ccb = _native.HWLM_CB(cb)
    rv = lib.hsgpu_hwlm_exec(t._h, buf.ctypes.data, buf.size, 0, ccb, scratch._h, hw.HWLM_ALL_GROUPS)
    assert rv == 0 and [(e, i) for e, i, _c in seen] == [(7, 7), (15, 7)]
    assert all(c == scratch._h.value for _e, _i, c in seen), "default context: the scratch"
    token = C.c_uint64(0)
    lib.hsgpu_scratch_set_context(scratch._h, C.addressof(token))
    assert lib.hsgpu_scratch_get_context(scratch._h) == C.addressof(token)
    del seen[:]
    rv = lib.hsgpu_hwlm_exec(t._h, buf.ctypes.data, buf.size, 9, ccb, scratch._h, hw.HWLM_ALL_GROUPS)  # start = 9
    assert rv == 0 and [(e, i, c) for e, i, c in seen] == [(15, 7, C.addressof(token))]
    lib.hsgpu_scratch_set_context(scratch._h, scratch._h)  # back to the default for the tests that follow


@pytest.mark.parametrize("env", [{"HSGPU_WG_PER_CU": "4", "HSGPU_WG_THREADS": "256"}, {"HSGPU_WG_PER_CU": "2"},
                                 {"HSGPU_MODE": "fused"}, {"HSGPU_MODE": "unfolded"}])
def test_delivery_order_under_other_geometries(env):
    """Where a share's records go is computed from partial sums over groups of 2^k regions, k chosen from the
    number of regions: other launch geometries (more and smaller workgroups: four times the regions; the fused
    pipeline: a quarter of them in use) must deliver the same records in the same order."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                          "-k", "delivery_order or workload_generators or resident_and_properties or overflow",
                          "-p", "no:cacheprovider"],
                         capture_output=True, text=True, env=dict(os.environ, **env), cwd=root, timeout=900)
    assert out.returncode == 0, out.stdout[-2500:] + out.stderr[-1500:]
    assert " passed" in out.stdout
