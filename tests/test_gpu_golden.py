"""GPU engine against the reference's golden vectors (through the C ABI), in
every engine configuration the build flags can force -- the analogue of
unit/internal/fdr.cpp running each case on every valid FDR/Teddy engine id."""
import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import hwlm as hw
from tests import golden_cases as gc
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

FORCE_REPL, FORCE_HASHED, FORCE_K2, FORCE_K1, FORCE_S1, FORCE_BLIND, FORCE_S2 = 1, 2, 4, 8, 16, 32, 64
ENGINES = {
    "auto": 0,
    "repl_k1_s1": FORCE_REPL | FORCE_K1 | FORCE_S1,
    "repl_k2_s2": FORCE_REPL | FORCE_K2 | FORCE_S2,
    "hash_k1_s1": FORCE_HASHED | FORCE_K1 | FORCE_S1,
    "hash_k2_s2_blind": FORCE_HASHED | FORCE_K2 | FORCE_S2 | FORCE_BLIND,
    "hash_k2_s1_blind": FORCE_HASHED | FORCE_K2 | FORCE_S1 | FORCE_BLIND,
}


def exec_collect(table, scratch, buf, start=0, groups=H.HWLM_ALL_GROUPS):
    out = []

    def cb(end, lit_id, _ctx):
        out.append((end, lit_id))
        return H.HWLM_CONTINUE_MATCHING

    assert H.hwlm_exec(table, buf, start, cb, scratch, groups) == H.HWLM_SUCCESS
    return out


def batch_cases(table, scratch, cases):
    """Scan many small golden buffers as ONE batch of blocks (also exercises block
    boundaries: neighbours must never leak matches into each other)."""
    bufs = [c["buf"] for c in cases]
    off = np.concatenate([[0], np.cumsum([len(b) for b in bufs])]).astype(np.uint64)
    corpus = np.frombuffer(b"".join(bufs), dtype=np.uint8) if off[-1] else np.zeros(0, np.uint8)
    recs = hw.hwlm_exec_batch(table, scratch, corpus, off)
    per = [[] for _ in cases]
    for b, e, i in zip(recs["block"].tolist(), recs["end"].tolist(), recs["id"].tolist()):
        per[b].append((e, i))
    return per


@pytest.mark.parametrize("engine", ENGINES)
def test_simple_and_norepeat(scratch, engine):
    for case in gc.simple_cases():
        t = H.hwlm_build(case["lits"], ENGINES[engine])
        assert exec_collect(t, scratch, case["buf"]) == case["expect"], case["name"]


@pytest.mark.parametrize("engine", ENGINES)
def test_multi_location(scratch, engine):
    cases = gc.multi_location_cases()
    t = H.hwlm_build(cases[0]["lits"], ENGINES[engine])
    for got, case in zip(batch_cases(t, scratch, cases), cases):
        assert got == case["expect"], case["name"]


@pytest.mark.parametrize("engine", ["auto", "repl_k2_s2", "hash_k1_s1", "hash_k2_s2_blind"])
def test_align_and_too_early(scratch, engine):
    cases = gc.align_too_early_cases()
    # group by literal so each group is one table + one batched scan
    groups = {}
    for c in cases:
        groups.setdefault(c["lits"][0].s, []).append(c)
    for lit, cs in groups.items():
        t = H.hwlm_build(cs[0]["lits"], ENGINES[engine])
        for got, case in zip(batch_cases(t, scratch, cs), cs):
            assert sorted(got) == case["expect"], case["name"]


@pytest.mark.parametrize("engine", ["auto", "repl_k1_s1", "hash_k2_s1_blind"])
@pytest.mark.parametrize("alphabet", gc.SHORT_ALPHABETS)
def test_short_writings(scratch, engine, alphabet):
    bufs, groups = gc.short_writings(alphabet)
    cases = [dict(buf=b) for b in bufs]
    for g in groups:
        t = H.hwlm_build([H.HwlmLiteral(p, False, i) for p, i in g], ENGINES[engine])
        for got, buf in zip(batch_cases(t, scratch, cases), bufs):
            assert sorted(got) == gc.naive_matches(buf, g)


@pytest.mark.parametrize("engine", ["auto", "hash_k2_s1_blind"])
def test_flood_counts(scratch, engine):
    # unit/internal/fdr_flood.cpp:148-240, all 256 byte values
    for c in range(256):
        lits, c_alt, _ = gc.flood_literals(c)
        t = H.hwlm_build(lits, ENGINES[engine])
        first, second = gc.flood_expected_counts(c)
        cases = [dict(buf=bytes([c]) * 1024), dict(buf=bytes([c_alt]) * 1024)]
        for got, want in zip(batch_cases(t, scratch, cases), (first, second)):
            cnt = {}
            for _e, i in got:
                cnt[i] = cnt.get(i, 0) + 1
            for i, n in want.items():
                assert cnt.get(i, 0) == n, (c, i)


def test_noodle_cases(scratch):
    cases = [c for c in gc.noodle_cases() if len(c["buf"])]
    by_lit = {}
    for c in cases:
        by_lit.setdefault((c["lits"][0].s, c["lits"][0].nocase), []).append(c)
    for _k, cs in by_lit.items():
        t = H.hwlm_build(cs[0]["lits"])
        for got, case in zip(batch_cases(t, scratch, cs), cs):
            assert sorted(got) == case["expect"], case["name"]


def test_termination(scratch):
    # unit/internal/fdr.cpp:725-744 FDRTermB
    t = H.hwlm_build([H.HwlmLiteral("f", False, 0), H.HwlmLiteral("ff", False, 1)])
    seen = []
    rv = H.hwlm_exec(t, b"f" * 17, 0, lambda e, i, _c: (seen.append((e, i)), H.HWLM_TERMINATE_MATCHING)[1], scratch)
    assert rv == H.HWLM_TERMINATED and len(seen) == 1


def test_groups_and_start(scratch):
    lits = [H.HwlmLiteral("ab", False, 0, groups=1), H.HwlmLiteral("b", False, 1, groups=2)]
    t = H.hwlm_build(lits)
    o = ob.Oracle(lits)
    assert exec_collect(t, scratch, b"abab", groups=0) == []
    out = []
    H.hwlm_exec(t, b"abab", 0, lambda e, i, _c: (out.append((e, i)), 2)[1], scratch, 2)
    assert out == [(1, 1), (3, 1)]
    # start: matches must begin at or after it (hwlm.h:108-111)
    lits = [H.HwlmLiteral("abc", False, 0), H.HwlmLiteral("c", False, 1)]
    t = H.hwlm_build(lits)
    o = ob.Oracle(lits)
    buf = b"abcabcabc"
    for start in range(0, 9):
        assert sorted(exec_collect(t, scratch, buf, start)) == sorted(o.collect(buf, start)), start


def test_msk_cmp(scratch):
    lits = [H.HwlmLiteral("bc", False, 7, msk=b"\xf0\xff\xff", cmp=b"\x30bc"),
            H.HwlmLiteral("xyz", True, 8, msk=b"\xff\x00\x00\x00", cmp=b"Q\x00\x00\x00")]
    t = H.hwlm_build(lits)
    buf = b"bc1bcabc2bc.QxYz.qxyz.QXYZxyz"
    assert sorted(exec_collect(t, scratch, buf)) == sorted(ob.Oracle(lits).collect(buf))
    assert len(exec_collect(t, scratch, buf)) == 4
