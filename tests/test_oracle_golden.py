"""Pins the CPU checkers (not gpu):
  * oracle/hwlm_oracle.c against the reference's golden vectors, and
  * against the reference itself (oracle/_ref, compiled from /root/reference)
    on seeded random inputs, including every FDR/Teddy engine the reference can
    be forced into (as unit/internal/fdr.cpp parameterises its tests)."""
import numpy as np
import pytest

import hyperscan_amd as H
from tests import golden_cases as gc
from tests import oracle_binding as ob
from tests.util import random_corpus, random_literals

needs_ref = pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libhsref.so not built")


def run_case(engine, case):
    got = engine.collect(case["buf"])
    if case["ordered"]:
        assert got == case["expect"], case["name"]
    else:
        assert sorted(got) == sorted(case["expect"]), case["name"]


@pytest.mark.parametrize("brute", [False, True])
def test_oracle_simple_and_norepeat(brute):
    for case in gc.simple_cases():
        run_case(ob.Oracle(case["lits"], brute=brute), case)


def test_oracle_multi_location():
    o = ob.Oracle(gc.multi_location_cases()[0]["lits"])
    for case in gc.multi_location_cases():
        run_case(o, case)


def test_oracle_align_and_too_early():
    for case in gc.align_too_early_cases(alignments=range(0, 32, 5)):
        run_case(ob.Oracle(case["lits"]), case)


@pytest.mark.parametrize("alphabet", gc.SHORT_ALPHABETS)
def test_oracle_short_writings(alphabet):
    bufs, groups = gc.short_writings(alphabet)
    for g in groups[::3]:
        o = ob.Oracle([H.HwlmLiteral(p, False, i) for p, i in g])
        for buf in bufs[::7]:
            assert sorted(o.collect(buf)) == gc.naive_matches(buf, g)


@pytest.mark.parametrize("c", [0, 0x20, ord("a"), ord("A"), ord("z"), ord("5"), 0x7F, 0x80, 0xDF, 0xFF])
def test_oracle_flood(c):
    lits, c_alt, _ = gc.flood_literals(c)
    o = ob.Oracle(lits)
    first, second = gc.flood_expected_counts(c)
    for byte, want in ((c, first), (c_alt, second)):
        got = {}
        for _e, i in o.collect(bytes([byte]) * 1024):
            got[i] = got.get(i, 0) + 1
        for i, n in want.items():
            assert got.get(i, 0) == n, (c, byte, i)


@pytest.mark.parametrize("c", [0, 0x20, ord("a"), ord("A"), ord("e"), ord("Z"), ord("5"), 0x5B, 0x7F, 0x80, 0xDF, 0xFF])
def test_oracle_flood_with_mask(c):
    # unit/internal/fdr_flood.cpp:242-403 FDRFloodp.WithMask
    lits, c_alt, _ = gc.flood_mask_literals(c)
    o = ob.Oracle(lits)
    first, second = gc.flood_mask_expected_counts(c)
    for byte, want in ((c, first), (c_alt, second)):
        got = {}
        for _e, i in o.collect(bytes([byte]) * 1024):
            got[i] = got.get(i, 0) + 1
        for i, n in want.items():
            assert got.get(i, 0) == n, (c, byte, i)


def test_oracle_noodle_cases():
    for case in gc.noodle_cases():
        run_case(ob.Oracle(case["lits"]), case)


def test_oracle_termination_and_groups():
    # unit/internal/fdr.cpp:697-744 FDRTermB: callback returning 0 => TERMINATED after one match
    lits = [H.HwlmLiteral("f", False, 0), H.HwlmLiteral("ff", False, 1)]
    o = ob.Oracle(lits)
    seen = []
    rv = o.exec(b"f" * 17, 0, lambda e, i: (seen.append((e, i)), 0)[1])
    assert rv == 1 and len(seen) == 1
    # groups == 0 => no scan (hwlm.c:178); disjoint groups => literal skipped
    lits = [H.HwlmLiteral("ab", False, 0, groups=1), H.HwlmLiteral("b", False, 1, groups=2)]
    o = ob.Oracle(lits)
    assert o.collect(b"abab", groups=0) == []
    # the callback's return value is the live group mask (fdr_confirm_runtime.h:91-96)
    out = []
    o.exec(b"abab", 0, lambda e, i: (out.append((e, i)), 2)[1], groups=2)
    assert out == [(1, 1), (3, 1)]
    out = []
    o.exec(b"abab", 0, lambda e, i: (out.append((e, i)), 1 if i == 1 else 3)[1], groups=2)
    assert out == [(1, 1), (3, 0), (3, 1)]  # after id 1 only group 1 is live; id 0 then re-enables both


def test_oracle_msk_cmp():
    # supplementary mask: "bc" preceded by a digit: msk/cmp over 3 bytes (hwlm_literal.h:88-104)
    lit = H.HwlmLiteral("bc", False, 7, msk=b"\xf0\xff\xff", cmp=b"\x30bc")
    o = ob.Oracle([lit])
    assert o.collect(b"abc1bc9bcxbc") == [(5, 7), (8, 7)]
    # mask longer than the literal: match may not overhang the start of the block
    assert o.collect(b"bc1bc") == [(4, 7)]


@needs_ref
def test_reference_golden_all_engines():
    engines = ob.valid_engines()
    assert 0 in engines
    for hint in [ob.NO_HINT] + engines:
        for case in gc.simple_cases():
            if hint == ob.NO_HINT and any(l.noruns for l in case["lits"]):
                continue  # noruns is advisory (hwlm_literal.h:66-71): Noodle ignores it, FDR quashes
            try:
                r = ob.Reference(case["lits"], hint=hint)
            except ValueError:
                continue  # Teddy engines may refuse a set (CHECK_WITH_TEDDY_OK_TO_FAIL)
            run_case(r, case)


@needs_ref
@pytest.mark.parametrize("nlits,lo,hi,seed", [(1, 1, 8, 1), (5, 1, 8, 2), (40, 3, 8, 3), (64, 4, 8, 4), (97, 2, 8, 5),
                                              (700, 3, 8, 6), (10000, 4, 8, 7)])
def test_oracle_equals_reference_random(nlits, lo, hi, seed):
    rng = np.random.default_rng(seed)
    lits = random_literals(rng, nlits, lo, hi)
    corpus = random_corpus(rng, 200_000, lits, plant_every=700)
    want = sorted(ob.Reference(lits).collect(corpus))
    got = sorted(ob.Oracle(lits).collect(corpus))
    assert got == want


@needs_ref
def test_oracle_equals_reference_forced_engines():
    rng = np.random.default_rng(11)
    lits = random_literals(rng, 24, 2, 8)
    corpus = random_corpus(rng, 50_000, lits, plant_every=300)
    want = sorted(ob.Oracle(lits).collect(corpus))
    ran = 0
    for hint in ob.valid_engines():
        try:
            r = ob.Reference(lits, hint=hint)
        except ValueError:
            continue
        assert sorted(r.collect(corpus)) == want, hint
        ran += 1
    assert ran >= 3


@needs_ref
def test_flood_with_mask_reference_agrees():
    for c in (ord("a"), ord("A"), 0x00, 0xFF, ord("5"), 0x7B):
        lits, c_alt, _ = gc.flood_mask_literals(c)
        first, second = gc.flood_mask_expected_counts(c)
        for byte, want in ((c, first), (c_alt, second)):
            buf = bytes([byte]) * 1024
            ref = sorted(ob.Reference(lits).collect(buf))
            assert ref == sorted(ob.Oracle(lits).collect(buf))
            got = {}
            for _e, i in ref:
                got[i] = got.get(i, 0) + 1
            for i, n in want.items():  # the reference meets its own test's expectations as restated
                assert got.get(i, 0) == n, (c, byte, i)


@needs_ref
def test_flood_reference_agrees():
    for c in (ord("a"), 0x00, 0xFF):
        lits, c_alt, _ = gc.flood_literals(c)
        for byte in (c, c_alt):
            buf = bytes([byte]) * 1024
            assert sorted(ob.Reference(lits).collect(buf)) == sorted(ob.Oracle(lits).collect(buf))
