"""Compile side of the accelerators (host only, no GPU): the forward-accel chooser
(buildForwardAccel, src/rose/rose_build_lit_accel.cpp:372-465) and the shufti mask builder
(shuftiBuildMasks, src/nfa/shufticompile.cpp:54-109) against the compiled reference."""
import ctypes as C

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import accel
from hyperscan_amd.hwlm import pack_literals
from tests import oracle_binding as ob

L = H.HwlmLiteral


def ref_forward(lits, groups):
    R = ob.href()
    arr, _keep = pack_literals(list(lits))
    out = (C.c_uint8 * 160)()
    R.hsref_forward_accel(arr, len(arr), groups, out)
    raw = bytes(out)
    return [dict(type=raw[k], offset=raw[k + 1], c1=raw[k + 2], c2=raw[k + 3], lo=raw[k + 16:k + 32],
                 hi=raw[k + 48:k + 64]) for k in (0, 80)]


def same_scheme(ours, ref):
    assert ours.type == ref["type"], (ours.type, ref)
    if ours.type == accel.ACCEL_NONE:
        return
    assert ours.offset == ref["offset"], (ours.offset, ref)
    if ours.type in (accel.ACCEL_VERM, accel.ACCEL_VERM_NOCASE):
        assert ours.c1 == ref["c1"]
    elif ours.type in (accel.ACCEL_DVERM, accel.ACCEL_DVERM_NOCASE):
        assert (ours.c1, ours.c2) == (ref["c1"], ref["c2"])
    elif ours.type == accel.ACCEL_SHUFTI:  # bucket numbering is free: compare the classes
        assert accel.CharClass.from_shufti(ours.mask_lo, ours.mask_hi).members() == \
            accel.CharClass.from_shufti(ref["lo"], ref["hi"]).members()
    elif ours.type == accel.ACCEL_TRUFFLE:
        assert (ours.mask_lo, ours.mask_hi) == (ref["lo"], ref["hi"])


def random_sets(rng, n_sets):
    alpha = np.frombuffer(b"abcdeABCDE0123_-/ \x00\x80\xff", dtype=np.uint8)
    for k in range(n_sets):
        n = int(rng.integers(1, 12))
        common = bytes(rng.choice(alpha, int(rng.integers(0, 3))))  # shared bytes make (d)verm possible
        lits = []
        for i in range(n):
            body = bytes(rng.choice(alpha, int(rng.integers(1, 9))))
            pos = int(rng.integers(0, len(body) + 1))
            s = (body[:pos] + common + body[pos:])[:8] or b"x"
            kw = {}
            if rng.random() < 0.15:  # a mask that overhangs a short literal
                s = s[:2]
                kw = dict(msk=bytes([0xF0, 0xFF]) + b"\xff" * len(s), cmp=bytes([0x30, int(rng.choice(alpha))]) + s)
            lits.append(L(s, nocase=bool(rng.random() < 0.3) and not kw, id=i, groups=int(rng.choice([1, 2, 3, 0xFF])), **kw))
        yield lits


def test_forward_accel_matches_reference():
    ob.require_ref()
    rng = np.random.default_rng(31)
    kinds = {}
    for lits in random_sets(rng, 600):
        for groups, which in ((1, 0), (0xFFFFFFFFFFFFFFFF, 1), (2, 0)):
            ref = ref_forward(lits, groups)[0]  # accel1 = scheme for `groups`
            if groups == 0xFFFFFFFFFFFFFFFF:
                assert ref == ref_forward(lits, 1)[1]  # accel0 = all groups, whatever was expected
            ours = accel.ForwardAccel.choose(lits, groups)
            same_scheme(ours, ref)
            kinds[ours.type] = kinds.get(ours.type, 0) + 1
    assert set(kinds) >= {accel.ACCEL_NONE, accel.ACCEL_VERM, accel.ACCEL_DVERM, accel.ACCEL_SHUFTI}, kinds
    assert accel.ACCEL_VERM_NOCASE in kinds or accel.ACCEL_DVERM_NOCASE in kinds, kinds


def test_forward_accel_known_cases():
    fa = accel.ForwardAccel.choose([L("foobar", False, 0), L("xfoo", False, 1)])
    # pairs both contain: "fo" (distinct bytes, offsets 0 / 1) beats "oo" (equal bytes)
    assert fa.type == accel.ACCEL_DVERM and fa.offset == 1 and (fa.c1, fa.c2) == (ord("f"), ord("o"))
    fa = accel.ForwardAccel.choose([L("Hello", True, 0), L("help", True, 1)])
    # "HE" (offset 0 in both) beats "EL" (offset 1): smaller offset wins among equals
    assert fa.type == accel.ACCEL_DVERM_NOCASE and (fa.c1, fa.c2) == (ord("H"), ord("E")) and fa.offset == 0
    fa = accel.ForwardAccel.choose([L("ab", False, 0), L("cd", False, 1), L("ef", False, 2)])
    assert fa.type == accel.ACCEL_SHUFTI and fa.offset == 0
    assert accel.CharClass.from_shufti(fa.mask_lo, fa.mask_hi).members() == sorted(b"ace")
    assert accel.ForwardAccel.choose([L("ab", False, 0, groups=2)], expected_groups=1).type == accel.ACCEL_NONE
    kind, cls = accel.ForwardAccel.choose([L("ab", False, 0), L("cd", False, 1), L("ef", False, 2)]).scanner()
    assert kind == "class" and cls.members() == sorted(b"ace")


def test_shufti_mask_builder_matches_reference():
    rng = np.random.default_rng(8)
    U = C.c_uint8 * 16
    for trial in range(300):
        k = int(rng.integers(1, 40))
        members = sorted(set(rng.choice(256 if trial % 2 else 96, k).tolist()))
        cls = accel.CharClass(members)
        got = cls.to_shufti()
        if got is not None:
            lo, hi, nb = got
            assert accel.CharClass.from_shufti(lo, hi).members() == members and 1 <= nb <= 8
        if ob.ref_available():
            rlo, rhi = U(), U()
            rnb = ob.href().hsref_shufti_build(cls.bitmap.ctypes.data, rlo, rhi)
            assert (got is None) == (rnb == -1)
            if got is not None:
                assert rnb == got[2]
    assert accel.CharClass(range(8)).to_shufti() is not None  # "always able to construct masks for 8 or fewer characters"


def test_do_accel_block_model_is_the_reference_function():
    """The expectation used for hsgpu_hwlm_forward_skip_dev on the GPU (tests/util.do_accel_block_model: the oracle's
    accelerators inside a restatement of do_accel_block) against the reference's OWN static do_accel_block
    (src/hwlm/hwlm.c:80-99, exported by oracle/ref_build/accel_block_shim.c which includes hwlm.c in place): every
    scheme, buffers of 0 .. 400 bytes, every kind of `start` (0, inside, len - 16, len - 15, len)."""
    from tests.util import do_accel_block_model

    R = ob.href()
    if not hasattr(R, "hsref_do_accel_block"):
        pytest.skip("oracle/_ref predates the do_accel_block shim")
    Lo = ob.hso()
    rng = np.random.default_rng(23)
    sets = [[L("needle", False, 0), L("xneed", False, 1)],                       # dverm
            [L("Hello", True, 0), L("shell", True, 1)],                          # dverm nocase
            [L("ab", False, 0), L("cd", False, 1), L("ef", False, 2)],            # shufti
            [L("qa", False, 0), L("zq", False, 1)],                              # verm
            [L("Q", True, 0)],                                                   # verm nocase
            [L(bytes([0x80 + i, 0x41]), False, i) for i in range(12)]]           # high bytes: truffle or shufti
    seen = set()
    for lits in sets:
        fa = accel.ForwardAccel.choose(lits)
        assert fa.type != accel.ACCEL_NONE
        seen.add(fa.type)
        arr, _keep = pack_literals(list(lits))
        out = (C.c_uint8 * 160)()
        R.hsref_forward_accel(arr, len(arr), 0xFFFFFFFFFFFFFFFF, out)
        img = (C.c_uint8 * 80).from_buffer_copy(bytes(out)[80:160])  # accel0: all groups
        assert img[0] == fa.type and img[1] == fa.offset
        kind, sc = fa.scanner()
        words = [l.s for l in lits] + [b"hello", b"SHELL", b"....", b"q", b"nee", b"\x80", b"A"]
        for n in list(range(0, 40)) + [63, 64, 65, 100, 257, 400]:
            for trial in range(6):
                text = b"".join(words[int(i)] + b"." * int(rng.integers(0, 12)) for i in rng.integers(0, len(words), n // 3 + 1))
                blk = np.frombuffer(text[:n].ljust(n, b"~"), dtype=np.uint8).copy()
                if trial == 0 and n:
                    blk[:] = ord("~")            # nothing to find
                if trial == 1 and n:
                    blk[-1] = lits[0].s[0]       # the last byte alone: double vermicelli's partial match
                for start in sorted(x for x in {0, 1, n // 2, max(0, n - 17), max(0, n - 16), max(0, n - 15), n} if x <= n):  # (hwlmExec: start <= len)
                    want = R.hsref_do_accel_block(img, blk.ctypes.data, n, start)
                    got = do_accel_block_model(Lo, kind, sc, fa.offset, blk, start)
                    assert got == want, (fa.type, n, start, trial, got, want)
    assert {accel.ACCEL_DVERM, accel.ACCEL_DVERM_NOCASE, accel.ACCEL_SHUFTI, accel.ACCEL_VERM, accel.ACCEL_VERM_NOCASE} <= seen
