"""Pins the accelerator half of the oracle (oracle/hwlm_oracle.c: shufti / truffle /
vermicelli and their two-byte forms) without a GPU:
  * against the golden vectors of the reference's own unit tests (tests/golden_accel.py,
    restated from unit/internal/{shufti,vermicelli,rvermicelli}.cpp with file:line), and
  * against the reference itself (oracle/_ref, compiled from /root/reference) on seeded
    random classes, pair sets and buffers at every 16-byte alignment,
and checks the product library's host-side mask builders (no device needed)."""
import ctypes as C

import numpy as np
import pytest

from hyperscan_amd import accel
from tests import golden_accel as ga
from tests import oracle_binding as ob

U8x16 = C.c_uint8 * 16
REF_VECTOR = 32  # oracle/ref_build/Makefile builds the reference with -mavx2: 32-byte shufti vectors


def aligned_copy(data, align):
    """a uint8 array holding `data` whose first byte sits at address = align (mod 64)"""
    raw = np.zeros(len(data) + 128, dtype=np.uint8)
    base = raw.ctypes.data
    shift = (align - base) % 64
    view = raw[shift: shift + len(data)]
    view[:] = np.frombuffer(bytes(data), dtype=np.uint8)
    assert view.ctypes.data % 64 == align % 64
    return raw, view


def oracle_eval(kind, params, buf):
    L = ob.hso()
    p, n = buf.ctypes.data, buf.size
    if kind == "verm":
        return L.hso_verm_fwd(params[0], params[1], 0, p, n)
    if kind == "nverm":
        return L.hso_verm_fwd(params[0], params[1], 1, p, n)
    if kind == "rverm":
        return L.hso_verm_rev(params[0], params[1], 0, p, n)
    if kind == "dverm":
        return L.hso_dverm_fwd(params[0], params[1], params[2], p, n)
    if kind == "dverm_masked":
        return L.hso_dverm_masked_fwd(params[0], params[1], params[2], params[3], p, n)
    if kind == "rdverm":
        return L.hso_rdverm(params[0], params[1], params[2], p, n)
    if kind == "dshufti":
        ps = accel.PairSet.build(params)
        lo1, hi1, lo2, hi2 = ps.masks
        return L.hso_dshufti_fwd(lo1, hi1, lo2, hi2, p, n)
    if kind in ("shufti", "rshufti"):
        lo, hi, _nb = accel.CharClass(params).to_shufti()  # the product's shuftiBuildMasks
        return (L.hso_shufti_fwd if kind == "shufti" else L.hso_shufti_rev)(lo, hi, p, n)
    if kind in ("truffle", "rtruffle"):
        m1, m2 = accel.CharClass(params).to_truffle()
        return (L.hso_truffle_fwd if kind == "truffle" else L.hso_truffle_rev)(m1, m2, p, n)
    raise AssertionError(kind)


def ref_eval(kind, params, buf):
    R = ob.href()
    p, n = buf.ctypes.data, buf.size
    if kind == "verm":
        return R.hsref_verm_exec(params[0], params[1], p, n)
    if kind == "rverm":
        return R.hsref_rverm_exec(params[0], params[1], p, n)
    if kind == "dverm":
        return R.hsref_dverm_exec(params[0], params[1], params[2], p, n)
    if kind == "dverm_masked":
        return R.hsref_dverm_masked_exec(params[0], params[1], params[2], params[3], p, n)
    if kind == "rdverm":
        return R.hsref_rdverm_exec(params[0], params[1], params[2], p, n)
    if kind == "dshufti":
        lo1, hi1, lo2, hi2 = ref_dshufti_masks(params)
        return R.hsref_dshufti_exec(lo1, hi1, lo2, hi2, p, n)
    if kind in ("shufti", "rshufti", "truffle", "rtruffle"):
        cls = accel.CharClass(params)
        a, b = U8x16(), U8x16()
        if kind.endswith("shufti"):
            assert R.hsref_shufti_build(cls.bitmap.ctypes.data, a, b) > 0
            return (R.hsref_shufti_exec if kind == "shufti" else R.hsref_rshufti_exec)(a, b, p, n)
        R.hsref_truffle_build(cls.bitmap.ctypes.data, a, b)
        return (R.hsref_truffle_exec if kind == "truffle" else R.hsref_rtruffle_exec)(a, b, p, n)
    raise AssertionError(kind)


def ref_dshufti_masks(pairs, onechar=None):
    R = ob.href()
    flat = bytes(b for p in pairs for b in p)
    oc = (onechar if onechar is not None else accel.CharClass()).bitmap
    m = [U8x16() for _ in range(4)]
    ok = R.hsref_dshufti_build(oc.ctypes.data, flat, len(flat) // 2, *m)
    return tuple(bytes(x) for x in m) if ok else None


def expected(case):
    name, kind, params, text, lo, hi, want_abs = case
    n = len(text) - lo - hi
    if want_abs is None:
        return -1 if kind in ("rverm", "rdverm", "rshufti", "rtruffle") else n
    return want_abs - lo


def test_oracle_on_reference_unit_test_vectors():
    cases = ga.cases()
    assert len(cases) > 1200
    for case in cases:
        name, kind, params, text, lo, hi, _ = case
        _raw, buf = aligned_copy(text[lo: len(text) - hi], 0)
        assert oracle_eval(kind, params, buf) == expected(case), name


def test_reference_on_its_own_unit_test_vectors():
    """the compiled reference reproduces the expectations as restated (guards the restating)"""
    ob.require_ref()
    for case in ga.cases():
        name, kind, params, text, lo, hi, _ = case
        # the tests scan t1 + i of ONE array: keep the alignment relation (t1 16-aligned here)
        _raw, whole = aligned_copy(text, 0)
        buf = whole[lo: len(text) - hi]
        assert ref_eval(kind, params, buf) == expected(case), name


def random_pairs(rng, n):
    alpha = np.frombuffer(b"abcdABCD01 \x80\xff", dtype=np.uint8)
    return sorted({(int(rng.choice(alpha)), int(rng.choice(alpha))) for _ in range(n)})


def test_pair_builder_is_exact_and_matches_reference_buckets():
    """hsgpu_pair_build accepts exactly the requested sequences (all 65536 pairs decoded), and
    succeeds/fails like shuftiBuildDoubleMasks (shufticompile.cpp:135-209)."""
    rng = np.random.default_rng(41)
    for trial in range(40):
        pairs = random_pairs(rng, int(rng.integers(1, 12)))
        onechar = accel.CharClass(rng.choice(np.arange(256), int(rng.integers(0, 3)), replace=False).tolist()) \
            if trial % 3 == 0 else None
        want = set(pairs) | ({(a, b) for a in onechar.members() for b in range(256)} if onechar else set())
        try:
            ps = accel.PairSet.build(pairs, onechar)
        except accel.HsgpuError as e:
            assert e.code == -4
            ps = None
        if ob.ref_available():
            assert (ref_dshufti_masks(pairs, onechar) is not None) == (ps is not None)
        if ps is None:
            continue
        got = {(a, b) for a in range(256) for b in range(256) if ps.test(a, b)}
        assert got == want
        if ob.ref_available():  # same accepted set as the reference's masks (bucket order may differ)
            rm = accel.PairSet.from_dshufti(*ref_dshufti_masks(pairs, onechar))
            assert all(rm.test(a, b) == ((a, b) in want) for a in range(0, 256, 3) for b in range(256))
    # nine sequences no two of which share a nibble in any position need nine buckets
    with pytest.raises(accel.HsgpuError):
        accel.PairSet.build([(0x10 * k + k, 0x10 * k + k) for k in range(9)])


def test_dverm_builders_decode_to_the_vermicelli_predicates():
    for c1, c2, nocase in ((ord("a"), ord("b"), 0), (ord("A"), ord("B"), 1), (ord("1"), ord("Z"), 1), (0, 255, 0)):
        ps = accel.PairSet.from_dverm(c1, c2, nocase)
        al = lambda c: chr(c).isalpha() and c < 128
        m1, m2 = (0xDF if nocase and al(c1) else 0xFF), (0xDF if nocase and al(c2) else 0xFF)
        for a in range(256):
            for b in range(0, 256, 5):
                assert ps.test(a, b) == ((a & m1) == (c1 & m1) and (b & m2) == (c2 & m2))
    for c1, c2, m1, m2 in ((0x41, 0x42, 0xDF, 0xDF), (0x30, 0x0A, 0xF0, 0xFF), (0x00, 0x80, 0x01, 0x80)):
        ps = accel.PairSet.from_dverm_masked(c1, c2, m1, m2)
        for a in range(256):
            for b in range(0, 256, 3):
                assert ps.test(a, b) == ((a & m1) == c1 and (b & m2) == c2)


def test_double_shufti_vector_model_equals_reference():
    """oracle/hwlm_oracle.c models the reference's 16-byte-vector early exits exactly
    (hso_dshufti_ref_model) and its exact form is never earlier-than-true / later-than-reference."""
    ob.require_ref()
    L, R = ob.hso(), ob.href()
    rng = np.random.default_rng(17)
    checked = early = 0
    sets = [tuple((ord(a), ord(b)) for a, b in ps) for ps, _ in ga.DSHUFTI_EDGE + ga.DSHUFTI_NOMATCH]
    sets += [tuple(random_pairs(rng, 3)) for _ in range(6)]
    for pairs in sets:
        masks = ref_dshufti_masks(pairs)
        if masks is None:
            continue
        for trial in range(30):
            n = int(rng.integers(16, 200))
            data = rng.choice(np.frombuffer(b"abcdABCDbbbbbeeeeeV01 \x80\xff", dtype=np.uint8), n)
            for align in (0, 1, 7, 15, 16, 31, 33):
                _raw, buf = aligned_copy(data, align)
                ref = R.hsref_dshufti_exec(*masks, buf.ctypes.data, n)
                model = L.hso_dshufti_ref_model(*masks, buf.ctypes.data, n, align, REF_VECTOR)
                exact = L.hso_dshufti_fwd(*masks, buf.ctypes.data, n)
                assert ref == model, (pairs, n, align)
                assert ref <= exact
                if ref < exact:  # only ever a first-byte hit
                    ps = accel.PairSet.from_dshufti(*masks)
                    assert any(ps.test(int(buf[ref]), b) for b in range(256))
                    early += 1
                checked += 1
    assert checked > 500 and early > 0
    # the reference's own expectations for the artefact cases (unit/internal/shufti.cpp:545-602)
    for ps, text in ga.DSHUFTI_EDGE:
        masks = ref_dshufti_masks(tuple((ord(a), ord(b)) for a, b in ps))
        for i in range(16):
            _raw, whole = aligned_copy(text.encode(), 0)
            buf = whole[i:]
            assert R.hsref_dshufti_exec(*masks, buf.ctypes.data, buf.size) == 15
            assert L.hso_dshufti_ref_model(*masks, buf.ctypes.data, buf.size, i, REF_VECTOR) == 15
            assert L.hso_dshufti_fwd(*masks, buf.ctypes.data, buf.size) == buf.size - 1  # partial match at the end
    for ps, text in ga.DSHUFTI_NOMATCH:
        masks = ref_dshufti_masks(tuple((ord(a), ord(b)) for a, b in ps))
        for i in range(16):
            _raw, whole = aligned_copy(text.encode(), 0)
            buf = whole[i:]
            assert L.hso_dshufti_fwd(*masks, buf.ctypes.data, buf.size) == buf.size


def test_oracle_equals_reference_on_random_accel_inputs():
    ob.require_ref()
    L, R = ob.hso(), ob.href()
    rng = np.random.default_rng(23)
    alpha = np.frombuffer(b"abABzZ09 \n\x00\x7f\x80\xff", dtype=np.uint8)
    for trial in range(300):
        n = int(rng.integers(16, 300))
        data = rng.choice(alpha, n)
        _raw, buf = aligned_copy(data, int(rng.integers(0, 16)))
        p = buf.ctypes.data
        c1, c2 = int(rng.choice(alpha)), int(rng.choice(alpha))
        al = lambda c: chr(c).isalpha() and c < 128
        nocase = int(al(c1) and al(c2) and rng.random() < 0.5)
        if nocase:
            c1, c2 = c1 & 0xDF, c2 & 0xDF  # "nocase already uppercase", vermicelli.h:47
        assert L.hso_verm_fwd(c1, nocase, 0, p, n) == R.hsref_verm_exec(c1, nocase, p, n)
        assert L.hso_verm_fwd(c1, nocase, 1, p, n) == R.hsref_nverm_exec(c1, nocase, p, n)
        assert L.hso_verm_rev(c1, nocase, 0, p, n) == R.hsref_rverm_exec(c1, nocase, p, n)
        assert L.hso_dverm_fwd(c1, c2, nocase, p, n) == R.hsref_dverm_exec(c1, c2, nocase, p, n)
        # reverse double: the reference leaves the head below its last 16-byte boundary unexamined
        rref = R.hsref_rdverm_exec(c1, c2, nocase, p, n)
        assert rref == L.hso_rdverm_ref_model(c1, c2, nocase, p, n, buf.ctypes.data % 16)
        exact = L.hso_rdverm(c1, c2, nocase, p, n)
        assert rref >= exact and (rref == exact or exact < 32)
        m1, m2 = int(rng.choice([0xFF, 0xDF, 0xF0, 0x0F])), int(rng.choice([0xFF, 0xDF, 0x80]))
        assert L.hso_dverm_masked_fwd(c1 & m1, c2 & m2, m1, m2, p, n) == \
            R.hsref_dverm_masked_exec(c1 & m1, c2 & m2, m1, m2, p, n)
        cls = accel.CharClass(rng.choice(alpha, int(rng.integers(1, 6))).tolist())
        lo, hi, t1, t2 = U8x16(), U8x16(), U8x16(), U8x16()
        R.hsref_truffle_build(cls.bitmap.ctypes.data, t1, t2)
        assert L.hso_truffle_fwd(t1, t2, p, n) == R.hsref_truffle_exec(t1, t2, p, n)
        assert L.hso_truffle_rev(t1, t2, p, n) == R.hsref_rtruffle_exec(t1, t2, p, n)
        assert L.hso_class_fwd(cls.bitmap.ctypes.data, p, n) == R.hsref_truffle_exec(t1, t2, p, n)
        if R.hsref_shufti_build(cls.bitmap.ctypes.data, lo, hi) > 0:
            assert L.hso_shufti_fwd(lo, hi, p, n) == R.hsref_shufti_exec(lo, hi, p, n)
            assert L.hso_shufti_rev(lo, hi, p, n) == R.hsref_rshufti_exec(lo, hi, p, n)
