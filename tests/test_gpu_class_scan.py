"""Character-class scanning on the GPU against the oracle's scalar restatements
(oracle/hwlm_oracle.c: hso_class_*, hso_shufti_*, hso_truffle_*, hso_verm_*) and,
when shipped, against the reference's own shuftiExec / truffleExec / vermicelliExec."""
import ctypes as C

import numpy as np
import pytest

from hyperscan_amd import accel
from hyperscan_amd import corpus as cp
from tests import oracle_binding as ob
from tests.util import do_accel_block_model

pytestmark = pytest.mark.gpu

CLASSES = {
    "lower": [c for c in range(ord("a"), ord("z") + 1)],
    "upper": [c for c in range(ord("A"), ord("Z") + 1)],
    "digit": [c for c in range(ord("0"), ord("9") + 1)],
    "hex": list(b"0123456789abcdef"),
    "space": list(b" \t\r\n\x0b\x0c"),
    "high": list(range(128, 256)),
    "vowel": list(b"aeiouAEIOU"),
    "nl": [10],
}


def oracle_bitmap(cls, buf):
    L = ob.hso()
    out = np.zeros((buf.size + 7) // 8, dtype=np.uint8)
    L.hso_class_bitmap(cls.bitmap.ctypes.data, buf.ctypes.data, buf.size, out.ctypes.data)
    return out


def test_bitmaps_and_first_last_match_oracle():
    import torch

    corpus, off = cp.line_corpus(3 << 20, seed=5)
    # sprinkle high bytes
    rng = np.random.default_rng(2)
    pos = rng.integers(0, corpus.size, 5000)
    corpus[pos] = rng.integers(128, 256, pos.size)
    classes = [accel.CharClass(m) for m in CLASSES.values()]
    dev = torch.device("cuda", 0)
    d_corpus = torch.from_numpy(corpus).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    nb = off.size - 1
    bitmaps, first, last = accel.class_scan(classes, d_corpus, corpus.size, d_off, nb, True, True)
    bm = bitmaps.cpu().numpy()
    fi = first.cpu().numpy().view(np.uint32)
    la = last.cpu().numpy().view(np.uint32)
    L = ob.hso()
    for ci, cls in enumerate(classes):
        want = oracle_bitmap(cls, corpus)
        assert np.array_equal(bm[ci][: want.size], want), list(CLASSES)[ci]
        for b in rng.integers(0, nb, 300).tolist() + [0, nb - 1]:
            blk = corpus[int(off[b]):int(off[b + 1])]
            f = L.hso_class_fwd(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
            r = L.hso_class_rev(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
            assert fi[ci, b] == f
            assert la[ci, b] == (r & 0xFFFFFFFF)


@pytest.mark.parametrize("total", [1, 15, 16, 17, 16383, 16384, 16385, 100_003])
def test_ragged_sizes(total):
    import torch

    rng = np.random.default_rng(total)
    buf = rng.integers(0, 256, total, dtype=np.uint8)
    cls = accel.CharClass(rng.integers(0, 256, 40).tolist())
    d = torch.from_numpy(buf).to("cuda:0")
    off = np.array([0, total // 2, total // 2, total], dtype=np.uint64)  # includes an empty block
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    bitmaps, first, last = accel.class_scan([cls], d, total, d_off, 3, True, True)
    want = oracle_bitmap(cls, buf)
    assert np.array_equal(bitmaps.cpu().numpy()[0][: want.size], want)
    L = ob.hso()
    for b in range(3):
        blk = np.ascontiguousarray(buf[int(off[b]):int(off[b + 1])])
        f = L.hso_class_fwd(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
        r = L.hso_class_rev(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
        assert first.cpu().numpy().view(np.uint32)[0, b] == f
        assert last.cpu().numpy().view(np.uint32)[0, b] == (r & 0xFFFFFFFF)


def _check_all_blocks(classes, corpus, off):
    import torch

    d = torch.from_numpy(corpus.copy()).to("cuda:0")
    d_off = torch.from_numpy(off.astype(np.uint64).view(np.int64).copy()).to("cuda:0")
    nb = off.size - 1
    bitmaps, first, last = accel.class_scan(classes, d, corpus.size, d_off, nb, True, True)
    fi = first.cpu().numpy().view(np.uint32)
    la = last.cpu().numpy().view(np.uint32)
    L = ob.hso()
    for ci, cls in enumerate(classes):
        want = oracle_bitmap(cls, corpus)
        assert np.array_equal(bitmaps.cpu().numpy()[ci][: want.size], want)
        for b in range(nb):
            blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
            f = L.hso_class_fwd(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
            r = L.hso_class_rev(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
            assert fi[ci, b] == f, (ci, b, int(off[b]), int(off[b + 1]))
            assert la[ci, b] == (r & 0xFFFFFFFF), (ci, b, int(off[b]), int(off[b + 1]))


@pytest.mark.parametrize("shape", ["one_block", "tile_aligned", "tiny", "mixed", "trailing_empty", "sparse_big"])
def test_fused_block_shapes(shape):
    """Blocks against the kernel's 16 KiB tiles: spanning many tiles, ending exactly on a tile
    boundary, thousands of 0/1-byte blocks in one tile, empty blocks at the very end, and a
    class so sparse that most tiles of a long block contribute nothing."""
    rng = np.random.default_rng(len(shape))
    T = 16384
    if shape == "one_block":
        total, off = 5 * T + 123, np.array([0, 5 * T + 123])
    elif shape == "tile_aligned":
        total, off = 4 * T, np.array([0, T, T, 2 * T, 3 * T - 1, 3 * T, 4 * T])
    elif shape == "tiny":
        lens = rng.choice([0, 1, 1, 2], 40000)
        off = np.concatenate([[0], np.cumsum(lens)])
        total = int(off[-1])
    elif shape == "mixed":
        lens = rng.choice([0, 3, 17, 120, 1460, 20000, 40000], 60, p=[.1, .2, .2, .2, .2, .05, .05])
        off = np.concatenate([[0], np.cumsum(lens)])
        total = int(off[-1])
    elif shape == "trailing_empty":
        total, off = 2 * T, np.array([0, 100, 2 * T, 2 * T, 2 * T])
    else:
        total, off = 6 * T, np.array([0, 10, 6 * T - 5, 6 * T])
    corpus = rng.choice(np.frombuffer(b"abcxyz ", dtype=np.uint8), total).astype(np.uint8)
    if shape == "sparse_big":
        corpus[:] = ord("x")
        corpus[[7, 3 * T + 5, 6 * T - 2]] = ord("Q")
    classes = [accel.CharClass(b"abc"), accel.CharClass(b" "), accel.CharClass(b"Q"), accel.CharClass(b"z")]
    _check_all_blocks(classes, corpus, np.asarray(off, dtype=np.uint64))


def test_decoders_match_reference_masks():
    """shufti / truffle masks built by the REFERENCE decode (on our side) to the class
    they were built from, and the GPU first-hit equals shuftiExec / truffleExec."""
    ob.require_ref()
    import torch

    R = ob.href()
    rng = np.random.default_rng(8)
    buf = np.frombuffer(bytes(rng.integers(32, 127, 5000, dtype=np.uint8)), dtype=np.uint8)
    d = torch.from_numpy(buf.copy()).to("cuda:0")
    d_off = torch.from_numpy(np.array([0, buf.size], dtype=np.int64)).to("cuda:0")
    for name, members in CLASSES.items():
        cls = accel.CharClass(members)
        lo, hi = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
        nb = R.hsref_shufti_build(cls.bitmap.ctypes.data, lo, hi)
        m1, m2 = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
        R.hsref_truffle_build(cls.bitmap.ctypes.data, m1, m2)
        assert accel.CharClass.from_truffle(bytes(m1), bytes(m2)).members() == sorted(set(members))
        assert cls.to_truffle() == (bytes(m1), bytes(m2))  # trufflecompile.cpp:59-72
        want_t = R.hsref_truffle_exec(m1, m2, buf.ctypes.data, buf.size)
        _bm, first, last = accel.class_scan([cls], d, buf.size, d_off, 1, True, True)
        assert int(first.cpu().numpy().view(np.uint32)[0, 0]) == want_t
        assert int(last.cpu().numpy()[0, 0]) == R.hsref_rtruffle_exec(m1, m2, buf.ctypes.data, buf.size)
        if nb > 0:
            assert accel.CharClass.from_shufti(bytes(lo), bytes(hi)).members() == sorted(set(members))
            assert want_t == R.hsref_shufti_exec(lo, hi, buf.ctypes.data, buf.size)


def test_vermicelli_semantics():
    import torch

    buf = np.frombuffer(b"xxxxxxxxxxxxxxxxxxxxxxxxxxxQxxxxxqxxxxxxxxxxxxxxxxxxxxxx", dtype=np.uint8)
    d = torch.from_numpy(buf.copy()).to("cuda:0")
    d_off = torch.from_numpy(np.array([0, buf.size], dtype=np.int64)).to("cuda:0")
    L = ob.hso()
    for ch, nocase, negate in ((ord("q"), 0, 0), (ord("Q"), 1, 0), (ord("x"), 0, 1), (ord("z"), 0, 0)):
        cls = accel.CharClass.from_verm(ch, nocase, negate)
        _bm, first, last = accel.class_scan([cls], d, buf.size, d_off, 1, True, True)
        assert int(first.cpu().numpy()[0, 0]) == L.hso_verm_fwd(ch, nocase, negate, buf.ctypes.data, buf.size)
        assert int(last.cpu().numpy()[0, 0]) == L.hso_verm_rev(ch, nocase, negate, buf.ctypes.data, buf.size)
        if ob.ref_available() and not negate:
            R = ob.href()
            assert int(first.cpu().numpy()[0, 0]) == R.hsref_verm_exec(ch, nocase, buf.ctypes.data, buf.size)


# ---- two-byte accelerators (double shufti / double vermicelli) ---------------------------

def _pair_for(kind, params):
    if kind == "dverm" or kind == "rdverm":
        return accel.PairSet.from_dverm(*params)
    if kind == "dverm_masked":
        return accel.PairSet.from_dverm_masked(*params)
    if kind == "dshufti":
        return accel.PairSet.build(params)
    raise AssertionError(kind)


def test_pair_scan_reference_unit_test_vectors():
    """Every two-byte golden vector (tests/golden_accel.py, from unit/internal/{vermicelli,
    rvermicelli,shufti}.cpp): each scanned slice is one block of a batch; all slices that
    share a set are scanned in one launch."""
    import torch

    from tests import golden_accel as ga
    from tests.test_oracle_accel import expected

    groups = {}
    for case in ga.cases():
        name, kind, params, text, lo, hi, _ = case
        if kind in ("dverm", "dverm_masked", "rdverm", "dshufti"):
            groups.setdefault((kind, params), []).append(case)
    assert len(groups) >= 25
    checked = 0
    for (kind, params), cs in groups.items():
        blocks = [c[3][c[4]: len(c[3]) - c[5]] for c in cs]
        corpus = np.frombuffer(b"".join(blocks), dtype=np.uint8)
        off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
        d = torch.from_numpy(corpus.copy()).to("cuda:0")
        d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
        _bm, first, last = accel.pair_scan([_pair_for(kind, params)], d, corpus.size, d_off, len(blocks), True, True)
        got = (last if kind == "rdverm" else first).cpu().numpy().view(np.uint32)[0]
        for b, case in enumerate(cs):
            assert int(got[b]) == (expected(case) & 0xFFFFFFFF), case[0]
            checked += 1
    assert checked > 400


def test_pair_scan_random_sets_match_oracle():
    import torch

    rng = np.random.default_rng(12)
    alpha = np.frombuffer(b"abcdABCD01 \n\x80\xff", dtype=np.uint8)
    total = 300_007
    corpus = rng.choice(alpha, total).astype(np.uint8)
    lens = rng.choice([0, 1, 2, 3, 15, 16, 17, 40, 200, 1500], 3000)
    off = np.concatenate([[0], np.cumsum(lens)])
    off = np.unique(np.concatenate([off[off < total], [total]])).astype(np.uint64)
    off = np.sort(np.concatenate([off, off[5:8]]))  # a few empty blocks
    nb = off.size - 1
    sets = [accel.PairSet.from_dverm(ord("a"), ord("b"), 0), accel.PairSet.from_dverm(ord("A"), ord("B"), 1),
            accel.PairSet.from_dverm_masked(0x30, 0x0A, 0xF0, 0xFF),
            accel.PairSet.build([(ord("a"), ord("a")), (ord("B"), ord("a")), (0x80, 0xFF)]),
            accel.PairSet.build([(ord("0"), ord("1"))], accel.CharClass([0xFF])),
            accel.PairSet.build([(int(rng.choice(alpha)), int(rng.choice(alpha))) for _ in range(4)])]
    d = torch.from_numpy(corpus).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    bitmaps, first, last = accel.pair_scan(sets, d, total, d_off, nb, True, True)
    bm = bitmaps.cpu().numpy()
    fi = first.cpu().numpy().view(np.uint32)
    la = last.cpu().numpy().view(np.uint32)
    L = ob.hso()
    for k, ps in enumerate(sets):
        m = ps.masks
        want = np.zeros((total + 7) // 8, dtype=np.uint8)
        L.hso_dshufti_bitmap(*m, corpus.ctypes.data, total, want.ctypes.data)
        assert np.array_equal(bm[k][: want.size], want), k
        for b in list(range(0, nb, 7)) + [nb - 1]:
            blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
            assert fi[k, b] == L.hso_dshufti_fwd(*m, blk.ctypes.data, blk.size), (k, b)
            assert la[k, b] == (L.hso_dshufti_rev(*m, blk.ctypes.data, blk.size) & 0xFFFFFFFF), (k, b)
    if ob.ref_available():  # never later than the reference's (vector-width dependent) answer allows
        R = ob.href()
        m = sets[0].masks
        for b in range(0, nb, 11):
            blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
            if blk.size >= 32:
                assert R.hsref_dshufti_exec(*m, blk.ctypes.data, blk.size) <= fi[0, b]


@pytest.mark.parametrize("total", [1, 2, 16, 17, 16384, 16385])
def test_pair_scan_tile_edges(total):
    import torch

    buf = np.full(total, ord("a"), dtype=np.uint8)
    ps = accel.PairSet.from_dverm(ord("a"), ord("a"), 0)
    d = torch.from_numpy(buf).to("cuda:0")
    d_off = torch.from_numpy(np.array([0, total], dtype=np.int64)).to("cuda:0")
    bitmaps, first, last = accel.pair_scan([ps], d, total, d_off, 1, True, True)
    bits = np.unpackbits(bitmaps.cpu().numpy()[0], bitorder="little")[:total]
    assert bits[: total - 1].all() and not bits[total - 1]  # the last byte has no successor
    assert int(first.cpu().numpy().view(np.uint32)[0, 0]) == 0
    assert int(last.cpu().numpy().view(np.uint32)[0, 0]) == (total - 1 if total > 1 else 0xFFFFFFFF)


def test_forward_skip_is_do_accel_block_for_a_batch():
    """hsgpu_hwlm_forward_skip_dev = hwlmExec's pre-skip (do_accel_block, src/hwlm/hwlm.c:48-99)
    per block, value for value -- for start 0, a common start > 0 and per-block starts -- and for
    start 0 the skip is SAFE: no literal of the set starts before it (HWLM oracle)."""
    import torch

    import hyperscan_amd as H

    rng = np.random.default_rng(3)
    sets = [[H.HwlmLiteral("needle", False, 0), H.HwlmLiteral("xneed", False, 1)],          # dverm
            [H.HwlmLiteral("Hello", True, 0), H.HwlmLiteral("shell", True, 1)],              # dverm nocase
            [H.HwlmLiteral("ab", False, 0), H.HwlmLiteral("cd", False, 1), H.HwlmLiteral("ef", False, 2)],  # shufti
            [H.HwlmLiteral("qa", False, 0), H.HwlmLiteral("zq", False, 1)]]                  # verm
    L = ob.hso()
    for lits in sets:
        fa = accel.ForwardAccel.choose(lits)
        assert fa.type != accel.ACCEL_NONE
        words = [l.s for l in lits] + [b"hello", b"SHELL", b"zzzz", b"    ", b"abcdef", b"q", b"nee"]
        lens = rng.choice([5, 16, 17, 40, 200, 1000], 300)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total = int(off[-1])
        nb = off.size - 1
        text = b"".join(words[int(i)] + b"." * int(rng.integers(0, 30)) for i in rng.integers(0, len(words), total // 8))
        corpus = np.frombuffer(text[:total].ljust(total, b"."), dtype=np.uint8).copy()
        # a block that ends in the pair's first byte alone, and blocks without any hit
        corpus[int(off[3]):int(off[4])] = ord("~")
        d = torch.from_numpy(corpus).to("cuda:0")
        d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
        orc = ob.Oracle(lits)
        kind, sc = fa.scanner()
        per_block = rng.integers(0, 24, nb).astype(np.int32)
        per_block = np.minimum(per_block, lens.astype(np.int32))
        for start in (0, 7, per_block):
            d_start = start if isinstance(start, int) else torch.from_numpy(start).to("cuda:0")
            skip = accel.forward_skip(fa, d, total, d_off, nb, d_start).cpu().numpy()
            for b in range(nb):
                blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
                s0 = start if isinstance(start, int) else int(start[b])
                if s0 > blk.size:
                    continue
                want = do_accel_block_model(L, kind, sc, fa.offset, blk, s0)
                assert skip[b] == want, (b, blk.size, s0, fa.offset, int(skip[b]), want)
                if s0 == 0:
                    for end, lid in orc.collect(blk):
                        size = len(lits[lid].s)
                        assert end + 1 - size >= skip[b], "a literal starts before the skip"
    # no scheme: every start is kept
    none = accel.ForwardAccel.choose([H.HwlmLiteral(bytes([v, v ^ 0x55]), False, v) for v in range(256)])
    if none.type == accel.ACCEL_NONE:
        skip = accel.forward_skip(none, d, total, d_off, nb, 5).cpu().numpy()
        assert (skip == 5).all()
