"""GPU parity of the class-sequence kernel (csrc/class_seq.hip) against the run-length model and Python `re`:
config 4's final (id, to) -- SURVEY section 8(d): "final (id,to) vs a brute-force regex oracle"."""
import numpy as np
import pytest

from tests import class_seq_model as M

pytestmark = pytest.mark.gpu

POOL = [bytes(range(ord("a"), ord("z") + 1)), bytes(range(ord("A"), ord("Z") + 1)), b"0123456789", b"0123456789abcdef",
        b" \t\r\n\x0b\x0c", bytes(range(128, 256)), b"aeiou", b",.;:",
        b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz_", b"()[]", b"-_", b"\n"]


def _run(corpus, off, seqs, classes, emit=None, cap=1 << 22):
    import torch

    from hyperscan_amd import accel

    dev = torch.device("cuda", 0)
    total, nb = int(corpus.size), int(off.size - 1)
    d_corpus = torch.from_numpy(np.concatenate([corpus, np.zeros(16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.astype(np.uint64).view(np.int64)).to(dev)
    bms = []
    for i in range(0, len(classes), 8):
        bm, _f, _l = accel.class_scan([accel.CharClass(c) for c in classes[i:i + 8]], d_corpus, total, d_off, nb, False, False)
        bms += [bm[k] for k in range(bm.shape[0])]
    emit = emit or (0, total)
    counts, recs, n_emit = accel.class_seq_scan(seqs, bms, total, d_off, nb, emit, cap)
    return counts.cpu().numpy(), recs, n_emit


def _check(corpus, off, seqs, classes, emit=None):
    counts, recs, n_emit = _run(corpus, off, seqs, classes, emit)
    lo, hi = emit or (0, int(corpus.size))
    assert n_emit == len(recs)
    for k, (a, b, m, n, pid) in enumerate(seqs):
        want = M.ends_numpy(corpus, off, classes[a], classes[b], m, n)
        assert counts[k] == len(want), (k, a, b, m, n, int(counts[k]), len(want))
        pos = off[want[:, 0]].astype(np.int64) + want[:, 1] if len(want) else np.zeros(0, np.int64)
        w = want[(pos >= lo) & (pos < hi)]
        g = recs[recs[:, 3] == k]
        assert np.all(g[:, 2] == pid)
        gs = g[np.lexsort((g[:, 1], g[:, 0]))][:, :2].astype(np.int64)
        assert np.array_equal(gs, w), (k, a, b, m, n, len(gs), len(w))


def test_line_corpus_256_patterns():
    from hyperscan_amd import corpus as cp

    corpus, off = cp.line_corpus(1 << 18, seed=5)
    rng = np.random.default_rng(5)
    seqs = []
    for k in range(256):
        a, b = rng.choice(len(POOL), 2, replace=False)
        seqs.append((int(a), int(b), int(rng.integers(3, 9)), 1, 1000 + k))
    _check(corpus, off, seqs, POOL)


def test_python_re_on_a_slice():
    from hyperscan_amd import corpus as cp

    corpus, off = cp.line_corpus(1 << 14, seed=7)
    seqs = [(0, 2, 3, 1, 7), (8, 0, 4, 2, 8), (3, 2, 1, 1, 9)]
    counts, recs, _ = _run(corpus, off, seqs, POOL)
    for k, (a, b, m, n, pid) in enumerate(seqs):
        got = recs[recs[:, 3] == k]
        for blk in range(len(off) - 1):
            want = M.ends_re(bytes(corpus[int(off[blk]):int(off[blk + 1])]), POOL[a], POOL[b], m, n)
            assert np.array_equal(np.sort(got[got[:, 0] == blk][:, 1].astype(np.int64)), want), (k, blk)


def test_ragged_empty_and_long_blocks_repeat_counts_and_emit_range():
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"abcXYZ019 _-\n", np.uint8)
    corpus = rng.choice(alpha, 300000).astype(np.uint8)
    corpus[100000:180000] = rng.choice(np.frombuffer(b"ab1", np.uint8), 80000)  # one long block dense in runs
    cuts = sorted(set(rng.integers(0, 100000, 900).tolist()) | {0, 100000, 180000, 300000}
                  | set(rng.integers(180000, 300000, 1500).tolist()))
    off = np.array(cuts + [300000, 300000], dtype=np.uint64)  # trailing empty blocks
    off = np.sort(np.concatenate([off, off[5:8]]))            # empty blocks in the middle
    classes = [b"abc", b"019", b"abcXYZ_", b"XYZ", b" \n", b"ab1"]
    seqs = [(0, 1, 1, 1, 1), (0, 1, 3, 2, 2), (2, 0, 2, 3, 3), (5, 5, 16, 1, 4), (5, 1, 9, 16, 5), (3, 4, 1, 1, 6),
            (2, 3, 5, 1, 7), (4, 0, 1, 4, 8)]
    _check(corpus, off, seqs, classes)
    _check(corpus, off, seqs, classes, emit=(99990, 100300))


def test_one_block_only_and_tiny():
    corpus = np.frombuffer(b"xxabc123yy", np.uint8).copy()
    off = np.array([0, 10], dtype=np.uint64)
    classes = [b"abc", b"123"]
    counts, recs, n = _run(corpus, off, [(0, 1, 3, 1, 42)], classes)
    assert counts.tolist() == [3] and sorted(recs[:, 1].tolist()) == [5, 6, 7] and set(recs[:, 2].tolist()) == {42}
    off2 = np.array([0, 4, 10], dtype=np.uint64)  # "xxab" | "c123yy": the run of A is cut by the block boundary
    counts, recs, n = _run(corpus, off2, [(0, 1, 3, 1, 42), (0, 1, 1, 1, 43)], classes)
    assert counts.tolist() == [0, 3]


# ---- through the public API: hs_compile routes literal-less class sequences to the kernel --------------------

def _re_events(pats, ids, data):
    """(id, to) of every match end of every pattern, Python re on the reversed data (all-matches semantics)"""
    import re

    rev = data[::-1]
    out = []
    for p, i in zip(pats, ids):
        a, qa, b, qb = re.fullmatch(r"(\[[^\]]+\]|\\[dws]|\.)(\+|\{\d+,\})(\[[^\]]+\]|\\[dws]|\.)(\+|\{\d+,\})", p).groups()
        rp = re.compile(("(?=%s%s%s%s)" % (b, qb.replace("+", "{1,}"), a, qa.replace("+", "{1}").replace(",}", "}"))).encode(), re.S)
        out += [(i, len(data) - mt.start()) for mt in rp.finditer(rev)]
    return sorted(set(out), key=lambda e: (e[1], e[0]))


def test_hs_compile_routes_class_sequences_and_mixes_them_with_literals():
    from hyperscan_amd import hs

    pats = [r"[a-z]{3,}\d+", r"foo[0-9]+", r"\s+[A-Z]{2,}", r"[a-f0-9]{4,}[g-z]+", r"bar"]
    ids = [10, 11, 12, 13, 14]
    db = hs.Database.compile(pats, [hs.HS_FLAG_DOTALL] * 5, ids)
    scratch = hs.HsScratch(db)
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"abcdefgxyz0123456789 ABC\nfoobar", np.uint8)
    data = bytes(rng.choice(alpha, 5000))
    got = []
    assert hs.scan(db, data, scratch, lambda i, f, t: got.append((i, t)) and False) == hs.HS_SUCCESS
    want = _re_events([pats[0], pats[2], pats[3]], [10, 12, 13], data)
    import re

    # foo[0-9]+ and bar by direct search: every end inside the run of digits after "foo"; every "bar"
    for m in re.finditer(rb"foo", data):
        k = m.end()
        while k < len(data) and 48 <= data[k] <= 57:
            k += 1
            want.append((11, k))
    want += [(14, m.start() + 3) for m in re.finditer(rb"(?=bar)", data)]
    assert got == sorted(set(want), key=lambda e: (e[1], e[0]))
    tos = [t for _i, t in got]
    assert tos == sorted(tos)


def test_hs_class_sequences_only_database_batch_singlematch_and_serialisation():
    from hyperscan_amd import hs

    pats = [r"[aeiou]{2,}[^aeiou]+", r"[A-C]+[x-z]{2,}", r".{3,}[\n]+"]
    flags = [0, hs.HS_FLAG_SINGLEMATCH | hs.HS_FLAG_CASELESS, 0]
    db = hs.Database.compile(pats, flags, [1, 2, 3])
    db2 = hs.Database.deserialize(db.serialize())
    rng = np.random.default_rng(4)
    alpha = np.frombuffer(b"aeioubcdxyzABC\n ", np.uint8)
    lens = rng.integers(0, 300, 40)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    data = bytes(rng.choice(alpha, int(off[-1])))
    for d in (db, db2):
        scratch = hs.HsScratch(d)
        got = {}
        hs.scan_batch(d, data, off, scratch, lambda blk, i, f, t: got.setdefault(blk, []).append((i, t)) and False)
        for blk in range(len(lens)):
            chunk = data[int(off[blk]):int(off[blk + 1])]
            want = _re_events([r"[aeiou]{2,}[^aeiou]+"], [1], chunk)
            w2 = _re_events([r"[A-Ca-c]+[x-zX-Z]{2,}"], [2], chunk)[:1]  # SINGLEMATCH: the first end only
            w3 = _re_events([r"[^\n]{3,}[\n]+"], [3], chunk)
            assert got.get(blk, []) == sorted(want + w2 + w3, key=lambda e: (e[1], e[0])), blk
    assert hs.expression_info(r"[a-z]{3,}\d+") == (4, 0xffffffff)


def test_hs_quiet_class_sequence_is_not_evaluated_and_reports_nothing():
    """HS_FLAG_QUIET on a class sequence: no event for it (src/hs.h: "ignore match reporting for this expression"), the
    other patterns unaffected; a database whose class sequences are all quiet scans with its literals alone."""
    from hyperscan_amd import hs

    pats = [r"[a-z]{3,}\d+", r"\s+[A-Z]{2,}", r"bar"]
    rng = np.random.default_rng(8)
    data = bytes(rng.choice(np.frombuffer(b"abcxyz0123 ABC\nbar", np.uint8), 4000))
    import re

    bars = [(3, m.start() + 3) for m in re.finditer(rb"(?=bar)", data)]
    for flags, want in (([hs.HS_FLAG_QUIET, 0, 0], sorted(set(_re_events([pats[1]], [2], data) + bars), key=lambda e: (e[1], e[0]))),
                        ([hs.HS_FLAG_QUIET, hs.HS_FLAG_QUIET, 0], bars)):
        db = hs.Database.compile(pats, flags, [1, 2, 3])
        scratch = hs.HsScratch(db)
        got = []
        assert hs.scan(db, data, scratch, lambda i, f, t: got.append((i, t)) and False) == hs.HS_SUCCESS
        assert got == want
