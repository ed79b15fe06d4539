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
