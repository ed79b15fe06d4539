"""The two CPU models of the class-sequence patterns agree (Python `re` on reversed blocks vs run lengths):
the GPU kernel is then compared with the run-length model at sizes `re` cannot walk (tests/test_gpu_class_seq.py)."""
import numpy as np

from tests import class_seq_model as M


def test_models_agree_on_random_blocks():
    rng = np.random.default_rng(1)
    alpha = np.frombuffer(b"ab1_ x\n", np.uint8)
    for _ in range(400):
        L = int(rng.integers(0, 40))
        blk = bytes(rng.choice(alpha, L))
        A = rng.choice(alpha, int(rng.integers(1, 4)), replace=False).tolist()
        B = rng.choice(alpha, int(rng.integers(1, 4)), replace=False).tolist()
        m, n = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        want = M.ends_re(blk, A, B, m, n)
        got = M.ends_numpy(np.frombuffer(blk, np.uint8), np.array([0, L], np.uint64), A, B, m, n)
        assert np.array_equal(want, got[:, 1] if L else np.zeros(0, np.int64)), (blk, A, B, m, n)


def test_overlapping_classes_any_split_counts():
    # A = [ab], B = [b]: "aabb" matches ending at 2 (split after "aa") and at 3; with A{3,} the split "aab|b" counts
    blk = b"aabb"
    assert M.ends_re(blk, b"ab", b"b", 2, 1).tolist() == [2, 3]
    assert M.ends_re(blk, b"ab", b"b", 3, 1).tolist() == [3]
    c = np.frombuffer(blk, np.uint8)
    assert M.ends_numpy(c, np.array([0, 4], np.uint64), b"ab", b"b", 3, 1)[:, 1].tolist() == [3]


def test_vectorised_model_equals_the_run_length_model():
    """bench.py gates megabytes of the class-sequence kernel's records against ends_vec: the same statement as ends_numpy
    (itself pinned to Python re above) without a loop over bytes -- ragged and empty blocks, overlapping classes, repeat counts"""
    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"ab1_ x\n", np.uint8)
    for it in range(60):
        nb = int(rng.integers(1, 30))
        lens = rng.integers(0, 50, nb)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        corpus = rng.choice(alpha, int(off[-1])).astype(np.uint8)
        vm = M.VecModel(corpus, off)
        for _ in range(6):
            A = rng.choice(alpha, int(rng.integers(1, 4)), replace=False).tolist()
            B = rng.choice(alpha, int(rng.integers(1, 4)), replace=False).tolist()
            m, n = int(rng.integers(1, 6)), int(rng.integers(1, 4))
            want = M.ends_numpy(corpus, off, A, B, m, n)
            got = vm.ends(A, B, m, n)
            assert np.array_equal(want, got), (it, A, B, m, n)
