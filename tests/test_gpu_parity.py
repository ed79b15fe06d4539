"""GPU parity: the HIP path, called through the C ABI, against the oracle on the
same seeded inputs. Bit-exact: identical multisets of (block, end, id)."""
import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import hwlm as hw
from tests import oracle_binding as ob
from tests.util import as_set, random_blocks, random_corpus, random_literals

pytestmark = pytest.mark.gpu


def gpu_collect(table, scratch, buf, start=0, groups=H.HWLM_ALL_GROUPS):
    out = []

    def cb(end, lit_id, _ctx):
        out.append((end, lit_id))
        return H.HWLM_CONTINUE_MATCHING

    rv = H.hwlm_exec(table, buf, start, cb, scratch, groups)
    assert rv == H.HWLM_SUCCESS
    return out


def test_simple_golden(scratch):
    # unit/internal/fdr.cpp:167-190 (FDRp.Simple): ends 5, 23, 83
    data = b"mnopqrabcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ12345678901234567890mnopqr\0"
    t = H.hwlm_build([H.HwlmLiteral("mnopqr", False, 0)])
    assert gpu_collect(t, scratch, data) == [(5, 0), (23, 0), (83, 0)]


def test_simple_single_golden(scratch):
    # unit/internal/fdr.cpp:192-216 (FDRp.SimpleSingle): ends 0, 18, 78, 80
    data = b"mnopqrabcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ12345678901234567890m0m\0"
    t = H.hwlm_build([H.HwlmLiteral("m", False, 0)])
    assert gpu_collect(t, scratch, data) == [(0, 0), (18, 0), (78, 0), (80, 0)]


@pytest.mark.parametrize("nlits,lo,hi,seed", [(1, 4, 8, 1), (8, 1, 8, 2), (64, 4, 8, 3), (64, 1, 3, 4),
                                              (500, 3, 8, 5), (3000, 1, 8, 6)])
def test_random_sets_single_block(scratch, nlits, lo, hi, seed):
    rng = np.random.default_rng(seed)
    lits = random_literals(rng, nlits, lo, hi)
    corpus = random_corpus(rng, 300_000 + seed * 7919, lits, plant_every=512)
    t = H.hwlm_build(lits)
    got = sorted(gpu_collect(t, scratch, corpus))
    want = sorted(ob.Oracle(lits).collect(corpus))
    assert len(got) == len(want)
    assert got == want


@pytest.mark.parametrize("nlits,seed", [(16, 11), (64, 12), (2000, 13)])
def test_random_sets_batched_blocks(scratch, nlits, seed):
    rng = np.random.default_rng(seed)
    lits = random_literals(rng, nlits, 1, 8)
    corpus = random_corpus(rng, 1_000_000, lits, plant_every=300)
    off = random_blocks(rng, corpus.size, mean_len=400)
    t = H.hwlm_build(lits)
    got = hw.hwlm_exec_batch(t, scratch, corpus, off)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    assert as_set(got) == as_set(want)
    # records arrive sorted by (block, end)
    key = got["block"].astype(np.uint64) << np.uint64(32) | got["end"].astype(np.uint64)
    assert np.all(key[1:] >= key[:-1])


def test_smoke_entry():
    import __graft_entry__ as ge

    ge.smoke()
