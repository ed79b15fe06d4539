"""Round 6: what the advisor and the guard-page tests found."""
import ctypes as C

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import hwlm as hw
from tests.util import random_corpus, random_literals

pytestmark = pytest.mark.gpu


def test_pipelined_batch_refuses_bad_offsets_before_reading_or_delivering(scratch):
    """advisor, round 5: hsgpu_hwlm_exec_batch_cb checked the offsets chunk by chunk beside the copies -- off = [0, 3 GiB, 50]
    made chunk 0 one 'valid' 3 GiB block that was copied from the caller's (much smaller) buffer before the descending pair
    was seen, and chunks in front of a bad one had been delivered. Every offset is checked before anything is read now."""
    rng = np.random.default_rng(1)
    lits = random_literals(rng, 50, 3, 8)
    t = H.hwlm_build(lits)
    corpus = random_corpus(rng, 1 << 20, lits, plant_every=64)
    calls = []

    def on_chunk(recs):
        calls.append(len(recs))
        return False

    for bad in ([0, 3 << 30, 50], [0, 100, 50, 1 << 20], [0, 1 << 19, (1 << 19) - 1, 1 << 20], [0, 5 << 32]):
        off = np.array(bad, dtype=np.uint64)
        with pytest.raises(H.HsgpuError) as e:
            hw.hwlm_exec_batch_pipelined(t, scratch, corpus, off, chunk_bytes=1 << 16, on_chunk=on_chunk)
        assert e.value.code == -1 and not calls, (bad, calls)
    # many blocks: the multi-threaded walk (above 2^20 blocks), a single descending pair near the end
    off = np.concatenate([np.zeros((1 << 20) + 100, dtype=np.uint64), np.arange(0, (1 << 20) + 1, 1024, dtype=np.uint64)])  # a million empty blocks, then 1 KiB blocks
    assert hw.hwlm_exec_batch_pipelined(t, scratch, corpus, off, chunk_bytes=1 << 18, on_chunk=on_chunk) in (0, -3)
    n_good = len(calls)
    assert n_good > 0
    off2 = off.copy()
    off2[-5] = off2[-6] - 1
    with pytest.raises(H.HsgpuError) as e:
        hw.hwlm_exec_batch_pipelined(t, scratch, corpus, off2, chunk_bytes=1 << 18, on_chunk=on_chunk)
    assert e.value.code == -1 and len(calls) == n_good


def test_bench_virtual_ranks_gather_equals_one_scan(scratch):
    """bench.py's also.virtual_ranks (multi_gpu.loopback in the line): N virtual ranks on this one GPU -- a shard each, pack, the
    step's transfers over the loopback transport, compact -- deliver, at every rank (all-gather) / at the root (to-root), exactly
    the records of ONE scan of the concatenated corpus with global block indices, in corpus order."""
    import argparse

    import bench

    n_ranks = 4
    lits = bench.build_workload("fdr10k", 8 << 20, 0)[0]
    shards = [bench._build_workload("fdr10k", 8 << 20, sid)[1:] for sid in range(n_ranks)]
    args = argparse.Namespace(steps=3)
    out = bench.run_virtual_ranks(args, n_ranks, lits, shards, keep_rows=True)
    bench._WORKLOAD_CACHE.clear()
    corpus = np.concatenate([c for c, _o in shards])
    off, base = [np.zeros(1, dtype=np.uint64)], 0
    for c, o in shards:
        off.append(o[1:] + np.uint64(base))
        base += int(c.size)
    off = np.concatenate(off)
    t = H.hwlm_build(lits)
    one = hw.hwlm_exec_batch(t, scratch, corpus, off)
    want = np.stack([one["block"], one["end"], one["id"]], axis=1).astype(np.uint32)
    assert sum(out["records_per_rank"]) == len(want)
    for r in range(n_ranks):
        assert np.array_equal(out["_rows"]["all_gather"][r], want), f"rank {r}: all-gather rows differ from one scan"
    assert np.array_equal(out["_rows"]["to_root"][0], want)
    assert all(len(out["_rows"]["to_root"][r]) == 0 for r in range(1, n_ranks))
    for m in ("all_gather", "to_root"):
        assert out[m]["step_ms"] > 0 and out[m]["pack_ms"] > 0 and out[m]["compact_ms"] > 0
