"""The N > 1 path on CPU: world_size-2 gloo. Each rank takes its shard
(hyperscan_amd.dist), produces its match records (here with the CPU oracle --
tests may use it; the scan itself is covered by the -m gpu tests), and the
records are all-gathered exactly as bench.py does over RCCL. The union must equal
a single-process scan of the whole corpus."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hyperscan_amd import corpus as cp
from hyperscan_amd import dist as hd
from tests import oracle_binding as ob


def test_shards_cover_and_balance():
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 3000, 5000)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        sh = hd.shard_blocks_by_bytes(off, world)
        assert sh[0][0] == 0 and sh[-1][1] == off.size - 1
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        sizes = [int(off[hi] - off[lo]) for lo, hi in sh]
        assert max(sizes) - min(sizes) <= 2 * 3000
    # degenerate: fewer blocks than ranks, empty corpus
    assert hd.shard_blocks_by_bytes(np.array([0, 10], dtype=np.uint64), 4)[-1][1] == 1
    assert hd.shard_blocks_by_bytes(np.array([0], dtype=np.uint64), 2) == [(0, 0), (0, 0)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lits = cp.teddy_literals(32, seed=5)
    corpus, off = cp.packet_corpus(1 << 20, lits, seed=9, match_every=1024)
    my_corpus, my_off, base = hd.local_shard(corpus, off, rank, world)
    recs = ob.Oracle(lits).collect_blocks(my_corpus, my_off)
    t = torch.zeros((max(1, recs.size), 4), dtype=torch.int32)
    t[: recs.size, 0] = torch.from_numpy(recs["block"].astype(np.int32))
    t[: recs.size, 1] = torch.from_numpy(recs["end"].astype(np.int32))
    t[: recs.size, 2] = torch.from_numpy(recs["id"].astype(np.int32))
    allr, counts = hd.all_gather_records(t, recs.size, base, dist, world)
    if rank == 0:
        np.save(out_path, allr.numpy())
    assert sum(counts) == allr.shape[0]
    # the exact-size exchanges deliver the same rows in the same order
    exact, counts2 = hd.all_gather_records_exact(t, recs.size, base, dist, world, rank)
    assert counts2 == counts and torch.equal(exact, allr)
    # the bench's per-step form: fixed-size buffers, counts on the "device", two steps through the same buffers
    rows = 4096
    pad = torch.zeros((rows, 4), dtype=torch.int32)
    pad[: recs.size] = t[: recs.size]
    ex = hd.RecordExchange(dist, world, rank, torch.device("cpu"), rows, base)
    for _ in range(2):
        ex.step(pad, torch.tensor([recs.size], dtype=torch.int64))
    fixed, counts4 = ex.compact()
    assert counts4 == counts and torch.equal(fixed, allr)
    for root in range(world):
        rooted, counts3 = hd.gather_records_to_root(t, recs.size, base, dist, world, rank, root=root)
        assert counts3 == counts
        assert (rooted is None) if rank != root else torch.equal(rooted, allr)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_equals_single_scan(tmp_path):
    out = str(tmp_path / "gathered.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    lits = cp.teddy_literals(32, seed=5)
    corpus, off = cp.packet_corpus(1 << 20, lits, seed=9, match_every=1024)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    g = sorted(zip(got[:, 0].tolist(), got[:, 1].tolist(), got[:, 2].tolist()))
    w = sorted(zip(want["block"].tolist(), want["end"].tolist(), want["id"].tolist()))
    assert len(w) > 100 and g == w
    # shards are contiguous block ranges, so rank order == global block order
    assert np.all(np.diff(got[:, 0]) >= 0)


def _skew_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 1 holds a flood-dense shard, rank 0 a few rows, rank 2 nothing at all
    n = [5, 40_000, 0][rank]
    base = [0, 100, 1000][rank]
    g = torch.Generator().manual_seed(rank)
    t = torch.randint(0, 1 << 20, (max(1, n), 4), dtype=torch.int32, generator=g)
    want = []
    for r in range(world):
        m = [5, 40_000, 0][r]
        w = torch.randint(0, 1 << 20, (max(1, m), 4), dtype=torch.int32, generator=torch.Generator().manual_seed(r))[:m].clone()
        w[:, 0] += [0, 100, 1000][r]
        want.append(w)
    want = torch.cat(want)
    padded, counts = hd.all_gather_records(t, n, base, dist, world)
    exact, counts2 = hd.all_gather_records_exact(t, n, base, dist, world, rank)
    assert counts == counts2 == [5, 40_000, 0]
    assert torch.equal(padded, want) and torch.equal(exact, want)
    # the step-wise exact form (counts agreed once, as bench.py does after its warm-up): several steps, same result
    ex = hd.ExactExchange(dist, world, rank, torch.device("cpu"), [5, 40_000, 0], [0, 100, 1000])
    for _ in range(3):
        ex.step(t, None)
    stepwise, counts3 = ex.compact()
    assert counts3 == [5, 40_000, 0] and torch.equal(stepwise, want)
    rooted, _ = hd.gather_records_to_root(t, n, base, dist, world, rank, root=2)  # the empty rank collects
    assert (rooted is None) if rank != 2 else torch.equal(rooted, want)
    # nobody has anything
    e, c = hd.all_gather_records_exact(t, 0, base, dist, world, rank)
    assert c == [0, 0, 0] and e.shape == (0, 4)
    r0, c = hd.gather_records_to_root(t, 0, base, dist, world, rank)
    assert c == [0, 0, 0] and ((r0 is None) if rank else r0.shape == (0, 4))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_rank_skewed_exact_exchanges():
    mp.spawn(_skew_worker, args=(3, _free_port()), nprocs=3, join=True)


def test_block_base_beyond_2_31_wraps_like_uint32():
    """records travel as int32 tensors while hsgpu_match_t.block is uint32: adding a rank's first global block
    (which passes 2^31 on big jobs) must behave as uint32 arithmetic, and a base that does not fit 32 bits at all
    must be refused rather than wrapped silently"""
    assert hd._as_i32(0) == 0 and hd._as_i32((1 << 31) - 1) == (1 << 31) - 1
    assert hd._as_i32(1 << 31) == -(1 << 31) and hd._as_i32((1 << 32) - 1) == -1
    t = torch.tensor([[5, 0, 0, 0], [7, 0, 0, 0]], dtype=torch.int32)
    t[:, 0] += hd._as_i32(3_000_000_000)
    assert (t[:, 0].to(torch.int64) & 0xFFFFFFFF).tolist() == [3_000_000_005, 3_000_000_007]
    with pytest.raises(ValueError):
        hd._as_i32(1 << 32)
    with pytest.raises(ValueError):
        hd._as_i32(-1)
