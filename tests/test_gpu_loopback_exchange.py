"""The N > 1 exchange step on ONE GPU (round 5): hsgpu_exchange_* over the in-process loopback transport
(hsgpu_exchange_loopback_id) -- 2 to 8 virtual ranks of this process, each a NativeExchange of its own. The step's logic (which
rank sends what to whom, slot offsets, bytes_of with and without agreed counts, the slot headers compact reads, overflow) is the
code the RCCL transport runs at N = 8; only the fabric underneath differs. Expectations = those of the gloo tests
(tests/test_dist_cpu.py): the union in rank order equals one scan of the whole corpus with global block indices, with skewed,
empty and overflowing ranks."""
import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import corpus as cp
from hyperscan_amd import dist as hd
from hyperscan_amd import hwlm as hw
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

TO_ROOT, ALL_GATHER = 0, 1


def _fake_records(rng, n, nblocks, cap):
    """a rank's scan output without a scan: n sorted rows (block, end, id, lit) in an int32 [cap, 4] device tensor + its counter"""
    import torch

    dev = torch.device("cuda", 0)
    rec = np.zeros((cap, 4), dtype=np.int32)
    m = min(n, cap)  # (n > cap: a scan that overflowed its record buffer -- the count says so, the buffer holds nothing usable)
    if m:
        b = np.sort(rng.integers(0, max(1, nblocks), m))
        rec[:m, 0], rec[:m, 1], rec[:m, 2], rec[:m, 3] = b, rng.integers(0, 1500, m), rng.integers(0, 10000, m), rng.integers(0, 10000, m)
    rec[m:] = -7  # what lies behind the count must never travel
    return torch.from_numpy(rec).to(dev), torch.tensor([n], dtype=torch.int64, device=dev), rec[:m]


def _run(world, mode, counts, rows, agreed=None, order=None, own_streams=False, steps=1, root=0, caps=None):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(world * 1000 + mode * 100 + sum(counts) % 97)
    nbl = [1000 + 10 * r for r in range(world)]
    bases = np.concatenate([[0], np.cumsum(nbl)])[:world].tolist()
    lid = hd.NativeExchange.loopback_id()
    xs = [hd.NativeExchange(None, world, r, dev, rows, bases[r], mode=mode, root=root, id_bytes=lid) for r in range(world)]
    caps = caps or [max(rows, c) + 8 for c in counts]
    data = [_fake_records(rng, counts[r], nbl[r], caps[r]) for r in range(world)]
    if agreed is not None:
        for x in xs:
            x.set_counts(agreed)
    streams = [torch.cuda.Stream(device=dev) if own_streams else torch.cuda.current_stream() for _ in range(world)]
    for _ in range(steps):
        for r in (order or range(world)):
            with torch.cuda.stream(streams[r]):
                xs[r].step(data[r][0], data[r][1], cap=caps[r])
    outs = []
    for r in range(world):
        with torch.cuda.stream(streams[r]):
            try:
                outs.append(xs[r].compact())
            except RuntimeError as e:
                outs.append(e)
    torch.cuda.synchronize()
    want = np.concatenate([np.stack([d[2][:, 0] + bases[r], d[2][:, 1], d[2][:, 2]], axis=1) for r, d in enumerate(data)]).astype(np.int32)
    wires = [x.wire_bytes() for x in xs]
    for x in xs:
        x.close()
    return outs, want, wires


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("mode", [TO_ROOT, ALL_GATHER])
@pytest.mark.parametrize("exact", [False, True])
def test_loopback_exchange_delivers_rank_order(world, mode, exact):
    counts = [int(c) for c in np.random.default_rng(world).integers(50, 4000, world)]
    rows = max(counts) + 100
    outs, want, wires = _run(world, mode, counts, rows, agreed=counts if exact else None)
    for r, o in enumerate(outs):
        assert not isinstance(o, Exception), o
        out, got_counts = o
        if mode == ALL_GATHER or r == 0:
            assert got_counts == counts
            assert np.array_equal(out.cpu().numpy(), want), f"rank {r}: gathered records differ"
        else:
            assert out.shape[0] == 0
    slot = lambda n: (16 + 12 * n + 15) & ~15
    per = [slot(c) if exact else slot(rows) for c in counts]
    if mode == TO_ROOT:
        assert wires[0] == (0, sum(per[1:])) and all(wires[r] == (per[r], 0) for r in range(1, world))
    else:
        assert all(wires[r] == (per[r] * (world - 1), sum(per) - per[r]) for r in range(world))


@pytest.mark.parametrize("mode", [TO_ROOT, ALL_GATHER])
def test_loopback_skewed_and_empty_ranks(mode):
    """the gloo tests' skew (40 000 : 5 : 0) and an all-empty step, exact and padded"""
    for counts in ([40000, 5, 0], [0, 0, 0], [0, 7, 0, 0, 1]):
        for exact in (False, True):
            outs, want, _ = _run(len(counts), mode, counts, max(counts) + 3, agreed=counts if exact else None)
            out, got = outs[0]
            assert got == counts and np.array_equal(out.cpu().numpy().reshape(-1, 3), want.reshape(-1, 3))


def test_loopback_any_driving_order_own_streams_and_repeated_steps():
    """ranks driven root-last / root-first / shuffled, each on a stream of its own, three steps in a row: the copy a Send and
    its Recv become is posted by whichever comes second, and a rank waits for what others posted before touching its buffers"""
    counts = [1200, 10, 3000, 777]
    for order in ([3, 2, 1, 0], [0, 1, 2, 3], [2, 0, 3, 1]):
        for mode in (TO_ROOT, ALL_GATHER):
            outs, want, _ = _run(4, mode, counts, 3100, order=order, own_streams=True, steps=3)
            out, got = outs[0]
            assert got == counts and np.array_equal(out.cpu().numpy(), want)
            if mode == ALL_GATHER:
                for o in outs[1:]:
                    assert np.array_equal(o[0].cpu().numpy(), want)


def test_loopback_root_other_than_zero():
    counts = [100, 200, 300]
    outs, want, wires = _run(3, TO_ROOT, counts, 400, root=2)
    assert outs[0][0].shape[0] == 0 and outs[1][0].shape[0] == 0
    out, got = outs[2]
    assert got == counts and np.array_equal(out.cpu().numpy(), want) and wires[2][0] == 0 and wires[2][1] > 0


def test_loopback_slot_overflow_and_scan_overflow():
    """a rank that found more than a slot holds -> compact says INSUFFICIENT_SPACE where the records arrive; and the advisor's
    case (round 4): the SCAN overflowed its record buffer (count > cap) while a slot would have held the count -- the slot
    travels with no rows, nothing is read past the buffer, and compact refuses"""
    outs, _w, _ = _run(3, TO_ROOT, [100, 5000, 10], rows=1000)
    assert isinstance(outs[0], RuntimeError) and "rc -12" in str(outs[0])
    assert not isinstance(outs[1], Exception) and not isinstance(outs[2], Exception)  # non-roots have nothing to refuse
    outs, _w, _ = _run(3, ALL_GATHER, [100, 5000, 10], rows=1000)
    assert all(isinstance(o, RuntimeError) for o in outs)
    # count 900 <= rows 1000, but the scan's buffer held only 500
    outs, _w, _ = _run(2, TO_ROOT, [900, 20], rows=1000, caps=[500, 1008])
    assert isinstance(outs[0], RuntimeError) and "rc -12" in str(outs[0])


def test_loopback_ranks_that_disagree_on_sizes_get_an_error_not_a_hang():
    import torch

    dev = torch.device("cuda", 0)
    lid = hd.NativeExchange.loopback_id()
    xs = [hd.NativeExchange(None, 2, r, dev, 1000, 0, mode=TO_ROOT, id_bytes=lid) for r in range(2)]
    xs[0].set_counts([10, 20])
    xs[1].set_counts([10, 30])
    rng = np.random.default_rng(1)
    d0, d1 = _fake_records(rng, 10, 50, 1008), _fake_records(rng, 30, 50, 1008)
    xs[0].step(d0[0], d0[1])
    with pytest.raises(RuntimeError, match="disagree"):
        xs[1].step(d1[0], d1[1])
    for x in xs:
        x.close()
    with pytest.raises(RuntimeError):  # an id is made for one world size
        lid = hd.NativeExchange.loopback_id()
        a = hd.NativeExchange(None, 2, 0, dev, 10, 0, id_bytes=lid)
        hd.NativeExchange(None, 3, 1, dev, 10, 0, id_bytes=lid)
    a.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_scans_through_the_loopback_exchange_equal_one_scan(world):
    """SURVEY 8(e) end to end on one device: the corpus sharded by contiguous block ranges (hyperscan_amd.dist.local_shard), every
    virtual rank scanning its shard with hsgpu_hwlm_scan_dev, ONE exchange step, and the root's compacted records = the oracle's
    (block, end, id) of the whole corpus in delivery order."""
    import torch

    dev = torch.device("cuda", 0)
    lits = cp.teddy_literals(64, seed=2)
    corpus, off = cp.packet_corpus(12 << 20, lits, seed=91, match_every=2048)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    order = np.lexsort((want["id"], want["end"], want["block"]))
    table = H.hwlm_build(lits)
    lid = hd.NativeExchange.loopback_id()
    cap = 1 << 16
    ranks = []
    for r in range(world):
        c, o, base = hd.local_shard(corpus, off, r, world)
        s = H.Scratch(0)
        d_c = torch.from_numpy(np.concatenate([c, np.zeros(16, np.uint8)])).to(dev)
        d_o = torch.from_numpy(o.view(np.int64)).to(dev)
        d_out = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
        d_n = torch.zeros(1, dtype=torch.int64, device=dev)
        x = hd.NativeExchange(None, world, r, dev, cap, base, mode=TO_ROOT, id_bytes=lid)
        ranks.append((s, d_c, int(c.size), d_o, int(o.size - 1), d_out, d_n, x))
    st = torch.cuda.current_stream().cuda_stream
    for s, d_c, total, d_o, nb, d_out, d_n, x in reversed(ranks):  # the root last: its receives meet sends already posted
        hw.hwlm_scan_dev(table, s, d_c.data_ptr(), total, d_o.data_ptr(), nb, d_out.data_ptr(), cap, d_n.data_ptr(), 0, st)
        x.step(d_out, d_n)
    out, counts = ranks[0][7].compact()
    assert sum(counts) == len(want)
    g = out.cpu().numpy().astype(np.uint32)
    # (ties inside one (block, end) are in literal-index order on the device; the oracle's multiset is compared sorted)
    gi = np.lexsort((g[:, 2], g[:, 1], g[:, 0]))
    assert np.array_equal(g[gi, 0], want["block"][order]) and np.array_equal(g[gi, 1], want["end"][order]) and \
        np.array_equal(g[gi, 2], want["id"][order])
    k = (g[:, 0].astype(np.uint64) << np.uint64(32)) | g[:, 1].astype(np.uint64)
    assert np.all(k[1:] >= k[:-1]), "rank order must be corpus order"
    for rk in ranks:
        rk[7].close()


def test_loopback_compact_before_the_peers_have_stepped_is_an_error():
    """Over RCCL a rank that gathers before its peers have sent blocks; the loopback transport blocks nobody, and a rank that
    compacted early would read the slots of the step before (advisor, round 5). Its own receives still waiting for their senders say
    so: hsgpu_exchange_compact refuses, and works once everybody has stepped."""
    import torch

    dev = torch.device("cuda", 0)
    lid = hd.NativeExchange.loopback_id()
    xs = [hd.NativeExchange(None, 3, r, dev, 500, 100 * r, mode=TO_ROOT, id_bytes=lid) for r in range(3)]
    rng = np.random.default_rng(3)
    data = [_fake_records(rng, 40 + r, 50, 508) for r in range(3)]
    xs[0].step(data[0][0], data[0][1])  # the root: its receives are posted, nobody has sent
    xs[1].step(data[1][0], data[1][1])
    with pytest.raises(RuntimeError, match="before 1 of its peers"):
        xs[0].compact()
    xs[2].step(data[2][0], data[2][1])
    out, got = xs[0].compact()
    torch.cuda.synchronize()
    assert got == [40, 41, 42] and out.shape[0] == 123
    for x in xs:
        x.close()
