"""GPU tests of round 5: the opt-in full-window Bloom gate of the confirm kernel (HSGPU_F_BLOOM, csrc/table.h) against the key
gate, no gate at all, the oracle and the compiled reference; solo scans (one launch for small batches) against the three-kernel
pipeline."""
import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import corpus as cp
from hyperscan_amd import hwlm as hw
from tests import oracle_binding as ob
from tests.util import random_blocks, random_corpus, random_literals

pytestmark = pytest.mark.gpu

F_GATE, F_BLOOM, NO_GATE, FORCE_BLOOM = 512, 2048, 4096, 8192


def as_sorted(r):
    a = np.stack([r["block"].astype(np.int64), r["end"].astype(np.int64), r["id"].astype(np.int64)], axis=1)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


@pytest.mark.parametrize("nocase_frac", [0.3, 0.0])
def test_bloom_gate_key_gate_and_no_gate_find_the_same_matches(nocase_frac):
    """3 000 random literals of 3-8 bytes plus literals whose msk / cmp reach in front of the string or carry wildcard bits, over a
    corpus in which they are planted in both cases and with near misses (the fifth byte from the end wrong: exactly what group 5
    of the Bloom gate rejects and the key gate lets through): the three tables deliver the oracle's and the reference's records
    in every pipeline that has a confirm kernel."""
    rng = np.random.default_rng(501 + int(nocase_frac * 10))
    lits = random_literals(rng, 3000, 3, 8, nocase_frac=nocase_frac)
    n0 = len(lits)
    lits += [H.HwlmLiteral("abcde", False, n0, msk=b"\xff\xf0\xff\xff\xff", cmp=b"a\x60cde"),
             H.HwlmLiteral("qrstuv", False, n0 + 1, msk=b"\x00\xff\xff\xff\xff\xff", cmp=b"\x00rstuv"),
             H.HwlmLiteral("wxyz", False, n0 + 2, msk=b"\x0f\xff\xff\xff\xff", cmp=b"\x01wxyz"),
             H.HwlmLiteral("mnop", False, n0 + 3, msk=b"\xff\xff\xff\xff\xff", cmp=b"Zmnop")]
    lits = [H.HwlmLiteral(l.s, l.nocase, i, msk=l.msk, cmp=l.cmp) for i, l in enumerate(lits)]
    corpus = random_corpus(rng, 3_000_000, lits, plant_every=150).copy()
    # near misses and the masked literals' own contexts
    for _ in range(4000):
        l = lits[int(rng.integers(0, len(lits)))]
        p = int(rng.integers(16, corpus.size - 16))
        corpus[p:p + len(l.s)] = np.frombuffer(l.s, dtype=np.uint8)
        if len(l.s) >= 5 and rng.random() < 0.5:
            corpus[p + len(l.s) - 5] ^= 0x01
    for ctx in (b"Zmnop", b"zmnop", b"\x31wxyz", b"\x32wxyz", b"a\x6fcde", b"abcde", b"?RSTUV", b"?rstuv"):
        for _ in range(50):
            p = int(rng.integers(16, corpus.size - 16))
            corpus[p:p + len(ctx)] = np.frombuffer(ctx, dtype=np.uint8)
    off = random_blocks(rng, corpus.size, mean_len=900)
    want = as_sorted(ob.Oracle(lits).collect_blocks(corpus, off))
    if ob.ref_available():
        assert np.array_equal(as_sorted(ob.Reference(lits).collect_blocks(corpus, off)), want)
    assert len(want) > 20000
    scratch = H.Scratch(0)
    seen = set()
    for flags in (FORCE_BLOOM, 0, NO_GATE):
        t = H.hwlm_build(lits, flags)
        fl = t.info()["flags"]
        seen.add(fl & (F_GATE | F_BLOOM))
        for tuning in (0, 2):  # folded (default); confirm kernel + record_sort_kernel
            scratch.set_tuning(tuning)
            got = hw.hwlm_exec_batch(t, scratch, corpus, off)
            assert np.array_equal(as_sorted(got), want), (flags, tuning, len(got), len(want))
    assert seen == {F_BLOOM, F_GATE, 0}


def test_bloom_gate_bench_set_identical_records_to_the_key_gate():
    """the 10 000-literal bench set on 32 MiB of its corpus: element for element the same record array with either gate"""
    lits, _ = cp.snort_like_literals(10000, seed=4)
    corpus, off = cp.packet_corpus(32 << 20, lits, seed=10)
    scratch = H.Scratch(0)
    out = []
    for flags in (FORCE_BLOOM, 0):
        t = H.hwlm_build(lits, flags)
        assert bool(t.info()["flags"] & F_BLOOM) == (flags == FORCE_BLOOM)
        r = hw.hwlm_exec_batch(t, scratch, corpus, off)
        out.append(np.stack([r["block"], r["end"], r["id"]], axis=1))
    assert len(out[0]) > 10000 and np.array_equal(out[0], out[1])


# ---- solo scans: one launch for small batches (csrc/scan_device.h solo_tail, csrc/runtime.hip launch_scan) ---------------

class _Resident:
    def __init__(self, table, corpus, off, cap, timing=False):
        import torch

        self.torch = torch
        dev = torch.device("cuda", 0)
        self.t, self.s = table, H.Scratch(0)
        if timing:
            self.s.enable_timing(True)
        self.total, self.nblocks, self.cap = int(corpus.size), int(off.size - 1), int(cap)
        self.d_corpus = torch.from_numpy(np.concatenate([corpus, np.zeros(16, np.uint8)])).to(dev)
        self.d_off = torch.from_numpy(off.astype(np.uint64).view(np.int64)).to(dev)
        self.d_out = torch.zeros(max(1, self.cap) * 4, dtype=torch.int32, device=dev)
        self.d_count = torch.zeros(1, dtype=torch.int64, device=dev)

    def scan(self, total=None, nblocks=None):
        st = self.torch.cuda.current_stream().cuda_stream
        hw.hwlm_scan_dev(self.t, self.s, self.d_corpus.data_ptr(), self.total if total is None else total, self.d_off.data_ptr(),
                         self.nblocks if nblocks is None else nblocks, self.d_out.data_ptr(), self.cap, self.d_count.data_ptr(), 0, st)
        self.torch.cuda.synchronize()
        n = int(self.d_count.item())
        return n, self.d_out[: min(n, self.cap) * 4].view(-1, 4).cpu().numpy().astype(np.uint32)


SOLO_OFF, SOLO_ALWAYS = 3, 4  # hsgpu_scratch_set_tuning: never the single launch / whenever the geometry allows


@pytest.mark.parametrize("workload", ["teddy64", "fdr10k", "mixed3000"])
def test_solo_scans_equal_the_three_kernel_pipeline_at_every_small_size(workload):
    """A resident batch of up to 64 KiB takes ONE launch (the fused kernel, its last workgroup placing the records). The same
    record array, element for element, as the three-kernel pipeline on the same scratch (set_tuning(3) switches the single
    launch off), and the oracle's multiset, at sizes around every edge: one byte, less than a chunk, a packet, tile and
    workgroup-share boundaries, exactly 64 KiB, one byte more (no longer solo), and on to 1 MiB. The scans alternate on ONE
    scratch, so each path must leave the other's control words as it found them."""
    rng = np.random.default_rng(77)
    if workload == "teddy64":
        lits = cp.teddy_literals(64, seed=2)
        corpus, off = cp.packet_corpus((1 << 20) + 4096, lits, seed=5, match_every=512)
    elif workload == "fdr10k":
        lits, _ = cp.snort_like_literals(10000, seed=4)
        corpus, off = cp.packet_corpus((1 << 20) + 4096, lits, seed=6)
    else:
        lits = random_literals(rng, 3000, 3, 8, nocase_frac=0.3)  # (1- and 2-byte literals match every few bytes: dense mode, where no scan is solo)
        lits = [H.HwlmLiteral(l.s, l.nocase, i) for i, l in enumerate(lits)]
        corpus = random_corpus(rng, (1 << 20) + 4096, lits, plant_every=300)
        off = random_blocks(rng, corpus.size, mean_len=700)
    table = H.hwlm_build(lits)
    oracle = ob.Oracle(lits)
    r = _Resident(table, corpus, off, cap=1 << 18, timing=True)
    sizes = [1, 7, 15, 16, 17, 1023, 1024, 1025, 1460, 9000, 16383, 16384, 16385, 40000, 65535, 65536, 65537, 300001, 1 << 20, (1 << 20) + 1]
    for sz in sizes:
        k = int(np.searchsorted(off, sz, side="right")) - 1
        if k < 1:  # the first block alone, cut to sz bytes
            sub_off = np.array([0, sz], dtype=np.uint64)
        else:
            sub_off = np.concatenate([off[:k], [np.uint64(sz)]]).astype(np.uint64) if int(off[k]) != sz else off[: k + 1].copy()
        nb = int(sub_off.size - 1)
        r.d_off[: nb + 1] = r.torch.from_numpy(sub_off.view(np.int64)).to(r.d_off.device)
        want = as_sorted(oracle.collect_blocks(corpus[:sz], sub_off))
        got = {}
        for tuning in (0, SOLO_OFF, 0):
            r.s.set_tuning(tuning)
            n, recs = r.scan(sz, nb)
            assert n == len(want), (workload, sz, tuning, n, len(want))
            key = (recs[:, 0].astype(np.uint64) << np.uint64(32)) | recs[:, 1].astype(np.uint64)
            assert np.all((key[1:] > key[:-1]) | ((key[1:] == key[:-1]) & (recs[1:, 3] > recs[:-1, 3]))), (workload, sz, tuning, "delivery order")
            got[tuning] = recs
        assert np.array_equal(got[0], got[SOLO_OFF]), (workload, sz)
        a = np.stack([got[0][:, 0], got[0][:, 1], got[0][:, 2]], axis=1).astype(np.int64)
        assert np.array_equal(a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))], want), (workload, sz)
    # the timing slots of a solo scan are filled like any other's
    f, c, t = r.s.timing(0)
    assert f > 0 and t > 0


@pytest.mark.parametrize("workload", ["teddy64", "fdr10k", "flood", "pair"])
def test_forced_solo_on_large_and_dense_batches(workload):
    """set_tuning(4) takes the single launch at any size (at most 64 workgroups, 1024 regions): 24 MiB of the bench workloads,
    a flood piece (regions of thousands of records: the large-region path of the placement, sort_share's LDS network and its
    merge passes) and the opt-in pair filter (late-keyed literals at share edges): element for element the three-kernel
    pipeline's array; a record buffer that is too small reports the exact count ("again" semantics) in both."""
    if workload == "teddy64":
        lits = cp.teddy_literals(64, seed=2)
        corpus, off = cp.packet_corpus(24 << 20, lits, seed=33, match_every=2048)
        flags = 0
    elif workload == "fdr10k":
        lits, _ = cp.snort_like_literals(10000, seed=4)
        corpus, off = cp.packet_corpus(24 << 20, lits, seed=34)
        flags = 0
    elif workload == "flood":
        lits = [H.HwlmLiteral(b"aaaa", False, 0), H.HwlmLiteral(b"aaaaaaaa", False, 1), H.HwlmLiteral(b"aaab", False, 2)] + \
               [H.HwlmLiteral(l.s, l.nocase, 3 + i) for i, l in enumerate(cp.teddy_literals(100, seed=12))]
        corpus = np.full(600_000, ord("z"), dtype=np.uint8)
        corpus[100_000:160_000] = ord("a")
        corpus[400_000:400_700] = ord("a")
        off = np.array([0, 50_000, 130_000, 130_001, 420_000, 600_000], dtype=np.uint64)
        flags = 0
    else:
        rng = np.random.default_rng(78)
        alpha = np.frombuffer(b"abcdef", dtype=np.uint8)
        corpus = rng.choice(alpha, 4 << 20).astype(np.uint8)
        strs = [b"abc", b"bcd", b"fed", b"cab", b"aabc", b"fbcd", b"efed", b"dcab", b"abcde", b"dea"]
        lits = [H.HwlmLiteral(s_, False, 100 + i) for i, s_ in enumerate(strs)]
        off = np.array([0, 1 << 20, (1 << 20) + 16384 * 3 + 1, 4 << 20], dtype=np.uint64)
        flags = 1024  # FORCE_PAIR
    table = H.hwlm_build(lits, flags)
    want_n = len(ob.Oracle(lits).collect_blocks(corpus, off))
    r = _Resident(table, corpus, off, cap=want_n + 4096)
    out = {}
    for tuning in (SOLO_ALWAYS, SOLO_OFF):
        r.s.set_tuning(tuning)
        n, recs = r.scan()
        tries = 0
        while n > r.cap and tries < 4:  # "again": a staging region (sized from cap for an even spread) overflowed -- more room
            tries += 1
            r.cap *= 2
            r.d_out = r.torch.zeros(r.cap * 4, dtype=r.torch.int32, device=r.d_out.device)
            n, recs = r.scan()
        assert n == want_n, (workload, tuning, n, want_n)
        out[tuning] = recs
    assert np.array_equal(out[SOLO_ALWAYS], out[SOLO_OFF]), workload
    # a buffer that cannot hold the records: a count above cap, nothing claimed complete
    small = _Resident(table, corpus, off, cap=max(1, want_n // 3))
    small.s.set_tuning(SOLO_ALWAYS)
    n, _ = small.scan()
    assert n > small.cap


def test_small_host_batches_take_no_copy_commands_and_return_the_same_records():
    """hsgpu_hwlm_exec / hsgpu_hwlm_exec_batch on a small host batch: the batch in mapped pinned memory, the kernel reading it and
    writing records and count back (csrc/runtime.hip scan_host_small). Same callbacks as the general path (set_tuning(3) on a
    second scratch keeps the three-kernel pipeline; a batch above 256 KiB takes the copying path), one packet at a time as
    hsbench drives hs_scan, with `start`, empty blocks and a batch whose matches exceed the small path's record buffer."""
    lits, _ = cp.snort_like_literals(10000, seed=4)
    corpus, off = cp.packet_corpus(2 << 20, lits, seed=6)
    table = H.hwlm_build(lits)
    oracle = ob.Oracle(lits)
    s1, s2 = H.Scratch(0), H.Scratch(0)
    s2.set_tuning(SOLO_OFF)
    n_cb = 0
    for b in list(range(0, 40)) + list(range(900, 930)):
        blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
        for start in (0, 5):
            want = oracle.collect_blocks(blk, np.array([0, blk.size], dtype=np.uint64), start=start)
            for s in (s1, s2):
                got = []
                rv = hw.hwlm_exec(table, blk, start, lambda end, lid, ctx: (got.append((end, lid)), hw.HWLM_ALL_GROUPS)[1], s)
                assert rv == 0
                assert sorted(got) == sorted(zip(want["end"].tolist(), want["id"].tolist())), (b, start)
                n_cb += len(got)
    assert n_cb > 20
    for nbytes in (1, 100_000, 262_144, 262_145, 700_000):
        k = max(1, int(np.searchsorted(off, nbytes, side="right")) - 1)
        sub_off = off[: k + 1].copy()
        sub_off[1:4] = sub_off[1]  # two empty blocks
        want = as_sorted(oracle.collect_blocks(corpus, sub_off))
        for s in (s1, s2):
            assert np.array_equal(as_sorted(hw.hwlm_exec_batch(table, s, corpus, sub_off)), want), nbytes
    # more matches than the small path's buffer (4096 records): the general path takes over
    flood = np.full(20_000, ord("a"), dtype=np.uint8)
    fl = [H.HwlmLiteral(b"aaaa", False, 0), H.HwlmLiteral(b"aa", False, 1)]
    got = hw.hwlm_exec_batch(H.hwlm_build(fl), s1, flood, np.array([0, 20_000], dtype=np.uint64))
    assert len(got) == (20_000 - 3) + (20_000 - 1)


@pytest.mark.parametrize("seed", [1, 2])
def test_dense_runs_of_any_length_among_quiet_blocks_spread_parts(seed):
    """Dense scans since round 5: a record region per PART (half batches), and a worker's parts spread over the corpus row by row
    (csrc/runtime.hip, hwlm_confirm_kernel: conf_spread), the block of the last drain cached (resolve_queued). Flood runs of
    uneven length (the reference's flood case, unit/internal/fdr_flood.cpp:148-557) between ordinary packets, sizes chosen so that
    the part count is no multiple of the worker count and the last row of parts is not full; runs that end inside a half batch;
    a run cut into many small blocks (the cached block changes inside a drain). Every record against the oracle, in delivery
    order, twice (the second scan is the dense one's repeat)."""
    rng = np.random.default_rng(seed)
    lits = [H.HwlmLiteral(b"aaaa", False, 7), H.HwlmLiteral(b"aaaaaaaa", False, 8), H.HwlmLiteral(b"aaax", False, 9)]
    lits += [H.HwlmLiteral(l.s, l.nocase, 100 + i) for i, l in enumerate(cp.teddy_literals(40, seed=12))]
    pieces, sizes = [], []
    quiet, qoff = cp.packet_corpus(6 << 20, lits[3:], seed=40 + seed, match_every=4096)
    qb = 0
    for run in (300 * 1024 + 5, 1300 * 1024 + 777, 77 * 1024, 33, 512 * 1024 + 1):
        take = int(rng.integers(200, 900))  # quiet blocks in front of the run
        for b in range(qb, min(qb + take, qoff.size - 1)):
            pieces.append(quiet[int(qoff[b]):int(qoff[b + 1])])
            sizes.append(pieces[-1].size)
        qb += take
        if run > 400 * 1024:  # one run as many small blocks of uneven size
            left = run
            while left:
                n = int(min(left, rng.integers(40, 1500)))
                pieces.append(np.full(n, ord("a"), np.uint8))
                sizes.append(n)
                left -= n
        else:
            pieces.append(np.full(run, ord("a"), np.uint8))
            sizes.append(run)
    corpus = np.concatenate(pieces)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    oracle = ob.Oracle(lits)
    want = []
    for b in range(off.size - 1):
        for e, i in oracle.collect(corpus[int(off[b]):int(off[b + 1])]):
            want.append((b, e, i))
    want = np.array(sorted(want), dtype=np.int64)
    from tests.test_gpu_round4 import Resident, _in_delivery_order

    r = Resident(lits, corpus, off, cap=want.shape[0] + (1 << 16))
    n, tries = r.scan(), 0
    while n > r.cap and tries < 8:
        tries += 1
        if tries > 1:
            r.cap *= 2
            r.d_out = r.torch.zeros(r.cap * 4, dtype=r.torch.int32, device=r.d_out.device)
        n = r.scan()
    assert tries >= 1, "the runs were meant to send the scratch to dense mode"
    for again in range(2):
        assert n == want.shape[0], (n, want.shape[0], tries)
        got = r.records(n)
        assert _in_delivery_order(got)
        a = np.stack([got[:, 0].astype(np.int64), got[:, 1].astype(np.int64), got[:, 2].astype(np.int64)], axis=1)
        a = a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        assert np.array_equal(a, want)
        n = r.scan()


def test_skewed_halves_of_the_shares_deliver_the_same_array_as_equal_halves():
    """Since round 5 the confirm kernel's workers take the two halves of a candidate share in proportion to their workgroup's
    dispatch rank, a record region belongs to a part and not to a worker (csrc/runtime.hip: conf_skew; hsgpu_scratch_set_tuning
    5 = equal halves, worker w = part w as before). Same record array either way, element for element, in delivery order, and the
    oracle's multiset; a corpus whose candidate density changes along its length (text, then random bytes, then text with many
    matches) so that the shares' batch counts differ and most cuts fall inside a batch."""
    from tests.test_gpu_round4 import Resident, _in_delivery_order

    lits, _ = cp.snort_like_literals(3000, seed=9)
    a, aoff = cp.packet_corpus(20 << 20, lits, seed=51)
    rng = np.random.default_rng(3)
    b = rng.integers(0, 256, 6 << 20, dtype=np.uint8)
    boff = np.arange(0, b.size + 1, 1024, dtype=np.uint64)
    c, coff = cp.packet_corpus(10 << 20, lits, seed=52, match_every=256)
    corpus = np.concatenate([a, b, c])
    off = np.concatenate([aoff, boff[1:] + np.uint64(a.size), coff[1:] + np.uint64(a.size + b.size)]).astype(np.uint64)
    r = Resident(lits, corpus, off, cap=1 << 21)
    got = {}
    for name, code in (("skewed", 0), ("equal", 5), ("skewed again", 0)):
        r.s.set_tuning(code)
        n = r.scan()
        assert 1000 < n <= r.cap
        got[name] = r.records(n)
        assert _in_delivery_order(got[name])
    assert np.array_equal(got["skewed"], got["equal"]) and np.array_equal(got["skewed"], got["skewed again"])
    oracle = ob.Oracle(lits)
    want = []
    for blk in range(off.size - 1):
        want += [(blk, e, i) for e, i in oracle.collect(corpus[int(off[blk]):int(off[blk + 1])])]
    want = np.array(sorted(want), dtype=np.int64)
    g = got["skewed"]
    have = np.stack([g[:, 0].astype(np.int64), g[:, 1].astype(np.int64), g[:, 2].astype(np.int64)], axis=1)
    have = have[np.lexsort((have[:, 2], have[:, 1], have[:, 0]))]
    assert np.array_equal(have, want)
