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


def test_hs_scan_batch_resident_equals_hs_scan_batch():
    """hs_scan_batch_resident (include/hs_gpu.h): the batch already on the device, only the literal hits cross the bus; the same
    events, in the same order, as hs_scan_batch on the host copy -- small batches (single-threaded confirm) and large ones."""
    import torch

    from hyperscan_amd import corpus as cp
    from hyperscan_amd import hs
    from tests import rose_model as RM

    rng = np.random.default_rng(8)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
    lits = sorted({bytes(rng.choice(alpha, int(rng.integers(6, 13)))) for _ in range(200)})
    pats = [l.decode() + RM.TAILS[i % 3] for i, l in enumerate(lits)]
    db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))
    sc = hs.HsScratch(db)
    lib = hs._lib()
    lib.hs_scan_batch_resident.restype = C.c_int
    lib.hs_scan_batch_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p, hs.BATCH_CB, C.c_void_p]

    class L:
        def __init__(self, s):
            self.s = s
    follow = [b"abc7", b"  key=", b"....END"]
    for total, every in ((1 << 20, 4096), (48 << 20, 256)):
        corpus, off = cp.packet_corpus(total, [L(l + follow[i % 3]) for i, l in enumerate(lits)], seed=3, match_every=every)
        offs = np.ascontiguousarray(off, dtype=np.uint64)
        a, b = [], []
        cb_a = hs.BATCH_CB(lambda blk, i, f, t, _fl, _c: (a.append((int(blk), int(i), int(t))), 0)[1])
        cb_b = hs.BATCH_CB(lambda blk, i, f, t, _fl, _c: (b.append((int(blk), int(i), int(t))), 0)[1])
        assert lib.hs_scan_batch(db._h, corpus.ctypes.data, offs.ctypes.data, offs.size - 1, 0, sc._h, cb_a, None) == 0
        d_corpus = torch.from_numpy(corpus).to("cuda:0")
        d_off = torch.from_numpy(offs.view(np.int64)).to("cuda:0")
        for _ in range(2):  # (the second call reuses the scratch's buffers)
            b.clear()
            assert lib.hs_scan_batch_resident(db._h, corpus.ctypes.data, offs.ctypes.data, offs.size - 1, d_corpus.data_ptr(), d_off.data_ptr(),
                                              sc._h, cb_b, None) == 0
            assert a == b and len(a) > 100, (total, len(a), len(b))


@pytest.mark.parametrize("mailbox", [1, 2])
def test_small_batch_server_parity_lifetime_and_fallbacks(mailbox):
    """(mailbox 1: the requests written into device memory through the PCIe BAR where the device has a large one; 2: through
    mapped host memory.)
    The small-batch server (include/hsgpu.h, hsgpu_scratch_enable_server; scan_device.h, hwlm_server_kernel): hwlm_exec and
    hwlm_exec_batch calls of up to 16 KiB served by ONE resident workgroup -- the same records as the launch path and as the
    oracle at every size up to its limit and beyond it (where the call falls back to a launch), over several table layouts; it
    ends by itself when idle and comes back with the next call; another table, another pipeline, a large scan on the same scratch
    and hsgpu_scratch_free all end it first."""
    import time

    from hyperscan_amd import corpus as cp
    from tests import oracle_binding as ob
    from tests.util import as_set, random_blocks

    rng = np.random.default_rng(66)
    sets = {"teddy64": (random_literals(rng, 64, 4, 8, nocase_frac=0.2), 0), "mixed3000": (random_literals(rng, 3000, 1, 8, nocase_frac=0.3), 0),
            "fdr10k": (cp.snort_like_literals(10000, seed=4)[0], 0), "one": ([H.HwlmLiteral(b"needle", nocase=True, id=7)], 0)}
    s = H.Scratch(0)
    s.enable_server(mailbox, idle_us=2000)
    plain = H.Scratch(0)
    served_before = 0
    for name, (lits, flags) in sets.items():
        t = H.hwlm_build(lits, flags)
        oracle = ob.Oracle(lits)
        for total in [1, 2, 7, 16, 17, 100, 1023, 1024, 1025, 1460, 4096, 8191, 8192, 8193, 16383, 16384, 16385, 20000, 70000]:
            corpus = random_corpus(rng, total, lits, plant_every=97)
            got = []
            assert H.hwlm_exec(t, corpus, 0, lambda e, i, c: got.append((e, i)) or H.HWLM_CONTINUE_MATCHING, s) == H.HWLM_SUCCESS
            assert sorted(got) == sorted(oracle.collect(corpus)), (name, total)
            got2 = []
            assert H.hwlm_exec(t, corpus, 3 if total > 3 else 0, lambda e, i, c: got2.append((e, i)) or H.HWLM_CONTINUE_MATCHING, s) == H.HWLM_SUCCESS
            assert sorted(got2) == sorted(oracle.collect(corpus, 3 if total > 3 else 0)), (name, total, "start")
            off = random_blocks(rng, total, mean_len=max(2, min(300, total // 3 + 1)))
            assert as_set(hw.hwlm_exec_batch(t, s, corpus, off)) == as_set(oracle.collect_blocks(corpus, off)) == as_set(hw.hwlm_exec_batch(t, plain, corpus, off))
        calls, launches, live = s.server_stats()
        if name != "mixed3000":  # (a table with 1- and 2-byte literals fills the LDS to the last byte: no room for the server's mailbox, its calls are launches)
            assert calls - served_before >= 30, (name, calls)
        served_before = calls
        t.close()  # (the next table: the server of this one is ended first)
    calls, launches, live = s.server_stats()
    assert launches <= 4 * len(sets), f"{launches} server launches for {calls} calls: it should stay resident across a burst"
    # idle: it ends by itself, and the next call brings it back
    lits, _ = sets["teddy64"]
    t = H.hwlm_build(lits)
    pkt = random_corpus(rng, 1460, lits, plant_every=97)
    want = sorted(ob.Oracle(lits).collect(pkt))

    def once():
        g = []
        assert H.hwlm_exec(t, pkt, 0, lambda e, i, c: g.append((e, i)) or H.HWLM_CONTINUE_MATCHING, s) == H.HWLM_SUCCESS
        assert sorted(g) == want
    once()
    assert s.server_stats()[2]
    time.sleep(0.05)
    l0 = s.server_stats()
    assert not l0[2], "the server did not end after its idle time"
    once()
    assert s.server_stats()[1] == l0[1] + 1 and s.server_stats()[2]
    # a large scan on the same scratch (its buffers are the server's): the server is ended first, and comes back afterwards
    big = random_corpus(rng, 3 << 20, lits, plant_every=500)
    one = np.array([0, big.size], dtype=np.uint64)
    assert as_set(hw.hwlm_exec_batch(t, s, big, one)) == as_set(ob.Oracle(lits).collect_blocks(big, one))
    assert not s.server_stats()[2]
    once()
    # another pipeline
    s.set_tuning(1)
    once()
    s.set_tuning(0)
    once()
    assert s.server_stats()[2]
    s.close()  # with the server resident
    plain.close()


@pytest.mark.parametrize("tables", [True, False])
@pytest.mark.parametrize("short", [False, True])
@pytest.mark.parametrize("start", [0, 5])
def test_flood_runs_replicated_from_one_confirm(short, start, tables):
    """The dense kernel's run shortcut (scan_device.h, `uni`; the reference: src/fdr/flood_runtime.h:86-335): a batch of chunks
    that are all one byte value, inside one block, is confirmed at ONE position; its other positions' records are a descriptor in
    the region's run table, which goes on over the following batches of the same run and is expanded by record_sort_kernel
    (`tables`; without them -- hsgpu_scratch_set_tuning(s, 5) -- the confirm kernel writes the records from that one position's).
    Runs of several byte values and lengths, blocks cut inside runs at odd offsets (the shortcut must stop short of a
    block's first bytes), literals that match only at a run's head (`xaaaa`), several literals per window, `start` > 0, stride-1
    tables (a 3-byte literal in the set) and stride-2 ones: every record against the oracle, in delivery order."""
    from tests import oracle_binding as ob
    from tests.util import as_set

    rng = np.random.default_rng(7 + short)
    lits = []
    for c in b"abz":
        lits += [H.HwlmLiteral(bytes([c]) * 4, id=len(lits)), H.HwlmLiteral(bytes([c]) * 8, id=len(lits) + 1),
                 H.HwlmLiteral(b"x" + bytes([c]) * 4, id=len(lits) + 2), H.HwlmLiteral(bytes([c]) * 5, nocase=True, id=len(lits) + 3)]
    if short:
        lits += [H.HwlmLiteral(b"aaa", id=len(lits)), H.HwlmLiteral(b"zz", id=len(lits) + 1)]
    lits += [H.HwlmLiteral(l.s, nocase=l.nocase, id=len(lits) + i) for i, l in enumerate(random_literals(rng, 60, 4, 8))]
    total = 3 << 20
    corpus = random_corpus(rng, total, lits, plant_every=3000)
    cuts = {0, total}
    pos = 1000
    for k in range(40):  # runs of 3 KiB .. 200 KiB of one value, some preceded by an 'x', with quiet stretches between them
        ln = int(rng.integers(3 << 10, 200 << 10))
        if pos + ln + 5000 > total:
            break
        v = b"abzAq"[k % 5]
        corpus[pos:pos + ln] = v
        if k % 3 == 0:
            corpus[pos - 1] = ord("x")
        for _ in range(int(rng.integers(0, 4))):  # block cuts inside the run, at any offset
            cuts.add(pos + int(rng.integers(1, ln)))
        cuts.add(pos + int(rng.integers(-20, 20)))
        pos += ln + int(rng.integers(100, 30000))
    off = np.array(sorted(cuts), dtype=np.uint64)
    t = H.hwlm_build(lits)
    s = H.Scratch(0)
    if not tables:
        s.set_tuning(5)
    want = ob.Oracle(lits).collect_blocks(corpus, off, start=start)
    for rep in range(3):  # the first scan says "again" and the scratch goes dense; the following ones run dense
        got = hw.hwlm_exec_batch(t, s, corpus, off, start=start)
        assert len(got) == len(want), (rep, len(got), len(want))
        assert as_set(got) == as_set(want), rep
        key = (got["block"].astype(np.uint64) << np.uint64(32)) | got["end"].astype(np.uint64)
        assert np.all((key[1:] > key[:-1]) | ((key[1:] == key[:-1]) & (got["lit"][1:] > got["lit"][:-1]))), "delivery order"
    s.close()


def test_flood_run_tables_full():
    """More runs in one part of the dense confirm stage than a region's run table holds (HSGPU_RUN_MAX = 3; the further ones are
    staged record by record), runs that go on over many batches and over several parts, runs of a byte that is a candidate
    everywhere and matches nothing: 512 MiB on the smallest launch geometry (1 024 filter wavefronts: shares of 512 KiB, parts
    of 16 KiB), quiet but for 2 MiB of runs -- every record against the oracle, in delivery order."""
    from tests import oracle_binding as ob
    from tests.util import as_set

    rng = np.random.default_rng(99)
    lits = []
    for c in b"abz":
        lits += [H.HwlmLiteral(bytes([c]) * 4, id=len(lits)), H.HwlmLiteral(bytes([c]) * 8, id=len(lits) + 1)]
    lits += [H.HwlmLiteral(b"qqqqqqqx", id=len(lits))]  # a run of q: candidates at every position, no match
    lits += [H.HwlmLiteral(l.s, nocase=l.nocase, id=len(lits) + i) for i, l in enumerate(random_literals(rng, 40, 5, 8))]
    total = 512 << 20
    corpus = np.full(total, ord("."), dtype=np.uint8)
    for at in rng.integers(0, total - 16, 2000):  # a few ordinary matches all over
        l = lits[int(rng.integers(7, len(lits)))].s
        corpus[at:at + len(l)] = np.frombuffer(l, dtype=np.uint8)
    pos = (100 << 20) + 777
    for k in range(96):  # 4 .. 7 KiB runs back to back, the values in turn: several runs per 16 KiB part
        ln = int(rng.integers(4 << 10, 7 << 10))
        corpus[pos:pos + ln] = b"abzq"[k % 4]
        pos += ln
    corpus[(200 << 20) + 5:(201 << 20) + 900] = ord("a")  # one run over ~64 parts
    corpus[(300 << 20):(300 << 20) + (256 << 10)] = ord("q")
    cuts = {0, total, (100 << 20) + 20000, (200 << 20) + 300000, (200 << 20) + 300001}
    cuts |= {int(x) for x in rng.integers(0, total, 300)}
    off = np.array(sorted(cuts), dtype=np.uint64)
    t = H.hwlm_build(lits)
    s = H.Scratch(0)
    s.set_tuning(0, 256, 1)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    for rep in range(3):
        got = hw.hwlm_exec_batch(t, s, corpus, off)
        assert len(got) == len(want), (rep, len(got), len(want))
        assert as_set(got) == as_set(want), rep
        key = (got["block"].astype(np.uint64) << np.uint64(32)) | got["end"].astype(np.uint64)
        assert np.all((key[1:] > key[:-1]) | ((key[1:] == key[:-1]) & (got["lit"][1:] > got["lit"][:-1]))), "delivery order"
    s.close()


def test_small_batch_server_mailbox_switches():
    """One scratch, the request mailbox switched between device memory (through the BAR) and mapped host memory and back with calls
    in between, and across an idle exit: no request is lost, none is served twice (the sequence numbers of the two mailboxes are
    brought together when a server starts: runtime.hip, server_start)."""
    import time

    from tests import oracle_binding as ob

    rng = np.random.default_rng(5)
    lits = random_literals(rng, 200, 4, 8)
    t = H.hwlm_build(lits)
    s = H.Scratch(0)
    oracle = ob.Oracle(lits)
    served = 0
    for rnd, kind in enumerate([1, 2, 1, 1, 2, 2, 1]):
        s.enable_server(kind, idle_us=400)
        for k in range(7):
            pkt = random_corpus(rng, int(rng.integers(16, 4000)), lits, plant_every=61)  # (a workgroup of 256 threads serves up to 4 KiB)
            g = []
            assert H.hwlm_exec(t, pkt, 0, lambda e, i, c: g.append((e, i)) or H.HWLM_CONTINUE_MATCHING, s) == H.HWLM_SUCCESS
            assert sorted(g) == sorted(oracle.collect(pkt)), (rnd, kind, k)
            served += 1
            if k == 3 and rnd % 2:
                time.sleep(0.01)  # the server ends by itself; the next call starts another
        assert s.server_stats()[0] == served
    s.close()
    t.close()


@pytest.mark.parametrize("mailbox", [1, 2])
def test_small_batch_server_answer_line(mailbox):
    """The server's answer line (scan_kernels.h, HsgpuServerCtl): a request with up to three records gets them back WITH its
    sequence number in one 64-byte store; four and more go through the mapped area. Packets with exactly 0 .. 6 matches, several
    hundred requests in a row with every packet new (a stale or torn line would deliver the records of the request before), one
    block per call and a few blocks per call, `start` > 0, and the stage stamps switched on and off in between -- each against the
    oracle, ends and ids in delivery order."""
    from tests import oracle_binding as ob

    rng = np.random.default_rng(77 + mailbox)
    lits = [H.HwlmLiteral(b"needle", id=11), H.HwlmLiteral(b"HayStack", nocase=True, id=12), H.HwlmLiteral(b"zq7", id=13)]
    lits += random_literals(rng, 300, 5, 8)
    t = H.hwlm_build(lits)
    oracle = ob.Oracle(lits)
    s = H.Scratch(0)
    s.enable_server(mailbox, idle_us=2000)
    lib = t._lib
    lib.hsgpu_debug_server_stamping.argtypes = [C.c_void_p, C.c_int]
    alphabet = np.frombuffer(b"0123456789 .,;:-_", dtype=np.uint8)
    plant = [b"needle", b"haystack", b"zq7", b"HAYSTACK", b"needle"]
    seen = {k: 0 for k in range(7)}
    for rnd in range(420):
        if rnd % 70 == 0:
            lib.hsgpu_debug_server_stamping(s._h, (rnd // 70) & 1)
        n = int(rng.integers(40, 1500))
        pkt = rng.choice(alphabet, n).astype(np.uint8)
        k = int(rng.integers(0, 7))
        for j in range(k):  # k planted literals on slots of their own (one may still be cut by the packet's end: counted below)
            p = (n // max(k, 1)) * j
            w = np.frombuffer(plant[(rnd + j) % len(plant)], dtype=np.uint8)
            if p + w.size <= n:
                pkt[p:p + w.size] = w
        start = int(rng.integers(0, 9)) if rnd % 5 == 0 else 0
        want = oracle.collect(pkt, start)
        got = []
        assert H.hwlm_exec(t, pkt, start, lambda e, i, c: got.append((e, i)) or H.HWLM_CONTINUE_MATCHING, s) == H.HWLM_SUCCESS
        assert [e for e, _ in got] == sorted(e for e, _ in got), (rnd, "delivery order")
        assert sorted(got) == sorted(want), (rnd, k, n, start)
        seen[min(len(want), 6)] += 1
        if rnd % 9 == 0:  # a few blocks per call: the staged path, the same answer line
            from tests.util import as_set, random_blocks
            off = random_blocks(rng, n, mean_len=max(2, n // 3))
            assert as_set(hw.hwlm_exec_batch(t, s, pkt, off)) == as_set(oracle.collect_blocks(pkt, off)), (rnd, "batch")
    calls, launches, _live = s.server_stats()
    assert calls >= 420 and launches <= 8, (calls, launches)
    assert all(seen[k] >= 10 for k in range(5)), seen  # (both sides of the line's three records)
    s.close()
    t.close()
