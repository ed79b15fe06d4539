"""GPU parity: the HIP path, called through the C ABI, against the oracle on the
same seeded inputs. Bit-exact: identical multisets of (block, end, id)."""
import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import corpus as cp
from hyperscan_amd import hwlm as hw
from tests import oracle_binding as ob
from tests.util import as_set, random_blocks, random_corpus, random_literals

pytestmark = pytest.mark.gpu

FORCE_REPL, FORCE_HASHED, FORCE_K2, FORCE_K1, FORCE_S1, FORCE_BLIND, FORCE_S2 = 1, 2, 4, 8, 16, 32, 64
FORCE_PAIR, FORCE_SMALL = 1024, 128


def gpu_collect(table, scratch, buf, start=0, groups=H.HWLM_ALL_GROUPS):
    out = []

    def cb(end, lit_id, _ctx):
        out.append((end, lit_id))
        return H.HWLM_CONTINUE_MATCHING

    rv = H.hwlm_exec(table, buf, start, cb, scratch, groups)
    assert rv == H.HWLM_SUCCESS
    return out


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


@pytest.mark.parametrize("flags", [FORCE_REPL | FORCE_K1 | FORCE_S1, FORCE_REPL | FORCE_K2 | FORCE_S2,
                                   FORCE_HASHED | FORCE_K1 | FORCE_S1, FORCE_HASHED | FORCE_K2 | FORCE_S2,
                                   FORCE_HASHED | FORCE_K2 | FORCE_S2 | FORCE_BLIND, FORCE_REPL | FORCE_BLIND])
@pytest.mark.parametrize("nlits,lo,hi,seed", [(16, 2, 8, 11), (200, 1, 8, 12), (2000, 3, 8, 13)])
def test_random_sets_forced_engines(scratch, flags, nlits, lo, hi, seed):
    rng = np.random.default_rng(seed)
    lits = random_literals(rng, nlits, lo, hi, nocase_frac=0.4)
    corpus = random_corpus(rng, 400_000, lits, plant_every=300)
    off = random_blocks(rng, corpus.size, mean_len=300)
    t = H.hwlm_build(lits, flags)
    got = hw.hwlm_exec_batch(t, scratch, corpus, off)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    assert as_set(got) == as_set(want)
    key = got["block"].astype(np.uint64) << np.uint64(32) | got["end"].astype(np.uint64)
    assert np.all(key[1:] >= key[:-1]), "records must arrive sorted by (block, end)"


@pytest.mark.parametrize("flags", [FORCE_PAIR, FORCE_PAIR | FORCE_SMALL])
@pytest.mark.parametrize("nlits,lo,hi,seed", [(40, 3, 8, 14), (600, 3, 8, 15), (3000, 3, 8, 16), (500, 4, 8, 17)])
def test_pair_filter_sets(scratch, flags, nlits, lo, hi, seed):
    """The pair filter (stride 2, 64-bit entries, 3-byte literals keyed one byte late at odd ends), forced
    for sets of every size it can hold; blocks cut at random so that both parities of every end offset occur."""
    rng = np.random.default_rng(seed)
    lits = random_literals(rng, nlits, lo, hi, nocase_frac=0.4)
    corpus = random_corpus(rng, 500_000, lits, plant_every=250)
    off = random_blocks(rng, corpus.size, mean_len=300)
    t = H.hwlm_build(lits, flags)
    assert t.info()["flags"] & 256
    got = hw.hwlm_exec_batch(t, scratch, corpus, off)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    assert as_set(got) == as_set(want)


def test_pair_filter_fdr10k_set_and_dense_candidates(scratch):
    lits, _full = cp.snort_like_literals(10000, seed=4)
    corpus, off = cp.packet_corpus(8 << 20, lits, seed=10)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    t = H.hwlm_build(lits, FORCE_PAIR)
    assert as_set(hw.hwlm_exec_batch(t, scratch, corpus, off)) == as_set(want)
    # a corpus made of one literal over and over (dense candidates; the fused pipeline is covered through HSGPU_MODE below)
    lits2 = [H.HwlmLiteral("abcd", False, 0), H.HwlmLiteral("bcda", False, 1), H.HwlmLiteral("cda", False, 2)]
    corpus2 = np.frombuffer(b"abcd" * 60_000, dtype=np.uint8)
    one = np.array([0, corpus2.size], dtype=np.uint64)
    got2 = hw.hwlm_exec_batch(H.hwlm_build(lits2, FORCE_PAIR), scratch, corpus2, one)
    assert as_set(got2) == as_set(ob.Oracle(lits2).collect_blocks(corpus2, one))


@pytest.mark.parametrize("tune", [0, 1, 2, 3, 4])
def test_pair_filter_short_literal_at_the_corpus_end_and_at_tile_edges(tune):
    """Round 6 (found by tests/test_gpu_guard_pages.py): the pair filter keys a 3-byte literal whose end has the other parity
    one byte LATE -- it is found at the lookup position behind its end. Behind the corpus' last byte there is none, and the
    first position of the partial last tile may belong to another wavefront: such ends are asked for explicitly
    (pair_edge_probe). Every pipeline, ends at total - 1 of both parities, at the last whole tile's last byte, one block / cut."""
    rng = np.random.default_rng(600 + tune)
    lits = random_literals(rng, 60, 3, 8, nocase_frac=0.3)
    shorts = [l for l in lits if len(l.s) == 3]
    assert len(shorts) >= 5
    t = H.hwlm_build(lits, FORCE_PAIR)
    s = H.Scratch(0)
    s.set_tuning(tune)
    oracle = ob.Oracle(lits)
    for total in [3, 4, 16, 17, 1023, 1024, 1025, 1026, 2047, 2048, 2051, 16384, 16385, 65536, 65537, 200_000, 200_001]:
        if tune == 4 and total > 100_000:
            continue
        for k, lit in enumerate(shorts[:4]):
            corpus = random_corpus(rng, total, lits, plant_every=200)
            b = np.frombuffer(lit.s, dtype=np.uint8)
            corpus[total - 3:total] = b                     # ends at the corpus' last byte
            if total > 2048:
                e = (total >> 10 << 10) - (k & 1)           # ... at the last whole tile's last byte, and one before
                corpus[e - 3:e] = b
                corpus[1024 - 3 + (k & 1):1024 + (k & 1)] = b
            for off in (np.array([0, total], dtype=np.uint64), np.array([0, total // 2, total // 2, total], dtype=np.uint64)):
                got = hw.hwlm_exec_batch(t, s, corpus, off)
                want = oracle.collect_blocks(corpus, off)
                assert as_set(got) == as_set(want), (tune, total, lit.s, len(got), len(want))
    s.close()


@pytest.mark.parametrize("total", [1, 15, 16, 17, 1023, 1024, 16383, 16384, 16385, 49152 + 5, 3 * 16384])
def test_sizes_around_tile_boundaries(scratch, total):
    rng = np.random.default_rng(total)
    lits = random_literals(rng, 40, 1, 8)
    corpus = random_corpus(rng, total, lits, plant_every=64)
    t = H.hwlm_build(lits)
    assert sorted(gpu_collect(t, scratch, corpus)) == sorted(ob.Oracle(lits).collect(corpus))


def test_empty_and_tiny_blocks(scratch):
    rng = np.random.default_rng(21)
    lits = random_literals(rng, 30, 1, 4)
    corpus = random_corpus(rng, 20_000, lits, plant_every=32)
    lens = rng.choice([0, 0, 1, 2, 3, 5, 8, 33, 70], 4000)
    off = np.concatenate([[0], np.cumsum(lens)])
    off = off[off <= corpus.size]
    off = np.unique(np.concatenate([off, [corpus.size]]))  # ends exactly at the corpus end
    off = np.sort(np.concatenate([off, off[5:50]])).astype(np.uint64)  # re-insert duplicates = empty blocks
    t = H.hwlm_build(lits)
    got = hw.hwlm_exec_batch(t, scratch, corpus, off)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    assert as_set(got) == as_set(want)


def test_many_literals_per_key_and_duplicate_ids(scratch):
    # literals sharing suffixes and ids (one key -> long literal list; ids may repeat)
    base = [b"xabcd", b"yabcd", b"zzabcd", b"abcd", b"Abcd", b"bcd", b"cd", b"d"]
    lits = [H.HwlmLiteral(s, nocase=(i % 2 == 1), id=i // 2) for i, s in enumerate(base * 4)]
    corpus = np.frombuffer(b"..xabcd..yAbCd..zzabcd..ABCD..d..cd.." * 300, dtype=np.uint8)
    t = H.hwlm_build(lits)
    assert sorted(gpu_collect(t, scratch, corpus)) == sorted(ob.Oracle(lits).collect(corpus))


def test_dense_matches_overflow_retry(scratch):
    # every byte matches several literals: the record buffer must grow and retry
    lits = [H.HwlmLiteral(b"a" * n, False, n) for n in range(1, 9)] + [H.HwlmLiteral("A", True, 100)]
    corpus = np.full(200_000, ord("a"), dtype=np.uint8)
    t = H.hwlm_build(lits)
    got = hw.hwlm_exec_batch(t, scratch, corpus, np.array([0, 100_000, 200_000], dtype=np.uint64), cap=1000)
    want = ob.Oracle(lits).collect_blocks(corpus, np.array([0, 100_000, 200_000], dtype=np.uint64))
    assert len(got) == len(want) == 2 * (9 * 100_000 - 28)
    assert as_set(got) == as_set(want)


def test_candidate_buffer_overflow_falls_back_to_fused(scratch):
    # a corpus where almost every chunk has a candidate overflows the two-phase
    # candidate regions; the fused fallback must produce the identical result
    lits = [H.HwlmLiteral("abcd", False, 0), H.HwlmLiteral("bcda", False, 1)]
    corpus = np.frombuffer(b"abcd" * 60_000, dtype=np.uint8)
    t = H.hwlm_build(lits)
    got = hw.hwlm_exec_batch(t, scratch, corpus, np.array([0, corpus.size], dtype=np.uint64))
    want = ob.Oracle(lits).collect_blocks(corpus, np.array([0, corpus.size], dtype=np.uint64))
    assert as_set(got) == as_set(want)


@pytest.mark.parametrize("flags", [FORCE_HASHED, FORCE_HASHED | 512, FORCE_HASHED | FORCE_S2, FORCE_HASHED | FORCE_K2,
                                   FORCE_HASHED | FORCE_K2 | FORCE_BLIND])
def test_few_three_byte_literals_folded_filter(scratch, flags):
    """HSGPU_F_BFOLD: 3-byte keys own their filter word and the filter kernel tests one class;
    512 = HSGPU_BUILD_NO_FOLD keeps the separate test. Same matches either way."""
    rng = np.random.default_rng(77)
    lits = random_literals(rng, 600, 4, 8, nocase_frac=0.3) + random_literals(rng, 25, 3, 3, nocase_frac=0.2)
    lits = [H.HwlmLiteral(l.s, nocase=l.nocase, id=i) for i, l in enumerate(lits)]
    corpus = random_corpus(rng, 600_000, lits, plant_every=200)
    off = random_blocks(rng, corpus.size, mean_len=700)
    t = H.hwlm_build(lits, flags)
    folded = bool(t.info()["flags"] & 128)
    if flags & 512:
        assert not folded
    else:  # folding is a stride-1 measure (stride-2 kernels are not VALU-bound)
        assert folded == (not flags & FORCE_S2)
    got = hw.hwlm_exec_batch(t, scratch, corpus, off)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    assert as_set(got) == as_set(want)


def test_folded_filter_overflow_falls_back_to_fused(scratch):
    lits = [H.HwlmLiteral("abcd", False, 0), H.HwlmLiteral("bcda", False, 1), H.HwlmLiteral("cda", False, 2)]
    corpus = np.frombuffer(b"abcd" * 60_000, dtype=np.uint8)
    t = H.hwlm_build(lits, FORCE_HASHED)
    assert t.info()["flags"] & 128
    got = hw.hwlm_exec_batch(t, scratch, corpus, np.array([0, corpus.size], dtype=np.uint64))
    want = ob.Oracle(lits).collect_blocks(corpus, np.array([0, corpus.size], dtype=np.uint64))
    assert as_set(got) == as_set(want)


def test_serialized_table_scans_identically(scratch):
    rng = np.random.default_rng(31)
    lits = random_literals(rng, 100, 2, 8)
    corpus = random_corpus(rng, 100_000, lits, plant_every=200)
    t = H.hwlm_build(lits)
    t2 = H.HwlmTable.deserialize(t.serialize())
    assert sorted(gpu_collect(t, scratch, corpus)) == sorted(gpu_collect(t2, scratch, corpus))


def test_workload_generators_parity(scratch):
    # the bench's own workloads at oracle-sized corpora
    for lits, seed in ((cp.teddy_literals(), 3), (cp.snort_like_literals(2000)[0], 10)):
        corpus, off = cp.packet_corpus(4 << 20, lits, seed=seed, match_every=2048)
        t = H.hwlm_build(lits)
        got = hw.hwlm_exec_batch(t, scratch, corpus, off)
        want = ob.Oracle(lits).collect_blocks(corpus, off)
        assert as_set(got) == as_set(want)


def _ref_or_skip():
    ob.require_ref()


def _recs_key(r):
    return np.lexsort((r["id"], r["end"], r["block"]))


def _same_records(got, want):
    """identical multisets of (block, end, id)"""
    assert len(got) == len(want)
    g, w = got[_recs_key(got)], want[_recs_key(want)]
    for f in ("block", "end", "id"):
        assert np.array_equal(g[f], w[f]), f


@pytest.mark.parametrize("nlits,lo,hi,seed", [(1, 4, 8, 41), (8, 2, 8, 42), (64, 4, 8, 43), (97, 3, 8, 44),
                                              (1000, 3, 8, 45), (10000, 3, 8, 46)])
def test_reference_parity_random_sets(scratch, nlits, lo, hi, seed):
    """The HIP path against the reference's own hwlmBuild / hwlmExec (oracle/_ref, compiled from the
    reference's sources), not against the restatement: literal counts on both sides of every engine
    switch of the reference (noodle / Teddy / fat Teddy / FDR strides and domains), 30% caseless."""
    _ref_or_skip()
    rng = np.random.default_rng(seed)
    lits = random_literals(rng, nlits, lo, hi, nocase_frac=0.3)
    corpus = random_corpus(rng, 600_000, lits, plant_every=300)
    off = random_blocks(rng, corpus.size, mean_len=500)
    got = hw.hwlm_exec_batch(H.hwlm_build(lits), scratch, corpus, off)
    _same_records(got, ob.Reference(lits).collect_blocks(corpus, off))


def test_reference_parity_msk_cmp_noruns_groups(scratch):
    """hwlmLiteral's other fields against the reference itself: msk/cmp beside and beyond the string,
    noruns, group masks, duplicate ids (unit/internal/fdr.cpp:549-555, 663-681; hwlm_literal.h:51-129)."""
    _ref_or_skip()
    rng = np.random.default_rng(47)
    lits = random_literals(rng, 120, 2, 8, nocase_frac=0.3)
    extra = [H.HwlmLiteral("bc", False, 300, msk=b"\xf0\xff\xff", cmp=b"\x30bc"),
             H.HwlmLiteral("xyz", True, 301, msk=b"\xff\x00\x00\x00", cmp=b"Q\x00\x00\x00"),
             H.HwlmLiteral("k", False, 302, msk=b"\xdf\x00\xff", cmp=b"A\x00k"),
             H.HwlmLiteral("aa", False, 303, noruns=True), H.HwlmLiteral("abab", True, 304, noruns=True),
             H.HwlmLiteral("zz9", False, 7), H.HwlmLiteral("q7", False, 7, groups=0x5)]
    lits = lits + extra
    corpus = random_corpus(rng, 300_000, lits, plant_every=150)
    corpus[1000:1400] = ord("a")
    corpus[5000:5400] = np.frombuffer(b"abAB" * 100, dtype=np.uint8)
    corpus[9000:9003] = np.frombuffer(b"Qxy", dtype=np.uint8)
    t = H.hwlm_build(lits)
    # whole-buffer exec: the sequential rules (noruns, groups) run in hsgpu_hwlm_replay
    for groups in (H.HWLM_ALL_GROUPS, 0x4, 0x2):
        got = gpu_collect(t, scratch, corpus, groups=groups)
        want = ob.Reference(lits).collect(corpus, groups=groups)
        assert sorted(got) == sorted(want), hex(groups)
        assert [e for e, _ in got] == sorted(e for e, _ in got), "callbacks in non-decreasing end"


def test_fdr10k_set_parity(scratch):
    """BASELINE config 3's literal set (10 000 snort-like literals, the set the bench headline is quoted
    on) over 16 MiB of its packet corpus: content-level identity with the reference, which runs it on
    FDR stride 1 / domain 15 (unit/internal/fdr.cpp:185-188 compares (end, id) lists the same way)."""
    _ref_or_skip()
    lits, _full = cp.snort_like_literals(10000, seed=4)
    corpus, off = cp.packet_corpus(16 << 20, lits, seed=10)
    t = H.hwlm_build(lits)
    got = hw.hwlm_exec_batch(t, scratch, corpus, off)
    ref = ob.Reference(lits)
    assert "fdr" in ref.info()
    _same_records(got, ref.collect_blocks(corpus, off))
    key = (got["block"].astype(np.uint64) << np.uint64(32)) | got["end"].astype(np.uint64)
    assert np.all(key[1:] >= key[:-1]), "delivery order"


def _scan_dev(t, scratch, corpus, off, cap):
    import torch

    dev = torch.device("cuda", 0)
    d_corpus = torch.from_numpy(np.ascontiguousarray(corpus)).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_out = torch.full((max(cap, 1) * 4,), -1, dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    hw.hwlm_scan_dev(t, scratch, d_corpus.data_ptr(), corpus.size, d_off.data_ptr(), off.size - 1, d_out.data_ptr(), cap,
                     d_count.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = int(d_count.item())
    return n, d_out.view(-1, 4).cpu().numpy().astype(np.uint32)


@pytest.mark.parametrize("case", ["sparse", "dense_lds", "dense_global", "mixed"])
def test_scan_dev_records_in_delivery_order(scratch, case):
    """hsgpu_hwlm_scan_dev delivers its records sorted by (block, end, literal index) -- hwlmExec's
    non-decreasing `end`, block by block (src/hwlm/hwlm.h:101-118, src/rose/match.c:396-476) -- with no
    host sort behind it: slices with a handful of records (ranked by counting), slices of hundreds
    (bitonic network in LDS) and of tens of thousands (in global memory)."""
    rng = np.random.default_rng(61)
    if case == "sparse":
        lits = random_literals(rng, 300, 3, 8)
        corpus = random_corpus(rng, 2_000_000, lits, plant_every=700)
    else:
        lits = [H.HwlmLiteral(b"a" * n, False, n) for n in (1, 2, 3)] + random_literals(rng, 50, 4, 8)
        lits = [H.HwlmLiteral(l.s, nocase=l.nocase, id=i) for i, l in enumerate(lits)]
        corpus = random_corpus(rng, 600_000, lits[3:], plant_every=900)
        corpus[corpus == ord("a")] = ord("b")
        if case == "dense_lds":      # runs of 300 x 'a': about 900 records in one 4 KiB slice
            for p in range(10_000, 500_000, 40_000):
                corpus[p:p + 300] = ord("a")
        elif case == "dense_global":  # 40 000 x 'a': 12 000 records per slice
            corpus[100_000:140_000] = ord("a")
        else:
            corpus[100_000:120_000] = ord("a")
            for p in range(200_000, 500_000, 30_000):
                corpus[p:p + 100] = ord("a")
    off = random_blocks(rng, corpus.size, mean_len=900)
    t = H.hwlm_build(lits)
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    cap = len(want) + 16
    n, recs = _scan_dev(t, scratch, corpus, off, cap)
    while n > cap:  # cap + 1: room enough in total, but a wavefront's staging region was too small for a dense run
        assert n == cap + 1
        cap *= 2
        n, recs = _scan_dev(t, scratch, corpus, off, cap)
    assert n == len(want)
    recs = recs[:n]
    lit_of = recs[:, 3]
    assert np.array_equal(np.array([lits[i].id for i in lit_of.tolist()], dtype=np.uint32), recs[:, 2])
    k1 = (recs[:, 0].astype(np.uint64) << np.uint64(32)) | recs[:, 1].astype(np.uint64)
    later = (k1[1:] > k1[:-1]) | ((k1[1:] == k1[:-1]) & (lit_of[1:] > lit_of[:-1]))
    assert later.all(), f"not in (block, end, lit) order at {np.flatnonzero(~later)[:5]}"
    got = np.zeros(n, dtype=want.dtype)
    got["block"], got["end"], got["id"] = recs[:, 0], recs[:, 1], recs[:, 2]
    _same_records(got, want)
    # a buffer that is too small: the exact count comes back and nothing is delivered
    n2, recs2 = _scan_dev(t, scratch, corpus, off, cap=max(1, len(want) // 2))
    assert n2 >= len(want) // 2 + 1  # (what the buffer holds then is unspecified, include/hsgpu.h; nothing is written past cap:
    # _scan_dev's buffer is exactly cap records long, and an out-of-bounds write would fault or corrupt the next tensor)


def test_scan_dev_resident_and_properties(scratch):
    """Device-resident path at a size the oracle cannot walk: size-independent
    properties -- identical results for two different block partitions of the
    same bytes wherever a match does not straddle a cut, idempotence, and count
    == number of records."""
    import torch

    rng = np.random.default_rng(51)
    lits = cp.teddy_literals()
    corpus, off = cp.packet_corpus(64 << 20, lits, seed=7, match_every=4096)
    t = H.hwlm_build(lits)
    dev = torch.device("cuda", 0)
    d_corpus = torch.from_numpy(corpus).to(dev)
    cap = 1 << 18
    d_out = torch.zeros(cap * 4, dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)

    def run(off_arr):
        d_off = torch.from_numpy(off_arr.view(np.int64)).to(dev)
        d_count.zero_()
        hw.hwlm_scan_dev(t, scratch, d_corpus.data_ptr(), corpus.size, d_off.data_ptr(), off_arr.size - 1,
                         d_out.data_ptr(), cap, d_count.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        n = int(d_count.item())
        assert n <= cap
        r = d_out[: n * 4].view(n, 4).cpu().numpy().astype(np.uint64)
        return r, d_off

    r1, _ = run(off)
    r2, _ = run(off)
    g1 = sorted((off[r1[:, 0].astype(np.int64)] + r1[:, 1]).tolist())
    assert g1 == sorted((off[r2[:, 0].astype(np.int64)] + r2[:, 1]).tolist()), "idempotence"
    one = np.array([0, corpus.size], dtype=np.uint64)
    r3, _ = run(one)
    # as ONE block nothing is cut: a superset; the extra matches are exactly those straddling a block start
    p1 = set(zip((off[r1[:, 0].astype(np.int64)] + r1[:, 1]).tolist(), r1[:, 3].tolist()))  # (corpus offset, literal)
    p3 = set(zip(r3[:, 1].tolist(), r3[:, 3].tolist()))
    assert p1 <= p3
    starts = off[:-1]
    sizes = np.array([len(l.s) for l in lits])
    n_straddling = 0
    for g, li in p3:
        b = int(np.searchsorted(starts, g, side="right") - 1)
        straddles = g - int(sizes[li]) + 1 < int(starts[b])
        n_straddling += straddles
        assert ((g, li) in p1) == (not straddles), (g, li, straddles)
    assert len(p1) > 1000 and n_straddling > 0


@pytest.mark.parametrize("env", [{"HSGPU_MODE": "fused"}, {"HSGPU_MODE": "unfolded"}, {"HSGPU_WG_THREADS": "1024"}, {"HSGPU_WG_THREADS": "512"}])
def test_golden_vectors_under_alternative_pipelines(env):
    """The golden-vector suite again with the always-correct fused pipeline (normally only the
    overflow fallback) and with each workgroup geometry forced for every table."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_golden.py"), "-q", "-x",
                          "-m", "gpu"], capture_output=True, text=True, env=dict(os.environ, **env), cwd=root, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
