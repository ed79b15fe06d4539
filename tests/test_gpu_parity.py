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


def test_reference_parity_when_available(scratch):
    if not ob.ref_available():
        pytest.skip("oracle/_ref not shipped")
    rng = np.random.default_rng(41)
    lits = random_literals(rng, 64, 4, 8)
    corpus = random_corpus(rng, 500_000, lits, plant_every=400)
    t = H.hwlm_build(lits)
    assert sorted(gpu_collect(t, scratch, corpus)) == sorted(ob.Reference(lits).collect(corpus))


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
    g3 = sorted(r3[:, 1].tolist())
    assert set(g1) <= set(g3)
    starts = off[:-1]
    sizes = np.array([len(l.s) for l in lits])
    for rec in r3:
        g = int(rec[1])
        b = int(np.searchsorted(starts, g, side="right") - 1)
        straddles = g - int(sizes[int(rec[3])]) + 1 < int(starts[b])
        assert (g in set(g1)) != straddles or not straddles
    assert len(g1) > 1000


@pytest.mark.parametrize("env", [{"HSGPU_MODE": "fused"}, {"HSGPU_WG_THREADS": "1024"}, {"HSGPU_WG_THREADS": "512"}])
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
