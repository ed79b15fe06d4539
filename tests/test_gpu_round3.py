"""GPU tests of round 3: the consumer side of the device boundary (hsgpu_hwlm_fetch_replay), the chunked
host-buffer pipeline, dense inputs, the run_accel-shaped entry."""
import ctypes as C

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import corpus as cp
from hyperscan_amd import hwlm as hw
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _resident(lits, corpus, off, cap=None):
    import torch

    dev = torch.device("cuda", 0)
    t = H.hwlm_build(lits)
    s = H.Scratch(0)
    d_corpus = torch.from_numpy(np.concatenate([corpus, np.zeros(16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.astype(np.uint64).view(np.int64)).to(dev)
    cap = cap or max(1 << 12, corpus.size // 64)
    d_out = torch.zeros(cap * 4, dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    hw.hwlm_scan_dev(t, s, d_corpus.data_ptr(), int(corpus.size), d_off.data_ptr(), int(off.size - 1), d_out.data_ptr(), cap,
                     d_count.data_ptr(), 0, st)
    return t, s, d_out, d_count, cap, st, (d_corpus, d_off)


@pytest.mark.parametrize("mib,threads", [(1, 1), (8, 4), (32, 16)])
def test_fetch_replay_counts_what_the_reference_delivers(mib, threads):
    lits = cp.teddy_literals(64, seed=2)
    corpus, off = cp.packet_corpus(mib << 20, lits, seed=21, match_every=512)  # dense enough for the 4-chunk path at 32 MiB
    t, s, d_out, d_count, cap, st, keep = _resident(lits, corpus, off)
    n_rec, n_del = hw.hwlm_fetch_replay_count(t, s, d_out.data_ptr(), cap, d_count.data_ptr(), threads, st)
    want = ob.Oracle(lits).count_blocks(corpus, off)
    assert n_rec == want and n_del == want and want > 100
    # a second call on the same scratch (pool and pinned buffer reused), other thread count
    n_rec2, n_del2 = hw.hwlm_fetch_replay_count(t, s, d_out.data_ptr(), cap, d_count.data_ptr(), max(1, threads // 2), st)
    assert (n_rec2, n_del2) == (want, want)


def test_fetch_replay_sequential_rules_and_overflow():
    """noruns literals and groups go through the same per-block rules as hsgpu_hwlm_replay_batch; a scan whose
    records did not fit reports HSGPU_INSUFFICIENT_SPACE and delivers nothing."""
    rng = np.random.default_rng(5)
    base = cp.teddy_literals(40, seed=9)
    lits = [H.HwlmLiteral(l.s, False, i, noruns=bool(i % 3 == 0), groups=[H.HWLM_ALL_GROUPS, 0x1, 0x2][i % 3]) for i, l in enumerate(base)]
    corpus, off = cp.packet_corpus(4 << 20, lits, seed=22, match_every=256)
    t, s, d_out, d_count, cap, st, keep = _resident(lits, corpus, off)
    n = int(d_count.item())
    recs = d_out[: n * 4].view(n, 4).cpu().numpy().astype(np.uint32)
    for groups in (H.HWLM_ALL_GROUPS, 0x1, 0x2):
        want = hw.hwlm_replay_count(t, recs, groups)
        n_rec, n_del = hw.hwlm_fetch_replay_count(t, s, d_out.data_ptr(), cap, d_count.data_ptr(), 5, st, groups)
        assert n_rec == n and n_del == want and 0 < want <= n
    with pytest.raises(hw.HsgpuError):
        hw.hwlm_fetch_replay_count(t, s, d_out.data_ptr(), 16, d_count.data_ptr(), 2, st)


@pytest.mark.parametrize("chunk", [1 << 16, 1 << 20, 0])
def test_chunked_host_pipeline_equals_the_one_shot_scan(chunk):
    """hsgpu_hwlm_exec_batch_cb: the same records, in the same (delivery) order, whatever the chunk size; empty blocks,
    a block larger than the chunk, records handed over chunk by chunk in block order."""
    lits, _ = cp.snort_like_literals(600, seed=8)
    corpus, off = cp.packet_corpus(6 << 20, lits, seed=23, match_every=1024)
    off = np.sort(np.concatenate([off, off[10:14], off[-1:]]))  # empty blocks
    off = np.concatenate([off[off < (3 << 20)], off[off >= (3 << 20) + (300 << 10)]])  # one block of >= 300 KiB
    t = H.hwlm_build(lits)
    s = H.Scratch(0)
    want = hw.hwlm_exec_batch(t, s, corpus, off)
    chunks = []
    assert hw.hwlm_exec_batch_pipelined(t, s, corpus, off, 0, chunk, lambda r: chunks.append(r) and False) == 0
    got = np.concatenate(chunks)
    assert np.array_equal(got, want) and want.size > 1000
    firsts = [int(c["block"][0]) for c in chunks if c.size]
    assert firsts == sorted(firsts) and (chunk == 0 or len(chunks) > 3)
    # stopping early
    seen = []
    assert hw.hwlm_exec_batch_pipelined(t, s, corpus, off, 0, 1 << 18, lambda r: seen.append(r) or True) == -3
    assert len(seen) == 1
    # the scratch is free again and scans on
    assert np.array_equal(hw.hwlm_exec_batch(t, s, corpus, off), want)


def test_hs_scan_batch_takes_the_pipeline_for_large_batches():
    from hyperscan_amd import hs

    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    lits = sorted({bytes(rng.choice(alpha, int(rng.integers(5, 9)))) for _ in range(50)})
    pats = [l.decode() + r"[0-9]+" for l in lits]
    db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))
    scratch = hs.HsScratch(db)

    class L:
        def __init__(self, s):
            self.s = s
    import torch

    corpus, off = cp.packet_corpus(100 << 20, [L(l + b"42") for l in lits], seed=24, match_every=8192)
    keep = torch.from_numpy(corpus).pin_memory()  # the pipeline is taken for page-locked batches
    corpus = keep.numpy()
    events = []
    assert hs.scan_batch(db, corpus, off, scratch, lambda b, i, f, t: events.append((b, i, t)) and False) == hs.HS_SUCCESS
    # the same batch in two halves below the pipeline's threshold
    k = int(np.searchsorted(off, 50 << 20))
    ev2 = []
    hs.scan_batch(db, corpus, off[: k + 1], scratch, lambda b, i, f, t: ev2.append((b, i, t)) and False)
    hs.scan_batch(db, corpus, off[k:], scratch, lambda b, i, f, t: ev2.append((b + k, i, t)) and False)
    assert events == ev2 and len(events) > 10000
    blocks = [e[0] for e in events]
    assert blocks == sorted(blocks)


def test_run_accel_all_ten_cases_against_the_reference():
    """hsgpu_run_accel_dev against the reference's own run_accel (src/nfa/accel.c:35-146, exported from oracle/_ref
    by oracle/ref_build/run_accel_shim.c) on the same raw AccelAux bytes: every dispatched type, offsets, block
    lengths around the minimum-length thresholds (15 / 16 / 17 bytes after the start), per-block starts."""
    import torch

    from hyperscan_amd import accel

    ob.require_ref()
    R = ob.href(ob.ref_variants()[0])
    R.hsref_run_accel.restype = C.c_size_t
    R.hsref_run_accel.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    rng = np.random.default_rng(17)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789 \n", np.uint8)
    lens = np.concatenate([np.arange(0, 40), rng.integers(40, 400, 400)])
    rng.shuffle(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    corpus = rng.choice(alpha, int(off[-1])).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_corpus = torch.from_numpy(np.concatenate([corpus, np.zeros(16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    nb = len(lens)
    starts = np.minimum(rng.integers(0, 12, nb), lens).astype(np.int32)
    d_starts = torch.from_numpy(starts).to(dev)
    shufti_cls = accel.CharClass(b"aeiou0")
    lo, hi, _nb = shufti_cls.to_shufti()
    t1, t2 = accel.CharClass(bytes(range(0x30, 0x3a)) + b"\n xyzXYZ").to_truffle()
    pair = accel.PairSet.build([(b"a", b"b"), (b"q", b"u"), (b"0", b"1")], accel.CharClass(b"Z"))
    cases = [("none", accel.AccelAux.make(accel.ACCEL_NONE)),
             ("red_tape", accel.AccelAux.make(accel.ACCEL_RED_TAPE, 3)),
             ("verm", accel.AccelAux.make(accel.ACCEL_VERM, 2, ord("q"))),
             ("verm_nc", accel.AccelAux.make(accel.ACCEL_VERM_NOCASE, 0, ord("E"))),
             ("dverm", accel.AccelAux.make(accel.ACCEL_DVERM, 1, ord("a"), ord("b"))),
             ("dverm_nc", accel.AccelAux.make(accel.ACCEL_DVERM_NOCASE, 4, ord("A"), ord("B"))),
             ("dverm_masked", accel.AccelAux.make(accel.ACCEL_DVERM_MASKED, 0, ord("a") & 0xdf, ord("0") & 0xf0, 0xdf, 0xf0)),
             ("shufti", accel.AccelAux.make(accel.ACCEL_SHUFTI, 5, masks=(lo, hi))),
             ("truffle", accel.AccelAux.make(accel.ACCEL_TRUFFLE, 1, masks=(t1, t2))),
             ("dshufti", accel.AccelAux.make(accel.ACCEL_DSHUFTI, 2, masks=pair.masks))]
    for name, aux in cases:
        for st_arg, st_np in ((0, np.zeros(nb, np.int32)), (d_starts, starts)):
            got = accel.run_accel(aux, d_corpus, int(corpus.size), d_off, nb, st_arg).cpu().numpy()
            raw = bytes(aux)
            want = np.array([R.hsref_run_accel(raw, corpus[int(off[b]):].ctypes.data if lens[b] else corpus.ctypes.data,
                                               int(lens[b]), int(st_np[b])) for b in range(nb)], dtype=np.int64)
            if name == "dshufti":
                # the reference may stop earlier at a first-byte-only hit in the last lane of one of its vectors
                # (include/hsgpu.h, "two-byte accelerators"); never later, and equal where it saw a real pair
                assert np.all(got >= want), name
                lo1, hi1, _lo2, _hi2 = pair.masks
                exact = []
                for b in range(nb):
                    L, s0 = int(lens[b]), int(st_np[b])
                    blk = corpus[int(off[b]):int(off[b]) + L]
                    if s0 + 16 >= L:
                        exact.append(s0)
                        continue
                    end = L - 1
                    rv = next((i for i in range(s0, end - 1) if pair.test(int(blk[i]), int(blk[i + 1]))), None)
                    if rv is None:
                        c = int(blk[end - 1])
                        rv = end - 1 if (lo1[c & 15] | hi1[c >> 4]) != 0xff else end
                    exact.append(max(s0 + aux.offset, rv) - aux.offset)
                assert np.array_equal(got, np.array(exact)), name
            else:
                assert np.array_equal(got, want), (name, np.nonzero(got != want)[0][:5], got[got != want][:5], want[got != want][:5])


@pytest.mark.parametrize("short", [0, 20])
@pytest.mark.parametrize("nocase", [0.0, 0.4])
def test_wide_filter_layout_equals_the_oracle_and_the_narrow_layout(short, nocase):
    """HSGPU_F_WIDE (64-bit filter entries, the two bits of a 4-byte key in words of their own, tests that are plain shifts):
    chosen for large stride-1 two-bit sets; same records as the oracle and as the 32-bit layout (HSGPU_BUILD_NO_WIDE), with and
    without folded 3-byte keys, case-blind or not, through the two-phase pipeline and through the fused kernel."""
    from tests.util import as_set, random_blocks, random_corpus, random_literals

    FORCE_HASHED, FORCE_K2, FORCE_S1, NO_WIDE, F_WIDE, F_BFOLD = 2, 4, 16, 2048, 1024, 128
    rng = np.random.default_rng(90 + short)
    base = random_literals(rng, 1500, 4, 8, nocase_frac=nocase) + random_literals(rng, short, 3, 3, nocase_frac=nocase)
    lits = [H.HwlmLiteral(l.s, nocase=l.nocase, id=i) for i, l in enumerate(base)]
    corpus = random_corpus(rng, 700_000, lits, plant_every=150)
    off = random_blocks(rng, corpus.size, mean_len=500)
    want = as_set(ob.Oracle(lits).collect_blocks(corpus, off))
    wide = H.hwlm_build(lits, FORCE_HASHED | FORCE_K2 | FORCE_S1)
    narrow = H.hwlm_build(lits, FORCE_HASHED | FORCE_K2 | FORCE_S1 | NO_WIDE)
    assert wide.info()["flags"] & F_WIDE and not narrow.info()["flags"] & F_WIDE
    assert bool(wide.info()["flags"] & F_BFOLD) == bool(short)
    s = H.Scratch(0)
    got_w = hw.hwlm_exec_batch(wide, s, corpus, off)
    got_n = hw.hwlm_exec_batch(narrow, s, corpus, off)
    assert as_set(got_w) == want and as_set(got_n) == want and len(want) > 1500
    assert np.array_equal(got_w, got_n)  # the same records in the same (delivery) order
    assert np.array_equal(hw.hwlm_exec_batch(H.HwlmTable.deserialize(wide.serialize()), s, corpus, off), got_w)
    f = H.Scratch(0)
    f.set_tuning(fused_only=True)
    assert as_set(hw.hwlm_exec_batch(wide, f, corpus, off)) == want
