"""Child process of tests/test_gpu_guard_pages.py (TEST INFRASTRUCTURE): runs one family of entry points with EVERY buffer
against an unmapped page -- the caller's buffers through hsgpu_debug_guard_malloc, the library's own through
hsgpu_debug_guard_mode (include/hsgpu_tuning.h, csrc/devmem.hip) -- and compares the results with the oracle.

  python -m tests.guard_worker <case> <front|back>

front: every buffer starts at the first mapped byte of its range (an access in FRONT of it faults);
back:  every buffer ends at the last mapped byte (an access BEHIND it faults; for a buffer whose size is no multiple of its
       required alignment the end is the last byte the alignment allows, < 16 bytes short).
A fault kills the process ("Memory access fault by GPU"); the parent reports the last progress line. Exit code 0 and a final
"GUARD-OK" line = every sub-case ran in bounds and equal to the oracle.

What the reference does about the same risk: zones copy a block's head and tail into a padded buffer (src/fdr/fdr.c:392-690),
vectoredLoad* likewise (src/fdr/teddy_runtime_common.h:126-391), unit/internal/fdr.cpp:496-561 scans at every alignment."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import hyperscan_amd as H  # noqa: E402
from hyperscan_amd import _native, accel  # noqa: E402
from hyperscan_amd import corpus as cp  # noqa: E402
from hyperscan_amd import hwlm as hw  # noqa: E402
from tests import class_seq_model as M  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402
from tests.util import as_set, do_accel_block_model, random_blocks, random_corpus, random_literals  # noqa: E402

LIB = _native.load_library()
LIB.hsgpu_debug_guard_mode.restype = C.c_int
LIB.hsgpu_debug_guard_mode.argtypes = [C.c_int]
LIB.hsgpu_debug_guard_malloc.restype = C.c_int
LIB.hsgpu_debug_guard_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_int]
LIB.hsgpu_debug_guard_free.restype = None
LIB.hsgpu_debug_guard_free.argtypes = [C.c_void_p]
LIB.hsgpu_debug_guard_probe.restype = C.c_int
LIB.hsgpu_debug_guard_probe.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
LIB.hsgpu_debug_guard_copy.restype = C.c_int
LIB.hsgpu_debug_guard_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
LIB.hsgpu_debug_guard_fill.restype = C.c_int
LIB.hsgpu_debug_guard_fill.argtypes = [C.c_void_p, C.c_int, C.c_size_t]

BACK = False


def say(*a):
    print(*a, flush=True)


class Mem:
    """guard-page device buffers of one sub-case, freed together"""

    def __init__(self):
        self.live = []

    def alloc(self, nbytes, align=16, fill=None):
        p = C.c_void_p()
        rv = LIB.hsgpu_debug_guard_malloc(C.byref(p), int(nbytes), align, 1 if BACK else 0)
        assert rv == 0, (rv, LIB.hsgpu_last_error())
        self.live.append(p.value)
        if fill is not None and nbytes:
            assert LIB.hsgpu_debug_guard_fill(p.value, fill, int(nbytes)) == 0
        return p.value

    def put(self, arr, align=16):
        a = np.ascontiguousarray(arr)
        p = self.alloc(a.nbytes, align)
        assert LIB.hsgpu_debug_guard_copy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p

    @staticmethod
    def get(ptr, dtype, count):
        out = np.zeros(count, dtype=dtype)
        assert LIB.hsgpu_debug_guard_copy(out.ctypes.data, ptr, out.nbytes, 0) == 0, LIB.hsgpu_last_error()
        return out

    def close(self):
        for p in self.live:
            LIB.hsgpu_debug_guard_free(p)
        self.live = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---- the literal scan -------------------------------------------------------------------------------------------------

PIPELINES = {"default": 0, "fused": 1, "unfolded": 2, "nosolo": 3, "solo": 4}
FORCE_PAIR, FORCE_SMALL, FORCE_MEDIUM, FORCE_BLOOM, FORCE_HASHED, FORCE_K2, FORCE_S1, NO_WIDE = 1024, 128, 256, 8192, 2, 4, 16, 2048


def literal_tables():
    rng = np.random.default_rng(77)
    sets = {
        "noodle1": ([H.HwlmLiteral(b"needle", nocase=True, id=5)], 0),
        "teddy64": (random_literals(rng, 64, 4, 8, nocase_frac=0.2), 0),
        "mixed3000": (random_literals(rng, 3000, 1, 8, nocase_frac=0.3), 0),
        "wide1500": (random_literals(rng, 1500, 4, 8, nocase_frac=0.0) + [H.HwlmLiteral(l.s, id=2000 + i) for i, l in
                                                                         enumerate(random_literals(rng, 20, 3, 3, nocase_frac=0.0))],
                     FORCE_HASHED | FORCE_K2 | FORCE_S1),
        "narrow1500": (random_literals(rng, 1500, 3, 8, nocase_frac=0.4), FORCE_HASHED | FORCE_K2 | FORCE_S1 | NO_WIDE),
        "pair600": (random_literals(rng, 600, 3, 8, nocase_frac=0.4), FORCE_PAIR),
        "small200": (random_literals(rng, 200, 2, 8, nocase_frac=0.3), FORCE_SMALL),
        "medium400": (random_literals(rng, 400, 3, 8, nocase_frac=0.3), FORCE_MEDIUM),
        "bloom2000": (random_literals(rng, 2000, 3, 8, nocase_frac=0.3), FORCE_BLOOM | FORCE_S1),
    }
    return sets


def scan_dev_once(mem, table, scratch, d_corpus, total, d_off, nblocks, start, cap):
    d_out = mem.alloc(cap * 16, 16, fill=0xEE)
    d_count = mem.alloc(8, 8, fill=0xEE)
    hw.hwlm_scan_dev(table, scratch, d_corpus, total, d_off, nblocks, d_out, cap, d_count, start=start)
    n = int(Mem.get(d_count, np.uint64, 1)[0])  # (the copy synchronises the device)
    recs = Mem.get(d_out, hw.MATCH_DTYPE, min(n, cap)) if n <= cap else None
    return n, recs


def scan_dev_guarded(table, scratch, corpus, off, start=0, first_cap=None):
    """hsgpu_hwlm_scan_dev with corpus, offsets, records and count in guard ranges; the "again" protocol of include/hsgpu.h; then
    once more with cap EXACTLY the count. -> records"""
    total, nblocks = int(corpus.size), int(off.size - 1)
    with Mem() as mem:
        d_corpus = mem.put(corpus, 16)
        d_off = mem.put(off.astype(np.uint64), 8)
        cap = first_cap if first_cap is not None else max(16, total // 64)
        for _ in range(40):
            n, recs = scan_dev_once(mem, table, scratch, d_corpus, total, d_off, nblocks, start, cap)
            if n <= cap:
                break
            cap = 2 * cap + 16 if n == cap + 1 else n
        else:
            raise AssertionError("no cap was ever enough")
        n2, recs2 = scan_dev_once(mem, table, scratch, d_corpus, total, d_off, nblocks, start, n)  # cap == count exactly
        # "again" (cap + 1) is a legal answer to an exact cap -- a wavefront's staging area is sized from cap, and a dense piece
        # of corpus may need several doublings --: then the exact buffer must simply not have been overrun (no fault)
        assert n2 in (n, n + 1), (n, n2)
        if n2 == n:
            assert recs2 is not None and np.array_equal(recs, recs2), (n, n2)
        return recs


def blocks_for(rng, total, shape):
    if shape == "one":
        return np.array([0, total], dtype=np.uint64)
    if shape == "ragged":  # empty blocks, tiny blocks, blocks cut at odd places
        off = random_blocks(rng, total, mean_len=max(2, min(300, total // 3 + 1)))
        off = np.sort(np.concatenate([off, off[1:4], [total]])).astype(np.uint64)
        return off
    raise AssertionError(shape)


SIZES_EXACT = [1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 47, 48, 64] + list(range(1016, 1033)) + list(range(16376, 16393))
SIZES_BIG = [65536, 65541, 200_003, (1 << 20) + 7, 3 << 20]


def case_literal(pipeline):
    tune = PIPELINES[pipeline]
    for name, (lits, flags) in literal_tables().items():
        table = H.hwlm_build(lits, flags)
        oracle = ob.Oracle(lits)
        rng = np.random.default_rng(len(name))
        for total in SIZES_EXACT + SIZES_BIG:
            if pipeline == "solo" and total > (1 << 20):
                continue
            corpus = random_corpus(rng, total, lits, plant_every=48 if total < 70000 else 400)
            for shape, start in (("one", 0), ("ragged", 0), ("ragged", 3)):
                if total in SIZES_BIG and shape == "ragged" and start == 0:
                    continue
                off = blocks_for(rng, total, shape)
                say(f"literal {pipeline} {name} total={total} blocks={shape} start={start}")
                scratch = H.Scratch(0)  # a fresh scratch: its buffers are sized for THIS scan, and end where their mapping ends
                scratch.set_tuning(tune)
                got = scan_dev_guarded(table, scratch, corpus, off, start)
                want = oracle.collect_blocks(corpus, off, start=start) if start else oracle.collect_blocks(corpus, off)
                assert as_set(got) == as_set(want), (name, total, shape, start, len(got), len(want))
                scratch.close()
        table.close()


def case_literal_dense():
    """the flood path: a scan that overflows its candidate regions says "again", the scratch goes dense (room for every chunk)"""
    lits = [H.HwlmLiteral(b"a", id=1), H.HwlmLiteral(b"aa", id=2), H.HwlmLiteral(b"aaaa", id=3), H.HwlmLiteral(b"bcd", id=4)]
    lits += random_literals(np.random.default_rng(3), 300, 3, 8)
    table = H.hwlm_build(lits)
    oracle = ob.Oracle(lits)
    for total in (16384, 70_001, 300_000 + 5, (1 << 20) + 16):
        rng = np.random.default_rng(total)
        corpus = random_corpus(rng, total, lits, plant_every=300)
        corpus[total // 4: total // 2] = ord("a")
        for shape in ("one", "ragged"):
            off = blocks_for(rng, total, shape)
            for tune in (0, 2):
                say(f"dense total={total} blocks={shape} tune={tune}")
                scratch = H.Scratch(0)
                scratch.set_tuning(tune)
                for rep in range(3):  # the first says "again" and goes dense, the following ones run dense
                    got = scan_dev_guarded(table, scratch, corpus, off, 0, first_cap=total // 64 + 16)
                    want = oracle.collect_blocks(corpus, off)
                    assert as_set(got) == as_set(want), (total, shape, tune, rep, len(got), len(want))
                scratch.close()


def case_host_entry():
    """hsgpu_hwlm_exec / _exec_batch / _exec_batch_cb: host buffers in, the library's own device buffers in guard ranges"""
    rng = np.random.default_rng(9)
    for name, (lits, flags) in literal_tables().items():
        if name not in ("noodle1", "teddy64", "mixed3000", "wide1500"):
            continue
        table = H.hwlm_build(lits, flags)
        oracle = ob.Oracle(lits)
        for total in [1, 7, 16, 33, 1023, 1460, 4096, 16385, 70_003, 300_001, 600_000, (2 << 20) + 3]:
            say(f"host {name} total={total}")
            corpus = random_corpus(rng, total, lits, plant_every=64 if total < 70000 else 500)
            scratch = H.Scratch(0)
            got = []
            rv = H.hwlm_exec(table, corpus, 0, lambda e, i, c: got.append((e, i)) or H.HWLM_CONTINUE_MATCHING, scratch)
            assert rv == H.HWLM_SUCCESS and sorted(got) == sorted(oracle.collect(corpus)), (name, total)
            off = blocks_for(rng, total, "ragged")
            want = as_set(oracle.collect_blocks(corpus, off))
            assert as_set(hw.hwlm_exec_batch(table, scratch, corpus, off)) == want
            assert as_set(hw.hwlm_exec_batch_pipelined(table, scratch, corpus, off, chunk_bytes=1 << 16)) == want
            scratch.close()
            if total <= 16385:  # the small-batch server: its staging copy of the batch, hints, regions and control block in guard ranges too
                srv = H.Scratch(0)
                srv.enable_server(True)
                for rep in range(2):
                    got = []
                    rv = H.hwlm_exec(table, corpus, 0, lambda e, i, c: got.append((e, i)) or H.HWLM_CONTINUE_MATCHING, srv)
                    assert rv == H.HWLM_SUCCESS and sorted(got) == sorted(oracle.collect(corpus)), (name, total, "server")
                    assert as_set(hw.hwlm_exec_batch(table, srv, corpus, off)) == want
                srv.close()
        table.close()


# ---- class scans ---------------------------------------------------------------------------------------------------------

CLASS_POOL = [bytes(range(ord("a"), ord("z") + 1)), bytes(range(ord("A"), ord("Z") + 1)), b"0123456789", b"0123456789abcdef",
              b" \t\r\n\x0b\x0c", bytes(range(128, 256)), b"aeiou", b",.;:",
              b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz_", b"()[]", b"-_", b"\n", b"xyz", b"q", b"019", b"Z"]


def oracle_bitmap(cls, buf):
    out = np.zeros((buf.size + 7) // 8, dtype=np.uint8)
    if buf.size:
        ob.hso().hso_class_bitmap(cls.bitmap.ctypes.data, buf.ctypes.data, buf.size, out.ctypes.data)
    return out


def class_scan_guarded(mem, classes, d_corpus, total, d_off, nblocks, first, last, bitmap_align=8):
    lib = accel._lib()
    n = len(classes)
    arr = (accel._Class * n)(*[c._to_c() for c in classes])
    bm_bytes = (total + 15) // 16 * 2
    bms = [mem.alloc(bm_bytes, bitmap_align, fill=0xEE) for _ in range(n)]
    ptrs = (C.c_void_p * n)(*bms)
    d_first = mem.alloc(4 * n * nblocks, 4, fill=0xEE) if first else None
    d_last = mem.alloc(4 * n * nblocks, 4, fill=0xEE) if last else None
    d_work = mem.alloc(accel.WORK_BYTES, 16, fill=0)
    rv = lib.hsgpu_class_scan_dev(arr, n, d_corpus, total, d_off, nblocks, ptrs, d_first, d_last, d_work, None)
    assert rv == 0, (rv, LIB.hsgpu_last_error())
    bm = [Mem.get(p, np.uint8, bm_bytes) for p in bms]
    fi = Mem.get(d_first, np.uint32, n * nblocks).reshape(n, nblocks) if first else None
    la = Mem.get(d_last, np.uint32, n * nblocks).reshape(n, nblocks) if last else None
    return bms, bm, fi, la


def case_class_scan():
    L = ob.hso()
    for total in [1, 15, 16, 17, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 16383, 16384, 16385, 100_003, (1 << 20) + 8]:
        rng = np.random.default_rng(total)
        corpus = rng.choice(np.frombuffer(b"abcxyzQZ 019\n_-,", np.uint8), total).astype(np.uint8)
        corpus[rng.integers(0, total, max(1, total // 50))] = rng.integers(128, 256, max(1, total // 50))
        for shape in ("one", "ragged"):
            off = blocks_for(rng, total, shape)
            nb = int(off.size - 1)
            for n_classes, fl in ((1, True), (8, True), (3, False), (12, False), (16, False)):
                say(f"class_scan total={total} blocks={shape} classes={n_classes} first_last={fl}")
                classes = [accel.CharClass(m) for m in CLASS_POOL[:n_classes]]
                with Mem() as mem:
                    d_corpus = mem.put(corpus, 16)
                    d_off = mem.put(off, 8)
                    _p, bm, fi, la = class_scan_guarded(mem, classes, d_corpus, total, d_off, nb, fl, fl)
                    for ci, cls in enumerate(classes):
                        want = oracle_bitmap(cls, corpus)
                        assert np.array_equal(bm[ci][: want.size], want), (total, shape, n_classes, ci)
                        if fl:
                            for b in range(nb):
                                blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
                                f = L.hso_class_fwd(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
                                r = L.hso_class_rev(cls.bitmap.ctypes.data, blk.ctypes.data, blk.size)
                                assert fi[ci, b] == f and la[ci, b] == (r & 0xFFFFFFFF), (total, shape, ci, b)


def case_pair_scan():
    lib = accel._lib()
    L = ob.hso()
    pairs = [accel.PairSet.from_dverm("a", "b"), accel.PairSet.from_dverm("Q", "z", True),
             accel.PairSet.build([(b"a", b"b"), (b"q", b"u"), (b"0", b"1")], accel.CharClass(b"Z")),
             accel.PairSet.from_dverm_masked(ord("a") & 0xdf, ord("0") & 0xf0, 0xdf, 0xf0)]
    for total in [1, 2, 15, 16, 17, 4095, 4096, 4097, 16384, 16385, 100_003]:
        rng = np.random.default_rng(total)
        corpus = rng.choice(np.frombuffer(b"abqu01ZQz x", np.uint8), total).astype(np.uint8)
        for shape in ("one", "ragged"):
            off = blocks_for(rng, total, shape)
            nb = int(off.size - 1)
            for n in (1, 4):
                say(f"pair_scan total={total} blocks={shape} pairs={n}")
                with Mem() as mem:
                    d_corpus = mem.put(corpus, 16)
                    d_off = mem.put(off, 8)
                    arr = (accel._Pair * n)(*[p._c for p in pairs[:n]])
                    bm_bytes = (total + 15) // 16 * 2
                    bms = [mem.alloc(bm_bytes, 8, fill=0xEE) for _ in range(n)]
                    ptrs = (C.c_void_p * n)(*bms)
                    d_first = mem.alloc(4 * n * nb, 4, fill=0xEE)
                    d_last = mem.alloc(4 * n * nb, 4, fill=0xEE)
                    d_work = mem.alloc(accel.PAIR_WORK_BYTES, 16, fill=0)
                    rv = lib.hsgpu_pair_scan_dev(arr, n, d_corpus, total, d_off, nb, ptrs, d_first, d_last, d_work, None)
                    assert rv == 0, (rv, LIB.hsgpu_last_error())
                    fi = Mem.get(d_first, np.uint32, n * nb).reshape(n, nb)
                    la = Mem.get(d_last, np.uint32, n * nb).reshape(n, nb)
                    for k in range(n):
                        m = pairs[k].masks
                        for b in range(nb):
                            blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
                            f = L.hso_dshufti_fwd(*m, blk.ctypes.data, blk.size)
                            r = L.hso_dshufti_rev(*m, blk.ctypes.data, blk.size)
                            assert fi[k, b] == f and la[k, b] == (r & 0xFFFFFFFF), (total, shape, k, b, int(fi[k, b]), f, int(la[k, b]), r)


def case_class_seq():
    lib = accel._lib()
    lib.hsgpu_class_seq_work_bytes.restype = C.c_size_t
    lib.hsgpu_class_seq_work_bytes.argtypes = [C.c_uint64]
    lib.hsgpu_class_seq_scan_dev.restype = C.c_int
    lib.hsgpu_class_seq_scan_dev.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]
    lib.hsgpu_class_seq_emit_dev.restype = C.c_int
    lib.hsgpu_class_seq_emit_dev.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p]
    classes_b = [b"abc", b"019", b"abcXYZ_", b"XYZ", b" \n", b"ab1"]
    for total in [1, 9, 63, 64, 65, 4031, 4032, 4033, 4096, 16384 + 3, 100_003, 300_000]:
        rng = np.random.default_rng(total)
        corpus = rng.choice(np.frombuffer(b"abcXYZ019 _-\n", np.uint8), total).astype(np.uint8)
        for shape in ("one", "ragged"):
            off = blocks_for(rng, total, shape)
            nb = int(off.size - 1)
            for n_seqs in (3, 70):
                seqs = [(int(rng.integers(0, 6)), int(rng.integers(0, 6)), int(rng.integers(1, 9)), int(rng.integers(1, 4)), 100 + k)
                        for k in range(n_seqs)]
                say(f"class_seq total={total} blocks={shape} seqs={n_seqs}")
                with Mem() as mem:
                    d_corpus = mem.put(corpus, 16)
                    d_off = mem.put(off, 8)
                    classes = [accel.CharClass(m) for m in classes_b]
                    bms, _bm, _f, _l = class_scan_guarded(mem, classes, d_corpus, total, d_off, nb, False, False)
                    arr = (accel._Seq * n_seqs)(*[accel._Seq(a, b, m, n, i) for a, b, m, n, i in seqs])
                    ptrs = (C.c_void_p * len(bms))(*bms)
                    wb = int(lib.hsgpu_class_seq_work_bytes(total))
                    d_work = mem.alloc(wb, 16, fill=0xEE)
                    d_counts = mem.alloc(8 * n_seqs, 8, fill=0xEE)
                    model = M.VecModel(corpus, off)
                    want = [model.ends(classes_b[a], classes_b[b], m, n) for a, b, m, n, _ in seqs]
                    n_all = sum(len(w) for w in want)
                    # counts + every record, cap exactly the number there is
                    d_out = mem.alloc(16 * n_all, 16, fill=0xEE)
                    d_count = mem.alloc(8, 8, fill=0xEE)
                    rv = lib.hsgpu_class_seq_scan_dev(arr, n_seqs, ptrs, len(bms), total, d_off, nb, 0, total, d_counts, d_out if n_all else None,
                                                      n_all, d_count, d_work, wb, None)
                    assert rv == 0, (rv, LIB.hsgpu_last_error())
                    counts = Mem.get(d_counts, np.uint64, n_seqs)
                    assert [int(c) for c in counts] == [len(w) for w in want], (total, shape)
                    assert int(Mem.get(d_count, np.uint64, 1)[0]) == n_all
                    recs = Mem.get(d_out, np.uint32, 4 * n_all).reshape(-1, 4)
                    for k in range(n_seqs):
                        g = recs[recs[:, 3] == k]
                        gs = g[np.lexsort((g[:, 1], g[:, 0]))][:, :2].astype(np.int64)
                        assert np.array_equal(gs, want[k]), (total, shape, k)
                    # the records of a range of whole blocks
                    if nb >= 3:
                        lo_b, hi_b = nb // 3, max(nb // 3 + 1, 2 * nb // 3)
                        lo, hi = int(off[lo_b]), int(off[hi_b])
                        sel = [w[(w[:, 0] >= lo_b) & (w[:, 0] < hi_b)] for w in want]
                        n_sel = sum(len(w) for w in sel)
                        d_out2 = mem.alloc(16 * n_sel, 16, fill=0xEE)
                        rv = lib.hsgpu_class_seq_emit_dev(arr, n_seqs, ptrs, len(bms), total, d_off, nb, lo, hi, d_out2 if n_sel else None, n_sel,
                                                          d_count, d_work, wb, None)
                        assert rv == 0, (rv, LIB.hsgpu_last_error())
                        assert int(Mem.get(d_count, np.uint64, 1)[0]) == n_sel, (total, shape)
                        recs = Mem.get(d_out2, np.uint32, 4 * n_sel).reshape(-1, 4)
                        for k in range(n_seqs):
                            g = recs[recs[:, 3] == k]
                            gs = g[np.lexsort((g[:, 1], g[:, 0]))][:, :2].astype(np.int64)
                            assert np.array_equal(gs, sel[k]), (total, shape, k, "emit")


def case_accel():
    """hsgpu_hwlm_forward_skip_dev (do_accel_block) and hsgpu_run_accel_dev (run_accel) with corpus, offsets, starts, results, bitmap
    and work area against unmapped pages; expectations: the oracle's accelerators (pinned to the reference in the CPU suite)"""
    lib = accel._lib()
    lib.hsgpu_run_accel_dev.restype = C.c_int
    lib.hsgpu_run_accel_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L = ob.hso()
    rng = np.random.default_rng(17)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789 \n", np.uint8)
    litsets = {"verm": [H.HwlmLiteral(b"qfoo", id=1), H.HwlmLiteral(b"qbar", id=2)],
               "shufti": [H.HwlmLiteral(s, id=i) for i, s in enumerate([b"abcd", b"efgh", b"ijkl", b"mnop", b"0123"])],
               "dverm": [H.HwlmLiteral(b"abxyz", id=1)]}
    for total_hint in (40, 700, 5000, 70_000):
        lens = np.concatenate([np.arange(0, 40), rng.integers(40, 400, max(1, total_hint // 200))])
        rng.shuffle(lens)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total = int(off[-1])
        nb = len(lens)
        corpus = rng.choice(alpha, total).astype(np.uint8)
        starts = np.minimum(rng.integers(0, 12, nb), lens).astype(np.uint32)
        for name, lits in litsets.items():
            fa = accel.ForwardAccel.choose(lits)
            sc = fa.scanner()
            for per_block in (False, True):
                say(f"forward_skip total={total} scheme={name} type={fa.type} per_block={per_block}")
                with Mem() as mem:
                    d_corpus = mem.put(corpus, 16)
                    d_off = mem.put(off, 8)
                    d_start = mem.put(starts, 4) if per_block else None
                    d_out = mem.alloc(4 * nb, 4, fill=0xEE)
                    d_bitmap = mem.alloc((total + 15) // 16 * 2, 8, fill=0xEE)
                    d_work = mem.alloc(accel.PAIR_WORK_BYTES, 16, fill=0)
                    rv = lib.hsgpu_hwlm_forward_skip_dev(C.byref(fa._c), d_corpus, total, d_off, nb, d_start, 0 if per_block else 2, d_out,
                                                         d_bitmap, d_work, None)
                    assert rv == 0, (rv, LIB.hsgpu_last_error())
                    got = Mem.get(d_out, np.uint32, nb)
                    for b in range(nb):
                        blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
                        st = int(starts[b]) if per_block else min(2, blk.size)
                        if sc is None:
                            want = st
                        else:
                            want = do_accel_block_model(L, sc[0], sc[1], fa.offset, blk, st)
                        if not per_block and blk.size < 2:
                            continue  # a common start beyond a shorter block: the caller's business
                        assert got[b] == want, (name, b, int(got[b]), want, blk.size, st)
        # run_accel: in-bounds only (value parity lives in tests/test_gpu_round3.py against the reference's run_accel)
        shufti_cls = accel.CharClass(b"aeiou0")
        lo, hi, _nb = shufti_cls.to_shufti()
        t1, t2 = accel.CharClass(bytes(range(0x30, 0x3a)) + b"\n xyzXYZ").to_truffle()
        pair = accel.PairSet.build([(b"a", b"b"), (b"q", b"u"), (b"0", b"1")], accel.CharClass(b"Z"))
        cases = [accel.AccelAux.make(accel.ACCEL_NONE), accel.AccelAux.make(accel.ACCEL_RED_TAPE, 3),
                 accel.AccelAux.make(accel.ACCEL_VERM, 2, ord("q")), accel.AccelAux.make(accel.ACCEL_VERM_NOCASE, 0, ord("E")),
                 accel.AccelAux.make(accel.ACCEL_DVERM, 1, ord("a"), ord("b")),
                 accel.AccelAux.make(accel.ACCEL_DVERM_NOCASE, 4, ord("A"), ord("B")),
                 accel.AccelAux.make(accel.ACCEL_DVERM_MASKED, 0, ord("a") & 0xdf, ord("0") & 0xf0, 0xdf, 0xf0),
                 accel.AccelAux.make(accel.ACCEL_SHUFTI, 5, masks=(lo, hi)), accel.AccelAux.make(accel.ACCEL_TRUFFLE, 1, masks=(t1, t2)),
                 accel.AccelAux.make(accel.ACCEL_DSHUFTI, 2, masks=pair.masks)]
        for aux in cases:
            for per_block in (False, True):
                say(f"run_accel total={total} type={aux.accel_type} per_block={per_block}")
                with Mem() as mem:
                    d_corpus = mem.put(corpus, 16)
                    d_off = mem.put(off, 8)
                    d_start = mem.put(starts, 4) if per_block else None
                    d_out = mem.alloc(4 * nb, 4, fill=0xEE)
                    d_bitmap = mem.alloc((total + 15) // 16 * 2, 8, fill=0xEE)
                    d_work = mem.alloc(accel.PAIR_WORK_BYTES, 16, fill=0)
                    rv = lib.hsgpu_run_accel_dev(C.byref(aux), d_corpus, total, d_off, nb, d_start, 0, d_out, d_bitmap, d_work, None)
                    assert rv == 0, (rv, LIB.hsgpu_last_error())
                    got = Mem.get(d_out, np.uint32, nb).astype(np.int64)
                    st = starts.astype(np.int64) if per_block else np.zeros(nb, np.int64)
                    assert np.all(got >= st) and np.all(got <= np.maximum(lens, st)), aux.accel_type


# ---- the exchange over the loopback transport -------------------------------------------------------------------------------------

def case_exchange():
    LIB.hsgpu_exchange_loopback_id.argtypes = [C.c_void_p]
    LIB.hsgpu_exchange_create.restype = C.c_int
    LIB.hsgpu_exchange_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint, C.c_int]
    LIB.hsgpu_exchange_step.restype = C.c_int
    LIB.hsgpu_exchange_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    LIB.hsgpu_exchange_compact.restype = C.c_int
    LIB.hsgpu_exchange_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    LIB.hsgpu_exchange_set_counts.restype = C.c_int
    LIB.hsgpu_exchange_set_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    LIB.hsgpu_exchange_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(4)
    for world in (2, 3, 8):
        for mode in (0, 1):
            for exact in (False, True):
                counts = [int(c) for c in rng.integers(0, 3000, world)]
                counts[world // 2] = 0
                rows = max(counts)  # a slot holds exactly the largest count
                say(f"exchange world={world} mode={mode} exact={exact} counts={counts}")
                with Mem() as mem:
                    ident = (C.c_uint8 * 128)()
                    assert LIB.hsgpu_exchange_loopback_id(ident) == 0
                    xs = []
                    for r in range(world):
                        x = C.c_void_p()
                        assert LIB.hsgpu_exchange_create(C.byref(x), ident, world, r, 0, max(1, rows), mode, 0) == 0, LIB.hsgpu_last_error()
                        xs.append(x)
                    if exact:
                        agreed = (C.c_uint64 * world)(*counts)
                        for x in xs:
                            assert LIB.hsgpu_exchange_set_counts(x, agreed, world) == 0
                    recs, bases = [], []
                    base = 0
                    for r in range(world):
                        a = np.zeros(counts[r], dtype=hw.MATCH_DTYPE)
                        a["block"] = np.sort(rng.integers(0, 500, counts[r]))
                        a["end"] = rng.integers(0, 1500, counts[r])
                        a["id"] = rng.integers(0, 9999, counts[r])
                        recs.append(a)
                        bases.append(base)
                        base += 500
                    for r in range(world):
                        d_rec = mem.put(recs[r], 16)  # cap == count: the record buffer ends with its last record
                        d_cnt = mem.put(np.array([counts[r]], dtype=np.uint64), 8)
                        assert LIB.hsgpu_exchange_step(xs[r], d_rec, counts[r], d_cnt, bases[r], None) == 0, LIB.hsgpu_last_error()
                    want = np.concatenate([np.stack([recs[r]["block"] + bases[r], recs[r]["end"], recs[r]["id"]], axis=1) for r in range(world)])
                    for r in range(world):
                        tot = sum(counts)
                        d_out = mem.alloc(12 * tot, 4, fill=0xEE)
                        got_counts = (C.c_uint64 * world)()
                        total = C.c_uint64()
                        rv = LIB.hsgpu_exchange_compact(xs[r], d_out, tot, got_counts, C.byref(total), None)
                        assert rv == 0, (rv, LIB.hsgpu_last_error())
                        if mode == 1 or r == 0:
                            assert list(got_counts) == counts and total.value == tot
                            out = Mem.get(d_out, np.uint32, 3 * tot).reshape(-1, 3)
                            assert np.array_equal(out, want.astype(np.uint32)), (world, mode, exact, r)
                        else:
                            assert total.value == 0
                    for x in xs:
                        LIB.hsgpu_exchange_free(x)


# ---- that the mechanism bites -----------------------------------------------------------------------------------------------------------

def case_probe(which):
    """one byte read / written by a kernel: inside a guard buffer (must succeed), one byte past its end, one byte in front of it (the
    process must die)"""
    with Mem() as mem:
        n = 4096 * 3
        p = mem.alloc(n, 16, fill=1)
        say(f"probe {which}")
        if which == "inside":
            assert LIB.hsgpu_debug_guard_probe(p, 0, 0) == 0 and LIB.hsgpu_debug_guard_probe(p, n - 1, 1) == 0
            return
        ofs, write = {"read_past_end": (n, 0), "write_past_end": (n, 1), "read_before_start": (-1, 0)}[which]
        rv = LIB.hsgpu_debug_guard_probe(p, ofs, write)
        say(f"probe returned {rv}: the access did NOT fault")
        sys.exit(0 if rv != 0 else 7)  # an error code from the runtime also counts as "caught"; 7 = silently succeeded


def main():
    global BACK
    case, mode = sys.argv[1], sys.argv[2]
    BACK = mode == "back"
    assert LIB.hsgpu_debug_guard_mode(2 if BACK else 1) == 0
    if case.startswith("literal:"):
        case_literal(case.split(":")[1])
    elif case.startswith("probe:"):
        case_probe(case.split(":")[1])
    else:
        {"literal_dense": case_literal_dense, "host_entry": case_host_entry, "class_scan": case_class_scan, "pair_scan": case_pair_scan,
         "class_seq": case_class_seq, "accel": case_accel, "exchange": case_exchange}[case]()
    say("GUARD-OK")


if __name__ == "__main__":
    main()
