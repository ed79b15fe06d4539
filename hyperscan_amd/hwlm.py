"""Python mirror of the reference's HWLM interface.

  reference (C++ / C)                                      here
  hwlmLiteral(s, nocase, noruns, id, groups, msk, cmp)      HwlmLiteral(...)
      src/hwlm/hwlm_literal.h:51-129
  hwlmBuildProto + hwlmBuild -> bytecode_ptr<HWLM>          hwlm_build(lits) -> HwlmTable
      src/hwlm/hwlm_build.h:112-120
  hwlmSize(HWLM*)                                           hwlm_size(table)
  hwlmExec(tab, buf, len, start, cb, scratch, groups)       hwlm_exec(table, buf, start, cb, scratch, groups)
      src/hwlm/hwlm.h:116-118
Argument meaning, return values (HWLM_SUCCESS / HWLM_TERMINATED) and callback
protocol (return value = live group mask, 0 terminates) are the reference's.
"""
import ctypes as C
import os

import numpy as np

from . import _native
from ._native import HWLM_CB, HsgpuInfo, HsgpuLit, HsgpuMatch

HWLM_ALL_GROUPS = 0xFFFFFFFFFFFFFFFF
HWLM_CONTINUE_MATCHING = HWLM_ALL_GROUPS
HWLM_TERMINATE_MATCHING = 0
HWLM_SUCCESS = 0
HWLM_TERMINATED = 1
HWLM_ERROR_UNKNOWN = 2

MATCH_DTYPE = np.dtype([("block", "<u4"), ("end", "<u4"), ("id", "<u4"), ("lit", "<u4")])


class HsgpuError(RuntimeError):
    def __init__(self, code, what):
        lib = _native.load_library()
        msg = lib.hsgpu_last_error().decode(errors="replace")
        super().__init__(f"{what} failed with {code}: {msg}")
        self.code = code


class HwlmLiteral:
    """One literal; same fields as the reference's hwlmLiteral."""

    __slots__ = ("s", "nocase", "noruns", "id", "groups", "msk", "cmp")

    def __init__(self, s, nocase=False, id=0, noruns=False, groups=HWLM_ALL_GROUPS, msk=b"", cmp=b""):
        self.s = s.encode("latin-1") if isinstance(s, str) else bytes(s)
        self.nocase = bool(nocase)
        self.noruns = bool(noruns)
        self.id = int(id)
        self.groups = int(groups)
        self.msk = bytes(msk)
        self.cmp = bytes(cmp)

    def __repr__(self):
        return f"HwlmLiteral({self.s!r}, nocase={self.nocase}, id={self.id})"


def pack_literals(lits, struct=HsgpuLit):
    """-> (ctypes array, keepalive list) in the C ABI's hsgpu_lit_t layout."""
    arr = (struct * len(lits))()
    keep = []
    for i, l in enumerate(lits):
        sb = C.create_string_buffer(l.s, len(l.s)) if l.s else None
        mb = C.create_string_buffer(l.msk, len(l.msk)) if l.msk else None
        cb = C.create_string_buffer(l.cmp, len(l.cmp)) if l.cmp else None
        keep += [sb, mb, cb]
        arr[i].s = C.cast(sb, C.c_void_p) if sb else None
        arr[i].len = len(l.s)
        arr[i].id = l.id
        arr[i].nocase = 1 if l.nocase else 0
        arr[i].noruns = 1 if l.noruns else 0
        arr[i].msk_len = len(l.msk)
        arr[i].groups = l.groups
        arr[i].msk = C.cast(mb, C.c_void_p) if mb else None
        arr[i].cmp = C.cast(cb, C.c_void_p) if cb else None
    return arr, keep


class HwlmTable:
    """Compiled literal table (immutable; shareable between scratches)."""

    def __init__(self, handle):
        self._h = handle
        self._lib = _native.load_library()

    @classmethod
    def build(cls, lits, flags=0):
        lib = _native.load_library()
        arr, _keep = pack_literals(list(lits))
        h = C.c_void_p()
        rv = lib.hsgpu_hwlm_build(arr, len(arr), flags, C.byref(h))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_hwlm_build")
        return cls(h)

    @classmethod
    def deserialize(cls, blob):
        lib = _native.load_library()
        h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        rv = lib.hsgpu_hwlm_deserialize(buf, len(blob), C.byref(h))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_hwlm_deserialize")
        return cls(h)

    def serialize(self):
        n = C.c_size_t()
        self._lib.hsgpu_hwlm_serialize(self._h, None, 0, C.byref(n))
        buf = (C.c_char * n.value)()
        rv = self._lib.hsgpu_hwlm_serialize(self._h, buf, n.value, C.byref(n))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_hwlm_serialize")
        return bytes(buf)

    @property
    def size(self):
        return self._lib.hsgpu_hwlm_size(self._h)

    def info(self):
        i = HsgpuInfo()
        self._lib.hsgpu_hwlm_get_info(self._h, C.byref(i))
        return {n: getattr(i, n) for n, _ in HsgpuInfo._fields_}

    def close(self):
        if self._h:
            self._lib.hsgpu_hwlm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scratch:
    """Per-caller device state (stream + buffers); the role of hs_scratch."""

    def __init__(self, device=-1):
        self._lib = _native.load_library()
        h = C.c_void_p()
        rv = self._lib.hsgpu_scratch_alloc(C.byref(h), device)
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_alloc")
        self._h = h
        # tests and tuning runs select another pipeline / launch geometry for a whole process through the environment
        # (include/hsgpu_tuning.h; the C library itself reads no environment variables)
        env = os.environ
        if env.get("HSGPU_MODE") or env.get("HSGPU_WG_THREADS") or env.get("HSGPU_WG_PER_CU"):
            self.set_tuning({"fused": 1, "unfolded": 2, "no_skew": 5}.get(env.get("HSGPU_MODE"), 0), int(env.get("HSGPU_WG_THREADS", "0")),
                            int(env.get("HSGPU_WG_PER_CU", "0")))
        if env.get("HSGPU_MODE") in ("server", "server_host"):  # every small host batch of the process through the resident workgroup (the parity suite, forced)
            self.enable_server(2 if env.get("HSGPU_MODE") == "server_host" else True)

    def enable_server(self, on=True, idle_us=0):
        """the small-batch server (include/hsgpu.h, hsgpu_scratch_enable_server): hwlm_exec / hwlm_exec_batch calls of up to
        16 KiB are served by one resident workgroup instead of a kernel launch per call (on = 2: the requests through mapped host
        memory even where the host can write device memory)"""
        f = self._lib.hsgpu_scratch_enable_server
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        rv = f(self._h, 2 if on == 2 else int(bool(on)), int(idle_us))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_enable_server")

    def server_stats(self):
        """(requests served, server launches, resident right now)"""
        f = self._lib.hsgpu_scratch_server_stats
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        c, l, v = C.c_uint64(), C.c_uint64(), C.c_int()
        f(self._h, C.byref(c), C.byref(l), C.byref(v))
        return int(c.value), int(l.value), bool(v.value)

    def set_tuning(self, fused_only=False, wg_threads=0, wg_per_cu=0):
        self._lib.hsgpu_scratch_set_tuning.restype = C.c_int
        self._lib.hsgpu_scratch_set_tuning.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint]
        rv = self._lib.hsgpu_scratch_set_tuning(self._h, int(fused_only), wg_threads, wg_per_cu)  # 1 fused only, 2 unfolded
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_set_tuning")

    def enable_timing(self, on=True):
        """on = 2: also a set of device-clock stamps per workgroup of the filter kernel (wg_stamps)"""
        rv = self._lib.hsgpu_scratch_enable_timing(self._h, int(on))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_enable_timing")

    def wg_stamps(self):
        """[workgroups][4] ms from the earliest start of the last scan's filter kernel: start, prologue done,
        wavefront 0's share done, end (enable_timing(2))"""
        import numpy as np

        f = self._lib.hsgpu_scratch_get_wg_stamps
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
        n = C.c_uint(0)
        out = np.zeros((4096, 4), dtype=np.float32)
        rv = f(self._h, out.ctypes.data, 4096, C.byref(n))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_get_wg_stamps")
        return out[: min(n.value, 4096)]

    def conf_stamps(self):
        """[confirm workers][6]: start ms, end ms, fresh steps, rest steps, sorted drains, entries (enable_timing(2), a library
        built with -DHSGPU_CONFIRM_STAMPS=1)"""
        import numpy as np

        f = self._lib.hsgpu_scratch_get_conf_stamps
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
        n = C.c_uint(0)
        out = np.zeros((16384, 6), dtype=np.float32)
        rv = f(self._h, out.ctypes.data, 16384, C.byref(n))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_get_conf_stamps")
        return out[: min(n.value, 16384)]

    def timing(self, back=0):
        """(filter_ms, confirm_ms, total_ms) of the hwlm_scan_dev `back` launches ago
        on this scratch (0 = the last one; a ring of 32 is kept)."""
        f, c, t = C.c_float(), C.c_float(), C.c_float()
        rv = self._lib.hsgpu_scratch_get_timing(self._h, back, C.byref(f), C.byref(c), C.byref(t))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_get_timing")
        return f.value, c.value, t.value

    def kernel_span(self, back=0):
        """Filter-kernel execution span in ms from the device wall clock (no dispatch gaps)."""
        f = C.c_float()
        rv = self._lib.hsgpu_scratch_get_kernel_span(self._h, back, C.byref(f))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_get_kernel_span")
        return f.value

    def stats(self):
        """(candidate entries spilled, scans that overflowed) since the previous call -- synchronises."""
        n, o = C.c_uint64(), C.c_int()
        rv = self._lib.hsgpu_scratch_get_stats(self._h, C.byref(n), C.byref(o))
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_scratch_get_stats")
        return n.value, o.value

    def close(self):
        if self._h:
            self._lib.hsgpu_scratch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hwlm_build(lits, flags=0):
    return HwlmTable.build(lits, flags)


def hwlm_size(table):
    return table.size


def _as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.dtype != np.uint8 or not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def hwlm_exec(table, buf, start, cb, scratch, groups=HWLM_ALL_GROUPS, ctx=None):
    """Mirror of hwlmExec. cb(end, id, ctx) -> live group mask (0 terminates)."""
    a = _as_u8(buf)

    def tramp(end, lit_id, _):
        return int(cb(end, lit_id, ctx)) & HWLM_ALL_GROUPS

    ccb = HWLM_CB(tramp)
    return table._lib.hsgpu_hwlm_exec(table._h, a.ctypes.data, a.size, start, ccb, scratch._h, groups)


def hwlm_exec_batch(table, scratch, base, off, start=0, cap=None):
    """Batched host form: returns a structured array of (block, end, id, lit)
    sorted by (block, end, lit)."""
    a = _as_u8(base)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    nblocks = off.size - 1
    cap = int(cap) if cap is not None else max(4096, a.size // 64)
    lib = table._lib
    while True:
        out = np.zeros(cap, dtype=MATCH_DTYPE)
        n = C.c_size_t()
        rv = lib.hsgpu_hwlm_exec_batch(table._h, scratch._h, a.ctypes.data, off.ctypes.data, nblocks, start,
                                       out.ctypes.data, cap, C.byref(n))
        if rv == -12:  # HSGPU_INSUFFICIENT_SPACE
            cap = n.value
            continue
        if rv != 0:
            raise HsgpuError(rv, "hsgpu_hwlm_exec_batch")
        return out[: n.value]


def hwlm_replay_count(table, recs, groups=HWLM_ALL_GROUPS):
    """The sorted records of a batch (uint32 [n, 4] or MATCH_DTYPE array) through hsgpu_hwlm_replay_batch into
    the library's native counting callback -- hsbench's delivery path (engine_hyperscan.cpp:89-97) -- and the
    number of callbacks delivered."""
    a = np.ascontiguousarray(recs)
    n = a.shape[0]
    lib = table._lib
    cnt = C.c_uint64(0)
    rv = lib.hsgpu_hwlm_replay_batch(table._h, a.ctypes.data, n, C.cast(lib.hsgpu_hwlm_count_cb, C.c_void_p),
                                     C.cast(C.byref(cnt), C.c_void_p), groups, None)
    if rv != HWLM_SUCCESS:
        raise HsgpuError(rv, "hsgpu_hwlm_replay_batch")
    return int(cnt.value)


CHUNK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


def hwlm_exec_batch_pipelined(table, scratch, base, off, start=0, chunk_bytes=0, on_chunk=None):
    """hsgpu_hwlm_exec_batch_cb: the host-buffer scan as a pipeline of chunks (copy of chunk i + 1 beside the scan
    of chunk i; the records of each finished chunk handed over while later chunks are in flight).
    on_chunk(records) -> truthy stops; without it the records of all chunks are returned as one MATCH_DTYPE array."""
    a = _as_u8(base)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    got = []

    def cb(ptr, n, _ctx):
        recs = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n * 4,)).view(MATCH_DTYPE).copy() if n else \
            np.zeros(0, dtype=MATCH_DTYPE)
        if on_chunk is not None:
            return 1 if on_chunk(recs) else 0
        got.append(recs)
        return 0

    ccb = CHUNK_CB(cb)
    lib = table._lib
    lib.hsgpu_hwlm_exec_batch_cb.restype = C.c_int
    lib.hsgpu_hwlm_exec_batch_cb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                             CHUNK_CB, C.c_void_p]
    rv = lib.hsgpu_hwlm_exec_batch_cb(table._h, scratch._h, a.ctypes.data, off.ctypes.data, off.size - 1, start, chunk_bytes, ccb, None)
    if rv not in (0, -3):
        raise HsgpuError(rv, "hsgpu_hwlm_exec_batch_cb")
    if on_chunk is not None:
        return rv
    return np.concatenate(got) if got else np.zeros(0, dtype=MATCH_DTYPE)


def _thread_counters(threads):
    """one uint64 counter per thread, a cache line apart, and the array of their addresses"""
    cnt = np.zeros(threads * 8, dtype=np.uint64)
    ctxs = (C.c_void_p * threads)(*[cnt.ctypes.data + 64 * i for i in range(threads)])
    return cnt, ctxs


def hwlm_replay_count_mt(table, recs, threads, groups=HWLM_ALL_GROUPS):
    """hwlm_replay_count on `threads` host threads (hsgpu_hwlm_replay_batch_mt: contiguous block ranges, one per
    thread, a counter per thread) -> callbacks delivered."""
    a = np.ascontiguousarray(recs)
    lib = table._lib
    cnt, ctxs = _thread_counters(threads)
    lib.hsgpu_hwlm_replay_batch_mt.restype = C.c_int
    lib.hsgpu_hwlm_replay_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint64,
                                               C.c_void_p]
    rv = lib.hsgpu_hwlm_replay_batch_mt(table._h, a.ctypes.data, a.shape[0], C.cast(lib.hsgpu_hwlm_count_cb, C.c_void_p), ctxs,
                                        threads, groups, None)
    if rv != HWLM_SUCCESS:
        raise HsgpuError(rv, "hsgpu_hwlm_replay_batch_mt")
    return int(cnt.sum())


def hwlm_fetch_replay_count(table, scratch, out_ptr, cap, count_ptr, threads, stream=None, groups=HWLM_ALL_GROUPS):
    """hsgpu_hwlm_fetch_replay into the native counting callback: the records of a device-resident scan to pinned
    host memory in chunks, replayed on `threads` threads while the next chunk copies.
    -> (records the scan found, callbacks delivered)"""
    lib = table._lib
    cnt, ctxs = _thread_counters(threads)
    n_rec = C.c_size_t(0)
    lib.hsgpu_hwlm_fetch_replay.restype = C.c_int
    lib.hsgpu_hwlm_fetch_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_uint, C.c_uint64, C.POINTER(C.c_size_t), C.c_void_p]
    rv = lib.hsgpu_hwlm_fetch_replay(table._h, scratch._h, out_ptr, cap, count_ptr, stream,
                                     C.cast(lib.hsgpu_hwlm_count_cb, C.c_void_p), ctxs, threads, groups, C.byref(n_rec), None)
    if rv != 0:
        raise HsgpuError(rv, "hsgpu_hwlm_fetch_replay")
    return int(n_rec.value), int(cnt.sum())


def hwlm_scan_dev(table, scratch, corpus_ptr, total_bytes, off_ptr, nblocks, out_ptr, cap, count_ptr,
                  start=0, stream=None):
    """Device-resident hot path: raw device pointers in, asynchronous."""
    rv = table._lib.hsgpu_hwlm_scan_dev(table._h, scratch._h, corpus_ptr, total_bytes, off_ptr, nblocks, start,
                                        out_ptr, cap, count_ptr, stream)
    if rv != 0:
        raise HsgpuError(rv, "hsgpu_hwlm_scan_dev")
