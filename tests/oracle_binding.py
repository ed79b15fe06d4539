"""ctypes bindings of the CPU checkers (TEST INFRASTRUCTURE):

  Oracle     oracle/_build/libhso.so   -- the plain-C restatement (oracle/hwlm_oracle.c)
  Reference  oracle/_ref/libhsref.so   -- the reference's own code compiled in place
                                           (oracle/ref_build); optional
"""
import ctypes as C
import os

import numpy as np

from hyperscan_amd.hwlm import HWLM_ALL_GROUPS, HwlmLiteral, pack_literals  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CB = C.CFUNCTYPE(C.c_uint64, C.c_size_t, C.c_uint32, C.c_void_p)
REC = np.dtype([("block", "<u4"), ("end", "<u4"), ("id", "<u4")])


def _u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a, dtype=np.uint8)


_hso = None


def hso():
    global _hso
    if _hso is None:
        path = os.path.join(ROOT, "oracle", "_build", "libhso.so")
        if not os.path.exists(path):
            import __graft_entry__ as ge

            ge.build_oracle()
        L = C.CDLL(path)
        L.hso_build.restype = C.c_void_p
        L.hso_build.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.hso_free.argtypes = [C.c_void_p]
        L.hso_exec.restype = C.c_int
        L.hso_exec.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, CB, C.c_void_p, C.c_uint64]
        L.hso_collect.restype = C.c_size_t
        L.hso_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_size_t]
        L.hso_count_blocks.restype = C.c_uint64
        L.hso_count_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64]
        L.hso_collect_blocks.restype = C.c_size_t
        L.hso_collect_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        for name in ("hso_shufti_fwd", "hso_shufti_rev", "hso_truffle_fwd", "hso_truffle_rev"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        for name in ("hso_class_fwd", "hso_class_rev"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.hso_class_bitmap.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.hso_truffle_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("hso_verm_fwd", "hso_verm_rev"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_uint8, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.hso_dverm_fwd.restype = C.c_int64
        L.hso_dverm_fwd.argtypes = [C.c_uint8, C.c_uint8, C.c_int, C.c_void_p, C.c_size_t]
        L.hso_rdverm.restype = C.c_int64
        L.hso_rdverm.argtypes = [C.c_uint8, C.c_uint8, C.c_int, C.c_void_p, C.c_size_t]
        L.hso_rdverm_ref_model.restype = C.c_int64
        L.hso_rdverm_ref_model.argtypes = [C.c_uint8, C.c_uint8, C.c_int, C.c_void_p, C.c_size_t, C.c_uint]
        L.hso_dverm_masked_fwd.restype = C.c_int64
        L.hso_dverm_masked_fwd.argtypes = [C.c_uint8] * 4 + [C.c_void_p, C.c_size_t]
        for name in ("hso_dshufti_fwd", "hso_dshufti_rev"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t]
        L.hso_dshufti_ref_model.restype = C.c_int64
        L.hso_dshufti_ref_model.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint]
        L.hso_dshufti_bitmap.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]
        _hso = L
    return _hso


class Oracle:
    """The restated HWLM contract (oracle/hwlm_oracle.c)."""

    def __init__(self, lits, brute=False):
        self.L = hso()
        arr, self._keep = pack_literals(list(lits))
        self.h = self.L.hso_build(arr, len(arr), 1 if brute else 0)
        if not self.h:
            raise ValueError("oracle rejected the literal set")

    def __del__(self):
        try:
            self.L.hso_free(self.h)
        except Exception:
            pass

    def exec(self, buf, start, cb, groups=HWLM_ALL_GROUPS):
        a = _u8(buf)
        ccb = CB(lambda e, i, _c: int(cb(e, i)) & HWLM_ALL_GROUPS)
        return self.L.hso_exec(self.h, a.ctypes.data, a.size, start, ccb, None, groups)

    def collect(self, buf, start=0, groups=HWLM_ALL_GROUPS):
        a = _u8(buf)
        cap = 1 << 12
        while True:
            ends = np.zeros(cap, dtype=np.uint64)
            ids = np.zeros(cap, dtype=np.uint32)
            n = self.L.hso_collect(self.h, a.ctypes.data, a.size, start, groups, ends.ctypes.data,
                                   ids.ctypes.data, cap)
            if n <= cap:
                return list(zip(ends[:n].tolist(), ids[:n].tolist()))
            cap = n

    def count_blocks(self, base, off, start=0, groups=HWLM_ALL_GROUPS):
        a = _u8(base)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return self.L.hso_count_blocks(self.h, a.ctypes.data, off.ctypes.data, off.size - 1, start, groups)

    def collect_blocks(self, base, off, start=0, groups=HWLM_ALL_GROUPS):
        a = _u8(base)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        cap = 1 << 14
        while True:
            b = np.zeros(cap, dtype=np.uint32)
            e = np.zeros(cap, dtype=np.uint32)
            i = np.zeros(cap, dtype=np.uint32)
            n = self.L.hso_collect_blocks(self.h, a.ctypes.data, off.ctypes.data, off.size - 1, start, groups,
                                          b.ctypes.data, e.ctypes.data, i.ctypes.data, cap)
            if n <= cap:
                out = np.zeros(n, dtype=REC)
                out["block"], out["end"], out["id"] = b[:n], e[:n], i[:n]
                return out
            cap = n


_ref = {}


def ref_available():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libhsref.so"))


def require_ref(what="oracle/_ref/libhsref.so"):
    """The compiled reference is the strongest checker the suite has. Where it must be there -- on a GPU box (it travels with the
    snapshot like the product library) and wherever /root/reference exists (build() makes it) -- its absence FAILS the test; only a
    machine with neither may skip (round 5's verdict: a snapshot that lost the file lost the parity tests silently)."""
    import pytest

    if ref_available():
        return
    gpu = False
    try:
        import torch

        gpu = torch.cuda.is_available()
    except Exception:
        pass
    if gpu or os.path.isdir("/root/reference/src"):
        pytest.fail(f"{what} is missing: the reference-parity tests cannot run (python -c 'import __graft_entry__ as g; g.build()' "
                    "in the container builds it; it must travel to the GPU box)")
    pytest.skip("oracle/_ref not built and no reference tree here")


def _cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def ref_variants():
    """The builds of the reference this host can run: 'avx2' always (when built), 'avx512' when
    oracle/_ref/libhsref_avx512.so exists and /proc/cpuinfo lists every flag it was compiled
    for (oracle/ref_build/Makefile VARIANT=_avx512)."""
    out = []
    if ref_available():
        out.append("avx2")
    need = {"avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512cd", "avx512vbmi", "avx2", "bmi2"}
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libhsref_avx512.so")) and need <= _cpu_flags():
        out.append("avx512")
    return out


def href(variant="avx2"):
    if variant not in _ref:
        name = "libhsref.so" if variant == "avx2" else "libhsref_avx512.so"
        L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
        L.hsref_hwlm_build.restype = C.c_void_p
        L.hsref_hwlm_build.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int]
        L.hsref_hwlm_free.argtypes = [C.c_void_p]
        L.hsref_hwlm_info.restype = C.c_char_p
        L.hsref_hwlm_info.argtypes = [C.c_void_p]
        L.hsref_hwlm_exec.restype = C.c_int
        L.hsref_hwlm_exec.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, CB, C.c_void_p, C.c_uint64]
        L.hsref_hwlm_count_blocks.restype = C.c_uint64
        L.hsref_hwlm_count_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                              C.c_uint64]
        L.hsref_hwlm_collect_blocks.restype = C.c_size_t
        L.hsref_hwlm_collect_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.hsref_hwlm_bench_threads.restype = C.c_int
        L.hsref_hwlm_bench_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                               C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.hsref_build_isa.restype = C.c_char_p
        L.hsref_shufti_build.restype = C.c_int
        L.hsref_shufti_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hsref_truffle_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hsref_truffle2cr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("hsref_shufti_exec", "hsref_rshufti_exec", "hsref_truffle_exec", "hsref_rtruffle_exec"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        for name in ("hsref_verm_exec", "hsref_nverm_exec", "hsref_rverm_exec"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_uint8, C.c_int, C.c_void_p, C.c_size_t]
        L.hsref_dverm_exec.restype = C.c_int64
        L.hsref_dverm_exec.argtypes = [C.c_uint8, C.c_uint8, C.c_int, C.c_void_p, C.c_size_t]
        L.hsref_dshufti_build.restype = C.c_int
        L.hsref_dshufti_build.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
        L.hsref_dshufti_exec.restype = C.c_int64
        L.hsref_dshufti_exec.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t]
        L.hsref_dverm_masked_exec.restype = C.c_int64
        L.hsref_dverm_masked_exec.argtypes = [C.c_uint8] * 4 + [C.c_void_p, C.c_size_t]
        L.hsref_rdverm_exec.restype = C.c_int64
        L.hsref_rdverm_exec.argtypes = [C.c_uint8, C.c_uint8, C.c_int, C.c_void_p, C.c_size_t]
        L.hsref_forward_accel.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        if hasattr(L, "hsref_do_accel_block"):
            L.hsref_do_accel_block.restype = C.c_size_t
            L.hsref_do_accel_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.hsref_valid_engines.restype = C.c_size_t
        L.hsref_valid_engines.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        _ref[variant] = L
    return _ref[variant]


NO_HINT = 0xFFFFFFFF


class Reference:
    """The reference's own hwlmBuild/hwlmExec (or fdrExec with an engine hint)."""

    def __init__(self, lits, hint=NO_HINT, make_small=False, isa=0, variant="avx2"):
        self.L = href(variant)
        self.variant = variant
        arr, self._keep = pack_literals(list(lits))
        self.h = self.L.hsref_hwlm_build(arr, len(arr), 1 if make_small else 0, hint, isa)
        if not self.h:
            raise ValueError("reference could not build this literal set / engine")

    def __del__(self):
        try:
            self.L.hsref_hwlm_free(self.h)
        except Exception:
            pass

    def info(self):
        return self.L.hsref_hwlm_info(self.h).decode() + " isa=" + self.L.hsref_build_isa().decode()

    def collect_blocks(self, base, off, start=0, groups=HWLM_ALL_GROUPS, cap=1 << 16):
        """Every match as (block, end, id), in the reference's own delivery order. `off` may be a slice of a larger
        offset array (absolute offsets into `base`): block indices then count from the slice's first block. Thread safe
        (the collecting context is thread local in oracle/ref_build/ref_driver.cpp), and the GIL is released while it runs."""
        a = _u8(base)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        cap = max(16, int(cap))
        while True:
            b = np.zeros(cap, dtype=np.uint32)
            e = np.zeros(cap, dtype=np.uint32)
            i = np.zeros(cap, dtype=np.uint32)
            n = self.L.hsref_hwlm_collect_blocks(self.h, a.ctypes.data, off.ctypes.data, off.size - 1, start, groups,
                                                 b.ctypes.data, e.ctypes.data, i.ctypes.data, cap)
            if n <= cap:
                out = np.zeros(n, dtype=REC)
                out["block"], out["end"], out["id"] = b[:n], e[:n], i[:n]
                return out
            cap = n

    def bench_threads(self, base, off, threads, seconds, pin=True, start=0, groups=HWLM_ALL_GROUPS):
        """hsbench's thread model in native code: `threads` pinned pthreads, each looping over its own
        slice of the blocks for `seconds`. Returns (bytes scanned, wall seconds, matches of one pass,
        passes of the slowest thread)."""
        a = _u8(base)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        out = (C.c_double * 4)()
        rv = self.L.hsref_hwlm_bench_threads(self.h, a.ctypes.data, off.ctypes.data, off.size - 1, start, groups,
                                             int(threads), float(seconds), 1 if pin else 0, out)
        if rv != 0:
            raise RuntimeError("reference bench threads could not start")
        return float(out[0]), float(out[1]), int(out[2]), int(out[3])

    def exec(self, buf, start, cb, groups=HWLM_ALL_GROUPS):
        a = _u8(buf)
        ccb = CB(lambda e, i, _c: int(cb(e, i)) & HWLM_ALL_GROUPS)
        return self.L.hsref_hwlm_exec(self.h, a.ctypes.data, a.size, start, ccb, None, groups)

    def collect(self, buf, start=0, groups=HWLM_ALL_GROUPS):
        out = []

        def cb(e, i):
            out.append((e, i))
            return HWLM_ALL_GROUPS

        self.exec(buf, start, cb, groups)
        return out

    def count_blocks(self, base, off, start=0, groups=HWLM_ALL_GROUPS):
        a = _u8(base)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return self.L.hsref_hwlm_count_blocks(self.h, a.ctypes.data, off.ctypes.data, off.size - 1, start, groups)


def valid_engines(isa=0):
    L = href()
    buf = (C.c_uint32 * 64)()
    n = L.hsref_valid_engines(buf, 64, isa)
    return list(buf[:n])
