"""ctypes binding of lib/libhsgpu.so (the C ABI declared in include/hsgpu.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    # HSGPU_LIB_VARIANT: a tuning build of the same sources (csrc/Makefile VARIANT=...), tools/ only
    return os.path.join(_HERE, "lib", "libhsgpu%s.so" % os.environ.get("HSGPU_LIB_VARIANT", ""))


class HsgpuLit(C.Structure):
    _fields_ = [
        ("s", C.c_void_p), ("len", C.c_uint32), ("id", C.c_uint32), ("nocase", C.c_uint8),
        ("noruns", C.c_uint8), ("pad", C.c_uint8 * 2), ("msk_len", C.c_uint32), ("groups", C.c_uint64),
        ("msk", C.c_void_p), ("cmp", C.c_void_p),
    ]


class HsgpuMatch(C.Structure):
    _fields_ = [("block", C.c_uint32), ("end", C.c_uint32), ("id", C.c_uint32), ("lit", C.c_uint32)]


class HsgpuInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "n_lits", "n_class_a", "n_class_b", "n_class_c", "filter_words", "filter_entries",
        "ht_a_slots", "ht_b_slots", "max_size", "blob_bytes", "flags")]


HWLM_CB = C.CFUNCTYPE(C.c_uint64, C.c_size_t, C.c_uint32, C.c_void_p)

_SIGS = {
    "hsgpu_hwlm_build": (C.c_int, [C.POINTER(HsgpuLit), C.c_size_t, C.c_uint, C.POINTER(C.c_void_p)]),
    "hsgpu_hwlm_free": (None, [C.c_void_p]),
    "hsgpu_hwlm_size": (C.c_size_t, [C.c_void_p]),
    "hsgpu_hwlm_get_info": (C.c_int, [C.c_void_p, C.POINTER(HsgpuInfo)]),
    "hsgpu_hwlm_serialize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsgpu_hwlm_deserialize": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "hsgpu_scratch_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "hsgpu_scratch_free": (None, [C.c_void_p]),
    "hsgpu_hwlm_exec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, HWLM_CB, C.c_void_p,
                                  C.c_uint64]),
    "hsgpu_scratch_set_context": (None, [C.c_void_p, C.c_void_p]),
    "hsgpu_scratch_get_context": (C.c_void_p, [C.c_void_p]),
    "hsgpu_hwlm_exec_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsgpu_hwlm_scan_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "hsgpu_scratch_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "hsgpu_scratch_get_timing": (C.c_int, [C.c_void_p, C.c_uint, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float)]),
    "hsgpu_scratch_get_kernel_span": (C.c_int, [C.c_void_p, C.c_uint, C.POINTER(C.c_float)]),
    "hsgpu_scratch_get_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "hsgpu_hwlm_replay": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, HWLM_CB, C.c_void_p, C.c_uint64]),
    "hsgpu_hwlm_replay_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.POINTER(C.c_size_t)]),
    "hsgpu_hwlm_count_cb": (C.c_uint64, [C.c_size_t, C.c_uint32, C.c_void_p]),
    "hsgpu_hwlm_replay_batch_mt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint64,
                                             C.c_void_p]),
    "hsgpu_hwlm_fetch_replay": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_uint, C.c_uint64, C.POINTER(C.c_size_t), C.c_void_p]),
    "hsgpu_hwlm_exec_batch_cb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                           C.c_void_p, C.c_void_p]),
    "hsgpu_class_seq_work_bytes": (C.c_size_t, [C.c_uint64]),
    "hsgpu_class_seq_scan_dev": (C.c_int, [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]),
    "hsgpu_class_seq_exec_batch": (C.c_int, [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsgpu_last_error": (C.c_char_p, []),
    "hsgpu_version": (C.c_char_p, []),
    "hsgpu_source_hash": (C.c_char_p, []),
}


def exported_symbols():
    """Every entry point include/hsgpu.h declares (used by the symbol test)."""
    return sorted(_SIGS)


def load_library():
    """Load libhsgpu.so; fails loudly if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    try:
        # make sure the HIP runtime torch ships is the one already mapped, so device
        # pointers from torch tensors and our launches share one runtime instance
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
