"""ctypes mirror of the public hs_* block-mode API served by libhsgpu.so (include/hs_gpu.h).
Same names, argument meaning and error codes as the reference's src/hs.h."""
import ctypes as C

import numpy as np

from . import _native

HS_SUCCESS, HS_INVALID, HS_NOMEM, HS_SCAN_TERMINATED, HS_COMPILER_ERROR = 0, -1, -2, -3, -4
HS_DB_VERSION_ERROR, HS_DB_PLATFORM_ERROR = -5, -6
HS_DB_MODE_ERROR, HS_SCRATCH_IN_USE, HS_UNKNOWN_ERROR = -7, -10, -13
HS_FLAG_CASELESS, HS_FLAG_DOTALL, HS_FLAG_MULTILINE, HS_FLAG_SINGLEMATCH = 1, 2, 4, 8
HS_FLAG_UTF8, HS_FLAG_SOM_LEFTMOST = 32, 256
HS_FLAG_ALLOWEMPTY, HS_FLAG_UCP, HS_FLAG_PREFILTER, HS_FLAG_COMBINATION, HS_FLAG_QUIET = 16, 64, 128, 512, 1024
HS_MODE_BLOCK, HS_MODE_STREAM, HS_MODE_VECTORED = 1, 2, 4
HS_EXT_FLAG_MIN_OFFSET, HS_EXT_FLAG_MAX_OFFSET, HS_EXT_FLAG_MIN_LENGTH = 1, 2, 4
HS_EXT_FLAG_EDIT_DISTANCE, HS_EXT_FLAG_HAMMING_DISTANCE = 8, 16


class ExprExt(C.Structure):
    """hs_expr_ext_t (src/hs_compile.h:244-310)"""
    _fields_ = [("flags", C.c_ulonglong), ("min_offset", C.c_ulonglong), ("max_offset", C.c_ulonglong),
                ("min_length", C.c_ulonglong), ("edit_distance", C.c_uint), ("hamming_distance", C.c_uint)]

    @classmethod
    def make(cls, min_offset=None, max_offset=None, min_length=None, edit_distance=None):
        e = cls()
        for flag, name, v in ((1, "min_offset", min_offset), (2, "max_offset", max_offset),
                              (4, "min_length", min_length), (8, "edit_distance", edit_distance)):
            if v is not None:
                e.flags |= flag
                setattr(e, name, v)
        return e


class ExprInfo(C.Structure):
    """hs_expr_info_t (src/hs_compile.h:160-236)"""
    _fields_ = [("min_width", C.c_uint), ("max_width", C.c_uint), ("unordered_matches", C.c_char),
                ("matches_at_eod", C.c_char), ("matches_only_at_eod", C.c_char)]


class CompileErrorStruct(C.Structure):
    _fields_ = [("message", C.c_char_p), ("expression", C.c_int)]


MATCH_CB = C.CFUNCTYPE(C.c_int, C.c_uint, C.c_ulonglong, C.c_ulonglong, C.c_uint, C.c_void_p)
BATCH_CB = C.CFUNCTYPE(C.c_int, C.c_ulonglong, C.c_uint, C.c_ulonglong, C.c_ulonglong, C.c_uint, C.c_void_p)


class HsError(RuntimeError):
    def __init__(self, code, message="", expression=-1):
        super().__init__(f"hs error {code}: {message} (expression {expression})")
        self.code, self.message, self.expression = code, message, expression


def _lib():
    lib = _native.load_library()
    if not getattr(lib, "_hs_sigs", False):
        P, E = C.POINTER, C.POINTER(C.POINTER(CompileErrorStruct))
        lib.hs_compile_multi.argtypes = [P(C.c_char_p), P(C.c_uint), P(C.c_uint), C.c_uint, C.c_uint, C.c_void_p,
                                         P(C.c_void_p), E]
        lib.hs_compile_lit_multi.argtypes = [P(C.c_char_p), P(C.c_uint), P(C.c_uint), P(C.c_size_t), C.c_uint,
                                             C.c_uint, C.c_void_p, P(C.c_void_p), E]
        lib.hs_free_compile_error.argtypes = [P(CompileErrorStruct)]
        lib.hs_free_database.argtypes = [C.c_void_p]
        lib.hs_alloc_scratch.argtypes = [C.c_void_p, P(C.c_void_p)]
        lib.hs_free_scratch.argtypes = [C.c_void_p]
        lib.hs_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, MATCH_CB, C.c_void_p]
        lib.hs_scan_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_uint, C.c_void_p,
                                      BATCH_CB, C.c_void_p]
        lib.hs_serialize_database.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_size_t)]
        lib.hs_deserialize_database.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p)]
        lib.hs_database_size.argtypes = [C.c_void_p, P(C.c_size_t)]
        lib.hs_version.restype = C.c_char_p
        lib.hs_compile_ext_multi.argtypes = [P(C.c_char_p), P(C.c_uint), P(C.c_uint), P(P(ExprExt)), C.c_uint, C.c_uint,
                                             C.c_void_p, P(C.c_void_p), E]
        lib.hs_expression_ext_info.argtypes = [C.c_char_p, C.c_uint, P(ExprExt), P(P(ExprInfo)), E]
        lib.hs_serialized_database_size.argtypes = [C.c_void_p, C.c_size_t, P(C.c_size_t)]
        lib.hs_serialized_database_info.argtypes = [C.c_void_p, C.c_size_t, P(C.c_char_p)]
        lib.hs_database_info.argtypes = [C.c_void_p, P(C.c_char_p)]
        lib._hs_sigs = True
    return lib


class Database:
    def __init__(self, handle):
        self._h = handle

    @classmethod
    def _compile(cls, fn_name, exprs, flags, ids, mode, lens=None, ext=None):
        lib = _lib()
        n = len(exprs)
        bufs = [e if isinstance(e, bytes) else e.encode("latin-1") for e in exprs]
        arr = (C.c_char_p * n)(*bufs)
        fl = (C.c_uint * n)(*(flags if flags is not None else [0] * n))
        idv = (C.c_uint * n)(*(ids if ids is not None else list(range(n))))
        db = C.c_void_p()
        err = C.POINTER(CompileErrorStruct)()
        if fn_name == "hs_compile_lit_multi":
            ln = (C.c_size_t * n)(*(lens if lens is not None else [len(b) for b in bufs]))
            rv = lib.hs_compile_lit_multi(arr, fl, idv, ln, n, mode, None, C.byref(db), C.byref(err))
        elif ext is not None:
            keep = [e for e in ext]
            ev = (C.POINTER(ExprExt) * n)(*[C.pointer(e) if e is not None else C.POINTER(ExprExt)() for e in keep])
            rv = lib.hs_compile_ext_multi(arr, fl, idv, ev, n, mode, None, C.byref(db), C.byref(err))
        else:
            rv = lib.hs_compile_multi(arr, fl, idv, n, mode, None, C.byref(db), C.byref(err))
        if rv != HS_SUCCESS:
            msg, ex = "", -1
            if err:
                msg, ex = err.contents.message.decode(errors="replace"), err.contents.expression
                lib.hs_free_compile_error(err)
            raise HsError(rv, msg, ex)
        return cls(db)

    @classmethod
    def compile(cls, exprs, flags=None, ids=None, mode=HS_MODE_BLOCK):
        return cls._compile("hs_compile_multi", exprs, flags, ids, mode)

    @classmethod
    def compile_ext(cls, exprs, flags=None, ids=None, ext=None, mode=HS_MODE_BLOCK):
        """hs_compile_ext_multi: ext = list of ExprExt or None per expression"""
        return cls._compile("hs_compile_ext_multi", exprs, flags, ids, mode, ext=ext if ext is not None else [None] * len(exprs))

    @classmethod
    def compile_lit(cls, exprs, flags=None, ids=None, mode=HS_MODE_BLOCK):
        return cls._compile("hs_compile_lit_multi", exprs, flags, ids, mode)

    def serialize(self):
        lib = _lib()
        p, n = C.c_void_p(), C.c_size_t()
        rv = lib.hs_serialize_database(self._h, C.byref(p), C.byref(n))
        if rv != 0:
            raise HsError(rv)
        out = C.string_at(p, n.value)
        C.CDLL(None).free(p)
        return out

    @classmethod
    def deserialize(cls, blob):
        db = C.c_void_p()
        rv = _lib().hs_deserialize_database(blob, len(blob), C.byref(db))
        if rv != 0:
            raise HsError(rv)
        return cls(db)

    def literals(self):
        """hs_database_literal for every branch: [(bytes, nocase, report id)] -- what the GPU
        matcher is keyed on, in the order of the literal ids it reports"""
        lib, out = _lib(), []
        lib.hs_database_literal.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_int), C.POINTER(C.c_uint)]
        while True:
            b, n, nc, rid = C.c_void_p(), C.c_size_t(), C.c_int(), C.c_uint()
            if lib.hs_database_literal(self._h, len(out), C.byref(b), C.byref(n), C.byref(nc), C.byref(rid)) != HS_SUCCESS:
                return out
            out.append((C.string_at(b, n.value), bool(nc.value), rid.value))

    def size(self):
        n = C.c_size_t()
        _lib().hs_database_size(self._h, C.byref(n))
        return n.value

    def info(self):
        """hs_database_info: "Version: ... Features: ... Mode: ..." """
        s = C.c_char_p()
        rv = _lib().hs_database_info(self._h, C.byref(s))
        if rv != HS_SUCCESS:
            raise HsError(rv, "hs_database_info")
        return s.value.decode()  # (the few bytes stay with the misc allocator, as a ctypes c_char_p cannot hand them back)

    def stream_size(self):
        """hs_stream_size -> error code (no database here is a streaming one)"""
        n = C.c_size_t()
        lib = _lib()
        lib.hs_stream_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        return lib.hs_stream_size(self._h, C.byref(n))

    def close(self):
        if self._h:
            _lib().hs_free_database(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HsScratch:
    def __init__(self, db):
        h = C.c_void_p()
        rv = _lib().hs_alloc_scratch(db._h, C.byref(h))
        if rv != 0:
            raise HsError(rv, "hs_alloc_scratch")
        self._h = h

    def close(self):
        if self._h:
            _lib().hs_free_scratch(self._h)
            self._h = None

    def enable_server(self, on=True, idle_us=0):
        """hs_scratch_enable_small_batch_server (include/hs_gpu.h): hs_scan / hs_scan_batch calls of up to 16 KiB served by one
        resident workgroup instead of a kernel launch per call (on = 2: the requests through mapped host memory)"""
        f = _lib().hs_scratch_enable_small_batch_server
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        rv = f(self._h, 2 if on == 2 else int(bool(on)), int(idle_us))
        if rv != 0:
            raise HsError(rv, "hs_scratch_enable_small_batch_server")

    def server_stats(self):
        """(calls served, server launches)"""
        f = _lib().hs_scratch_small_batch_server_stats
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        c, l = C.c_ulonglong(0), C.c_ulonglong(0)
        rv = f(self._h, C.byref(c), C.byref(l))
        if rv != 0:
            raise HsError(rv, "hs_scratch_small_batch_server_stats")
        return c.value, l.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scan(db, data, scratch, on_event=None):
    """hs_scan: on_event(id, from, to) -> truthy stops. Returns the hs_error_t."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    cb = MATCH_CB(lambda i, f, t, _fl, _c: 1 if (on_event and on_event(i, f, t)) else 0)
    return _lib().hs_scan(db._h, buf.ctypes.data if buf.size else b"", buf.size, 0, scratch._h, cb, None)


def scan_vector(db, segments, scratch, on_event=None):
    """hs_scan_vector over a list of bytes-like segments"""
    n = len(segments)
    keep = [bytes(s) for s in segments]
    arr = (C.c_char_p * n)(*keep)
    lens = (C.c_uint * n)(*[len(s) for s in keep])
    cb = MATCH_CB(lambda i, f, t, _fl, _c: 1 if (on_event and on_event(i, f, t)) else 0)
    lib = _lib()
    lib.hs_scan_vector.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_void_p, MATCH_CB,
                                   C.c_void_p]
    return lib.hs_scan_vector(db._h, arr, lens, n, 0, scratch._h, cb, None)


def scan_batch(db, data, off, scratch, on_event=None):
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    off = np.ascontiguousarray(off, dtype=np.uint64)
    cb = BATCH_CB(lambda b, i, f, t, _fl, _c: 1 if (on_event and on_event(b, i, f, t)) else 0)
    return _lib().hs_scan_batch(db._h, buf.ctypes.data, off.ctypes.data, off.size - 1, 0, scratch._h, cb, None)


def scan_batch_resident(db, data, off, d_corpus_ptr, d_off_ptr, scratch, on_event=None):
    """hs_scan_batch_resident (include/hs_gpu.h): the batch is already on the device (raw device pointers to the same bytes and
    offsets, as hsgpu_hwlm_scan_dev takes them); `data` / `off` are its host copy, which the confirm reads the bytes behind every
    literal hit from. Only the hits cross the bus."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    off = np.ascontiguousarray(off, dtype=np.uint64)
    cb = BATCH_CB(lambda b, i, f, t, _fl, _c: 1 if (on_event and on_event(b, i, f, t)) else 0)
    lib = _lib()
    lib.hs_scan_batch_resident.restype = C.c_int
    lib.hs_scan_batch_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p, BATCH_CB,
                                           C.c_void_p]
    return lib.hs_scan_batch_resident(db._h, buf.ctypes.data, off.ctypes.data, off.size - 1, d_corpus_ptr, d_off_ptr, scratch._h, cb, None)


def expression_info(expr, flags=0, ext=None):
    """hs_expression_ext_info -> (min_width, max_width)"""
    lib = _lib()
    info = C.POINTER(ExprInfo)()
    err = C.POINTER(CompileErrorStruct)()
    e = expr if isinstance(expr, bytes) else expr.encode("latin-1")
    rv = lib.hs_expression_ext_info(e, flags, C.byref(ext) if ext is not None else None, C.byref(info), C.byref(err))
    if rv != HS_SUCCESS:
        msg = err.contents.message.decode(errors="replace") if err else ""
        if err:
            lib.hs_free_compile_error(err)
        raise HsError(rv, msg, 0)
    out = (info.contents.min_width, info.contents.max_width)
    C.CDLL(None).free(info)
    return out
