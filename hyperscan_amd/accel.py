"""Python mirror of the reference's character-class accelerators, batched on the GPU.

  reference (C)                                              here
  shuftiExec(mask_lo, mask_hi, buf, buf_end)                  CharClass.from_shufti(lo, hi)  + class_scan(...)
      src/nfa/shufti.h:46-52                                      -> first[c][block]
  rshuftiExec / rtruffleExec / rvermicelliExec                -> last[c][block]
  truffleExec(mask1, mask2, buf, buf_end)                     CharClass.from_truffle(m1, m2)
      src/nfa/truffle.h:45-50
  vermicelliExec(c, nocase, buf, buf_end) / nvermicelliExec   CharClass.from_verm(c, nocase, negate)
      src/nfa/vermicelli.h:42-104
  truffleBuildMasks(CharReach) src/nfa/trufflecompile.cpp:59  CharClass(...).to_truffle()

  shuftiDoubleExec(lo1, hi1, lo2, hi2, buf, buf_end)          PairSet.from_dshufti(...) + pair_scan(...)
      src/nfa/shufti.c:319-361
  shuftiBuildDoubleMasks(onechar, twochar, ...)               PairSet.build(pairs, onechar)
      src/nfa/shufticompile.cpp:135-209
  vermicelliDoubleExec / ...MaskedExec / rvermicelliDoubleExec PairSet.from_dverm / from_dverm_masked
      src/nfa/vermicelli.h:169-317,464-518

Return convention as in the reference: offset of the first member in the block, the
block length when there is none; for the reverse scans the last member, -1 when none.
"""
import ctypes as C

import numpy as np

from . import _native
from .hwlm import HsgpuError

CLASS_MAX = 8           # classes per call with first / last
CLASS_MAX_BITMAPS = 16  # ... for the bitmaps alone: one read of the corpus
WORK_BYTES = 8192
PAIR_MAX = 8
PAIR_WORK_BYTES = 8256


class _Class(C.Structure):
    _fields_ = [("bitmap", C.c_uint8 * 32)]


class _Accel(C.Structure):
    _fields_ = [("type", C.c_uint8), ("offset", C.c_uint8), ("c1", C.c_uint8), ("c2", C.c_uint8),
                ("mask_lo", C.c_uint8 * 16), ("mask_hi", C.c_uint8 * 16)]


ACCEL_NONE, ACCEL_VERM, ACCEL_VERM_NOCASE, ACCEL_DVERM, ACCEL_DVERM_NOCASE = 0, 1, 2, 3, 4
ACCEL_SHUFTI, ACCEL_TRUFFLE = 13, 15  # enum AccelType, src/nfa/accel.h:46-65


class _Pair(C.Structure):
    _fields_ = [("lo1", C.c_uint8 * 16), ("hi1", C.c_uint8 * 16), ("lo2", C.c_uint8 * 16), ("hi2", C.c_uint8 * 16)]


def _lib():
    lib = _native.load_library()
    if not getattr(lib, "_class_sigs", False):
        lib.hsgpu_class_from_shufti.restype = C.c_int
        lib.hsgpu_class_from_shufti.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Class)]
        lib.hsgpu_class_from_truffle.restype = C.c_int
        lib.hsgpu_class_from_truffle.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Class)]
        lib.hsgpu_class_from_verm.restype = C.c_int
        lib.hsgpu_class_from_verm.argtypes = [C.c_uint8, C.c_int, C.c_int, C.POINTER(_Class)]
        lib.hsgpu_class_to_truffle.restype = C.c_int
        lib.hsgpu_class_to_truffle.argtypes = [C.POINTER(_Class), C.c_void_p, C.c_void_p]
        lib.hsgpu_class_scan_dev.restype = C.c_int
        lib.hsgpu_class_scan_dev.argtypes = [C.POINTER(_Class), C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
        lib.hsgpu_class_to_shufti.restype = C.c_int
        lib.hsgpu_class_to_shufti.argtypes = [C.POINTER(_Class), C.c_void_p, C.c_void_p]
        lib.hsgpu_accel_forward.restype = C.c_int
        lib.hsgpu_accel_forward.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(_Accel)]
        lib.hsgpu_pair_from_dshufti.restype = C.c_int
        lib.hsgpu_pair_from_dshufti.argtypes = [C.c_void_p] * 4 + [C.POINTER(_Pair)]
        lib.hsgpu_pair_from_dverm.restype = C.c_int
        lib.hsgpu_pair_from_dverm.argtypes = [C.c_uint8, C.c_uint8, C.c_int, C.POINTER(_Pair)]
        lib.hsgpu_pair_from_dverm_masked.restype = C.c_int
        lib.hsgpu_pair_from_dverm_masked.argtypes = [C.c_uint8] * 4 + [C.POINTER(_Pair)]
        lib.hsgpu_pair_build.restype = C.c_int
        lib.hsgpu_pair_build.argtypes = [C.POINTER(_Class), C.c_void_p, C.c_size_t, C.POINTER(_Pair)]
        lib.hsgpu_pair_test.restype = C.c_int
        lib.hsgpu_pair_test.argtypes = [C.POINTER(_Pair), C.c_uint8, C.c_uint8]
        lib.hsgpu_pair_scan_dev.restype = C.c_int
        lib.hsgpu_pair_scan_dev.argtypes = [C.POINTER(_Pair), C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p,
                                            C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]
        lib.hsgpu_hwlm_forward_skip_dev.restype = C.c_int
        lib.hsgpu_hwlm_forward_skip_dev.argtypes = [C.POINTER(_Accel), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                    C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
        lib._class_sigs = True
    return lib


class CharClass:
    """A 256-bit character class (the reference's CharReach)."""

    def __init__(self, members=()):
        self.bitmap = np.zeros(32, dtype=np.uint8)
        for v in members:
            v = v if isinstance(v, int) else ord(v)
            self.bitmap[v >> 3] |= 1 << (v & 7)

    @classmethod
    def _from_c(cls, c):
        o = cls()
        o.bitmap = np.frombuffer(bytes(c.bitmap), dtype=np.uint8).copy()
        return o

    def _to_c(self):
        c = _Class()
        C.memmove(c.bitmap, self.bitmap.ctypes.data, 32)
        return c

    @classmethod
    def from_shufti(cls, lo, hi):
        c = _Class()
        lo, hi = bytes(lo), bytes(hi)
        if _lib().hsgpu_class_from_shufti(lo, hi, C.byref(c)) != 0:
            raise HsgpuError(-1, "hsgpu_class_from_shufti")
        return cls._from_c(c)

    @classmethod
    def from_truffle(cls, m1, m2):
        c = _Class()
        if _lib().hsgpu_class_from_truffle(bytes(m1), bytes(m2), C.byref(c)) != 0:
            raise HsgpuError(-1, "hsgpu_class_from_truffle")
        return cls._from_c(c)

    @classmethod
    def from_verm(cls, ch, nocase=False, negate=False):
        c = _Class()
        ch = ch if isinstance(ch, int) else ord(ch)
        _lib().hsgpu_class_from_verm(ch, int(nocase), int(negate), C.byref(c))
        return cls._from_c(c)

    def to_shufti(self):
        """shuftiBuildMasks -> (lo, hi, n_buckets) or None when the class needs more than 8 buckets"""
        lo, hi = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
        c = self._to_c()
        nb = _lib().hsgpu_class_to_shufti(C.byref(c), lo, hi)
        return (bytes(lo), bytes(hi), nb) if nb > 0 else None

    def to_truffle(self):
        m1, m2 = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
        c = self._to_c()
        _lib().hsgpu_class_to_truffle(C.byref(c), m1, m2)
        return bytes(m1), bytes(m2)

    def members(self):
        return [v for v in range(256) if self.bitmap[v >> 3] >> (v & 7) & 1]


def class_scan(classes, d_corpus, total, d_off=None, nblocks=0, want_first=True, want_last=False, stream=None,
               buffers=None):
    """Evaluate <= 8 classes (<= 16 when neither first nor last is asked for: one read of the corpus) over a
    device-resident block batch (torch tensors).
    -> (bitmaps uint8 [n][ceil(total/16)*2], first int64-view uint32 [n][nblocks] | None, last | None)"""
    import torch

    lib = _lib()
    n = len(classes)
    assert 1 <= n <= (CLASS_MAX if ((want_first or want_last) and nblocks) else CLASS_MAX_BITMAPS)
    dev = d_corpus.device
    arr = (_Class * n)(*[c._to_c() for c in classes])
    words = (total + 15) // 16
    if buffers is not None:  # reuse (bitmaps, first, last, work) from an earlier call: no allocation, no sync
        bitmaps, first, last, work = buffers
    else:
        # rows padded to 8 bytes: 8-byte aligned bitmaps let the first/last kernel read 64-bit words
        bitmaps = torch.empty((n, (max(1, words) * 2 + 7) // 8 * 8), dtype=torch.uint8, device=dev)
        work = torch.zeros(WORK_BYTES, dtype=torch.uint8, device=dev)
        first = torch.zeros((n, nblocks), dtype=torch.int32, device=dev) if (want_first and nblocks) else None
        last = torch.zeros((n, nblocks), dtype=torch.int32, device=dev) if (want_last and nblocks) else None
    ptrs = (C.c_void_p * n)(*[bitmaps[i].data_ptr() for i in range(n)])
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    rv = lib.hsgpu_class_scan_dev(arr, n, d_corpus.data_ptr(), total, d_off.data_ptr() if d_off is not None else None,
                                  nblocks, ptrs, first.data_ptr() if first is not None else None,
                                  last.data_ptr() if last is not None else None, work.data_ptr(), st)
    if rv != 0:
        raise HsgpuError(rv, "hsgpu_class_scan_dev")
    if buffers is None:
        _wait_for(stream)  # `work` must outlive the launch (the stream it went to, not torch's current one)
        return bitmaps, first, last
    return bitmaps, first, last, work


class _Seq(C.Structure):
    _fields_ = [("a", C.c_uint8), ("b", C.c_uint8), ("m", C.c_uint8), ("n", C.c_uint8), ("id", C.c_uint32)]


SEQ_MAX, SEQ_MAX_REPEAT = 1024, 16


def class_seq_scan(seqs, bitmaps, total, d_off, nblocks, emit=(0, 0), cap=0, stream=None, buffers=None, emit_only=False):
    """A{m,}B{n,} class-sequence patterns over the membership bitmaps of class_scan (device-resident, torch).
    seqs: iterable of (a, b, m, n, id) with a, b indices into `bitmaps` (a list of 1-D uint8 device tensors,
    one per class). emit: corpus byte range [lo, hi) whose match ends are written as records (cap of them).
    -> (counts int64 [n_seqs] on the device, records uint32 [k][4] = (block, end, id, pattern index) on the host
    in no particular order, number of match ends in the emit range).
    buffers = (work, counts, out, count) from class_seq_buffers: nothing is allocated and nothing waits; the
    device tensors come back as they are (counts, out, count).
    Reference: what an accelerated NFA / DFA engine reports for such patterns (src/nfa/accel.c:35-146,
    src/nfa/limex_accel.c:49-74)."""
    import torch

    lib = _lib()
    seqs = list(seqs)
    arr = (_Seq * len(seqs))(*[_Seq(a, b, m, n, i) for a, b, m, n, i in seqs])
    ptrs = (C.c_void_p * len(bitmaps))(*[b.data_ptr() for b in bitmaps])
    work, counts, out, count = buffers if buffers is not None else class_seq_buffers(len(seqs), total, cap, d_off.device)
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    lib.hsgpu_class_seq_scan_dev.restype = C.c_int
    lib.hsgpu_class_seq_scan_dev.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]
    if emit_only:  # hsgpu_class_seq_emit_dev: the records of a range of whole blocks, nothing counted, only that range walked
        lib.hsgpu_class_seq_emit_dev.restype = C.c_int
        lib.hsgpu_class_seq_emit_dev.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_uint64,
                                                 C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.c_void_p]
        rv = lib.hsgpu_class_seq_emit_dev(arr, len(seqs), ptrs, len(bitmaps), total, d_off.data_ptr(), nblocks, int(emit[0]), int(emit[1]),
                                          out.data_ptr(), cap, count.data_ptr(), work.data_ptr(), work.numel() - 16, st)
    else:
        rv = lib.hsgpu_class_seq_scan_dev(arr, len(seqs), ptrs, len(bitmaps), total, d_off.data_ptr(), nblocks, int(emit[0]),
                                          int(emit[1]), counts.data_ptr(), out.data_ptr() if cap else None, cap, count.data_ptr(),
                                          work.data_ptr(), work.numel() - 16, st)
    if rv != 0:
        raise HsgpuError(rv, _native.load_library().hsgpu_last_error().decode())
    if buffers is not None:
        return counts, out, count
    if stream is not None:
        torch.cuda.synchronize()
    n_emit = int(count.item())  # synchronises the launch stream: `work` and the argument arrays may go
    recs = out[: min(n_emit, cap)].cpu().numpy().astype(np.uint32) if cap else np.zeros((0, 4), np.uint32)
    return counts, recs, n_emit


def class_seq_buffers(n_seqs, total, cap, device):
    """(work, counts, out, count) device tensors for class_seq_scan(buffers=...)"""
    import torch

    lib = _lib()
    lib.hsgpu_class_seq_work_bytes.restype = C.c_size_t
    lib.hsgpu_class_seq_work_bytes.argtypes = [C.c_uint64]
    wb = int(lib.hsgpu_class_seq_work_bytes(total))
    return (torch.empty(wb + 16, dtype=torch.uint8, device=device), torch.zeros(n_seqs, dtype=torch.int64, device=device),
            torch.zeros((max(1, cap), 4), dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int64, device=device))


class PairSet:
    """A set of two-byte sequences in double-shufti form (<= 8 buckets, 0-active masks)."""

    def __init__(self, c=None):
        self._c = c if c is not None else _Pair()

    @property
    def masks(self):
        return bytes(self._c.lo1), bytes(self._c.hi1), bytes(self._c.lo2), bytes(self._c.hi2)

    @classmethod
    def from_dshufti(cls, lo1, hi1, lo2, hi2):
        c = _Pair()
        if _lib().hsgpu_pair_from_dshufti(bytes(lo1), bytes(hi1), bytes(lo2), bytes(hi2), C.byref(c)) != 0:
            raise HsgpuError(-1, "hsgpu_pair_from_dshufti")
        return cls(c)

    @classmethod
    def from_dverm(cls, c1, c2, nocase=False):
        c = _Pair()
        o = lambda x: x if isinstance(x, int) else ord(x)
        _lib().hsgpu_pair_from_dverm(o(c1), o(c2), int(nocase), C.byref(c))
        return cls(c)

    @classmethod
    def from_dverm_masked(cls, c1, c2, m1, m2):
        c = _Pair()
        _lib().hsgpu_pair_from_dverm_masked(c1, c2, m1, m2, C.byref(c))
        return cls(c)

    @classmethod
    def build(cls, pairs=(), onechar=None):
        """shuftiBuildDoubleMasks: `pairs` = iterable of (first, second) byte values, `onechar` = CharClass of
        single bytes with a wildcard second byte. Raises HsgpuError(-4) when > 8 buckets are needed."""
        c = _Pair()
        flat = bytes(b for p in pairs for b in (p[0] if isinstance(p[0], int) else ord(p[0]),
                                                p[1] if isinstance(p[1], int) else ord(p[1])))
        oc = onechar._to_c() if onechar is not None else None
        rv = _lib().hsgpu_pair_build(C.byref(oc) if oc is not None else None, flat, len(flat) // 2, C.byref(c))
        if rv != 0:
            raise HsgpuError(rv, _native.load_library().hsgpu_last_error().decode())
        return cls(c)

    def test(self, a, b):
        return bool(_lib().hsgpu_pair_test(C.byref(self._c), a, b))


def pair_scan(pairs, d_corpus, total, d_off=None, nblocks=0, want_first=True, want_last=False, stream=None):
    """Evaluate <= 8 two-byte sets over a device-resident block batch (torch tensors).
    -> (bitmaps uint8 [n][ceil(total/16)*2], first int32-view uint32 [n][nblocks] | None, last | None)"""
    import torch

    lib = _lib()
    n = len(pairs)
    assert 1 <= n <= PAIR_MAX
    dev = d_corpus.device
    arr = (_Pair * n)(*[p._c for p in pairs])
    words = (total + 15) // 16
    bitmaps = torch.empty((n, max(1, words) * 2), dtype=torch.uint8, device=dev)
    work = torch.zeros(PAIR_WORK_BYTES, dtype=torch.uint8, device=dev)
    first = torch.zeros((n, nblocks), dtype=torch.int32, device=dev) if (want_first and nblocks) else None
    last = torch.zeros((n, nblocks), dtype=torch.int32, device=dev) if (want_last and nblocks) else None
    ptrs = (C.c_void_p * n)(*[bitmaps[i].data_ptr() for i in range(n)])
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    rv = lib.hsgpu_pair_scan_dev(arr, n, d_corpus.data_ptr(), total, d_off.data_ptr() if d_off is not None else None,
                                 nblocks, ptrs, first.data_ptr() if first is not None else None,
                                 last.data_ptr() if last is not None else None, work.data_ptr(), st)
    if rv != 0:
        raise HsgpuError(rv, "hsgpu_pair_scan_dev")
    _wait_for(stream)  # `work` must outlive the launch (the stream it went to, not torch's current one)
    return bitmaps, first, last


class ForwardAccel:
    """The pre-skip scheme of a literal set (buildForwardAccel, rose_build_lit_accel.cpp:459-465)."""

    def __init__(self, c):
        self._c = c
        self.type, self.offset, self.c1, self.c2 = c.type, c.offset, c.c1, c.c2
        self.mask_lo, self.mask_hi = bytes(c.mask_lo), bytes(c.mask_hi)

    @classmethod
    def choose(cls, lits, expected_groups=0xFFFFFFFFFFFFFFFF):
        from .hwlm import pack_literals

        arr, _keep = pack_literals(list(lits))
        out = _Accel()
        rv = _lib().hsgpu_accel_forward(arr, len(arr), expected_groups, C.byref(out))
        if rv != 0:
            raise HsgpuError(rv, _native.load_library().hsgpu_last_error().decode())
        return cls(out)

    def scanner(self):
        """-> ("class", CharClass) | ("pair", PairSet) | None: what to hand to class_scan / pair_scan"""
        if self.type in (ACCEL_VERM, ACCEL_VERM_NOCASE):
            return "class", CharClass.from_verm(self.c1, self.type == ACCEL_VERM_NOCASE)
        if self.type in (ACCEL_DVERM, ACCEL_DVERM_NOCASE):
            return "pair", PairSet.from_dverm(self.c1, self.c2, self.type == ACCEL_DVERM_NOCASE)
        if self.type == ACCEL_SHUFTI:
            return "class", CharClass.from_shufti(self.mask_lo, self.mask_hi)
        if self.type == ACCEL_TRUFFLE:
            return "class", CharClass.from_truffle(self.mask_lo, self.mask_hi)
        return None


def forward_skip(fa, d_corpus, total, d_off, nblocks, start=0, stream=None):
    """do_accel_block (src/hwlm/hwlm.c:80-99) for a whole batch, through hsgpu_hwlm_forward_skip_dev:
    per block with at least 16 bytes after its start, max(0, hit - offset) where hit is
    run_hwlm_accel's return value over [start, len) (the block length when nothing is found);
    other blocks, and sets without a scheme, keep their start. `start`: one int for every block or
    an int32/uint32 device tensor [nblocks]. -> int64 tensor [nblocks]."""
    import torch

    lib = _lib()
    dev = d_corpus.device
    out = torch.empty(nblocks, dtype=torch.int32, device=dev)
    bitmap = torch.empty(max(1, (total + 15) // 16) * 2, dtype=torch.uint8, device=dev)
    work = torch.zeros(PAIR_WORK_BYTES, dtype=torch.uint8, device=dev)
    per_block = None if isinstance(start, int) else start.to(device=dev, dtype=torch.int32).contiguous()
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    rv = lib.hsgpu_hwlm_forward_skip_dev(C.byref(fa._c), d_corpus.data_ptr(), total, d_off.data_ptr(), nblocks,
                                         per_block.data_ptr() if per_block is not None else None,
                                         0 if per_block is not None else int(start), out.data_ptr(),
                                         bitmap.data_ptr(), work.data_ptr(), st)
    if rv != 0:
        raise HsgpuError(rv, "hsgpu_hwlm_forward_skip_dev")
    _wait_for(stream)  # `work` / `bitmap` must outlive the launch, `out` must be complete
    return out.to(torch.int64) & 0xFFFFFFFF


def _wait_for(stream):
    """wait for the stream the kernels were launched on: torch's current stream, or -- a raw stream handle torch
    knows nothing about -- the whole device"""
    import torch

    if stream is None:
        torch.cuda.current_stream().synchronize()
    else:
        torch.cuda.synchronize()


class AccelAux(C.Structure):
    """union AccelAux of the reference (src/nfa/accel.h:66-113), byte for byte: 80 bytes"""
    _fields_ = [("accel_type", C.c_uint8), ("offset", C.c_uint8), ("c1", C.c_uint8), ("c2", C.c_uint8), ("m1", C.c_uint8),
                ("m2", C.c_uint8), ("pad", C.c_uint8 * 10), ("mask", (C.c_uint8 * 16) * 4)]

    @classmethod
    def make(cls, accel_type, offset=0, c1=0, c2=0, m1=0, m2=0, masks=()):
        a = cls()
        a.accel_type, a.offset, a.c1, a.c2, a.m1, a.m2 = accel_type, offset, c1, c2, m1, m2
        for k, m in enumerate(masks):
            C.memmove(a.mask[k], bytes(m), 16)
        return a


ACCEL_DSHUFTI, ACCEL_RED_TAPE, ACCEL_DVERM_MASKED = 14, 16, 17


def run_accel(aux, d_corpus, total, d_off, nblocks, start=0, stream=None):
    """run_accel (src/nfa/accel.c:35-146) for every block of a device-resident batch through hsgpu_run_accel_dev:
    -> int64 tensor [nblocks], run_accel(aux, buf + start, buf + len) - buf. `start`: an int or a device tensor."""
    import torch

    lib = _lib()
    dev = d_corpus.device
    out = torch.empty(nblocks, dtype=torch.int32, device=dev)
    bitmap = torch.empty(max(1, (total + 15) // 16) * 2, dtype=torch.uint8, device=dev)
    work = torch.zeros(PAIR_WORK_BYTES, dtype=torch.uint8, device=dev)
    per_block = None if isinstance(start, int) else start.to(device=dev, dtype=torch.int32).contiguous()
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    lib.hsgpu_run_accel_dev.restype = C.c_int
    lib.hsgpu_run_accel_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rv = lib.hsgpu_run_accel_dev(C.byref(aux), d_corpus.data_ptr(), total, d_off.data_ptr(), nblocks,
                                 per_block.data_ptr() if per_block is not None else None,
                                 0 if per_block is not None else int(start), out.data_ptr(), bitmap.data_ptr(), work.data_ptr(), st)
    if rv != 0:
        raise HsgpuError(rv, _native.load_library().hsgpu_last_error().decode())
    _wait_for(stream)
    return out.to(torch.int64) & 0xFFFFFFFF
