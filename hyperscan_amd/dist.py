"""Multi-GPU sharding of a block corpus and the one exchange step of the path.

Blocks are independent scans (the reference shares nothing between hs_scan calls:
the database is immutable, all mutable state lives in the caller's scratch), so
the corpus shards by contiguous block ranges balanced by bytes, every rank scans
its shard with no data-path collective, and afterwards the match records are
all-gathered (RCCL over xGMI with backend "nccl"; gloo in the CPU tests):
counts first, then records padded to the largest count.
"""
import numpy as np


def shard_blocks_by_bytes(off, world):
    """off: nblocks+1 ascending offsets. -> [(b_lo, b_hi)] per rank, contiguous,
    covering all blocks, balanced by bytes (not by block count)."""
    off = np.asarray(off, dtype=np.uint64)
    nblocks = off.size - 1
    total = int(off[-1] - off[0])
    cuts = [0]
    for r in range(1, world):
        target = int(off[0]) + total * r // world
        b = int(np.searchsorted(off, target, side="left"))
        b = min(max(b, cuts[-1]), nblocks)
        cuts.append(b)
    cuts.append(nblocks)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def local_shard(corpus, off, rank, world):
    """-> (corpus slice, rebased offsets, first global block index) of this rank."""
    lo, hi = shard_blocks_by_bytes(off, world)[rank]
    o = np.asarray(off[lo:hi + 1], dtype=np.uint64)
    base = int(o[0]) if o.size else 0
    return corpus[base:int(o[-1]) if o.size else base], (o - np.uint64(base)).astype(np.uint64), lo


def all_gather_records(records, count, block_base, dist, world, device=None):
    """records: int32 tensor [cap, 4] (block, end, id, lit) with `count` valid rows and
    rank-local block indices; block_base: this rank's first global block index.
    Returns (int32 tensor [total, 4] with GLOBAL block indices, ordered by rank, counts list).
    Two collectives: all_gather(counts), all_gather(records padded to max count)."""
    import torch

    device = device if device is not None else records.device
    cnt = torch.tensor([int(count), int(block_base)], dtype=torch.int64, device=device)
    allc = torch.empty(world * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, cnt)
    allc = allc.view(world, 2).cpu()
    counts = allc[:, 0].tolist()
    bases = allc[:, 1].tolist()
    mx = max(counts)
    if mx == 0:
        return torch.zeros((0, 4), dtype=torch.int32, device=device), counts
    pad = torch.zeros((mx, 4), dtype=torch.int32, device=device)
    n = int(count)
    pad[:n] = records[:n]
    buf = torch.empty((world * mx, 4), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(buf, pad)
    parts = []
    for r in range(world):
        p = buf[r * mx:r * mx + counts[r]].clone()
        p[:, 0] += _as_i32(bases[r])  # uint32 arithmetic on the int32 view: exact up to 2^32 blocks per job
        parts.append(p)
    return torch.cat(parts, dim=0), counts


def _as_i32(v):
    """the int32 bit pattern of a uint32 value (records travel as int32 tensors; hsgpu_match_t.block is uint32)"""
    v = int(v)
    if not 0 <= v < 1 << 32:
        raise ValueError("global block index %d does not fit the record's 32-bit block field" % v)
    return v - (1 << 32) if v >= 1 << 31 else v


class RecordExchange:
    """The per-step exchange of bench.py --gpus N with nothing allocated and no host synchronisation
    inside the step: one all-gather of {count, first global block} per rank and one all-gather of the
    record buffers padded to `rows` rows (a fixed size agreed before the timed steps, so no rank has to
    learn another rank's count before it can post its receive). Counts stay on the device; `compact()`
    (which synchronises) turns the last step's buffers into the rows of all ranks with GLOBAL block
    indices, ordered by rank -- the order of the corpus, since shards are contiguous block ranges.

    Block indices are 32 bits in a record (hsgpu_match_t.block): the add of the rank's first block wraps
    like uint32 arithmetic, so global indices are exact up to 2^32 blocks per job."""

    def __init__(self, dist, world, rank, device, rows, block_base):
        import torch

        self.dist, self.world, self.rank, self.rows = dist, world, rank, int(rows)
        self.cnt_in = torch.tensor([0, int(block_base)], dtype=torch.int64, device=device)
        self.cnt_out = torch.zeros((world, 2), dtype=torch.int64, device=device)
        self.rec_out = torch.zeros((world, self.rows, 4), dtype=torch.int32, device=device)

    def step(self, records, d_count):
        """records: int32 [>= rows, 4] device tensor of this rank's scan; d_count: int64 [1] device tensor
        the scan wrote its match count to. Enqueues both collectives on the current stream."""
        self.cnt_in[0:1].copy_(d_count[0:1], non_blocking=True)
        self.dist.all_gather_into_tensor(self.cnt_out.view(-1), self.cnt_in)
        self.dist.all_gather_into_tensor(self.rec_out.view(-1, 4), records[: self.rows])

    def compact(self):
        import torch

        cnt = self.cnt_out.cpu()
        counts, bases = cnt[:, 0].tolist(), cnt[:, 1].tolist()
        assert max(counts) <= self.rows, "exchange buffer smaller than a rank's match count"
        parts = []
        for r in range(self.world):
            p = self.rec_out[r, : counts[r]].clone()
            p[:, 0] += _as_i32(bases[r])  # uint32 add in an int32 tensor
            parts.append(p)
        return torch.cat(parts, dim=0), counts


class ExactExchange:
    """The per-step exchange without padding: every rank's count and first global block are agreed ONCE (the steps of
    a bench scan the same shard every time, so the counts do not change; `counts` / `bases` come from the warm-up),
    and a step is `world` broadcasts of exactly counts[r] rows each into their final place in one output tensor --
    sum(counts) rows per rank on the wire instead of world x max(counts). The form to take when shards are skewed
    (one flood-dense shard among quiet ones); like RecordExchange nothing is allocated and nothing waits for the
    host inside a step. compact() returns the rows of all ranks, global block indices, rank order."""

    def __init__(self, dist, world, rank, device, counts, bases):
        import torch

        self.dist, self.world, self.rank = dist, world, rank
        self.counts = [int(c) for c in counts]
        self.base = _as_i32(bases[rank])
        self.starts = np.concatenate([[0], np.cumsum(self.counts)]).tolist()
        self.out = torch.zeros((max(1, self.starts[-1]), 4), dtype=torch.int32, device=device)
        self.rows = max(self.counts) if self.counts else 0

    def step(self, records, d_count):
        n = self.counts[self.rank]
        if n:
            mine = self.out[self.starts[self.rank]:self.starts[self.rank] + n]
            mine.copy_(records[:n], non_blocking=True)
            mine[:, 0] += self.base  # uint32 add in an int32 tensor
        pending = [self.dist.broadcast(self.out[self.starts[r]:self.starts[r + 1]], src=r, async_op=True)
                   for r in range(self.world) if self.counts[r]]
        for h in pending:
            h.wait()

    def compact(self):
        return self.out[: self.starts[-1]], list(self.counts)


def _exchange_counts(count, block_base, dist, world, device):
    import torch

    cnt = torch.tensor([int(count), int(block_base)], dtype=torch.int64, device=device)
    allc = torch.empty(world * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, cnt)
    allc = allc.view(world, 2).cpu()
    return allc[:, 0].tolist(), allc[:, 1].tolist()


def all_gather_records_exact(records, count, block_base, dist, world, rank, device=None):
    """Same result as all_gather_records, but no rank is padded to the largest count: after
    the counts, rank r's rows travel as one broadcast of exactly counts[r] rows into their
    final place in the output (all `world` broadcasts in flight at once). With skewed shards
    (one flood-dense shard among quiet ones) this moves sum(counts) rows per rank instead of
    world * max(counts)."""
    import torch

    device = device if device is not None else records.device
    counts, bases = _exchange_counts(count, block_base, dist, world, device)
    total = sum(counts)
    out = torch.empty((total, 4), dtype=torch.int32, device=device)
    starts = np.concatenate([[0], np.cumsum(counts)]).tolist()
    n = int(count)
    if n:
        mine = out[starts[rank]:starts[rank] + n]
        mine.copy_(records[:n])
        mine[:, 0] += _as_i32(block_base)
    pending = [dist.broadcast(out[starts[r]:starts[r + 1]], src=r, async_op=True) for r in range(world) if counts[r]]
    for h in pending:
        h.wait()
    return out, counts


def gather_records_to_root(records, count, block_base, dist, world, rank, root=0, device=None):
    """When only one rank's host delivers the callbacks: counts to everyone, then every other
    rank sends exactly its rows to `root` (1/world of the all-gather's traffic per link).
    Returns (records with GLOBAL block indices ordered by rank, counts) on root and
    (None, counts) elsewhere."""
    import torch

    device = device if device is not None else records.device
    counts, bases = _exchange_counts(count, block_base, dist, world, device)
    n = int(count)
    if rank != root:
        if n:
            dist.send(records[:n].contiguous(), dst=root)
        return None, counts
    out = torch.empty((sum(counts), 4), dtype=torch.int32, device=device)
    starts = np.concatenate([[0], np.cumsum(counts)]).tolist()
    pending = [dist.irecv(out[starts[r]:starts[r + 1]], src=r) for r in range(world) if r != root and counts[r]]
    if n:
        out[starts[root]:starts[root] + n].copy_(records[:n])
    for h in pending:
        h.wait()
    for r in range(world):
        out[starts[r]:starts[r + 1], 0] += _as_i32(bases[r])
    return out, counts


class NativeExchange:
    """Mirror of the C ABI's exchange (include/hsgpu.h, hsgpu_exchange_*; csrc/exchange.hip): RCCL directly, 12-byte wire
    records {global block, end, id}, to the root (default: one host delivers the callbacks) or to every rank.

    The 128-byte RCCL id travels from rank 0 to the others over the process group the job already has (`dist`, any backend);
    world == 1 needs no group. step() takes the scan's device buffers as they are (nothing allocated, no host sync);
    compact() synchronises and returns (int32 tensor [total, 3] of the ranks' records in corpus order -- empty where nothing
    arrives --, counts per rank)."""

    TO_ROOT, ALL_GATHER = 0, 1

    def __init__(self, dist, world, rank, device, rows, block_base, mode=TO_ROOT, root=0, with_comm=None, id_bytes=None):
        import ctypes as C

        import torch

        from . import _native

        self._lib = lib = _native.load_library()
        self.world, self.rank, self.rows, self.base, self.mode, self.root = world, rank, int(rows), int(block_base), mode, root
        self.device = device
        lib.hsgpu_exchange_create.restype = C.c_int
        lib.hsgpu_exchange_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint, C.c_int]
        lib.hsgpu_exchange_step.restype = C.c_int
        lib.hsgpu_exchange_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.hsgpu_exchange_compact.restype = C.c_int
        lib.hsgpu_exchange_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
        lib.hsgpu_exchange_set_counts.restype = C.c_int
        lib.hsgpu_exchange_set_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.hsgpu_exchange_wire_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.hsgpu_exchange_free.argtypes = [C.c_void_p]
        want_comm = world > 1 if with_comm is None else with_comm
        idbuf = None
        if id_bytes is not None:  # the caller brought the id (loopback_id(): virtual ranks of one process)
            idbuf = (C.c_uint8 * 128)(*bytes(id_bytes))
        elif want_comm:
            idt = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_uint8 * 128)()
                rv = lib.hsgpu_exchange_unique_id(raw)
                if rv != 0:
                    raise RuntimeError("hsgpu_exchange_unique_id: " + lib.hsgpu_last_error().decode())
                idt = torch.tensor(list(raw), dtype=torch.uint8)
            if world > 1:
                on = idt.to(device) if dist.get_backend() == "nccl" else idt
                dist.broadcast(on, src=0)
                idt = on.cpu()
            idbuf = (C.c_uint8 * 128)(*idt.tolist())
        h = C.c_void_p()
        rv = lib.hsgpu_exchange_create(C.byref(h), idbuf, world, rank, device.index if device.index is not None else 0, self.rows, mode, root)
        if rv != 0:
            raise RuntimeError("hsgpu_exchange_create: " + lib.hsgpu_last_error().decode())
        self._h = h
        self.out = torch.zeros((max(1, world * self.rows), 3), dtype=torch.int32, device=device)

    def set_counts(self, counts):
        import ctypes as C

        arr = (C.c_uint64 * self.world)(*[int(c) for c in counts]) if counts is not None else None
        rv = self._lib.hsgpu_exchange_set_counts(self._h, arr, self.world)
        if rv != 0:
            raise RuntimeError("hsgpu_exchange_set_counts: " + self._lib.hsgpu_last_error().decode())

    def wire_bytes(self):
        import ctypes as C

        s, r = C.c_uint64(), C.c_uint64()
        self._lib.hsgpu_exchange_wire_bytes(self._h, C.byref(s), C.byref(r))
        return s.value, r.value

    @staticmethod
    def loopback_id():
        """an id for VIRTUAL ranks: `world` NativeExchange objects of this process made with it (id_bytes=) exchange by device
        copies instead of RCCL -- the N > 1 step's own logic on a 1-GPU box (include/hsgpu.h, hsgpu_exchange_loopback_id)"""
        import ctypes as C

        from . import _native

        raw = (C.c_uint8 * 128)()
        lib = _native.load_library()
        lib.hsgpu_exchange_loopback_id.restype = C.c_int
        assert lib.hsgpu_exchange_loopback_id(raw) == 0
        return bytes(raw)

    def step(self, records, d_count, cap=None):
        """records: the scan's int32 [cap, 4] device tensor, d_count its int64 [1] counter; on torch's current stream"""
        import torch

        rv = self._lib.hsgpu_exchange_step(self._h, records.data_ptr(), records.shape[0] if cap is None else cap, d_count.data_ptr(), self.base,
                                           torch.cuda.current_stream().cuda_stream)
        if rv != 0:
            raise RuntimeError("hsgpu_exchange_step: " + self._lib.hsgpu_last_error().decode())

    def compact(self):
        import ctypes as C

        import torch

        counts = (C.c_uint64 * self.world)()
        total = C.c_uint64()
        rv = self._lib.hsgpu_exchange_compact(self._h, self.out.data_ptr(), self.out.shape[0], counts, C.byref(total),
                                              torch.cuda.current_stream().cuda_stream)
        if rv != 0:
            raise RuntimeError(f"hsgpu_exchange_compact: rc {rv} (counts {list(counts)}, slot {self.rows} rows) " + self._lib.hsgpu_last_error().decode())
        return self.out[: total.value], [int(c) for c in counts]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hsgpu_exchange_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
