/*
 * exchange.hip -- the path's one exchange step behind the C ABI (include/hsgpu.h, hsgpu_exchange_*).
 *
 * Blocks are independent scans, so a multi-GPU job shards the corpus by contiguous block ranges (one process per GPU, the
 * table replicated) and needs exactly one collective per step: the match records of every shard to the rank whose host
 * delivers the callbacks -- hsbench's threads each report into one result table (tools/hsbench/main.cpp:957-963, 990-1030) --
 * or, on request, to every rank. Round 3 had this in the Python bench harness over torch.distributed; here it is RCCL
 * directly (ncclSend / ncclRecv in one group, or ncclAllGather), so that a C caller of INTEGRATION.md section 1 can shard.
 *
 *   wire record  12 bytes {global block, end, id}: the rank's first global block is added while the records are packed, the
 *                scan's internal literal index stays home (25 % less on the wire than the 16-byte hsgpu_match_t)
 *   step         pack (one small kernel: reads the scan's count on the device, no host synchronisation) + the collective,
 *                both on the caller's stream. A rank's slot is a fixed number of rows agreed at creation, or -- after
 *                hsgpu_exchange_set_counts -- exactly the rows the ranks agreed on (steps that repeat a scan)
 *   to root      xGMI is point to point: every peer has a link of its own to the root, the (world - 1) transfers run side
 *                by side, each carrying ONE rank's records; a ring all-gather carries (world - 1) ranks' worth over every link
 *   compact      the slots of the last step packed into one array in rank order = corpus order, counts to the host
 *
 * RCCL is loaded at run time (the copy already in the process when there is one -- PyTorch brings its own -- else the
 * system's): the library links without it and a single-GPU caller never touches it.
 *
 * Round 5: the collectives sit behind a small transport interface with two implementations -- RCCL, and an in-process
 * LOOPBACK (hsgpu_exchange_loopback_id: 2-8 virtual ranks of one process on one device, a Send meeting its Recv becomes one
 * hipMemcpyAsync ordered by events). The step's own logic -- which rank sends what to whom, slot offsets, bytes_of, agreed
 * counts, the headers compact reads -- is the same code over either, so a 1-GPU box runs all of it (tests/test_gpu_exchange.py).
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "internal.h"

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            hsgpu_set_error("%s failed: %s", #expr, hipGetErrorString(e_));        \
            return (e_ == hipErrorOutOfMemory) ? HSGPU_NOMEM : HSGPU_UNKNOWN_ERROR; \
        }                                                                          \
    } while (0)

namespace {

/* the few RCCL entry points the exchange needs (rccl.h: ncclResult_t = int, ncclSuccess = 0, ncclInt8 = 0) */
typedef struct ncclComm *ncclComm_t;
struct Id128 { /* ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128, passed by value */
    char b[128];
};
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *id) = nullptr;
    int (*CommInitRank)(ncclComm_t *comm, int nranks, Id128 id, int rank) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
typedef decltype(Rccl::CommInitRank) InitFn;

Rccl *rccl() {
    static std::mutex mu;
    static Rccl r;
    static bool tried = false;
    std::lock_guard<std::mutex> g(mu);
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) /* the copy the process already holds (PyTorch's), if any */
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (const char *n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) return nullptr;
    auto sym = [&](const char *s) { return dlsym(r.handle, s); };
    r.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
    r.CommInitRank = (InitFn)sym("ncclCommInitRank");
    r.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
    r.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t, hipStream_t))sym("ncclAllGather");
    r.Send = (int (*)(const void *, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
    r.Recv = (int (*)(void *, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
    r.GroupStart = (int (*)())sym("ncclGroupStart");
    r.GroupEnd = (int (*)())sym("ncclGroupEnd");
    r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}

#define NCCL_TRY(R, expr)                                                                                  \
    do {                                                                                                   \
        int e_ = (expr);                                                                                   \
        if (e_ != 0) {                                                                                     \
            hsgpu_set_error("%s failed: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(e_) : "?"); \
            return HSGPU_UNKNOWN_ERROR;                                                                    \
        }                                                                                                  \
    } while (0)


/* what the step needs from a fabric; every call posts work on `st` and returns (0 = ok, else hsgpu_set_error was called) */
struct Transport {
    virtual ~Transport() {}
    virtual int group_start() = 0;
    virtual int send(const void *p, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int recv(void *p, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int group_end() = 0;
    virtual int all_gather(const void *mine, void *all, size_t bytes, hipStream_t st) = 0; /* in place: mine == all + rank * bytes */
    virtual int before_use(hipStream_t) { return 0; } /* called before a rank touches its buffers again */
    virtual int unmet_receives() { return 0; }         /* loopback: receives of this rank whose sender has not come by yet */
    virtual const char *name() const = 0;
};

struct RcclTransport : Transport {
    Rccl *R;
    ncclComm_t comm = nullptr;
    explicit RcclTransport(Rccl *r) : R(r) {}
    ~RcclTransport() override {
        if (comm) (void)R->CommDestroy(comm);
    }
    int check(int e, const char *what) {
        if (e == 0) return 0;
        hsgpu_set_error("%s failed: %s", what, R->GetErrorString ? R->GetErrorString(e) : "?");
        return HSGPU_UNKNOWN_ERROR;
    }
    int group_start() override { return check(R->GroupStart(), "ncclGroupStart"); }
    int send(const void *p, size_t bytes, int peer, hipStream_t st) override { return check(R->Send(p, bytes, 0 /* ncclInt8 */, peer, comm, st), "ncclSend"); }
    int recv(void *p, size_t bytes, int peer, hipStream_t st) override { return check(R->Recv(p, bytes, 0, peer, comm, st), "ncclRecv"); }
    int group_end() override { return check(R->GroupEnd(), "ncclGroupEnd"); }
    int all_gather(const void *mine, void *all, size_t bytes, hipStream_t st) override {
        return check(R->AllGather(mine, all, bytes, 0, comm, st), "ncclAllGather");
    }
    const char *name() const override { return "rccl"; }
};

/* Loopback: the virtual ranks of one id share a hub. A Send and the Recv it meets are one device-to-device copy, posted by
 * whichever half arrives second on ITS stream after waiting for an event the first half recorded on its own; the first half's
 * owner waits for the copy (an event again) before it touches its buffers the next time (before_use). Nothing blocks the host,
 * so one thread may drive the ranks one after the other, in any order; sizes that do not agree are an error as they are a
 * hang over RCCL. */
struct LoopHub {
    std::mutex mu;
    int world = 0;
    struct Half {
        bool is_send;
        const void *ptr;
        size_t bytes;
        hipEvent_t ready;
        struct LoopTransport *owner;
    };
    std::map<std::pair<int, int>, std::deque<Half>> pending; /* (from, to) -> halves waiting for their other side, in order */
};
std::mutex g_hubs_mu;
std::map<uint64_t, std::weak_ptr<LoopHub>> g_hubs;
const char LOOP_MAGIC[16] = {'h', 's', 'g', 'p', 'u', '-', 'l', 'o', 'o', 'p', 'b', 'a', 'c', 'k', 0, 1};

struct LoopTransport : Transport {
    std::shared_ptr<LoopHub> hub;
    int rank = 0;
    std::vector<hipEvent_t> wait_for; /* copies other ranks posted out of / into this rank's buffers (guarded by hub->mu) */
    ~LoopTransport() override {
        std::lock_guard<std::mutex> g(hub->mu);
        for (auto &kv : hub->pending)
            for (auto it = kv.second.begin(); it != kv.second.end();)
                if (it->owner == this) {
                    (void)hipEventDestroy(it->ready);
                    it = kv.second.erase(it);
                } else
                    ++it;
        for (hipEvent_t e : wait_for) (void)hipEventDestroy(e);
    }
    int fail(hipError_t e, const char *what) {
        hsgpu_set_error("loopback %s failed: %s", what, hipGetErrorString(e));
        return HSGPU_UNKNOWN_ERROR;
    }
    int post(bool is_send, const void *p, size_t bytes, int peer, hipStream_t st) {
        if (peer < 0 || peer >= hub->world) return HSGPU_INVALID;
        const std::pair<int, int> key = is_send ? std::make_pair(rank, peer) : std::make_pair(peer, rank);
        std::lock_guard<std::mutex> g(hub->mu);
        auto &q = hub->pending[key];
        if (!q.empty() && q.front().is_send != is_send) { /* the other half is waiting: this call posts the copy */
            LoopHub::Half o = q.front();
            q.pop_front();
            hipError_t e;
            if (o.bytes != bytes) {
                (void)hipEventDestroy(o.ready);
                hsgpu_set_error("loopback: rank %d %s %zu bytes, rank %d %s %zu (the ranks disagree on a slot's size)", rank,
                                is_send ? "sends" : "expects", bytes, peer, is_send ? "expects" : "sends", o.bytes);
                return HSGPU_INVALID;
            }
            e = hipStreamWaitEvent(st, o.ready, 0);
            (void)hipEventDestroy(o.ready); /* (the half has been taken off the queue: released on every path, once the wait has been queued) */
            if (e != hipSuccess) return fail(e, "hipStreamWaitEvent");
            void *dst = is_send ? (void *)o.ptr : (void *)p;
            const void *src = is_send ? p : o.ptr;
            if ((e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)) != hipSuccess) return fail(e, "hipMemcpyAsync");
            hipEvent_t done;
            if ((e = hipEventCreateWithFlags(&done, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
            if ((e = hipEventRecord(done, st)) != hipSuccess) {
                (void)hipEventDestroy(done);
                return fail(e, "hipEventRecord");
            }
            o.owner->wait_for.push_back(done);
            return 0;
        }
        LoopHub::Half h{is_send, p, bytes, nullptr, this};
        hipError_t e;
        if ((e = hipEventCreateWithFlags(&h.ready, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventRecord(h.ready, st)) != hipSuccess) {
            (void)hipEventDestroy(h.ready);
            return fail(e, "hipEventRecord");
        }
        q.push_back(h);
        return 0;
    }
    int group_start() override { return 0; }
    int group_end() override { return 0; }
    int send(const void *p, size_t bytes, int peer, hipStream_t st) override { return post(true, p, bytes, peer, st); }
    int recv(void *p, size_t bytes, int peer, hipStream_t st) override { return post(false, p, bytes, peer, st); }
    int all_gather(const void *mine, void *all, size_t bytes, hipStream_t st) override {
        for (int r = 0; r < hub->world; r++) {
            if (r == rank) continue;
            int rv = send(mine, bytes, r, st);
            if (!rv) rv = recv((uint8_t *)all + (size_t)r * bytes, bytes, r, st);
            if (rv) return rv;
        }
        return 0;
    }
    int before_use(hipStream_t st) override {
        std::vector<hipEvent_t> evs;
        {
            std::lock_guard<std::mutex> g(hub->mu);
            evs.swap(wait_for);
        }
        int rv = 0;
        for (hipEvent_t e : evs) {
            const hipError_t err = hipStreamWaitEvent(st, e, 0);
            if (err != hipSuccess && !rv) rv = fail(err, "hipStreamWaitEvent"); /* (the buffers may be in use by a copy nobody waits for: the caller must not go on) */
            (void)hipEventDestroy(e);
        }
        return rv;
    }
    /* Over RCCL a rank that gathers before its peers have sent BLOCKS; here nothing blocks, and a rank that compacted before
     * every peer had called step() would read the previous step's slots. Its own receives still in the queue say so. */
    int unmet_receives() override {
        std::lock_guard<std::mutex> g(hub->mu);
        int n = 0;
        for (auto &kv : hub->pending)
            if (kv.first.second == rank)
                for (const auto &h : kv.second) n += (!h.is_send && h.owner == this) ? 1 : 0;
        return n;
    }
    const char *name() const override { return "loopback"; }
};

constexpr uint32_t SLOT_HEADER = 16; /* {uint64 count of the scan, uint64 rows in the slot} in front of a rank's wire records */

/* the scan's records -> this rank's slot: {count, rows} + rows wire records with global block indices */
/* (a scan that found more than its record buffer holds wrote nothing usable -- hsgpu_hwlm_scan_dev's `count > cap` protocol --: its
 * slot travels with 0 rows and the true count, and compact says HSGPU_INSUFFICIENT_SPACE; round 4 read past the buffer there) */
__global__ void exchange_pack_kernel(const hsgpu_match_t *recs, const unsigned long long *count, uint64_t record_cap, uint64_t first_block,
                                     uint64_t rows_max, uint8_t *slot) {
    const unsigned long long n = *count;
    const uint64_t rows = n > record_cap ? 0 : (n < rows_max ? n : rows_max);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ((unsigned long long *)slot)[0] = n;
        ((unsigned long long *)slot)[1] = rows;
    }
    hsgpu_wire_t *w = (hsgpu_wire_t *)(slot + SLOT_HEADER);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * blockDim.x) {
        const hsgpu_match_t r = recs[i];
        hsgpu_wire_t o;
        o.block = (uint32_t)(r.block + first_block); /* uint32 arithmetic: exact up to 2^32 blocks per job */
        o.end = r.end;
        o.id = r.id;
        w[i] = o;
    }
}

/* the slots of all ranks -> one array in rank order; d_counts[r] = the scan count of rank r, d_counts[world] = rows delivered,
 * d_counts[world + 1 + r] = the rows rank r's slot carried */
__global__ void exchange_compact_kernel(const uint8_t *slots, uint64_t slot_bytes, uint32_t world, hsgpu_wire_t *out, uint64_t cap,
                                        unsigned long long *d_counts) {
    uint64_t at = 0;
    for (uint32_t r = 0; r < world; r++) {
        const unsigned long long *h = (const unsigned long long *)(slots + (uint64_t)r * slot_bytes);
        const uint64_t n = h[0], rows = h[1];
        const hsgpu_wire_t *w = (const hsgpu_wire_t *)(slots + (uint64_t)r * slot_bytes + SLOT_HEADER);
        if (blockIdx.x == 0 && threadIdx.x == 0) d_counts[r] = n, d_counts[world + 1 + r] = rows;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * blockDim.x)
            if (at + i < cap) out[at + i] = w[i];
        at += rows;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) d_counts[world] = at;
}

} // namespace

struct hsgpu_exchange {
    int world = 1, rank = 0, root = 0, device = 0;
    unsigned mode = HSGPU_XCHG_TO_ROOT;
    uint64_t rows = 0;       /* rows per slot */
    uint64_t slot_bytes = 0; /* header + rows wire records, a multiple of 16 */
    Transport *tr = nullptr;  /* RCCL or loopback; NULL: world == 1 without an id (nothing to exchange) */
    uint8_t *send = nullptr;  /* this rank's slot */
    uint8_t *slots = nullptr; /* [world] slots: what the last step received (root, or every rank) */
    unsigned long long *d_counts = nullptr, *h_counts = nullptr; /* [2 * world + 1] */
    std::vector<uint64_t> agreed; /* hsgpu_exchange_set_counts: the rows every rank sends, else empty (fixed slots) */
    bool receives() const { return mode == HSGPU_XCHG_ALL_GATHER || rank == root; }
};

extern "C" int hsgpu_exchange_unique_id(void *id) {
    if (!id) return HSGPU_INVALID;
    Rccl *R = rccl();
    if (!R) {
        hsgpu_set_error("RCCL (librccl.so.1) could not be loaded");
        return HSGPU_UNKNOWN_ERROR;
    }
    NCCL_TRY(R, R->GetUniqueId(id));
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_exchange_loopback_id(void *id) {
    static std::mutex mu;
    static uint64_t next = 1;
    if (!id) return HSGPU_INVALID;
    std::lock_guard<std::mutex> g(mu);
    memset(id, 0, HSGPU_XCHG_ID_BYTES);
    memcpy(id, LOOP_MAGIC, sizeof(LOOP_MAGIC));
    const uint64_t serial = next++;
    memcpy((char *)id + sizeof(LOOP_MAGIC), &serial, 8);
    return HSGPU_SUCCESS;
}

extern "C" void hsgpu_exchange_free(hsgpu_exchange_t *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    delete x->tr;
    hsgpu_dev_free(x->send);
    hsgpu_dev_free(x->slots);
    hsgpu_dev_free(x->d_counts);
    if (x->h_counts) (void)hipHostFree(x->h_counts);
    delete x;
}

extern "C" int hsgpu_exchange_create(hsgpu_exchange_t **out, const void *id, int world, int rank, int device, uint64_t rows_per_rank,
                                     unsigned mode, int root) {
    if (!out || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || (world > 1 && !id) ||
        (mode != HSGPU_XCHG_TO_ROOT && mode != HSGPU_XCHG_ALL_GATHER))
        return HSGPU_INVALID;
    *out = nullptr;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    HIP_TRY(hipSetDevice(device));
    hsgpu_exchange *x = new (std::nothrow) hsgpu_exchange;
    if (!x) return HSGPU_NOMEM;
    x->world = world, x->rank = rank, x->root = root, x->device = device, x->mode = mode;
    x->rows = std::max<uint64_t>(1, rows_per_rank);
    x->slot_bytes = (SLOT_HEADER + x->rows * sizeof(hsgpu_wire_t) + 15) & ~(uint64_t)15;
    auto fail = [&](int rv) {
        hsgpu_exchange_free(x);
        return rv;
    };
    if (hsgpu_dev_alloc((void **)&x->send, x->slot_bytes) != HSGPU_SUCCESS || hipMemset(x->send, 0, SLOT_HEADER) != hipSuccess) return fail(HSGPU_NOMEM);
    if (x->receives() && (hsgpu_dev_alloc((void **)&x->slots, x->slot_bytes * (uint64_t)world) != HSGPU_SUCCESS ||
                          hipMemset(x->slots, 0, x->slot_bytes * (uint64_t)world) != hipSuccess))
        return fail(HSGPU_NOMEM);
    if (hsgpu_dev_alloc((void **)&x->d_counts, (2 * world + 1) * sizeof(unsigned long long)) != HSGPU_SUCCESS ||
        hipHostMalloc((void **)&x->h_counts, (2 * world + 1) * sizeof(unsigned long long)) != hipSuccess)
        return fail(HSGPU_NOMEM);
    if (id && !memcmp(id, LOOP_MAGIC, sizeof(LOOP_MAGIC))) { /* virtual ranks of this process (hsgpu_exchange_loopback_id) */
        uint64_t serial;
        memcpy(&serial, (const char *)id + sizeof(LOOP_MAGIC), 8);
        std::lock_guard<std::mutex> g(g_hubs_mu);
        std::shared_ptr<LoopHub> hub = g_hubs[serial].lock();
        if (!hub) {
            hub = std::make_shared<LoopHub>();
            hub->world = world;
            g_hubs[serial] = hub;
        }
        if (hub->world != world) {
            hsgpu_set_error("loopback id %llu was created for %d ranks, not %d", (unsigned long long)serial, hub->world, world);
            return fail(HSGPU_INVALID);
        }
        LoopTransport *t = new (std::nothrow) LoopTransport;
        if (!t) return fail(HSGPU_NOMEM);
        t->hub = hub, t->rank = rank;
        x->tr = t;
    } else if (id) { /* a communicator also at world size 1 when the caller brought an id: the one-GPU test of this path */
        Rccl *R = rccl();
        if (!R) {
            hsgpu_set_error("RCCL (librccl.so.1) could not be loaded");
            return fail(HSGPU_UNKNOWN_ERROR);
        }
        Id128 uid;
        memcpy(uid.b, id, sizeof(uid.b));
        RcclTransport *t = new (std::nothrow) RcclTransport(R);
        if (!t) return fail(HSGPU_NOMEM);
        x->tr = t;
        const int e = R->CommInitRank(&t->comm, world, uid, rank);
        if (e != 0) {
            hsgpu_set_error("ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(e) : "?");
            t->comm = nullptr;
            return fail(HSGPU_UNKNOWN_ERROR);
        }
    }
    *out = x;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_exchange_set_counts(hsgpu_exchange_t *x, const uint64_t *rows, int n) {
    if (!x || (rows && n != x->world)) return HSGPU_INVALID;
    x->agreed.clear();
    if (!rows) return HSGPU_SUCCESS;
    for (int r = 0; r < n; r++) {
        if (rows[r] > x->rows) {
            hsgpu_set_error("rank %d sends %llu rows, a slot holds %llu", r, (unsigned long long)rows[r], (unsigned long long)x->rows);
            return HSGPU_INVALID;
        }
        x->agreed.push_back(rows[r]);
    }
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_exchange_wire_bytes(const hsgpu_exchange_t *x, uint64_t *sent, uint64_t *received) {
    if (!x) return HSGPU_INVALID;
    auto bytes_of = [&](int r) { return x->agreed.empty() ? x->slot_bytes : (SLOT_HEADER + x->agreed[r] * sizeof(hsgpu_wire_t) + 15) & ~(uint64_t)15; };
    uint64_t s = 0, rcv = 0;
    for (int r = 0; r < x->world; r++) {
        if (r == x->rank) continue;
        if (x->mode == HSGPU_XCHG_ALL_GATHER) s += bytes_of(x->rank), rcv += bytes_of(r);
        else if (x->rank == x->root) rcv += bytes_of(r);
        else if (r == x->root) s += bytes_of(x->rank);
    }
    if (sent) *sent = s;
    if (received) *received = rcv;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_exchange_step(hsgpu_exchange_t *x, const void *d_records, uint64_t record_cap, const void *d_count, uint64_t first_block,
                                   void *stream) {
    if (!x || !d_records || !d_count) return HSGPU_INVALID;
    HIP_TRY(hipSetDevice(x->device));
    hipStream_t st = (hipStream_t)stream;
    int rv;
    if (x->tr && (rv = x->tr->before_use(st)) != HSGPU_SUCCESS) return rv; /* loopback: copies other ranks posted on their streams out of / into this rank's buffers */
    /* packed straight into its place where this rank also receives */
    uint8_t *mine = x->receives() ? x->slots + (uint64_t)x->rank * x->slot_bytes : x->send;
    const uint64_t my_rows = x->agreed.empty() ? x->rows : x->agreed[x->rank];
    hipLaunchKernelGGL(exchange_pack_kernel, dim3((unsigned)std::min<uint64_t>(1024, (my_rows + 255) / 256 + 1)), dim3(256), 0, st,
                       (const hsgpu_match_t *)d_records, (const unsigned long long *)d_count, record_cap, first_block, my_rows, mine);
    HIP_TRY(hipGetLastError());
    Transport *T = x->tr;
    if (!T) return HSGPU_SUCCESS; /* world == 1 and no id: there is nobody to exchange with */
    auto bytes_of = [&](int r) { return x->agreed.empty() ? x->slot_bytes : (SLOT_HEADER + x->agreed[r] * sizeof(hsgpu_wire_t) + 15) & ~(uint64_t)15; };
    if ((x->mode == HSGPU_XCHG_ALL_GATHER && x->agreed.empty()) || x->world == 1) {
        /* in place: every rank's slot already sits at its place in `slots` (at world size 1 with a communicator the collective
         * still runs -- one rank gathering from itself --, so that a 1-GPU box exercises the RCCL call: advisor, round 4) */
        return T->all_gather(mine, x->slots, x->slot_bytes, st);
    }
    /* point to point, all transfers of the step in one group: to the root, or (exact sizes) everybody to everybody. An error
     * inside the group still closes it: an open RCCL group would swallow the next call's transfers. */
    rv = T->group_start();
    if (rv) return rv;
    for (int r = 0; r < x->world && !rv; r++) {
        if (r == x->rank) continue;
        const bool i_send = x->mode == HSGPU_XCHG_ALL_GATHER || r == x->root;
        const bool i_recv = x->mode == HSGPU_XCHG_ALL_GATHER || x->rank == x->root;
        if (i_send) rv = T->send(mine, bytes_of(x->rank), r, st);
        if (i_recv && !rv) rv = T->recv(x->slots + (uint64_t)r * x->slot_bytes, bytes_of(r), r, st);
    }
    const int rv_end = T->group_end();
    return rv ? rv : rv_end;
}

extern "C" int hsgpu_exchange_compact(hsgpu_exchange_t *x, void *d_out, uint64_t cap, uint64_t *counts, uint64_t *total, void *stream) {
    if (!x || (cap && !d_out)) return HSGPU_INVALID;
    if (total) *total = 0;
    if (!x->receives()) return HSGPU_SUCCESS; /* nothing arrives here: the root has it */
    HIP_TRY(hipSetDevice(x->device));
    hipStream_t st = (hipStream_t)stream;
    if (x->tr) {
        const int unmet = x->tr->unmet_receives();
        if (unmet) {
            hsgpu_set_error("hsgpu_exchange_compact on rank %d before %d of its peers have called hsgpu_exchange_step: their slots hold the step before (over RCCL this call would block)",
                            x->rank, unmet);
            return HSGPU_INVALID;
        }
        const int rvb = x->tr->before_use(st);
        if (rvb != HSGPU_SUCCESS) return rvb;
    }
    hipLaunchKernelGGL(exchange_compact_kernel, dim3(256), dim3(256), 0, st, x->slots, x->slot_bytes, (uint32_t)x->world, (hsgpu_wire_t *)d_out, cap,
                       x->d_counts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(x->h_counts, x->d_counts, (2 * x->world + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    bool short_slot = false;
    for (int r = 0; r < x->world; r++) {
        if (counts) counts[r] = x->h_counts[r];
        const uint64_t room = x->agreed.empty() ? x->rows : x->agreed[r];
        /* a scan found more than its slot holds (or than was agreed), or more than its own record buffer held (0 rows travelled) */
        short_slot = short_slot || x->h_counts[r] > room || x->h_counts[x->world + 1 + r] != x->h_counts[r];
    }
    if (total) *total = x->h_counts[x->world];
    if (short_slot || x->h_counts[x->world] > cap) return HSGPU_INSUFFICIENT_SPACE;
    return HSGPU_SUCCESS;
}
