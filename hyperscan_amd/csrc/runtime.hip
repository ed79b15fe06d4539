/*
 * runtime.hip -- run side of the C ABI (include/hsgpu.h).
 *
 * hsgpu_hwlm_exec mirrors the reference's hwlmExec (src/hwlm/hwlm.c:172-199):
 * same arguments, same return values, callbacks delivered on the calling thread
 * in non-decreasing `end`. What the reference does inline inside its confirm
 * loop (group gate fdr_confirm_runtime.h:91, NOREPEAT :73-75, termination
 * fdr.c:719-721) cannot run on the GPU because each callback's return value
 * feeds the next decision; the GPU therefore emits the group-independent
 * superset of matches and hsgpu_hwlm_replay applies those three rules on the
 * host while walking the sorted records.
 *
 * hsgpu_hwlm_scan_dev is the hot path proper: everything resident in HBM, one
 * kernel launch per <= 2^36-byte corpus, fully asynchronous.
 */
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <functional>
#include <vector>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>

#include "internal.h"
#include "scan_kernels.h"
#include "../../include/hsgpu_tuning.h"

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            hsgpu_set_error("%s failed: %s", #expr, hipGetErrorString(e_));        \
            return (e_ == hipErrorOutOfMemory) ? HSGPU_NOMEM : HSGPU_UNKNOWN_ERROR; \
        }                                                                          \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return HSGPU_SUCCESS;
        hsgpu_dev_free(p);
        p = nullptr;
        cap = 0;
        /* guard mode (devmem.hip): exactly what was asked for, so that the buffer ends where the mapping ends */
        size_t want = hsgpu_dev_guard_mode() ? std::max<size_t>(bytes, 16) : std::max<size_t>(bytes + bytes / 4, 4096);
        int rv = hsgpu_dev_alloc(&p, want);
        if (rv != HSGPU_SUCCESS) return rv;
        cap = want;
        return HSGPU_SUCCESS;
    }
    void release() {
        hsgpu_dev_free(p);
        p = nullptr;
        cap = 0;
    }
};

/* A few resident host threads for the consumer side of the boundary (callbacks of disjoint block ranges): threads
 * created once per scratch, jobs handed over through one mutex; the calling thread works too. */
class WorkerPool {
  public:
    explicit WorkerPool(unsigned n) {
        for (unsigned i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    unsigned size() const { return (unsigned)th_.size(); }
    /* f(j) for j in [0, jobs): on the pool and on the caller; returns when all are done */
    void run(unsigned jobs, const std::function<void(unsigned)> &f) {
        if (jobs == 0) return;
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &f;
            next_ = 0;
            jobs_ = jobs;
            left_ = jobs;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return left_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            unsigned j;
            const std::function<void(unsigned)> *f;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!fn_ || next_ >= jobs_) return;
                j = next_++;
                f = fn_;
            }
            (*f)(j);
            std::lock_guard<std::mutex> g(mu_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return stop_ || (fn_ && next_ < jobs_); });
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned next_ = 0, jobs_ = 0, left_ = 0;
    bool stop_ = false;
};

struct hsgpu_scratch {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;           /* block hints are computed beside the filter kernel */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool timing = false;                   /* hsgpu_scratch_enable_timing */
    void *user_ctx = nullptr;              /* hsgpu_scratch_set_context: the callbacks' third argument */
    bool has_user_ctx = false;
    /* ring of event sets {start, filter done, confirm done, packed}: one per scan */
    static const int kRing = 32;
    hipEvent_t ev_ring[kRing][4] = {};
    hipEvent_t *ev_t = nullptr; /* the set of the scan being launched */
    uint64_t n_timed = 0;       /* scans launched with timing on */
    DevBuf tstamp;              /* [kRing][4] device wall clock: filter start (min) / end (max), confirm-stage end, pipeline end */
    bool timing_wg = false;     /* enable == 2: per-workgroup stamps of the filter kernel too */
    DevBuf wg_stamps;
    unsigned wg_stamps_n = 0;   /* workgroups of the last stamped scan */
    DevBuf conf_stamps;
    unsigned conf_stamps_n = 0; /* confirm workers of the last stamped scan */
    double wall_clock_khz = 100000.0;
    DevBuf corpus, off, out, count, hint, cand, ctl, rec_stage, stats, run_tab;
    DevBuf pipe_corpus[2], pipe_off[2], pipe_out[2], pipe_count; /* hsgpu_hwlm_exec_batch_cb: two chunks in flight */
    hipEvent_t ev_copied[2] = {}, ev_scanned[2] = {};
    /* ... its page-locked staging: the chunk's relative offsets on their way in, its records on their way out. (Round 4 copied
     * both through pageable vectors: a pageable copy is staged by the runtime and synchronises, and the offsets' copy sat in the
     * copy stream BETWEEN two chunks' corpus copies -- 0.1-0.2 ms of idle bus per chunk, the "0.2-0.6 ms per chunk that no copy
     * hides" of DESIGN 6.) */
    uint64_t *h_rel[2] = {nullptr, nullptr};
    size_t h_rel_cap[2] = {0, 0};
    hsgpu_match_t *h_prec[2] = {nullptr, nullptr};
    size_t h_prec_cap[2] = {0, 0};
    DevBuf cs_bitmaps, cs_work, cs_counts; /* hsgpu_class_seq_exec_batch: class bitmaps, work areas, per-pattern counts */
    bool ctl_clean = false;                /* the control block the next scan will use is zero (left so by the scan before last) */
    unsigned ctl_parity = 0;               /* which half of the control buffer the next scan uses */
    unsigned long long stats_seen[2] = {0, 0};
    unsigned long long *h_count = nullptr; /* pinned */
    uint32_t *h_note = nullptr, *d_note = nullptr; /* mapped pinned word the fused fallback sets (see HsgpuScanArgs::overflow_note) */
    bool tune_no_skew = false;             /* ... 5: equal halves of the shares for the confirm kernel's workers (A/B of conf_skew); dense scans: no run tables (A/B) */
    int tune_fused = 0;                    /* hsgpu_scratch_set_tuning (tests / tuning runs) */
    int tune_solo = 0;                     /* fused_only == 3: never the single-launch path for small batches; 4: whenever the geometry allows */
    DevBuf solo_ctl;                       /* solo scans: rec_counts | rec_super | ticket, left zeroed by the scan itself */
    const void *res_corpus = nullptr, *res_off = nullptr; /* where the batch this scratch took in last lives (reuse_resident): its device buffers, or the mapped small-batch area */
    uint8_t *h_small = nullptr, *d_small = nullptr; /* small host batches: mapped pinned {count | offsets | corpus | records}, read and written by the kernel itself */
    /* the small-batch server (hsgpu_scratch_set_server; scan_device.h, hwlm_server_kernel): one resident workgroup that scans what the
     * host puts into the mapped area above, without a launch per call */
    bool srv_enabled = false, srv_live = false;
    hipStream_t srv_stream = nullptr;
    HsgpuServerCtl *h_srv = nullptr, *d_srv = nullptr;
    /* the request side of the mailbox in DEVICE memory that the host writes through the PCIe BAR (fine-grained; devices with a
     * large BAR only): {request lines of an HsgpuServerCtl | offsets | corpus}, laid out like the mapped area. The workgroup then
     * polls and reads its batch locally -- a 1.5 KB request's round trip is 4.4 us instead of 10.5 (profiles/r06_bar_mailbox.txt) */
    uint8_t *bar_small = nullptr;
    HsgpuServerCtl *bar_ctl = nullptr; /* ... its request lines (behind offsets and corpus) */
    int bar_state = 0;          /* 0 not tried, 1 in use, -1 none (no large BAR, guard-page mode, or asked for: enable == 2) */
    bool srv_req_bar = false;   /* the live server reads its requests from bar_small */
    bool srv_host_mailbox = false; /* hsgpu_scratch_enable_server(s, 2, ..): the A/B */
    bool srv_debug = false;        /* hsgpu_debug_server_stamping: the requests ask for their stage stamps */
    const hsgpu_hwlm *srv_table = nullptr;
    uint32_t srv_seq = 0;
    unsigned srv_idle_us = 300; /* an idle server ends after this long: nothing it holds outlives a burst of calls by more */
    uint64_t srv_calls = 0, srv_launches = 0;
    bool solo_ctl_clean = false;
    int tune_unfolded = 0;                 /* fused_only == 2: two-phase with record_sort_kernel behind the confirm kernel */
    unsigned tune_wg_threads = 0, tune_wg_per_cu = 0;
    uint64_t cand_div = 64;                /* corpus bytes per candidate entry of capacity: 16 (room for every chunk) once a scan overflowed */
    unsigned dense_span = 0, dense_left = 0; /* dense mode lasts dense_span scans (doubling each time it is re-entered) */
    bool dense_unfolded = false;             /* a dense scan of the folded pipeline asked for "again" (more matches per position than its queue orders) */
    hsgpu_match_t *h_recs = nullptr;       /* pinned: hsgpu_hwlm_fetch_replay's landing area for the records */
    size_t h_recs_cap = 0;
    hipEvent_t ev_chunk[4] = {};           /* its D2H chunks */
    std::unique_ptr<WorkerPool> pool;      /* its replay threads (created on first use) */
    int n_cu = 0;
    size_t lds_per_cu = 0;
    bool in_use = false;
};

/* ---- device residency of compiled tables ---------------------------------- */

static int table_on_device(const hsgpu_hwlm *ct, int device, const uint8_t **out) {
    hsgpu_hwlm *t = const_cast<hsgpu_hwlm *>(ct);
    std::lock_guard<std::mutex> g(t->mu);
    auto it = t->dev_blob.find(device);
    if (it != t->dev_blob.end()) {
        *out = (const uint8_t *)it->second;
        return HSGPU_SUCCESS;
    }
    void *d = nullptr;
    int rv = hsgpu_dev_alloc(&d, t->blob.size());
    if (rv != HSGPU_SUCCESS) return rv;
    hipError_t e = hipMemcpy(d, t->blob.data(), t->blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hsgpu_dev_free(d);
        hsgpu_set_error("table upload failed: %s", hipGetErrorString(e));
        return HSGPU_UNKNOWN_ERROR;
    }
    t->dev_blob[device] = d;
    *out = (const uint8_t *)d;
    return HSGPU_SUCCESS;
}

void hsgpu_release_device_copies(hsgpu_hwlm *t) {
    std::lock_guard<std::mutex> g(t->mu);
    int cur = 0;
    bool have_cur = (hipGetDevice(&cur) == hipSuccess);
    for (auto &kv : t->dev_blob) {
        if (hipSetDevice(kv.first) == hipSuccess) hsgpu_dev_free(kv.second);
    }
    if (have_cur) (void)hipSetDevice(cur);
    t->dev_blob.clear();
}

/* ---- scratch ---------------------------------------------------------------- */

extern "C" int hsgpu_scratch_alloc(hsgpu_scratch_t **out, int device) {
    if (!out) return HSGPU_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        hsgpu_set_error("no HIP device available (%s)", hipGetErrorString(e));
        return HSGPU_UNKNOWN_ERROR;
    }
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return HSGPU_INVALID;
    HIP_TRY(hipSetDevice(device));
    hsgpu_scratch *s = new (std::nothrow) hsgpu_scratch;
    if (!s) return HSGPU_NOMEM;
    s->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || hipStreamCreate(&s->stream) != hipSuccess ||
        hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&s->h_count, 4 * sizeof(unsigned long long)) != hipSuccess) {
        hsgpu_set_error("scratch setup failed");
        hsgpu_scratch_free(s);
        return HSGPU_UNKNOWN_ERROR;
    }
    if (hipHostMalloc((void **)&s->h_note, sizeof(uint32_t), hipHostMallocMapped) == hipSuccess) {
        *s->h_note = 0;
        if (hipHostGetDevicePointer((void **)&s->d_note, s->h_note, 0) != hipSuccess) s->d_note = nullptr;
    } else {
        (void)hipGetLastError();
        s->h_note = nullptr;
    }
    s->n_cu = prop.multiProcessorCount;
    s->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
    if (s->count.ensure(sizeof(unsigned long long)) != HSGPU_SUCCESS ||
        s->stats.ensure(2 * sizeof(unsigned long long)) != HSGPU_SUCCESS ||
        hipMemset(s->stats.p, 0, 2 * sizeof(unsigned long long)) != hipSuccess) {
        hsgpu_scratch_free(s);
        return HSGPU_NOMEM;
    }
    *out = s;
    return HSGPU_SUCCESS;
}

static void server_stop(hsgpu_scratch *s);
extern "C" void hsgpu_scratch_free(hsgpu_scratch_t *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    server_stop(s);
    s->corpus.release();
    s->off.release();
    s->out.release();
    s->count.release();
    s->hint.release();
    s->cand.release();
    s->ctl.release();
    s->stats.release();
    s->tstamp.release();
    s->wg_stamps.release();
    s->conf_stamps.release();
    s->rec_stage.release();
    s->run_tab.release();
    s->solo_ctl.release();
    for (int i = 0; i < 2; i++) {
        s->pipe_corpus[i].release();
        s->pipe_off[i].release();
        s->pipe_out[i].release();
        if (s->h_rel[i]) (void)hipHostFree(s->h_rel[i]);
        if (s->h_prec[i]) (void)hipHostFree(s->h_prec[i]);
        if (s->ev_copied[i]) (void)hipEventDestroy(s->ev_copied[i]);
        if (s->ev_scanned[i]) (void)hipEventDestroy(s->ev_scanned[i]);
    }
    s->pipe_count.release();
    s->cs_bitmaps.release();
    s->cs_work.release();
    s->cs_counts.release();
    if (s->h_count) (void)hipHostFree(s->h_count);
    if (s->h_note) (void)hipHostFree(s->h_note);
    if (s->h_small) (void)hipHostFree(s->h_small);
    if (s->h_srv) (void)hipHostFree(s->h_srv);
    if (s->bar_small) (void)hipFree(s->bar_small);
    if (s->srv_stream) (void)hipStreamDestroy(s->srv_stream);
    if (s->h_recs) (void)hipHostFree(s->h_recs);
    for (int i = 0; i < 4; i++)
        if (s->ev_chunk[i]) (void)hipEventDestroy(s->ev_chunk[i]);
    s->pool.reset();
    for (int r = 0; r < hsgpu_scratch::kRing; r++)
        for (int i = 0; i < 4; i++)
            if (s->ev_ring[r][i]) (void)hipEventDestroy(s->ev_ring[r][i]);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" int hsgpu_scratch_enable_timing(hsgpu_scratch_t *s, int enable) {
    if (!s) return HSGPU_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    if (enable && !s->ev_ring[0][0]) {
        for (int r = 0; r < hsgpu_scratch::kRing; r++)
            for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreate(&s->ev_ring[r][i]));
    }
    if (enable) {
        int rv = s->tstamp.ensure(hsgpu_scratch::kRing * 4 * sizeof(unsigned long long));
        if (rv != HSGPU_SUCCESS) return rv;
        std::vector<unsigned long long> init(hsgpu_scratch::kRing * 4, 0);
        for (int r = 0; r < hsgpu_scratch::kRing; r++) init[4 * r] = ~0ull;
        HIP_TRY(hipMemcpy(s->tstamp.p, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device) == hipSuccess && khz > 0)
            s->wall_clock_khz = khz;
    }
    s->timing = enable != 0;
    s->timing_wg = enable == 2;
    s->n_timed = 0;
    return HSGPU_SUCCESS;
}

/* tuning: the per-workgroup stamps of the last scan's filter kernel, in milliseconds relative to the earliest start:
 * out[4 * w + {0: start, 1: prologue done, 2: wavefront 0's share done, 3: end}] (synchronises the device) */
extern "C" int hsgpu_scratch_get_wg_stamps(hsgpu_scratch_t *s, float *out, unsigned max_wgs, unsigned *n_wgs) {
    if (!s || !n_wgs || (max_wgs && !out)) return HSGPU_INVALID;
    *n_wgs = s->wg_stamps_n;
    if (!s->wg_stamps_n || !s->wg_stamps.p) return HSGPU_SUCCESS;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> v((size_t)s->wg_stamps_n * 4);
    HIP_TRY(hipMemcpy(v.data(), s->wg_stamps.p, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < s->wg_stamps_n; w++) t0 = std::min(t0, v[4 * w]);
    for (size_t w = 0; w < std::min<size_t>(s->wg_stamps_n, max_wgs); w++)
        for (int k = 0; k < 4; k++) out[4 * w + k] = (float)((double)(v[4 * w + k] - t0) / s->wall_clock_khz);
    return HSGPU_SUCCESS;
}

/* tuning builds (HSGPU_CONFIRM_STAMPS=1): the confirm kernel's per-worker stamps of the last scan:
 * out[6 * w + {0: start, 1: end (ms from the earliest start), 2: fresh steps, 3: steps on the rest queue, 4: sorted drains, 5: entries}] */
extern "C" int hsgpu_scratch_get_conf_stamps(hsgpu_scratch_t *s, float *out, unsigned max_workers, unsigned *n_workers) {
    if (!s || !n_workers || (max_workers && !out)) return HSGPU_INVALID;
    *n_workers = s->conf_stamps_n;
    if (!s->conf_stamps_n || !s->conf_stamps.p) return HSGPU_SUCCESS;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> v((size_t)s->conf_stamps_n * 4);
    HIP_TRY(hipMemcpy(v.data(), s->conf_stamps.p, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < s->conf_stamps_n; w++)
        if (v[4 * w]) t0 = std::min(t0, v[4 * w]);
    for (size_t w = 0; w < std::min<size_t>(s->conf_stamps_n, max_workers); w++) {
        out[6 * w] = v[4 * w] ? (float)((double)(v[4 * w] - t0) / s->wall_clock_khz) : -1.f;
        out[6 * w + 1] = v[4 * w + 3] ? (float)((double)(v[4 * w + 3] - t0) / s->wall_clock_khz) : -1.f;
        out[6 * w + 2] = (float)(v[4 * w + 1] & 0xffff);
        out[6 * w + 3] = (float)(v[4 * w + 1] >> 16 & 0xffff);
        out[6 * w + 4] = (float)(v[4 * w + 1] >> 32);
        out[6 * w + 5] = (float)v[4 * w + 2];
    }
    return HSGPU_SUCCESS;
}

/* device wall-clock stamps of one timed scan (synchronises the device: tuning / bench API) */
static int read_stamps(hsgpu_scratch_t *s, unsigned back, unsigned long long t[4]) {
    if (!s || back >= hsgpu_scratch::kRing || back >= s->n_timed) return HSGPU_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t slot = (s->n_timed - 1 - back) % hsgpu_scratch::kRing;
    HIP_TRY(hipMemcpy(t, (const unsigned long long *)s->tstamp.p + 4 * slot, 4 * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_scratch_get_kernel_span(hsgpu_scratch_t *s, unsigned back, float *filter_ms) {
    /* the filter kernel's own execution span (first workgroup start to last workgroup
     * end) from the device wall clock -- what a kernel trace reports as its duration */
    if (!filter_ms) return HSGPU_INVALID;
    unsigned long long t[4];
    int rv = read_stamps(s, back, t);
    if (rv != HSGPU_SUCCESS) return rv;
    if (t[1] < t[0]) return HSGPU_INVALID;
    *filter_ms = (float)((double)(t[1] - t[0]) / s->wall_clock_khz);
    return HSGPU_SUCCESS;
}

/* filter_ms: HIP events recorded on the launch stream right before and after the filter
 * kernel (the only two events a timed scan records: every extra event costs ~5 us of
 * dispatch gap, and four of them were 6% of a 0.3 ms scan). confirm_ms (filter end ->
 * confirm stage end) and total_ms (filter start -> end of the scan's last kernel) come from
 * the device wall clock the kernels stamp themselves. */
extern "C" int hsgpu_scratch_get_timing(hsgpu_scratch_t *s, unsigned back, float *filter_ms, float *confirm_ms,
                                        float *total_ms) {
    unsigned long long t[4];
    int rv = read_stamps(s, back, t);
    if (rv != HSGPU_SUCCESS) return rv;
    hipEvent_t *ev = s->ev_ring[(s->n_timed - 1 - back) % hsgpu_scratch::kRing];
    float f = 0;
    HIP_TRY(hipEventElapsedTime(&f, ev[0], ev[1]));
    if (filter_ms) *filter_ms = f;
    if (confirm_ms) *confirm_ms = (t[2] > t[1]) ? (float)((double)(t[2] - t[1]) / s->wall_clock_khz) : 0.f;
    if (total_ms) *total_ms = (t[3] > t[0] && t[0] != ~0ull) ? (float)((double)(t[3] - t[0]) / s->wall_clock_khz) : 0.f;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_scratch_get_stats(hsgpu_scratch_t *s, uint64_t *cand_entries, int *overflowed) {
    /* synchronous; for tuning and tests: 32-byte candidate entries spilled by the
     * two-phase scans since the previous call, and how many of those scans overflowed
     * a candidate region (and were redone by the fused fallback kernel) */
    if (!s) return HSGPU_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long cur[2] = {0, 0};
    HIP_TRY(hipMemcpy(cur, s->stats.p, sizeof(cur), hipMemcpyDeviceToHost));
    if (cand_entries) *cand_entries = cur[0] - s->stats_seen[0];
    if (overflowed) *overflowed = (int)(cur[1] - s->stats_seen[1]);
    s->stats_seen[0] = cur[0];
    s->stats_seen[1] = cur[1];
    return HSGPU_SUCCESS;
}

/* ---- the launch --------------------------------------------------------------- */

static int set_dyn_lds(const void *fn, size_t lds) {
    /* once per (function, size): raising the dynamic-LDS limit past 64 KiB */
    static std::mutex mu;
    static std::map<const void *, size_t> done;
    std::lock_guard<std::mutex> g(mu);
    auto it = done.find(fn);
    if (it != done.end() && it->second >= lds) return HSGPU_SUCCESS;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done[fn] = lds;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_scratch_set_tuning(hsgpu_scratch_t *s, int fused_only, unsigned wg_threads, unsigned wg_per_cu) {
    if (!s || (wg_threads && (wg_threads % 64 || wg_threads < 256 || wg_threads > 1024)) || wg_per_cu > 4) return HSGPU_INVALID;
    server_stop(s); /* (a resident server was sized for the old geometry) */
    s->tune_fused = fused_only == 1;
    s->tune_unfolded = fused_only == 2;
    s->tune_solo = fused_only == 3 ? 1 : fused_only == 4 ? 2 : 0;
    s->tune_no_skew = fused_only == 5;
    s->tune_wg_threads = wg_threads;
    s->tune_wg_per_cu = wg_per_cu;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_scratch_enable_server(hsgpu_scratch_t *s, int enable, unsigned idle_us) {
    if (!s) return HSGPU_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    if (!enable || (enable == 2) != s->srv_host_mailbox) server_stop(s);
    s->srv_enabled = enable != 0;
    s->srv_host_mailbox = enable == 2;
    if (idle_us) s->srv_idle_us = std::min(idle_us, 1000000u);
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_scratch_server_last_us(hsgpu_scratch_t *s, float *copy_us, float *scan_us) {
    if (!s || !s->h_srv) return HSGPU_INVALID;
    if (copy_us) *copy_us = (float)s->h_srv->done_copy_ticks / 100.f; /* 100 MHz ticks */
    if (scan_us) *scan_us = (float)s->h_srv->done_body_ticks / 100.f;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_debug_server_stamping(hsgpu_scratch_t *s, int on) {
    if (!s) return HSGPU_INVALID;
    s->srv_debug = on != 0;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_debug_server_stamps(hsgpu_scratch_t *s, float *us) {
    if (!s || !s->h_srv || !us) return HSGPU_INVALID;
    const volatile unsigned long long *st = (const volatile unsigned long long *)s->h_srv->stamps;
    for (int i = 0; i < 3; i++) us[i] = (float)(long long)(st[i + 1] - st[i]) / 100.f; /* 100 MHz ticks */
    return HSGPU_SUCCESS;
}

/* ... and, from the body's start: us[0] -> the first loads are out, us[1] -> the wavefront stands in front of the body's first barrier */
extern "C" int hsgpu_debug_server_head_stamps(hsgpu_scratch_t *s, float *us) {
    if (!s || !s->h_srv || !us) return HSGPU_INVALID;
    const volatile unsigned long long *st = (const volatile unsigned long long *)s->h_srv->stamps;
    us[0] = (float)(long long)(st[4] - st[0]) / 100.f;
    us[1] = (float)(long long)(st[5] - st[0]) / 100.f;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_scratch_server_stats(hsgpu_scratch_t *s, uint64_t *calls, uint64_t *launches, int *live) {
    if (!s) return HSGPU_INVALID;
    if (calls) *calls = s->srv_calls;
    if (launches) *launches = s->srv_launches;
    if (live) *live = s->srv_live && !__atomic_load_n(&s->h_srv->exited, __ATOMIC_ACQUIRE);
    return HSGPU_SUCCESS;
}

extern "C" unsigned hsgpu_confirm_partition(unsigned n_shares, unsigned max_workers, unsigned *q_out, unsigned *k_out) {
    if (!n_shares || !max_workers) return 0;
    double best = -1;
    unsigned bq = 1, bk = 1;
    for (unsigned q = 1; q <= 8; q++) {
        const uint64_t parts = (uint64_t)n_shares * q, k = (parts + max_workers - 1) / max_workers;
        const double eff = (double)parts / (double)(k * max_workers) - 1e-3 * q; /* the slots busy in steady state; ties: fewer parts */
        if (eff > best) best = eff, bq = q, bk = (unsigned)k;
    }
    if (q_out) *q_out = bq;
    if (k_out) *k_out = bk;
    return (unsigned)(((uint64_t)n_shares * bq + bk - 1) / bk);
}

/* how many confirm workgroups the device holds at once (per kernel instantiation: the register count differs) */
static unsigned confirm_resident_workgroups(hsgpu_scratch *s, const void *f_conf, size_t gate_lds) {
    static std::mutex mu;
    static std::map<std::pair<const void *, size_t>, int> per_cu;
    std::lock_guard<std::mutex> g(mu);
    const std::pair<const void *, size_t> key(f_conf, gate_lds);
    auto it = per_cu.find(key);
    if (it == per_cu.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f_conf, HSGPU_CONFIRM_THREADS, gate_lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 4;
        }
        it = per_cu.emplace(key, nb).first;
    }
    return (unsigned)it->second * (unsigned)std::max(1, s->n_cu);
}

/* The arguments and buffers of a SOLO scan -- the fused kernel with the placement in its last workgroup (scan_device.h, solo_tail):
 * one launch for a small batch (launch_scan), or the body the small-batch server runs per request (server_start). `args` comes in
 * with the table fields set; total / cap: the largest batch and record count this setup has to hold. */
static int solo_setup(hsgpu_scratch *s, HsgpuScanArgs &args, uint64_t total, uint64_t cap, unsigned solo_grid, unsigned wg_threads,
                      hipStream_t stream) {
    int rv;
    const uint32_t n_reg = solo_grid * (wg_threads / 64);
    /* (the hints: every wavefront writes those of its own tiles in the kernel's prologue) */
    args.n_hint = (total >> HSGPU_HINT_SHIFT) + 1;
    if ((rv = s->hint.ensure(args.n_hint * sizeof(uint32_t))) != HSGPU_SUCCESS) return rv;
    args.hint = (const uint32_t *)s->hint.p;
    args.hint_in_filter = 0;
    args.fold = 0;
    args.conf_q = args.conf_k = 1;
    args.conf_spread = 0, args.conf_skew = 0;
    args.run_tab = nullptr;
    args.img_keep_words = 0;
    args.srv_inline_at = 0;
    args.conf_cus = (uint32_t)std::max(1, s->n_cu);
    args.cand = nullptr;
    args.cand_cap = 0;
    args.cand_waves = 0;
    args.cand_counts = nullptr;
    args.rec_regions = n_reg;
    args.rec_cap = (uint32_t)std::min<uint64_t>(1u << 30, std::max<uint64_t>(256, 2 * (cap / n_reg + 1)));
    if ((rv = s->rec_stage.ensure((uint64_t)args.rec_cap * n_reg * sizeof(uint4))) != HSGPU_SUCCESS) return rv;
    args.rec_stage = (uint4 *)s->rec_stage.p;
    const size_t super_ofs = ((size_t)2 * 1024 + 1) & ~(size_t)1, ticket_ofs = super_ofs + 2 * HSGPU_SUPER_WORDS;
    const size_t ctl_words = (ticket_ofs + 4) & ~(size_t)3;
    const size_t cap_before = s->solo_ctl.cap;
    if ((rv = s->solo_ctl.ensure(ctl_words * sizeof(uint32_t))) != HSGPU_SUCCESS) return rv;
    if (s->solo_ctl.cap != cap_before || !s->solo_ctl_clean) HIP_TRY(hipMemsetAsync(s->solo_ctl.p, 0, s->solo_ctl.cap, stream));
    s->solo_ctl_clean = true; /* (the scan's last workgroup leaves it zeroed) */
    args.rec_counts = (uint32_t *)s->solo_ctl.p;
    args.rec_super = (unsigned long long *)((uint32_t *)s->solo_ctl.p + super_ofs);
    args.solo_ticket = (uint32_t *)s->solo_ctl.p + ticket_ofs;
    args.solo_ctl_words = (uint32_t)ctl_words;
    args.super_shift = 5;
    args.group_regions = 1;
    args.ctl_other = nullptr;
    args.ctl_other_words = 0;
    args.stats = (unsigned long long *)s->stats.p;
    args.overflow_note = s->d_note;
    args.solo = 1;
    args.tstamp = nullptr;
    args.tstamp_next = nullptr;
    args.wg_stamps = nullptr;
    args.conf_stamps = nullptr;
    return HSGPU_SUCCESS;
}

/* the table fields every kernel reads from its arguments (no dependent read of the header) */
static void table_args(const HsgpuTableHeader *h, HsgpuScanArgs &args) {
    args.t_flags = h->flags;
    args.fold_shift = (h->flags & (HSGPU_F_BFOLD | HSGPU_F_PAIR)) ? 16u : 0u; /* one filter test stands for every key class */
    args.t_hash_mask = h->hash_mask;
    args.t_filter_log2 = h->filter_log2;
    args.t_ht_a_log2 = h->ht_a_log2;
    args.t_ht_b_log2 = h->ht_b_log2;
    args.t_off_filter = h->off_filter;
    args.t_off_c2bits = h->off_c2bits;
    args.t_off_ht_a = h->off_ht_a;
    args.t_off_ht_b = h->off_ht_b;
    args.t_off_c2ref = h->off_c2ref;
    args.t_off_lists = h->off_lists;
    args.t_off_lits = h->off_lits;
}

/* the filter workgroup's size for a table (launch_scan's choice, also the server's) */
static unsigned filter_wg_threads(const HsgpuTableHeader *h, const hsgpu_scratch *s, bool *small_out) {
    const bool small = hsgpu_filter_lds_bytes(h->flags, h->filter_log2, true, 512) * 3 <= s->lds_per_cu;
    const bool light = (h->flags & HSGPU_F_STRIDE2) && !(h->flags & (HSGPU_F_K2 | HSGPU_F_HAS_C | HSGPU_F_PAIR));
    if (small_out) *small_out = small;
    unsigned wg_threads = small ? 512 : light ? 768 : HSGPU_WG_THREADS;
    if (s->tune_wg_threads) wg_threads = s->tune_wg_threads; /* hsgpu_scratch_set_tuning */
    return wg_threads;
}

/* ---- the small-batch server: host side (scan_device.h, hwlm_server_kernel) ------------------------------------------------ */
/* the request lines of the mailbox as the host writes them: in the BAR area when the live server reads them there */
constexpr uint32_t SRV_POISON_COUNT = ~0u - 1u;
static inline HsgpuServerCtl *server_req(hsgpu_scratch *s) { return s->srv_req_bar ? s->bar_ctl : s->h_srv; }
static inline void bar_flush() {
#if defined(__x86_64__)
    __builtin_ia32_sfence(); /* (device memory through the BAR is write-combining: without it a store sits in the CPU's buffers) */
#endif
}
static void server_stop(hsgpu_scratch *s) { /* ends a live server at once and waits until it has gone */
    if (!s->srv_live) return;
    __atomic_store_n(&server_req(s)->stop, 1u, __ATOMIC_RELEASE);
    bar_flush();
    (void)hipStreamSynchronize(s->srv_stream);
    server_req(s)->stop = 0;
    bar_flush();
    s->h_srv->exited = 0;
    s->srv_live = false;
    s->srv_table = nullptr;
}

static int launch_scan(const hsgpu_hwlm *t, hsgpu_scratch *s, const HsgpuScanArgs &a, hipStream_t stream) {
    const HsgpuTableHeader *h = t->hdr();
    const void *f_two = hsgpu_filter_kernel_for(h->flags, false);
    const void *f_fused = hsgpu_filter_kernel_for(h->flags, true);
    const void *f_conf = hsgpu_confirm_kernel_for(h->flags, false);
    if (!f_two || !f_fused || !f_conf) {
        hsgpu_set_error("no kernel for table flags %u", h->flags);
        return HSGPU_UNKNOWN_ERROR;
    }
    /* Geometry: big tables (up to 128 KiB of filter) run one 16-wavefront workgroup per
     * CU; tables of <= 40 KiB run three 8-wavefront workgroups per CU (24 wavefronts
     * hide more latency; the SGPR budget admits no second 16-wavefront workgroup). */
    server_stop(s); /* (a resident server works in this scratch's buffers) */
    const bool small = hsgpu_filter_lds_bytes(h->flags, h->filter_log2, true, 512) * 3 <= s->lds_per_cu;
    /* Stride-2 single-bit filters are light enough (~3 VALU instructions per byte) that the
     * kernel is bound by the memory side, and there 8 wavefronts per CU measured 7-13% faster
     * than 16 (teddy64: 0.269 vs 0.288 ms; 1000 literals: 0.269 vs 0.309 ms); the VALU-bound
     * variants (stride 1, two filter bits) want all 16 (fdr10k: 0.39 vs 0.45 ms). */
    const bool light = (h->flags & HSGPU_F_STRIDE2) && !(h->flags & (HSGPU_F_K2 | HSGPU_F_HAS_C | HSGPU_F_PAIR));
    /* Round 5: the light filters run TWELVE wavefronts per CU, three per SIMD -- the sizes in between had never been tried.
     * teddy64, 1 GiB, one box (tools/kbench.py, HSGPU_WG_THREADS): 384 threads 0.2676 ms, 512 0.2528, 640 0.2430, 704 0.2141,
     * 768 0.2062, 832 0.2212, 896 0.2412, 960 0.2389, 1024 0.2641; the 10 000-literal filter: 704 0.3139, 768 0.3088, 832
     * 0.3320, 896 0.3174, 960 0.3050, 1024 0.2919 (profiles/r05_wg_threads_sweep.txt). */
    unsigned wg_threads = small ? 512 : light ? 768 : HSGPU_WG_THREADS;
    unsigned wg_per_cu = small ? 3 : 1;
    if (s->tune_wg_threads) wg_threads = s->tune_wg_threads; /* hsgpu_scratch_set_tuning */
    if (!small) { /* a 64 KiB filter admits a second workgroup when the kernel's registers do */
        int nb = 0;
        const size_t l2 = hsgpu_filter_lds_bytes(h->flags, h->filter_log2, false, wg_threads);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f_two, (int)wg_threads, l2) == hipSuccess && nb >= 2)
            wg_per_cu = 2;
    }
    if (s->tune_wg_per_cu) wg_per_cu = s->tune_wg_per_cu;
    const uint32_t super_shift = wg_threads <= 256 ? 12 : wg_threads <= 512 ? 13 : 14; /* >= 1 KiB per wavefront: the grid of a small corpus */
    const size_t lds = hsgpu_filter_lds_bytes(h->flags, h->filter_log2, true, wg_threads);      /* fused */
    const size_t lds_two = hsgpu_filter_lds_bytes(h->flags, h->filter_log2, false, wg_threads); /* two-phase filter */
    if (lds > s->lds_per_cu) {
        hsgpu_set_error("filter needs %zu bytes of LDS, device has %zu", lds, s->lds_per_cu);
        return HSGPU_UNKNOWN_ERROR;
    }
    const uint64_t n_tiles = (a.total + (1ull << super_shift) - 1) >> super_shift;
    if (n_tiles == 0) return HSGPU_SUCCESS;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)s->n_cu * wg_per_cu);

    HsgpuScanArgs args = a;
    args.solo = 0;
    args.solo_ctl_words = 0;
    args.solo_ticket = nullptr;
    table_args(h, args);
    int rv;
    /* Small batches: ONE launch (the fused kernel with the placement in its last workgroup; scan_device.h, solo_tail). The
     * reference serves short blocks from dedicated small matchers (src/rose/block.c:382-391, src/runtime.c:401-413); here
     * three launches were ~20 us of fixed cost per scan whatever its size. Up to SOLO_BYTES by default -- measured, resident
     * scans without timing events, solo vs three kernels: 1 460 B 10.4 vs 15.6 us, 16 KiB 17.0 vs 20.9, 64 KiB 20.2 vs 21.2,
     * 256 KiB 28.9 vs 22.1 (the in-kernel confirm's dependent reads and a placement by one workgroup stop paying) --, the default
     * pipeline only, not in dense mode. */
    constexpr uint64_t SOLO_BYTES = 64ull << 10;
    const bool solo_ok = !s->tune_fused && !s->tune_unfolded && s->tune_solo != 1 && s->cand_div == 64 && !(s->h_note && *s->h_note) &&
                         (size_t)hsgpu_filter_words(h->flags, h->filter_log2) * 4 >= 28 * 1024;
    if (solo_ok && (a.total <= SOLO_BYTES || s->tune_solo == 2)) {
        /* 16 KiB (8 KiB for 512-thread workgroups) per workgroup; forced on big corpora (tests): at most 1024 regions */
        const unsigned solo_grid = (unsigned)std::min<uint64_t>(n_tiles, 1024 / (wg_threads / 64));
        if ((rv = solo_setup(s, args, a.total, a.cap, solo_grid, wg_threads, stream)) != HSGPU_SUCCESS) return rv;
        if (s->timing) {
            const size_t slot = s->n_timed % hsgpu_scratch::kRing;
            s->ev_t = s->ev_ring[slot];
            args.tstamp = (unsigned long long *)s->tstamp.p + 4 * slot;
            args.tstamp_next = (unsigned long long *)s->tstamp.p + 4 * ((slot + 1) % hsgpu_scratch::kRing);
            HIP_TRY(hipEventRecord(s->ev_t[0], stream));
        }
        void *skargs[] = {&args};
        if ((rv = set_dyn_lds(f_fused, lds)) != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipLaunchKernel(f_fused, dim3(solo_grid), dim3(wg_threads), skargs, lds, stream));
        if (s->timing) {
            HIP_TRY(hipEventRecord(s->ev_t[1], stream));
            s->n_timed++;
        }
        return HSGPU_SUCCESS;
    }
    /* phase 0: per-KiB block hints */
    args.n_hint = (a.total >> HSGPU_HINT_SHIFT) + 1;
    if ((rv = s->hint.ensure(args.n_hint * sizeof(uint32_t))) != HSGPU_SUCCESS) return rv;
    args.hint = (const uint32_t *)s->hint.p;
    const bool two_phase = !s->tune_fused;
    /* Block hints are only needed by the confirm step. Two-phase: the filter kernel writes
     * them in its prologue (while its LDS image loads). Fused: the filter itself needs
     * them, so a hint kernel runs first on the same stream. */
    auto launch_hints = [&](hipStream_t hs) -> int {
        const uint64_t *off = a.off;
        uint64_t nblocks = a.nblocks, total = a.total, n_hint = args.n_hint;
        uint32_t *hint = (uint32_t *)s->hint.p;
        void *hargs[] = {&off, &nblocks, &total, &hint, &n_hint};
        HIP_TRY(hipLaunchKernel(hsgpu_hint_kernel(), dim3((unsigned)((nblocks + 1 + 255) / 256)), dim3(256), hargs, 0, hs));
        return HSGPU_SUCCESS;
    };
    args.hint_in_filter = two_phase ? 1u : 0u;
    if (!two_phase && (rv = launch_hints(stream)) != HSGPU_SUCCESS) return rv;
    /* Dense input (the reference's flood case, src/fdr/flood_runtime.h:86-335): once a scan on this scratch ran out of
     * candidate room (and told its caller to scan again), later scans give every 16-byte chunk an entry of its own -- the
     * two-phase path can then not overflow, at the price of a candidate buffer twice the size of the corpus. */
    if (two_phase) {
        if (s->h_note && *s->h_note) {
            /* (an overflow seen: dense mode for dense_span scans, twice as long every time it has to be re-entered) */
            *s->h_note = 0;
            if (s->cand_div == 16) s->dense_unfolded = true; /* candidates cannot overflow in dense mode: the folded confirm kernel emitted out of order */
            s->cand_div = 16;
            s->dense_span = std::min<unsigned>(s->dense_span ? s->dense_span * 2 : 16, 1u << 16);
            s->dense_left = s->dense_span;
        } else if (s->cand_div == 16 && s->dense_left && --s->dense_left == 0) {
            /* Dense mode is not for ever (it costs a candidate buffer twice the size of the corpus and the unfolded pipeline):
             * after dense_span scans without a note the scratch tries the ordinary sizing again and gives the big buffer back.
             * If the input is still dense that scan reports "again", sets the note, and the span doubles. */
            s->cand_div = 64;
            s->dense_unfolded = false;
            s->cand.release();
        }
    }
    /* The folded pipeline: the confirm wavefronts emit their regions in order and place them themselves. Not for fused-only
     * scratches and not without the mapped "again" note (the fused kernel then has to run between confirm and sort). In dense
     * mode (fold = 2) dense batches are confirmed position by position; a set with more matches per position than the queue
     * orders that way makes the scan say "again" once more, and the scratch goes on with record_sort_kernel, which sorts
     * whatever it is given. */
    /* the confirm kernel's gate in dynamic LDS: the 64 Kbit key gate, or the opt-in Bloom gate's three planes (table.h) */
    const size_t gate_lds = (h->flags & HSGPU_F_PAIR) ? 16 : (h->flags & HSGPU_F_BLOOM) ? (size_t)HSGPU_BLOOM_WORDS * 4 : 8192;
    const bool fold = two_phase && s->d_note && !s->tune_unfolded && !(s->cand_div == 16 && s->dense_unfolded);
    args.fold = fold ? (s->cand_div == 16 ? 2u : 1u) : 0u;
    if (args.fold == 2 && !(f_conf = hsgpu_confirm_kernel_for(h->flags, true))) return HSGPU_INVALID;
    /* staged match records: one region per producing wavefront -- the confirm kernel's workers (two-phase) or the filter
     * wavefronts (fused) --, packed into the caller's buffer in delivery order. 2x headroom over an even split. */
    const uint32_t n_waves = grid * (wg_threads / 64);
    uint32_t n_rec = n_waves * HSGPU_CONFIRM_SPLIT; /* the fused kernel uses the first n_waves of them */
    unsigned conf_grid = 0;
    args.conf_q = args.conf_k = 1;
    args.conf_spread = 0, args.conf_skew = 0;
    args.run_tab = nullptr;
    args.img_keep_words = 0;
    args.srv_inline_at = 0;
    args.conf_cus = (uint32_t)std::max(1, s->n_cu);
    if (two_phase) {
        /* The confirm kernel's partition: share = one filter wavefront's candidates, cut into Q parts, K consecutive parts per
         * worker wavefront, so that the parts go round the workers the device holds at once as evenly as whole numbers allow
         * (4 096 shares on 8 192 workers: Q = 2, K = 1; on 6 144: Q = 3, K = 2). */
        const unsigned w_max = confirm_resident_workgroups(s, f_conf, gate_lds) * (HSGPU_CONFIRM_THREADS / 64);
        unsigned q = 1, k = 1;
        const unsigned workers = hsgpu_confirm_partition(n_waves, w_max, &q, &k);
        args.conf_q = q, args.conf_k = k;
        args.conf_cus = (uint32_t)std::max(1, s->n_cu);
        conf_grid = (unsigned)((workers + HSGPU_CONFIRM_THREADS / 64 - 1) / (HSGPU_CONFIRM_THREADS / 64));
        n_rec = conf_grid * (HSGPU_CONFIRM_THREADS / 64);
#ifndef HSGPU_CONFIRM_SKEW
#define HSGPU_CONFIRM_SKEW 3473 /* 0.053 x 2^16: the oldest of eight ranks takes 0.553 of its share, the fourth 0.508 (profiles/r05_confirm_workers.txt) */
#endif
        /* the headline's geometry -- every share in two parts, a part per worker, a whole even number of workgroups per CU: the halves
         * of a share go to workers of mirrored dispatch ranks, unequal (hwlm_confirm_kernel, conf_skew) */
        if (args.fold == 1 && !(h->flags & HSGPU_F_PAIR) && !s->tune_no_skew && q == 2 && k == 1 && (uint64_t)conf_grid * (HSGPU_CONFIRM_THREADS / 64) == (uint64_t)n_waves * 2 &&
            s->n_cu > 0 && conf_grid % (unsigned)s->n_cu == 0 && (conf_grid / (unsigned)s->n_cu) % 2 == 0)
            args.conf_skew = HSGPU_CONFIRM_SKEW;
        if (args.fold == 2 && !(h->flags & HSGPU_F_PAIR)) {
            /* Dense scans: parts of half a batch or a few (a dense half batch is 1 024 lookup positions, each of which may match), a
             * region per part, the parts of a worker spread over the corpus (hwlm_confirm_kernel). */
            const uint64_t share_entries = ((a.total >> 10) + n_waves - 1) / n_waves * 64 + 64; /* = cand_cap below */
            const unsigned nb = (unsigned)std::min<uint64_t>((share_entries + 63) / 64, 1u << 20); /* half batches */
            q = std::max(1u, std::min(nb, 32u));
            const uint64_t parts = (uint64_t)n_waves * q;
            if (parts < (1u << 24)) {
                const unsigned w_use = (unsigned)std::min<uint64_t>(w_max, parts);
                k = (unsigned)((parts + w_use - 1) / w_use);
                const unsigned wk = (unsigned)((parts + k - 1) / k);
                conf_grid = (wk + HSGPU_CONFIRM_THREADS / 64 - 1) / (HSGPU_CONFIRM_THREADS / 64);
                n_rec = (uint32_t)parts;
                args.conf_q = q, args.conf_k = k;
                args.conf_spread = 1;
                /* the run tables (hwlm_confirm_kernel: a run of one byte value costs one confirm and a descriptor) */
                if (!s->tune_no_skew) { /* (hsgpu_scratch_set_tuning(s, 5, ..): every record staged, the A/B) */
                    if ((rv = s->run_tab.ensure(parts * HSGPU_RUN_STRIDE * sizeof(uint4))) != HSGPU_SUCCESS) return rv;
                    args.run_tab = (uint4 *)s->run_tab.p;
                }
            }
        }
        if (!s->d_note) n_rec = std::max(n_rec, n_waves); /* (the fused kernel behind the confirm kernel writes one region per filter wavefront) */
    }
    args.rec_regions = n_rec;
    /* (dense mode has a region per PART -- tens of thousands of them for a 64 MiB chunk: a floor of 256 records each was 285 MiB
     * of staging whatever the caller's cap said; advisor, round 5. A part that outgrows its region says "again" like any other.) */
    args.rec_cap = (uint32_t)std::min<uint64_t>(1u << 30, std::max<uint64_t>(args.conf_spread ? 64 : 256, 2 * (a.cap / n_rec + 1)));
    if ((rv = s->rec_stage.ensure((uint64_t)args.rec_cap * n_rec * sizeof(uint4))) != HSGPU_SUCCESS) return rv;
    /* Two control blocks that alternate from scan to scan, each rec_counts[2 n_rec] | cand_counts[n_waves + 1] |
     * rec_super[257] (64-bit): a scan works in one and its last kernel zeroes the other (the previous scan's), so
     * the next scan finds its block zeroed without a memset. */
    const size_t cand_ofs = (size_t)2 * n_rec;
    const size_t super_ofs = (cand_ofs + n_waves + 1 + 1) & ~(size_t)1; /* 8-byte aligned */
    const size_t blk_words = (super_ofs + 2 * HSGPU_SUPER_WORDS + 3) & ~(size_t)3;
    /* a reallocated control buffer is garbage whatever its address: hipMalloc may hand the
     * freed range straight back, so growth is detected by capacity, never by pointer */
    const size_t ctl_cap_before = s->ctl.cap;
    if ((rv = s->ctl.ensure(2 * blk_words * sizeof(uint32_t))) != HSGPU_SUCCESS) return rv;
    if (s->ctl.cap != ctl_cap_before) s->ctl_clean = false;
    const size_t half_words = (s->ctl.cap / 8) & ~(size_t)3; /* the second block starts at the same place whatever this scan's size */
    if (!s->ctl_clean) { /* a fresh (or possibly dirty) buffer: both blocks */
        HIP_TRY(hipMemsetAsync(s->ctl.p, 0, s->ctl.cap, stream));
        s->ctl_parity = 0;
    }
    s->ctl_clean = false; /* until this scan's record_sort_kernel has been queued */
    uint32_t *blk = (uint32_t *)s->ctl.p + (s->ctl_parity ? half_words : 0);
    args.ctl_other = (uint32_t *)s->ctl.p + (s->ctl_parity ? 0 : half_words);
    args.ctl_other_words = (uint32_t)half_words;
    args.rec_stage = (uint4 *)s->rec_stage.p;
    args.rec_counts = blk;
    args.rec_super = (unsigned long long *)(blk + super_ofs);
    /* up to 256 supers of 2^super_shift regions, one atomic per region (atomics on one address go one after the other) */
    args.super_shift = 5;
    while (((n_rec + (1u << args.super_shift) - 1) >> args.super_shift) > 256) args.super_shift++;
    /* consecutive regions that one record_sort_kernel workgroup places (consecutive pieces of the corpus) */
    args.group_regions = HSGPU_CONFIRM_SPLIT;
#ifndef HSGPU_FOLD_GROUP
#define HSGPU_FOLD_GROUP 4 /* folded pipeline: regions (consecutive sorted runs) that one gather workgroup places; tuning builds: the gather of
                            * the headline step's 8 192 regions takes 14.3 us with groups of 2, 12.9 with 4, 13.5 with 8, 18.2 with 16, 35.0 with 32 */
#endif
    if (fold) args.group_regions = HSGPU_FOLD_GROUP;
    args.stats = (unsigned long long *)s->stats.p;
    args.overflow_note = s->d_note;

    void *kargs[] = {&args};
    args.tstamp = nullptr;
    args.tstamp_next = nullptr;
    args.wg_stamps = nullptr;
    args.conf_stamps = nullptr;
    if (s->timing_wg && two_phase) {
        if ((rv = s->wg_stamps.ensure((size_t)grid * 4 * sizeof(unsigned long long))) != HSGPU_SUCCESS) return rv;
        args.wg_stamps = (unsigned long long *)s->wg_stamps.p;
        s->wg_stamps_n = grid;
        if ((rv = s->conf_stamps.ensure((size_t)n_rec * 4 * sizeof(unsigned long long))) != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipMemsetAsync(s->conf_stamps.p, 0, (size_t)n_rec * 4 * sizeof(unsigned long long), stream));
        args.conf_stamps = (unsigned long long *)s->conf_stamps.p;
        s->conf_stamps_n = n_rec;
    }
    if (s->timing) {
        const size_t slot = s->n_timed % hsgpu_scratch::kRing;
        s->ev_t = s->ev_ring[slot];
        args.tstamp = (unsigned long long *)s->tstamp.p + 4 * slot;
        args.tstamp_next = (unsigned long long *)s->tstamp.p + 4 * ((slot + 1) % hsgpu_scratch::kRing);
        HIP_TRY(hipEventRecord(s->ev_t[0], stream));
    }
    if (!two_phase) {
        args.cand = nullptr;
        args.cand_cap = 0;
        args.cand_waves = 0;
        args.cand_counts = nullptr;
        if ((rv = set_dyn_lds(f_fused, lds)) != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipLaunchKernel(f_fused, dim3(grid), dim3(wg_threads), kargs, lds, stream));
        if (s->timing) {
            HIP_TRY(hipEventRecord(s->ev_t[1], stream));
        }
    } else {
        /* phase 1 + 2. Every filter wavefront owns a private region of the candidate buffer: one 32-byte
         * entry per 64 corpus bytes on average, i.e. room for candidates in a quarter of all chunks. */
        args.cand_waves = n_waves;
        uint64_t cand_div = s->cand_div;
        args.cand_cap = (uint32_t)std::max<uint64_t>(256, (a.total / cand_div + n_waves - 1) / n_waves);
        /* dense mode must not be able to overflow: sized from the kernel's own partition -- a wavefront owns
         * ceil(tiles / wavefronts) whole 1 KiB tiles of 64 chunks, and the one that takes the partial last tile 64 more
         * (total / 16 / n_waves rounds the other way: a fully dense share overflowed on every retry; advisor, round 3) */
        if (cand_div == 16) args.cand_cap = (uint32_t)std::max<uint64_t>(256, (((a.total >> 10) + n_waves - 1) / n_waves + 1) * 64);
        if ((rv = s->cand.ensure((uint64_t)args.cand_cap * n_waves * 32 + 64)) != HSGPU_SUCCESS) return rv; /* + slack: the confirm kernel reads one dword past an entry */
        args.cand = (uint4 *)s->cand.p;
        args.cand_counts = blk + cand_ofs;
        if ((rv = set_dyn_lds(f_two, lds_two)) != HSGPU_SUCCESS) return rv;
        if ((rv = set_dyn_lds(f_fused, lds)) != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipLaunchKernel(f_two, dim3(grid), dim3(wg_threads), kargs, lds_two, stream));
        if (s->timing) HIP_TRY(hipEventRecord(s->ev_t[1], stream));
        /* the confirm kernel's workers: at most as many as the device holds at once */
        HIP_TRY(hipLaunchKernel(f_conf, dim3(conf_grid), dim3(HSGPU_CONFIRM_THREADS), kargs, gate_lds, stream));
        /* No fused kernel behind it (it used to be launched on every scan, to return at once): a scan whose candidate regions
         * overflowed reports count = cap + 1 like one whose staging regions did -- "again" -- and sets the scratch's note,
         * so that the next scan has room for every chunk. Without the note (mapped host memory unavailable) the always-correct
         * fused kernel still redoes such a scan in place. */
        if (!s->d_note) HIP_TRY(hipLaunchKernel(f_fused, dim3(grid), dim3(wg_threads), kargs, lds, stream));
    }
    if (s->timing) s->n_timed++;
    /* one workgroup per group of regions: its records into place (folded: gathered, the regions are sorted runs; else
     * sorted); the control block back to zero */
    HIP_TRY(hipLaunchKernel(hsgpu_record_sort_kernel(), dim3((n_rec + args.group_regions - 1) / args.group_regions),
                            dim3(!fold && s->cand_div == 16 ? 1024 : 256), kargs, 0, stream));
    s->ctl_clean = true;
    s->ctl_parity ^= 1u;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_hwlm_scan_dev(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const void *d_corpus,
                                   uint64_t total_bytes, const void *d_off, uint64_t nblocks, uint64_t start,
                                   void *d_out, uint64_t cap, void *d_count, void *stream) {
    if (!t || !s || !d_off || !d_count || (cap && !d_out) || (total_bytes && !d_corpus)) return HSGPU_INVALID;
    if (((uintptr_t)d_corpus & 15) || ((uintptr_t)d_out & 15)) {
        hsgpu_set_error("corpus and record buffers must be 16-byte aligned");
        return HSGPU_INVALID;
    }
    if (cap >= (1ull << 32)) {
        hsgpu_set_error("record buffer larger than 2^32 records");
        return HSGPU_INVALID;
    }
    if (nblocks >= (1ull << 32)) { /* hsgpu_match_t.block is 32 bits */
        hsgpu_set_error("more than 2^32 - 1 blocks per launch");
        return HSGPU_INVALID;
    }
    if (total_bytes >= (1ull << 36)) { /* chunk index is 32 bits of 16-byte chunks */
        hsgpu_set_error("corpus larger than 64 GiB per launch");
        return HSGPU_INVALID;
    }
    HIP_TRY(hipSetDevice(s->device));
    if (nblocks == 0 || total_bytes == 0) {
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream ? (hipStream_t)stream : s->stream));
        return HSGPU_SUCCESS;
    }
    const uint8_t *d_blob = nullptr;
    int rv = table_on_device(t, s->device, &d_blob);
    if (rv != HSGPU_SUCCESS) return rv;
    HsgpuScanArgs a;
    a.corpus = (const uint8_t *)d_corpus;
    a.total = total_bytes;
    a.off = (const uint64_t *)d_off;
    a.nblocks = nblocks;
    a.start = start;
    a.blob = d_blob;
    a.out = (hsgpu_match_t *)d_out;
    a.cap = cap;
    a.count = (unsigned long long *)d_count;
    return launch_scan(t, s, a, stream ? (hipStream_t)stream : s->stream);
}

/* ---- host-buffer forms ------------------------------------------------------- */

struct RecLess { /* a functor, so that std::sort / std::merge inline the comparison */
    bool operator()(const hsgpu_match_t &a, const hsgpu_match_t &b) const {
        const uint64_t ka = (uint64_t)a.block << 32 | a.end, kb = (uint64_t)b.block << 32 | b.end;
        return ka != kb ? ka < kb : a.lit < b.lit;
    }
};
static const RecLess rec_less;

/* delivery order on the host. Large sets are cut into per-thread chunks, sorted concurrently and
 * merged pairwise (ping-pong through one scratch copy): the single-threaded std::sort of a few
 * hundred thousand records was the largest host cost of a host-buffer scan. */
extern "C" void hsgpu_match_sort_host(hsgpu_match_t *r, size_t n) {
    if (!r || n < 2) return;
    unsigned T = 1;
    if (n >= (1u << 16)) T = (unsigned)std::min<size_t>(std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency())), n >> 15);
    if (T <= 1) {
        std::sort(r, r + n, rec_less);
        return;
    }
    std::vector<size_t> b(T + 1);
    for (unsigned i = 0; i <= T; i++) b[i] = n * i / T;
    std::vector<hsgpu_match_t> tmp;
    try {
        tmp.resize(n);
    } catch (...) {
        std::sort(r, r + n, rec_less);
        return;
    }
    /* run f(i) for i in [0, jobs) on up to `jobs` threads; whatever cannot be started runs here */
    auto parallel = [](unsigned jobs, const std::function<void(unsigned)> &f) {
        std::vector<std::thread> pool;
        unsigned started = 1; /* job 0 is ours */
        try {
            for (; started < jobs; started++) pool.emplace_back(f, started);
        } catch (...) {
            for (unsigned i = started; i < jobs; i++) f(i);
        }
        f(0);
        for (std::thread &th : pool) th.join();
    };
    parallel(T, [&](unsigned i) { std::sort(r + b[i], r + b[i + 1], rec_less); });
    hsgpu_match_t *src = r, *dst = tmp.data();
    for (unsigned width = 1; width < T; width *= 2) {
        const unsigned jobs = (T + 2 * width - 1) / (2 * width);
        parallel(jobs, [&](unsigned j) {
            const unsigned lo = j * 2 * width, mid = std::min(T, lo + width), hi = std::min(T, lo + 2 * width);
            std::merge(src + b[lo], src + b[mid], src + b[mid], src + b[hi], dst + b[lo], rec_less);
        });
        std::swap(src, dst);
    }
    if (src != r) memcpy(r, src, n * sizeof(hsgpu_match_t));
}

struct InUse {
    hsgpu_scratch *s;
    bool ok;
    explicit InUse(hsgpu_scratch *s_) : s(s_), ok(!s_->in_use) {
        if (ok) s->in_use = true;
    }
    ~InUse() {
        if (ok) s->in_use = false;
    }
};

/* a host batch onto the device: corpus[off[0] .. off[nblocks]) and the offsets relative to off[0] into the scratch's
 * buffers, asynchronously on its stream */
static int upload_batch(hsgpu_scratch *s, const uint8_t *base, const uint64_t *off, size_t nblocks) {
    if ((uint64_t)nblocks >= (1ull << 32)) { /* hsgpu_match_t.block is 32 bits */
        hsgpu_set_error("more than 2^32 - 1 blocks per call");
        return HSGPU_INVALID;
    }
    const uint64_t lo = off[0], total = off[nblocks] - off[0];
    for (size_t i = 0; i < nblocks; i++) {
        if (off[i + 1] < off[i]) {
            hsgpu_set_error("block offsets must be ascending");
            return HSGPU_INVALID;
        }
        if (off[i + 1] - off[i] > 0xffffffffull) {
            hsgpu_set_error("block %zu longer than 4 GiB", i); /* hs_scan length is unsigned */
            return HSGPU_INVALID;
        }
    }
    HIP_TRY(hipSetDevice(s->device));
    int rv;
    if ((rv = s->corpus.ensure(total + 16)) != HSGPU_SUCCESS) return rv;
    if ((rv = s->off.ensure((nblocks + 1) * sizeof(uint64_t))) != HSGPU_SUCCESS) return rv;
    std::vector<uint64_t> rel(nblocks + 1);
    for (size_t i = 0; i <= nblocks; i++) rel[i] = off[i] - lo;
    if (total) HIP_TRY(hipMemcpyAsync(s->corpus.p, base + lo, total, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->off.p, rel.data(), rel.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream)); /* `rel` goes out of scope */
    s->res_corpus = s->corpus.p, s->res_off = s->off.p;
    return HSGPU_SUCCESS;
}

constexpr size_t SMALL_BYTES = 256 << 10, SMALL_BLOCKS = 4096, SMALL_RECS = 4096;
constexpr size_t SMALL_OFF_AT = 64, SMALL_CORPUS_AT = SMALL_OFF_AT + (SMALL_BLOCKS + 1) * 8 + 56 /* -> a multiple of 64 */,
                 SMALL_RECS_AT = SMALL_CORPUS_AT + SMALL_BYTES + 64, SMALL_TOTAL = SMALL_RECS_AT + SMALL_RECS * sizeof(hsgpu_match_t);
static_assert(SMALL_CORPUS_AT % 64 == 0 && SMALL_RECS_AT % 64 == 0, "aligned sections");
/* the request side in device memory (bar_area): offsets and corpus where the mapped area has them, the request lines of the mailbox behind them */
constexpr size_t BAR_CTL_AT = SMALL_RECS_AT, BAR_TOTAL = BAR_CTL_AT + sizeof(HsgpuServerCtl);

/* The request side of the server's mailbox in device memory (hsgpu_scratch::bar_small), once per scratch. */
static bool bar_area(hsgpu_scratch *s) {
    if (s->srv_host_mailbox || hsgpu_dev_guard_mode()) return false; /* (guard-page runs keep to the buffers the guard allocator made) */
    if (s->bar_state) return s->bar_state > 0;
    s->bar_state = -1;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, s->device) != hipSuccess || !pr.isLargeBar) {
        (void)hipGetLastError();
        return false;
    }
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, BAR_TOTAL, hipDeviceMallocFinegrained) != hipSuccess || hipMemsetAsync(p, 0, BAR_TOTAL, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        return false;
    }
    /* isLargeBar says the host can reach device memory; that THIS range is writable from here is asked of the kernel, without a
     * signal handler: read(2) from /dev/zero into it fails with EFAULT where a store would fault */
    bool writable = false;
    const int zfd = open("/dev/zero", O_RDONLY | O_CLOEXEC);
    if (zfd >= 0) {
        writable = read(zfd, p, 64) == 64 && read(zfd, (uint8_t *)p + BAR_TOTAL - 64, 64) == 64;
        (void)close(zfd);
    }
    if (!writable) {
        (void)hipFree(p);
        return false;
    }
    s->bar_small = (uint8_t *)p;
    s->bar_ctl = (HsgpuServerCtl *)(s->bar_small + BAR_CTL_AT);
    s->bar_state = 1;
    return true;
}

/* one resident workgroup for this (scratch, table): its arguments are a solo scan's, with corpus, offsets, records and count
 * in the scratch's mapped area. -> HSGPU_SUCCESS, 1: no server for this table / device (the caller launches), < 0: an error */
static int server_start(const hsgpu_hwlm *t, hsgpu_scratch *s) {
    const HsgpuTableHeader *h = t->hdr();
    const void *f = hsgpu_server_kernel_for(h->flags);
    if (!f || !s->d_small || !s->d_note) return 1;
    const unsigned wg_threads = filter_wg_threads(h, s, nullptr);
    const size_t lds = hsgpu_filter_lds_bytes(h->flags, h->filter_log2, true, wg_threads) + 192; /* + the mailbox, the answer line and the stage stamps */
    if (lds > s->lds_per_cu || (size_t)hsgpu_filter_words(h->flags, h->filter_log2) * 4 < 28 * 1024) return 1;
    if (!s->h_srv) {
        if (hipHostMalloc((void **)&s->h_srv, sizeof(HsgpuServerCtl), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&s->d_srv, s->h_srv, 0) != hipSuccess ||
            hipStreamCreateWithFlags(&s->srv_stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            if (s->h_srv) (void)hipHostFree(s->h_srv);
            s->h_srv = s->d_srv = nullptr;
            return 1;
        }
        memset(s->h_srv, 0, sizeof(HsgpuServerCtl));
    }
    const uint8_t *d_blob = nullptr;
    int rv = table_on_device(t, s->device, &d_blob);
    if (rv != HSGPU_SUCCESS) return rv;
    HsgpuScanArgs args;
    memset(&args, 0, sizeof(args));
    /* the batch is copied from the mapped area to device memory by the workgroup itself at the head of every request */
    const uint32_t super_shift = wg_threads <= 256 ? 12 : wg_threads <= 512 ? 13 : 14;
    if ((rv = s->corpus.ensure((1ull << super_shift) + 64)) != HSGPU_SUCCESS) return rv;
    if ((rv = s->off.ensure((SMALL_BLOCKS + 1) * sizeof(uint64_t) + 64)) != HSGPU_SUCCESS) return rv;
    args.corpus = (const uint8_t *)s->corpus.p;
    args.off = (const uint64_t *)s->off.p;
    args.blob = d_blob;
    args.out = (hsgpu_match_t *)(s->d_small + SMALL_RECS_AT);
    args.cap = SMALL_RECS;
    args.count = (unsigned long long *)s->d_small;
    table_args(h, args);
    if ((rv = solo_setup(s, args, 1ull << super_shift, SMALL_RECS, 1, wg_threads, s->srv_stream)) != HSGPU_SUCCESS) return rv;
    if ((rv = set_dyn_lds(f, lds)) != HSGPU_SUCCESS) return rv;
    s->srv_req_bar = bar_area(s);
    if (s->srv_req_bar) { /* (a request written for a server that went idle before it saw it is served by this one: same words) */
        HsgpuServerCtl *q = (HsgpuServerCtl *)(s->bar_small + BAR_CTL_AT);
        q->req_seq = s->h_srv->req_seq;
        q->total = s->h_srv->total, q->nblocks = s->h_srv->nblocks, q->start = s->h_srv->start, q->debug = s->h_srv->debug;
    }
    server_req(s)->stop = 0;
    bar_flush();
    s->h_srv->exited = 0;
    HsgpuServerCtl *ctl = s->d_srv, *req = s->srv_req_bar ? (HsgpuServerCtl *)(s->bar_small + BAR_CTL_AT) : s->d_srv;
    unsigned long long idle_ticks = (unsigned long long)s->srv_idle_us * 100ull; /* the 100 MHz wall clock */
    uint8_t *const area = s->srv_req_bar ? s->bar_small : s->d_small;
    const void *src_corpus = area + SMALL_CORPUS_AT, *src_off = area + SMALL_OFF_AT;
    void *kargs[] = {&args, &ctl, &req, &idle_ticks, &src_corpus, &src_off};
    HIP_TRY(hipLaunchKernel(f, dim3(1), dim3(wg_threads), kargs, lds, s->srv_stream));
    s->srv_live = true;
    s->srv_table = t;
    s->srv_launches++;
    return HSGPU_SUCCESS;
}

/* a batch the resident workgroup could serve on this scratch as it is set up */
static bool server_takes(const hsgpu_hwlm *t, hsgpu_scratch *s, uint64_t total) {
    if (!s->srv_enabled || !total) return false;
    const HsgpuTableHeader *h = t->hdr();
    const unsigned wg_threads = filter_wg_threads(h, s, nullptr);
    const uint32_t super_shift = wg_threads <= 256 ? 12 : wg_threads <= 512 ? 13 : 14;
    /* what ONE workgroup scans as a solo scan, under the conditions a solo scan has (launch_scan) */
    return !(total > (1ull << super_shift) || s->tune_fused || s->tune_unfolded || s->tune_solo == 1 || s->cand_div != 64 || (s->h_note && *s->h_note));
}

/* A server for this table and a batch of this size, resident (started when there is none): -> 0, where it reads its requests is
 * s->srv_req_bar; 1: none to be had: the caller launches; < 0: an error. BEFORE the batch is laid out: it goes where the server reads. */
static int server_ready(const hsgpu_hwlm *t, hsgpu_scratch *s, uint64_t total) {
    if (!server_takes(t, s, total)) return 1;
    if (s->srv_live && s->srv_table != t) server_stop(s);
    if (s->srv_live && __atomic_load_n(&s->h_srv->exited, __ATOMIC_ACQUIRE)) { /* it went idle and ended */
        (void)hipStreamSynchronize(s->srv_stream);
        s->srv_live = false;
    }
    return s->srv_live ? 0 : server_start(t, s);
}

/* One request to the resident server (server_ready said 0; the batch is laid out where it reads). -> 0: served (count and records
 * are in the mapped area), < 0: an error */
static int server_call(const hsgpu_hwlm *t, hsgpu_scratch *s, uint64_t total, size_t nblocks, size_t start) {
    int rv;
    HsgpuServerCtl *c = s->h_srv, *q = server_req(s);
    const uint32_t seq = ++s->srv_seq;
    /* the answer line poisoned: should its 64 bytes ever arrive in pieces (they are one store, one bus write), a piece that is not
     * there yet shows -- no literal has index ~0, no count is ~0 - 1 -- and scan_host_small waits for it */
    c->done_count = SRV_POISON_COUNT;
    for (int i = 0; i < HSGPU_SRV_INLINE_RECS; i++) c->done_rec[4 * i + 3] = ~0u;
    if (q != c) c->total = total, c->nblocks = nblocks, c->start = start, c->debug = s->srv_debug, c->req_seq = seq; /* (what a restarted server is told: server_start) */
    q->total = total, q->nblocks = nblocks, q->start = start, q->debug = s->srv_debug;
    bar_flush(); /* the batch (scan_host_small) and the parameters are out before the sequence number */
    __atomic_store_n(&q->req_seq, seq, __ATOMIC_RELEASE);
    bar_flush();
    s->srv_calls++;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
        if (__atomic_load_n(&c->done_seq, __ATOMIC_ACQUIRE) == seq) return 0;
        if (__atomic_load_n(&c->exited, __ATOMIC_ACQUIRE)) {
            /* it ended (idle, as this request was being written) without having seen it: the next one starts from done_seq */
            (void)hipStreamSynchronize(s->srv_stream);
            s->srv_live = false;
            if (__atomic_load_n(&c->done_seq, __ATOMIC_ACQUIRE) == seq) return 0;
            if ((rv = server_start(t, s)) != HSGPU_SUCCESS) return rv;
        }
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            hsgpu_set_error("the small-batch server did not answer request %u (done %u, exited %u)", seq, c->done_seq, c->exited);
            server_stop(s);
            return HSGPU_UNKNOWN_ERROR;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

/* Small host batches (one packet per hwlmExec call is the reference's normal diet: src/rose/block.c:382-391,
 * tools/hsbench/engine_hyperscan.cpp:132-145): no copy commands at all. The batch is laid out in mapped pinned memory owned by
 * the scratch -- {count | offsets | corpus | records} -- and the scan kernel reads it over the bus and writes records and count
 * back into it: ONE kernel launch (a solo scan, launch_scan) and one stream synchronisation per call. Round 4: an H2D copy, a
 * synchronisation, a memset, three launches, a D2H of the count, a synchronisation and a blocking D2H of the records --
 * 68-72 us per 1 460-byte call. -> HSGPU_SUCCESS, or 1 when the batch does not fit this path (the caller takes the general one). */
static int scan_host_small(const hsgpu_hwlm *t, hsgpu_scratch *s, const uint8_t *base, const uint64_t *off, size_t nblocks,
                           size_t start, std::vector<hsgpu_match_t> &recs) {
    const uint64_t lo = off[0], total = off[nblocks] - off[0];
    if (total > SMALL_BYTES || nblocks > SMALL_BLOCKS) return 1;
    for (size_t i = 0; i < nblocks; i++)
        if (off[i + 1] < off[i]) return 1; /* (the general path reports it) */
    HIP_TRY(hipSetDevice(s->device));
    if (!s->h_small) {
        if (hipHostMalloc((void **)&s->h_small, SMALL_TOTAL, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&s->d_small, s->h_small, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (s->h_small) (void)hipHostFree(s->h_small);
            s->h_small = s->d_small = nullptr;
            return 1;
        }
    }
    /* the batch into the area the scan reads it from: the mapped one -- or, for the resident workgroup on a device whose memory the
     * host can write (bar_area), device memory */
    auto put = [&](uint8_t *area) {
        uint64_t *a_off = (uint64_t *)(area + SMALL_OFF_AT);
        for (size_t i = 0; i <= nblocks; i++) a_off[i] = off[i] - lo;
        memcpy(area + SMALL_CORPUS_AT, base + lo, total);
        memset(area + SMALL_CORPUS_AT + total, 0, 16);
    };
    int rv = server_ready(t, s, total); /* the resident workgroup, when the scratch has one enabled and the batch is its size */
    if (rv < 0) return rv;
    const bool served = rv == 0, to_bar = served && s->srv_req_bar;
    put(to_bar ? s->bar_small : s->h_small);
    *(volatile unsigned long long *)s->h_small = ~0ull;
    uint8_t *const d_area = to_bar ? s->bar_small : s->d_small;
    s->res_corpus = d_area + SMALL_CORPUS_AT, s->res_off = d_area + SMALL_OFF_AT;
    if (served && (rv = server_call(t, s, total, nblocks, start)) < 0) return rv;
    if (served) {
        /* the answer line (HsgpuServerCtl): up to three records came with the sequence number; ~0: count and records are in the mapped area */
        uint32_t n_line;
        for (unsigned spin = 0;; spin++) { /* (see server_call: whole on the first look unless the line came in pieces) */
            n_line = __atomic_load_n(&s->h_srv->done_count, __ATOMIC_ACQUIRE);
            bool whole = n_line != SRV_POISON_COUNT;
            for (uint32_t i = 0; whole && n_line <= HSGPU_SRV_INLINE_RECS && i < n_line; i++) /* (~0: count and records are in the mapped area) */
                whole = __atomic_load_n(&s->h_srv->done_rec[4 * i + 3], __ATOMIC_ACQUIRE) != ~0u;
            if (whole) break;
            if (spin > 100000000u) {
                hsgpu_set_error("the small-batch server's answer line stayed incomplete");
                server_stop(s);
                return HSGPU_UNKNOWN_ERROR;
            }
        }
        if (n_line <= HSGPU_SRV_INLINE_RECS) {
            recs.resize(n_line);
            if (n_line) memcpy(recs.data(), (const void *)s->h_srv->done_rec, n_line * sizeof(hsgpu_match_t));
            return HSGPU_SUCCESS;
        }
        const uint64_t n = *(volatile unsigned long long *)s->h_small;
        if (n <= SMALL_RECS) {
            recs.resize(n);
            if (n) memcpy(recs.data(), s->h_small + SMALL_RECS_AT, n * sizeof(hsgpu_match_t));
            return HSGPU_SUCCESS;
        }
        *(volatile unsigned long long *)s->h_small = ~0ull; /* more records than the area holds, or "again": the launch path decides */
    }
    if (to_bar) { /* (rare: the launch path reads the mapped area) */
        put(s->h_small);
        s->res_corpus = s->d_small + SMALL_CORPUS_AT, s->res_off = s->d_small + SMALL_OFF_AT;
    }
    rv = hsgpu_hwlm_scan_dev(t, s, s->d_small + SMALL_CORPUS_AT, total, s->d_small + SMALL_OFF_AT, nblocks, start,
                                 s->d_small + SMALL_RECS_AT, SMALL_RECS, s->d_small, s->stream);
    if (rv != HSGPU_SUCCESS) return rv;
    HIP_TRY(hipStreamSynchronize(s->stream));
    const uint64_t n = *(volatile unsigned long long *)s->h_small;
    if (n > SMALL_RECS) return 1; /* more matches than this path's buffer holds (or "again"): the general path */
    recs.resize(n);
    if (n) memcpy(recs.data(), s->h_small + SMALL_RECS_AT, n * sizeof(hsgpu_match_t));
    return HSGPU_SUCCESS;
}

/* scan host blocks; on return recs holds ALL matches sorted by (block,end,lit) */
static int scan_host(const hsgpu_hwlm *t, hsgpu_scratch *s, const uint8_t *base, const uint64_t *off,
                     size_t nblocks, size_t start, std::vector<hsgpu_match_t> &recs) {
    recs.clear();
    const uint64_t total = off[nblocks] - off[0];
    if (total == 0) return HSGPU_SUCCESS;
    int rv = scan_host_small(t, s, base, off, nblocks, start, recs);
    if (rv != 1) return rv;
    recs.clear();
    rv = upload_batch(s, base, off, nblocks);
    if (rv != HSGPU_SUCCESS) return rv;
    uint64_t cap = std::max<uint64_t>(4096, total / 256);
    for (int attempt = 0; attempt < 8; attempt++) {
        if ((rv = s->out.ensure(cap * sizeof(hsgpu_match_t))) != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipMemsetAsync(s->count.p, 0, sizeof(unsigned long long), s->stream));
        rv = hsgpu_hwlm_scan_dev(t, s, s->corpus.p, total, s->off.p, nblocks, start, s->out.p, cap, s->count.p,
                                 s->stream);
        if (rv != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipMemcpyAsync(s->h_count, s->count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        uint64_t n = *s->h_count;
        if (n <= cap) {
            recs.resize(n); /* already in delivery order: (block, end, literal index) */
            if (n) HIP_TRY(hipMemcpy(recs.data(), s->out.p, n * sizeof(hsgpu_match_t), hipMemcpyDeviceToHost));
            return HSGPU_SUCCESS;
        }
        /* overflow: the count is exact; rerun with room for all of them, and since the
         * records are staged in per-wavefront regions sized from cap, leave headroom
         * for skew (doubling per attempt) */
        cap = std::max<uint64_t>(n + n / 4, cap * 2);
    }
    hsgpu_set_error("match buffer overflow persisted");
    return HSGPU_UNKNOWN_ERROR;
}

/* Class-sequence patterns over a host batch (or over the batch this scratch scanned last, still resident):
 * class bitmaps in passes of <= 8, then the sequence kernel over all of them; records of the whole batch in
 * delivery order, counts per pattern. */
extern "C" int hsgpu_class_seq_exec_batch(const hsgpu_class_t *classes, unsigned n_classes, const hsgpu_class_seq_t *seqs,
                                          unsigned n_seqs, hsgpu_scratch_t *s, const uint8_t *base, const uint64_t *off,
                                          size_t nblocks, int reuse_resident, uint64_t *counts, hsgpu_match_t *out,
                                          size_t cap, size_t *nout) {
    if (!classes || !n_classes || !seqs || !n_seqs || !s || !off || !nout || (cap && !out)) return HSGPU_INVALID;
    *nout = 0;
    if (counts) memset(counts, 0, (size_t)n_seqs * sizeof(uint64_t));
    if (nblocks == 0) return HSGPU_SUCCESS;
    const uint64_t total = off[nblocks] - off[0];
    if (total == 0) return HSGPU_SUCCESS;
    if (!base && !reuse_resident) return HSGPU_INVALID;
    InUse guard(s);
    if (!guard.ok) return HSGPU_SCRATCH_IN_USE;
    int rv;
    if (!reuse_resident && (rv = upload_batch(s, base, off, nblocks)) != HSGPU_SUCCESS) return rv;
    HIP_TRY(hipSetDevice(s->device));
    const size_t row = ((total + 15) / 16 * 2 + 15) & ~(size_t)15;
    const size_t seq_work = hsgpu_class_seq_work_bytes(total);
    if ((rv = s->cs_bitmaps.ensure(row * n_classes)) != HSGPU_SUCCESS) return rv;
    if ((rv = s->cs_work.ensure(HSGPU_CLASS_WORK_BYTES + 64 + seq_work)) != HSGPU_SUCCESS) return rv;
    if ((rv = s->cs_counts.ensure((size_t)n_seqs * sizeof(uint64_t))) != HSGPU_SUCCESS) return rv;
    std::vector<void *> ptrs(n_classes);
    for (unsigned c = 0; c < n_classes; c++) ptrs[c] = (uint8_t *)s->cs_bitmaps.p + row * c;
    for (unsigned c = 0; c < n_classes; c += HSGPU_CLASS_MAX_BITMAPS) { /* bitmaps alone: 16 classes per read of the corpus */
        const unsigned k = std::min<unsigned>(HSGPU_CLASS_MAX_BITMAPS, n_classes - c);
        rv = hsgpu_class_scan_dev(classes + c, k, s->res_corpus, total, s->res_off, nblocks, ptrs.data() + c, nullptr, nullptr,
                                  s->cs_work.p, s->stream);
        if (rv != HSGPU_SUCCESS) return rv;
    }
    void *work2 = (uint8_t *)s->cs_work.p + ((HSGPU_CLASS_WORK_BYTES + 63) & ~(size_t)63);
    uint64_t dcap = std::max<uint64_t>(cap, 4096);
    for (int attempt = 0; attempt < 3; attempt++) {
        if ((rv = s->out.ensure(dcap * sizeof(hsgpu_match_t))) != HSGPU_SUCCESS) return rv;
        rv = hsgpu_class_seq_scan_dev(seqs, n_seqs, ptrs.data(), n_classes, total, s->res_off, nblocks, 0, total,
                                      s->cs_counts.p, s->out.p, dcap, s->count.p, work2, seq_work, s->stream);
        if (rv != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipMemcpyAsync(s->h_count, s->count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        const uint64_t n = *s->h_count;
        *nout = (size_t)n;
        if (counts) HIP_TRY(hipMemcpy(counts, s->cs_counts.p, (size_t)n_seqs * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (n > cap) return HSGPU_INSUFFICIENT_SPACE; /* *nout = the room the caller needs */
        if (n <= dcap) {
            if (n) HIP_TRY(hipMemcpy(out, s->out.p, n * sizeof(hsgpu_match_t), hipMemcpyDeviceToHost));
            hsgpu_match_sort_host(out, n); /* the kernel emits in no particular order */
            return HSGPU_SUCCESS;
        }
        dcap = n;
    }
    return HSGPU_UNKNOWN_ERROR;
}

/* The same for the blocks [block_lo, block_hi) only (include/hsgpu.h): a batch whose patterns match several times per byte holds
 * more records than any buffer -- 256 class-heavy patterns over 1 GiB are 16 bytes x 800 M -- and the reference delivers such
 * matches through callbacks in O(1) memory; the facade walks the batch range by range through this (advisor, round 3). */
extern "C" int hsgpu_class_seq_exec_blocks(const hsgpu_class_t *classes, unsigned n_classes, const hsgpu_class_seq_t *seqs,
                                           unsigned n_seqs, hsgpu_scratch_t *s, const uint8_t *base, const uint64_t *off, size_t nblocks,
                                           int reuse_resident, int reuse_bitmaps, size_t block_lo, size_t block_hi, hsgpu_match_t *out,
                                           size_t cap, size_t *nout) {
    if (!classes || !n_classes || !seqs || !n_seqs || !s || !off || !nout || (cap && !out) || block_lo > block_hi || block_hi > nblocks)
        return HSGPU_INVALID;
    *nout = 0;
    if (block_lo == block_hi) return HSGPU_SUCCESS;
    const uint64_t total = off[nblocks] - off[0];
    if (total == 0) return HSGPU_SUCCESS;
    /* a range of empty blocks with batch and bitmaps already in place: nothing to emit (an empty emit range would select the
     * counting kernel and walk the corpus for nothing; advisor, round 4). A first call still uploads and classifies: its caller
     * goes on with reuse_resident / reuse_bitmaps. */
    if (off[block_lo] == off[block_hi] && reuse_resident && reuse_bitmaps) return HSGPU_SUCCESS;
    if (!base && !reuse_resident) return HSGPU_INVALID;
    InUse guard(s);
    if (!guard.ok) return HSGPU_SCRATCH_IN_USE;
    int rv;
    if (!reuse_resident && (rv = upload_batch(s, base, off, nblocks)) != HSGPU_SUCCESS) return rv;
    HIP_TRY(hipSetDevice(s->device));
    const size_t row = ((total + 15) / 16 * 2 + 15) & ~(size_t)15;
    const size_t seq_work = hsgpu_class_seq_work_bytes(total);
    if (reuse_bitmaps && (s->cs_bitmaps.cap < row * n_classes || s->cs_work.cap < HSGPU_CLASS_WORK_BYTES + 64 + seq_work)) return HSGPU_INVALID;
    if ((rv = s->cs_bitmaps.ensure(row * n_classes)) != HSGPU_SUCCESS) return rv;
    if ((rv = s->cs_work.ensure(HSGPU_CLASS_WORK_BYTES + 64 + seq_work)) != HSGPU_SUCCESS) return rv;
    std::vector<void *> ptrs(n_classes);
    for (unsigned c = 0; c < n_classes; c++) ptrs[c] = (uint8_t *)s->cs_bitmaps.p + row * c;
    for (unsigned c = 0; !reuse_bitmaps && c < n_classes; c += HSGPU_CLASS_MAX_BITMAPS) {
        const unsigned k = std::min<unsigned>(HSGPU_CLASS_MAX_BITMAPS, n_classes - c);
        rv = hsgpu_class_scan_dev(classes + c, k, s->res_corpus, total, s->res_off, nblocks, ptrs.data() + c, nullptr, nullptr,
                                  s->cs_work.p, s->stream);
        if (rv != HSGPU_SUCCESS) return rv;
    }
    void *work2 = (uint8_t *)s->cs_work.p + ((HSGPU_CLASS_WORK_BYTES + 63) & ~(size_t)63);
    if ((rv = s->out.ensure(std::max<size_t>(cap, 4096) * sizeof(hsgpu_match_t))) != HSGPU_SUCCESS) return rv;
    rv = hsgpu_class_seq_emit_dev(seqs, n_seqs, ptrs.data(), n_classes, total, s->res_off, nblocks, off[block_lo] - off[0], off[block_hi] - off[0],
                                  s->out.p, cap, s->count.p, work2, seq_work, s->stream);
    if (rv != HSGPU_SUCCESS) return rv;
    HIP_TRY(hipMemcpyAsync(s->h_count, s->count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const uint64_t n = *s->h_count;
    *nout = (size_t)n;
    if (n > cap) return HSGPU_INSUFFICIENT_SPACE;
    if (n) HIP_TRY(hipMemcpy(out, s->out.p, n * sizeof(hsgpu_match_t), hipMemcpyDeviceToHost));
    hsgpu_match_sort_host(out, n); /* the kernel emits in no particular order */
    return HSGPU_SUCCESS;
}

/* ---- the chunked host-buffer pipeline ---------------------------------------------------------
 * A host batch cut into chunks of whole blocks (64 MiB by default). A producer thread drives the device: the copy
 * of chunk i + 1 (side stream) runs beside the scan of chunk i (scratch stream), two device slots alternate; the
 * records of every finished chunk, block indices made global, are handed IN BLOCK ORDER to the calling thread,
 * which runs the caller's function on them while later chunks copy and scan. The reference has no device boundary
 * (doc/dev-reference/performance.rst:56-61); for a caller whose data is in host memory this is what keeps the
 * bus busy end to end: everything on the host side of a chunk (sorting, confirm, callbacks) hides behind the copies
 * of the chunks after it. */
#include <deque>
namespace {
struct ChunkResult {
    std::vector<hsgpu_match_t> recs;
    int rv = HSGPU_SUCCESS;
    bool last = false;
};
} // namespace

static int produce_chunks(const hsgpu_hwlm *t, hsgpu_scratch *s, const uint8_t *base, const uint64_t *off, size_t nblocks,
                          size_t start, const std::vector<size_t> &cuts, std::mutex &mu, std::condition_variable &cv,
                          std::deque<ChunkResult> &queue, std::atomic<bool> &abort, std::atomic<bool> &producer_dead) {
    auto fail = [&](int rv) {
        /* a copy from the caller's buffer may still be on its way: the caller is free to release that buffer once the call
         * has returned, so nothing of it may be in flight by then (advisor, round 3) */
        (void)hipStreamSynchronize(s->side);
        (void)hipStreamSynchronize(s->stream);
        try {
            ChunkResult r;
            r.rv = rv;
            r.last = true;
            {
                std::lock_guard<std::mutex> g(mu);
                queue.push_back(std::move(r));
            }
        } catch (...) { /* not even the end marker could be queued: the consumer watches this flag of its own (advisor, round 4:
                         * `abort` also means "the caller asked to stop", and a consumer that had asked waited for ever) */
            std::lock_guard<std::mutex> g(mu);
            abort = true;
            producer_dead = true;
        }
        cv.notify_all();
        return rv;
    };
    try {
    if (hipSetDevice(s->device) != hipSuccess) return fail(HSGPU_UNKNOWN_ERROR);
    const size_t n_chunks = cuts.size() - 1;
    /* page-locked memory that grows on demand (freed with the scratch) */
    auto pinned = [&](void **p, size_t *cap, size_t bytes) -> int {
        if (*cap >= bytes) return HSGPU_SUCCESS;
        if (*p) (void)hipHostFree(*p);
        *p = nullptr, *cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(p, want) != hipSuccess) {
            (void)hipGetLastError();
            *p = nullptr;
            return HSGPU_NOMEM;
        }
        *cap = want;
        return HSGPU_SUCCESS;
    };
    auto copy_in = [&](size_t i) -> int {
        const int slot = (int)(i & 1);
        const size_t b0 = cuts[i], b1 = cuts[i + 1];
        const uint64_t lo = off[b0], bytes = off[b1] - lo;
        int rv;
        /* (the slot's last copy out of this staging area belonged to chunk i - 2, whose scan this thread has waited for) */
        if ((rv = pinned((void **)&s->h_rel[slot], &s->h_rel_cap[slot], (b1 - b0 + 1) * sizeof(uint64_t))) != HSGPU_SUCCESS) return rv;
        uint64_t *rel = s->h_rel[slot];
        uint64_t bad = 0; /* (branch-free: the loop vectorises) */
        for (size_t k = 0; k < b1 - b0; k++) {
            const uint64_t a = off[b0 + k], b = off[b0 + k + 1];
            rel[k] = a - lo;
            bad |= (uint64_t)(b < a) | ((b - a) >> 32);
        }
        rel[b1 - b0] = off[b1] - lo;
        if (bad) {
            hsgpu_set_error("block offsets must be ascending and blocks shorter than 4 GiB (blocks %zu .. %zu)", b0, b1);
            return HSGPU_INVALID;
        }
        if ((rv = s->pipe_corpus[slot].ensure(bytes + 16)) != HSGPU_SUCCESS) return rv;
        if ((rv = s->pipe_off[slot].ensure((b1 - b0 + 1) * sizeof(uint64_t))) != HSGPU_SUCCESS) return rv;
        /* the small copy FIRST: behind the corpus it would wait for it, and the scan for both */
        HIP_TRY(hipMemcpyAsync(s->pipe_off[slot].p, rel, (b1 - b0 + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s->side));
        if (bytes) HIP_TRY(hipMemcpyAsync(s->pipe_corpus[slot].p, base + lo, bytes, hipMemcpyHostToDevice, s->side));
        HIP_TRY(hipEventRecord(s->ev_copied[slot], s->side));
        return HSGPU_SUCCESS;
    };
    int rv;
    for (int i = 0; i < 2; i++) {
        if (!s->ev_copied[i] && hipEventCreateWithFlags(&s->ev_copied[i], hipEventDisableTiming) != hipSuccess) return fail(HSGPU_UNKNOWN_ERROR);
        if (!s->ev_scanned[i] && hipEventCreateWithFlags(&s->ev_scanned[i], hipEventDisableTiming) != hipSuccess) return fail(HSGPU_UNKNOWN_ERROR);
    }
    if ((rv = s->pipe_count.ensure(2 * sizeof(unsigned long long))) != HSGPU_SUCCESS) return fail(rv);
    if ((rv = copy_in(0)) != HSGPU_SUCCESS) return fail(rv);
    for (size_t i = 0; i < n_chunks && !abort; i++) {
        const int slot = (int)(i & 1);
        const size_t b0 = cuts[i], b1 = cuts[i + 1];
        const uint64_t bytes = off[b1] - off[b0];
        uint64_t cap = std::max<uint64_t>(4096, std::max<uint64_t>(bytes / 256, s->pipe_out[slot].cap / sizeof(hsgpu_match_t)));
        uint64_t n = 0;
        bool next_issued = false;
        for (int attempt = 0;; attempt++) {
            if ((rv = s->pipe_out[slot].ensure(cap * sizeof(hsgpu_match_t))) != HSGPU_SUCCESS) return fail(rv);
            unsigned long long *d_count = (unsigned long long *)s->pipe_count.p + slot;
            if (hipStreamWaitEvent(s->stream, s->ev_copied[slot], 0) != hipSuccess) return fail(HSGPU_UNKNOWN_ERROR);
            if (bytes) {
                rv = hsgpu_hwlm_scan_dev(t, s, s->pipe_corpus[slot].p, bytes, s->pipe_off[slot].p, b1 - b0, start,
                                         s->pipe_out[slot].p, cap, d_count, s->stream);
                if (rv != HSGPU_SUCCESS) return fail(rv);
            } else if (hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s->stream) != hipSuccess) {
                return fail(HSGPU_UNKNOWN_ERROR);
            }
            if (hipMemcpyAsync(s->h_count + slot, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                hipEventRecord(s->ev_scanned[slot], s->stream) != hipSuccess)
                return fail(HSGPU_UNKNOWN_ERROR);
            /* the next chunk's copy goes out now, beside this chunk's scan; its slot was last read by the scan of
             * chunk i - 1, which this thread has already waited for */
            if (!next_issued && i + 1 < n_chunks) {
                if ((rv = copy_in(i + 1)) != HSGPU_SUCCESS) return fail(rv);
                next_issued = true;
            }
            if (hipEventSynchronize(s->ev_scanned[slot]) != hipSuccess) return fail(HSGPU_UNKNOWN_ERROR);
            n = s->h_count[slot];
            if (n <= cap) break;
            if (attempt == 7) return fail(HSGPU_UNKNOWN_ERROR);
            cap = std::max<uint64_t>(n + n / 4, cap * 2); /* the count is exact: again with room (and headroom for skew) */
        }
        ChunkResult r;
        if (n) { /* the records: to page-locked memory on the scan's own stream (no default-stream copy, nothing staged), then into the chunk's vector */
            if ((rv = pinned((void **)&s->h_prec[slot], &s->h_prec_cap[slot], n * sizeof(hsgpu_match_t))) != HSGPU_SUCCESS) return fail(rv);
            if (hipMemcpyAsync(s->h_prec[slot], s->pipe_out[slot].p, n * sizeof(hsgpu_match_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                hipStreamSynchronize(s->stream) != hipSuccess)
                return fail(HSGPU_UNKNOWN_ERROR);
            try {
                r.recs.assign(s->h_prec[slot], s->h_prec[slot] + n);
            } catch (...) {
                return fail(HSGPU_NOMEM);
            }
        }
        for (hsgpu_match_t &m : r.recs) m.block += (uint32_t)b0;
        r.last = i + 1 == n_chunks;
        {
            std::lock_guard<std::mutex> g(mu);
            queue.push_back(std::move(r));
        }
        cv.notify_all();
    }
    if (abort) { /* the consumer stopped early: let it see an end */
        return fail(HSGPU_SUCCESS);
    }
    return HSGPU_SUCCESS;
    } catch (...) { /* std::bad_alloc from a vector on this thread: no exception may leave it (std::terminate) */
        return fail(HSGPU_NOMEM);
    }
}

/* off[0] <= off[1] <= ... <= off[nblocks], every block shorter than 4 GiB: branch-free (it vectorises), on up to eight threads
 * above a million blocks */
static bool offsets_ascending(const uint64_t *off, size_t nblocks) {
    auto walk = [off](size_t lo, size_t hi) -> uint64_t {
        uint64_t bad = 0;
        for (size_t k = lo; k < hi; k++) {
            const uint64_t a = off[k], b = off[k + 1];
            bad |= (uint64_t)(b < a) | ((b - a) >> 32);
        }
        return bad;
    };
    if (nblocks < (1u << 20)) return walk(0, nblocks) == 0;
    const unsigned T = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<uint64_t> bad(T, 0);
    std::vector<std::thread> th;
    try {
        for (unsigned i = 1; i < T; i++) th.emplace_back([&, i] { bad[i] = walk(nblocks * i / T, nblocks * (i + 1) / T); });
    } catch (...) { /* no more threads: whatever was not started is walked here */
    }
    const size_t started = th.size() + 1;
    bad[0] = walk(0, nblocks / T);
    uint64_t any = 0;
    for (size_t i = started; i < T; i++) any |= walk(nblocks * i / T, nblocks * (i + 1) / T);
    for (std::thread &t : th) t.join();
    for (unsigned i = 0; i < T; i++) any |= bad[i];
    return any == 0;
}

/* is this host memory page-locked (hipHostMalloc / hipHostRegister)? Asynchronous copies from pageable memory are
 * staged by the runtime chunk by chunk and gain nothing from being cut up further. */
int hsgpu_host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return a.type == hipMemoryTypeHost ? 1 : 0;
}

extern "C" int hsgpu_hwlm_exec_batch_cb(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const uint8_t *base, const uint64_t *off,
                                        size_t nblocks, size_t start, size_t chunk_bytes, hsgpu_chunk_cb on_chunk, void *ctx) {
    if (!t || !s || !off || !on_chunk) return HSGPU_INVALID;
    if (nblocks == 0) return HSGPU_SUCCESS;
    if (!base && off[nblocks] != off[0]) return HSGPU_INVALID;
    if ((uint64_t)nblocks >= (1ull << 32)) {
        hsgpu_set_error("more than 2^32 - 1 blocks per call");
        return HSGPU_INVALID;
    }
    /* Every offset is checked BEFORE anything is read through it (advisor, round 5: with the check done chunk by chunk beside the
     * copies, off = [0, 3 GiB, 50] sent a 3 GiB copy past the caller's buffer before the descending pair in the next chunk was
     * seen, and earlier chunks had been delivered when the call failed). A single-threaded walk of all of them -- 28 MB for the
     * 3.5 M packets of 2 GiB -- was 3 ms of a 40 ms call in which nothing else happened (round 5, tools/chunk_sweep.py: 7 % of the
     * bus), so large arrays are walked by a few threads: ~0.4 ms. The producer's per-chunk check stays as it costs nothing there. */
    if (!offsets_ascending(off, nblocks)) {
        hsgpu_set_error("block offsets must be ascending and blocks shorter than 4 GiB");
        return HSGPU_INVALID;
    }
    InUse guard(s);
    if (!guard.ok) return HSGPU_SCRATCH_IN_USE;
#ifndef HSGPU_CHUNK_MIB
#define HSGPU_CHUNK_MIB 64 /* tuning builds */
#endif
    /* the caller's chunk size as it is; the default ramps up -- 8, 16, 32, then 64 MiB -- so that the pipeline's fill (nothing is
     * scanned before the first chunk has arrived: 1.2 ms for 64 MiB at 57 GB/s) costs an eighth of that */
    const bool ramp = chunk_bytes == 0;
    if (!chunk_bytes) chunk_bytes = (size_t)HSGPU_CHUNK_MIB << 20;
    std::vector<size_t> cuts; /* block indices: chunk i = blocks [cuts[i], cuts[i + 1]) */
    try {
        cuts.push_back(0);
        while (cuts.back() < nblocks) {
            const uint64_t lo = off[cuts.back()];
            const size_t k = cuts.size() - 1;
            const size_t this_chunk = (ramp && k < 3) ? std::max<size_t>(chunk_bytes >> (3 - k), 1 << 20) : chunk_bytes;
            size_t b = std::upper_bound(off + cuts.back(), off + nblocks + 1, lo + this_chunk) - off; /* first offset past the chunk */
            b = std::max<size_t>(b - 1, cuts.back() + 1); /* at least one block (a block larger than the chunk goes alone) */
            cuts.push_back(std::min(b, nblocks));
        }
        std::mutex mu;
        std::condition_variable cv;
        std::deque<ChunkResult> queue;
        std::atomic<bool> abort{false}, producer_dead{false};
        std::thread producer([&] { (void)produce_chunks(t, s, base, off, nblocks, start, cuts, mu, cv, queue, abort, producer_dead); });
        struct Join { /* whatever happens below (the caller's function may throw): the producer winds down and is joined */
            std::thread &th;
            std::atomic<bool> &abort;
            ~Join() {
                abort = true;
                if (th.joinable()) th.join();
            }
        } join{producer, abort};
        int rv = HSGPU_SUCCESS, stop = 0;
        for (;;) {
            ChunkResult r;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return !queue.empty() || producer_dead; });
                if (queue.empty()) { /* the producer failed without being able to say so */
                    rv = HSGPU_NOMEM;
                    break;
                }
                r = std::move(queue.front());
                queue.pop_front();
            }
            if (r.rv != HSGPU_SUCCESS) {
                rv = r.rv;
            } else if (!stop && !r.recs.empty() && on_chunk(r.recs.data(), r.recs.size(), ctx) != 0) {
                stop = 1; /* the caller has seen enough: the producer winds down */
                abort = true;
            }
            if (r.last) break;
        }
        return rv != HSGPU_SUCCESS ? rv : (stop ? HSGPU_SCAN_TERMINATED : HSGPU_SUCCESS);
    } catch (const std::bad_alloc &) {
        return HSGPU_NOMEM;
    } catch (...) {
        return HSGPU_UNKNOWN_ERROR;
    }
}

extern "C" int hsgpu_hwlm_replay(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n, hsgpu_hwlm_cb cb,
                                 void *ctx, uint64_t groups) {
    if (!t || (n && !recs) || !cb) return HSGPU_HWLM_ERROR_UNKNOWN;
    const HsgpuDevLit *lits = t->lits();
    const uint32_t n_lits = t->hdr()->n_lits;
    uint64_t control = groups;
    uint32_t last_match = 0xffffffffu; /* INVALID_MATCH_ID */
    for (size_t i = 0; i < n; i++) {
        if (recs[i].lit >= n_lits) return HSGPU_HWLM_ERROR_UNKNOWN;
        const HsgpuDevLit &li = lits[recs[i].lit];
        if ((li.flags & HSGPU_LIT_NORUNS) && last_match == li.id) continue; /* fdr_confirm_runtime.h:73-75 */
        if (!(li.groups & control)) continue;                                /* :91 */
        last_match = li.id;
        control = cb(recs[i].end, li.id, ctx);
        if (!control) return HSGPU_HWLM_TERMINATED; /* fdr.c:719-721 */
    }
    return HSGPU_HWLM_SUCCESS;
}

/* the records of a whole batch: the sequential rules start afresh in every block (each block is a scan of
 * its own), and a callback that returns 0 ends ITS block only */
extern "C" int hsgpu_hwlm_replay_batch(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n, hsgpu_hwlm_cb cb,
                                       void *ctx, uint64_t groups, size_t *n_terminated) {
    if (!t || (n && !recs) || !cb) return HSGPU_HWLM_ERROR_UNKNOWN;
    size_t term = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && recs[j].block == recs[i].block) j++;
        const int rv = hsgpu_hwlm_replay(t, recs + i, j - i, cb, ctx, groups);
        if (rv == HSGPU_HWLM_ERROR_UNKNOWN) return rv;
        term += rv == HSGPU_HWLM_TERMINATED;
        i = j;
    }
    if (n_terminated) *n_terminated = term;
    return HSGPU_HWLM_SUCCESS;
}

/* The same on several threads: the blocks are cut into n_threads contiguous ranges (whole blocks: the sequential
 * rules restart in every block, so ranges are independent) and range i is walked in order by one thread with
 * ctxs[i] as the callbacks' context. What hsbench's -T threads do with their own slices of the corpus
 * (tools/hsbench/main.cpp:957-963), applied to the consumer side of the device boundary. */
static void cut_block_ranges(const hsgpu_match_t *recs, size_t n, unsigned parts, std::vector<size_t> &cut) {
    cut.assign(parts + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < parts; t++) {
        size_t c = std::max(cut[t - 1], n * t / parts);
        while (c < n && c > 0 && recs[c].block == recs[c - 1].block) c++;
        cut[t] = c;
    }
}

static int replay_ranges(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n, hsgpu_hwlm_cb cb, void *const *ctxs,
                         unsigned n_threads, uint64_t groups, WorkerPool *pool, size_t *n_terminated) {
    std::vector<size_t> cut;
    cut_block_ranges(recs, n, n_threads, cut);
    std::vector<size_t> term(n_threads, 0);
    std::atomic<int> bad{0};
    auto job = [&](unsigned i) {
        size_t nt = 0;
        if (cut[i + 1] > cut[i] &&
            hsgpu_hwlm_replay_batch(t, recs + cut[i], cut[i + 1] - cut[i], cb, ctxs[i], groups, &nt) != HSGPU_HWLM_SUCCESS)
            bad = 1;
        term[i] = nt;
    };
    if (pool) {
        pool->run(n_threads, job);
    } else {
        std::vector<std::thread> th;
        for (unsigned i = 1; i < n_threads; i++) th.emplace_back(job, i);
        job(0);
        for (std::thread &x : th) x.join();
    }
    if (n_terminated)
        for (size_t v : term) *n_terminated += v;
    return bad ? HSGPU_HWLM_ERROR_UNKNOWN : HSGPU_HWLM_SUCCESS;
}

extern "C" int hsgpu_hwlm_replay_batch_mt(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n, hsgpu_hwlm_cb cb,
                                          void *const *ctxs, unsigned n_threads, uint64_t groups, size_t *n_terminated) {
    if (!t || (n && !recs) || !cb || !ctxs || !n_threads || n_threads > 256) return HSGPU_HWLM_ERROR_UNKNOWN;
    if (n_terminated) *n_terminated = 0;
    try {
        return replay_ranges(t, recs, n, cb, ctxs, n_threads, groups, nullptr, n_terminated);
    } catch (...) {
        return HSGPU_HWLM_ERROR_UNKNOWN;
    }
}

/* From a device-resident scan to callbacks: the records come to the host in four chunks (pinned memory owned by
 * the scratch) and the blocks that have arrived completely are replayed on n_threads threads while the next chunk
 * is on the wire. The consumer side of hsgpu_hwlm_scan_dev at device speed: the single-threaded replay of 0.8 M
 * records took five times as long as the scan that found them. */
extern "C" int hsgpu_hwlm_fetch_replay(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const void *d_out, uint64_t cap,
                                       const void *d_count, void *stream, hsgpu_hwlm_cb cb, void *const *ctxs,
                                       unsigned n_threads, uint64_t groups, size_t *n_records, size_t *n_terminated) {
    if (!t || !s || !d_count || !cb || !ctxs || !n_threads || n_threads > 256 || (cap && !d_out)) return HSGPU_INVALID;
    if (n_records) *n_records = 0;
    if (n_terminated) *n_terminated = 0;
    hipStream_t st = stream ? (hipStream_t)stream : s->stream;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(s->h_count, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t n = *s->h_count;
    if (n_records) *n_records = (size_t)n;
    if (n > cap) return HSGPU_INSUFFICIENT_SPACE; /* the scan delivered nothing: rerun it with room for n */
    if (n == 0) return HSGPU_SUCCESS;
    if (n > s->h_recs_cap) {
        if (s->h_recs) (void)hipHostFree(s->h_recs);
        s->h_recs = nullptr;
        s->h_recs_cap = 0;
        const size_t want = (size_t)(n + n / 4 + 4096);
        HIP_TRY(hipHostMalloc((void **)&s->h_recs, want * sizeof(hsgpu_match_t)));
        s->h_recs_cap = want;
    }
    try {
        if (!s->pool || s->pool->size() + 1 < n_threads) s->pool.reset(new WorkerPool(n_threads - 1));
    } catch (...) {
        return HSGPU_NOMEM;
    }
    const unsigned K = n >= (1u << 16) ? 4 : 1;
    size_t edge[5];
    for (unsigned k = 0; k <= K; k++) edge[k] = (size_t)(n * k / K);
    for (unsigned k = 0; k < K; k++) {
        if (!s->ev_chunk[k]) HIP_TRY(hipEventCreateWithFlags(&s->ev_chunk[k], hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(s->h_recs + edge[k], (const hsgpu_match_t *)d_out + edge[k],
                               (edge[k + 1] - edge[k]) * sizeof(hsgpu_match_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(s->ev_chunk[k], st));
    }
    size_t done = 0;
    int rv = HSGPU_SUCCESS;
    for (unsigned k = 0; k < K; k++) {
        HIP_TRY(hipEventSynchronize(s->ev_chunk[k]));
        size_t upto = edge[k + 1];
        if (k + 1 < K) /* the last block of the prefix may continue in the next chunk: leave it for then */
            while (upto > done && s->h_recs[upto - 1].block == s->h_recs[edge[k + 1] - 1].block) upto--;
        if (upto > done) {
            try {
                if (replay_ranges(t, s->h_recs + done, upto - done, cb, ctxs, n_threads, groups, s->pool.get(), n_terminated) !=
                    HSGPU_HWLM_SUCCESS)
                    rv = HSGPU_UNKNOWN_ERROR;
            } catch (...) {
                rv = HSGPU_NOMEM;
            }
            done = upto;
        }
    }
    return rv;
}

/* hsbench's counting callback (tools/hsbench/engine_hyperscan.cpp:89-97) for the two replay functions:
 * ctx = uint64_t counter */
extern "C" uint64_t hsgpu_hwlm_count_cb(size_t, uint32_t, void *ctx) {
    ++*(uint64_t *)ctx;
    return ~0ull;
}

/* what the callback receives as its third argument: the reference hands the scratch itself to
 * HWLMCallback (src/hwlm/hwlm.h:80-99); a caller that wants its own pointer hangs it here */
extern "C" void hsgpu_scratch_set_context(hsgpu_scratch_t *s, void *ctx) {
    if (s) {
        s->user_ctx = ctx;
        s->has_user_ctx = true;
    }
}
extern "C" void *hsgpu_scratch_get_context(const hsgpu_scratch_t *s) {
    return s ? (s->has_user_ctx ? s->user_ctx : (void *)s) : nullptr;
}

/* hwlmExec's own argument order (src/hwlm/hwlm.h:116-118) */
extern "C" int hsgpu_hwlm_exec(const hsgpu_hwlm_t *t, const uint8_t *buf, size_t len, size_t start,
                               hsgpu_hwlm_cb cb, hsgpu_scratch_t *s, uint64_t groups) {
    if (!t || !s || !cb || (len && !buf)) return HSGPU_HWLM_ERROR_UNKNOWN;
    void *ctx = hsgpu_scratch_get_context(s);
    if (!groups) return HSGPU_HWLM_SUCCESS; /* hwlm.c:178 */
    if (len == 0 || start >= len) return HSGPU_HWLM_SUCCESS;
    InUse guard(s);
    if (!guard.ok) {
        hsgpu_set_error("scratch in use");
        return HSGPU_HWLM_ERROR_UNKNOWN;
    }
    uint64_t off[2] = {0, len};
    std::vector<hsgpu_match_t> recs;
    if (scan_host(t, s, buf, off, 1, start, recs) != HSGPU_SUCCESS) return HSGPU_HWLM_ERROR_UNKNOWN;
    return hsgpu_hwlm_replay(t, recs.data(), recs.size(), cb, ctx, groups);
}

/* hsgpu_hwlm_exec `calls` times from native code, the way hsbench walks its blocks (tools/hsbench/engine_hyperscan.cpp:132-145):
 * what a C caller pays per call (a ctypes call costs ~1 us by itself). */
extern "C" int hsgpu_debug_exec_repeat(const hsgpu_hwlm_t *t, const uint8_t *buf, size_t len, size_t start, hsgpu_hwlm_cb cb, hsgpu_scratch_t *s,
                                       uint64_t groups, unsigned calls, double *us_per_call) {
    if (!calls || !us_per_call) return HSGPU_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 0; i < calls; i++) {
        const int rv = hsgpu_hwlm_exec(t, buf, len, start, cb, s, groups);
        if (rv != HSGPU_HWLM_SUCCESS) return rv;
    }
    *us_per_call = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_hwlm_exec_resident(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const void *d_corpus, uint64_t total_bytes,
                                        const void *d_off, uint64_t nblocks, uint64_t start, const hsgpu_match_t **recs, size_t *nout) {
    if (!t || !s || !recs || !nout || !d_off || (total_bytes && !d_corpus)) return HSGPU_INVALID;
    *recs = nullptr;
    *nout = 0;
    if (nblocks == 0 || total_bytes == 0) return HSGPU_SUCCESS;
    InUse guard(s);
    if (!guard.ok) return HSGPU_SCRATCH_IN_USE;
    HIP_TRY(hipSetDevice(s->device));
    int rv;
    uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(4096, total_bytes / 1024), s->out.cap / sizeof(hsgpu_match_t));
    for (int attempt = 0; attempt < 8; attempt++) {
        if ((rv = s->out.ensure(cap * sizeof(hsgpu_match_t))) != HSGPU_SUCCESS) return rv;
        rv = hsgpu_hwlm_scan_dev(t, s, d_corpus, total_bytes, d_off, nblocks, start, s->out.p, cap, s->count.p, s->stream);
        if (rv != HSGPU_SUCCESS) return rv;
        HIP_TRY(hipMemcpyAsync(s->h_count, s->count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        const uint64_t n = *s->h_count;
        if (n <= cap) {
            if (n > s->h_recs_cap) {
                if (s->h_recs) (void)hipHostFree(s->h_recs);
                s->h_recs = nullptr;
                s->h_recs_cap = 0;
                const size_t want = (size_t)(n + n / 4 + 4096);
                HIP_TRY(hipHostMalloc((void **)&s->h_recs, want * sizeof(hsgpu_match_t)));
                s->h_recs_cap = want;
            }
            if (n) {
                HIP_TRY(hipMemcpyAsync(s->h_recs, s->out.p, n * sizeof(hsgpu_match_t), hipMemcpyDeviceToHost, s->stream));
                HIP_TRY(hipStreamSynchronize(s->stream));
            }
            *recs = s->h_recs;
            *nout = (size_t)n;
            return HSGPU_SUCCESS;
        }
        cap = std::max<uint64_t>(n + n / 4, cap * 2); /* "again": the count is exact or cap + 1; room for skew */
    }
    hsgpu_set_error("match buffer overflow persisted");
    return HSGPU_UNKNOWN_ERROR;
}

extern "C" int hsgpu_hwlm_exec_batch(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const uint8_t *base,
                                     const uint64_t *off, size_t nblocks, size_t start, hsgpu_match_t *out,
                                     size_t cap, size_t *nout) {
    if (!t || !s || !off || !nout || (cap && !out)) return HSGPU_INVALID;
    *nout = 0;
    if (nblocks == 0) return HSGPU_SUCCESS;
    if (!base && off[nblocks] != off[0]) return HSGPU_INVALID;
    InUse guard(s);
    if (!guard.ok) return HSGPU_SCRATCH_IN_USE;
    std::vector<hsgpu_match_t> recs;
    int rv = scan_host(t, s, base, off, nblocks, start, recs);
    if (rv != HSGPU_SUCCESS) return rv;
    *nout = recs.size();
    size_t n = std::min(cap, recs.size());
    if (n) memcpy(out, recs.data(), n * sizeof(hsgpu_match_t));
    return recs.size() > cap ? HSGPU_INSUFFICIENT_SPACE : HSGPU_SUCCESS;
}
