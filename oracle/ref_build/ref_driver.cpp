/*
 * oracle/ref_build/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin C-ABI shim around the *reference's own* literal-matcher hot path,
 * compiled in place from /root/reference/src (never copied into this repo):
 *
 *   hwlmBuildProto / hwlmBuild   src/hwlm/hwlm_build.cpp:121-215
 *   hwlmExec                     src/hwlm/hwlm.c:172-199
 *   fdrBuildProtoHinted          src/fdr/fdr_compile.cpp:900-911  (engine forcing, as unit/internal/fdr.cpp:140-150)
 *   fdrExec                      src/fdr/fdr.c:827-851
 *   shuftiBuildMasks/shuftiExec  src/nfa/shufticompile.cpp:54, src/nfa/shufti.c:150
 *   truffleBuildMasks/truffleExec src/nfa/trufflecompile.cpp:59, src/nfa/truffle.c:118
 *   vermicelliExec & friends     src/nfa/vermicelli.h:42-518
 *
 * Built into oracle/_ref/libhsref.so by oracle/ref_build/Makefile.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / the timed CPU baseline -- never from the product path.
 */
#include "config.h"

#include "hs_compile.h"
#include "grey.h"
#include "hwlm/hwlm.h"
#include "hwlm/hwlm_build.h"
#include "hwlm/hwlm_internal.h"
#include "hwlm/hwlm_literal.h"
#include "fdr/fdr.h"
#include "fdr/fdr_compile.h"
#include "fdr/fdr_compile_internal.h"
#include "fdr/fdr_engine_description.h"
#include "fdr/teddy_engine_description.h"
#include "fdr/fdr_internal.h"
#include "nfa/accel.h"
#include "nfa/shufticompile.h"
#include "nfa/trufflecompile.h"
#include "rose/rose_build_lit_accel.h"
#include "util/alloc.h"
#include "util/bytecode_ptr.h"
#include "util/compare.h"
#include "util/charreach.h"
#include "util/compile_context.h"
#include "util/target_info.h"
#include "scratch.h"

extern "C" {
#include "nfa/shufti.h"
#include "nfa/truffle.h"
#include "nfa/vermicelli.h"
}

#include <cstdint>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace ue2;

extern "C" {

typedef struct hsref_lit {
    const uint8_t *s;
    uint32_t len;
    uint32_t id;
    uint8_t nocase;
    uint8_t noruns;
    uint8_t pad[2];
    uint32_t msk_len;
    uint64_t groups;
    const uint8_t *msk;
    const uint8_t *cmp;
} hsref_lit_t;

typedef uint64_t (*hsref_cb_t)(size_t end, uint32_t id, void *ctx);

} // extern "C"

namespace {

struct RefTable {
    bytecode_ptr<HWLM> hwlm; // normal path (hwlmExec)
    bytecode_ptr<FDR> fdr;   // hinted path (fdrExec directly)
    std::string info;
};

struct CallCtx {
    hsref_cb_t cb;
    void *user;
    uint64_t count;
};

// The reference callback receives only (end, id, scratch); route the caller's
// context through a thread-local exactly as a unit test would through a global.
static thread_local CallCtx *tl_ctx = nullptr;

extern "C" hwlmcb_rv_t trampoline(size_t end, u32 id, struct hs_scratch *) {
    return (hwlmcb_rv_t)tl_ctx->cb(end, id, tl_ctx->user);
}

extern "C" hwlmcb_rv_t counting_cb(size_t, u32, struct hs_scratch *) {
    tl_ctx->count++;
    return HWLM_CONTINUE_MATCHING;
}

target_t make_target(int isa) {
    if (isa == 0) {
        return get_current_target();
    }
    hs_platform_info pi;
    memset(&pi, 0, sizeof(pi));
    pi.tune = HS_TUNE_FAMILY_GENERIC;
    pi.cpu_features = (isa >= 2) ? HS_CPU_FEATURES_AVX2 : 0;
    return target_t(pi);
}

std::vector<hwlmLiteral> to_lits(const hsref_lit_t *lits, size_t n) {
    std::vector<hwlmLiteral> v;
    v.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const hsref_lit_t &l = lits[i];
        std::vector<u8> msk, cmp;
        if (l.msk_len) {
            msk.assign(l.msk, l.msk + l.msk_len);
            cmp.assign(l.cmp, l.cmp + l.msk_len);
        }
        v.emplace_back(std::string((const char *)l.s, l.len), l.nocase != 0,
                       l.noruns != 0, l.id, (hwlm_group_t)l.groups, msk, cmp);
    }
    return v;
}

CharReach to_cr(const uint8_t bitmap[32]) {
    CharReach cr;
    for (unsigned c = 0; c < 256; c++) {
        if (bitmap[c / 8] & (1u << (c % 8))) {
            cr.set(c);
        }
    }
    return cr;
}

} // namespace

extern "C" {

/* isa: 0 = this host's ISA (get_current_target), 1 = force no-AVX2 target,
 * 2 = force AVX2 target.  hint: engine id as in unit/internal/fdr.cpp, or
 * 0xffffffff for the reference's own choice via hwlmBuildProto. */
void *hsref_hwlm_build(const hsref_lit_t *lits, size_t n, int make_small,
                       uint32_t hint, int isa) {
    try {
        std::vector<hwlmLiteral> v = to_lits(lits, n);
        target_t target = make_target(isa);
        Grey grey;
        std::unique_ptr<RefTable> t(new RefTable);
        char buf[160];
        if (hint != 0xffffffffu) {
            auto proto = fdrBuildProtoHinted(HWLM_ENGINE_FDR, v, make_small != 0,
                                             hint, target, grey);
            if (!proto) {
                return nullptr;
            }
            t->fdr = fdrBuildTable(*proto, grey);
            if (!t->fdr) {
                return nullptr;
            }
            snprintf(buf, sizeof(buf), "fdrExec engineID=%u size=%zu (hinted %u)",
                     t->fdr->engineID, fdrSize(t->fdr.get()), hint);
        } else {
            CompileContext cc(false, false, target, grey);
            auto proto = hwlmBuildProto(v, make_small != 0, cc);
            if (!proto) {
                return nullptr;
            }
            t->hwlm = hwlmBuild(*proto, cc);
            if (!t->hwlm) {
                return nullptr;
            }
            const HWLM *h = t->hwlm.get();
            if (h->type == HWLM_ENGINE_NOOD) {
                snprintf(buf, sizeof(buf), "hwlmExec type=noodle size=%zu",
                         hwlmSize(h));
            } else {
                const FDR *f = (const FDR *)HWLM_C_DATA(h);
                snprintf(buf, sizeof(buf),
                         "hwlmExec type=fdr engineID=%u domain=%u stride=%u size=%zu",
                         f->engineID, (unsigned)f->domain, (unsigned)f->stride,
                         hwlmSize(h));
            }
        }
        t->info = buf;
        return t.release();
    } catch (...) {
        return nullptr;
    }
}

void hsref_hwlm_free(void *h) { delete (RefTable *)h; }

const char *hsref_hwlm_info(void *h) { return ((RefTable *)h)->info.c_str(); }

static int run_one(RefTable *t, const uint8_t *buf, size_t len, size_t start,
                   HWLMCallback cb, uint64_t groups) {
    struct hs_scratch scratch; // as unit/internal/fdr.cpp:180-183
    scratch.fdr_conf = NULL;
    if (t->fdr) {
        return (int)fdrExec(t->fdr.get(), buf, len, start, cb, &scratch,
                            (hwlm_group_t)groups);
    }
    return (int)hwlmExec(t->hwlm.get(), buf, len, start, cb, &scratch,
                         (hwlm_group_t)groups);
}

/* Mirror of hwlmExec (src/hwlm/hwlm.h:116-118) with a user context. */
int hsref_hwlm_exec(void *h, const uint8_t *buf, size_t len, size_t start,
                    hsref_cb_t cb, void *ctx, uint64_t groups) {
    CallCtx c{cb, ctx, 0};
    CallCtx *prev = tl_ctx;
    tl_ctx = &c;
    int rv = run_one((RefTable *)h, buf, len, start, trampoline, groups);
    tl_ctx = prev;
    return rv;
}

/* hsbench-style block loop (tools/hsbench/main.cpp:502-528 with the counting
 * callback of engine_hyperscan.cpp:89-97): scan blocks [off[i], off[i+1]) of
 * base, return total number of matches. Used as the timed CPU baseline. */
uint64_t hsref_hwlm_count_blocks(void *h, const uint8_t *base,
                                 const uint64_t *off, size_t nblocks,
                                 size_t start, uint64_t groups) {
    CallCtx c{nullptr, nullptr, 0};
    CallCtx *prev = tl_ctx;
    tl_ctx = &c;
    for (size_t i = 0; i < nblocks; i++) {
        run_one((RefTable *)h, base + off[i], (size_t)(off[i + 1] - off[i]),
                start, counting_cb, groups);
    }
    tl_ctx = prev;
    return c.count;
}

/* Every match of every block as (block, end, id) in the reference's own delivery order
 * (block by block, callbacks as hwlmExec issues them). Returns the total number of matches;
 * only the first `cap` are stored. Used for content-level parity checks. */
struct CollectCtx {
    uint32_t *block, *end, *id;
    size_t cap, n;
    uint32_t cur;
};
static thread_local CollectCtx *tl_collect = nullptr;
extern "C" hwlmcb_rv_t collecting_cb(size_t end, u32 id, struct hs_scratch *) {
    CollectCtx *c = tl_collect;
    if (c->n < c->cap) {
        c->block[c->n] = c->cur;
        c->end[c->n] = (uint32_t)end;
        c->id[c->n] = id;
    }
    c->n++;
    return HWLM_CONTINUE_MATCHING;
}
size_t hsref_hwlm_collect_blocks(void *h, const uint8_t *base, const uint64_t *off, size_t nblocks,
                                 size_t start, uint64_t groups, uint32_t *block, uint32_t *end,
                                 uint32_t *id, size_t cap) {
    CollectCtx c{block, end, id, cap, 0, 0};
    CollectCtx *prev = tl_collect;
    tl_collect = &c;
    for (size_t i = 0; i < nblocks; i++) {
        c.cur = (uint32_t)i;
        run_one((RefTable *)h, base + off[i], (size_t)(off[i + 1] - off[i]), start, collecting_cb, groups);
    }
    tl_collect = prev;
    return c.n;
}

/* hsbench's thread model (tools/hsbench/main.cpp:957-963, 990-1030): T native threads, each
 * pinned to its own CPU (-T style affinity) and scanning its own contiguous slice of the
 * blocks (balanced by bytes) over and over until `seconds` have passed; all threads start
 * behind one barrier. out[0] = bytes scanned (all threads, all passes), out[1] = wall seconds,
 * out[2] = matches of ONE pass over all slices (must equal the single-thread count),
 * out[3] = passes completed by the slowest thread. Returns 0, or -1 when threads cannot start. */
struct BenchArg {
    RefTable *t;
    const uint8_t *base;
    const uint64_t *off;
    size_t lo, hi, start;
    uint64_t groups;
    double seconds;
    int cpu;
    pthread_barrier_t *bar;
    uint64_t bytes, matches_one_pass, passes;
    double t_begin, t_end;
};
static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *bench_thread(void *p) {
    BenchArg *a = (BenchArg *)p;
    if (a->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(a->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    CallCtx c{nullptr, nullptr, 0};
    tl_ctx = &c;
    pthread_barrier_wait(a->bar);
    a->t_begin = now_s();
    const double deadline = a->t_begin + a->seconds;
    const uint64_t slice = a->off[a->hi] - a->off[a->lo];
    do {
        c.count = 0;
        for (size_t i = a->lo; i < a->hi; i++)
            run_one(a->t, a->base + a->off[i], (size_t)(a->off[i + 1] - a->off[i]), a->start, counting_cb, a->groups);
        a->bytes += slice;
        a->passes++;
        a->matches_one_pass = c.count;
    } while (now_s() < deadline);
    a->t_end = now_s();
    tl_ctx = nullptr;
    return nullptr;
}
int hsref_hwlm_bench_threads(void *h, const uint8_t *base, const uint64_t *off, size_t nblocks, size_t start,
                             uint64_t groups, int nthreads, double seconds, int pin, double out[4]) {
    if (nthreads < 1 || nblocks == 0) return -1;
    if ((size_t)nthreads > nblocks) nthreads = (int)nblocks;
    std::vector<BenchArg> args((size_t)nthreads);
    std::vector<pthread_t> th((size_t)nthreads);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, (unsigned)nthreads);
    /* the CPUs this process may run on, in order: thread i is pinned to the i-th of them */
    std::vector<int> cpus;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (pin && sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
    /* slices balanced by bytes */
    const uint64_t total = off[nblocks] - off[0];
    size_t lo = 0;
    for (int i = 0; i < nthreads; i++) {
        const uint64_t want = off[0] + total * (uint64_t)(i + 1) / (uint64_t)nthreads;
        size_t hi = lo;
        while (hi < nblocks && (off[hi + 1] <= want || hi == lo)) hi++;
        if (i == nthreads - 1) hi = nblocks;
        if (hi > nblocks) hi = nblocks;
        BenchArg &a = args[(size_t)i];
        a = BenchArg{(RefTable *)h, base, off, lo, hi, start, groups, seconds,
                     cpus.empty() ? -1 : cpus[(size_t)i % cpus.size()], &bar, 0, 0, 0, 0.0, 0.0};
        lo = hi;
    }
    int started = 0;
    for (; started < nthreads; started++)
        if (pthread_create(&th[(size_t)started], nullptr, bench_thread, &args[(size_t)started]) != 0) break;
    if (started != nthreads) { /* cannot release the barrier with fewer threads: give up cleanly */
        for (int i = 0; i < started; i++) pthread_cancel(th[(size_t)i]);
        for (int i = 0; i < started; i++) pthread_join(th[(size_t)i], nullptr);
        pthread_barrier_destroy(&bar);
        return -1;
    }
    for (int i = 0; i < nthreads; i++) pthread_join(th[(size_t)i], nullptr);
    pthread_barrier_destroy(&bar);
    double t0 = args[0].t_begin, t1 = args[0].t_end;
    uint64_t bytes = 0, matches = 0, min_passes = ~0ull;
    for (const BenchArg &a : args) {
        t0 = a.t_begin < t0 ? a.t_begin : t0;
        t1 = a.t_end > t1 ? a.t_end : t1;
        bytes += a.bytes;
        matches += a.matches_one_pass;
        min_passes = a.passes < min_passes ? a.passes : min_passes;
    }
    out[0] = (double)bytes;
    out[1] = t1 - t0;
    out[2] = (double)matches;
    out[3] = (double)min_passes;
    return 0;
}

static m128 ld128(const uint8_t *p) {
    m128 v;
    memcpy(&v, p, 16);
    return v;
}

/* Config 4's CPU side: for every block (line) and every class, the first and the last member of the class --
 * what the reference's accelerators are asked for (shuftiExec / rshuftiExec, truffleExec / rtruffleExec:
 * src/nfa/shufti.c:150-199, src/nfa/truffle.c:118-165). classes: n x 33 bytes {kind 0 shufti / 1 truffle,
 * mask lo[16] (mask1), mask hi[16] (mask2)}. Same thread model as hsref_hwlm_bench_threads. out[0] = corpus bytes
 * walked (all threads, all passes; one pass = every class over every block), out[1] = wall seconds,
 * out[2] = a checksum of the first/last offsets of one pass, out[3] = passes of the slowest thread. */
struct ClassBenchArg {
    const uint8_t *classes;
    size_t n_classes;
    const uint8_t *base;
    const uint64_t *off;
    size_t lo, hi;
    double seconds;
    int cpu;
    pthread_barrier_t *bar;
    uint64_t bytes, checksum, passes;
    double t_begin, t_end;
};
static void *class_bench_thread(void *p) {
    ClassBenchArg *a = (ClassBenchArg *)p;
    if (a->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(a->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    pthread_barrier_wait(a->bar);
    a->t_begin = now_s();
    const double deadline = a->t_begin + a->seconds;
    const uint64_t slice = a->off[a->hi] - a->off[a->lo];
    do {
        uint64_t sum = 0;
        for (size_t c = 0; c < a->n_classes; c++) {
            const uint8_t *cl = a->classes + 33 * c;
            const m128 lo = ld128(cl + 1), hi = ld128(cl + 17);
            for (size_t i = a->lo; i < a->hi; i++) {
                const uint8_t *b = a->base + a->off[i], *e = a->base + a->off[i + 1];
                const uint8_t *f = cl[0] ? truffleExec(lo, hi, b, e) : shuftiExec(lo, hi, b, e);
                const uint8_t *l = cl[0] ? rtruffleExec(lo, hi, b, e) : rshuftiExec(lo, hi, b, e);
                sum += (uint64_t)(f - b) + 3 * (uint64_t)(l - b + 1);
            }
        }
        a->checksum = sum;
        a->bytes += slice;
        a->passes++;
    } while (now_s() < deadline);
    a->t_end = now_s();
    return nullptr;
}
int hsref_class_bench_threads(const uint8_t *classes, size_t n_classes, const uint8_t *base, const uint64_t *off,
                              size_t nblocks, int nthreads, double seconds, int pin, double out[4]) {
    if (nthreads < 1 || nblocks == 0 || n_classes == 0) return -1;
    if ((size_t)nthreads > nblocks) nthreads = (int)nblocks;
    std::vector<ClassBenchArg> args((size_t)nthreads);
    std::vector<pthread_t> th((size_t)nthreads);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, (unsigned)nthreads);
    std::vector<int> cpus;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (pin && sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
    const uint64_t total = off[nblocks] - off[0];
    size_t lo = 0;
    for (int i = 0; i < nthreads; i++) {
        const uint64_t want = off[0] + total * (uint64_t)(i + 1) / (uint64_t)nthreads;
        size_t hi = lo;
        while (hi < nblocks && (off[hi + 1] <= want || hi == lo)) hi++;
        if (i == nthreads - 1) hi = nblocks;
        args[(size_t)i] = ClassBenchArg{classes, n_classes, base, off, lo, hi, seconds,
                                        cpus.empty() ? -1 : cpus[(size_t)i % cpus.size()], &bar, 0, 0, 0, 0.0, 0.0};
        lo = hi;
    }
    int started = 0;
    for (; started < nthreads; started++)
        if (pthread_create(&th[(size_t)started], nullptr, class_bench_thread, &args[(size_t)started]) != 0) break;
    if (started != nthreads) {
        for (int i = 0; i < started; i++) pthread_cancel(th[(size_t)i]);
        for (int i = 0; i < started; i++) pthread_join(th[(size_t)i], nullptr);
        pthread_barrier_destroy(&bar);
        return -1;
    }
    for (int i = 0; i < nthreads; i++) pthread_join(th[(size_t)i], nullptr);
    pthread_barrier_destroy(&bar);
    double t0 = args[0].t_begin, t1 = args[0].t_end;
    uint64_t bytes = 0, sum = 0, min_passes = ~0ull;
    for (const ClassBenchArg &a : args) {
        t0 = a.t_begin < t0 ? a.t_begin : t0;
        t1 = a.t_end > t1 ? a.t_end : t1;
        bytes += a.bytes;
        sum += a.checksum;
        min_passes = a.passes < min_passes ? a.passes : min_passes;
    }
    out[0] = (double)bytes;
    out[1] = t1 - t0;
    out[2] = (double)(sum & 0xffffffffffffull);
    out[3] = (double)min_passes;
    return 0;
}

/* the ISA this copy of the reference was compiled for (the Teddy / FDR variants are chosen
 * at compile time, src/util/arch.h) */
const char *hsref_build_isa(void) {
#if defined(__AVX512VBMI__)
    return "avx512vbmi";
#elif defined(__AVX512BW__)
    return "avx512bw";
#elif defined(__AVX2__)
    return "avx2";
#else
    return "sse";
#endif
}

/* ---- character-class accelerators ---- */

/* returns number of shufti buckets used, or -1 if the class is not
 * representable (src/nfa/shufticompile.cpp:54-109). */
int hsref_shufti_build(const uint8_t bitmap[32], uint8_t lo[16], uint8_t hi[16]) {
    return shuftiBuildMasks(to_cr(bitmap), lo, hi);
}

void hsref_truffle_build(const uint8_t bitmap[32], uint8_t m1[16], uint8_t m2[16]) {
    truffleBuildMasks(to_cr(bitmap), m1, m2);
}

void hsref_truffle2cr(const uint8_t m1[16], const uint8_t m2[16], uint8_t bitmap[32]) {
    CharReach cr = truffle2cr(m1, m2);
    memset(bitmap, 0, 32);
    for (size_t c = cr.find_first(); c != cr.npos; c = cr.find_next(c)) {
        bitmap[c / 8] |= 1u << (c % 8);
    }
}


/* All exec wrappers return an offset relative to buf (len = "not found" for
 * forward scans; -1 for reverse scans), see src/nfa/shufti.h:40-52. */
int64_t hsref_shufti_exec(const uint8_t lo[16], const uint8_t hi[16],
                          const uint8_t *buf, size_t len) {
    return shuftiExec(ld128(lo), ld128(hi), buf, buf + len) - buf;
}
int64_t hsref_rshufti_exec(const uint8_t lo[16], const uint8_t hi[16],
                           const uint8_t *buf, size_t len) {
    return rshuftiExec(ld128(lo), ld128(hi), buf, buf + len) - buf;
}
int64_t hsref_truffle_exec(const uint8_t m1[16], const uint8_t m2[16],
                           const uint8_t *buf, size_t len) {
    return truffleExec(ld128(m1), ld128(m2), buf, buf + len) - buf;
}
int64_t hsref_rtruffle_exec(const uint8_t m1[16], const uint8_t m2[16],
                            const uint8_t *buf, size_t len) {
    return rtruffleExec(ld128(m1), ld128(m2), buf, buf + len) - buf;
}
int64_t hsref_verm_exec(uint8_t c, int nocase, const uint8_t *buf, size_t len) {
    return vermicelliExec((char)c, (char)nocase, buf, buf + len) - buf;
}
int64_t hsref_nverm_exec(uint8_t c, int nocase, const uint8_t *buf, size_t len) {
    return nvermicelliExec((char)c, (char)nocase, buf, buf + len) - buf;
}
int64_t hsref_dverm_exec(uint8_t c1, uint8_t c2, int nocase, const uint8_t *buf,
                         size_t len) {
    return vermicelliDoubleExec((char)c1, (char)c2, (char)nocase, buf, buf + len) - buf;
}
int64_t hsref_rverm_exec(uint8_t c, int nocase, const uint8_t *buf, size_t len) {
    return rvermicelliExec((char)c, (char)nocase, buf, buf + len) - buf;
}

/* double shufti: masks for a set of 2-byte sequences plus single bytes whose second
 * byte is a wildcard (shuftiBuildDoubleMasks, src/nfa/shufticompile.cpp:135-209);
 * returns 1 on success, 0 when more than 8 buckets would be needed. */
int hsref_dshufti_build(const uint8_t onechar[32], const uint8_t *pairs, size_t npairs,
                        uint8_t lo1[16], uint8_t hi1[16], uint8_t lo2[16], uint8_t hi2[16]) {
    flat_set<std::pair<u8, u8>> two;
    for (size_t i = 0; i < npairs; i++) two.insert(std::make_pair(pairs[2 * i], pairs[2 * i + 1]));
    return shuftiBuildDoubleMasks(to_cr(onechar), two, lo1, hi1, lo2, hi2) ? 1 : 0;
}
int64_t hsref_dshufti_exec(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                           const uint8_t hi2[16], const uint8_t *buf, size_t len) {
    return shuftiDoubleExec(ld128(lo1), ld128(hi1), ld128(lo2), ld128(hi2), buf, buf + len) - buf;
}
int64_t hsref_dverm_masked_exec(uint8_t c1, uint8_t c2, uint8_t m1, uint8_t m2, const uint8_t *buf,
                                size_t len) {
    return vermicelliDoubleMaskedExec((char)c1, (char)c2, (char)m1, (char)m2, buf, buf + len) - buf;
}
int64_t hsref_rdverm_exec(uint8_t c1, uint8_t c2, int nocase, const uint8_t *buf, size_t len) {
    return rvermicelliDoubleExec((char)c1, (char)c2, (char)nocase, buf, buf + len) - buf;
}

/* buildForwardAccel (src/rose/rose_build_lit_accel.cpp:459-465): the pre-skip scheme Rose
 * attaches to a literal matcher. `lits` in the driver's hsref_lit_t form. out[0] = accel1
 * (literals of `expected_groups`), out[1] = accel0 (all groups); each 80 bytes:
 * [0] type [1] offset [2] c/c1 [3] c2 [16..48) lo / mask1 [48..80) hi / mask2 */
void hsref_forward_accel(const hsref_lit_t *lits, size_t n, uint64_t expected_groups, uint8_t out[2][80]) {
    std::vector<AccelString> v;
    for (size_t i = 0; i < n; i++) {
        std::string s((const char *)lits[i].s, lits[i].len);
        if (lits[i].nocase) {
            for (auto &c : s) c = (char)mytoupper((unsigned char)c); /* as hwlmLiteral does */
        }
        std::vector<u8> msk(lits[i].msk, lits[i].msk + lits[i].msk_len), cmp(lits[i].cmp, lits[i].cmp + lits[i].msk_len);
        v.emplace_back(s, lits[i].nocase != 0, msk, cmp, lits[i].groups);
    }
    HWLM *h = (HWLM *)aligned_zmalloc(sizeof(HWLM));
    buildForwardAccel(h, v, expected_groups);
    const AccelAux *aux[2] = {&h->accel1, &h->accel0};
    for (int k = 0; k < 2; k++) {
        memset(out[k], 0, 80);
        const AccelAux &a = *aux[k];
        out[k][0] = a.accel_type;
        out[k][1] = a.generic.offset;
        switch (a.accel_type) {
        case ACCEL_VERM: case ACCEL_VERM_NOCASE: out[k][2] = a.verm.c; break;
        case ACCEL_DVERM: case ACCEL_DVERM_NOCASE: out[k][2] = a.dverm.c1; out[k][3] = a.dverm.c2; break;
        case ACCEL_SHUFTI: memcpy(out[k] + 16, &a.shufti.lo, 16); memcpy(out[k] + 48, &a.shufti.hi, 16); break;
        case ACCEL_TRUFFLE: memcpy(out[k] + 16, &a.truffle.mask1, 16); memcpy(out[k] + 48, &a.truffle.mask2, 16); break;
        default: break;
        }
    }
    aligned_free(h);
}

/* which engine ids are valid for hints on this host (unit/internal/fdr.cpp:114-137) */
size_t hsref_valid_engines(uint32_t *out, size_t cap, int isa) {
    target_t target = make_target(isa);
    std::vector<uint32_t> ret;
    std::vector<FDREngineDescription> fd;
    getFdrDescriptions(&fd);
    for (const auto &d : fd) {
        if (d.isValidOnTarget(target)) ret.push_back(d.getID());
    }
    std::vector<TeddyEngineDescription> td;
    getTeddyDescriptions(&td);
    for (const auto &d : td) {
        if (d.isValidOnTarget(target)) ret.push_back(d.getID());
    }
    for (size_t i = 0; i < ret.size() && i < cap; i++) out[i] = ret[i];
    return ret.size();
}

} // extern "C"
