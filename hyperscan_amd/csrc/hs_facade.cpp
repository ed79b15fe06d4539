/*
 * hs_facade.cpp -- the public hs_* block-mode API (include/hs_gpu.h) on top of the
 * GPU literal engine, with a host-side "Rose-lite" confirm.
 *
 * What each piece stands in for in the reference:
 *   hs_pattern.{h,cpp}   the subset of src/parser/ + src/nfagraph/ this engine needs: an
 *                        expression becomes branches R1 LIT R2 around a mandatory literal
 *   build_database       rose_build_matchers.cpp:701-744: one HWLM literal per branch =
 *                        the last <= 8 bytes of its literal, id = branch index
 *   collect_block_events roseCallback -> roseRunProgram (src/rose/match.c:479-523,
 *                        program_runtime.c): CHECK_MED/LONG_LIT = full-literal compare,
 *                        then R1 backwards / R2 forwards; CHECK_BOUNDS / CHECK_MIN_LENGTH = the
 *                        ext parameters; REPORT -> the user's match_event_handler;
 *                        SINGLEMATCH = the exhaustion vector (src/report.h)
 *   confirm_and_deliver  blocks are independent: worker threads per slice of whole blocks,
 *                        delivery in block order on the caller's thread
 * Matches are reported as the reference does: `to` = offset after the last byte,
 * `from` = 0 unless HS_FLAG_SOM_LEFTMOST, every distinct (id, to) once, in
 * non-decreasing `to`.
 */
#include "../../include/hs_gpu.h"
#include "../../include/hsgpu_tuning.h"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include "hs_pattern.h"
#include "internal.h"

#include <algorithm>
#include <bitset>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

using namespace hsf;

struct hs_database {
    unsigned magic = 0x48534744; /* "HSGD" */
    unsigned mode = HS_MODE_BLOCK;
    std::vector<Pattern> pats;
    hsgpu_hwlm_t *hwlm = nullptr;
    size_t min_width = 0;
    std::vector<std::string> sources; /* for serialisation: original expressions + flags */
    std::vector<unsigned> src_flags, src_ids;
    std::vector<unsigned char> src_is_lit;
    std::vector<hs_expr_ext_t> src_ext; /* flags == 0: none */
    std::set<unsigned> single_ids;      /* report ids carrying HS_FLAG_SINGLEMATCH: the reference's exhaustion keys ("all patterns
                                         * with the same report id share an ekey", src/util/report_manager.cpp:254-257) */
    /* literal-less class sequences A{m,}B{n,}: evaluated on the GPU from the class bitmaps (csrc/class_seq.hip) */
    std::vector<ClassSeq> cseq;
    std::vector<hsgpu_class_t> cs_classes;   /* the distinct classes of cseq */
    std::vector<hsgpu_class_seq_t> cs_seqs;  /* id = index into cseq */
};

/* hs_deserialize_database_at: what the caller's memory holds. A database of this engine owns device memory (the
 * literal table in HBM) and host containers, so it cannot live inside a caller's buffer as the reference's flat
 * bytecode does (src/database.c:170-230); the buffer receives this header, which every entry point follows to the
 * database proper. */
struct hs_database_at {
    unsigned magic; /* "HSGF" */
    unsigned pad;
    hs_database *real;
};
static const unsigned kDbMagic = 0x48534744, kDbAtMagic = 0x48534746;
static inline const hs_database *resolve_db(const hs_database_t *db) {
    if (db && ((const hs_database_at *)db)->magic == kDbAtMagic) return ((const hs_database_at *)db)->real;
    return db;
}

/* The host confirm's worker threads, kept for the life of a scratch: hs_scan_batch's chunked pipeline confirms one chunk of the
 * batch after the other, and starting (and joining) the threads for every chunk was ~0.4 ms of each chunk's budget -- with
 * chunks smaller than 64 MiB the confirm stopped hiding behind the next chunk's copy. run(jobs, f): f(0 .. jobs - 1)
 * on the workers and the calling thread; returns when all are done. f must not throw. */
class HsConfirmPool {
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_idle_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned jobs_ = 0, active_ = 0;
    unsigned long long run_ = 0;
    std::atomic<unsigned> next_{0};
    bool stop_ = false;
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *f;
            unsigned jobs;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || run_ != seen; });
                if (stop_) return;
                seen = run_;
                f = fn_, jobs = jobs_;
                active_++;
            }
            for (unsigned j; (j = next_.fetch_add(1, std::memory_order_relaxed)) < jobs;) (*f)(j);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--active_ == 0) cv_idle_.notify_all();
            }
        }
    }

public:
    size_t size() const { return threads_.size(); }
    void ensure(unsigned n) { /* (a thread that cannot be started is done without) */
        try {
            while (threads_.size() < n) threads_.emplace_back([this] { loop(); });
        } catch (...) {
        }
    }
    void run(unsigned jobs, const std::function<void(unsigned)> &f) {
        if (threads_.empty() || jobs <= 1) {
            for (unsigned j = 0; j < jobs; j++) f(j);
            return;
        }
        {
            /* (a worker that woke up late for the run before still holds that run's function and job count: the job counter
             * goes back to zero only when nobody is looking at it) */
            std::unique_lock<std::mutex> lk(mu_);
            cv_idle_.wait(lk, [&] { return active_ == 0; });
            fn_ = &f, jobs_ = jobs;
            next_.store(0, std::memory_order_relaxed);
            run_++;
        }
        cv_work_.notify_all();
        for (unsigned j; (j = next_.fetch_add(1, std::memory_order_relaxed)) < jobs;) f(j);
        /* every job has been TAKEN; wait for the workers that are still inside one (a worker that wakes up late finds none) */
        std::unique_lock<std::mutex> lk(mu_);
        cv_idle_.wait(lk, [&] { return active_ == 0; });
    }
    ~HsConfirmPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
};

struct hs_scratch {
    unsigned magic = 0x48534753; /* "HSGS" */
    HsConfirmPool pool; /* the host confirm's threads (started with the first large batch) */
    hsgpu_scratch_t *gpu = nullptr;
    bool in_use = false;
    /* record buffer: grown on demand, never value-initialised (a std::vector::resize of the
     * worst-case capacity cost more than the scan) */
    hsgpu_match_t *recs = nullptr;
    size_t recs_cap = 0;
    std::vector<hsgpu_match_t> cs_recs; /* class-sequence matches of the batch */
    ~hs_scratch() { free(recs); }
    bool reserve(size_t n) {
        if (n <= recs_cap) return true;
        free(recs);
        recs = (hsgpu_match_t *)malloc(n * sizeof(hsgpu_match_t));
        recs_cap = recs ? n : 0;
        return recs != nullptr;
    }
};

namespace {

/* allocation hooks, src/alloc.c / src/hs_common.h:273-439 */
struct Hooks {
    hs_alloc_t alloc = nullptr;
    hs_free_t free = nullptr;
};
Hooks g_db, g_misc, g_scratch, g_stream;
void *hook_alloc(const Hooks &h, size_t n) { return h.alloc ? h.alloc(n) : malloc(n); }
void hook_free(const Hooks &h, void *p) {
    if (!p) return;
    if (h.free) h.free(p);
    else free(p);
}
/* hs_check_alloc, src/alloc.c: NULL -> HS_NOMEM, misaligned -> HS_BAD_ALIGN */
hs_error_t check_alloc(const void *p) {
    if (!p) return HS_NOMEM;
    return ((uintptr_t)p & 7) ? HS_BAD_ALIGN : HS_SUCCESS;
}
char *misc_strdup(const std::string &sv) {
    char *m = (char *)hook_alloc(g_misc, sv.size() + 1);
    if (m) memcpy(m, sv.c_str(), sv.size() + 1);
    return m;
}

hs_compile_error_t *make_error(const std::string &msg, int expr) {
    hs_compile_error_t *e = (hs_compile_error_t *)hook_alloc(g_misc, sizeof(*e));
    if (!e) return nullptr;
    e->message = misc_strdup(msg);
    e->expression = expr;
    return e;
}

void destroy_db(hs_database *d);

/* hs_expr_ext_t validation as in src/compiler/compiler.cpp:97-130 (flags known, bounds
 * consistent); approximate matching belongs to the graph compiler and is refused */
void apply_ext(Pattern &p, const hs_expr_ext_t &e) {
    const unsigned long long known = HS_EXT_FLAG_MIN_OFFSET | HS_EXT_FLAG_MAX_OFFSET | HS_EXT_FLAG_MIN_LENGTH |
                                     HS_EXT_FLAG_EDIT_DISTANCE | HS_EXT_FLAG_HAMMING_DISTANCE;
    if (e.flags & ~known) throw ParseError{"Invalid hs_expr_ext flag set."};
    if (e.flags & (HS_EXT_FLAG_EDIT_DISTANCE | HS_EXT_FLAG_HAMMING_DISTANCE))
        throw ParseError{"Approximate matching (edit/Hamming distance) is not supported by the GPU literal engine."};
    if ((e.flags & HS_EXT_FLAG_MIN_OFFSET) && (e.flags & HS_EXT_FLAG_MAX_OFFSET) && e.min_offset > e.max_offset)
        throw ParseError{"In hs_expr_ext, min_offset must be less than or equal to max_offset."};
    if ((e.flags & HS_EXT_FLAG_MIN_LENGTH) && (e.flags & HS_EXT_FLAG_MAX_OFFSET) && e.min_length > e.max_offset)
        throw ParseError{"In hs_expr_ext, min_length must be less than or equal to max_offset."};
    p.ext_flags = e.flags;
    p.min_offset = e.min_offset;
    p.max_offset = e.max_offset;
    p.min_length = e.min_length;
}

/* ext parameters no match could satisfy, with the reference's messages
 * (propagateExtendedParams, src/nfagraph/ng_extparam.cpp:871-905) */
void check_ext_widths(const Pattern *b, size_t n, const hs_expr_ext_t &e) {
    unsigned long long minw = kInf64, maxw = 0;
    bool unbounded = false, anchored = true;
    for (size_t k = 0; k < n; k++) {
        unsigned long long lo, hi;
        bool inf;
        raw_widths(b[k], lo, hi, inf);
        minw = std::min(minw, lo);
        maxw = std::max(maxw, hi);
        unbounded |= inf;
        anchored &= b[k].bol && !b[k].bol_ml;
    }
    char msg[200];
    if ((e.flags & HS_EXT_FLAG_MIN_OFFSET) && anchored && !unbounded && e.min_offset > maxw) {
        snprintf(msg, sizeof(msg), "Expression is anchored and cannot satisfy min_offset=%llu as it can only produce matches "
                 "of length %llu bytes at most.", e.min_offset, maxw);
        throw ParseError{msg};
    }
    if ((e.flags & HS_EXT_FLAG_MAX_OFFSET) && minw > e.max_offset) {
        snprintf(msg, sizeof(msg), "Expression has max_offset=%llu but requires %llu bytes to match.", e.max_offset, minw);
        throw ParseError{msg};
    }
    if ((e.flags & HS_EXT_FLAG_MIN_LENGTH) && !unbounded && maxw < e.min_length) {
        snprintf(msg, sizeof(msg), "Expression has min_length=%llu but can only produce matches of length %llu bytes at most.",
                 e.min_length, maxw);
        throw ParseError{msg};
    }
}

hs_error_t build_database(const std::vector<std::string> &exprs, const std::vector<unsigned char> &is_lit,
                          const unsigned *flags, const unsigned *ids, const hs_expr_ext_t *const *ext, unsigned mode,
                          hs_database_t **db, hs_compile_error_t **error,
                          const std::vector<unsigned char> *gpu_table = nullptr) {
    /* checkMode, src/hs.cpp:78-118: the reference's rules and messages first, then ours */
    const unsigned som_modes = HS_MODE_SOM_HORIZON_LARGE | HS_MODE_SOM_HORIZON_MEDIUM | HS_MODE_SOM_HORIZON_SMALL;
    const unsigned scan_modes = mode & (HS_MODE_BLOCK | HS_MODE_STREAM | HS_MODE_VECTORED);
    const char *mode_err = nullptr;
    if (mode & ~(HS_MODE_BLOCK | HS_MODE_STREAM | HS_MODE_VECTORED | som_modes))
        mode_err = "Invalid parameter: unrecognised mode flags.";
    else if (scan_modes == 0 || (scan_modes & (scan_modes - 1)))
        mode_err = "Invalid parameter: mode must have one (and only one) of HS_MODE_BLOCK, HS_MODE_STREAM or "
                   "HS_MODE_VECTORED set.";
    else if ((mode & som_modes) && !(mode & HS_MODE_STREAM))
        mode_err = "Invalid parameter: the HS_MODE_SOM_HORIZON_ mode flags may only be set in streaming mode.";
    else if ((mode & som_modes) & ((mode & som_modes) - 1))
        mode_err = "Invalid parameter: only one HS_MODE_SOM_HORIZON_ mode flag can be set.";
    else if (mode != HS_MODE_BLOCK && mode != HS_MODE_VECTORED)
        mode_err = "Only HS_MODE_BLOCK and HS_MODE_VECTORED are supported by the GPU literal engine.";
    if (mode_err) {
        *error = make_error(mode_err, -1);
        return HS_COMPILER_ERROR;
    }
    void *mem = hook_alloc(g_db, sizeof(hs_database));
    if (hs_error_t ae = check_alloc(mem)) {
        hook_free(g_db, mem);
        *error = make_error(ae == HS_BAD_ALIGN ? "Database allocator returned misaligned memory."
                                               : "Unable to allocate memory.", -1);
        return HS_COMPILER_ERROR;
    }
    hs_database *d = new (mem) hs_database;
    std::vector<std::string> hw_s;
    try {
        for (size_t i = 0; i < exprs.size(); i++) {
            const unsigned f = flags ? flags[i] : 0, id = ids ? ids[i] : 0;
            const size_t first_of_expr = d->pats.size();
            try {
                if (is_lit[i]) {
                    check_flags(f, true);
                    /* src/compiler/compiler.cpp:405-419: flags the pure-literal API refuses */
                    const unsigned bad = HS_FLAG_DOTALL | HS_FLAG_ALLOWEMPTY | HS_FLAG_UTF8 | HS_FLAG_UCP |
                                         HS_FLAG_PREFILTER | HS_FLAG_COMBINATION | HS_FLAG_QUIET | HS_FLAG_MULTILINE;
                    if (f & bad)
                        throw ParseError{"Only HS_FLAG_CASELESS, HS_FLAG_SINGLEMATCH and HS_FLAG_SOM_LEFTMOST are "
                                         "supported in literal API."};
                    if (exprs[i].empty()) throw ParseError{"Pure literal API doesn't support empty string."};
                    Pattern p;
                    p.lit = exprs[i];
                    p.nocase = f & HS_FLAG_CASELESS;
                    p.single = f & HS_FLAG_SINGLEMATCH;
                    p.som = f & HS_FLAG_SOM_LEFTMOST;
                    p.id = id;
                    d->pats.push_back(p);
                } else {
                    ClassSeq cs;
                    const bool has_ext = ext && ext[i] && ext[i]->flags;
                    if (!has_ext && parse_class_seq(exprs[i], f, cs)) { /* no literal to find: the class-sequence kernel's */
                        cs.id = id;
                        cs.expr = (unsigned)i;
                        d->cseq.push_back(cs);
                        continue;
                    }
                    for (Pattern &b : parse_pattern(exprs[i], f, id)) d->pats.push_back(std::move(b));
                }
                for (size_t k = first_of_expr; k < d->pats.size(); k++) {
                    d->pats[k].expr = (unsigned)i;
                    if (ext && ext[i]) apply_ext(d->pats[k], *ext[i]);
                    finish_pattern(d->pats[k]);
                }
                if (ext && ext[i] && !is_lit[i])
                    check_ext_widths(&d->pats[first_of_expr], d->pats.size() - first_of_expr, *ext[i]);
            } catch (const ParseError &pe) {
                *error = make_error(pe.msg, (int)i);
                destroy_db(d);
                return HS_COMPILER_ERROR;
            }
        }
        /* registerExtReport (src/util/report_manager.cpp:212-234): expressions sharing a match id must agree on
         * HS_FLAG_SINGLEMATCH (they share one exhaustion key) */
        {
            std::map<unsigned, std::pair<bool, unsigned>> seen; /* id -> (single, first expression index) */
            auto check = [&](unsigned id, bool single, unsigned expr) -> bool {
                auto it = seen.find(id);
                if (it == seen.end()) {
                    seen.emplace(id, std::make_pair(single, expr));
                    return true;
                }
                if (it->second.first == single || it->second.second == expr) return true;
                char msg[256];
                snprintf(msg, sizeof(msg),
                         "Expression (index %u) with match ID %u %s HS_FLAG_SINGLEMATCH whereas previous expression (index %u) with the "
                         "same match ID did%s.",
                         expr, id, single ? "specified" : "did not specify", it->second.second, single ? " not" : "");
                *error = make_error(msg, (int)expr);
                return false;
            };
            std::vector<std::pair<unsigned, std::pair<unsigned, bool>>> all; /* (expr, (id, single)) in expression order */
            for (const Pattern &p : d->pats) all.push_back({p.expr, {p.id, p.single}});
            for (const ClassSeq &c : d->cseq) all.push_back({c.expr, {c.id, c.single}});
            std::sort(all.begin(), all.end());
            for (const auto &e : all)
                if (!check(e.second.first, e.second.second, e.first)) {
                    destroy_db(d);
                    return HS_COMPILER_ERROR;
                }
        }
        /* the patterns are final (nothing moves or copies them from here on): equal shift-and tables are shared */
        {
            std::map<std::vector<unsigned long long>, const unsigned long long *> seen;
            for (Pattern &p : d->pats) {
                if (p.reach.empty()) continue;
                auto it = seen.find(p.reach);
                if (it == seen.end()) {
                    seen.emplace(p.reach, p.reach.data());
                } else {
                    p.reach_shared = it->second;
                    std::vector<unsigned long long>().swap(p.reach); /* (its own copy is not needed any more) */
                }
            }
        }
        /* one HWLM literal per pattern: the last <= 8 bytes of the literal prefix */
        std::vector<hsgpu_lit_t> lits(d->pats.size());
        hw_s.resize(d->pats.size());
        d->min_width = ~(size_t)0;
        for (size_t i = 0; i < d->pats.size(); i++) {
            const Pattern &p = d->pats[i];
            hw_s[i] = p.lit.size() > 8 ? p.lit.substr(p.lit.size() - 8) : p.lit;
            memset(&lits[i], 0, sizeof(lits[i]));
            lits[i].s = (const uint8_t *)hw_s[i].data();
            lits[i].len = (uint32_t)hw_s[i].size();
            lits[i].id = (uint32_t)i; /* the "Rose program" of this literal = the pattern index */
            lits[i].nocase = p.nocase;
            lits[i].groups = HSGPU_ALL_GROUPS;
            size_t w = p.lit.size();
            for (const Unit &u : p.tail) w += u.optional ? 0 : 1;
            if (p.general) w += p.g.wmin;
            if (p.has_pre) w += p.pre.wmin;
            d->min_width = std::min(d->min_width, w);
        }
        /* the class sequences: distinct classes, patterns as indices into them */
        for (size_t k = 0; k < d->cseq.size(); k++) {
            const ClassSeq &c = d->cseq[k];
            if (c.quiet) continue; /* (reports nothing: not evaluated at all -- advisor, round 3) */
            auto class_index = [&](const ByteSet &bs) {
                hsgpu_class_t hc;
                memset(&hc, 0, sizeof(hc));
                for (unsigned v = 0; v < 256; v++)
                    if (bs[v]) hc.bitmap[v >> 3] |= (uint8_t)(1u << (v & 7));
                for (size_t q = 0; q < d->cs_classes.size(); q++)
                    if (!memcmp(&d->cs_classes[q], &hc, sizeof(hc))) return q;
                d->cs_classes.push_back(hc);
                return d->cs_classes.size() - 1;
            };
            hsgpu_class_seq_t hs;
            const size_t ia = class_index(c.a), ib = class_index(c.b);
            if (ia > 254 || ib > 254 || d->cseq.size() > HSGPU_SEQ_MAX) {
                *error = make_error("Too many class-sequence patterns for one database.", (int)c.expr);
                destroy_db(d);
                return HS_COMPILER_ERROR;
            }
            hs.a = (uint8_t)ia, hs.b = (uint8_t)ib, hs.m = (uint8_t)c.m, hs.n = (uint8_t)c.n, hs.id = (uint32_t)k;
            d->cs_seqs.push_back(hs);
            d->min_width = std::min<size_t>(d->min_width, c.m + c.n);
        }
        /* a deserialised database brings its GPU table along: no literal compile on load */
        int rv = lits.empty() && !d->cseq.empty() ? HSGPU_SUCCESS /* class sequences only: no literal table */
                 : gpu_table && !gpu_table->empty() ? hsgpu_hwlm_deserialize(gpu_table->data(), gpu_table->size(), &d->hwlm)
                                                    : hsgpu_hwlm_build(lits.data(), lits.size(), 0, &d->hwlm);
        /* a stored table of another table version is simply compiled again (a damaged one of this version is refused:
         * tests/test_hs_reference_api_cpu.py::test_serialised_database_carries_the_gpu_table) */
        if (rv == HSGPU_DB_VERSION_ERROR && !lits.empty() && gpu_table && !gpu_table->empty()) {
            d->hwlm = nullptr;
            rv = hsgpu_hwlm_build(lits.data(), lits.size(), 0, &d->hwlm);
        }
        /* a stored table must be the one these patterns compile to (a blob from a build whose pattern
         * compiler chose other literals, or a spliced one, is not): otherwise compile afresh */
        if (rv == HSGPU_SUCCESS && d->hwlm && gpu_table && !gpu_table->empty() &&
            !hsgpu_table_agrees(d->hwlm, lits.data(), lits.size())) {
            hsgpu_hwlm_free(d->hwlm);
            d->hwlm = nullptr;
            rv = hsgpu_hwlm_build(lits.data(), lits.size(), 0, &d->hwlm);
        }
        if (rv != HSGPU_SUCCESS) {
            *error = make_error(hsgpu_last_error(), -1);
            destroy_db(d);
            return HS_COMPILER_ERROR;
        }
    } catch (const std::bad_alloc &) {
        destroy_db(d);
        *error = make_error("Unable to allocate memory.", -1);
        return HS_COMPILER_ERROR;
    }
    d->mode = mode;
    d->sources = exprs;
    d->src_is_lit = is_lit;
    for (size_t i = 0; i < exprs.size(); i++) {
        d->src_flags.push_back(flags ? flags[i] : 0);
        d->src_ids.push_back(ids ? ids[i] : 0);
        hs_expr_ext_t none;
        memset(&none, 0, sizeof(none));
        d->src_ext.push_back(ext && ext[i] ? *ext[i] : none);
    }
    for (const Pattern &p : d->pats) /* (an expression may have several branches: walk the branches) */
        if (p.single) d->single_ids.insert(p.id);
    for (const ClassSeq &c : d->cseq)
        if (c.single) d->single_ids.insert(c.id);
    *db = d;
    *error = nullptr;
    return HS_SUCCESS;
}

void destroy_db(hs_database *d) {
    if (!d) return;
    hsgpu_hwlm_free(d->hwlm);
    d->~hs_database();
    hook_free(g_db, d);
}

bool lit_matches_at(const Pattern &p, const unsigned char *buf, size_t end /* offset after the literal */) {
    const size_t n = p.lit.size();
    if (end < n) return false;
    const unsigned char *b = buf + end - n;
    if (!p.nocase) return memcmp(b, p.lit.data(), n) == 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char x = b[i], y = (unsigned char)p.lit[i];
        if (is_alpha(x)) x &= 0xdf;
        if (is_alpha(y)) y &= 0xdf;
        if (x != y) return false;
    }
    return true;
}

/* `grp`: what keeps two reports with the same id apart in the reference -- an exhaustion key of its own
 * (SINGLEMATCH) or a start of match (SOM_LEFTMOST) make an expression's Report distinct, so those
 * carry their expression index; plain expressions sharing an id share one Report (kNoGrp) and are
 * reported once per offset (ReportManager::getInternalId, src/util/report_manager.cpp) */
static const unsigned kNoGrp = 0xffffffffu, kSingleGrp = 0xfffffffeu; /* SINGLEMATCH reports: one per id (their ekey) */
struct Event {
    unsigned long long to, from;
    unsigned id, grp;
    bool operator<(const Event &o) const {
        if (to != o.to) return to < o.to;
        if (id != o.id) return id < o.id;
        return grp != o.grp ? grp < o.grp : from < o.from;
    }
    bool operator==(const Event &o) const { return to == o.to && id == o.id && grp == o.grp; }
};

/* turn the HWLM hits of ONE block into the user events it owes, in delivery order:
 * appended to `out` (sorted by to, one per (id, to), SINGLEMATCH ids once).
 * Cost: the hits of one pattern whose report does not depend on where the match began (no
 * start of match, no min_length, no \b layers) share ONE forward pass of the pattern's automaton
 * over the block, with the start state injected at every hit -- O(block length) per pattern hit
 * in the block, not O(hits x length); the other patterns run from each hit as before. */
void collect_block_events(const hs_database *db, const unsigned char *buf, size_t len, const hsgpu_match_t *recs,
                          size_t n, std::vector<Event> &out, std::vector<std::pair<unsigned, size_t>> &by_pat,
                          std::vector<size_t> &starts, const hsgpu_match_t *cs = nullptr, size_t n_cs = 0) {
    const size_t base = out.size();
    /* class-sequence matches of the block (csrc/class_seq.hip): already final, `to` = end + 1 */
    for (size_t k = 0; k < n_cs; k++) {
        if (cs[k].id >= db->cseq.size()) continue;
        const ClassSeq &c = db->cseq[cs[k].id];
        if (c.quiet || (size_t)cs[k].end + 1 > len) continue;
        out.push_back(Event{(unsigned long long)cs[k].end + 1, 0, c.id, c.single ? kSingleGrp : kNoGrp});
    }
    const size_t n_pats = db->pats.size();
    by_pat.clear();
    for (size_t k = 0; k < n; k++) {
        if (recs[k].id >= n_pats) continue; /* not a record of this database's table */
        const Pattern &p = db->pats[recs[k].id];
        if (p.quiet) continue;
        const size_t lit_end = (size_t)recs[k].end + 1;
        if (lit_end > len) continue;
        by_pat.emplace_back(recs[k].id, lit_end);
    }
    std::sort(by_pat.begin(), by_pat.end()); /* (pattern, lit_end) ascending */
    for (size_t g = 0; g < by_pat.size();) {
        size_t g_end = g;
        while (g_end < by_pat.size() && by_pat[g_end].first == by_pat[g].first) g_end++;
        const Pattern &p = db->pats[by_pat[g].first];
        const bool from_matters = p.som || (p.ext_flags & HS_EXT_FLAG_MIN_LENGTH);
        const unsigned grp = p.single ? kSingleGrp : p.som ? p.expr : kNoGrp;
        const bool shared = !from_matters && g_end - g > 1 && (p.general ? !p.g.has_cond : !p.tail.empty());
        /* hs_expr_ext_t bounds: the job of the reference's CHECK_BOUNDS / CHECK_MIN_LENGTH
         * program instructions (src/rose/program_runtime.c) */
        auto in_bounds = [&](unsigned long long to, unsigned long long start) {
            /* `$`: at the end of the data or before its final newline; multiline: before any newline */
            if (p.eol && to != len && !(buf[to] == '\n' && (p.eol_ml || (p.eol_nl && to + 1 == len)))) return false;
            if (!assert_ok(p.as_end, buf, len, to)) return false;
            if ((p.ext_flags & HS_EXT_FLAG_MIN_OFFSET) && to < p.min_offset) return false;
            if ((p.ext_flags & HS_EXT_FLAG_MAX_OFFSET) && to > p.max_offset) return false;
            if ((p.ext_flags & HS_EXT_FLAG_MIN_LENGTH) && to - start < p.min_length) return false;
            return true;
        };
        starts.clear();
        for (size_t k = g; k < g_end; k++) {
            const size_t lit_end = by_pat[k].second;
            if (!lit_matches_at(p, buf, lit_end)) continue; /* long-literal check */
            unsigned long long start = lit_end - p.lit.size();
            if (!assert_ok(p.as_lit_pre, buf, len, start) || !assert_ok(p.as_lit_post, buf, len, lit_end)) continue;
            if (p.has_pre) { /* the part in front of the literal, backwards; `start` becomes the match start */
                size_t f = 0;
                if (!TailNfa::run_reverse(p, buf, len, start, from_matters, f)) continue;
                start = f;
            } else if ((p.bol && start != 0 && !(p.bol_ml && buf[start - 1] == '\n')) || !assert_ok(p.as_start, buf, len, start)) {
                continue;
            }
            if (shared) {
                starts.push_back(lit_end);
                continue;
            }
            const unsigned long long from = p.som ? start : 0;
            auto on_to = [&](size_t to) {
                if (in_bounds(to, start)) out.push_back(Event{to, from, p.id, grp});
                return true;
            };
            if (p.general) {
                TailNfa::run_general(p.g, buf, len, lit_end, on_to);
            } else if (p.tail.empty()) {
                on_to(lit_end);
            } else {
                TailNfa::run64(p, buf, len, lit_end, on_to);
            }
        }
        if (shared && !starts.empty()) {
            auto on_to = [&](size_t to) {
                if (in_bounds(to, 0)) out.push_back(Event{to, 0, p.id, grp});
                return true;
            };
            if (p.general)
                TailNfa::run_general_multi(p.g, buf, len, starts.data(), starts.size(), on_to);
            else
                TailNfa::run64_multi(p, buf, len, starts.data(), starts.size(), on_to);
        }
        g = g_end;
    }
    std::sort(out.begin() + base, out.end());
    out.erase(std::unique(out.begin() + base, out.end()), out.end()); /* one report per (id, to) */
    if (!db->single_ids.empty()) {
        std::set<unsigned> exhausted; /* SINGLEMATCH ids already reported in this block (the reference's exhaustion vector) */
        size_t w = base;
        for (size_t r = base; r < out.size(); r++) {
            if (out[r].grp == kSingleGrp && !exhausted.insert(out[r].id).second) continue;
            out[w++] = out[r];
        }
        out.resize(w);
    }
}

/* one contiguous slice of the record array, whole blocks only: events of every block in it */
struct BlockRun {
    unsigned long long block;
    size_t ev_begin, ev_end;
};
void collect_slice(const hs_database *db, const unsigned char *data, const unsigned long long *off,
                   const hsgpu_match_t *recs, size_t lo, size_t hi, std::vector<Event> &events,
                   std::vector<BlockRun> &runs, const hsgpu_match_t *cs = nullptr, size_t cs_lo = 0, size_t cs_hi = 0) {
    size_t k = lo, c = cs_lo;
    std::vector<std::pair<unsigned, size_t>> by_pat;
    std::vector<size_t> starts;
    /* A literal hit every few KiB of a multi-GiB batch: the hit's offsets, then its bytes, are two DEPENDENT cache misses per
     * hit (round 6: 270 ns per hit cold against 140 warm on one core). The records say where the next hits are, so their
     * offsets are asked for 16 records ahead and their bytes -- the line the literal ends in and the one behind it, where the
     * tail runs -- 8 ahead, by which time the offsets have arrived. */
    constexpr size_t PF_OFF = 16, PF_DATA = 8;
    size_t pf = lo; /* records [lo, pf) have had their prefetches issued */
    while (k < hi || c < cs_hi) { /* both lists are sorted by (block, end): one run per block, blocks in order */
        while (pf < hi && pf <= k) { /* keep the window [k, k + PF_OFF) primed */
            if (pf + PF_OFF < hi) __builtin_prefetch(&off[recs[pf + PF_OFF].block]);
            if (pf + PF_DATA < hi) {
                const hsgpu_match_t &r = recs[pf + PF_DATA];
                const unsigned char *p = data + off[r.block] + r.end;
                __builtin_prefetch(p);
                __builtin_prefetch(p + 64);
                __builtin_prefetch(&db->pats[r.id < db->pats.size() ? r.id : 0]);
            }
            pf++;
        }
        const unsigned long long b = std::min<unsigned long long>(k < hi ? recs[k].block : ~0ull, c < cs_hi ? cs[c].block : ~0ull);
        size_t e = k, ce = c;
        while (e < hi && recs[e].block == b) e++;
        while (ce < cs_hi && cs[ce].block == b) ce++;
        const size_t len = (size_t)(off[b + 1] - off[b]);
        const size_t ev0 = events.size();
        if (len >= db->min_width)
            collect_block_events(db, data + off[b], len, recs + k, e - k, events, by_pat, starts, cs ? cs + c : nullptr, ce - c);
        if (events.size() > ev0) runs.push_back(BlockRun{b, ev0, events.size()});
        k = e;
        c = ce;
    }
}

} // namespace

extern "C" {

hs_error_t hs_compile_ext_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                                const hs_expr_ext_t *const *ext, unsigned int elements, unsigned int mode,
                                const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
    if (platform) { /* checkPlatform, src/hs.cpp:122-146 */
        const char *perr = nullptr;
        if (platform->cpu_features & ~(HS_CPU_FEATURES_AVX2 | HS_CPU_FEATURES_AVX512 | HS_CPU_FEATURES_AVX512VBMI))
            perr = "Invalid cpu features specified in the platform information.";
        else if (platform->tune > HS_TUNE_FAMILY_ICX)
            perr = "Invalid tuning value specified in the platform information.";
        if (perr) {
            if (db) *db = nullptr;
            if (error) *error = make_error(perr, -1);
            return HS_COMPILER_ERROR;
        }
    }
    if (!error) {
        if (db) *db = nullptr;
        return HS_COMPILER_ERROR;
    }
    if (!db) { *error = make_error("Invalid parameter: db is NULL", -1); return HS_COMPILER_ERROR; }
    *db = nullptr;
    if (!expressions) { *error = make_error("Invalid parameter: expressions is NULL", -1); return HS_COMPILER_ERROR; }
    if (elements == 0) { *error = make_error("Invalid parameter: elements is zero", -1); return HS_COMPILER_ERROR; }
    std::vector<std::string> ex;
    for (unsigned i = 0; i < elements; i++) {
        if (!expressions[i]) { *error = make_error("Invalid parameter: expression is NULL", (int)i); return HS_COMPILER_ERROR; }
        ex.push_back(expressions[i]);
    }
    return build_database(ex, std::vector<unsigned char>(elements, 0), flags, ids, ext, mode, db, error);
}

hs_error_t hs_compile_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                            unsigned int elements, unsigned int mode, const hs_platform_info_t *platform,
                            hs_database_t **db, hs_compile_error_t **error) {
    return hs_compile_ext_multi(expressions, flags, ids, nullptr, elements, mode, platform, db, error);
}

hs_error_t hs_compile(const char *expression, unsigned int flags, unsigned int mode, const hs_platform_info_t *platform,
                      hs_database_t **db, hs_compile_error_t **error) {
    if (!expression) {
        if (db) *db = nullptr;
        if (error) *error = make_error("Invalid parameter: expression is NULL", -1);
        return HS_COMPILER_ERROR;
    }
    unsigned id = 0;
    return hs_compile_multi(&expression, &flags, &id, 1, mode, platform, db, error);
}

hs_error_t hs_compile_lit_multi(const char *const *expressions, const unsigned *flags, const unsigned *ids,
                                const size_t *lens, unsigned elements, unsigned mode,
                                const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
    if (platform) { /* checkPlatform, src/hs.cpp:122-146 */
        const char *perr = nullptr;
        if (platform->cpu_features & ~(HS_CPU_FEATURES_AVX2 | HS_CPU_FEATURES_AVX512 | HS_CPU_FEATURES_AVX512VBMI))
            perr = "Invalid cpu features specified in the platform information.";
        else if (platform->tune > HS_TUNE_FAMILY_ICX)
            perr = "Invalid tuning value specified in the platform information.";
        if (perr) {
            if (db) *db = nullptr;
            if (error) *error = make_error(perr, -1);
            return HS_COMPILER_ERROR;
        }
    }
    if (!error) {
        if (db) *db = nullptr;
        return HS_COMPILER_ERROR;
    }
    if (!db) { *error = make_error("Invalid parameter: db is NULL", -1); return HS_COMPILER_ERROR; }
    *db = nullptr;
    if (!expressions) { *error = make_error("Invalid parameter: expressions is NULL", -1); return HS_COMPILER_ERROR; }
    if (!lens) { *error = make_error("Invalid parameter: len is NULL", -1); return HS_COMPILER_ERROR; }
    if (elements == 0) { *error = make_error("Invalid parameter: elements is zero", -1); return HS_COMPILER_ERROR; }
    std::vector<std::string> ex;
    for (unsigned i = 0; i < elements; i++) ex.emplace_back(expressions[i] ? expressions[i] : "", expressions[i] ? lens[i] : 0);
    return build_database(ex, std::vector<unsigned char>(elements, 1), flags, ids, nullptr, mode, db, error);
}

hs_error_t hs_compile_lit(const char *expression, unsigned flags, const size_t len, unsigned mode,
                          const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
    if (!expression) {
        if (db) *db = nullptr;
        if (error) *error = make_error("Invalid parameter: expression is NULL", -1);
        return HS_COMPILER_ERROR;
    }
    unsigned id = 0;
    return hs_compile_lit_multi(&expression, &flags, &id, &len, 1, mode, platform, db, error);
}

hs_error_t hs_free_compile_error(hs_compile_error_t *error) {
    if (!error) return HS_SUCCESS;
    hook_free(g_misc, error->message);
    hook_free(g_misc, error);
    return HS_SUCCESS;
}

hs_error_t hs_free_database(hs_database_t *db) {
    if (!db) return HS_SUCCESS;
    if (((hs_database_at *)db)->magic == kDbAtMagic) { /* placed by hs_deserialize_database_at: the memory is the caller's */
        hs_database_at *h = (hs_database_at *)db;
        if (h->real) destroy_db(h->real);
        h->real = nullptr;
        h->magic = 0;
        return HS_SUCCESS;
    }
    if (db->magic != 0x48534744) return HS_INVALID;
    destroy_db(db);
    return HS_SUCCESS;
}

hs_error_t hs_database_size(const hs_database_t *db, size_t *size) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || !size || db->magic != 0x48534744) return HS_INVALID;
    size_t s = sizeof(*db) + hsgpu_hwlm_size(db->hwlm);
    for (const Pattern &p : db->pats) s += sizeof(p) + p.lit.size() + p.tail.size() * sizeof(Unit) + p.reach.size() * 8 + p.g.bytes() + p.pre.bytes();
    s += db->cseq.size() * (sizeof(ClassSeq) + sizeof(hsgpu_class_seq_t)) + db->cs_classes.size() * sizeof(hsgpu_class_t);
    *size = s;
    return HS_SUCCESS;
}

hs_error_t hs_database_info(const hs_database_t *db, char **info) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || !info || db->magic != 0x48534744) return HS_INVALID;
    char buf[160];
    snprintf(buf, sizeof(buf), "Version: %s Features: gfx950 Mode: %s", hs_version(),
             db->mode == HS_MODE_VECTORED ? "VECTORED" : "BLOCK");
    *info = misc_strdup(buf);
    if (hs_error_t ae = check_alloc(*info)) {
        hook_free(g_misc, *info);
        *info = nullptr;
        return ae;
    }
    return HS_SUCCESS;
}

/* src/hs_runtime.h:294 / src/runtime.c hs_stream_size: a database that was not compiled for
 * streaming answers HS_DB_MODE_ERROR (unit/hyperscan/single.cpp:72-83), and none here is. */
hs_error_t hs_stream_size(const hs_database_t *db, size_t *stream_size) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || !stream_size || db->magic != 0x48534744) return HS_INVALID;
    return HS_DB_MODE_ERROR;
}

/* serialised form: magic "HSGH", CRC-32 of everything after it, count, front-end revision, then per pattern
 * (top bit: vectored mode) {is_lit, flags, id, len, ext flags, min_offset, max_offset, min_length, bytes},
 * then the GPU table section {u64 length, the hsgpu_hwlm_serialize image}. On load the host
 * automata are rebuilt from the sources (microseconds each) and the GPU table is taken from its
 * section as it is. The reference guards its bytecode with a CRC too
 * (src/database.c:119-168): a damaged blob is HS_INVALID, never a different database. */
static const unsigned kSerialMagic = 0x48534748; /* "HSGH": version 3 = sources + front-end revision + GPU table section */
/* bumped whenever the pattern compiler may pick other literals / another branch order for the same
 * sources: a blob from another revision still loads (the sources are what it stores), but its GPU
 * table is compiled afresh instead of being trusted */
static const unsigned kFrontEndRevision = 3;
static const unsigned kSerialVectored = 0x80000000u; /* top bit of the count word: HS_MODE_VECTORED */

static unsigned crc32_of(const unsigned char *p, size_t n) {
    struct Table {
        unsigned v[256];
        Table() {
            for (unsigned i = 0; i < 256; i++) {
                unsigned c = i;
                for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                v[i] = c;
            }
        }
    };
    static const Table table; /* initialised once, thread-safely (C++11 function-local static) */
    unsigned c = ~0u;
    for (size_t i = 0; i < n; i++) c = table.v[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}

hs_error_t hs_serialize_database(const hs_database_t *db, char **bytes, size_t *length) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || !bytes || !length || db->magic != 0x48534744) return HS_INVALID;
    std::string out;
    auto put32 = [&](unsigned v) { out.append((const char *)&v, 4); };
    auto put64 = [&](unsigned long long v) { out.append((const char *)&v, 8); };
    put32(kSerialMagic);
    put32(0); /* CRC, filled in below */
    put32((unsigned)db->sources.size() | (db->mode == HS_MODE_VECTORED ? kSerialVectored : 0));
    put32(kFrontEndRevision);
    for (size_t i = 0; i < db->sources.size(); i++) {
        put32(db->src_is_lit[i]);
        put32(db->src_flags[i]);
        put32(db->src_ids[i]);
        put32((unsigned)db->sources[i].size());
        put64(db->src_ext[i].flags);
        put64(db->src_ext[i].min_offset);
        put64(db->src_ext[i].max_offset);
        put64(db->src_ext[i].min_length);
        out += db->sources[i];
    }
    /* the GPU literal table, as hsgpu_hwlm_serialize writes it: a deserialised database scans
     * without compiling its literals again */
    size_t tlen = 0;
    if (db->hwlm) { /* (a database of class sequences only has no literal table: an empty section) */
        if (hsgpu_hwlm_serialize(db->hwlm, nullptr, 0, &tlen) != HSGPU_SUCCESS) return HS_UNKNOWN_ERROR;
        std::string table(tlen, '\0');
        if (hsgpu_hwlm_serialize(db->hwlm, &table[0], table.size(), &tlen) != HSGPU_SUCCESS) return HS_UNKNOWN_ERROR;
        put64(tlen);
        out += table;
    } else {
        put64(0);
    }
    const unsigned crc = crc32_of((const unsigned char *)out.data() + 8, out.size() - 8);
    memcpy(&out[4], &crc, 4);
    *bytes = (char *)hook_alloc(g_misc, out.size());
    if (hs_error_t ae = check_alloc(*bytes)) {
        hook_free(g_misc, *bytes);
        *bytes = nullptr;
        return ae;
    }
    memcpy(*bytes, out.data(), out.size());
    *length = out.size();
    return HS_SUCCESS;
}

namespace {
struct Serial {
    std::vector<std::string> ex;
    std::vector<unsigned char> is_lit;
    std::vector<unsigned> flags, ids;
    std::vector<hs_expr_ext_t> ext;
    unsigned mode = HS_MODE_BLOCK;
    std::vector<unsigned char> table; /* the GPU table section */
};
hs_error_t parse_serial(const char *bytes, size_t length, Serial &out) {
    if (!bytes) return HS_INVALID;
    size_t off = 0;
    auto get32 = [&](unsigned &v) {
        if (off + 4 > length) return false;
        memcpy(&v, bytes + off, 4);
        off += 4;
        return true;
    };
    auto get64 = [&](unsigned long long &v) {
        if (off + 8 > length) return false;
        memcpy(&v, bytes + off, 8);
        off += 8;
        return true;
    };
    unsigned magic, n;
    if (!get32(magic)) return HS_INVALID;
    if ((magic & 0xffffff00u) == (kSerialMagic & 0xffffff00u) && magic != kSerialMagic) return HS_DB_VERSION_ERROR;
    unsigned crc = 0;
    if (magic != kSerialMagic || !get32(crc) || length < 12 ||
        crc != crc32_of((const unsigned char *)bytes + 8, length - 8))
        return HS_INVALID;
    if (!get32(n)) return HS_INVALID;
    if (n & kSerialVectored) out.mode = HS_MODE_VECTORED;
    n &= ~kSerialVectored;
    if (n == 0) return HS_INVALID;
    unsigned revision = 0;
    if (!get32(revision)) return HS_INVALID;
    for (unsigned i = 0; i < n; i++) {
        unsigned l, f, id, len;
        hs_expr_ext_t e;
        memset(&e, 0, sizeof(e));
        if (!get32(l) || !get32(f) || !get32(id) || !get32(len) || !get64(e.flags) || !get64(e.min_offset) ||
            !get64(e.max_offset) || !get64(e.min_length) || off + len > length)
            return HS_INVALID;
        out.ex.emplace_back(bytes + off, len);
        off += len;
        out.is_lit.push_back((unsigned char)l);
        out.flags.push_back(f);
        out.ids.push_back(id);
        out.ext.push_back(e);
    }
    unsigned long long tlen = 0;
    if (!get64(tlen) || tlen > length - off) return HS_INVALID;
    if (revision == kFrontEndRevision) /* else: compiled afresh from the sources */
        out.table.assign((const unsigned char *)bytes + off, (const unsigned char *)bytes + off + tlen);
    off += tlen;
    return off == length ? HS_SUCCESS : HS_INVALID;
}
} // namespace

hs_error_t hs_deserialize_database(const char *bytes, const size_t length, hs_database_t **db) {
    if (!bytes || !db) return HS_INVALID;
    *db = nullptr;
    Serial sr;
    if (hs_error_t rv = parse_serial(bytes, length, sr)) return rv;
    std::vector<const hs_expr_ext_t *> ext;
    for (const hs_expr_ext_t &e : sr.ext) ext.push_back(e.flags ? &e : nullptr);
    hs_compile_error_t *err = nullptr;
    hs_error_t rv = build_database(sr.ex, sr.is_lit, sr.flags.data(), sr.ids.data(), ext.data(), sr.mode, db, &err, &sr.table);
    hs_free_compile_error(err);
    return rv == HS_SUCCESS ? HS_SUCCESS : HS_INVALID;
}

/* hs_deserialize_database_at (src/hs_common.h:147-169, src/database.c:170-230): the reference rebuilds its flat
 * bytecode inside memory the caller provides (>= hs_serialized_database_size bytes, 8-byte aligned). A database of
 * this engine owns device memory and host containers, which no caller's buffer can hold: the buffer receives a
 * 16-byte header that every entry point follows to the database proper. One difference follows and is documented
 * in include/hs_gpu.h: the caller still frees its buffer itself, but calls hs_free_database(db) first to release
 * what the database owns outside it (the reference's caller just frees the buffer). */
hs_error_t hs_deserialize_database_at(const char *bytes, const size_t length, hs_database_t *db) {
    if (!bytes || !db) return HS_INVALID;
    if ((uintptr_t)db & 7) return HS_BAD_ALIGN; /* database.c:187-189 */
    hs_database_t *real = nullptr;
    if (hs_error_t rv = hs_deserialize_database(bytes, length, &real)) return rv;
    hs_database_at *h = (hs_database_at *)db;
    h->magic = kDbAtMagic;
    h->pad = 0;
    h->real = real;
    return HS_SUCCESS;
}

/* src/hs_common.h:196-271: how much a deserialised database will occupy / what it is,
 * without building it */
hs_error_t hs_serialized_database_size(const char *bytes, const size_t length, size_t *size) {
    if (!size) return HS_INVALID;
    Serial sr;
    if (hs_error_t rv = parse_serial(bytes, length, sr)) return rv;
    hs_database_t *db = nullptr;
    if (hs_error_t rv = hs_deserialize_database(bytes, length, &db)) return rv;
    hs_error_t rv = hs_database_size(db, size);
    hs_free_database(db);
    return rv;
}

hs_error_t hs_serialized_database_info(const char *bytes, size_t length, char **info) {
    if (!info) return HS_INVALID;
    *info = nullptr;
    Serial sr;
    if (hs_error_t rv = parse_serial(bytes, length, sr)) return rv;
    char buf[160];
    snprintf(buf, sizeof(buf), "Version: %s Features: gfx950 Mode: %s", hs_version(),
             sr.mode == HS_MODE_VECTORED ? "VECTORED" : "BLOCK");
    *info = misc_strdup(buf);
    if (hs_error_t ae = check_alloc(*info)) {
        hook_free(g_misc, *info);
        *info = nullptr;
        return ae;
    }
    return HS_SUCCESS;
}

/* ---- allocators (src/hs_common.h:273-439, src/alloc.c) ---- */
hs_error_t hs_set_database_allocator(hs_alloc_t a, hs_free_t f) { g_db.alloc = a; g_db.free = f; return HS_SUCCESS; }
hs_error_t hs_set_misc_allocator(hs_alloc_t a, hs_free_t f) { g_misc.alloc = a; g_misc.free = f; return HS_SUCCESS; }
hs_error_t hs_set_scratch_allocator(hs_alloc_t a, hs_free_t f) { g_scratch.alloc = a; g_scratch.free = f; return HS_SUCCESS; }
hs_error_t hs_set_stream_allocator(hs_alloc_t a, hs_free_t f) { g_stream.alloc = a; g_stream.free = f; return HS_SUCCESS; }
hs_error_t hs_set_allocator(hs_alloc_t a, hs_free_t f) {
    hs_set_database_allocator(a, f);
    hs_set_misc_allocator(a, f);
    hs_set_scratch_allocator(a, f);
    hs_set_stream_allocator(a, f);
    return HS_SUCCESS;
}

hs_error_t hs_populate_platform(hs_platform_info_t *platform) {
    if (!platform) return HS_INVALID;
    memset(platform, 0, sizeof(*platform)); /* tune = HS_TUNE_FAMILY_GENERIC, no CPU features: the engine is a GPU */
    return HS_SUCCESS;
}

/* hs_expression_info / _ext_info (src/hs.cpp:413-516): widths of the supported pattern
 * subset; never unordered, never EOD-anchored */
hs_error_t hs_expression_ext_info(const char *expression, unsigned int flags, const hs_expr_ext_t *ext,
                                  hs_expr_info_t **info, hs_compile_error_t **error) {
    if (!error) return HS_COMPILER_ERROR;
    *error = nullptr;
    if (!info) { *error = make_error("Invalid parameter: info is NULL", -1); return HS_COMPILER_ERROR; }
    *info = nullptr;
    if (!expression) { *error = make_error("Invalid parameter: expression is NULL", -1); return HS_COMPILER_ERROR; }
    std::vector<Pattern> branches;
    ClassSeq cseq;
    if (!(ext && ext->flags) && parse_class_seq(expression, flags, cseq)) { /* A{m,}B{n,}: at least m + n bytes, no upper bound */
        hs_expr_info_t *out = (hs_expr_info_t *)hook_alloc(g_misc, sizeof(*out));
        if (hs_error_t ae = check_alloc(out)) {
            hook_free(g_misc, out);
            *error = make_error(ae == HS_BAD_ALIGN ? "Allocator returned misaligned memory." : "Unable to allocate memory.", -1);
            return HS_COMPILER_ERROR;
        }
        out->min_width = cseq.m + cseq.n;
        out->max_width = 0xffffffffu;
        out->unordered_matches = 0;
        out->matches_at_eod = 0;
        out->matches_only_at_eod = 0;
        *info = out;
        return HS_SUCCESS;
    }
    try {
        branches = parse_pattern(expression, flags, 0);
        if (ext) {
            for (Pattern &b : branches) apply_ext(b, *ext);
            check_ext_widths(branches.data(), branches.size(), *ext);
        }
    } catch (const ParseError &pe) {
        *error = make_error(pe.msg, 0);
        return HS_COMPILER_ERROR;
    }
    unsigned long long minw = kInf64, maxw = 0;
    bool unbounded = false, any_eol = false, all_eol = true, eol_multiline = false, unordered = false, at_eod = false;
    for (const Pattern &p : branches) {
        unsigned long long lo, hi;
        bool inf;
        raw_widths(p, lo, hi, inf);
        if (p.ext_flags & HS_EXT_FLAG_MIN_LENGTH) lo = std::max(lo, p.min_length);
        if (p.ext_flags & HS_EXT_FLAG_MAX_OFFSET) {
            hi = inf ? p.max_offset : std::min(hi, p.max_offset);
            inf = false;
        }
        minw = std::min(minw, lo);
        maxw = std::max(maxw, hi);
        unbounded |= inf;
        any_eol |= p.eol;
        all_eol &= p.eol;
        eol_multiline |= p.eol && p.eol_ml;
        unordered |= (p.eol && p.eol_nl) || p.as_end;
        at_eod |= p.eol || p.as_end;
    }
    hs_expr_info_t *out = (hs_expr_info_t *)hook_alloc(g_misc, sizeof(*out));
    if (hs_error_t ae = check_alloc(out)) {
        hook_free(g_misc, out);
        *error = make_error(ae == HS_BAD_ALIGN ? "Allocator returned misaligned memory." : "Unable to allocate memory.", -1);
        return HS_COMPILER_ERROR;
    }
    out->min_width = (unsigned)std::min<unsigned long long>(minw, 0xffffffffu);
    out->max_width = unbounded ? 0xffffffffu : (unsigned)std::min<unsigned long long>(maxw, 0xffffffffu);
    /* a branch may be completed by the end of the data (expr_info.cpp:190-206): "foobar$" and \Z
     * are unordered / at EOD / only at EOD (the multiline `$` also matches before inner newlines),
     * \z is ordered and only at EOD, a closing \b is unordered and at EOD but not only there */
    out->unordered_matches = unordered;
    out->matches_at_eod = at_eod;
    out->matches_only_at_eod = all_eol && !eol_multiline;
    *info = out;
    return HS_SUCCESS;
}

hs_error_t hs_expression_info(const char *expression, unsigned int flags, hs_expr_info_t **info,
                              hs_compile_error_t **error) {
    return hs_expression_ext_info(expression, flags, nullptr, info, error);
}

hs_error_t hs_alloc_scratch(const hs_database_t *db, hs_scratch_t **scratch) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || !scratch || db->magic != 0x48534744) return HS_INVALID;
    if (*scratch) return (*scratch)->magic == 0x48534753 ? ((*scratch)->in_use ? HS_SCRATCH_IN_USE : HS_SUCCESS) : HS_INVALID;
    void *mem = hook_alloc(g_scratch, sizeof(hs_scratch));
    if (hs_error_t ae = check_alloc(mem)) {
        hook_free(g_scratch, mem);
        return ae;
    }
    hs_scratch *s = new (mem) hs_scratch;
    int rv = hsgpu_scratch_alloc(&s->gpu, -1);
    if (rv != HSGPU_SUCCESS) {
        s->~hs_scratch();
        hook_free(g_scratch, mem);
        return rv == HSGPU_NOMEM ? HS_NOMEM : HS_UNKNOWN_ERROR;
    }
    *scratch = s;
    return HS_SUCCESS;
}

hs_error_t hs_clone_scratch(const hs_scratch_t *src, hs_scratch_t **dest) {
    if (!src || !dest || src->magic != 0x48534753) return HS_INVALID;
    *dest = nullptr;
    void *mem = hook_alloc(g_scratch, sizeof(hs_scratch));
    if (hs_error_t ae = check_alloc(mem)) {
        hook_free(g_scratch, mem);
        return ae;
    }
    hs_scratch *s = new (mem) hs_scratch;
    if (hsgpu_scratch_alloc(&s->gpu, -1) != HSGPU_SUCCESS) {
        s->~hs_scratch();
        hook_free(g_scratch, mem);
        return HS_UNKNOWN_ERROR;
    }
    *dest = s;
    return HS_SUCCESS;
}

hs_error_t hs_scratch_size(const hs_scratch_t *scratch, size_t *size) {
    if (!scratch || !size || scratch->magic != 0x48534753) return HS_INVALID;
    *size = sizeof(*scratch);
    return HS_SUCCESS;
}

hs_error_t hs_free_scratch(hs_scratch_t *scratch) {
    if (!scratch) return HS_SUCCESS;
    if (scratch->magic != 0x48534753) return HS_INVALID;
    if (scratch->in_use) return HS_SCRATCH_IN_USE;
    hsgpu_scratch_free(scratch->gpu);
    scratch->~hs_scratch();
    hook_free(g_scratch, scratch);
    return HS_SUCCESS;
}

/* include/hs_gpu.h: the small-batch server of the scratch's device side (csrc/runtime.hip, hsgpu_scratch_enable_server) */
hs_error_t hs_scratch_enable_small_batch_server(hs_scratch_t *scratch, int enable, unsigned int idle_us) {
    if (!scratch || scratch->magic != 0x48534753 || enable < 0 || enable > 2) return HS_INVALID;
    if (scratch->in_use) return HS_SCRATCH_IN_USE;
    const int rv = hsgpu_scratch_enable_server(scratch->gpu, enable, idle_us);
    return rv == HSGPU_SUCCESS ? HS_SUCCESS : rv == HSGPU_NOMEM ? HS_NOMEM : HS_UNKNOWN_ERROR;
}

hs_error_t hs_scratch_small_batch_server_stats(hs_scratch_t *scratch, unsigned long long *calls, unsigned long long *launches) {
    if (!scratch || scratch->magic != 0x48534753) return HS_INVALID;
    uint64_t c = 0, l = 0;
    if (hsgpu_scratch_server_stats(scratch->gpu, &c, &l, nullptr) != HSGPU_SUCCESS) return HS_UNKNOWN_ERROR;
    if (calls) *calls = c;
    if (launches) *launches = l;
    return HS_SUCCESS;
}

/* The host confirm of a batch ("Rose-lite"): literal hits -> events, delivered in block order
 * on the calling thread. recs: sorted by (block, end), id = pattern index, as the literal
 * engine emits them. Returns 1 if some callback asked to stop (its block only), 0 if none did, -1
 * when the confirm ran out of memory (nothing is delivered then). */
static int confirm_and_deliver_impl(const hs_database *db, const char *data, const unsigned long long *off,
                                    const hsgpu_match_t *recs, size_t n, hs_batch_event_handler onEvent, void *context,
                                    const hsgpu_match_t *cs, size_t n_cs, HsConfirmPool *pool);
static int confirm_and_deliver(const hs_database *db, const char *data, const unsigned long long *off,
                               const hsgpu_match_t *recs, size_t n, hs_batch_event_handler onEvent, void *context,
                               const hsgpu_match_t *cs = nullptr, size_t n_cs = 0, HsConfirmPool *pool = nullptr) {
    try {
        return confirm_and_deliver_impl(db, data, off, recs, n, onEvent, context, cs, n_cs, pool);
    } catch (...) { /* bad_alloc while sizing the per-worker vectors */
        return -1;
    }
}
/* The confirm is bound by the latency of the misses behind every hit (offsets, then bytes), not by arithmetic: it scales with
 * the threads that have misses in flight. Up to 64 (round 5: 16). On the bench's host, 517 575 hits of config 5: 8 threads 15.4 ms,
 * 16 8.3, 32 4.1, 64 1.9-2.8, 128 1.8-3.5 with outliers (profiles/r06_confirm_threads.txt). */
static unsigned g_confirm_threads = 0;
extern "C" void hsgpu_debug_confirm_threads(unsigned n) { g_confirm_threads = n; }
static unsigned confirm_threads() {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return g_confirm_threads ? std::min(g_confirm_threads, 256u) : std::min(64u, hw);
}

/* tuning aid (include/hsgpu_tuning.h, hsgpu_debug_confirm_timing): the last large confirm of this process in seconds --
 * setup (cuts, vectors), the parallel part's wall time, the slowest and the fastest slice, the delivery loop */
static double g_confirm_timing[5];
extern "C" void hsgpu_debug_confirm_timing(double out[5]) {
    for (int i = 0; i < 5; i++) out[i] = g_confirm_timing[i];
}
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int confirm_and_deliver_impl(const hs_database *db, const char *data, const unsigned long long *off,
                                    const hsgpu_match_t *recs, size_t n, hs_batch_event_handler onEvent, void *context,
                                    const hsgpu_match_t *cs, size_t n_cs, HsConfirmPool *pool) {
    const double t_in = now_s();
    double t_par0 = t_in, t_par1 = t_in;
    /* host confirm: the events of different blocks are independent, so large batches are cut
     * into slices of whole blocks handled by worker threads; delivery stays on the calling
     * thread, in block order, as the callback contract requires */
    unsigned n_thr = 1;
    if (onEvent && n + n_cs >= 8192) n_thr = confirm_threads();
    /* more slices than threads: the threads take them as they come (round 6: with one slice per thread the slowest of 16
     * took 7.3 ms and the fastest 2.1 for the same number of hits -- a box shares its cores, and a late thread was the call) */
    const unsigned n_jobs = n_thr == 1 ? 1 : (unsigned)std::min<size_t>(n_thr * 4u, std::max<size_t>(n_thr, (n + n_cs) / 1024));
    std::vector<std::vector<Event>> ev(n_jobs);
    std::vector<std::vector<BlockRun>> runs(n_jobs);
    if (onEvent) {
        /* slices of whole blocks: cut the longer list evenly, snap to a block boundary, cut the other list at the
         * same block */
        std::vector<size_t> cut(n_jobs + 1, n), ccut(n_jobs + 1, n_cs);
        cut[0] = ccut[0] = 0;
        const bool by_cs = n_cs > n;
        const hsgpu_match_t *lead = by_cs ? cs : recs, *follow = by_cs ? recs : cs;
        const size_t n_lead = by_cs ? n_cs : n, n_follow = by_cs ? n : n_cs;
        std::vector<size_t> &lcut = by_cs ? ccut : cut, &fcut = by_cs ? cut : ccut;
        for (unsigned t = 1; t < n_jobs; t++) {
            size_t c = std::max(lcut[t - 1], n_lead * t / n_jobs);
            while (c < n_lead && c > 0 && lead[c].block == lead[c - 1].block) c++; /* snap to a block boundary */
            lcut[t] = c;
            const unsigned long long b = c < n_lead ? lead[c].block : ~0ull;
            size_t f = n_follow;
            if (c < n_lead)
                f = std::lower_bound(follow, follow + n_follow, b,
                                     [](const hsgpu_match_t &r, unsigned long long blk) { return r.block < blk; }) - follow;
            fcut[t] = std::max(f, fcut[t - 1]);
        }
        std::atomic<bool> failed{false}; /* no exception may leave a worker thread (std::terminate) or this extern "C" path */
        std::vector<double> busy(n_jobs, 0.0);
        auto work = [&](unsigned t) {
            try {
                const double t0 = now_s();
                collect_slice(db, (const unsigned char *)data, off, recs, cut[t], cut[t + 1], ev[t], runs[t], cs, ccut[t], ccut[t + 1]);
                busy[t] = now_s() - t0;
            } catch (...) {
                failed = true;
            }
        };
        t_par0 = now_s();
        if (n_thr == 1) {
            work(0);
        } else { /* the scratch's own threads and this one; without a scratch (hs_confirm_batch): threads for this call */
            HsConfirmPool local;
            HsConfirmPool *p = pool ? pool : &local;
            p->ensure(n_thr - 1);
            p->run(n_jobs, work);
        }
        if (failed) return -1;
        t_par1 = now_s();
        if (n_thr > 1) {
            g_confirm_timing[0] = t_par0 - t_in, g_confirm_timing[1] = t_par1 - t_par0;
            g_confirm_timing[2] = *std::max_element(busy.begin(), busy.end()), g_confirm_timing[3] = *std::min_element(busy.begin(), busy.end());
        }
    }
    struct Done { double t0; bool on; ~Done() { if (on) g_confirm_timing[4] = now_s() - t0; } } done{now_s(), n_thr > 1};
    int any_terminated = 0;
    for (unsigned t = 0; t < n_jobs; t++)
        for (const BlockRun &r : runs[t])
            for (size_t i = r.ev_begin; i < r.ev_end; i++) {
                const Event &e = ev[t][i];
                if (onEvent(r.block, e.id, e.from, e.to, 0, context) != 0) { /* stops THIS block only */
                    any_terminated = 1;
                    break;
                }
            }
    return any_terminated;
}

static hs_error_t scan_blocks(const hs_database_t *db, const char *data, const unsigned long long *off,
                              unsigned long long nblocks, hs_scratch_t *scratch, hs_batch_event_handler onEvent,
                              void *context);

hs_error_t hs_scan_batch(const hs_database_t *db, const char *data, const unsigned long long *off,
                         unsigned long long nblocks, unsigned int flags, hs_scratch_t *scratch,
                         hs_batch_event_handler onEvent, void *context) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    (void)flags;
    if (!scratch || !data || !off) return HS_INVALID;
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_BLOCK) return HS_DB_MODE_ERROR;
    return scan_blocks(db, data, off, nblocks, scratch, onEvent, context);
}

static hs_error_t scan_blocks(const hs_database_t *db, const char *data, const unsigned long long *off,
                              unsigned long long nblocks, hs_scratch_t *scratch, hs_batch_event_handler onEvent,
                              void *context) {
    if (scratch->magic != 0x48534753) return HS_INVALID;
    if (scratch->in_use) return HS_SCRATCH_IN_USE;
    scratch->in_use = true;
    struct Guard { hs_scratch *s; ~Guard() { s->in_use = false; } } guard{scratch};
    if (nblocks == 0) return HS_SUCCESS;
    /* large batches of a literal-only database: the chunked pipeline (csrc/runtime.hip): the host confirm and the
     * callbacks of chunk i run here while the chunks after it are copied and scanned */
    if (db->hwlm && db->cs_seqs.empty() && off[nblocks] - off[0] >= ((unsigned long long)96 << 20) &&
        hsgpu_host_is_pinned(data)) { /* (copies from pageable memory are staged by the runtime and gain nothing from chunks) */
        struct Ctx {
            const hs_database *db;
            const char *data;
            const unsigned long long *off;
            hs_batch_event_handler onEvent;
            void *context;
            HsConfirmPool *pool;
            int terminated, failed;
        } c{db, data, off, onEvent, context, &scratch->pool, 0, 0};
        const int rv = hsgpu_hwlm_exec_batch_cb(db->hwlm, scratch->gpu, (const uint8_t *)data, (const uint64_t *)off, (size_t)nblocks, 0, 0,
                                                [](const hsgpu_match_t *recs, size_t n, void *p) -> int {
                                                    Ctx *c = (Ctx *)p;
                                                    const int t = confirm_and_deliver(c->db, c->data, c->off, recs, n, c->onEvent, c->context, nullptr, 0, c->pool);
                                                    if (t < 0) {
                                                        c->failed = 1;
                                                        return 1;
                                                    }
                                                    c->terminated |= t; /* a callback's stop ends its own block only: go on */
                                                    return 0;
                                                },
                                                &c);
        if (c.failed || rv == HSGPU_NOMEM) return HS_NOMEM;
        if (rv != HSGPU_SUCCESS) return HS_UNKNOWN_ERROR;
        return c.terminated ? HS_SCAN_TERMINATED : HS_SUCCESS;
    }
    size_t cap = std::max<size_t>(std::max<size_t>(4096, scratch->recs_cap), (size_t)(off[nblocks] - off[0]) / 1024), n = 0;
    for (int attempt = 0; db->hwlm && attempt < 8; attempt++) {
        if (!scratch->reserve(cap)) return HS_NOMEM;
        int rv = hsgpu_hwlm_exec_batch(db->hwlm, scratch->gpu, (const uint8_t *)data, (const uint64_t *)off,
                                       (size_t)nblocks, 0, scratch->recs, cap, &n);
        if (rv == HSGPU_SUCCESS) break;
        if (rv != HSGPU_INSUFFICIENT_SPACE) return rv == HSGPU_NOMEM ? HS_NOMEM : HS_UNKNOWN_ERROR;
        cap = n + n / 4; /* n = the exact total */
        if (attempt == 7) return HS_UNKNOWN_ERROR;
    }
    if (db->cs_seqs.empty()) {
        const int any_terminated = confirm_and_deliver(db, data, off, scratch->recs, n, onEvent, context, nullptr, 0, &scratch->pool);
        if (any_terminated < 0) return HS_NOMEM;
        return any_terminated ? HS_SCAN_TERMINATED : HS_SUCCESS;
    }
    /* Literal-less class sequences: on the GPU from the class bitmaps, over the batch the literal scan left resident (or
     * uploaded here when the database has no literal at all). Such patterns match about once per byte ("[a-z]{3,}\d+",
     * "\w+\w+"): the records of a whole batch are K x 16 bytes per corpus byte, so the batch is delivered in RANGES of whole
     * blocks through a record buffer of bounded size -- the bitmaps are computed once, every range is one launch over its own
     * part of the corpus, its events (merged with the literal path's of the same blocks, in delivery order) go out before the
     * next range is asked for. A range that does not fit is halved. (Round 3 materialised the whole batch: advisor.) */
    int any_terminated = 0;
    try {
        const size_t room = (size_t)1 << 20; /* records per range: 16 MiB of host buffer */
        if (scratch->cs_recs.size() < room) scratch->cs_recs.resize(room);
        const hsgpu_match_t *lit = scratch->recs;
        size_t li = 0; /* literal records delivered so far (they are sorted by block) */
        unsigned long long span = std::max<unsigned long long>(1, nblocks / 64); /* blocks per range: adapts to what fits */
        int resident = db->hwlm != nullptr ? 1 : 0, have_bitmaps = 0;
        for (unsigned long long b0 = 0; b0 < nblocks;) {
            const unsigned long long b1 = std::min<unsigned long long>(nblocks, b0 + span);
            size_t n_cs = 0;
            const int rv = hsgpu_class_seq_exec_blocks(db->cs_classes.data(), (unsigned)db->cs_classes.size(), db->cs_seqs.data(),
                                                       (unsigned)db->cs_seqs.size(), scratch->gpu, (const uint8_t *)data, (const uint64_t *)off,
                                                       (size_t)nblocks, resident, have_bitmaps, (size_t)b0, (size_t)b1,
                                                       scratch->cs_recs.data(), scratch->cs_recs.size(), &n_cs);
            if (rv == HSGPU_INSUFFICIENT_SPACE) {
                resident = have_bitmaps = 1; /* (the batch and its bitmaps are on the device now whatever else happened) */
                if (b1 - b0 > 1) {
                    /* n_cs = the room this range needs: the next one is sized from the density it has just shown (three quarters of
                     * the buffer), not by blind halving -- every failed attempt is a discarded launch (advisor, round 4) */
                    const unsigned long long fit = (unsigned long long)((double)(b1 - b0) * 0.75 * (double)scratch->cs_recs.size() / (double)std::max<size_t>(1, n_cs));
                    span = std::max<unsigned long long>(1, std::min<unsigned long long>((b1 - b0) / 2, fit));
                    continue;
                }
                scratch->cs_recs.resize(n_cs + n_cs / 8 + 16); /* one block alone holds more: it gets the room it needs */
                continue;
            }
            if (rv != HSGPU_SUCCESS) return rv == HSGPU_NOMEM ? HS_NOMEM : HS_UNKNOWN_ERROR;
            resident = have_bitmaps = 1;
            size_t lj = li;
            while (lj < n && lit[lj].block < b1) lj++;
            const int t = confirm_and_deliver(db, data, off, lit + li, lj - li, onEvent, context, n_cs ? scratch->cs_recs.data() : nullptr, n_cs, &scratch->pool);
            if (t < 0) return HS_NOMEM;
            any_terminated |= t;
            li = lj;
            b0 = b1;
            /* the next range from this one's density: three quarters of the buffer, at most twice this span (a sparse range says
             * little about the next one), never doubled blindly after one sparse range into a span that has just failed */
            const unsigned long long fit = n_cs ? (unsigned long long)((double)(b1 - b0 ? b1 - b0 : 1) * 0.75 * (double)scratch->cs_recs.size() / (double)n_cs) : 2 * span;
            span = std::max<unsigned long long>(1, std::min<unsigned long long>(std::min<unsigned long long>(2 * span, fit), nblocks));
        }
    } catch (const std::bad_alloc &) {
        return HS_NOMEM;
    }
    return any_terminated ? HS_SCAN_TERMINATED : HS_SUCCESS;
}

hs_error_t hs_scan_batch_resident(const hs_database_t *db, const char *data, const unsigned long long *off,
                                  unsigned long long nblocks, const void *d_corpus, const void *d_off, hs_scratch_t *scratch,
                                  hs_batch_event_handler onEvent, void *context) {
    db = resolve_db(db);
    if (!scratch || !data || !off || !d_off) return HS_INVALID;
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_BLOCK) return HS_DB_MODE_ERROR;
    if (scratch->magic != 0x48534753 || !db->cs_seqs.empty() || off[0] != 0) return HS_INVALID;
    if (scratch->in_use) return HS_SCRATCH_IN_USE;
    scratch->in_use = true;
    struct Guard { hs_scratch *s; ~Guard() { s->in_use = false; } } guard{scratch};
    if (nblocks == 0 || !db->hwlm) return HS_SUCCESS;
    const hsgpu_match_t *recs = nullptr;
    size_t n = 0;
    const int rv = hsgpu_hwlm_exec_resident(db->hwlm, scratch->gpu, d_corpus, off[nblocks], d_off, nblocks, 0, &recs, &n);
    if (rv != HSGPU_SUCCESS) return rv == HSGPU_NOMEM ? HS_NOMEM : rv == HSGPU_INVALID ? HS_INVALID : HS_UNKNOWN_ERROR;
    const int t = confirm_and_deliver(db, data, off, recs, n, onEvent, context, nullptr, 0, &scratch->pool);
    if (t < 0) return HS_NOMEM;
    return t ? HS_SCAN_TERMINATED : HS_SUCCESS;
}

/* the literal the GPU matcher holds for one branch (hs_gpu.h) */
hs_error_t hs_database_literal(const hs_database_t *db, unsigned int index, const char **bytes, size_t *len,
                               int *nocase, unsigned int *id) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || db->magic != 0x48534744 || index >= db->pats.size()) return HS_INVALID;
    const Pattern &p = db->pats[index];
    const size_t n = std::min<size_t>(p.lit.size(), 8);
    if (bytes) *bytes = p.lit.data() + p.lit.size() - n;
    if (len) *len = n;
    if (nocase) *nocase = p.nocase;
    if (id) *id = p.id;
    return HS_SUCCESS;
}

/* Extension: the host confirm alone, for callers that bring their own literal hits (another
 * literal engine, a replayed capture, a test): recs as hsgpu_hwlm_exec_batch would return them
 * for this database's literals -- sorted by (block, end), id = index of the pattern in
 * compile order. No device is touched. */
hs_error_t hs_confirm_batch(const hs_database_t *db, const char *data, const unsigned long long *off,
                            unsigned long long nblocks, const void *records, unsigned long long n_records,
                            hs_batch_event_handler onEvent, void *context) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!db || db->magic != 0x48534744 || !data || !off || (n_records && !records)) return HS_INVALID;
    const hsgpu_match_t *recs = (const hsgpu_match_t *)records;
    /* the caller's records are checked before anything follows them into the batch -- on the threads that confirm them
     * afterwards (round 6: a serial walk was a quarter of a large call's time: every record reads two offsets) */
    auto valid = [&](unsigned long long lo, unsigned long long hi) -> bool {
        for (unsigned long long i = lo; i < hi; i++) {
            if (recs[i].block >= nblocks || recs[i].id >= db->pats.size()) return false;
            if (i && (recs[i].block < recs[i - 1].block ||
                      (recs[i].block == recs[i - 1].block && recs[i].end < recs[i - 1].end)))
                return false;
            if (recs[i].end >= off[recs[i].block + 1] - off[recs[i].block]) return false;
        }
        return true;
    };
    if (n_records < 8192) {
        if (!valid(0, n_records)) return HS_INVALID;
        const int rv = confirm_and_deliver(db, data, off, recs, (size_t)n_records, onEvent, context);
        return rv < 0 ? HS_NOMEM : (rv ? HS_SCAN_TERMINATED : HS_SUCCESS);
    }
    try {
        HsConfirmPool pool; /* (no scratch in this call: threads of its own, for the check and for the confirm) */
        const unsigned n_thr = confirm_threads();
        pool.ensure(n_thr - 1);
        std::atomic<bool> bad{false};
        const std::function<void(unsigned)> check = [&](unsigned t) {
            if (!valid(n_records * t / n_thr, n_records * (t + 1) / n_thr)) bad = true;
        };
        pool.run(n_thr, check);
        if (bad) return HS_INVALID;
        const int rv = confirm_and_deliver(db, data, off, recs, (size_t)n_records, onEvent, context, nullptr, 0, &pool);
        return rv < 0 ? HS_NOMEM : (rv ? HS_SCAN_TERMINATED : HS_SUCCESS);
    } catch (...) {
        return HS_NOMEM;
    }
}

hs_error_t hs_scan(const hs_database_t *db, const char *data, unsigned int length, unsigned int flags,
                   hs_scratch_t *scratch, match_event_handler onEvent, void *context) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    if (!scratch || !data) return HS_INVALID; /* src/runtime.c:320-322 */
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_BLOCK) return HS_DB_MODE_ERROR; /* src/runtime.c:334-336 */
    struct Ctx { match_event_handler cb; void *user; } c{onEvent, context};
    const unsigned long long off[2] = {0, length};
    if (length < db->min_width) { /* src/runtime.c:346-350 */
        if (scratch->magic != 0x48534753) return HS_INVALID;
        return scratch->in_use ? HS_SCRATCH_IN_USE : HS_SUCCESS;
    }
    return hs_scan_batch(db, data, off, 1, flags, scratch,
                         onEvent ? +[](unsigned long long, unsigned id, unsigned long long from, unsigned long long to,
                                       unsigned fl, void *cc) { return ((Ctx *)cc)->cb(id, from, to, fl, ((Ctx *)cc)->user); }
                                 : (hs_batch_event_handler) nullptr,
                         &c);
}

/* src/runtime.c:1106-1174 hs_scan_vector: the segments are one logical buffer (offsets run
 * through them). Here they are gathered into one block and scanned as such -- on this engine a
 * vectored scan IS a block scan of the concatenation, so the semantics hold by construction. */
hs_error_t hs_scan_vector(const hs_database_t *db, const char *const *data, const unsigned int *length,
                          unsigned int count, unsigned int flags, hs_scratch_t *scratch, match_event_handler onEvent,
                          void *context) {
    db = resolve_db(db); /* a database placed by hs_deserialize_database_at */
    (void)flags;
    if (!scratch || !data || !length) return HS_INVALID; /* src/runtime.c:1113-1115 */
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_VECTORED) return HS_DB_MODE_ERROR;
    if (scratch->magic != 0x48534753) return HS_INVALID;
    unsigned long long total = 0;
    for (unsigned i = 0; i < count; i++) {
        if (length[i] && !data[i]) return HS_INVALID;
        total += length[i];
    }
    if (total < db->min_width) return scratch->in_use ? HS_SCRATCH_IN_USE : HS_SUCCESS;
    std::string joined;
    joined.reserve((size_t)total);
    for (unsigned i = 0; i < count; i++) joined.append(data[i], length[i]);
    struct Ctx { match_event_handler cb; void *user; } c{onEvent, context};
    const unsigned long long off[2] = {0, total};
    return scan_blocks(db, joined.data(), off, 1, scratch,
                       onEvent ? +[](unsigned long long, unsigned id, unsigned long long from, unsigned long long to,
                                     unsigned fl, void *cc) { return ((Ctx *)cc)->cb(id, from, to, fl, ((Ctx *)cc)->user); }
                               : (hs_batch_event_handler) nullptr,
                       &c);
}

int hs_batch_count_handler(unsigned long long, unsigned int, unsigned long long, unsigned long long, unsigned int,
                           void *context) {
    if (context) ++*(unsigned long long *)context;
    return 0;
}

const char *hs_version(void) { return "5.4.2-hsgpu-gfx950"; }

hs_error_t hs_valid_platform(void) {
    hsgpu_scratch_t *s = nullptr;
    if (hsgpu_scratch_alloc(&s, -1) != HSGPU_SUCCESS) return HS_ARCH_ERROR;
    hsgpu_scratch_free(s);
    return HS_SUCCESS;
}

} // extern "C"
