/*
 * class_seq.hip -- class-sequence patterns A{m,}B{n,} ("[a-z]{3,}\d+" style) evaluated on the GPU from the
 * class-membership bitmaps that class_scan.hip produces: the consumer of the accelerators' answer.
 *
 * In the reference such a pattern has no literal for HWLM to find; it is compiled to an NFA / DFA whose idle
 * states are skipped with the class accelerators (run_accel, src/nfa/accel.c:35-146; callers
 * src/nfa/limex_accel.c:49-74, src/nfa/mcclellan.c:92-120) and every match end is reported through the
 * engine's callback. Here the class bitmaps make the whole pattern bit-parallel (one bit per corpus byte, 64
 * bytes per machine word), so the engine is three shift-and steps and one add:
 *
 *   qa = A & ~start                       A, and the byte before it is in the same block
 *   R_m(i)  = A(i-m+1) & qa(i-m+2 .. i)   m members of A end at i inside one block (run masks by doubling)
 *   G(s)    = R_m(s-1) & ~start(s)        a match of A{m,} may end right before s
 *   X(e)    = G(e-n+1) & B(e-n+1) & qb(e-n+2 .. e)       the mandatory B{n} ends at e
 *   Y       = (((qb + X) ^ qb) & qb) | X  X carried upwards through the rest of its run of B (the add)
 *
 * Y(e) <=> the pattern matches ending at byte e of its block: exactly the set of `to - 1` offsets hs_scan
 * reports for the expression (all matches, no start of match; unit/hyperscan/behaviour.cpp documents the
 * semantics). A and B may overlap: Y holds e iff SOME split point exists, as the regex semantics demand.
 *
 * Mapping: one lane per PATTERN, one wavefront per (group of 64 patterns, contiguous share of the corpus);
 * every lane walks its share word by word with its own classes, repeat counts and carry state, so nothing
 * crosses lanes and a pattern's count is one register. A share is made of whole blocks (the blocks that START
 * inside its byte range), found by bisection of the offsets: no state enters a share.
 *
 * Match density is a property of the pattern set, not of the engine: 256 class-heavy patterns over text report
 * several matches per corpus byte, more records than any buffer holds. The entry point therefore always COUNTS
 * (per pattern, the whole corpus: matches/s as hsbench reports it) and EMITS 16-byte records only for the byte
 * range the caller names (parity checks, or a caller that wants the ends for a subset of the blocks).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
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

constexpr int SEQ_THREADS = 256;
constexpr uint32_t SEQ_HEADER = 32768; /* work area: pattern array, bitmap pointers, (class, repeat) pairs in front of the start bitmap */
constexpr uint32_t SEQ_BATCH = 8;     /* words per hand-over of G between the pair lanes and the pattern lanes */

struct SeqArgs {
    const hsgpu_class_seq_t *seqs;
    const uint16_t *const *bitmaps; /* [n_classes] device pointers, (total + 15) / 16 * 2 bytes each */
    const uint64_t *starts;         /* bit i <=> a block starts at corpus byte i */
    const uint64_t *off;
    uint64_t nblocks, total;
    uint32_t n_seqs, n_groups, n_shares, any_n;
    uint64_t share_bytes;
    uint64_t emit_lo, emit_hi;
    unsigned long long *counts;
    hsgpu_match_t *out;
    uint64_t cap;
    unsigned long long *count;
    /* the patterns of a workgroup (256 of them) share their distinct (A, m) pairs: G is computed once per pair */
    const uint16_t *pair_of;   /* [n_seqs] index of the pattern's pair inside its group of 256 */
    const uint16_t *pairs;     /* [n_wgroups][256]: class | (m - 1) << 8 */
    const uint16_t *n_pairs;   /* [n_wgroups] */
    uint32_t n_wgroups;
};

__global__ void seq_starts_kernel(const uint64_t *off, uint64_t nblocks, uint64_t total, uint32_t *starts32) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint64_t o = off[b];
    if (o < total) atomicOr(&starts32[o >> 5], 1u << (o & 31));
}

/* first index in off[0 .. n] (n + 1 ascending entries) whose value is >= x; n + 1 when none */
__device__ __forceinline__ uint64_t lower_bound_off(const uint64_t *off, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n + 1;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

/* 64 membership bits of word w; the buffer is a whole number of 16-bit pieces, not of words */
__device__ __forceinline__ uint64_t load_word(const uint16_t *bm, uint64_t w, uint64_t n16) {
    if ((w + 1) * 4 <= n16 && (((uintptr_t)bm) & 7) == 0) return ((const uint64_t *)bm)[w];
    uint64_t v = 0;
    for (uint32_t k = 0; k < 4; k++)
        if (w * 4 + k < n16) v |= (uint64_t)bm[w * 4 + k] << (16 * k);
    return v;
}

/* (cur:prev) << k, the part that lands in cur; k in 0..63 */
__device__ __forceinline__ uint64_t shl2(uint64_t cur, uint64_t prev, uint32_t k) {
    return k ? (cur << k) | (prev >> (64 - k)) : cur;
}

/* T_k(i) = q(i-k+1 .. i) all set, k in 0..15, from the run masks of length 1, 2, 4, 8 of this word and the word
 * before it (t*[0] = this word, t*[1] = the previous one) */
struct RunMasks {
    uint64_t t1[2], t2[2], t4[2], t8[2];
};
__device__ __forceinline__ void advance_runs(RunMasks &r, uint64_t q) {
    r.t1[1] = r.t1[0], r.t2[1] = r.t2[0], r.t4[1] = r.t4[0], r.t8[1] = r.t8[0];
    r.t1[0] = q;
    r.t2[0] = r.t1[0] & shl2(r.t1[0], r.t1[1], 1);
    r.t4[0] = r.t2[0] & shl2(r.t2[0], r.t2[1], 2);
    r.t8[0] = r.t4[0] & shl2(r.t4[0], r.t4[1], 4);
}
__device__ __forceinline__ uint64_t run_of(const RunMasks &r, uint32_t k) {
    uint64_t acc = ~0ull;
    uint32_t ofs = 0;
    if (k & 1) acc &= r.t1[0], ofs = 1; /* ofs 0: the word itself */
    if (k & 2) acc &= shl2(r.t2[0], r.t2[1], ofs), ofs += 2;
    if (k & 4) acc &= shl2(r.t4[0], r.t4[1], ofs), ofs += 4;
    if (k & 8) acc &= shl2(r.t8[0], r.t8[1], ofs);
    return acc;
}

__global__ __launch_bounds__(SEQ_THREADS) void class_seq_kernel(SeqArgs args) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (SEQ_THREADS / 64) + (threadIdx.x >> 6);
    const uint32_t group = wave % args.n_groups, share = wave / args.n_groups;
    if (share >= args.n_shares) return;
    const uint32_t p = group * 64 + lane;
    const bool active = p < args.n_seqs;
    const hsgpu_class_seq_t P = args.seqs[active ? p : 0];
    const uint16_t *pa = args.bitmaps[P.a], *pb = args.bitmaps[P.b];
    const uint32_t km = (uint32_t)P.m - 1, kn = (uint32_t)P.n - 1;
    const uint64_t n16 = (args.total + 15) / 16;

    /* the share: the blocks that start inside [lo_b, hi_b) */
    const uint64_t lo_b = (uint64_t)share * args.share_bytes, hi_b = min(args.total, lo_b + args.share_bytes);
    const uint64_t b_lo = lower_bound_off(args.off, args.nblocks, lo_b), b_hi = lower_bound_off(args.off, args.nblocks, hi_b);
    if (b_lo >= b_hi || b_lo >= args.nblocks) return;
    const uint64_t s0 = args.off[b_lo], s1 = args.off[min(b_hi, args.nblocks)];
    if (s0 >= s1) return;
    const uint64_t w0 = s0 >> 6, w1 = (s1 - 1) >> 6;

    RunMasks ra, rb;
    memset(&ra, 0, sizeof(ra));
    memset(&rb, 0, sizeof(rb));
    uint64_t prev_a = 0, prev_b = 0, prev_r = 0, prev_g = 0, cin = 0;
    unsigned long long n_match = 0;
    uint64_t na = load_word(pa, w0, n16), nb = load_word(pb, w0, n16), ne = args.starts[w0];
    for (uint64_t w = w0; w <= w1; w++) {
        uint64_t a = na, b = nb;
        const uint64_t e = ne;
        if (w < w1) na = load_word(pa, w + 1, n16), nb = load_word(pb, w + 1, n16), ne = args.starts[w + 1];
        if ((w + 1) * 64 > args.total) { /* bits past the corpus */
            const uint64_t valid = ~0ull >> (64 - (args.total - w * 64));
            a &= valid, b &= valid;
        }
        const uint64_t nst = ~e;
        advance_runs(ra, a & nst);
        const uint64_t r = shl2(a, prev_a, km) & run_of(ra, km); /* R_m */
        const uint64_t g = shl2(r, prev_r, 1) & nst;             /* G */
        const uint64_t qb = b & nst;
        uint64_t x;
        if (args.any_n) { /* some pattern of the call has n > 1 */
            advance_runs(rb, qb);
            x = shl2(g, prev_g, kn) & shl2(b, prev_b, kn) & run_of(rb, kn);
        } else {
            x = g & b;
        }
        const uint64_t s_1 = qb + x, c1 = s_1 < qb ? 1 : 0, s_2 = s_1 + cin, c2 = s_2 < s_1 ? 1 : 0;
        uint64_t y = ((s_2 ^ qb) & qb) | x;
        cin = c1 | c2;
        prev_a = a, prev_b = b, prev_r = r, prev_g = g;
        if (w == w0) y &= ~0ull << (s0 & 63);
        if (w == w1) y &= ~0ull >> (63 - ((s1 - 1) & 63));
        if (!active) y = 0;
        n_match += (unsigned)__popcll(y);
        const uint64_t base = w * 64;
        if (y && base < args.emit_hi && base + 64 > args.emit_lo) {
            while (y) {
                const uint32_t j = __builtin_ctzll(y);
                y &= y - 1;
                const uint64_t pos = base + j;
                if (pos < args.emit_lo || pos >= args.emit_hi) continue;
                const unsigned long long at = atomicAdd(args.count, 1ull);
                if (at >= args.cap) continue;
                /* the block of pos: the last one starting at or before it (empty blocks share an offset) */
                const uint64_t blk = lower_bound_off(args.off, args.nblocks, pos + 1) - 1;
                hsgpu_match_t rec;
                rec.block = (uint32_t)blk;
                rec.end = (uint32_t)(pos - args.off[blk]);
                rec.id = P.id;
                rec.lit = p;
                args.out[at] = rec;
            }
        }
    }
    if (active && n_match) atomicAdd(&args.counts[p], n_match);
}

/* The same with the A side shared: a workgroup of 4 wavefronts takes 256 patterns and one share of the corpus. The distinct
 * (A, m) pairs of those patterns (72 for the bench's 256) are computed once, by the first n_pairs threads, SEQ_BATCH words at a
 * time into LDS; then every pattern lane reads the G of its pair and does only its own B side. 1.5x fewer vector instructions
 * than one lane per pattern doing everything (the kernel is 95 % VALU-busy, so instructions are its time). */
__global__ __launch_bounds__(SEQ_THREADS) void class_seq_shared_kernel(SeqArgs args) {
    __shared__ uint64_t gbuf[SEQ_BATCH][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t wg = blockIdx.x % args.n_wgroups, share = blockIdx.x / args.n_wgroups;
    if (share >= args.n_shares) return;
    const uint32_t p = wg * 256 + tid;
    const bool active = p < args.n_seqs;
    const hsgpu_class_seq_t P = args.seqs[active ? p : 0];
    const uint32_t my_pair = active ? args.pair_of[p] : 0;
    const uint32_t n_pairs = args.n_pairs[wg];
    const bool pair_lane = tid < n_pairs;
    const uint32_t pw = args.pairs[wg * 256 + (pair_lane ? tid : 0)];
    const uint16_t *pa = args.bitmaps[pw & 0xff], *pb = args.bitmaps[P.b];
    const uint32_t km = pw >> 8, kn = (uint32_t)P.n - 1;
    const uint64_t n16 = (args.total + 15) / 16;

    const uint64_t lo_b = (uint64_t)share * args.share_bytes, hi_b = min(args.total, lo_b + args.share_bytes);
    const uint64_t b_lo = lower_bound_off(args.off, args.nblocks, lo_b), b_hi = lower_bound_off(args.off, args.nblocks, hi_b);
    if (b_lo >= b_hi || b_lo >= args.nblocks) return; /* (uniform: the whole workgroup) */
    const uint64_t s0 = args.off[b_lo], s1 = args.off[min(b_hi, args.nblocks)];
    if (s0 >= s1) return;
    const uint64_t w0 = s0 >> 6, w1 = (s1 - 1) >> 6;

    RunMasks ra, rb;
    memset(&ra, 0, sizeof(ra));
    memset(&rb, 0, sizeof(rb));
    uint64_t prev_a = 0, prev_b = 0, prev_r = 0, prev_g = 0, cin = 0;
    unsigned long long n_match = 0;
    for (uint64_t wb = w0; wb <= w1; wb += SEQ_BATCH) {
        const uint32_t nw = (uint32_t)min((uint64_t)SEQ_BATCH, w1 - wb + 1);
        if (pair_lane) { /* G of this thread's (A, m) pair for the batch's words */
            for (uint32_t k = 0; k < nw; k++) {
                const uint64_t w = wb + k;
                uint64_t a = load_word(pa, w, n16);
                const uint64_t e = args.starts[w];
                if ((w + 1) * 64 > args.total) a &= ~0ull >> (64 - (args.total - w * 64));
                const uint64_t nst = ~e;
                advance_runs(ra, a & nst);
                const uint64_t r = shl2(a, prev_a, km) & run_of(ra, km);
                gbuf[k][tid] = shl2(r, prev_r, 1) & nst;
                prev_a = a, prev_r = r;
            }
        }
        __syncthreads();
        for (uint32_t k = 0; k < nw; k++) {
            const uint64_t w = wb + k;
            uint64_t b = load_word(pb, w, n16);
            const uint64_t e = args.starts[w];
            if ((w + 1) * 64 > args.total) b &= ~0ull >> (64 - (args.total - w * 64));
            const uint64_t g = gbuf[k][my_pair];
            const uint64_t qb = b & ~e;
            uint64_t x;
            if (args.any_n) {
                advance_runs(rb, qb);
                x = shl2(g, prev_g, kn) & shl2(b, prev_b, kn) & run_of(rb, kn);
            } else {
                x = g & b;
            }
            const uint64_t s_1 = qb + x, c1 = s_1 < qb ? 1 : 0, s_2 = s_1 + cin, c2 = s_2 < s_1 ? 1 : 0;
            uint64_t y = ((s_2 ^ qb) & qb) | x;
            cin = c1 | c2;
            prev_b = b, prev_g = g;
            if (w == w0) y &= ~0ull << (s0 & 63);
            if (w == w1) y &= ~0ull >> (63 - ((s1 - 1) & 63));
            if (!active) y = 0;
            n_match += (unsigned)__popcll(y);
            const uint64_t base = w * 64;
            if (y && base < args.emit_hi && base + 64 > args.emit_lo) {
                while (y) {
                    const uint32_t j = __builtin_ctzll(y);
                    y &= y - 1;
                    const uint64_t pos = base + j;
                    if (pos < args.emit_lo || pos >= args.emit_hi) continue;
                    const unsigned long long at = atomicAdd(args.count, 1ull);
                    if (at >= args.cap) continue;
                    const uint64_t blk = lower_bound_off(args.off, args.nblocks, pos + 1) - 1;
                    hsgpu_match_t rec;
                    rec.block = (uint32_t)blk;
                    rec.end = (uint32_t)(pos - args.off[blk]);
                    rec.id = P.id;
                    rec.lit = p;
                    args.out[at] = rec;
                }
            }
        }
        __syncthreads(); /* the next batch overwrites gbuf */
    }
    if (active && n_match) atomicAdd(&args.counts[p], n_match);
}

} // namespace

extern "C" size_t hsgpu_class_seq_work_bytes(uint64_t total_bytes) { return SEQ_HEADER + ((total_bytes + 63) / 64) * 8 + 8; }

extern "C" int hsgpu_class_seq_scan_dev(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps,
                                        unsigned n_classes, uint64_t total_bytes, const void *d_off, uint64_t nblocks,
                                        uint64_t emit_lo, uint64_t emit_hi, void *d_counts, void *d_out, uint64_t cap,
                                        void *d_count, void *d_work, size_t work_bytes, void *stream) {
    if (!seqs || !n_seqs || !d_bitmaps || !n_classes || !d_off || !d_counts || !d_count || !d_work || (cap && !d_out))
        return HSGPU_INVALID;
    if (n_seqs > HSGPU_SEQ_MAX || n_classes > 255) {
        hsgpu_set_error("at most %u class sequences and 255 classes per call", HSGPU_SEQ_MAX);
        return HSGPU_INVALID;
    }
    if (work_bytes < hsgpu_class_seq_work_bytes(total_bytes) || ((uintptr_t)d_work & 15)) {
        hsgpu_set_error("class-sequence work area: %zu bytes, 16-byte aligned (hsgpu_class_seq_work_bytes)",
                        hsgpu_class_seq_work_bytes(total_bytes));
        return HSGPU_INVALID;
    }
    bool any_n = false;
    for (unsigned i = 0; i < n_seqs; i++) {
        if (seqs[i].a >= n_classes || seqs[i].b >= n_classes || seqs[i].m < 1 || seqs[i].m > HSGPU_SEQ_MAX_REPEAT ||
            seqs[i].n < 1 || seqs[i].n > HSGPU_SEQ_MAX_REPEAT) {
            hsgpu_set_error("class sequence %u: class index or repeat count out of range (1..%u)", i, HSGPU_SEQ_MAX_REPEAT);
            return HSGPU_INVALID;
        }
        any_n |= seqs[i].n > 1;
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)n_seqs * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st));
    if (!total_bytes || !nblocks) return HSGPU_SUCCESS;
    /* work area: patterns | bitmap pointers | (class, repeat) pairs per group of 256 patterns | block-start bitmap */
    uint8_t *w = (uint8_t *)d_work;
    const size_t seq_bytes = (size_t)n_seqs * sizeof(hsgpu_class_seq_t), ptr_ofs = (seq_bytes + 15) & ~(size_t)15;
    const uint32_t n_wgroups = (n_seqs + 255) / 256;
    const size_t pairof_ofs = (ptr_ofs + (size_t)n_classes * sizeof(void *) + 15) & ~(size_t)15;
    const size_t pairs_ofs = pairof_ofs + (((size_t)n_seqs * 2 + 15) & ~(size_t)15);
    const size_t npairs_ofs = pairs_ofs + (size_t)n_wgroups * 256 * 2;
    if (npairs_ofs + (size_t)n_wgroups * 2 > SEQ_HEADER) return HSGPU_INVALID;
    std::vector<uint16_t> pair_of(n_seqs), pairs((size_t)n_wgroups * 256, 0), n_pairs(n_wgroups, 0);
    for (unsigned i = 0; i < n_seqs; i++) {
        const unsigned g = i / 256;
        const uint16_t key = (uint16_t)(seqs[i].a | (unsigned)(seqs[i].m - 1) << 8);
        unsigned k = 0;
        while (k < n_pairs[g] && pairs[(size_t)g * 256 + k] != key) k++;
        if (k == n_pairs[g]) pairs[(size_t)g * 256 + n_pairs[g]++] = key;
        pair_of[i] = (uint16_t)k;
    }
    /* (pageable sources: the runtime stages them before the call returns) */
    HIP_TRY(hipMemcpyAsync(w, seqs, seq_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w + ptr_ofs, d_bitmaps, (size_t)n_classes * sizeof(void *), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w + pairof_ofs, pair_of.data(), pair_of.size() * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w + pairs_ofs, pairs.data(), pairs.size() * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w + npairs_ofs, n_pairs.data(), n_pairs.size() * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st)); /* the staging vectors go out of scope */
    const uint64_t n_words = (total_bytes + 63) / 64;
    uint64_t *starts = (uint64_t *)(w + SEQ_HEADER);
    HIP_TRY(hipMemsetAsync(starts, 0, n_words * 8 + 8, st));
    hipLaunchKernelGGL(seq_starts_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, (const uint64_t *)d_off,
                       nblocks, total_bytes, (uint32_t *)starts);
    SeqArgs a;
    a.seqs = (const hsgpu_class_seq_t *)w;
    a.bitmaps = (const uint16_t *const *)(w + ptr_ofs);
    a.starts = starts;
    a.off = (const uint64_t *)d_off;
    a.nblocks = nblocks;
    a.total = total_bytes;
    a.n_seqs = n_seqs;
    a.n_groups = (n_seqs + 63) / 64;
    a.any_n = any_n ? 1u : 0u;
    /* ~16 K wavefronts in all, a share of at least 16 KiB (whole blocks each: a share smaller than the blocks buys
     * nothing) */
    const uint64_t want = std::max<uint64_t>(1, 16384 / a.n_groups);
    uint64_t share = std::max<uint64_t>(16384, (total_bytes + want - 1) / want);
    share = (share + 63) & ~63ull;
    a.share_bytes = share;
    a.n_shares = (uint32_t)((total_bytes + share - 1) / share);
    a.emit_lo = emit_lo;
    a.emit_hi = std::min(emit_hi, total_bytes);
    a.counts = (unsigned long long *)d_counts;
    a.out = (hsgpu_match_t *)d_out;
    a.cap = cap;
    a.count = (unsigned long long *)d_count;
    a.pair_of = (const uint16_t *)(w + pairof_ofs);
    a.pairs = (const uint16_t *)(w + pairs_ofs);
    a.n_pairs = (const uint16_t *)(w + npairs_ofs);
    a.n_wgroups = n_wgroups;
    if (n_seqs > 64) { /* several patterns per (A, m) pair are likely: share the A side inside a workgroup */
        hipLaunchKernelGGL(class_seq_shared_kernel, dim3(n_wgroups * a.n_shares), dim3(SEQ_THREADS), 0, st, a);
    } else {
        const uint64_t waves = (uint64_t)a.n_groups * a.n_shares;
        hipLaunchKernelGGL(class_seq_kernel, dim3((unsigned)((waves + SEQ_THREADS / 64 - 1) / (SEQ_THREADS / 64))),
                           dim3(SEQ_THREADS), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return HSGPU_SUCCESS;
}
