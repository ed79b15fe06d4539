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
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
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


/* ---- round 4: one lane per 64-byte WORD, the patterns looped in scalar registers ------------------------------------
 * The kernels above give a lane a pattern: every shift amount, class pointer and repeat count is a per-lane value, the
 * 64-bit shifts and the add are emulated per lane, ~130 vector instructions per word and pattern. Here a wavefront takes
 * a TILE of 63 consecutive words (lane 0 repeats the word before the tile: its upper bits feed lane 1's shifts, nothing
 * of it is reported) and walks a little program that the host sorted by (A, m, B, n):
 *   class A   the run masks of qa = A & ~start by doubling (t1, t2, t4, t8) and each one's copy from the lane below
 *             (ds_bpermute), once per distinct class
 *   pair      R_m and G for one (A, m): the shift amounts are wave-uniform, a 64-bit funnel shift across two words is two
 *             v_alignbit on 32-bit halves; once per distinct pair (72 of the bench's 256 patterns)
 *   pattern   x = G & qb(B) (qb of every class sits in LDS), s = qb + x with the carry BETWEEN the words of the tile by
 *             carry-lookahead on the wavefront: the add's carry-out mask (generate) and "s is all ones" (propagate) are
 *             64-bit lane masks in scalar registers, one scalar add gives every lane its carry-in ((P|G) + G ^ P), one
 *             v_addc applies it; Y = bfi(s, x, qb); the carry out of the tile's last word waits in a lane of a register
 *             for the pattern's next tile. Two patterns' counts are packed into one DPP reduction.
 * ~18 vector instructions per word and pattern + ~10 for the shared side, all shift amounts scalar. (profiles/r04_class_seq*.txt) */
constexpr uint32_t TILE_WORDS = 63;
constexpr uint32_t TILE_MAX_CLASSES = 32;
constexpr uint32_t QB_STRIDE = 65; /* a class' qb words in LDS: [0] = 0 (the halo lane adds nothing to a carry chain), [1..63] the tile's words, [64] the halo's true word */

enum : uint8_t { ENTRY_ONE = 0, ENTRY_TWO = 1 }; /* an entry's kind bit */
struct TileOp { /* a word of the program: 8 bytes, read with one scalar load */
    uint8_t kind, k;     /* class header: k = its pairs | pair header: k = m - 1 | entry: kind = ENTRY_ONE / ENTRY_TWO, k = n - 1 (ENTRY_ONE) */
    uint16_t ofs1, ofs2; /* entries: the byte offsets of the B classes' qb words in the wavefront's LDS (class * QB_STRIDE * 8); headers: 0 */
    uint16_t cls;        /* class header: A | pair header: its entries | ENTRY_ONE: B */
};
__host__ __device__ inline size_t tile_lds_per_wave(uint32_t n_classes, uint32_t n_pats) { /* qb words | carries | counts */
    return (size_t)n_classes * (QB_STRIDE * 8) + (size_t)(n_pats / 64 + 2) * 8 + (((((size_t)n_pats + 3) / 2) * 4 + 15) & ~(size_t)15); /* counts: two 16-bit fields per word */
}
__device__ __forceinline__ unsigned long long rfl64u(unsigned long long v) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32 |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

struct TileArgs {
    const TileOp *ops;
    const uint16_t *order; /* [n_pats]: the caller's index of the program's j-th pattern */
    uint32_t n_groups, n_pats, n_classes, n_shares; /* n_groups: class headers in the program */
    uint32_t only_emit; /* hsgpu_class_seq_emit_dev: only the shares whose blocks can end in [emit_lo, emit_hi) are walked (no counts) */
    const hsgpu_class_seq_t *seqs;
    const uint64_t *const *bitmaps; /* 8-byte aligned */
    const uint64_t *starts;
    const uint64_t *off;
    uint64_t nblocks, total, share_bytes, emit_lo, emit_hi, cap;
    unsigned long long *counts, *count;
    hsgpu_match_t *out;
};

struct W2 {
    uint32_t lo, hi;
};
/* a wave-uniform 64-bit value for a vector use, through an explicit move: a scalar value carried round a loop that has ONE vector user
 * (a store, a per-lane shift) is otherwise kept in vector registers altogether and read back with v_readfirstlane at every scalar use */
__device__ __forceinline__ unsigned long long to_vector(unsigned long long v) {
    uint32_t lo, hi;
    asm("v_mov_b32 %0, %1" : "=v"(lo) : "s"((uint32_t)v));
    asm("v_mov_b32 %0, %1" : "=v"(hi) : "s"((uint32_t)(v >> 32)));
    return (unsigned long long)hi << 32 | lo;
}
__device__ __forceinline__ W2 w2(uint64_t v) { return {(uint32_t)v, (uint32_t)(v >> 32)}; }
__device__ __forceinline__ W2 operator&(W2 a, W2 b) { return {a.lo & b.lo, a.hi & b.hi}; }
/* the value the lane below holds (lane 0: whatever; it is the halo) */
__device__ __forceinline__ W2 from_below(W2 v, uint32_t addr_below) {
    return {(uint32_t)__builtin_amdgcn_ds_bpermute((int)addr_below, (int)v.lo), (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr_below, (int)v.hi)};
}
/* (cur : prev) << k, the part that lands in cur; k wave-uniform, 0 .. 31 */
__device__ __forceinline__ W2 shl2u(W2 cur, W2 prev, uint32_t k) {
    if (k == 0) return cur;
    return {__builtin_amdgcn_alignbit(cur.lo, prev.hi, 32u - k), __builtin_amdgcn_alignbit(cur.hi, cur.lo, 32u - k)};
}
struct Runs { /* run masks of length 1, 2, 4, 8 of this lane's word and of the lane below */
    W2 t1, t2, t4, t8, p1, p2, p4, p8;
};
__device__ __forceinline__ void make_runs(Runs &r, W2 q, uint32_t below) {
    r.t1 = q, r.p1 = from_below(r.t1, below);
    r.t2 = r.t1 & shl2u(r.t1, r.p1, 1), r.p2 = from_below(r.t2, below);
    r.t4 = r.t2 & shl2u(r.t2, r.p2, 2), r.p4 = from_below(r.t4, below);
    r.t8 = r.t4 & shl2u(r.t4, r.p4, 4), r.p8 = from_below(r.t8, below);
}
/* q(i - k + 1 .. i) all set, k wave-uniform in 0 .. 15 */
__device__ __forceinline__ W2 run_of_u(const Runs &r, uint32_t k) {
    W2 acc = {~0u, ~0u};
    uint32_t ofs = 0;
    if (k & 1) acc = acc & r.t1, ofs = 1;
    if (k & 2) acc = acc & shl2u(r.t2, r.p2, ofs), ofs += 2;
    if (k & 4) acc = acc & shl2u(r.t4, r.p4, ofs), ofs += 4;
    if (k & 8) acc = acc & shl2u(r.t8, r.p8, ofs);
    return acc;
}
/* wave64 sum, the total in lane 63 (row_shr 1 2 4 8 inside the rows of 16, then row_bcast 15 and 31 across them) */
__device__ __forceinline__ uint32_t wave_sum_to_63(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

/* EMIT = false: the counting pass (no record can be asked for: the emit range is empty) without the record path's registers and branches */
template <bool EMIT>
__global__ __launch_bounds__(SEQ_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void class_seq_tile_kernel(TileArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tile_lds[];
    const uint32_t lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const uint32_t share = blockIdx.x * (SEQ_THREADS / 64) + wave_in_wg;
    /* per wavefront: qb of every class for the tile's words [n_classes][64], the patterns' counts [n_pats] */
    const size_t per_wave = tile_lds_per_wave(args.n_classes, args.n_pats);
    uint64_t *qb_lds = (uint64_t *)(tile_lds + wave_in_wg * per_wave);
    unsigned long long *carry_lds = (unsigned long long *)((uint8_t *)qb_lds + (size_t)args.n_classes * (QB_STRIDE * 8)); /* [n_pats / 64 + 1] */
    uint32_t *cnt = (uint32_t *)(carry_lds + (args.n_pats / 64 + 2));
    for (uint32_t j = lane; j < (args.n_pats + 3) / 2; j += 64) cnt[j] = 0;
    for (uint32_t j = lane; j < args.n_pats / 64 + 2; j += 64) carry_lds[j] = 0;
    if (share >= args.n_shares) return;

    /* the share: the blocks that START inside [lo_b, hi_b) (no state enters a share) */
    const uint64_t lo_b = (uint64_t)share * args.share_bytes, hi_b = min(args.total, lo_b + args.share_bytes);
    /* (an emit range is made of whole blocks: the blocks that START in it, i.e. in the shares it touches) */
    if (EMIT && args.only_emit && (hi_b <= args.emit_lo || lo_b >= args.emit_hi)) return;
    const uint64_t b_lo = lower_bound_off(args.off, args.nblocks, lo_b), b_hi = lower_bound_off(args.off, args.nblocks, hi_b);
    if (b_lo >= b_hi || b_lo >= args.nblocks) return;
    const uint64_t s0 = args.off[b_lo], s1 = args.off[min(b_hi, args.nblocks)];
    if (s0 >= s1) return;
    const uint64_t w0 = s0 >> 6, w1 = (s1 - 1) >> 6;
    const uint64_t n_words_total = (args.total + 63) >> 6;
    const uint32_t below = ((lane + 63u) & 63u) << 2; /* ds_bpermute address of the lane below */
    const uint32_t row_shift = (0x00201030u >> ((lane >> 4) * 8)) & 0xffu; /* rows 0 1 2 3 hold the sums of operations 0 2 1 3 */

    /* the counts live in LDS as 16-bit fields (63 words x 64 bits per tile: 16 tiles fit) and go to the caller's counters,
     * the program's pattern order back to the caller's, every 16 tiles and at the end */
    auto flush_counts = [&]() {
        if (!args.counts) return;
        for (uint32_t wd = lane; wd < (args.n_pats + 1) / 2; wd += 64) {
            const uint32_t v = cnt[wd];
            cnt[wd] = 0;
            if (v & 0xffffu) atomicAdd(&args.counts[args.order[2 * wd]], (unsigned long long)(v & 0xffffu));
            if ((v >> 16) && 2 * wd + 1 < args.n_pats) atomicAdd(&args.counts[args.order[2 * wd + 1]], (unsigned long long)(v >> 16));
        }
    };
    uint32_t tiles_counted = 0;
    for (uint64_t wb = w0; wb <= w1; wb += TILE_WORDS) {
        if (++tiles_counted > 16) flush_counts(), tiles_counted = 1;
        /* this lane's word: wb + lane - 1 (lane 0 = the word before the tile) */
        const uint64_t w = wb + lane - 1;
        const bool have = (lane || wb > 0) && w <= w1 && w < n_words_total; /* words past the share read as zero: no runs, no carries */
        const bool full = have && (w + 1) * 64 <= args.total;          /* the corpus' last word may be short of 8 bytes of bitmap */
        auto load = [&](uint32_t c) -> uint64_t {
            if (full) return args.bitmaps[c][w];
            if (!have) return 0;
            const uint16_t *bm = (const uint16_t *)args.bitmaps[c];
            const uint64_t n16 = (args.total + 15) / 16;
            uint64_t v = 0;
            for (uint32_t k = 0; k < 4; k++)
                if (w * 4 + k < n16) v |= (uint64_t)bm[w * 4 + k] << (16 * k);
            return v & (~0ull >> (64 - (args.total - w * 64)));
        };
        const uint64_t st64 = have ? args.starts[w] : 0;
        const W2 nst = w2(~st64);
        /* which of this word's match ends are this share's to report */
        uint64_t vm = 0;
        if (lane && have && w >= w0) {
            vm = ~0ull;
            if (w == w0) vm &= ~0ull << (s0 & 63);
            if (w == w1) vm &= ~0ull >> (63 - ((s1 - 1) & 63));
        }
        /* (the mask goes into the qb words: a match end is a bit of qb, and what lies outside the share starts or ends at a block
         * start, where runs and carries stop anyway; the halo's true word keeps its bits for the runs that reach into lane 1) */
        for (uint32_t c = 0; c < args.n_classes; c++) {
            const uint64_t q = load(c) & ~st64;
            qb_lds[c * QB_STRIDE + lane] = q & vm;
            if (!lane) qb_lds[c * QB_STRIDE + 64] = q;
        }
        const uint64_t base = w * 64;
        const bool emit_tile = EMIT && wb * 64 < args.emit_hi && (wb + TILE_WORDS) * 64 > args.emit_lo; /* (uniform) */

        Runs ra;
        W2 a = {0, 0}, pa = {0, 0}, g = {0, 0};
        uint32_t pj = 0; /* patterns done in this tile (their order in the program) */
        /* the carries the patterns bring along from the tile before: 64 patterns' worth in a scalar register pair, the rest in LDS */
        unsigned long long cb = rfl64u(carry_lds[0]);
        /* one pattern with n = 1 on this lane's word: s = qb + x word for word, the carry BETWEEN the words of the tile by
         * carry-lookahead on the wavefront; -> the match ends of the word (masked to what this share reports) */
        auto core = [&](W2 qb, W2 x, uint32_t &y_lo, uint32_t &y_hi) {
            uint32_t s_lo, s_hi, and_s;
            unsigned long long c_lo, gmask, pmask;
            asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s_lo), "=s"(c_lo) : "v"(qb.lo), "v"(x.lo));
            asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s_hi), "=s"(gmask) : "v"(qb.hi), "v"(x.hi), "s"(c_lo));
            and_s = s_lo & s_hi;
            asm("v_cmp_eq_u32_e64 %0, -1, %1" : "=s"(pmask) : "v"(and_s));
            /* lane 0 stands for everything below the tile: its own word is zero (no generate, no propagate), it generates
             * exactly the carry the pattern brought along: bit 0 of cb, a shift register of the group's 64 carries */
            /* The scalar side in ONE statement whose operands are all scalar. (gmask / pmask come out of statements that also have a
             * vector output, which makes them divergent to the compiler: plain C on them may be selected as vector instructions,
             * and an "s" operand held in a vector register is an assembler error, not a copy. Their halves are passed as they are.)
             * G = generate | the carry the pattern brought along (bit 0 of cb, a shift register of the group's 64 carries: lane 0 stands
             * for everything below the tile; its own word is zero); S = (P|G) + G; the carry INTO every lane is S ^ P, the one out of
             * lane 63 the add's own carry. */
            /* (readfirstlane: in one of the two instantiations the compiler carries cb round the loops in vector registers) */
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)cb & 1u));
            uint32_t cin_lo, cin_hi, cout31, t0, t1, t2;
            asm("s_or_b32 %3, %6, %10\n\t"   /* t0 = G.lo */
                "s_or_b32 %4, %8, %3\n\t"    /* t1 = A.lo = P.lo | G.lo */
                "s_or_b32 %5, %9, %7\n\t"    /* t2 = A.hi */
                "s_add_u32 %4, %4, %3\n\t"   /* S.lo */
                "s_addc_u32 %5, %5, %7\n\t"  /* S.hi */
                "s_cselect_b32 %2, 0x80000000, 0\n\t"
                "s_xor_b32 %0, %4, %8\n\t"
                "s_xor_b32 %1, %5, %9"
                : "=&s"(cin_lo), "=&s"(cin_hi), "=&s"(cout31), "=&s"(t0), "=&s"(t1), "=&s"(t2)
                : "s"((uint32_t)gmask), "s"((uint32_t)(gmask >> 32)), "s"((uint32_t)pmask), "s"((uint32_t)(pmask >> 32)), "s"(c0)
                : "scc");
            const unsigned long long cin = (unsigned long long)cin_hi << 32 | cin_lo;
            cb = (cb >> 1) | (unsigned long long)cout31 << 32;
            uint32_t t_lo, t_hi;
            unsigned long long c2, c3;
            asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(t_lo), "=s"(c2) : "v"(s_lo), "s"(cin));
            asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(t_hi), "=s"(c3) : "v"(s_hi), "s"(c2));
            /* Y = x | (qb & ~t): the run of B above x up to where the add's carry died */
            y_lo = (t_lo & x.lo) | (~t_lo & qb.lo), y_hi = (t_hi & x.hi) | (~t_hi & qb.hi);
        };
        auto emit = [&](uint32_t y_lo, uint32_t y_hi, uint32_t index) { /* (rare: a caller asked for the records of a byte range) */
            uint64_t y = (uint64_t)y_hi << 32 | y_lo;
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
                rec.id = args.seqs[index].id;
                rec.lit = index;
                args.out[at] = rec;
            }
        };
        auto next_group = [&]() { /* 64 patterns done: their carries to LDS, the next 64 patterns' from there */
            if ((pj & 63) == 0) {
                if (lane == 0) carry_lds[(pj >> 6) - 1] = to_vector(cb);
                cb = rfl64u(carry_lds[pj >> 6]);
            }
        };
        /* the program is read one operation ahead (a scalar load and its wait at the head of every iteration were most of an
         * iteration's time), and so are the next operation's qb words from LDS */
        /* (two OP_NOP behind the program: reading ahead needs no bound; the constant address space: scalar loads) */
        const __attribute__((address_space(4))) uint64_t *ops = (const __attribute__((address_space(4))) uint64_t *)(uintptr_t)args.ops;
        const uint8_t *qb_lane = (const uint8_t *)(qb_lds + lane);
        auto qb_at = [&](uint32_t ofs) { return w2(*(const uint64_t *)(qb_lane + ofs)); };
        TileOp op1 = __builtin_bit_cast(TileOp, ops[0]);
        uint64_t raw2 = ops[1];
        const __attribute__((address_space(4))) uint64_t *ops_ahead = ops + 1;
        W2 qb_nxt = qb_at(op1.ofs1), qb2_nxt = qb_at(op1.ofs2);
        /* the counts: a lane's two popcounts are one packed word (16 bits each: 63 words x 64 bits fit); FOUR operations' words
         * are summed over the wavefront together -- v_permlane32_swap + add folds two words into the halves of one, then
         * v_permlane16_swap + add folds two of those into the four rows of one, four DPP steps finish the rows, and the last lane
         * of every row adds its two sums to the counts of ITS operation (the host numbers the counting operations 0 1 2 3 in
         * k's top bits and pads their number to a multiple of four with OP_COUNT0: nothing is pending at the end of a tile).
         * One reduction per operation was 6 DPP steps, each waiting on the one before: a third of the loop's issue slots. */
        uint32_t stash = 0, x1 = 0;
        unsigned long long slots = 0; /* the four operations' first count slots, 16 bits each, the oldest on top */
        auto account = [&](uint32_t v, uint32_t slot, uint32_t q) {
            slots = slots << 16 | slot;
            if (!(q & 1)) {
                stash = v;
            } else {
                const auto r = __builtin_amdgcn_permlane32_swap(stash, v, false, false);
                const uint32_t x = r[0] + r[1]; /* lanes 0..31: the even operation's, 32..63: the odd one's */
                if (q == 1) {
                    x1 = x;
                } else {
                    const auto r2 = __builtin_amdgcn_permlane16_swap(x1, x, false, false);
                    uint32_t t = r2[0] + r2[1]; /* rows: operations 0, 2, 1, 3 */
                    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xf, 0xf, false);
                    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xf, 0xf, false);
                    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xf, 0xf, false);
                    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xf, 0xf, false);
                    if ((lane & 15) == 15) { /* (ds_add without return: nothing waits for LDS here) */
                        const uint32_t at = (uint32_t)(to_vector(slots) >> row_shift) & 0xffffu;
                        /* (slots 2i and 2i + 1 share a word; a pair of patterns sits at an even slot: its packed sums go in as they are) */
                        __hip_atomic_fetch_add(&cnt[at >> 1], (at & 1u) ? t << 16 : t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
        };
        /* The program is nested, not dispatched: class header {cls, pairs}, per pair a header {m - 1, entries}, per entry one bit
         * (two patterns with n = 1 | one pattern, any n). Three counted loops; an operation kind is never decoded. (With one flat
         * list and a switch on the kind, the loop around the carry chains was three quarters of the kernel's instructions:
         * profiles/r04_class_seq_v2.txt.) Words are still read two ahead, their qb words one ahead. */
        W2 qb = {0, 0}, qb2 = {0, 0};
        auto next_word = [&]() -> TileOp {
            /* what the word before asked for is waited for HERE, before this word's loads are issued: scalar and LDS loads share
             * one counter that can only be waited to zero, a wait further down would wait for the new ones too */
            /* (two statements: an asm with ANY vector output makes all its outputs divergent to the compiler -- the program word, and with
             * it every loop bound and everything carried round the loops, ended up in vector registers behind v_readfirstlane) */
            asm volatile("" : "+s"(raw2) : : "memory");
            asm volatile("" : "+v"(qb_nxt.lo), "+v"(qb_nxt.hi), "+v"(qb2_nxt.lo), "+v"(qb2_nxt.hi) : : "memory");
            const TileOp op = op1;
            qb = qb_nxt, qb2 = qb2_nxt;
            op1 = __builtin_bit_cast(TileOp, raw2);
            raw2 = *++ops_ahead;
            qb_nxt = qb_at(op1.ofs1), qb2_nxt = qb_at(op1.ofs2); /* (headers carry offset 0) */
            return op;
        };
        uint32_t nq = 0; /* counting operations so far: four of them share a reduction */
        for (uint32_t c = 0; c < args.n_groups; c++) {
            const TileOp ch = next_word(); /* class A: cls, k = its pairs */
            a = w2(load(ch.cls));
            pa = from_below(a, below);
            make_runs(ra, a & nst, below);
            for (uint32_t p = 0; p < ch.k; p++) {
                const TileOp ph = next_word(); /* pair (A, m): k = m - 1, cls = its entries */
                const W2 r = shl2u(a, pa, ph.k) & run_of_u(ra, ph.k); /* R_m: m members of A end here, inside one block */
                g = shl2u(r, from_below(r, below), 1) & nst;          /* G: a match of A{m,} may end right before this byte */
                for (uint32_t e = 0; e < ph.cls; e++) {
                    const TileOp op = next_word();
                    if (op.kind & 1) {
                        /* two patterns of the pair with n = 1 (g has no block starts: g & b = g & qb), their two carry chains side by
                         * side in one instruction stream, their counts in one reduction */
                        uint32_t y0l, y0h, y1l, y1h;
                        core(qb, g & qb, y0l, y0h);
                        core(qb2, g & qb2, y1l, y1h);
                        const uint32_t pc0 = __builtin_popcount(y0l) + __builtin_popcount(y0h), pc1 = __builtin_popcount(y1l) + __builtin_popcount(y1h);
                        account(pc0 | pc1 << 16, pj, nq++ & 3u);
                        if (emit_tile) emit(y0l, y0h, args.order[pj]), emit(y1l, y1h, args.order[pj + 1]);
                        pj += 2; /* (the host pairs patterns at even positions: both in one group of 64) */
                    } else {
                        const uint32_t k = op.k;
                        W2 x;
                        if (k == 0) {
                            x = g & qb;
                        } else { /* the mandatory B{n}: n members of B end here, the first of them anywhere in the block */
                            Runs rb;
                            make_runs(rb, w2(qb_lds[op.cls * QB_STRIDE + (lane ? lane : 64)]), below);
                            const W2 b = w2(load(op.cls));
                            x = shl2u(g, from_below(g, below), k) & shl2u(b, from_below(b, below), k) & run_of_u(rb, k) & qb; /* (& qb: nothing on the halo lane, nothing outside the share) */
                        }
                        uint32_t y_lo, y_hi;
                        core(qb, x, y_lo, y_hi);
                        account(__builtin_popcount(y_lo) + __builtin_popcount(y_hi), pj, nq++ & 3u);
                        if (emit_tile) emit(y_lo, y_hi, args.order[pj]);
                        pj++;
                    }
                    next_group();
                }
            }
        }
        while (nq & 3u) account(0, args.n_pats, nq++ & 3u); /* (the last reduction's missing operations: nobody's counts) */
        if ((pj & 63) && lane == 0) carry_lds[pj >> 6] = to_vector(cb >> (64 - (pj & 63))); /* (the shift register, part of the way round) */
    }
    /* the share's counts: the program's pattern order back to the caller's */
    if (!args.counts) return;
    flush_counts();
}

} // namespace

extern "C" size_t hsgpu_class_seq_work_bytes(uint64_t total_bytes) { return SEQ_HEADER + ((total_bytes + 63) / 64) * 8 + 8; }

static int class_seq_scan(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps, unsigned n_classes,
                          uint64_t total_bytes, const void *d_off, uint64_t nblocks, uint64_t emit_lo, uint64_t emit_hi, void *d_counts,
                          void *d_out, uint64_t cap, void *d_count, void *d_work, size_t work_bytes, void *stream, bool only_emit);

extern "C" int hsgpu_class_seq_scan_dev(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps,
                                        unsigned n_classes, uint64_t total_bytes, const void *d_off, uint64_t nblocks,
                                        uint64_t emit_lo, uint64_t emit_hi, void *d_counts, void *d_out, uint64_t cap,
                                        void *d_count, void *d_work, size_t work_bytes, void *stream) {
    if (!d_counts) return HSGPU_INVALID;
    return class_seq_scan(seqs, n_seqs, d_bitmaps, n_classes, total_bytes, d_off, nblocks, emit_lo, emit_hi, d_counts, d_out, cap, d_count,
                          d_work, work_bytes, stream, false);
}

/* the records of a range of whole blocks only: the shares outside it are not walked, nothing is counted */
extern "C" int hsgpu_class_seq_emit_dev(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps,
                                        unsigned n_classes, uint64_t total_bytes, const void *d_off, uint64_t nblocks,
                                        uint64_t emit_lo, uint64_t emit_hi, void *d_out, uint64_t cap, void *d_count, void *d_work,
                                        size_t work_bytes, void *stream) {
    return class_seq_scan(seqs, n_seqs, d_bitmaps, n_classes, total_bytes, d_off, nblocks, emit_lo, emit_hi, nullptr, d_out, cap, d_count,
                          d_work, work_bytes, stream, true);
}

static int class_seq_scan(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps, unsigned n_classes,
                          uint64_t total_bytes, const void *d_off, uint64_t nblocks, uint64_t emit_lo, uint64_t emit_hi, void *d_counts,
                          void *d_out, uint64_t cap, void *d_count, void *d_work, size_t work_bytes, void *stream, bool only_emit) {
    if (!seqs || !n_seqs || !d_bitmaps || !n_classes || !d_off || !d_count || !d_work || (cap && !d_out))
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
    if (d_counts) HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)n_seqs * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st));
    if (!total_bytes || !nblocks) return HSGPU_SUCCESS;
    uint8_t *w = (uint8_t *)d_work;
    const size_t seq_bytes = (size_t)n_seqs * sizeof(hsgpu_class_seq_t), ptr_ofs = (seq_bytes + 15) & ~(size_t)15;
    const size_t tab_ofs = (ptr_ofs + (size_t)n_classes * sizeof(void *) + 15) & ~(size_t)15;
    bool aligned = true;
    for (unsigned c = 0; c < n_classes; c++) aligned = aligned && (((uintptr_t)d_bitmaps[c]) & 7) == 0;
    const bool tiled = aligned && n_classes <= TILE_MAX_CLASSES;
    if (!tiled && !d_counts) { /* (the lane-per-pattern kernels always count) */
        hsgpu_set_error("hsgpu_class_seq_emit_dev needs 8-byte aligned bitmaps and at most %u classes", TILE_MAX_CLASSES);
        return HSGPU_INVALID;
    }
    /* the header of the work area: patterns | bitmap pointers | the kernel's tables. Built on the host and uploaded on every
     * call (the work area is the caller's: it may have been freed, reused and zeroed since the last call), asynchronously, from
     * a copy kept per work area -- nothing on the stack goes out of scope under the DMA, and the call does not wait for the
     * stream (round 3 synchronised it on every call). Only a call whose header differs from the kept copy waits, for the
     * upload that may still be reading the old one. */
    std::vector<uint8_t> hdr(tab_ofs, 0);
    memcpy(hdr.data(), seqs, seq_bytes);
    memcpy(hdr.data() + ptr_ofs, d_bitmaps, (size_t)n_classes * sizeof(void *));
    uint32_t n_groups = 0;
    const uint32_t n_wgroups = (n_seqs + 255) / 256;
    size_t pairof_ofs = 0, pairs_ofs = 0, npairs_ofs = 0, order_ofs = 0;
    if (tiled) { /* the program of class_seq_tile_kernel: patterns sorted by (A, m, n, B) */
        std::vector<uint32_t> order(n_seqs);
        for (unsigned i = 0; i < n_seqs; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const hsgpu_class_seq_t &p = seqs[x], &q = seqs[y];
            if (p.a != q.a) return p.a < q.a;
            if (p.m != q.m) return p.m < q.m;
            if (p.n != q.n) return p.n < q.n;
            if (p.b != q.b) return p.b < q.b;
            return x < y;
        });
        std::vector<TileOp> ops;
        std::vector<uint16_t> prog_order;
        unsigned n_pat_done = 0;
        size_t class_at = 0, pair_at = 0; /* the open headers: their counts are filled in as the program grows */
        for (unsigned k = 0; k < n_seqs; k++) {
            const hsgpu_class_seq_t &p = seqs[order[k]];
            const bool new_a = k == 0 || seqs[order[k - 1]].a != p.a;
            if (new_a) {
                class_at = ops.size();
                ops.push_back(TileOp{0, 0, 0, 0, p.a}); /* class header: k = pairs */
                n_groups++;
            }
            if (new_a || seqs[order[k - 1]].m != p.m) {
                pair_at = ops.size();
                ops.push_back(TileOp{0, (uint8_t)(p.m - 1), 0, 0, 0}); /* pair header: cls = entries */
                ops[class_at].k++; /* (at most 16 repeat counts per class) */
            }
            /* two patterns of the same (A, m) with n = 1 become one entry -- at an even position of the program, so that
             * both carries sit in the same group of 64 */
            const bool pairable = p.n == 1 && k + 1 < n_seqs && seqs[order[k + 1]].a == p.a && seqs[order[k + 1]].m == p.m &&
                                  seqs[order[k + 1]].n == 1 && (n_pat_done & 1) == 0;
            if (pairable) {
                ops.push_back(TileOp{ENTRY_TWO, 0, (uint16_t)(p.b * QB_STRIDE * 8), (uint16_t)(seqs[order[k + 1]].b * QB_STRIDE * 8), 0});
                prog_order.push_back((uint16_t)order[k]), prog_order.push_back((uint16_t)order[k + 1]);
                n_pat_done += 2;
                k++;
            } else {
                ops.push_back(TileOp{ENTRY_ONE, (uint8_t)(p.n - 1), (uint16_t)(p.b * QB_STRIDE * 8), 0, p.b});
                prog_order.push_back((uint16_t)order[k]);
                n_pat_done += 1;
            }
            ops[pair_at].cls++;
        }
        ops.push_back(TileOp{0, 0, 0, 0, 0}), ops.push_back(TileOp{0, 0, 0, 0, 0}); /* (the kernel reads two ahead) */
        order_ofs = tab_ofs + ops.size() * sizeof(TileOp);
        hdr.resize(order_ofs + prog_order.size() * sizeof(uint16_t));
        memcpy(hdr.data() + tab_ofs, ops.data(), ops.size() * sizeof(TileOp));
        memcpy(hdr.data() + order_ofs, prog_order.data(), prog_order.size() * sizeof(uint16_t));
    } else { /* (class, repeat) pairs per group of 256 patterns for class_seq_shared_kernel */
        pairof_ofs = tab_ofs;
        pairs_ofs = pairof_ofs + (((size_t)n_seqs * 2 + 15) & ~(size_t)15);
        npairs_ofs = pairs_ofs + (size_t)n_wgroups * 256 * 2;
        hdr.resize(npairs_ofs + (size_t)n_wgroups * 2, 0);
        uint16_t *pair_of = (uint16_t *)(hdr.data() + pairof_ofs), *pairs = (uint16_t *)(hdr.data() + pairs_ofs),
                 *n_pairs = (uint16_t *)(hdr.data() + npairs_ofs);
        for (unsigned i = 0; i < n_seqs; i++) {
            const unsigned g = i / 256;
            const uint16_t key = (uint16_t)(seqs[i].a | (unsigned)(seqs[i].m - 1) << 8);
            unsigned k = 0;
            while (k < n_pairs[g] && pairs[(size_t)g * 256 + k] != key) k++;
            if (k == n_pairs[g]) pairs[(size_t)g * 256 + n_pairs[g]++] = key;
            pair_of[i] = (uint16_t)k;
        }
    }
    if (hdr.size() > SEQ_HEADER) return HSGPU_INVALID;
    {
        /* the header goes up from page-locked memory, a ring of four staging areas each with an event: a copy out of one is
         * awaited (that copy alone) before the area is written again. Round 4 kept a host copy per work-area pointer in a static
         * map -- never erased, and a changed header meant hipDeviceSynchronize(): a device-wide stall on a no-wait path (advisor). */
        /* (round 6, advisor: the ring's one mutex was held across hipEventSynchronize -- with more than four scans in flight one
         * caller's wait for another stream's copy held up every other caller. The ring's mutex now only hands out the next area;
         * an area has its own, held from the wait to the record, so that a caller waits for the users of ITS area alone.) */
        struct Stage {
            std::mutex m;
            uint8_t *p = nullptr;
            hipEvent_t ev = nullptr;
            bool used = false;
        };
        struct Ring {
            Stage st[4];
            unsigned next = 0;
        };
        static std::mutex mu;
        static std::map<int, std::unique_ptr<Ring>> rings; /* per device; process lifetime, like the device's context */
        int dev_now = 0;
        HIP_TRY(hipGetDevice(&dev_now));
        Stage *sgp;
        {
            std::lock_guard<std::mutex> lock(mu);
            std::unique_ptr<Ring> &ring = rings[dev_now];
            if (!ring) ring.reset(new Ring());
            sgp = &ring->st[ring->next++ % 4];
        }
        Stage &sg = *sgp;
        std::lock_guard<std::mutex> area(sg.m);
        if (!sg.p) {
            HIP_TRY(hipHostMalloc((void **)&sg.p, SEQ_HEADER));
            HIP_TRY(hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
        }
        if (sg.used) HIP_TRY(hipEventSynchronize(sg.ev));
        memcpy(sg.p, hdr.data(), hdr.size());
        HIP_TRY(hipMemcpyAsync(w, sg.p, hdr.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(sg.ev, st));
        sg.used = true;
    }
    const uint64_t n_words = (total_bytes + 63) / 64;
    uint64_t *starts = (uint64_t *)(w + SEQ_HEADER);
    HIP_TRY(hipMemsetAsync(starts, 0, n_words * 8 + 8, st));
    hipLaunchKernelGGL(seq_starts_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, (const uint64_t *)d_off,
                       nblocks, total_bytes, (uint32_t *)starts);
    if (tiled) {
        TileArgs t;
        t.ops = (const TileOp *)(w + tab_ofs);
        t.order = (const uint16_t *)(w + order_ofs);
        t.n_groups = n_groups;
        t.n_pats = n_seqs;
        t.n_classes = n_classes;
        t.seqs = (const hsgpu_class_seq_t *)w;
        t.bitmaps = (const uint64_t *const *)(w + ptr_ofs);
        t.starts = starts;
        t.off = (const uint64_t *)d_off;
        t.nblocks = nblocks;
        t.total = total_bytes;
#ifndef HSGPU_SEQ_WG_PER_CU
#define HSGPU_SEQ_WG_PER_CU 0 /* tuning builds: workgroups per CU (the dynamic LDS is padded so that no more fit; 0: what the device holds) */
#endif
        size_t lds = tile_lds_per_wave(n_classes, n_seqs) * (SEQ_THREADS / 64);
        if (HSGPU_SEQ_WG_PER_CU) lds = std::max<size_t>(lds, ((160u << 10) / HSGPU_SEQ_WG_PER_CU) & ~(size_t)255);
        /* the counting pass (empty emit range) has an instantiation without the record path */
        const bool emits = std::min(emit_hi, total_bytes) > emit_lo;
        const void *kfn = emits ? (const void *)class_seq_tile_kernel<true> : (const void *)class_seq_tile_kernel<false>;
        /* a wavefront per share of whole blocks, every pattern. As many shares as the device holds wavefronts of this kernel at
         * once, times HSGPU_SEQ_ROUNDS (8 192 shares on 5 120 slots were 1.6 rounds: the second one 60 % full), at least 16 KiB (four tiles) each.
         * (The device's answers are kept: asking on every call was a visible part of a small scan.) */
        int dev = 0, n_cu = 256, per_cu = 0;
        HIP_TRY(hipGetDevice(&dev));
        {
            static std::mutex mu2;
            static std::map<std::pair<const void *, int>, size_t> lds_set;      /* (kernel, device) -> dynamic LDS allowed so far */
            static std::map<std::tuple<const void *, int, size_t>, std::pair<int, int>> geo; /* -> (CUs, workgroups per CU) */
            std::lock_guard<std::mutex> lock(mu2);
            size_t &have = lds_set[{kfn, dev}];
            if (lds > have) {
                HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                have = lds;
            }
            auto it = geo.find({kfn, dev, lds});
            if (it == geo.end()) {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, SEQ_THREADS, lds) != hipSuccess || per_cu < 1) {
                    (void)hipGetLastError();
                    per_cu = 4;
                }
                it = geo.emplace(std::make_tuple(kfn, dev, lds), std::make_pair(n_cu, per_cu)).first;
            }
            n_cu = it->second.first, per_cu = it->second.second;
        }
#ifndef HSGPU_SEQ_ROUNDS
#define HSGPU_SEQ_ROUNDS 4 /* shares per wavefront slot. 256 patterns, ms per GiB at 1 GiB / 4 GiB: 1 round 7.18 / --, 2 (round 4) 6.03 / 5.92, 3 5.88 / --,
                            * 4 5.64 / 5.59, 6 6.06 / 5.40, 8 6.04 (the 16 KiB floor) / 5.43; fewer workgroups per CU than the device holds: 5 6.93, 4 6.73,
                            * 3 10.5 (profiles/r05_wg_threads_sweep.txt) */
#endif
        /* (four shares per slot up to ~1 GiB, more on larger corpora -- shares of at least ~40 KiB, at most eight per slot: the
         * table above has 6 and 8 ahead of 4 at 4 GiB and behind it at 1 GiB) */
        const uint64_t slots1 = (uint64_t)n_cu * per_cu * (SEQ_THREADS / 64);
        const uint64_t rounds = std::min<uint64_t>(2 * HSGPU_SEQ_ROUNDS, std::max<uint64_t>(HSGPU_SEQ_ROUNDS, total_bytes / (slots1 * (40u << 10))));
        const uint64_t slots = slots1 * rounds;
        uint64_t share = std::max<uint64_t>(16384, (total_bytes + slots - 1) / slots);
        share = (share + 63) & ~63ull;
        t.share_bytes = share;
        t.n_shares = (uint32_t)((total_bytes + share - 1) / share);
        t.only_emit = only_emit ? 1u : 0u;
        t.emit_lo = emit_lo;
        t.emit_hi = std::min(emit_hi, total_bytes);
        t.cap = cap;
        t.counts = (unsigned long long *)d_counts;
        t.count = (unsigned long long *)d_count;
        t.out = (hsgpu_match_t *)d_out;
        void *kargs[] = {&t};
        HIP_TRY(hipLaunchKernel(kfn, dim3((t.n_shares + SEQ_THREADS / 64 - 1) / (SEQ_THREADS / 64)), dim3(SEQ_THREADS), kargs, lds, st));
        HIP_TRY(hipGetLastError());
        return HSGPU_SUCCESS;
    }
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
