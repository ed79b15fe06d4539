/*
 * class_scan.hip -- character-class scanning on the GPU: the block-batch form of
 * the reference's "find the first/last byte in a class" accelerators
 *   shuftiExec / rshuftiExec        src/nfa/shufti.c:150-199
 *   truffleExec / rtruffleExec      src/nfa/truffle.c:118-230
 *   vermicelliExec / nverm / rverm  src/nfa/vermicelli.h:42-518
 *   run_accel dispatcher            src/nfa/accel.c:35-146
 * whose x86 form is two PSHUFB nibble lookups per 16 bytes. Every one of those
 * schemes decodes to a 256-bit class (shufti2cr, truffle2cr src/nfa/trufflecompile.cpp:77-94),
 * so the GPU takes classes as bitmaps and evaluates up to 8 of them per pass:
 *
 *   class_bitmap_kernel  streams the corpus once (16 B per lane, coalesced); per
 *       byte ONE 128-bit LDS read of a 256-entry table whose entry holds, for 8
 *       classes, a 16-bit field that is 1 when the byte value is in the class;
 *       acc |= entry << j over the lane's 16 bytes leaves the 16 membership bits
 *       of every class in its own field (no per-class work at all); each lane then
 *       stores 2 bytes per class: bit i of bitmap c <=> corpus[i] in class c.
 *       The table is replicated 16x so that the 16 lanes of every ds_read_b128
 *       lane group read 16 different 16-byte slots: conflict-free.
 *       Algorithmic bytes: 1 read + n_classes/8 written per corpus byte.
 *   class_first_last_kernel  one lane per (block, class): the accelerators' return
 *       value (first / last member offset in the block, len / -1 when none), read
 *       off the bitmaps with early exit.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <vector>

#include "internal.h"

namespace {

constexpr int CS_THREADS = 1024;
constexpr int CS_TILE = CS_THREADS * 16;

__global__ __launch_bounds__(CS_THREADS) void class_bitmap_kernel(const uint8_t *corpus, uint64_t total,
                                                                   const uint4 *lut /* [256] */, uint32_t n_classes,
                                                                   uint16_t *const *bitmaps /* [8] device ptrs */) {
    extern __shared__ __attribute__((aligned(16))) uint4 table[]; /* [256][16] */
    for (uint32_t i = threadIdx.x; i < 256 * 16; i += CS_THREADS) table[i] = lut[i >> 4];
    __syncthreads();
    const uint32_t col = threadIdx.x & 15;
    uint16_t *bm[8];
#pragma unroll
    for (int c = 0; c < 8; c++) bm[c] = bitmaps[c];
    const uint64_t n_tiles = (total + CS_TILE - 1) / CS_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t off = tile * CS_TILE + (uint64_t)threadIdx.x * 16;
        if (off >= total) continue;
        uint32_t d[4] = {0, 0, 0, 0};
        uint32_t valid = 16;
        if (off + 16 <= total) {
            const uint4 v = *(const uint4 *)(corpus + off);
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
        } else {
            valid = (uint32_t)(total - off);
            for (uint32_t i = 0; i < valid; i++) d[i >> 2] |= (uint32_t)corpus[off + i] << (8 * (i & 3));
        }
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t b = (d[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const uint4 e = table[(b << 4) | col];
            acc.x |= e.x << j;
            acc.y |= e.y << j;
            acc.z |= e.z << j;
            acc.w |= e.w << j;
        }
        const uint32_t keep = valid >= 16 ? 0xffffu : ((1u << valid) - 1u);
        const uint32_t f[8] = {acc.x & 0xffffu, acc.x >> 16, acc.y & 0xffffu, acc.y >> 16,
                               acc.z & 0xffffu, acc.z >> 16, acc.w & 0xffffu, acc.w >> 16};
#pragma unroll
        for (int c = 0; c < 8; c++)
            if ((uint32_t)c < n_classes) bm[c][off >> 4] = (uint16_t)(f[c] & keep);
    }
}

/* first / last set bit of a bitmap inside [lo, hi); 0xffffffff when none */
__device__ __forceinline__ uint32_t first_bit(const uint16_t *bm, uint64_t lo, uint64_t hi) {
    if (lo >= hi) return 0xffffffffu;
    for (uint64_t w = lo >> 4; w <= (hi - 1) >> 4; w++) {
        uint32_t m = bm[w];
        if (w == lo >> 4) m &= 0xffffu << (lo & 15);
        if (w == (hi - 1) >> 4) m &= 0xffffu >> (15 - ((hi - 1) & 15));
        if (m) return (uint32_t)((w << 4) + __builtin_ctz(m) - lo);
    }
    return 0xffffffffu;
}
__device__ __forceinline__ uint32_t last_bit(const uint16_t *bm, uint64_t lo, uint64_t hi) {
    if (lo >= hi) return 0xffffffffu;
    for (uint64_t w = (hi - 1) >> 4;; w--) {
        uint32_t m = bm[w];
        if (w == lo >> 4) m &= 0xffffu << (lo & 15);
        if (w == (hi - 1) >> 4) m &= 0xffffu >> (15 - ((hi - 1) & 15));
        if (m) return (uint32_t)((w << 4) + 31 - __builtin_clz(m) - lo);
        if (w == lo >> 4) break;
    }
    return 0xffffffffu;
}

/* the same over 64-bit words (bitmaps that are 8-byte aligned): a 120-byte line is 2-3 reads
 * instead of 8 */
__device__ __forceinline__ uint32_t first_bit64(const uint64_t *bm, uint64_t lo, uint64_t hi) {
    if (lo >= hi) return 0xffffffffu;
    const uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
    for (uint64_t w = w0; w <= w1; w++) {
        uint64_t m = bm[w];
        if (w == w0) m &= ~0ull << (lo & 63);
        if (w == w1) m &= ~0ull >> (63 - ((hi - 1) & 63));
        if (m) return (uint32_t)((w << 6) + __builtin_ctzll(m) - lo);
    }
    return 0xffffffffu;
}
__device__ __forceinline__ uint32_t last_bit64(const uint64_t *bm, uint64_t lo, uint64_t hi) {
    if (lo >= hi) return 0xffffffffu;
    const uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
    for (uint64_t w = w1;; w--) {
        uint64_t m = bm[w];
        if (w == w0) m &= ~0ull << (lo & 63);
        if (w == w1) m &= ~0ull >> (63 - ((hi - 1) & 63));
        if (m) return (uint32_t)((w << 6) + 63 - __builtin_clzll(m) - lo);
        if (w == w0) break;
    }
    return 0xffffffffu;
}

/* first/last member of class c inside block b, from bitmap c. (Fusing this into the
 * classification kernel -- fields kept in LDS, a per-tile block index, atomics for blocks
 * crossing tiles -- measured 2.1 ms/GiB against 1.27 ms for these two streaming kernels: the
 * per-tile barriers and dependent offset reads cost more than re-reading the bitmaps.) */
__global__ void class_first_last_kernel(const uint64_t *off, uint64_t nblocks, uint32_t n_classes,
                                        uint16_t *const *bitmaps, uint64_t total, uint32_t *first, uint32_t *last) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks * n_classes) return;
    const uint32_t c = (uint32_t)(i / nblocks);
    const uint64_t b = i % nblocks;
    const uint64_t lo = off[b], hi = off[b + 1];
    const uint16_t *bm = bitmaps[c];
    /* the bitmap holds (total + 15) / 16 16-bit words: whole 64-bit words only below that */
    const bool wide = (((uintptr_t)bm & 7) == 0) && (((hi + 63) >> 6) << 2) <= ((total + 15) >> 4);
    if (first) {
        const uint32_t f = wide ? first_bit64((const uint64_t *)bm, lo, hi) : first_bit(bm, lo, hi);
        first[i] = f == 0xffffffffu ? (uint32_t)(hi - lo) : f;
    }
    if (last) last[i] = wide ? last_bit64((const uint64_t *)bm, lo, hi) : last_bit(bm, lo, hi);
}

/* ---- two-byte sets (double shufti / double vermicelli) --------------------------
 * Table entry v: T1 (first uint4) field k (16 bits) = t_k(v), T2 (second uint4) field k =
 * u_k(v); unused fields 0xff. A pair matches set k iff (t_k | u_k) != 0xff. */
__global__ __launch_bounds__(CS_THREADS) void pair_bitmap_kernel(const uint8_t *corpus, uint64_t total,
                                                                  const uint4 *lut /* [256][2] */, uint32_t n_pairs,
                                                                  uint16_t *const *bitmaps) {
    __shared__ __attribute__((aligned(16))) uint4 table[256 * 2];
    for (uint32_t i = threadIdx.x; i < 256 * 2; i += CS_THREADS) table[i] = lut[i];
    __syncthreads();
    uint16_t *bm[8];
#pragma unroll
    for (int c = 0; c < 8; c++) bm[c] = bitmaps[c];
    const uint64_t n_tiles = (total + CS_TILE - 1) / CS_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t off = tile * CS_TILE + (uint64_t)threadIdx.x * 16;
        if (off >= total) continue;
        uint32_t d[5] = {0, 0, 0, 0, 0};
        uint32_t valid = 16; /* positions whose SECOND byte exists too */
        if (off + 17 <= total) {
            const uint4 v = *(const uint4 *)(corpus + off);
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
            d[4] = corpus[off + 16];
        } else {
            const uint32_t have = (uint32_t)(total - off); /* 1..16 */
            for (uint32_t i = 0; i < have; i++) d[i >> 2] |= (uint32_t)corpus[off + i] << (8 * (i & 3));
            valid = have - 1;
        }
        uint4 acc = make_uint4(0, 0, 0, 0);
        uint4 t1 = table[2 * (d[0] & 0xffu)];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t nb = (d[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu;
            const uint4 u = table[2 * nb + 1];
            const uint4 v = make_uint4(t1.x | u.x, t1.y | u.y, t1.z | u.z, t1.w | u.w);
            /* per 16-bit field: 1 iff its low byte is not 0xff */
            auto alive = [](uint32_t x) { return (((~x & 0x00ff00ffu) + 0x00ff00ffu) >> 8) & 0x00010001u; };
            acc.x |= alive(v.x) << j;
            acc.y |= alive(v.y) << j;
            acc.z |= alive(v.z) << j;
            acc.w |= alive(v.w) << j;
            t1 = table[2 * nb];
        }
        const uint32_t keep = valid >= 16 ? 0xffffu : ((1u << valid) - 1u);
        const uint32_t f[8] = {acc.x & 0xffffu, acc.x >> 16, acc.y & 0xffffu, acc.y >> 16,
                               acc.z & 0xffffu, acc.z >> 16, acc.w & 0xffffu, acc.w >> 16};
#pragma unroll
        for (int c = 0; c < 8; c++)
            if ((uint32_t)c < n_pairs) bm[c][off >> 4] = (uint16_t)(f[c] & keep);
    }
}

/* per (set, block): forward = first pair inside the block, else the partial match at the
 * block's last byte, else len; reverse = second byte of the last pair */
__global__ void pair_first_last_kernel(const uint8_t *corpus, const uint64_t *off, uint64_t nblocks, uint32_t n_pairs,
                                       const uint4 *lut, uint16_t *const *bitmaps, uint32_t *first, uint32_t *last) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks * n_pairs) return;
    const uint32_t k = (uint32_t)(i / nblocks);
    const uint64_t b = i % nblocks;
    const uint64_t lo = off[b], hi = off[b + 1];
    const uint16_t *bm = bitmaps[k];
    const uint64_t hi_pairs = hi > lo ? hi - 1 : lo; /* a pair's first byte: [lo, hi - 1) */
    if (first) {
        uint32_t f = first_bit(bm, lo, hi_pairs);
        if (f == 0xffffffffu) {
            f = (uint32_t)(hi - lo);
            if (hi > lo) {
                const uint32_t *t1 = (const uint32_t *)&lut[2 * corpus[hi - 1]];
                const uint32_t tk = (t1[k >> 1] >> (16 * (k & 1))) & 0xffu;
                if (tk != 0xffu) f -= 1;
            }
        }
        first[i] = f;
    }
    if (last) {
        const uint32_t l = last_bit(bm, lo, hi_pairs);
        last[i] = l == 0xffffffffu ? l : l + 1;
    }
}

} // namespace

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            hsgpu_set_error("%s failed: %s", #expr, hipGetErrorString(e_));        \
            return (e_ == hipErrorOutOfMemory) ? HSGPU_NOMEM : HSGPU_UNKNOWN_ERROR; \
        }                                                                          \
    } while (0)

/* ---- class construction: the decoders of the reference's accel schemes ------ */

extern "C" int hsgpu_class_from_shufti(const uint8_t lo[16], const uint8_t hi[16], hsgpu_class_t *out) {
    if (!lo || !hi || !out) return HSGPU_INVALID;
    memset(out, 0, sizeof(*out));
    for (unsigned c = 0; c < 256; c++) /* member iff lo[c & 15] & hi[c >> 4], src/nfa/shufti.c:75-87 */
        if (lo[c & 15] & hi[c >> 4]) out->bitmap[c >> 3] |= (uint8_t)(1u << (c & 7));
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_class_from_truffle(const uint8_t mask1[16], const uint8_t mask2[16], hsgpu_class_t *out) {
    if (!mask1 || !mask2 || !out) return HSGPU_INVALID;
    memset(out, 0, sizeof(*out));
    for (unsigned c = 0; c < 256; c++) { /* truffle2cr, src/nfa/trufflecompile.cpp:77-94 */
        const uint8_t *m = (c & 0x80) ? mask2 : mask1;
        if ((m[c & 15] >> ((c >> 4) & 7)) & 1) out->bitmap[c >> 3] |= (uint8_t)(1u << (c & 7));
    }
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_class_from_verm(uint8_t c, int nocase, int negate, hsgpu_class_t *out) {
    if (!out) return HSGPU_INVALID;
    memset(out, 0, sizeof(*out));
    const bool alpha = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    const uint8_t mask = (nocase && alpha) ? 0xdf : 0xff; /* vermicelli.h:42-104, CASE_CLEAR */
    for (unsigned v = 0; v < 256; v++) {
        const bool eq = (v & mask) == (unsigned)(c & mask);
        if (eq != (negate != 0)) out->bitmap[v >> 3] |= (uint8_t)(1u << (v & 7));
    }
    return HSGPU_SUCCESS;
}

/* Build masks from a class, for callers that want the reference's own encodings
 * (truffle can represent every class: src/nfa/trufflecompile.cpp:59-72). */
extern "C" int hsgpu_class_to_truffle(const hsgpu_class_t *cls, uint8_t mask1[16], uint8_t mask2[16]) {
    if (!cls || !mask1 || !mask2) return HSGPU_INVALID;
    memset(mask1, 0, 16);
    memset(mask2, 0, 16);
    for (unsigned c = 0; c < 256; c++)
        if (cls->bitmap[c >> 3] >> (c & 7) & 1) {
            uint8_t *m = (c & 0x80) ? mask2 : mask1;
            m[c & 15] |= (uint8_t)(1u << ((c >> 4) & 7));
        }
    return HSGPU_SUCCESS;
}

/* ---- the scan ---------------------------------------------------------------- */

struct ClassScratch {
    void *lut = nullptr;  /* uint4[256] */
    void *ptrs = nullptr; /* uint16_t*[8] */
};

extern "C" int hsgpu_class_scan_dev(const hsgpu_class_t *classes, unsigned n_classes, const void *d_corpus,
                                    uint64_t total_bytes, const void *d_off, uint64_t nblocks,
                                    void *const *d_bitmaps, void *d_first, void *d_last, void *d_work,
                                    void *stream) {
    if (!classes || n_classes == 0 || n_classes > HSGPU_CLASS_MAX || !d_bitmaps || !d_work) return HSGPU_INVALID;
    if (((uintptr_t)d_corpus & 15) || ((uintptr_t)d_work & 15)) return HSGPU_INVALID;
    if ((d_first || d_last) && (!d_off || nblocks == 0)) return HSGPU_INVALID;
    hipStream_t st = (hipStream_t)stream;
    /* host-built 256-entry table: entry[v] field c (16 bits) = 1 iff v in class c */
    uint32_t lut[256][4];
    memset(lut, 0, sizeof(lut));
    for (unsigned c = 0; c < n_classes; c++)
        for (unsigned v = 0; v < 256; v++)
            if (classes[c].bitmap[v >> 3] >> (v & 7) & 1) lut[v][c >> 1] |= 1u << (16 * (c & 1));
    /* d_work: [4096 B table][64 B pointer array] supplied by the caller (no hidden allocation) */
    uint8_t *work = (uint8_t *)d_work;
    void *ptrs[8] = {nullptr};
    for (unsigned c = 0; c < n_classes; c++) {
        if (!d_bitmaps[c] || ((uintptr_t)d_bitmaps[c] & 1)) return HSGPU_INVALID;
        ptrs[c] = d_bitmaps[c];
    }
    HIP_TRY(hipMemcpyAsync(work, lut, sizeof(lut), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(work + sizeof(lut), ptrs, sizeof(ptrs), hipMemcpyHostToDevice, st));
    if (total_bytes == 0) return HSGPU_SUCCESS;
    const uint4 *d_lut = (const uint4 *)work;
    uint16_t *const *d_ptrs = (uint16_t *const *)(work + sizeof(lut));
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    static bool attr_set = false;
    const size_t lds = 256 * 16 * sizeof(uint4);
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)class_bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        attr_set = true;
    }
    const uint64_t n_tiles = (total_bytes + CS_TILE - 1) / CS_TILE;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)n_cu * 2);
    const uint8_t *corpus = (const uint8_t *)d_corpus;
    hipLaunchKernelGGL(class_bitmap_kernel, dim3(grid), dim3(CS_THREADS), lds, st, corpus, total_bytes, d_lut, n_classes,
                       d_ptrs);
    HIP_TRY(hipGetLastError());
    if (d_first || d_last) {
        const uint64_t n = nblocks * n_classes;
        hipLaunchKernelGGL(class_first_last_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const uint64_t *)d_off, nblocks, n_classes, d_ptrs, total_bytes, (uint32_t *)d_first, (uint32_t *)d_last);
        HIP_TRY(hipGetLastError());
    }
    return HSGPU_SUCCESS;
}

/* ---- two-byte sets: construction ------------------------------------------------ */

static void pair_clear(hsgpu_pair_t *p) { memset(p, 0xff, sizeof(*p)); }

/* bucket `bit` accepts first bytes {lo nibble in la} x {hi nibble in ha} and second bytes
 * {lb} x {hb} (16-bit nibble sets): clear the bucket's bit in the accepting mask entries */
static void pair_add_bucket(hsgpu_pair_t *p, unsigned bit, uint16_t la, uint16_t ha, uint16_t lb, uint16_t hb) {
    const uint8_t clr = (uint8_t)~(1u << bit);
    for (unsigned n = 0; n < 16; n++) {
        if (la >> n & 1) p->lo1[n] &= clr;
        if (ha >> n & 1) p->hi1[n] &= clr;
        if (lb >> n & 1) p->lo2[n] &= clr;
        if (hb >> n & 1) p->hi2[n] &= clr;
    }
}

extern "C" int hsgpu_pair_from_dshufti(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                                       const uint8_t hi2[16], hsgpu_pair_t *out) {
    if (!lo1 || !hi1 || !lo2 || !hi2 || !out) return HSGPU_INVALID;
    memcpy(out->lo1, lo1, 16);
    memcpy(out->hi1, hi1, 16);
    memcpy(out->lo2, lo2, 16);
    memcpy(out->hi2, hi2, 16);
    return HSGPU_SUCCESS;
}

/* {b : (b & m) == c} is a product of a low-nibble set and a high-nibble set */
static void masked_byte_sets(uint8_t c, uint8_t m, uint16_t *lo, uint16_t *hi) {
    *lo = *hi = 0;
    for (unsigned n = 0; n < 16; n++) {
        if ((n & (m & 15u)) == (c & 15u)) *lo |= (uint16_t)(1u << n);
        if ((n & (m >> 4)) == (unsigned)(c >> 4)) *hi |= (uint16_t)(1u << n);
    }
}

extern "C" int hsgpu_pair_from_dverm_masked(uint8_t c1, uint8_t c2, uint8_t m1, uint8_t m2, hsgpu_pair_t *out) {
    if (!out) return HSGPU_INVALID;
    pair_clear(out); /* (b & m) == c, vermicelli.h:241-317; an unsatisfiable c leaves the set empty */
    uint16_t la, ha, lb, hb;
    masked_byte_sets(c1, m1, &la, &ha);
    masked_byte_sets(c2, m2, &lb, &hb);
    if ((c1 & ~m1) || (c2 & ~m2)) return HSGPU_SUCCESS;
    pair_add_bucket(out, 0, la, ha, lb, hb);
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_pair_from_dverm(uint8_t c1, uint8_t c2, int nocase, hsgpu_pair_t *out) {
    auto alpha = [](uint8_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
    /* caseless: compare with CASE_CLEAR (0xdf) on alphabetic bytes, vermicelli.h:169-239 */
    const uint8_t m1 = (nocase && alpha(c1)) ? 0xdf : 0xff, m2 = (nocase && alpha(c2)) ? 0xdf : 0xff;
    return hsgpu_pair_from_dverm_masked(c1 & m1, c2 & m2, m1, m2, out);
}

/* Bucketing as the reference does it (shufticompile.cpp:135-209): every sequence starts as a
 * rectangle of four single-nibble sets; rectangles that agree in three of the four sets are
 * merged by uniting the fourth, one dimension after the other. Exact (no over-approximation). */
extern "C" int hsgpu_pair_build(const hsgpu_class_t *onechar, const uint8_t *pairs, size_t npairs, hsgpu_pair_t *out) {
    if (!out || (npairs && !pairs)) return HSGPU_INVALID;
    typedef std::array<uint16_t, 4> Rect;
    std::vector<Rect> rects;
    for (size_t i = 0; i < npairs; i++) {
        const uint8_t a = pairs[2 * i], b = pairs[2 * i + 1];
        rects.push_back(Rect{{(uint16_t)(1u << (a & 15)), (uint16_t)(1u << (a >> 4)), (uint16_t)(1u << (b & 15)),
                              (uint16_t)(1u << (b >> 4))}});
    }
    if (onechar)
        for (unsigned v = 0; v < 256; v++)
            if (onechar->bitmap[v >> 3] >> (v & 7) & 1)
                rects.push_back(Rect{{(uint16_t)(1u << (v & 15)), (uint16_t)(1u << (v >> 4)), 0xffff, 0xffff}});
    for (int dim = 0; dim < 4; dim++) {
        std::map<Rect, uint16_t> merged; /* the other three sets -> union of this one */
        for (const Rect &r : rects) {
            Rect key = r;
            key[dim] = 0;
            merged[key] |= r[dim];
        }
        rects.clear();
        for (const auto &kv : merged) {
            Rect r = kv.first;
            r[dim] = kv.second;
            rects.push_back(r);
        }
    }
    if (rects.size() > HSGPU_PAIR_MAX) {
        hsgpu_set_error("two-byte set needs %zu buckets (> 8)", rects.size());
        return HSGPU_COMPILER_ERROR;
    }
    pair_clear(out);
    for (size_t i = 0; i < rects.size(); i++) pair_add_bucket(out, (unsigned)i, rects[i][0], rects[i][1], rects[i][2], rects[i][3]);
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_pair_test(const hsgpu_pair_t *p, uint8_t a, uint8_t b) {
    if (!p) return 0;
    return (uint8_t)(p->lo1[a & 15] | p->hi1[a >> 4] | p->lo2[b & 15] | p->hi2[b >> 4]) != 0xff;
}

extern "C" int hsgpu_pair_scan_dev(const hsgpu_pair_t *pairs, unsigned n_pairs, const void *d_corpus,
                                   uint64_t total_bytes, const void *d_off, uint64_t nblocks, void *const *d_bitmaps,
                                   void *d_first, void *d_last, void *d_work, void *stream) {
    if (!pairs || n_pairs == 0 || n_pairs > HSGPU_PAIR_MAX || !d_bitmaps || !d_work) return HSGPU_INVALID;
    if (((uintptr_t)d_corpus & 15) || ((uintptr_t)d_work & 15)) return HSGPU_INVALID;
    if ((d_first || d_last) && (!d_off || nblocks == 0)) return HSGPU_INVALID;
    hipStream_t st = (hipStream_t)stream;
    /* entry v: [T1: 8 x 16-bit fields t_k(v)] [T2: 8 x 16-bit fields u_k(v)]; unused sets 0xff (dead) */
    uint32_t lut[256][8];
    for (unsigned v = 0; v < 256; v++) {
        for (unsigned w = 0; w < 8; w++) lut[v][w] = 0x00ff00ffu;
        for (unsigned k = 0; k < n_pairs; k++) {
            const uint32_t t = pairs[k].lo1[v & 15] | pairs[k].hi1[v >> 4];
            const uint32_t u = pairs[k].lo2[v & 15] | pairs[k].hi2[v >> 4];
            const unsigned sh = 16 * (k & 1);
            lut[v][k >> 1] = (lut[v][k >> 1] & ~(0xffffu << sh)) | t << sh;
            lut[v][4 + (k >> 1)] = (lut[v][4 + (k >> 1)] & ~(0xffffu << sh)) | u << sh;
        }
    }
    uint8_t *work = (uint8_t *)d_work;
    void *ptrs[8] = {nullptr};
    for (unsigned k = 0; k < n_pairs; k++) {
        if (!d_bitmaps[k] || ((uintptr_t)d_bitmaps[k] & 1)) return HSGPU_INVALID;
        ptrs[k] = d_bitmaps[k];
    }
    static_assert(sizeof(lut) + sizeof(ptrs) == HSGPU_PAIR_WORK_BYTES, "work area layout");
    HIP_TRY(hipMemcpyAsync(work, lut, sizeof(lut), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(work + sizeof(lut), ptrs, sizeof(ptrs), hipMemcpyHostToDevice, st));
    if (total_bytes == 0) return HSGPU_SUCCESS;
    const uint4 *d_lut = (const uint4 *)work;
    uint16_t *const *d_ptrs = (uint16_t *const *)(work + sizeof(lut));
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    const uint64_t n_tiles = (total_bytes + CS_TILE - 1) / CS_TILE;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)n_cu * 2);
    const uint8_t *corpus = (const uint8_t *)d_corpus;
    hipLaunchKernelGGL(pair_bitmap_kernel, dim3(grid), dim3(CS_THREADS), 0, st, corpus, total_bytes, d_lut, n_pairs, d_ptrs);
    HIP_TRY(hipGetLastError());
    if (d_first || d_last) {
        const uint64_t n = nblocks * n_pairs;
        hipLaunchKernelGGL(pair_first_last_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, corpus,
                           (const uint64_t *)d_off, nblocks, n_pairs, d_lut, d_ptrs, (uint32_t *)d_first,
                           (uint32_t *)d_last);
        HIP_TRY(hipGetLastError());
    }
    return HSGPU_SUCCESS;
}

/* ---- hwlmExec's pre-skip for a whole batch ------------------------------------------
 * do_accel_block (src/hwlm/hwlm.c:80-99) per block: when at least 16 bytes follow `start`,
 * start' = max(0, hit - offset) where hit = run_hwlm_accel over [start, len) (hwlm.c:48-77):
 * the first member (class schemes) / the first pair, else len - 1 on a partial match at the
 * last byte, else len (double vermicelli). Blocks with fewer than 16 bytes after `start`, and
 * ACCEL_NONE, keep their start. The result may lie BEFORE the old start (hit - offset < start):
 * that is what the reference computes, and it is what this returns. */
__global__ void forward_skip_kernel(const uint8_t *corpus, const uint64_t *off, uint64_t nblocks,
                                    const uint16_t *bm, const uint4 *lut, int is_pair, uint32_t offset,
                                    const uint32_t *start_in, uint32_t start_all, uint32_t *start_out) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint64_t lo = off[b], hi = off[b + 1];
    const uint64_t len = hi - lo;
    const uint64_t s = start_in ? start_in[b] : start_all;
    if (s > len || len - s < 16 || !bm) {
        start_out[b] = (uint32_t)s;
        return;
    }
    uint64_t hit;
    if (!is_pair) {
        const uint32_t f = first_bit(bm, lo + s, hi);
        hit = f == 0xffffffffu ? len : s + f;
    } else {
        const uint32_t f = first_bit(bm, lo + s, hi - 1);
        if (f != 0xffffffffu) {
            hit = s + f;
        } else {
            const uint32_t *t1 = (const uint32_t *)&lut[2 * corpus[hi - 1]];
            hit = (t1[0] & 0xffu) != 0xffu ? len - 1 : len;
        }
    }
    start_out[b] = (uint32_t)(hit > offset ? hit - offset : 0);
}

extern "C" int hsgpu_hwlm_forward_skip_dev(const hsgpu_accel_t *aux, const void *d_corpus, uint64_t total_bytes,
                                           const void *d_off, uint64_t nblocks, const void *d_start_in,
                                           uint32_t start, void *d_start_out, void *d_bitmap, void *d_work,
                                           void *stream) {
    if (!aux || !d_off || nblocks == 0 || !d_start_out || !d_work) return HSGPU_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int is_pair = 0;
    const uint16_t *bm = (const uint16_t *)d_bitmap;
    void *bitmaps[1] = {d_bitmap};
    int rv = HSGPU_SUCCESS;
    switch (aux->type) {
    case HSGPU_ACCEL_NONE:
        bm = nullptr;
        break;
    case HSGPU_ACCEL_VERM:
    case HSGPU_ACCEL_VERM_NOCASE:
    case HSGPU_ACCEL_SHUFTI:
    case HSGPU_ACCEL_TRUFFLE: {
        hsgpu_class_t cls;
        if (aux->type == HSGPU_ACCEL_SHUFTI)
            rv = hsgpu_class_from_shufti(aux->mask_lo, aux->mask_hi, &cls);
        else if (aux->type == HSGPU_ACCEL_TRUFFLE)
            rv = hsgpu_class_from_truffle(aux->mask_lo, aux->mask_hi, &cls);
        else
            rv = hsgpu_class_from_verm(aux->c1, aux->type == HSGPU_ACCEL_VERM_NOCASE, 0, &cls);
        if (rv != HSGPU_SUCCESS) return rv;
        if (!d_bitmap) return HSGPU_INVALID;
        rv = hsgpu_class_scan_dev(&cls, 1, d_corpus, total_bytes, nullptr, 0, bitmaps, nullptr, nullptr, d_work, stream);
        break;
    }
    case HSGPU_ACCEL_DVERM:
    case HSGPU_ACCEL_DVERM_NOCASE: {
        hsgpu_pair_t pr;
        rv = hsgpu_pair_from_dverm(aux->c1, aux->c2, aux->type == HSGPU_ACCEL_DVERM_NOCASE, &pr);
        if (rv != HSGPU_SUCCESS) return rv;
        if (!d_bitmap) return HSGPU_INVALID;
        is_pair = 1;
        rv = hsgpu_pair_scan_dev(&pr, 1, d_corpus, total_bytes, nullptr, 0, bitmaps, nullptr, nullptr, d_work, stream);
        break;
    }
    default:
        return HSGPU_INVALID;
    }
    if (rv != HSGPU_SUCCESS) return rv;
    hipLaunchKernelGGL(forward_skip_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st,
                       (const uint8_t *)d_corpus, (const uint64_t *)d_off, nblocks, bm, (const uint4 *)d_work, is_pair,
                       (uint32_t)aux->offset, (const uint32_t *)d_start_in, start, (uint32_t *)d_start_out);
    HIP_TRY(hipGetLastError());
    return HSGPU_SUCCESS;
}
