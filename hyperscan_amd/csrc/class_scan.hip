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
#include <cstring>

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

/* first/last member of class c inside block b, from bitmap c */
__global__ void class_first_last_kernel(const uint64_t *off, uint64_t nblocks, uint32_t n_classes,
                                        uint16_t *const *bitmaps, uint32_t *first, uint32_t *last) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks * n_classes) return;
    const uint32_t c = (uint32_t)(i / nblocks);
    const uint64_t b = i % nblocks;
    const uint64_t lo = off[b], hi = off[b + 1];
    const uint16_t *bm = bitmaps[c];
    uint32_t f = (uint32_t)(hi - lo), l = 0xffffffffu;
    if (first) {
        for (uint64_t w = lo >> 4; w <= (hi ? (hi - 1) >> 4 : 0) && lo < hi; w++) {
            uint32_t m = bm[w];
            if (w == lo >> 4) m &= 0xffffu << (lo & 15);
            if (w == (hi - 1) >> 4) m &= 0xffffu >> (15 - ((hi - 1) & 15));
            if (m) {
                f = (uint32_t)((w << 4) + __builtin_ctz(m) - lo);
                break;
            }
        }
        first[i] = f;
    }
    if (last) {
        if (lo < hi) {
            for (uint64_t w = (hi - 1) >> 4;; w--) {
                uint32_t m = bm[w];
                if (w == lo >> 4) m &= 0xffffu << (lo & 15);
                if (w == (hi - 1) >> 4) m &= 0xffffu >> (15 - ((hi - 1) & 15));
                if (m) {
                    l = (uint32_t)((w << 4) + 31 - __builtin_clz(m) - lo);
                    break;
                }
                if (w == lo >> 4) break;
            }
        }
        last[i] = l;
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
                           (const uint64_t *)d_off, nblocks, n_classes, d_ptrs, (uint32_t *)d_first, (uint32_t *)d_last);
        HIP_TRY(hipGetLastError());
    }
    return HSGPU_SUCCESS;
}
