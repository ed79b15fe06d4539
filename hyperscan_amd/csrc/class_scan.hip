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
 *   class_tile_{fl,bm}_kernel  stream the corpus once: membership bitmaps (bit i of bitmap c <=>
 *       corpus[i] in class c) and, for the blocks inside a wavefront's 4 KiB tile, the
 *       accelerators' return value (first / last member offset in the block, len / -1 when
 *       none) straight from the tile's bits in LDS.
 *       Algorithmic bytes: 1 read + n_classes/8 written per corpus byte (+ 8 per block read,
 *       8 per block and class written for first / last).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <vector>

#include "internal.h"

/* the CU count of a device, asked once (hipGetDeviceProperties on every call was a visible part of a small scan: advisor, round 4) */
static int cached_cu_count(int dev) {
    static std::mutex mu;
    static std::map<int, int> known;
    std::lock_guard<std::mutex> g(mu);
    auto it = known.find(dev);
    if (it != known.end()) return it->second;
    hipDeviceProp_t prop;
    const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    known[dev] = n;
    return n;
}

namespace {

constexpr int CS_THREADS = 1024;
constexpr int CS_TILE = CS_THREADS * 16;

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

/* ---- the classification kernel ------------------------------------------------------
 * class_tile_*_kernel: every wavefront streams ONE contiguous share of the corpus in 4 KiB tiles, 64
 * contiguous bytes per lane (the next tile's loads are issued before the current one is worked on).
 *   classify   per byte ONE ds_read_b64 of a 256-entry table whose entry holds, for 8 classes, an
 *              8-bit field that is 1 when the byte value is in the class; acc |= entry << j over 8
 *              bytes leaves their 8 membership bits of every class in that class's field: no
 *              per-class work. The table is replicated 32x (64 KiB) so that the 32 lanes of each
 *              half of a ds_read_b64 hit 32 different bank pairs: conflict-free. Eight 4x4 byte
 *              transposes (v_perm) turn the 8 accumulators into one 64-bit membership word per class.
 *   store      8 bytes per lane and class: 512 contiguous bytes per wavefront and class.
 *   first/last (FL) the same words go to a 4 KiB LDS area of the wavefront, and the blocks that lie
 *              wholly inside the tile get their first / last member per class from there, one lane
 *              per block -- the bitmaps are never read back from HBM. The block offsets a tile needs
 *              are requested one tile ahead (which blocks start in a tile depends on the offsets
 *              alone, not on the classification), so no load is waited for. Wavefront-local: no
 *              workgroup barrier after the table is staged. A block that crosses a tile boundary stays
 *              "open": what is known of its first / last members is carried to the next tile in
 *              scalar registers (found with ballots over the lanes' words, no LDS); a wavefront whose
 *              share ends inside a block reads on past its share until that block ends.
 * HBM traffic per corpus byte: 1 read + n_classes / 8 written (+ offsets and first / last per block):
 * the algorithmic bytes. (Round 1's two kernels re-read every bitmap and every offset per class.) */
constexpr int CT_THREADS = 1024;
constexpr uint32_t CT_TILE = 4096;                     /* bytes per wavefront tile */
constexpr uint32_t CT_TABLE_BYTES = 256 * 32 * 8;       /* 64 KiB */
constexpr uint32_t CT_PAD = 16;                         /* two zero words on either side of a class's 64 tile words */
constexpr uint32_t CT_ROW = CT_PAD + 64 * 8 + CT_PAD;
constexpr uint32_t CT_WAVE_LDS = 8 * CT_ROW + 64;       /* 8 classes per wavefront + the open block's {first, last} so far per class */
constexpr uint32_t CT_LDS_BYTES = CT_TABLE_BYTES + (CT_THREADS / 64) * CT_WAVE_LDS;

__device__ __forceinline__ uint2 lds_read64(uint32_t addr) {
    const uint64_t v = *(const __attribute__((address_space(3))) uint64_t *)(uintptr_t)addr;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ uint64_t lds_word(uint32_t addr) {
    return *(const __attribute__((address_space(3))) uint64_t *)(uintptr_t)addr;
}
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
/* index of the lowest / number of zeros above the highest set bit; 0xffffffff for 0 */
__device__ __forceinline__ uint32_t ffbl(uint32_t x) {
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t ffbh(uint32_t x) {
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

/* rows r0..r3 (4 bytes each) -> columns: o[c] = {r0.byte c, r1.byte c, r2.byte c, r3.byte c} */
__device__ __forceinline__ void transpose4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t o[4]) {
    const uint32_t t0 = perm(r1, r0, 0x05010400u), t1 = perm(r1, r0, 0x07030602u);
    const uint32_t t2 = perm(r3, r2, 0x05010400u), t3 = perm(r3, r2, 0x07030602u);
    o[0] = perm(t2, t0, 0x05040100u);
    o[1] = perm(t2, t0, 0x07060302u);
    o[2] = perm(t3, t1, 0x05040100u);
    o[3] = perm(t3, t1, 0x07060302u);
}

/* 8 bytes (two dwords) -> the 8-bit membership fields of 8 classes */
__device__ __forceinline__ uint2 classify8(uint32_t d0, uint32_t d1, uint32_t col8) {
    uint2 acc = make_uint2(0, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t d = j < 4 ? d0 : d1;
        const int q = j & 3;
        const uint32_t sh = q == 0 ? d << 8 : (q == 1 ? d : d >> (8 * (q - 1)));
        const uint2 e = lds_read64((sh & 0xff00u) | col8);
        acc.x |= e.x << j;
        acc.y |= e.y << j;
    }
    return acc;
}

/* First / last set bit of one class's tile bits inside [az, zz), zz > az, as tile bit numbers (0xffffffff:
 * none). `row`: LDS byte address of the class's word 0. Two windows of three words, one from the block's
 * first word up (masked below az), one from its last word down (masked above zz - 1): whatever a window finds
 * outside the block -- bits of the neighbouring blocks, the zero padding around the row -- fails the range
 * check, so no mask depends on how many words the block covers. Blocks of more than three words whose
 * windows found nothing go on word by word. */
__device__ __forceinline__ void tile_first_last(uint32_t row, uint32_t az, uint32_t zz, uint32_t &f, uint32_t &l) {
    const uint32_t w0 = az >> 6, w1 = (zz - 1u) >> 6;
    const uint32_t fa = row + w0 * 8u, la = row + w1 * 8u;
    const uint64_t a0 = lds_word(fa) & (~0ull << (az & 63u)), a1 = lds_word(fa + 8u), a2 = lds_word(fa + 16u);
    const uint64_t b0 = lds_word(la) & (~0ull >> (63u - ((zz - 1u) & 63u))), b1 = lds_word(la - 8u), b2 = lds_word(la - 16u);
    /* v_ffbl / v_ffbh give 0xffffffff for 0: OR-ing in the dword's base keeps "none" the largest value */
    const uint32_t fr = min(min3u(ffbl((uint32_t)a0), ffbl((uint32_t)(a0 >> 32)) | 32u, ffbl((uint32_t)a1) | 64u),
                            min3u(ffbl((uint32_t)(a1 >> 32)) | 96u, ffbl((uint32_t)a2) | 128u, ffbl((uint32_t)(a2 >> 32)) | 160u));
    const uint32_t lr = min(min3u(ffbh((uint32_t)(b0 >> 32)), ffbh((uint32_t)b0) | 32u, ffbh((uint32_t)(b1 >> 32)) | 64u),
                            min3u(ffbh((uint32_t)b1) | 96u, ffbh((uint32_t)(b2 >> 32)) | 128u, ffbh((uint32_t)b2) | 160u));
    f = w0 * 64u + fr; /* none: far beyond zz */
    f = (fr != 0xffffffffu && f < zz) ? f : 0xffffffffu;
    l = w1 * 64u + 63u - lr;
    l = (lr != 0xffffffffu && (int32_t)l >= (int32_t)az) ? l : 0xffffffffu;
    if (__any(w1 > w0 + 2 && (f == 0xffffffffu || l == 0xffffffffu))) {
        if (w1 > w0 + 2) {
            if (f == 0xffffffffu)
                for (uint32_t w = w0 + 3; w <= w1; w++) {
                    uint64_t v = lds_word(row + w * 8u);
                    if (w == w1) v &= ~0ull >> (63u - ((zz - 1u) & 63u));
                    if (v) {
                        f = w * 64u + (uint32_t)__builtin_ctzll(v);
                        break;
                    }
                }
            if (l == 0xffffffffu && f != 0xffffffffu) /* (no first: no last either) */
                for (uint32_t w = w1 - 3;; w--) {
                    uint64_t v = lds_word(row + w * 8u);
                    if (w == w0) v &= ~0ull << (az & 63u);
                    if (v) {
                        l = w * 64u + 63u - (uint32_t)__builtin_clzll(v);
                        break;
                    }
                    if (w == w0) break;
                }
        }
    }
}

/* number of offsets off[0 .. n) below `target` (off ascending), by the whole wavefront: 64-ary search */
__device__ __forceinline__ uint64_t wave_lower_bound(const uint64_t *off, uint64_t n, uint64_t target, uint32_t lane) {
    uint64_t lo = 0, hi = n; /* answer in [lo, hi] */
    while (hi - lo > 64) {
        const uint64_t step = (hi - lo + 63) / 64;
        const uint64_t probe = min(lo + (uint64_t)lane * step, hi - 1);
        const uint64_t below = __ballot(off[probe] < target); /* a prefix of the lanes */
        const uint32_t k = (uint32_t)__popcll(below);
        if (k == 0) return lo; /* off[lo] >= target */
        const uint64_t nlo = min(lo + (uint64_t)(k - 1) * step, hi - 1) + 1; /* that probe is below: the answer is after it */
        const uint64_t nhi = k < 64 ? min(lo + (uint64_t)k * step, hi - 1) : hi;
        lo = nlo;
        hi = nhi < lo ? lo : nhi;
    }
    const uint64_t idx = lo + lane;
    const uint64_t below = __ballot(idx < hi && off[idx] < target);
    return lo + (uint64_t)__popcll(below);
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), (int)l) << 32 |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}

template <bool FL>
__device__ __forceinline__ void class_tile_body(const uint8_t *corpus, uint64_t total, const uint2 *lut /* [256] */,
                                                uint32_t n_classes, uint16_t *const *bitmaps, int aligned8, const uint64_t *off,
                                                uint64_t nblocks, uint32_t *first, uint32_t *last) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[]; /* [256][32] table, then [16][8][64] tile words */
    for (uint32_t i = threadIdx.x; i < 256 * 32; i += CT_THREADS) {
        const uint2 e = lut[i >> 5];
        lds64[i] = (uint64_t)e.y << 32 | e.x;
    }
    if (FL)
        for (uint32_t i = CT_TABLE_BYTES / 8 + threadIdx.x; i < CT_LDS_BYTES / 8; i += CT_THREADS) lds64[i] = 0; /* row padding */
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t col8 = (lane & 31u) << 3;
    const uint32_t tb = CT_TABLE_BYTES + wave * CT_WAVE_LDS; /* this wavefront's tile words: byte address of [class][lane] */
    /* first / last: one lane per (block, class) pair, 64 >> lg blocks at a time */
    const uint32_t lg = n_classes > 4 ? 3u : n_classes > 2 ? 2u : n_classes > 1 ? 1u : 0u;
    const uint32_t cls = lane & ((1u << lg) - 1u), sub = lane >> lg, per_round = 64u >> lg;
    const uint32_t row = tb + cls * CT_ROW + CT_PAD; /* this lane's class: byte address of tile word 0 */
    const uint32_t part = tb + 8 * CT_ROW + cls * 8u; /* the open block's {first, last} so far for this lane's class */
    const uint64_t cls_off = (uint64_t)cls * nblocks;
    uint16_t *bm[8];
#pragma unroll
    for (int c = 0; c < 8; c++) bm[c] = bitmaps[c];

    const uint64_t n_tiles = (total + CT_TILE - 1) / CT_TILE;
    const uint64_t n_waves = (uint64_t)gridDim.x * (CT_THREADS / 64);
    const uint64_t per_wave = (n_tiles + n_waves - 1) / n_waves;
    const uint64_t wave_global = (uint64_t)blockIdx.x * (CT_THREADS / 64) + wave;
    uint64_t tile = min(n_tiles, wave_global * per_wave);
    const uint64_t tile_end = min(n_tiles, tile + per_wave);
    if (tile >= tile_end) return;

    /* a tile's 4 loads through a buffer descriptor over the tile's whole 16-byte pieces (a ragged last
     * piece is read byte by byte where it is used); a disabled tile gets an empty descriptor: zeros, no
     * memory touched */
    auto issue = [&](uint64_t t, bool enable, uint4 d[4]) {
        const uint64_t base = enable ? t * CT_TILE : 0;
        const uint64_t left = enable ? total - base : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(corpus + base), 0,
                                                                             (int)(min<uint64_t>(left, (uint64_t)CT_TILE) & ~15ull), 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 64u + k * 16u, 0, 0);
            d[k] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    /* the offsets of 64 blocks from `cur` on (lane i: block cur + i), clamped: nothing branches */
    auto window = [&](uint64_t cur, uint64_t &s, uint64_t &e) {
        s = off[min(cur + lane, nblocks)];
        e = off[min(cur + lane + 1, nblocks)];
    };

    uint64_t cursor = 0, ws = 0, we = 0;
    if (FL) {
        cursor = wave_lower_bound(off, nblocks, tile * CT_TILE, lane); /* first block that starts in this share */
        window(cursor, ws, we);
    }
    /* the block that began in an earlier tile and has not ended yet (at most one): wave-uniform; what is
     * known of its first / last members so far (relative to its start) sits in LDS, per class */
    bool open = false;
    uint64_t ob = 0, os = 0, oe = 0;

    /* two tiles in flight ahead of the one being worked on */
    uint4 nxt[4], nx2[4];
    issue(tile, true, nxt);
    issue(tile + 1, tile + 1 < tile_end, nx2);
    bool nx2_loaded = tile + 1 < tile_end;
    for (;; tile++) {
        const bool extra = tile >= tile_end; /* past the share: only to finish the open block */
        if ((extra && !open) || tile >= n_tiles) break; /* (offsets that run past the corpus must not keep a wavefront here) */
        uint4 d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = nxt[k], nxt[k] = nx2[k];
        const bool nxt_loaded = nx2_loaded;
        const uint64_t lo = tile * CT_TILE, hi = lo + CT_TILE;
        const bool last_tile = tile + 1 == n_tiles;

        /* Which blocks start in this tile is known from the offsets alone: count them now and ask for the next
         * tile's window a whole tile of work ahead of its use. The same tells whether a block will still be
         * open after this tile, i.e. whether a tile past the share must be fetched. */
        const uint64_t cur0 = cursor, s0 = ws, e0 = we;
        uint32_t n0 = 0;
        bool open_after = open && oe > hi;
        if (FL && !extra) {
            const bool started = cur0 + lane < nblocks && (last_tile || s0 < hi);
            n0 = (uint32_t)__popcll(__ballot(started));
            if (n0 < 64) {
                cursor = cur0 + n0;
                window(cursor, ws, we);
            }
            if (n0 == 64) open_after = true;
            else if (n0) open_after = open_after || readlane64(e0, n0 - 1) > hi;
        }
        nx2_loaded = tile + 2 < tile_end;
        issue(tile + 2, nx2_loaded, nx2);
        if (tile + 1 >= tile_end && !nxt_loaded) /* a tile past the share, only when a block is still open there */
            issue(tile + 1, tile + 1 < n_tiles && open_after, nxt);

        if (last_tile && (total & 15)) { /* the ragged last piece of the corpus: one lane, byte by byte */
            const uint32_t piece = (uint32_t)((total - lo) >> 4);
            if (lane == piece >> 2) {
                uint32_t r[4] = {0, 0, 0, 0};
                const uint8_t *src = corpus + lo + (uint64_t)piece * 16;
                for (uint32_t i = 0; i < (uint32_t)(total & 15); i++) r[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((piece & 3) == (uint32_t)k) d[k] = make_uint4(r[0], r[1], r[2], r[3]);
            }
        }

        /* classify: 8 accumulators of 8 bytes each, then one 64-bit word per class */
        uint2 a[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            a[2 * k] = classify8(d[k].x, d[k].y, col8);
            a[2 * k + 1] = classify8(d[k].z, d[k].w, col8);
        }
        uint32_t xl[4], xh[4], yl[4], yh[4];
        transpose4(a[0].x, a[1].x, a[2].x, a[3].x, xl);
        transpose4(a[4].x, a[5].x, a[6].x, a[7].x, xh);
        transpose4(a[0].y, a[1].y, a[2].y, a[3].y, yl);
        transpose4(a[4].y, a[5].y, a[6].y, a[7].y, yh);
        uint64_t m[8];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            m[c] = (uint64_t)xh[c] << 32 | xl[c];
            m[4 + c] = (uint64_t)yh[c] << 32 | yl[c];
        }
        const uint64_t lane_base = lo + lane * 64ull;
        if (last_tile) { /* bytes past the end are not members of anything */
            const uint64_t left = lane_base < total ? total - lane_base : 0;
            const uint64_t keep = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
#pragma unroll
            for (int c = 0; c < 8; c++) m[c] &= keep;
        }
        if (!extra) {
            if (aligned8 && !last_tile) {
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if ((uint32_t)c < n_classes) *(uint64_t *)(bm[c] + (lane_base >> 4)) = m[c];
            } else { /* bitmaps hold (total + 15) / 16 16-bit words */
                const uint64_t words = (total + 15) >> 4;
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if ((uint32_t)c < n_classes)
                        for (uint32_t q = 0; q < 4; q++)
                            if ((lane_base >> 4) + q < words) bm[c][(lane_base >> 4) + q] = (uint16_t)(m[c] >> (16 * q));
            }
        }
        if (!FL) continue;

#pragma unroll
        for (int c = 0; c < 8; c++)
            *(__attribute__((address_space(3))) uint64_t *)(uintptr_t)(tb + c * CT_ROW + CT_PAD + lane * 8u) = m[c];
        __builtin_amdgcn_wave_barrier();

        /* the open block's share of this tile, bits [0, z): lanes 0 .. classes - 1 of the first group */
        if (open) {
            const uint32_t z = (uint32_t)(min(oe, hi) - lo);
            const bool mine = sub == 0 && cls < n_classes;
            uint32_t f = 0xffffffffu, l = 0xffffffffu;
            tile_first_last(row, 0, mine ? z : 1u, f, l);
            if (mine) {
                const uint2 sofar = lds_read64(part);
                const uint32_t base = (uint32_t)(lo - os);
                const uint32_t pf = sofar.x != 0xffffffffu ? sofar.x : (f != 0xffffffffu ? base + f : f);
                const uint32_t pl = l != 0xffffffffu ? base + l : sofar.y;
                if (oe <= hi) { /* it ends here */
                    if (first) first[cls_off + ob] = pf == 0xffffffffu ? (uint32_t)(oe - os) : pf;
                    if (last) last[cls_off + ob] = pl;
                } else {
                    *(__attribute__((address_space(3))) uint64_t *)(uintptr_t)part = (uint64_t)pl << 32 | pf;
                }
            }
            if (oe <= hi) open = false;
        }
        if (extra) continue;

        /* the blocks that start in the tile, 64 at a time (more than one round only for blocks under 64 bytes) */
        uint64_t cur = cur0, s = s0, e = e0;
        uint32_t n = n0;
        for (;;) {
            const bool started = cur + lane < nblocks && (last_tile || s < hi);
            /* [s, e) in tile bits; a block that ends in a later tile takes part with what it has here */
            const uint32_t packed = started ? (uint32_t)(s - lo) | (uint32_t)(min(e, hi) - lo) << 16 | (e > hi ? 1u << 30 : 0u)
                                            : 0xffffffffu;
            const uint32_t rounds = (min(n, 64u) + per_round - 1) >> (6 - lg);
            for (uint32_t r = 0; r < rounds; r++) {
                const uint32_t j = r * per_round + sub; /* this lane's block of the window, its class is `cls` */
                const uint32_t pj = (uint32_t)__shfl((int)packed, (int)j);
                const bool active = pj != 0xffffffffu && cls < n_classes;
                const uint32_t az = active ? pj & 0xffffu : 0, zz = active ? (pj >> 16) & 0x3fffu : 0;
                uint32_t f = 0xffffffffu, l = 0xffffffffu;
                const bool some = zz > az;
                tile_first_last(row, some ? az : 0, some ? zz : 1u, f, l);
                if (!some) f = l = 0xffffffffu;
                if (active) {
                    if (pj >> 30) { /* the head of a block that goes on: what is known so far, relative to its start */
                        const uint32_t pf = f == 0xffffffffu ? f : f - az, pl = l == 0xffffffffu ? l : l - az;
                        *(__attribute__((address_space(3))) uint64_t *)(uintptr_t)part = (uint64_t)pl << 32 | pf;
                    } else {
                        const uint64_t at = cls_off + cur + j;
                        if (first) first[at] = f == 0xffffffffu ? zz - az : f - az;
                        if (last) last[at] = l == 0xffffffffu ? l : l - az;
                    }
                }
            }
            /* the one block that starts here and ends in a later tile */
            const uint64_t over = __ballot(started && e > hi);
            if (over) {
                const uint32_t L = (uint32_t)__builtin_ctzll(over);
                open = true;
                ob = cur + L;
                os = readlane64(s, L);
                oe = readlane64(e, L);
            }
            if (n < 64) break;
            cur += 64;
            window(cur, s, e);
            n = (uint32_t)__popcll(__ballot(cur + lane < nblocks && (last_tile || s < hi)));
            if (n < 64) {
                cursor = cur + n;
                window(cursor, ws, we);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* with first / last: 133 KiB of LDS, one workgroup (4 wavefronts per SIMD, up to 128 VGPRs) per CU */
__global__ __launch_bounds__(CT_THREADS) void class_tile_fl_kernel(const uint8_t *corpus, uint64_t total, const uint2 *lut,
                                                                    uint32_t n_classes, uint16_t *const *bitmaps, int aligned8,
                                                                    const uint64_t *off, uint64_t nblocks, uint32_t *first,
                                                                    uint32_t *last) {
    class_tile_body<true>(corpus, total, lut, n_classes, bitmaps, aligned8, off, nblocks, first, last);
}
/* bitmaps only: the 64 KiB table alone. (Two workgroups per CU, 64 VGPRs each, measured 0.52 ms/GiB against 0.45:
 * the kernel is bound by its vector and LDS instructions, not by latency.) */
__global__ __launch_bounds__(CT_THREADS) void class_tile_bm_kernel(const uint8_t *corpus, uint64_t total, const uint2 *lut,
                                                                       uint32_t n_classes, uint16_t *const *bitmaps, int aligned8) {
    class_tile_body<false>(corpus, total, lut, n_classes, bitmaps, aligned8, nullptr, 0, nullptr, nullptr);
}

/* ---- up to 16 classes, membership bitmaps only, ONE read of the corpus (round 4) -----------------------------------
 * The tile kernels above hold 8 classes per pass, and their instruction count is dominated by what first / last need
 * (block windows, open blocks, scalar state that no longer fits the scalar registers: 311 v_readlane in the bitmaps-only
 * instantiation). A caller that wants the bitmaps alone -- the class-sequence patterns of config 4 do -- gets this
 * kernel: the table entry of a byte value is 128 bits, an 8-bit field per class (1 = member), read with ONE ds_read_b128;
 * acc[q] |= entry[q] << j over the 8 bytes of a group leaves every class's 8 membership bits in its field (four
 * v_lshl_or per byte for 16 classes), and 4 x 4 byte transposes (v_perm) turn the eight groups of a lane's 64 bytes into
 * one 64-bit word per class. The table is replicated 8 times (32 KiB): the 8 lanes that share a cycle of a 128-bit LDS
 * read hit 8 different quadruples of banks. Per corpus byte: 1 B read, n_classes / 8 B written, nothing else. */
/* The 16 8-bit fields of 8 corpus bytes: eight ds_read_b128 of the replicated table, 32 shift-ors. A function of its own,
 * NOT inlined: inlined four times into the tile loop, the scheduler moves all 32 reads of a tile (128 result registers) to the
 * front and the kernel spills ~70 dwords per tile whatever the barriers between the groups. */
__device__ __attribute__((noinline)) uint4 b16_classify8(uint32_t d0, uint32_t d1, uint32_t rep) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t d = j < 4 ? d0 : d1;
        const int q = j & 3;
        /* byte q of d, times 128 (8 replicas of 16 bytes), + this lane's replica */
        const uint32_t sh = q == 0 ? d << 7 : d >> (8 * q - 7);
        const v4u e = *(const __attribute__((address_space(3))) v4u *)(uintptr_t)((sh & 0x7f80u) | rep); /* one ds_read_b128 */
        acc[0] |= e[0] << j, acc[1] |= e[1] << j, acc[2] |= e[2] << j, acc[3] |= e[3] << j;
    }
    return make_uint4(acc[0], acc[1], acc[2], acc[3]);
}

constexpr int B16_THREADS = 256;
constexpr uint32_t B16_TABLE_BYTES = 256 * 8 * 16;
constexpr uint32_t B16_TILE = 2048; /* bytes per wavefront tile: 32 per lane -> one dword of every class's bitmap per lane */

/* (registers for five wavefronts per SIMD = five 32 KiB workgroups per CU; FOUR are launched since round 5, which is faster:
 * HSGPU_B16_WG_PER_CU below. A 4 KiB tile -- 64 bytes and a 64-bit word per class and lane --
 * needs ~110 registers and spilled a hundred dwords at 96; 32 bytes per lane need 60.) */
__global__ __launch_bounds__(B16_THREADS) __attribute__((amdgpu_waves_per_eu(5, 8))) void class_bitmap16_kernel(
    const uint8_t *corpus, uint64_t total, const uint4 *lut /* [256] */, uint32_t n_classes, uint16_t *const *bitmaps) {
    extern __shared__ __attribute__((aligned(16))) uint4 b16_tab[]; /* [256][8] */
    for (uint32_t i = threadIdx.x; i < 256 * 8; i += B16_THREADS) b16_tab[i] = lut[i >> 3];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rep = (lane & 7u) << 4;
    const uint64_t n_tiles = (total + B16_TILE - 1) / B16_TILE;
    const uint64_t n_waves = (uint64_t)gridDim.x * (B16_THREADS / 64);
    const uint64_t per_wave = (n_tiles + n_waves - 1) / n_waves;
    const uint64_t wave_global = (uint64_t)blockIdx.x * (B16_THREADS / 64) + wave;
    uint64_t tile = min(n_tiles, wave_global * per_wave);
    const uint64_t tile_end = min(n_tiles, tile + per_wave);
    if (tile >= tile_end) return;
    const uint64_t words16 = (total + 15) >> 4; /* what a bitmap holds */

    auto issue = [&](uint64_t t, bool enable, uint4 d[2]) { /* (a disabled tile gets an empty descriptor: zeros, no memory touched) */
        const uint64_t base = enable ? t * B16_TILE : 0;
        const uint64_t left = enable ? total - base : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(corpus + base), 0,
                                                                             (int)(min<uint64_t>(left, (uint64_t)B16_TILE) & ~15ull), 0x00020000);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 32u + k * 16u, 0, 0);
            d[k] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto classify = [&](uint32_t d0, uint32_t d1, uint32_t acc[4]) {
        const uint4 r = b16_classify8(d0, d1, rep);
        acc[0] = r.x, acc[1] = r.y, acc[2] = r.z, acc[3] = r.w;
    };
    uint4 nxt[2]; /* (one tile ahead; five wavefronts per SIMD cover the rest of the latency) */
    issue(tile, true, nxt);
    for (; tile < tile_end; tile++) {
        uint4 d[2];
#pragma unroll
        for (int k = 0; k < 2; k++) d[k] = nxt[k];
        issue(tile + 1, tile + 1 < tile_end, nxt);
        const uint64_t lo = tile * B16_TILE;
        const bool last_tile = tile + 1 == n_tiles;
        if (last_tile && (total & 15)) { /* the ragged last piece of the corpus: one lane, byte by byte */
            const uint32_t piece = (uint32_t)((total - lo) >> 4);
            if (lane == piece >> 1) {
                uint32_t r[4] = {0, 0, 0, 0};
                const uint8_t *src = corpus + lo + (uint64_t)piece * 16;
                for (uint32_t i = 0; i < (uint32_t)(total & 15); i++) r[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
                if (piece & 1) d[1] = make_uint4(r[0], r[1], r[2], r[3]);
                else d[0] = make_uint4(r[0], r[1], r[2], r[3]);
            }
        }
        uint32_t A[4][4];
        classify(d[0].x, d[0].y, A[0]);
        classify(d[0].z, d[0].w, A[1]);
        classify(d[1].x, d[1].y, A[2]);
        classify(d[1].z, d[1].w, A[3]);
        const uint64_t lane_base = lo + lane * 32ull;
        uint32_t keep = ~0u;
        if (last_tile) { /* bytes past the end are not members of anything */
            const uint64_t left = lane_base < total ? total - lane_base : 0;
            keep = left >= 32 ? ~0u : ((1u << left) - 1u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { /* four classes at a time: transposed and stored (their words do not wait for the others') */
            uint32_t w[4];
            transpose4(A[0][q], A[1][q], A[2][q], A[3][q], w);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int c = 4 * q + r;
                if ((uint32_t)c >= n_classes) continue;
                const uint32_t m = w[r] & keep;
                uint16_t *bm = bitmaps[c];
                if (!last_tile) {
                    *(uint32_t *)(bm + (lane_base >> 4)) = m;
                } else {
                    for (uint32_t k = 0; k < 2; k++)
                        if ((lane_base >> 4) + k < words16) bm[(lane_base >> 4) + k] = (uint16_t)(m >> (16 * k));
                }
            }
        }
    }
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
    /* (first / last: 8 classes per call; bitmaps alone: 16) */
    if (!classes || n_classes == 0 || n_classes > ((d_first || d_last) ? HSGPU_CLASS_MAX : HSGPU_CLASS_MAX_BITMAPS) || !d_bitmaps || !d_work)
        return HSGPU_INVALID;
    if (((uintptr_t)d_corpus & 15) || ((uintptr_t)d_work & 15)) return HSGPU_INVALID;
    if ((d_first || d_last) && (!d_off || nblocks == 0)) return HSGPU_INVALID;
    hipStream_t st = (hipStream_t)stream;
    bool all_aligned8 = true;
    for (unsigned c = 0; c < n_classes; c++) {
        if (!d_bitmaps[c] || ((uintptr_t)d_bitmaps[c] & 1)) return HSGPU_INVALID;
        all_aligned8 = all_aligned8 && !((uintptr_t)d_bitmaps[c] & 7);
    }
    if (!d_first && !d_last && (n_classes > HSGPU_CLASS_MAX || all_aligned8)) {
        /* bitmaps alone: the 16-class kernel, one read of the corpus whatever the number of classes */
        if (!all_aligned8) {
            hsgpu_set_error("more than %d class bitmaps per call need 8-byte aligned bitmaps", HSGPU_CLASS_MAX);
            return HSGPU_INVALID;
        }
        uint32_t lut16[256][4];
        memset(lut16, 0, sizeof(lut16));
        for (unsigned c = 0; c < n_classes; c++)
            for (unsigned v = 0; v < 256; v++)
                if (classes[c].bitmap[v >> 3] >> (v & 7) & 1) lut16[v][c >> 2] |= 1u << (8 * (c & 3));
        uint8_t *work = (uint8_t *)d_work;
        void *ptrs16[HSGPU_CLASS_MAX_BITMAPS] = {nullptr};
        for (unsigned c = 0; c < n_classes; c++) ptrs16[c] = d_bitmaps[c];
        static_assert(sizeof(lut16) + sizeof(ptrs16) <= HSGPU_CLASS_WORK_BYTES, "work area layout");
        HIP_TRY(hipMemcpyAsync(work, lut16, sizeof(lut16), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(work + sizeof(lut16), ptrs16, sizeof(ptrs16), hipMemcpyHostToDevice, st));
        if (total_bytes == 0) return HSGPU_SUCCESS;
        int dev = 0, n_cu = 256;
        HIP_TRY(hipGetDevice(&dev));
        n_cu = cached_cu_count(dev);
        const uint64_t n_tiles = (total_bytes + B16_TILE - 1) / B16_TILE;
        /* four 32 KiB workgroups of four wavefronts per CU (five fit) */
#ifndef HSGPU_B16_WG_PER_CU
#define HSGPU_B16_WG_PER_CU 4 /* 12 classes, bitmaps only, one box: 2 workgroups per CU 0.729 ms per GiB, 3 0.590, 4 0.554, 5 (rounds 4-5: what the LDS holds) 0.659;
                                * 4 GiB: 5 2.451 ms, 4 1.992 (0.674 of the roofline), 3 2.088: profiles/r05_wg_threads_sweep.txt */
#endif
        const unsigned grid = (unsigned)std::min<uint64_t>((n_tiles + 3) / 4, (uint64_t)n_cu * HSGPU_B16_WG_PER_CU);
        hipLaunchKernelGGL(class_bitmap16_kernel, dim3(grid), dim3(B16_THREADS), B16_TABLE_BYTES, st, (const uint8_t *)d_corpus, total_bytes,
                           (const uint4 *)work, n_classes, (uint16_t *const *)(work + sizeof(lut16)));
        HIP_TRY(hipGetLastError());
        return HSGPU_SUCCESS;
    }
    /* host-built 256-entry table: entry[v] field c (8 bits) = 1 iff v in class c */
    uint32_t lut[256][2];
    memset(lut, 0, sizeof(lut));
    for (unsigned c = 0; c < n_classes; c++)
        for (unsigned v = 0; v < 256; v++)
            if (classes[c].bitmap[v >> 3] >> (v & 7) & 1) lut[v][c >> 2] |= 1u << (8 * (c & 3));
    /* d_work: [2048 B table][64 B pointer array] supplied by the caller (no hidden allocation) */
    uint8_t *work = (uint8_t *)d_work;
    void *ptrs[8] = {nullptr};
    int aligned8 = 1;
    for (unsigned c = 0; c < n_classes; c++) {
        if (!d_bitmaps[c] || ((uintptr_t)d_bitmaps[c] & 1)) return HSGPU_INVALID;
        ptrs[c] = d_bitmaps[c];
        if ((uintptr_t)d_bitmaps[c] & 7) aligned8 = 0;
    }
    static_assert(sizeof(lut) + sizeof(ptrs) <= HSGPU_CLASS_WORK_BYTES, "work area layout");
    HIP_TRY(hipMemcpyAsync(work, lut, sizeof(lut), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(work + sizeof(lut), ptrs, sizeof(ptrs), hipMemcpyHostToDevice, st));
    if (total_bytes == 0) return HSGPU_SUCCESS;
    const uint2 *d_lut = (const uint2 *)work;
    uint16_t *const *d_ptrs = (uint16_t *const *)(work + sizeof(lut));
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    n_cu = cached_cu_count(dev);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)class_tile_fl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)CT_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)class_tile_bm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)CT_TABLE_BYTES));
        attr_set = true;
    }
    const uint64_t n_tiles = (total_bytes + CT_TILE - 1) / CT_TILE;
    const unsigned grid = (unsigned)std::min<uint64_t>((n_tiles + 15) / 16, (uint64_t)n_cu); /* one workgroup per CU */
    const uint8_t *corpus = (const uint8_t *)d_corpus;
    if (d_first || d_last) {
        hipLaunchKernelGGL(class_tile_fl_kernel, dim3(grid), dim3(CT_THREADS), CT_LDS_BYTES, st, corpus, total_bytes, d_lut,
                           n_classes, d_ptrs, aligned8, (const uint64_t *)d_off, nblocks, (uint32_t *)d_first, (uint32_t *)d_last);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(class_tile_bm_kernel, dim3(grid), dim3(CT_THREADS), CT_TABLE_BYTES, st, corpus, total_bytes, d_lut,
                           n_classes, d_ptrs, aligned8);
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
    n_cu = cached_cu_count(dev);
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
    if (!aux || !d_off || nblocks == 0 || !d_start_out) return HSGPU_INVALID;
    if (aux->type != HSGPU_ACCEL_NONE && !d_work) return HSGPU_INVALID; /* (ACCEL_NONE uses neither the bitmap nor the work area) */
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

/* ---- run_accel for a block batch (src/nfa/accel.c:35-146) --------------------------------------
 * out[b] = run_accel(aux, buf + start, buf + len) - buf for every block: the ten cases the reference dispatches,
 * their "too short to bother" thresholds (c + 15 >= c_end for the one-byte schemes, c + 16 + 1 for double
 * vermicelli, c + 15 + 1 for double shufti: the block keeps its start), the two-byte schemes searched over
 * [c, c_end - 1) ("need to stop one early to get an accurate end state", accel.c:68,79,90,122), ACCEL_RED_TAPE = c_end,
 * and the offset adjustment max(c + offset, rv) - offset (accel.c:138-141). */
enum { RA_KEEP = 0, RA_SINGLE = 1, RA_PAIR = 2, RA_END = 3 };
__global__ void run_accel_kernel(const uint8_t *corpus, const uint64_t *off, uint64_t nblocks, const uint16_t *bm,
                                 const uint4 *lut, int kind, uint32_t min_tail, uint32_t offset, const uint32_t *start_in,
                                 uint32_t start_all, uint32_t *out) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint64_t lo = off[b], hi = off[b + 1], len = hi - lo;
    const uint64_t s = start_in ? start_in[b] : start_all;
    if (kind == RA_KEEP || s > len) {
        out[b] = (uint32_t)s;
        return;
    }
    uint64_t rv;
    if (kind == RA_END) {
        rv = len;
    } else {
        if (s + min_tail >= len) { /* c + 15 (+ 1 (+ 1)) >= c_end */
            out[b] = (uint32_t)s;
            return;
        }
        if (kind == RA_SINGLE) {
            const uint32_t f = first_bit(bm, lo + s, hi);
            rv = f == 0xffffffffu ? len : s + f;
        } else { /* pairs inside [s, len - 1): the second byte of a pair lies before len - 1 */
            const uint64_t end = len - 1;
            const uint32_t f = first_bit(bm, lo + s, lo + end - 1);
            if (f != 0xffffffffu) {
                rv = s + f;
            } else { /* the last byte of the range alone passing the first-byte test: reported for re-examination */
                const uint32_t *t1 = (const uint32_t *)&lut[2 * corpus[lo + end - 1]];
                rv = (t1[0] & 0xffu) != 0xffu ? end - 1 : end;
            }
        }
    }
    rv = max(s + offset, rv) - offset;
    out[b] = (uint32_t)rv;
}

extern "C" int hsgpu_run_accel_dev(const hsgpu_accel_aux_t *aux, const void *d_corpus, uint64_t total_bytes,
                                   const void *d_off, uint64_t nblocks, const void *d_start_in, uint32_t start,
                                   void *d_out, void *d_bitmap, void *d_work, void *stream) {
    if (!aux || !d_off || nblocks == 0 || !d_out) return HSGPU_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int kind = RA_KEEP, rv = HSGPU_SUCCESS;
    uint32_t min_tail = 15;
    void *bitmaps[1] = {d_bitmap};
    hsgpu_class_t cls;
    hsgpu_pair_t pr;
    switch (aux->accel_type) {
    case HSGPU_ACCEL_NONE:
        break;
    case HSGPU_ACCEL_RED_TAPE:
        kind = RA_END;
        break;
    case HSGPU_ACCEL_VERM:
    case HSGPU_ACCEL_VERM_NOCASE:
        kind = RA_SINGLE;
        rv = hsgpu_class_from_verm(aux->c1, aux->accel_type == HSGPU_ACCEL_VERM_NOCASE, 0, &cls);
        break;
    case HSGPU_ACCEL_SHUFTI:
        kind = RA_SINGLE;
        rv = hsgpu_class_from_shufti(aux->mask[0], aux->mask[1], &cls);
        break;
    case HSGPU_ACCEL_TRUFFLE:
        kind = RA_SINGLE;
        rv = hsgpu_class_from_truffle(aux->mask[0], aux->mask[1], &cls);
        break;
    case HSGPU_ACCEL_DVERM:
    case HSGPU_ACCEL_DVERM_NOCASE:
        kind = RA_PAIR, min_tail = 17;
        rv = hsgpu_pair_from_dverm(aux->c1, aux->c2, aux->accel_type == HSGPU_ACCEL_DVERM_NOCASE, &pr);
        break;
    case HSGPU_ACCEL_DVERM_MASKED:
        kind = RA_PAIR, min_tail = 17;
        rv = hsgpu_pair_from_dverm_masked(aux->c1, aux->c2, aux->m1, aux->m2, &pr);
        break;
    case HSGPU_ACCEL_DSHUFTI:
        kind = RA_PAIR, min_tail = 16;
        rv = hsgpu_pair_from_dshufti(aux->mask[0], aux->mask[1], aux->mask[2], aux->mask[3], &pr);
        break;
    default: /* the reverse and EOD types are not dispatched by run_accel either (accel.c:132-135: "not here") */
        hsgpu_set_error("accel type %u is not one run_accel dispatches", aux->accel_type);
        return HSGPU_INVALID;
    }
    if (rv != HSGPU_SUCCESS) return rv;
    if (kind == RA_SINGLE || kind == RA_PAIR) {
        if (!d_bitmap || !d_work || !d_corpus) return HSGPU_INVALID;
        rv = kind == RA_SINGLE
                 ? hsgpu_class_scan_dev(&cls, 1, d_corpus, total_bytes, nullptr, 0, bitmaps, nullptr, nullptr, d_work, stream)
                 : hsgpu_pair_scan_dev(&pr, 1, d_corpus, total_bytes, nullptr, 0, bitmaps, nullptr, nullptr, d_work, stream);
        if (rv != HSGPU_SUCCESS) return rv;
    }
    hipLaunchKernelGGL(run_accel_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_corpus,
                       (const uint64_t *)d_off, nblocks, (const uint16_t *)d_bitmap, (const uint4 *)d_work, kind, min_tail,
                       (uint32_t)aux->offset, (const uint32_t *)d_start_in, start, (uint32_t *)d_out);
    HIP_TRY(hipGetLastError());
    return HSGPU_SUCCESS;
}
