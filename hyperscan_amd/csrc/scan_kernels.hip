/*
 * scan_kernels.hip -- the block-mode multi-literal scan kernel for gfx950.
 *
 * Replaces, on the GPU, the reference's FDR / Teddy / Noodle main loops and
 * their confirm step:
 *   FDR_MAIN_LOOP + get_conf_stride_N   src/fdr/fdr.c:157-327,694-723
 *   prep_conf_teddy_m1..m4, CONFIRM_TEDDY src/fdr/teddy.c:893-1064
 *   noodle scan/final                   src/hwlm/noodle_engine.c:113-138
 *   do_confirm_fdr / confWithBit        src/fdr/fdr.c:330-364,
 *                                       src/fdr/fdr_confirm_runtime.h:43-102
 * It is a new design, not a translation: there are no buckets, no shift-or
 * state and no zones. See DESIGN.md "Kernel".
 *
 * Mapping. The corpus is the concatenation of all blocks (CSR offsets). A
 * workgroup of 16 wavefronts owns a 16 KiB super-tile per iteration; each
 * wavefront owns 1 KiB of it, each lane one 16-byte chunk, loaded with one
 * coalesced global_load_dwordx4. The three bytes in front of a chunk come from
 * the neighbouring lane (cross-lane move), not from memory.
 *
 * Filter (per byte position, all lanes): hash the 3-byte suffix with one
 * v_mul_u32_u24, read ONE 32-bit word of the LDS-resident filter, test one bit
 * chosen by the 4th byte. Candidates are collected as one 16-bit mask per lane.
 *
 * Confirm (per candidate, compacted): lanes with a non-empty mask push
 * {chunk, mask} into a per-wavefront LDS queue; whenever 64 entries are queued
 * the wavefront drains them with all lanes busy: exact hash-table lookup of the
 * suffix in HBM/L2, (window & msk) == v per listed literal, block lookup and
 * bound checks, then an atomically reserved 16-byte record store.
 *
 * Block boundaries are invisible to the filter; the confirm step resolves the
 * block of a candidate by binary search over the offsets and rejects matches
 * that would start before their block (or before `start` within it).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernels.h"
#include "table.h"

namespace {

constexpr int WG_THREADS = HSGPU_WG_THREADS;
constexpr int WAVES = WG_THREADS / 64;
constexpr int CHUNK = 16;                   /* bytes per lane per iteration */
constexpr int WAVE_TILE = 64 * CHUNK;       /* 1 KiB */
constexpr int SUPER_TILE = WAVES * WAVE_TILE; /* 16 KiB */
constexpr int QCAP = 128;                   /* queue entries per wavefront */

__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { return __umul24(a, b); }
__device__ __forceinline__ uint32_t bfe1(uint32_t v, uint32_t bit) { return __builtin_amdgcn_ubfe(v, bit, 1); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n) {
    return __builtin_amdgcn_alignbyte(hi, lo, n);
}

/* 32-bit read at an absolute LDS byte address. The kernel has no static
 * __shared__ data, so its dynamic LDS segment starts at LDS address 0 (checked
 * once at kernel entry); addressing it absolutely saves the per-lookup
 * "v_add base" the compiler otherwise emits. */
typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_word(uint32_t byte_addr) { return *(lds_u32_t *)(uintptr_t)byte_addr; }

struct Tables {
    const HsgpuHtSlot *ht_a, *ht_b;
    const uint32_t *c2ref, *lists;
    const HsgpuDevLit *lits;
    uint32_t ht_a_log2, ht_b_log2;
};

/* the 8 bytes ending at g, little-endian, bytes before the corpus read as 0
 * (conf_key of fdr.c:360 in block mode). Aligned dword loads + funnel shift. */
__device__ __forceinline__ uint64_t window8(const uint8_t *corpus, uint64_t g) {
    if (g < 7) {
        uint64_t w = 0;
        for (uint64_t i = 0; i <= g; i++) w |= (uint64_t)corpus[i] << (8 * (7 - g + i));
        return w;
    }
    uint64_t first = g - 7;
    const uint32_t *p = (const uint32_t *)(corpus + (first & ~3ull));
    uint32_t sh = (uint32_t)(first & 3);
    uint32_t d0 = p[0], d1 = p[1];
    if (sh == 0) return (uint64_t)d1 << 32 | d0;
    uint32_t d2 = p[2];
    uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh * 8);
    uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh * 8);
    return (uint64_t)hi << 32 | lo;
}

/* index of the block containing corpus offset g: last b with off[b] <= g */
__device__ __forceinline__ uint64_t find_block(const uint64_t *off, uint64_t nblocks, uint64_t g) {
    uint64_t lo = 0, hi = nblocks; /* invariant: off[lo] <= g < off[hi] */
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void walk_list(const HsgpuScanArgs &args, const Tables &t, uint32_t ref, uint64_t w,
                                          uint64_t g) {
    uint32_t i = ref - 1;
    uint32_t e;
    do {
        e = t.lists[i++];
        uint32_t li = e & ~HSGPU_LIST_END;
        const HsgpuDevLit lit = t.lits[li];
        if ((w & lit.msk) != lit.v) continue;
        uint64_t b = find_block(args.off, args.nblocks, g);
        uint64_t end = g - args.off[b];
        /* left bound (fdr_confirm_runtime.h:77-88) and `start` (hwlm.h:108-111) */
        if (end + 1 < lit.size || end + 1 - lit.size < args.start) continue;
        unsigned long long slot = atomicAdd(args.count, 1ull);
        if (slot < args.cap) {
            uint4 rec = make_uint4((uint32_t)b, (uint32_t)end, lit.id, li);
            ((uint4 *)args.out)[slot] = rec;
        }
    } while (!(e & HSGPU_LIST_END));
}

__device__ __forceinline__ void probe(const HsgpuScanArgs &args, const Tables &t, const HsgpuHtSlot *ht,
                                      uint32_t log2, uint32_t key, uint64_t w, uint64_t g) {
    uint32_t mask = (1u << log2) - 1;
    uint32_t s = hsgpu_ht_slot(key, log2);
    for (;;) {
        HsgpuHtSlot sl = ht[s];
        if (!sl.ref) return;
        if (sl.key == key) {
            walk_list(args, t, sl.ref, w, g);
            return;
        }
        s = (s + 1) & mask;
    }
}

template <bool HAS_A, bool HAS_B, bool HAS_C>
__device__ __forceinline__ void confirm(const HsgpuScanArgs &args, const Tables &t, uint64_t g) {
    uint64_t w = window8(args.corpus, g);
    uint32_t w4 = (uint32_t)(w >> 32);
    if (HAS_A) probe(args, t, t.ht_a, t.ht_a_log2, w4, w, g);
    if (HAS_B) probe(args, t, t.ht_b, t.ht_b_log2, w4 >> 8, w, g);
    if (HAS_C) {
        uint32_t ref = t.c2ref[w4 >> 16];
        if (ref) walk_list(args, t, ref, w, g);
    }
}

template <bool HAS_A, bool HAS_B, bool HAS_C>
__device__ __forceinline__ void drain_entry(const HsgpuScanArgs &args, const Tables &t, uint2 e) {
    uint32_t m = e.y;
    while (m) {
        uint32_t j = __builtin_ctz(m);
        m &= m - 1;
        confirm<HAS_A, HAS_B, HAS_C>(args, t, (uint64_t)e.x * CHUNK + j);
    }
}

template <bool HAS_A, bool HAS_B, bool HAS_C>
__global__ __launch_bounds__(WG_THREADS) void hwlm_scan_kernel(HsgpuScanArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const HsgpuTableHeader *hdr = (const HsgpuTableHeader *)args.blob;
    const uint32_t k = hdr->filter_log2_words;
    const uint32_t nw = 1u << k;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds != 0) __builtin_trap();
    uint32_t *filter = lds;
    const uint32_t c2_base = nw * 4;
    uint32_t *c2bits = lds + nw;
    uint2 *queues = (uint2 *)(lds + nw + (HAS_C ? 2048 : 0));

    /* stage the filter(s) in LDS: once per workgroup, 16 B per lane per step */
    {
        const uint4 *src = (const uint4 *)(args.blob + hdr->off_filter);
        for (uint32_t i = threadIdx.x; i < nw / 4; i += WG_THREADS) ((uint4 *)filter)[i] = src[i];
        if (HAS_C) {
            const uint4 *src2 = (const uint4 *)(args.blob + hdr->off_c2bits);
            for (uint32_t i = threadIdx.x; i < 512; i += WG_THREADS) ((uint4 *)c2bits)[i] = src2[i];
        }
    }
    __syncthreads();

    Tables t;
    t.ht_a = (const HsgpuHtSlot *)(args.blob + hdr->off_ht_a);
    t.ht_b = (const HsgpuHtSlot *)(args.blob + hdr->off_ht_b);
    t.c2ref = (const uint32_t *)(args.blob + hdr->off_c2ref);
    t.lists = (const uint32_t *)(args.blob + hdr->off_lists);
    t.lits = (const HsgpuDevLit *)(args.blob + hdr->off_lits);
    t.ht_a_log2 = hdr->ht_a_log2;
    t.ht_b_log2 = hdr->ht_b_log2;

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint2 *queue = queues + wave * QCAP;
    uint32_t qcount = 0;

    const uint32_t shift = 30u - k;
    const uint32_t amask = (nw - 1u) << 2;
    const uint8_t *corpus = args.corpus;
    const uint64_t total = args.total;
    const uint64_t n_tiles = (total + SUPER_TILE - 1) / SUPER_TILE;

    auto load_chunk = [&](uint64_t tile) -> uint4 {
        uint64_t off = tile * SUPER_TILE + (uint64_t)wave * WAVE_TILE + lane * CHUNK;
        if (off + CHUNK <= total) return *(const uint4 *)(corpus + off);
        uint4 r = make_uint4(0, 0, 0, 0);
        if (off < total) {
            uint32_t tmp[4] = {0, 0, 0, 0};
            uint32_t n = (uint32_t)(total - off);
            for (uint32_t i = 0; i < n; i++) tmp[i >> 2] |= (uint32_t)corpus[off + i] << (8 * (i & 3));
            r = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
        return r;
    };

    uint64_t tile = blockIdx.x;
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (tile < n_tiles) cur = load_chunk(tile);

    for (; tile < n_tiles; tile += gridDim.x) {
        const uint64_t wbase = tile * SUPER_TILE + (uint64_t)wave * WAVE_TILE;
        const uint64_t coff = wbase + lane * CHUNK;
        /* prefetch the next tile's chunk while this one is processed */
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (tile + gridDim.x < n_tiles) nxt = load_chunk(tile + gridDim.x);

        /* the dword in front of this chunk: neighbour lane's last dword; lane 0
         * takes the last dword of the previous 1 KiB (zero at corpus start). */
        uint32_t dp = __shfl_up(cur.w, 1);
        if (lane == 0) dp = (wbase >= 4 && wbase <= total) ? *(const uint32_t *)(corpus + wbase - 4) : 0;

        const uint32_t d0 = cur.x, d1 = cur.y, d2 = cur.z, d3 = cur.w;
        /* x[j] = bytes c[j-2 .. j+1] of the chunk (c[-4..-1] = dp); x[-1] too */
        uint32_t x[17];
        x[0] = alignbyte(d0, dp, 1);  /* x[-1]: c[-3..0] */
        x[1] = alignbyte(d0, dp, 2);  /* j=0 */
        x[2] = alignbyte(d0, dp, 3);
        x[3] = d0;
        x[4] = alignbyte(d1, d0, 1);
        x[5] = alignbyte(d1, d0, 2);
        x[6] = alignbyte(d1, d0, 3);
        x[7] = d1;
        x[8] = alignbyte(d2, d1, 1);
        x[9] = alignbyte(d2, d1, 2);
        x[10] = alignbyte(d2, d1, 3);
        x[11] = d2;
        x[12] = alignbyte(d3, d2, 1);
        x[13] = alignbyte(d3, d2, 2);
        x[14] = alignbyte(d3, d2, 3);
        x[15] = d3;
        x[16] = d3 >> 8; /* j=15: c[13..15] */

        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t xj = x[j + 1];
            const uint32_t prod = mul_u24(xj, HSGPU_FILTER_MUL);
            const uint32_t a = prod >> shift;
            const uint32_t word = lds_word(a & amask);
            uint32_t hit = 0;
            if (HAS_A) hit = bfe1(word, x[j] + a);
            if (HAS_B) hit |= bfe1(word, prod >> 8);
            if (HAS_C) {
                const uint32_t kc = __builtin_amdgcn_ubfe(xj, 8, 16);
                hit |= bfe1(lds_word(c2_base + ((kc >> 5) << 2)), kc);
            }
            acc |= hit << j;
        }
        /* positions at/after the end of the corpus are not positions */
        if (coff + CHUNK > total) acc &= (coff < total) ? ((1u << (uint32_t)(total - coff)) - 1u) : 0u;

        /* enqueue one {chunk, mask} entry per lane with candidates */
        const bool has = acc != 0;
        const unsigned long long bal = __ballot(has);
        if (bal) {
            if (has) {
                uint32_t idx = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                queue[idx] = make_uint2((uint32_t)(coff >> 4), acc);
            }
            qcount += __popcll(bal);
            if (qcount >= 64) {
                qcount -= 64;
                drain_entry<HAS_A, HAS_B, HAS_C>(args, t, queue[qcount + lane]);
            }
        }
        cur = nxt;
    }
    if (lane < qcount) drain_entry<HAS_A, HAS_B, HAS_C>(args, t, queue[lane]);
}

} // namespace

#define HSGPU_INST(A, B, C) \
    case ((A ? 1 : 0) | (B ? 2 : 0) | (C ? 4 : 0)): return (const void *)hwlm_scan_kernel<A, B, C>;

const void *hsgpu_scan_kernel_for(uint32_t flags) {
    switch (flags & 7u) {
        HSGPU_INST(true, false, false)
        HSGPU_INST(false, true, false)
        HSGPU_INST(true, true, false)
        HSGPU_INST(false, false, true)
        HSGPU_INST(true, false, true)
        HSGPU_INST(false, true, true)
        HSGPU_INST(true, true, true)
    default: return nullptr;
    }
}

size_t hsgpu_scan_lds_bytes(uint32_t flags, uint32_t filter_log2_words) {
    size_t words = ((size_t)1 << filter_log2_words) + ((flags & HSGPU_F_HAS_C) ? 2048 : 0);
    return words * 4 + (size_t)WAVES * QCAP * sizeof(uint2);
}

uint32_t hsgpu_scan_super_tile(void) { return SUPER_TILE; }
