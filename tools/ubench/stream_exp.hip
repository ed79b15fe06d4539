// stream_exp.hip -- experiments on the streaming skeleton of the literal-scan filter kernel (tuning tool,
// not part of the library): the same tile loop (buffer loads through wave-uniform descriptors, STAGES tiles
// in flight per wavefront, 1 KiB per wavefront per tile) around the library's own chunk filters
// (scan_device.h), with the knobs that are compile-time in the product exposed as template parameters.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../hyperscan_amd/csrc -o stream_exp stream_exp.hip
#include "scan_device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct ExpArgs {
    const uint8_t *corpus;
    uint64_t total;
    const uint32_t *table; /* 128 KiB filter image */
    uint32_t super_shift, flog2, hmask;
    unsigned long long *hits;
    uint4 *cand; /* SPILL: candidate regions as in the product, cand_cap entries per wavefront */
    uint32_t cand_cap;
};

// MODE 0: no lookups (xor the bytes), 1: classic stride-2 one-bit (teddy64 shape), 2: classic stride-1 two-bit
// blind (fdr10k shape), 3: pair filter.  AUX: buffer-load cache policy bits.  EARLY: first tiles requested
// before the filter image is staged in LDS.
template <int STAGES, int AUX, int MODE, bool EARLY, int MAP = 0, bool SPILL = false>
__global__ __launch_bounds__(1024) void exp_kernel(ExpArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t super_shift = args.super_shift;
    const uint8_t *corpus = args.corpus;
    const uint64_t n_full = args.total >> super_shift;
    const uint32_t G = gridDim.x;
    const uint32_t lane_off = wave * WAVE_TILE + lane * CHUNK;
    const uint64_t n_wt = args.total >> 10, n_wv = (uint64_t)G * (blockDim.x >> 6), per_wv = (n_wt + n_wv - 1) / n_wv;
    const uint64_t wv_global = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const uint64_t tile_end_ = MAP == 2 ? min(n_wt, (wv_global + 1) * per_wv)
                               : MAP == 1 ? min(n_full, (uint64_t)(blockIdx.x + 1) * ((n_full + G - 1) / G)) : n_full;
    const uint32_t lane_off2 = lane * CHUNK;
    auto issue = [&](uint64_t tile) -> Chunk {
        if (MAP == 2) { /* tile counts 1 KiB wavefront tiles here */
            const uint8_t *base = corpus + (tile << 10) - 8;
            const int records = (tile < tile_end_ && tile) ? (int)0x7ffffff0 : 0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, records, 0x00020000);
            Chunk c;
            const auto d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off2, 8, AUX);
            c.d = make_uint4(d[0], d[1], d[2], d[3]);
            const auto h = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off2, 0, AUX);
            c.h = make_uint2(h[0], h[1]);
            return c;
        }
        const uint8_t *base = corpus + (tile << super_shift) - 8;
        const int records = (tile < tile_end_ && tile) ? (int)0x7ffffff0 : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, records, 0x00020000);
        Chunk c;
        const auto d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, 8, AUX);
        c.d = make_uint4(d[0], d[1], d[2], d[3]);
        const auto h = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, 0, AUX);
        c.h = make_uint2(h[0], h[1]);
        return c;
    };
    Chunk c[STAGES];
    /* MAP 0: tile = blockIdx + k * G (all workgroups walk the corpus together); MAP 1: every workgroup streams
     * its own contiguous share (tile = blockIdx * per + k) */
    const uint64_t per = (n_full + G - 1) / G;
    const uint64_t step = MAP ? 1 : G;
    const uint64_t tile_end = MAP == 2 ? tile_end_ : MAP == 1 ? min(n_full, (uint64_t)(blockIdx.x + 1) * per) : n_full;
    uint64_t tile = MAP == 2 ? max((uint64_t)1, wv_global * per_wv)
                    : MAP == 1 ? max((uint64_t)1, (uint64_t)blockIdx.x * per) : blockIdx.x + G; /* tile 0 skipped */
    if (EARLY) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++) c[s] = issue(tile + (uint64_t)s * step);
    }
    const uint32_t nw = 32768;
    for (uint32_t i = threadIdx.x; i < nw / 4; i += blockDim.x) ((uint4 *)lds)[i] = ((const uint4 *)args.table)[i];
    __syncthreads();
    if (!EARLY) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; s++) c[s] = issue(tile + (uint64_t)s * step);
    }
    FilterCfg f;
    f.shift = MODE == 3 ? 29u - args.flog2 : 30u - args.flog2;
    f.amask = MODE == 3 ? ((1u << args.flog2) - 1u) << 3 : (nw - 1u) << 2;
    f.lane4 = (lane & 31u) << 2;
    f.c2base = nw * 4;
    f.hmask = args.hmask;
    uint32_t nhit = 0, x = 0;
    HsgpuScanArgs pa;
    pa.cand_cap = args.cand_cap;
    pa.fold_shift = 0;
    SpillState sp;
    sp.region = args.cand + 2ull * (blockIdx.x * (blockDim.x >> 6) + wave) * args.cand_cap;
    sp.written = 0;
    sp.overflow = 0;
    while (tile < tile_end) {
#pragma unroll
        for (int s = 0; s < STAGES; s++) {
            c[(s + STAGES - 1) % STAGES] = issue(tile + (uint64_t)(STAGES - 1) * step);
            const Chunk &cur = c[s];
            uint32_t acc;
            if (MODE == 0) acc = (cur.d.x ^ cur.d.y ^ cur.d.z ^ cur.d.w ^ cur.h.x ^ cur.h.y) == 0x12345678u;
            else if (MODE == 1) acc = filter_chunk<true, false, false, false, false, true, false>(cur, f);
            else if (MODE == 2) acc = filter_chunk<true, false, false, false, true, false, true>(cur, f);
            else acc = pair_filter_chunk(cur, f);
            if (SPILL) {
                spill(pa, sp, (tile << 10) + lane * 16, acc, cur);
            } else {
                const unsigned long long bal = __ballot(acc != 0);
                if (bal) nhit += __popcll(bal), x ^= acc;
            }
            tile += step; /* tiles past the end have empty descriptors: zeros, no hits */
        }
    }
    if (SPILL) nhit = sp.written;
    if (lane == 0 && (nhit | x)) atomicAdd(args.hits, (unsigned long long)nhit);
}

typedef void (*kfn)(ExpArgs);
struct Variant { const char *name; kfn f; };
#define V(S, A, M, E) {"stages=" #S " aux=" #A " mode=" #M " early=" #E, exp_kernel<S, A, M, E>}
#define VS(S, A, M, E) {"stages=" #S " aux=" #A " mode=" #M " early=" #E " WAVE-BLOCKED SPILL", exp_kernel<S, A, M, E, 2, true>}
#define VW(S, A, M, E) {"stages=" #S " aux=" #A " mode=" #M " early=" #E " WAVE-BLOCKED", exp_kernel<S, A, M, E, 2>}
#define VB(S, A, M, E) {"stages=" #S " aux=" #A " mode=" #M " early=" #E " BLOCKED", exp_kernel<S, A, M, E, 1>}

int main(int argc, char **argv) {
    const uint64_t total = 1ull << 30;
    uint8_t *d_corpus;
    uint32_t *d_table;
    unsigned long long *d_hits;
    CHECK(hipMalloc(&d_corpus, total + 64));
    CHECK(hipMalloc(&d_table, 131072));
    CHECK(hipMalloc(&d_hits, 8));
    uint4 *d_cand;
    const uint32_t cand_cap = 8192; /* per wavefront; 16384 wavefronts at most */
    CHECK(hipMalloc(&d_cand, (size_t)16384 * cand_cap * 32));
    {   // random text-like bytes; table with ~0.05% of bits set
        std::vector<uint8_t> h(total);
        uint64_t s = 88172645463325252ull;
        for (uint64_t i = 0; i < total; i += 8) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            uint64_t v = s & 0x1f1f1f1f1f1f1f1full;
            v |= 0x6060606060606060ull;
            memcpy(&h[i], &v, 8);
        }
        CHECK(hipMemcpy(d_corpus, h.data(), total, hipMemcpyHostToDevice));
        std::vector<uint32_t> t(32768, 0);
        for (int i = 0; i < 512; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; t[s % 32768] |= 1u << ((s >> 20) & 31); }
        CHECK(hipMemcpy(d_table, t.data(), 131072, hipMemcpyHostToDevice));
    }
    std::vector<Variant> vs = {
        V(8, 0, 0, false), V(8, 0, 1, false), V(8, 0, 2, false), V(8, 0, 3, false),
        V(8, 2, 0, false), V(8, 2, 1, false), V(8, 2, 3, false),
        V(8, 0, 1, true), V(4, 0, 1, false), V(12, 0, 1, false), V(12, 0, 0, false), V(12, 2, 1, false),
        VB(8, 0, 0, false), VB(8, 0, 1, false), VB(8, 0, 2, false), VB(8, 2, 0, false), VB(8, 2, 1, false), VB(4, 0, 1, false),
        VW(8, 0, 0, false), VW(8, 0, 1, false), VW(8, 0, 2, false), VW(8, 2, 1, false), VW(4, 0, 1, false), VW(8, 0, 3, false), VB(8, 0, 3, false), VS(8, 0, 1, false), VS(8, 0, 2, false), VS(4, 0, 1, false),
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    for (auto &v : vs) CHECK(hipFuncSetAttribute((const void *)v.f, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    struct Geo { int threads, per_cu; size_t lds; } geos[] = {{1024, 1, 131072}, {512, 1, 131072}, {512, 2, 65536 + 8192}, {256, 4, 32768 + 4096}};
    const bool one = argc > 1 && !strcmp(argv[1], "one");
    for (auto &g : geos) {
        if (one && g.threads != 1024) continue;
        printf("---- %d threads x %d per CU, %zu B LDS each\n", g.threads, g.per_cu, g.lds);
        for (auto &v : vs) {
            if (one && strcmp(v.name, "stages=8 aux=0 mode=1 early=false")) continue;
            ExpArgs a;
            a.corpus = d_corpus; a.total = total; a.table = d_table; a.hits = d_hits; a.cand = d_cand; a.cand_cap = cand_cap;
            a.super_shift = g.threads == 1024 ? 14 : g.threads == 512 ? 13 : 12;
            a.flog2 = (strstr(v.name, "mode=3") ? 14 : 15); a.hmask = 0x1fdfdf;
            float best = 1e9, sum = 0;
            for (int it = 0; it < 6; it++) {
                CHECK(hipMemset(d_hits, 0, 8));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.f, dim3(ncu * g.per_cu), dim3(g.threads), g.lds, 0, a);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (it) { sum += ms; if (ms < best) best = ms; }
            }
            unsigned long long hh = 0;
            CHECK(hipMemcpy(&hh, d_hits, 8, hipMemcpyDeviceToHost));
            printf("%-44s avg %.4f ms best %.4f ms  %.0f GB/s  hits %llu\n", v.name, sum / 5, best, total / (sum / 5) / 1e6, hh);
        }
    }
    return 0;
}
