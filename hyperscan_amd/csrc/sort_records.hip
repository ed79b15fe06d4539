/*
 * sort_records.hip -- match records into delivery order on the device.
 *
 * The reference delivers callbacks in non-decreasing `end` within a scan
 * (src/hwlm/hwlm.h:101-118); a batch delivers block by block. The scan kernels emit records
 * grouped by the wavefront that found them, so the host-facing entry points sort them by
 * (block, end, literal index) -- the key hsgpu_hwlm_replay walks. Doing that with std::sort
 * after the copy-out cost more than the scan for large result sets (60 ms for 0.8 M
 * records); here it is two stable LSD radix sorts (rocPRIM through hipCUB: a library sort,
 * nothing on the scan path): by literal index, then by (block, end).
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <stdint.h>

#include "internal.h"

namespace {

__global__ void sr_lit_keys(const uint4 *rec, uint32_t n, uint32_t *key, uint32_t *idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = rec[i].w; /* literal index */
    idx[i] = i;
}
__global__ void sr_pos_keys(const uint4 *rec, const uint32_t *idx, uint32_t n, uint64_t *key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = rec[idx[i]];
    key[i] = (uint64_t)r.x << 32 | r.y; /* block, end */
}
__global__ void sr_gather(const uint4 *rec, const uint32_t *idx, uint32_t n, uint4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = rec[idx[i]];
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
    size_t key32_a, key32_b, idx_a, idx_b, idx_c, key64_a, key64_b, out, temp, total;
};

bool plan(uint32_t n, Layout &L) {
    size_t t32 = 0, t64 = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, t32, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n) != hipSuccess)
        return false;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, t64, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n) != hipSuccess)
        return false;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes);
        return at;
    };
    L.key32_a = take((size_t)n * 4);
    L.key32_b = take((size_t)n * 4);
    L.idx_a = take((size_t)n * 4);
    L.idx_b = take((size_t)n * 4);
    L.idx_c = take((size_t)n * 4);
    L.key64_a = take((size_t)n * 8);
    L.key64_b = take((size_t)n * 8);
    L.out = take((size_t)n * 16);
    L.temp = take(std::max(t32, t64));
    L.total = o;
    return true;
}

} // namespace

/* bytes of device workspace hsgpu_sort_records needs for n records (0: n not sortable here) */
size_t hsgpu_sort_workspace_bytes(uint64_t n) {
    Layout L;
    if (n == 0 || n >= (1ull << 31) || !plan((uint32_t)n, L)) return 0;
    return L.total;
}

/* sorts d_rec[0 .. n) in place by (block, end, lit); asynchronous on st */
int hsgpu_sort_records(void *d_rec, uint64_t n, void *d_ws, size_t ws_bytes, hipStream_t st) {
    if (n < 2) return HSGPU_SUCCESS;
    Layout L;
    if (n >= (1ull << 31) || !plan((uint32_t)n, L) || ws_bytes < L.total) {
        hsgpu_set_error("record sort: %llu records do not fit the workspace", (unsigned long long)n);
        return HSGPU_INVALID;
    }
    uint8_t *ws = (uint8_t *)d_ws;
    const uint4 *rec = (const uint4 *)d_rec;
    const uint32_t cnt = (uint32_t)n;
    const dim3 grid((cnt + 255) / 256), wg(256);
    uint32_t *k32a = (uint32_t *)(ws + L.key32_a), *k32b = (uint32_t *)(ws + L.key32_b);
    uint32_t *ia = (uint32_t *)(ws + L.idx_a), *ib = (uint32_t *)(ws + L.idx_b), *ic = (uint32_t *)(ws + L.idx_c);
    uint64_t *k64a = (uint64_t *)(ws + L.key64_a), *k64b = (uint64_t *)(ws + L.key64_b);
    size_t temp_bytes = ws_bytes - L.temp;
    hipLaunchKernelGGL(sr_lit_keys, grid, wg, 0, st, rec, cnt, k32a, ia);
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.temp, temp_bytes, (const uint32_t *)k32a, k32b, (const uint32_t *)ia, ib,
                                           (int)cnt, 0, 32, st) != hipSuccess)
        return HSGPU_UNKNOWN_ERROR;
    hipLaunchKernelGGL(sr_pos_keys, grid, wg, 0, st, rec, (const uint32_t *)ib, cnt, k64a);
    temp_bytes = ws_bytes - L.temp;
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.temp, temp_bytes, (const uint64_t *)k64a, k64b, (const uint32_t *)ib, ic,
                                           (int)cnt, 0, 64, st) != hipSuccess)
        return HSGPU_UNKNOWN_ERROR;
    uint4 *out = (uint4 *)(ws + L.out);
    hipLaunchKernelGGL(sr_gather, grid, wg, 0, st, rec, (const uint32_t *)ic, cnt, out);
    if (hipMemcpyAsync(d_rec, out, (size_t)cnt * 16, hipMemcpyDeviceToDevice, st) != hipSuccess) return HSGPU_UNKNOWN_ERROR;
    if (hipGetLastError() != hipSuccess) return HSGPU_UNKNOWN_ERROR;
    return HSGPU_SUCCESS;
}
