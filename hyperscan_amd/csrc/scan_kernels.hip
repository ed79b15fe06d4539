/*
 * scan_kernels.hip -- kernel selection for the literal scan: maps a table's flags to the
 * instantiation of hwlm_filter_kernel / hwlm_confirm_kernel that serves it (the kernels
 * themselves, and the design notes, are in scan_device.h; the filter variants are compiled
 * in parallel translation units, scan_inst_r?f?k?.hip).
 */
#include "scan_device.h"

/* filter-kernel variants live in scan_inst_r?f?k?.hip (compiled in parallel) */
const void *hsgpu_filter_kernels_r0f0k0(uint32_t flags);
const void *hsgpu_filter_kernels_r0f0k1(uint32_t flags);
const void *hsgpu_filter_kernels_r0f1k0(uint32_t flags);
const void *hsgpu_filter_kernels_r0f1k1(uint32_t flags);
const void *hsgpu_filter_kernels_r1f0k0(uint32_t flags);
const void *hsgpu_filter_kernels_r1f0k1(uint32_t flags);
const void *hsgpu_filter_kernels_r1f1k0(uint32_t flags);
const void *hsgpu_filter_kernels_r1f1k1(uint32_t flags);

const void *hsgpu_server_kernels_r0f1k0(uint32_t flags); /* the small-batch server: the fused units */
const void *hsgpu_server_kernels_r0f1k1(uint32_t flags);
const void *hsgpu_server_kernels_r1f1k0(uint32_t flags);
const void *hsgpu_server_kernels_r1f1k1(uint32_t flags);

const void *hsgpu_pair_filter_kernel(uint32_t flags, bool fused); /* scan_inst_pair.hip */
const void *hsgpu_pair_confirm_kernel(uint32_t flags);

const void *hsgpu_filter_kernel_for(uint32_t flags, bool fused) {
    if (flags & HSGPU_F_PAIR) return hsgpu_pair_filter_kernel(flags, fused);
    typedef const void *(*pick_t)(uint32_t);
    static const pick_t tab[2][2][2] = {
        {{hsgpu_filter_kernels_r0f0k0, hsgpu_filter_kernels_r0f0k1}, {hsgpu_filter_kernels_r0f1k0, hsgpu_filter_kernels_r0f1k1}},
        {{hsgpu_filter_kernels_r1f0k0, hsgpu_filter_kernels_r1f0k1}, {hsgpu_filter_kernels_r1f1k0, hsgpu_filter_kernels_r1f1k1}}};
    return tab[(flags & HSGPU_F_REPL) ? 1 : 0][fused ? 1 : 0][(flags & HSGPU_F_K2) ? 1 : 0](flags);
}

const void *hsgpu_server_kernel_for(uint32_t flags) {
    if (flags & HSGPU_F_PAIR) return nullptr; /* (pair tables take the launch path) */
    typedef const void *(*pick_t)(uint32_t);
    static const pick_t tab[2][2] = {{hsgpu_server_kernels_r0f1k0, hsgpu_server_kernels_r0f1k1}, {hsgpu_server_kernels_r1f1k0, hsgpu_server_kernels_r1f1k1}};
    return tab[(flags & HSGPU_F_REPL) ? 1 : 0][(flags & HSGPU_F_K2) ? 1 : 0](flags);
}

template <bool S2> static const void *pick_confirm(uint32_t flags) {
    if (flags & HSGPU_F_HAS_C) return (const void *)hwlm_confirm_kernel<true, true, true, S2>;
    if (flags & HSGPU_F_HAS_B) return (const void *)hwlm_confirm_kernel<true, true, false, S2>;
    return (const void *)hwlm_confirm_kernel<true, false, false, S2>;
}
const void *hsgpu_dense_confirm_kernel(uint32_t flags); /* scan_inst_dense.hip */
const void *hsgpu_confirm_kernel_for(uint32_t flags, bool dense) {
    if (dense) return hsgpu_dense_confirm_kernel(flags);
    if (flags & HSGPU_F_PAIR) return hsgpu_pair_confirm_kernel(flags);
    return (flags & HSGPU_F_STRIDE2) ? pick_confirm<true>(flags) : pick_confirm<false>(flags);
}

const void *hsgpu_hint_kernel(void) { return (const void *)block_hint_kernel; }
const void *hsgpu_record_sort_kernel(void) { return (const void *)record_sort_kernel; }

size_t hsgpu_filter_lds_bytes(uint32_t flags, uint32_t filter_log2, bool fused, uint32_t wg_threads) {
    size_t words = (size_t)hsgpu_filter_words(flags, filter_log2) + ((flags & HSGPU_F_HAS_C) ? 2048 : 0);
    return words * 4 + (fused ? (size_t)(wg_threads / 64) * sizeof(WaveLds) : 64 /* the workgroup's progress sum */);
}
