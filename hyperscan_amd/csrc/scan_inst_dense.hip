/* scan_inst_dense.hip -- hwlm_confirm_kernel for dense scans of the folded pipeline (HsgpuScanArgs::fold == 2: every chunk has
 * a candidate entry, dense batches are confirmed position by position: scan_device.h). Instantiations of their own, so that the
 * ordinary confirm kernels keep their registers. */
#include "scan_device.h"

template <bool S2> static const void *pick_dense(uint32_t flags) {
    if (flags & HSGPU_F_HAS_C) return (const void *)hwlm_confirm_kernel<true, true, true, S2, false, true>;
    if (flags & HSGPU_F_HAS_B) return (const void *)hwlm_confirm_kernel<true, true, false, S2, false, true>;
    return (const void *)hwlm_confirm_kernel<true, false, false, S2, false, true>;
}
const void *hsgpu_dense_confirm_kernel(uint32_t flags) {
    if (flags & HSGPU_F_PAIR)
        return (flags & HSGPU_F_HAS_B) ? (const void *)hwlm_confirm_kernel<true, true, false, true, true, true>
                                       : (const void *)hwlm_confirm_kernel<true, false, false, true, true, true>;
    return (flags & HSGPU_F_STRIDE2) ? pick_dense<true>(flags) : pick_dense<false>(flags);
}
