/* scan_inst_pair.hip -- the pair-filter kernels (table.h HSGPU_F_PAIR): two-phase filter, fused fallback,
 * confirm. The class parameters of the shared templates are fixed (4- and 3-byte exact keys, stride 2). */
#include "scan_device.h"

const void *hsgpu_pair_filter_kernel(uint32_t flags, bool fused) {
    const bool b = (flags & HSGPU_F_HAS_B) != 0;
    if (fused)
        return b ? (const void *)hwlm_filter_kernel<true, true, false, false, false, true, false, true, true>
                 : (const void *)hwlm_filter_kernel<true, false, false, false, false, true, false, true, true>;
    /* the two-phase filter does not depend on the classes at all */
    return (const void *)hwlm_filter_kernel<true, false, false, false, false, true, false, false, true>;
}

const void *hsgpu_pair_confirm_kernel(uint32_t flags) {
    return (flags & HSGPU_F_HAS_B) ? (const void *)hwlm_confirm_kernel<true, true, false, true, true>
                                   : (const void *)hwlm_confirm_kernel<true, false, false, true, true>;
}
