/*
 * scan_kernels.hip -- the block-mode multi-literal scan kernel for gfx950.
 *
 * Replaces, on the GPU, the reference's FDR / Teddy / Noodle main loops and
 * their confirm step:
 *   FDR_MAIN_LOOP + get_conf_stride_N     src/fdr/fdr.c:157-327,694-723
 *   prep_conf_teddy_m1..m4, CONFIRM_TEDDY src/fdr/teddy.c:893-1064
 *   noodle scan/final                     src/hwlm/noodle_engine.c:113-138
 *   do_confirm_fdr / confWithBit          src/fdr/fdr.c:330-364,
 *                                         src/fdr/fdr_confirm_runtime.h:43-102
 * It is a new design, not a translation: there are no buckets, no shift-or
 * state and no zones. See DESIGN.md "Kernel".
 *
 * Mapping. The corpus is the concatenation of all blocks (CSR offsets). A
 * workgroup of 16 wavefronts owns a 16 KiB super-tile per iteration; each
 * wavefront owns 1 KiB of it, each lane one 16-byte chunk, loaded with one
 * coalesced global_load_dwordx4 two iterations ahead of its use (register
 * double-buffering; nothing in the steady-state loop waits for HBM).
 *
 * Filter (per lookup position, all lanes): hash the 3 bytes ending there with
 * one v_mul_u32_u24, read ONE 32-bit word of the LDS-resident filter, test one
 * bit chosen by the 4th byte (optionally a second bit). With stride 2 only every
 * second byte is a lookup position: the table then also holds every literal
 * keyed one byte early, so a lookup at q catches literals ending at q and q + 1
 * (the kernel is VALU-bound at ~7 instructions per lookup, so this doubles its
 * rate). Two filter layouts:
 *   REPL   small literal sets ("Teddy class"): 32 identical columns, lane l reads
 *          column l & 31 -> every lane of a 32-lane LDS group hits its own bank,
 *          conflict-free by construction;
 *   hashed large sets ("FDR class"): one 2^k-word table, up to 128 KiB.
 * Candidates are collected as one 16-bit mask per lane and class.
 *
 * Confirm. Candidates are rare (a fraction of a percent of positions) but each
 * one needs a chain of dependent HBM/L2 reads (window -> hash bucket -> literal
 * -> block offsets). Inside the streaming kernel every link of that chain queues
 * behind the wavefront's own prefetches, so the default pipeline is two-phase:
 *   hwlm_filter_kernel  streams the corpus, compacts {chunk, masks} candidate
 *                       entries through a per-wavefront LDS queue and writes
 *                       them to HBM 64 at a time (one reservation per 512 B);
 *   hwlm_confirm_kernel one lane per candidate entry, all of them in flight at
 *                       once: exact hash-table bucket (32 B), (window & msk) ==
 *                       v per listed literal, block lookup through a per-KiB
 *                       hint table, bound checks, records staged per wavefront
 *                       in LDS and stored with one reservation per flush.
 * A fused variant (confirm inside the streaming kernel) is kept as the
 * always-correct fallback for inputs so dense that the candidate buffer
 * overflows (the role of the reference's flood path, flood_runtime.h:86-335);
 * it is launched after the other two and returns at once unless they overflowed.
 *
 * Block boundaries are invisible to the filter; the confirm step resolves the
 * block of a candidate and rejects matches that would start before their block
 * (or before `start` within it).
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

const void *hsgpu_filter_kernel_for(uint32_t flags, bool fused) {
    typedef const void *(*pick_t)(uint32_t);
    static const pick_t tab[2][2][2] = {
        {{hsgpu_filter_kernels_r0f0k0, hsgpu_filter_kernels_r0f0k1}, {hsgpu_filter_kernels_r0f1k0, hsgpu_filter_kernels_r0f1k1}},
        {{hsgpu_filter_kernels_r1f0k0, hsgpu_filter_kernels_r1f0k1}, {hsgpu_filter_kernels_r1f1k0, hsgpu_filter_kernels_r1f1k1}}};
    return tab[(flags & HSGPU_F_REPL) ? 1 : 0][fused ? 1 : 0][(flags & HSGPU_F_K2) ? 1 : 0](flags);
}

template <bool S2> static const void *pick_confirm(uint32_t flags) {
    if (flags & HSGPU_F_HAS_C) return (const void *)hwlm_confirm_kernel<true, true, true, S2>;
    if (flags & HSGPU_F_HAS_B) return (const void *)hwlm_confirm_kernel<true, true, false, S2>;
    return (const void *)hwlm_confirm_kernel<true, false, false, S2>;
}
const void *hsgpu_confirm_kernel_for(uint32_t flags) {
    return (flags & HSGPU_F_STRIDE2) ? pick_confirm<true>(flags) : pick_confirm<false>(flags);
}

const void *hsgpu_hint_kernel(void) { return (const void *)block_hint_kernel; }
const void *hsgpu_record_scan_kernel(void) { return (const void *)record_scan_kernel; }
const void *hsgpu_record_pack_kernel(void) { return (const void *)record_pack_kernel; }
const void *hsgpu_control_reset_kernel(void) { return (const void *)control_reset_kernel; }

size_t hsgpu_filter_lds_bytes(uint32_t flags, uint32_t filter_log2, bool fused, uint32_t wg_threads) {
    size_t words = (size_t)hsgpu_filter_words(flags, filter_log2) + ((flags & HSGPU_F_HAS_C) ? 2048 : 0);
    return words * 4 + (fused ? (size_t)(wg_threads / 64) * sizeof(WaveLds) : 0);
}
