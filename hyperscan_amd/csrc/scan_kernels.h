/* scan_kernels.h -- launch interface between runtime.hip and the kernels. */
#ifndef HSGPU_SCAN_KERNELS_H
#define HSGPU_SCAN_KERNELS_H

#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "../../include/hsgpu.h"

#define HSGPU_WG_THREADS 1024
#ifndef HSGPU_CONFIRM_THREADS
#define HSGPU_CONFIRM_THREADS 256 /* tuning builds: 512 (eight wavefronts share one staged key gate: 36 KiB, four workgroups per CU) */
#endif
#ifndef HSGPU_CONFIRM_SPLIT
#define HSGPU_CONFIRM_SPLIT 4 /* confirm wavefronts per candidate region (tuning builds: 2) */
#endif
#define HSGPU_HINT_SHIFT 10 /* one block hint per KiB of corpus */
/* rec_super: 256 supers and a flags word, each on a 64-byte line of its own (thousands of wavefronts poll and add to them:
 * side by side in 2 KiB they were one hot spot in one memory channel) */
#define HSGPU_SUPER(i) ((i) << 3)
#define HSGPU_SUPER_FLAGS HSGPU_SUPER(256)
#define HSGPU_SUPER_WORDS (257 * 8)
#define HSGPU_RUN_MAX 3
#define HSGPU_RUN_STRIDE 4

struct HsgpuScanArgs {
    const uint8_t *corpus;      /* all blocks, concatenated; 16-byte aligned */
    uint64_t total;             /* bytes in corpus */
    const uint64_t *off;        /* nblocks + 1 ascending offsets, off[nblocks] == total */
    uint64_t nblocks;
    uint64_t start;             /* hwlmExec's `start`, applied inside every block */
    const uint8_t *blob;        /* compiled table in HBM */
    /* table header fields the kernels need, copied here by the host so that no
     * kernel starts with a dependent read of the header */
    uint32_t t_flags, t_filter_log2, t_ht_a_log2, t_ht_b_log2, t_hash_mask;
    uint32_t t_off_filter, t_off_c2bits, t_off_ht_a, t_off_ht_b, t_off_c2ref, t_off_lists, t_off_lits;
    hsgpu_match_t *out;         /* match records */
    uint64_t cap;               /* capacity of out */
    unsigned long long *count;  /* total matches (may exceed cap) */
    uint32_t hint_in_filter;    /* 1: the two-phase filter kernel writes the block hints in its prologue */
    uint32_t fold_shift;        /* 16 for HSGPU_F_BFOLD tables (candidate masks: 4-byte-key hits copied to the other half), else 0 */
    const uint32_t *hint;       /* hint[t] = block containing byte t << HSGPU_HINT_SHIFT */
    uint64_t n_hint;
    uint4 *cand;                /* two-phase: 32-byte candidate entries (2 x uint4), one region per filter wavefront */
    uint32_t cand_cap;          /* entries per region */
    uint32_t cand_waves;        /* number of regions = filter grid x 16 */
    uint32_t *cand_counts;      /* [cand_waves] entries written per region, [cand_waves] = overflow flag */
    uint4 *rec_stage;           /* staged match records, one region of rec_cap per producing wavefront */
    uint32_t rec_cap;
    uint32_t rec_regions;       /* number of record regions */
    uint32_t *rec_counts;       /* [rec_regions][2]: records at the front / at the back of each region */
    /* sums of the region fills over "supers" of 2^super_shift consecutive regions (at most 256 of them), added up by
     * the producing wavefronts (one atomic per region); [256] = some region overflowed. With them every sort
     * workgroup finds where its records go from <= 256 sums + fewer than 2^super_shift fills: no scan kernel. */
    unsigned long long *rec_super;
    uint32_t super_shift;
    uint32_t group_regions;     /* consecutive regions that hold the records of one filter workgroup's corpus share */
    /* the folded pipeline: hwlm_confirm_kernel emits every region in delivery order (sorted drains) and record_sort_kernel only
     * gathers them (1; 2 = dense scan: dense batches position by position); 0 = regions in any order, sorted by record_sort_kernel.
     * rec_super[256] = regions that lost records + (regions emitted out of order) << 32 */
    uint32_t fold;
    /* the confirm kernel's partition: every share (= one filter wavefront's candidates) in conf_q parts of whole batches, conf_k
     * consecutive parts per worker wavefront; rec_regions = its workers */
    uint32_t conf_q, conf_k;
    /* dense scans: conf_spread = 1: a record region per part (rec_regions = shares x conf_q), and the conf_k parts of a worker
     * are spread over the corpus row by row (hwlm_confirm_kernel) -- runs of dense input go round all workers */
    uint32_t conf_spread;
    /* ... and a run table per region (round 6, the reference's flood case proper: src/fdr/flood_runtime.h:86-335 replays ONE confirm
     * along a run of one byte value): [rec_regions][HSGPU_RUN_STRIDE], [0].x = the region's runs (<= HSGPU_RUN_MAX), [1 + k] =
     * {index in the region of the records of the run's first lookup, records per lookup, further lookups, what a lookup adds to
     * `end`} -- the further lookups' records exist only here; the region's count (rec_counts) counts them, record_sort_kernel
     * writes them straight to where they go. nullptr: no run tables (every record staged) */
    uint4 *run_tab;
    /* ordinary scans, one part per worker and two per share: the older workgroup of two mirrored dispatch ranks takes
     * 1/2 + conf_skew / 2^16 (scaled by the ranks' distance) of a share's batches (hwlm_confirm_kernel); 0: equal halves */
    uint32_t conf_skew;
    uint32_t conf_cus;          /* the device's CUs: workgroup index / conf_cus = the workgroup's rank among those resident on its CU */
    /* the control block of the PREVIOUS scan on this scratch (the blocks alternate): zeroed by this scan's last
     * kernel, whose workgroups read each other's words of the current block and so cannot zero that one */
    uint32_t *ctl_other;
    uint32_t ctl_other_words;
    unsigned long long *stats;  /* [2] cumulative: candidate entries spilled, overflowed scans */
    uint32_t *overflow_note;    /* host-visible word (mapped pinned memory): set by the fused kernel when it has to redo a scan
                                 * whose candidate regions overflowed; the next scan on this scratch then gives every chunk room */
    unsigned long long *tstamp; /* timing only: [2] min start / max end of the filter kernel (device wall clock) */
    unsigned long long *tstamp_next; /* slot the next scan will use: re-armed by the scan's last kernel */
    /* SOLO scans (round 5: batches up to ~1 MiB in ONE launch): the fused kernel with the placement inside -- the workgroup that
     * finishes last (a ticket in solo_ticket) orders and places every region's records itself, writes the count and leaves the
     * scan's own small control block zeroed. No hint kernel (hint == nullptr: block_of bisects the offsets), no sort kernel. */
    uint32_t solo;
    uint32_t solo_ctl_words;    /* words of the control block that starts at rec_counts (rec_counts | rec_super | ticket) */
    uint32_t *solo_ticket;
    uint32_t srv_inline_at;     /* ... inside the small-batch server: LDS word index of the 16-word answer line a scan with <= HSGPU_SRV_INLINE_RECS records leaves its records and count in (0: none) */
    uint32_t img_keep_words;    /* fused kernel's body inside the small-batch server: the table image is in LDS already but for its first img_keep_words words (0: load all of it) */
    unsigned long long *wg_stamps;   /* tuning (hsgpu_scratch_enable_timing(s, 2)): [filter grid][4] device wall clock per
                                      * workgroup: start, image staged / hints written, wavefront 0's share done, end */
    unsigned long long *conf_stamps; /* the same for the confirm kernel's workers (tuning builds, HSGPU_CONFIRM_STAMPS=1):
                                      * [worker][4]: start, fresh steps | rest steps << 16 | sorted drains << 32, entries, end */
};

/* the small-batch server's mailbox (scan_device.h, hwlm_server_kernel; runtime.hip, server_call): four 64-byte lines.
 *   line 0  host -> device, the WHOLE request: the workgroup's poll is ONE 64-byte read (a dword per lane) that brings the
 *           parameters with the sequence number (the host writes them first: a read that sees the new number sees them)
 *   line 2  device -> host, the WHOLE answer of a request with <= HSGPU_SRV_INLINE_RECS records: records, count, stamps and the
 *           sequence number LAST, written by ONE store instruction (four lanes x 16 bytes = one 64-byte write over the bus, whose
 *           bytes land in address order) -- nothing else has to be ordered in front of it. done_count = ~0: count and records
 *           are in the scratch's mapped area as before (released at system scope in front of this line)
 *   line 3  exited and the debug stamps */
#define HSGPU_SRV_INLINE_RECS 3
struct HsgpuServerCtl {
    uint32_t req_seq, stop;                       /* host -> device */
    uint64_t total, nblocks, start;               /* ... the request (written before req_seq) */
    uint32_t debug, pad0[7];                      /* debug: stamp the request's stages (hsgpu_debug_server_stamping; each stamp is a clock read the wavefront waits for) */
    uint64_t pad1[8];
    uint32_t done_rec[4 * HSGPU_SRV_INLINE_RECS]; /* device -> host: hsgpu_match_t x 3 */
    uint32_t done_count, done_copy_ticks, done_body_ticks, done_seq;
    uint32_t exited, pad3;
    unsigned long long stamps[6];                 /* hsgpu_debug_server_stamps: the body's four stages; [4] first loads out, [5] in front of the first barrier */
    uint32_t pad4[2];
};
static_assert(sizeof(HsgpuServerCtl) == 256 && offsetof(HsgpuServerCtl, done_rec) == 128 && offsetof(HsgpuServerCtl, done_seq) == 188 &&
                  offsetof(HsgpuServerCtl, exited) == 192,
              "the request is line 0, the answer line 2 with its sequence number last");

const void *hsgpu_filter_kernel_for(uint32_t table_flags, bool fused);
/* the resident small-batch server with the fused kernel's body (scan_device.h, hwlm_server_kernel; nullptr: none for this table):
 * launched with (HsgpuScanArgs, HsgpuServerCtl *ctl, HsgpuServerCtl *req, unsigned long long idle_ticks, const uint4 *src_corpus, const uint4 *src_off), ONE
 * workgroup, hsgpu_filter_lds_bytes(fused) + 192 of LDS; src_*: where the host puts a request's batch (mapped memory) -- copied to
 * args.corpus / args.off (device memory) at the head of every request */
const void *hsgpu_server_kernel_for(uint32_t table_flags);
const void *hsgpu_confirm_kernel_for(uint32_t table_flags, bool dense); /* dense: the folded pipeline's kernel for dense scans (fold == 2) */
const void *hsgpu_hint_kernel(void);
const void *hsgpu_record_sort_kernel(void);
size_t hsgpu_filter_lds_bytes(uint32_t table_flags, uint32_t filter_log2, bool fused, uint32_t wg_threads);

#endif
