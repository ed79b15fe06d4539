/* scan_kernels.h -- launch interface between runtime.hip and the kernels. */
#ifndef HSGPU_SCAN_KERNELS_H
#define HSGPU_SCAN_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/hsgpu.h"

#define HSGPU_WG_THREADS 1024

struct HsgpuScanArgs {
    const uint8_t *corpus;      /* all blocks, concatenated; 16-byte aligned */
    uint64_t total;             /* bytes in corpus */
    const uint64_t *off;        /* nblocks + 1 ascending offsets, off[nblocks] == total */
    uint64_t nblocks;
    uint64_t start;             /* hwlmExec's `start`, applied inside every block */
    const uint8_t *blob;        /* compiled table in HBM */
    hsgpu_match_t *out;         /* match records */
    uint64_t cap;               /* capacity of out */
    unsigned long long *count;  /* total matches (may exceed cap) */
};

const void *hsgpu_scan_kernel_for(uint32_t table_flags);
size_t hsgpu_scan_lds_bytes(uint32_t table_flags, uint32_t filter_log2_words);
uint32_t hsgpu_scan_super_tile(void);

#endif
