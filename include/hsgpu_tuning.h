/* hsgpu_tuning.h -- engine forcing for tests and tuning runs. Not part of the drop-in boundary: a caller of
 * include/hsgpu.h passes flags = 0 and never touches a scratch's geometry. */
#ifndef HSGPU_TUNING_H
#define HSGPU_TUNING_H

#include "hsgpu.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden and an export list (csrc/Makefile: libhsgpu.map, made from these headers):
 * what is declared between here and the pop is the whole exported surface, as hs.def / hs_runtime.def are the reference's */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* hsgpu_hwlm_build flags: engine forcing for tests, like the reference's
 * fdrBuildProtoHinted hook (src/fdr/fdr_compile.cpp:900-911). 0 = automatic. */
#define HSGPU_BUILD_FORCE_REPL 1u   /* bank-replicated ("Teddy class") filter */
#define HSGPU_BUILD_FORCE_HASHED 2u /* hashed ("FDR class") filter */
#define HSGPU_BUILD_FORCE_K2 4u     /* two filter bits per key */
#define HSGPU_BUILD_FORCE_K1 8u     /* one filter bit per key */
#define HSGPU_BUILD_FORCE_STRIDE1 16u /* look up every byte position (no stride-2 keys) */
#define HSGPU_BUILD_FORCE_STRIDE2 64u /* stride 2 even with 2- and 3-byte literals */
#define HSGPU_BUILD_FORCE_SMALL 128u /* 32 KiB hashed filter, run as three 8-wavefront workgroups per CU */
#define HSGPU_BUILD_FORCE_MEDIUM 256u /* 64 KiB hashed filter: two 16-wavefront workgroups per CU where registers allow */
#define HSGPU_BUILD_FORCE_BLIND 32u /* case-blind hash keys even without caseless literals */
#define HSGPU_BUILD_NO_FOLD 512u    /* keep the separate 3-byte-key filter test even when only few such keys exist */
#define HSGPU_BUILD_FORCE_PAIR 1024u /* the stride-2 pair filter (opt-in; an error for sets it cannot hold) */
#define HSGPU_BUILD_NO_WIDE 2048u    /* 32-bit filter words even where 64-bit entries {lo, hi} would be chosen (HSGPU_F_WIDE) */
#define HSGPU_BUILD_FORCE_BLOOM 8192u /* the full-window Bloom gate (HSGPU_F_BLOOM, csrc/table.h) in place of the 4-byte key gate on stride-1 sets: opt-in --
                                      * it stops 70 % of the exact-table probes and the confirm stage is 5 % SLOWER with it (profiles/r05_bloom_gate_ab.txt) */
#define HSGPU_BUILD_NO_GATE 4096u    /* no key gate in front of the exact hash tables (the confirm kernel then probes a table for every candidate) */


/* Launch geometry / pipeline of the scans on one scratch, for tests that must cover every pipeline and for tuning
 * runs (the library reads no environment variables): fused_only == 1 runs the always-correct fused kernel alone
 * (normally the overflow fallback), fused_only == 2 the two-phase pipeline with record_sort_kernel as a kernel of its own
 * sorting behind the confirm kernel (regions in any order, sorted there; by default the confirm workers emit in delivery order and the
 * kernel behind them only gathers); fused_only == 3 / 4 switch the one-launch path of small batches off / force it at any size; fused_only == 5 gives the confirm kernel's workers equal halves of the candidate shares
 * instead of halves in proportion to their workgroup's dispatch rank (csrc/runtime.hip, conf_skew: the A/B of that choice), and makes
 * dense scans stage every record of a run of one byte value instead of a descriptor per run (run_tab: the A/B of that);
 * wg_threads in {0, 256, 512, 1024} and wg_per_cu in {0, 1..4} override the workgroup size / workgroups per CU the runtime would
 * choose (0 = its choice). */
int hsgpu_scratch_set_tuning(hsgpu_scratch_t *s, int fused_only, unsigned wg_threads, unsigned wg_per_cu);

/* The small-batch server's last request on this scratch, stage by stage on the device (microseconds): us[0] request taken ->
 * table image and first tiles in place, us[1] -> wavefront 0 has filtered and confirmed its share, us[2] -> records placed and count
 * written. (hsgpu_scratch_server_last_us, include/hsgpu.h: the request as a whole.) */
int hsgpu_debug_server_stamps(hsgpu_scratch_t *s, float *us /* [3] */);
/* The stamps (and hsgpu_scratch_server_last_us's times) are taken only for requests made while stamping is on (off by default:
 * every stamp is a clock read the wavefront waits for, ~0.1 us each of a ~5 us request). */
int hsgpu_debug_server_stamping(hsgpu_scratch_t *s, int on);
/* ... and inside the first stage, from the body's start: us[0] -> the first tiles' (and the partial last tile's) loads are issued,
 * us[1] -> wavefront 0 stands in front of the body's first barrier (the rest of the stage is the wait for the loads). */
int hsgpu_debug_server_head_stamps(hsgpu_scratch_t *s, float *us /* [2] */);
/* hsgpu_hwlm_exec `calls` times in a row from native code, as hsbench walks its blocks (tools/hsbench/engine_hyperscan.cpp:132-145);
 * *us_per_call = the mean: what a C caller pays per call (a Python ctypes call costs ~1 us by itself). Stops at the first call
 * that does not return HSGPU_HWLM_SUCCESS and returns its code. */
int hsgpu_debug_exec_repeat(const hsgpu_hwlm_t *t, const uint8_t *buf, size_t len, size_t start, hsgpu_hwlm_cb cb, hsgpu_scratch_t *s,
                            uint64_t groups, unsigned calls, double *us_per_call);

/* hsgpu_scratch_enable_timing(s, 2) also stamps every workgroup of the filter kernel (device wall clock); this returns the last
 * scan's stamps in milliseconds from the earliest start: out[4 w + {0 start, 1 image staged and hints written, 2 wavefront 0's
 * share streamed, 3 end}], w < min(*n_wgs, max_wgs). Synchronises the device. */
int hsgpu_scratch_get_wg_stamps(hsgpu_scratch_t *s, float *out, unsigned max_wgs, unsigned *n_wgs);

/* The same for the confirm kernel's worker wavefronts, from a tuning build of the library (-DHSGPU_CONFIRM_STAMPS=1; the product
 * build writes no stamps and this returns zeros): out[6 w + {0 start, 1 end (ms from the earliest start), 2 fresh steps, 3 steps
 * on the rest queue, 4 sorted drains, 5 candidate entries}]. Synchronises the device. */
int hsgpu_scratch_get_conf_stamps(hsgpu_scratch_t *s, float *out, unsigned max_workers, unsigned *n_workers);

/* The confirm kernel's partition (csrc/runtime.hip): n_shares candidate regions (one per filter wavefront), each cut into *q parts
 * of whole batches, *k consecutive parts per worker wavefront, for a device that holds max_workers of them at once: the pair that
 * keeps most worker slots busy with every worker getting the same number of parts. Returns the workers used. Host arithmetic only. */
unsigned hsgpu_confirm_partition(unsigned n_shares, unsigned max_workers, unsigned *q, unsigned *k);

/* ---- guard pages: the out-of-bounds tests (csrc/devmem.hip, tests/test_gpu_guard_pages.py) ----------------------
 * The reference never touches a byte outside [buf, buf + len) (zones, src/fdr/fdr.c:392-690; vectoredLoad*,
 * src/fdr/teddy_runtime_common.h:126-391; unit/internal/fdr.cpp:496-561 scans at every alignment). hipMalloc's 2 MiB
 * granules would hide an over-read here, so the tests place buffers against UNMAPPED pages instead:
 *   hsgpu_debug_guard_malloc  device memory of exactly `bytes` inside a reserved address range whose neighbouring granules are
 *                             not mapped; back == 0: the buffer starts at the first mapped byte; back != 0: it ends at the last
 *                             one (moved down to a multiple of `align`, a power of two: with bytes % align == 0 an access one
 *                             byte past the end faults)
 *   hsgpu_debug_guard_mode    1 / 2: every device buffer the LIBRARY allocates from now on (scratch buffers, table images,
 *                             exchange slots) is placed the same way, front (1) or back (2), sized exactly, without the
 *                             growth slack; 0: hipMalloc again
 *   hsgpu_debug_guard_probe   one byte read (or written) by a kernel at p + byte_offset, then a synchronisation: the test's proof
 *                             that the mechanism faults on this box (run in a child process: the fault kills it)
 * None of this is on any scan's path; with mode 0 (the default) an allocation is one hipMalloc as before. */
int hsgpu_debug_guard_mode(int mode);
int hsgpu_debug_guard_malloc(void **p, size_t bytes, size_t align, int back);
void hsgpu_debug_guard_free(void *p);
int hsgpu_debug_guard_probe(void *p, long long byte_offset, int write);
/* synchronous hipMemcpy / hipMemset through the library's own HIP runtime (the tests fill and read guard buffers without torch) */
int hsgpu_debug_guard_copy(void *dst, const void *src, size_t bytes, int to_device);
int hsgpu_debug_guard_fill(void *dst, int value, size_t bytes);

/* The host confirm's last large batch in this process (csrc/hs_facade.cpp, tools/confirm_prof.py), seconds: out[0] setup, [1] the
 * parallel part's wall time, [2] / [3] the slowest / fastest worker's own time, [4] the delivery loop (callbacks). */
void hsgpu_debug_confirm_timing(double out[5]);
/* the host confirm's thread count for large batches, for sweeps (0 = the default: min(64, hardware threads)) */
void hsgpu_debug_confirm_threads(unsigned n);

/* The first 32 hex digits of the sha256 over the sources this library was built from (csrc/Makefile, STAMPED): the built
 * library is not in the repository, and a test compares this with the tree it runs in. */
const char *hsgpu_source_hash(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
