/*
 * hsgpu.h -- C ABI of the MI355X-native block-mode literal scan engine.
 *
 * This is the drop-in boundary for Hyperscan's literal-matcher layer ("HWLM"):
 * every entry point names the reference interface it replaces.  Plain pointers
 * and sizes only; no C++ or torch types cross this boundary.
 *
 *   reference interface                          replaced by
 *   -------------------------------------------  ---------------------------------
 *   struct hwlmLiteral   src/hwlm/hwlm_literal.h:51-129   hsgpu_lit_t
 *   hwlmBuildProto+hwlmBuild src/hwlm/hwlm_build.h:112-120 hsgpu_hwlm_build
 *   hwlmSize             src/hwlm/hwlm_build.h:127         hsgpu_hwlm_size
 *   HWLMCallback         src/hwlm/hwlm.h:98-99             hsgpu_hwlm_cb
 *   hwlmExec             src/hwlm/hwlm.h:116-118           hsgpu_hwlm_exec
 *   (no equivalent: one hs_scan per block,                 hsgpu_hwlm_exec_batch,
 *    tools/hsbench/main.cpp:502-528)                       hsgpu_hwlm_scan_dev
 *   struct hs_scratch    src/scratch.h (per-thread state)  hsgpu_scratch_t
 *   hs_serialize_database src/hs_common.h:108-169          hsgpu_hwlm_serialize/_deserialize
 *   shuftiExec/truffleExec/vermicelliExec                  hsgpu_class_* (hsgpu_class.h section below)
 *     src/nfa/shufti.h:46, truffle.h:45, vermicelli.h:42
 *
 * Error convention: the hs_error_t values of src/hs_common.h:478-588.
 * hsgpu_hwlm_exec additionally returns the hwlm_error_t values of
 * src/hwlm/hwlm.h:62-72 (0 success, 1 terminated by callback, 2 error).
 */
#ifndef HSGPU_H
#define HSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden and an export list (csrc/Makefile: libhsgpu.map, made from these headers):
 * what is declared between here and the pop is the whole exported surface, as hs.def / hs_runtime.def are the reference's */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* hs_error_t mirror (src/hs_common.h:478-588) */
#define HSGPU_SUCCESS 0
#define HSGPU_INVALID (-1)
#define HSGPU_NOMEM (-2)
#define HSGPU_SCAN_TERMINATED (-3)
#define HSGPU_COMPILER_ERROR (-4)
#define HSGPU_DB_VERSION_ERROR (-5)
#define HSGPU_SCRATCH_IN_USE (-10)
#define HSGPU_INSUFFICIENT_SPACE (-12)
#define HSGPU_UNKNOWN_ERROR (-13)

/* hwlm_error_t mirror (src/hwlm/hwlm.h:62-72) */
#define HSGPU_HWLM_SUCCESS 0
#define HSGPU_HWLM_TERMINATED 1
#define HSGPU_HWLM_ERROR_UNKNOWN 2

#define HSGPU_ALL_GROUPS 0xffffffffffffffffULL /* HWLM_ALL_GROUPS */
#define HSGPU_LITERAL_MAX_LEN 8                /* HWLM_LITERAL_MAX_LEN, hwlm.h:75 */
#define HSGPU_MASKLEN 8                        /* HWLM_MASKLEN, hwlm_literal.h:46 */

/* One literal, field for field the reference's hwlmLiteral:
 * s/len (<= 8 bytes), id passed to the callback, nocase, noruns, groups and
 * the supplementary msk/cmp pair ((v & msk) == cmp over the final msk_len <= 8
 * bytes, memory order, last byte of msk aligned with last byte of s). */
typedef struct hsgpu_lit {
    const uint8_t *s;
    uint32_t len;
    uint32_t id;
    uint8_t nocase;
    uint8_t noruns;
    uint8_t pad[2];
    uint32_t msk_len;
    uint64_t groups;
    const uint8_t *msk;
    const uint8_t *cmp;
} hsgpu_lit_t;

/* One match record as written by the GPU: `end` is the offset of the literal's
 * last byte inside block `block` (hwlm.h:101-105); `lit` is the index of the
 * literal in the array given to hsgpu_hwlm_build. 16 bytes, 16-byte aligned. */
typedef struct hsgpu_match {
    uint32_t block;
    uint32_t end;
    uint32_t id;
    uint32_t lit;
} hsgpu_match_t;

typedef struct hsgpu_hwlm hsgpu_hwlm_t;       /* compiled literal table (immutable, shareable) */
typedef struct hsgpu_scratch hsgpu_scratch_t; /* per-caller device state: stream + buffers */

/* Callback: as HWLMCallback. Return value = the live group mask; 0 terminates. */
typedef uint64_t (*hsgpu_hwlm_cb)(size_t end, uint32_t id, void *ctx);

typedef struct hsgpu_hwlm_info {
    uint32_t n_lits;
    uint32_t n_class_a, n_class_b, n_class_c; /* literals keyed on 4-, 3-, <=2-byte suffix */
    uint32_t filter_words;                    /* LDS filter size in 32-bit words */
    uint32_t filter_entries;                  /* enumerated key variants inserted */
    uint32_t ht_a_slots, ht_b_slots;
    uint32_t max_size;                        /* longest literal/mask */
    uint32_t blob_bytes;
    uint32_t flags;                           /* HSGPU_F_* of csrc/table.h (classes, filter layout) */
} hsgpu_hwlm_info_t;

/* hsgpu_hwlm_build's `flags`: 0 = the compiler chooses the layout. The engine-forcing values that tests and tuning
 * runs pass (the role of the reference's fdrBuildProtoHinted hook, src/fdr/fdr_compile.cpp:900-911) live in
 * hsgpu_tuning.h, with the scratch's launch-geometry override. */

/* ---- build side ---------------------------------------------------------- */

/* Compile n literals. Fails with HSGPU_COMPILER_ERROR on an invalid literal
 * (len > 8, msk_len > 8, empty literal, inconsistent msk/cmp, id 0xffffffff). */
int hsgpu_hwlm_build(const hsgpu_lit_t *lits, size_t n, unsigned flags, hsgpu_hwlm_t **out);
void hsgpu_hwlm_free(hsgpu_hwlm_t *t);
size_t hsgpu_hwlm_size(const hsgpu_hwlm_t *t);
int hsgpu_hwlm_get_info(const hsgpu_hwlm_t *t, hsgpu_hwlm_info_t *info);
/* Position-independent blob; two-call pattern: buf == NULL returns the size. */
int hsgpu_hwlm_serialize(const hsgpu_hwlm_t *t, void *buf, size_t cap, size_t *len);
int hsgpu_hwlm_deserialize(const void *buf, size_t len, hsgpu_hwlm_t **out);

/* ---- run side ------------------------------------------------------------ */

/* device < 0: the calling thread's current HIP device. */
int hsgpu_scratch_alloc(hsgpu_scratch_t **out, int device);
void hsgpu_scratch_free(hsgpu_scratch_t *s);

/* The small-batch server. hsbench block mode is ONE hs_scan per block (tools/hsbench/engine_hyperscan.cpp:132-145) and the
 * reference serves a packet in about a microsecond from dedicated small matchers (src/rose/block.c:382-391,
 * src/runtime.c:401-413); a kernel launch per call costs this engine ~25 us. With the server enabled, hsgpu_hwlm_exec /
 * hsgpu_hwlm_exec_batch (and hs_scan through them) hand batches of up to 16 KiB to ONE resident workgroup that polls a request
 * word in mapped page-locked memory: no launch, no copy command, no sleeping synchronisation. The workgroup ends by itself after
 * idle_us (default 300) without a request -- it holds a compute unit no longer than that after a burst of calls, and a device
 * synchronisation never waits longer for it -- and is started again by the next small call; any other scan on the scratch,
 * hsgpu_scratch_free and disabling end it at once. Off by default. On a device whose memory the host can write (a large PCIe
 * BAR: hipDeviceProp_t::isLargeBar) the request -- sequence number, parameters, offsets, bytes -- is written straight into device
 * memory, so that the workgroup polls and reads locally; records, count and the done word come back through page-locked host
 * memory either way. enable == 2 keeps the request in mapped host memory on every device (what a device without a large BAR
 * does; the A/B). hsgpu_scratch_server_stats: requests served, server launches, whether one is resident right now (any pointer
 * may be NULL). */
int hsgpu_scratch_enable_server(hsgpu_scratch_t *s, int enable, unsigned idle_us /* 0: keep */);
int hsgpu_scratch_server_stats(hsgpu_scratch_t *s, uint64_t *calls, uint64_t *launches, int *live);
/* the last request's device-side times: the batch's copy over the bus, the scan itself (device wall clock, microseconds); taken
 * for requests made while hsgpu_debug_server_stamping (include/hsgpu_tuning.h) is on, zero otherwise */
int hsgpu_scratch_server_last_us(hsgpu_scratch_t *s, float *copy_us, float *scan_us);

/* hwlmExec (src/hwlm/hwlm.h:116-118, src/hwlm/hwlm.c:172-199), argument for argument: scan one host
 * block, deliver callbacks in non-decreasing `end` on the calling thread, honouring the group
 * mask returned by the callback, noruns and termination. Returns HSGPU_HWLM_*. The callback's
 * third argument is the scratch itself, as in the reference (HWLMCallback, src/hwlm/hwlm.h:80-99),
 * unless the caller hung a pointer of its own on the scratch with hsgpu_scratch_set_context. */
int hsgpu_hwlm_exec(const hsgpu_hwlm_t *t, const uint8_t *buf, size_t len, size_t start,
                    hsgpu_hwlm_cb cb, hsgpu_scratch_t *scratch, uint64_t groups);
void hsgpu_scratch_set_context(hsgpu_scratch_t *s, void *ctx);
void *hsgpu_scratch_get_context(const hsgpu_scratch_t *s);

/* Batched host form: nblocks independent blocks, block i = base[off[i], off[i+1]).
 * Writes up to cap records sorted by (block, end, lit) and the total in *nout
 * (HSGPU_INSUFFICIENT_SPACE if *nout > cap; the first cap are valid). */
int hsgpu_hwlm_exec_batch(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const uint8_t *base,
                          const uint64_t *off, size_t nblocks, size_t start,
                          hsgpu_match_t *out, size_t cap, size_t *nout);

/* The same for a batch that is already on the device (d_corpus, d_off as hsgpu_hwlm_scan_dev takes them): the scan, then its
 * records -- in delivery order -- into page-locked host memory OWNED BY THE SCRATCH: *recs points at them until the next call on
 * this scratch, *nout is their number. Repeats the scan itself when its record buffer was too small. */
int hsgpu_hwlm_exec_resident(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const void *d_corpus, uint64_t total_bytes,
                             const void *d_off, uint64_t nblocks, uint64_t start, const hsgpu_match_t **recs, size_t *nout);

/* Device-resident form (the hot path): corpus, offsets, records and counter all
 * live in HBM; asynchronous on `stream` (a hipStream_t passed as void*, NULL =
 * the scratch's own stream); no host synchronisation. d_off holds nblocks+1
 * ascending uint64 offsets with d_off[0] == 0 and d_off[nblocks] == total_bytes. The records arrive in delivery
 * order, sorted by (block, end, lit) -- hwlmExec's non-decreasing `end` (src/hwlm/hwlm.h:101-118),
 * block by block -- and *d_count receives the TOTAL number of matches. *d_count > cap means the
 * buffer was too small and nothing is delivered (what the buffer then holds is unspecified: the shares of the corpus that
 * still fitted may or may not have been written; never anything past cap): scan again with room for at least *d_count
 * records, or with twice the room when *d_count == cap + 1 (then the staging area of one
 * wavefront, sized from cap, or -- the first dense scan on a scratch -- a wavefront's candidate region
 * was too small for a dense run of matches; in that case *d_count is a lower bound, and the scratch
 * remembers: its next scans give every 16-byte chunk candidate room of its own).
 * hsgpu_hwlm_exec_batch repeats the scan itself. cap < 2^32. d_corpus must be
 * 16-byte aligned. Successive scans on one scratch must be ordered after each other (the same stream, or
 * synchronised): a scratch is one set of working buffers, as an hs_scratch is (one per concurrent scan). */
int hsgpu_hwlm_scan_dev(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const void *d_corpus,
                        uint64_t total_bytes, const void *d_off, uint64_t nblocks, uint64_t start,
                        void *d_out, uint64_t cap, void *d_count, void *stream);

/* Measurement aid: when enabled, hsgpu_hwlm_scan_dev records two HIP events on the launch
 * stream, right before and right after its dominant (filter) kernel, and the kernels stamp
 * the device wall clock at their own start / end (a ring of the last 32 scans).
 * hsgpu_scratch_get_timing synchronises the device and returns, for the scan `back`
 * launches ago (0 = the last), the filter kernel's duration between the two events, and --
 * from the device clock -- the confirm stage (filter end to confirm end) and the whole
 * pipeline (filter start to the end of the scan's last kernel), in milliseconds. (The confirm stage's span ends where the
 * kernel that places the records starts -- a gather behind the default pipeline, whose confirm workers emit in delivery
 * order; a sort behind fused scans -- the pipeline's where that kernel's first workgroup ends.) */
int hsgpu_scratch_enable_timing(hsgpu_scratch_t *s, int enable);
int hsgpu_scratch_get_timing(hsgpu_scratch_t *s, unsigned back, float *filter_ms, float *confirm_ms,
                             float *total_ms);
/* The filter kernel's own execution span (first workgroup start .. last workgroup end, device
 * wall clock), i.e. what a kernel trace reports; HIP events additionally contain dispatch gaps. */
int hsgpu_scratch_get_kernel_span(hsgpu_scratch_t *s, unsigned back, float *filter_ms);

/* Tuning / test aid (synchronises the device): 32-byte candidate entries spilled by the
 * filter kernel since the previous call, and how many scans since then overflowed a
 * candidate region (and were redone by the fused fallback kernel). */
int hsgpu_scratch_get_stats(hsgpu_scratch_t *s, uint64_t *cand_entries, int *overflowed);

/* The host-buffer scan as a pipeline: the batch is cut into chunks of whole blocks (chunk_bytes, 0 = 64 MiB); the
 * copy of chunk i + 1 runs beside the scan of chunk i, and on_chunk receives the records of every finished chunk --
 * delivery order, block indices of the whole batch, chunks in block order -- on the CALLING thread while later
 * chunks are still being copied and scanned: whatever the caller does per chunk (confirm, callbacks) hides behind
 * the bus. A non-zero return from on_chunk stops the scan (HSGPU_SCAN_TERMINATED). Offsets that are not ascending (or a
 * block of 4 GiB or more) are refused with HSGPU_INVALID before a byte of the batch has been read and before on_chunk has
 * been called; a failure later on (device memory, a device error) ends the call after the chunks in front of it have been
 * delivered. Nothing of the batch stays resident afterwards. No reference counterpart: the reference has no device boundary
 * (doc/dev-reference/performance.rst:56-61). */
typedef int (*hsgpu_chunk_cb)(const hsgpu_match_t *recs, size_t n, void *ctx);
int hsgpu_hwlm_exec_batch_cb(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const uint8_t *base, const uint64_t *off,
                             size_t nblocks, size_t start, size_t chunk_bytes, hsgpu_chunk_cb on_chunk, void *ctx);

/* The delivery order (block, end, lit) for records on the host, e.g. after merging the records of
 * several scans; multi-threaded above 64 Ki records. Touches no device. */
void hsgpu_match_sort_host(hsgpu_match_t *recs, size_t n);

/* Replay sorted records of ONE block through a callback with the reference's
 * sequential semantics (groups gate, noruns, terminate). Host only. */
int hsgpu_hwlm_replay(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n,
                      hsgpu_hwlm_cb cb, void *ctx, uint64_t groups);
/* The same for the sorted records of a whole batch (hsgpu_hwlm_scan_dev / _exec_batch): the rules start
 * afresh in every block, a callback that returns 0 ends its own block only; *n_terminated (may be NULL)
 * counts those blocks. Returns HSGPU_HWLM_SUCCESS or HSGPU_HWLM_ERROR_UNKNOWN. */
int hsgpu_hwlm_replay_batch(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n, hsgpu_hwlm_cb cb,
                            void *ctx, uint64_t groups, size_t *n_terminated);
/* The same on n_threads host threads: the blocks are cut into n_threads contiguous ranges and range i is walked,
 * in order, by one thread with ctxs[i] as the callbacks' context -- callbacks of DIFFERENT blocks run
 * concurrently (hsbench's -T model, tools/hsbench/main.cpp:957-963); within a block nothing changes. */
int hsgpu_hwlm_replay_batch_mt(const hsgpu_hwlm_t *t, const hsgpu_match_t *recs, size_t n, hsgpu_hwlm_cb cb,
                               void *const *ctxs, unsigned n_threads, uint64_t groups, size_t *n_terminated);
/* From a device-resident scan (hsgpu_hwlm_scan_dev's d_out / d_count, `cap` as given there) to callbacks: waits
 * for the scan on `stream`, brings the records to pinned host memory owned by the scratch in chunks and replays
 * the blocks that have arrived on n_threads threads (as hsgpu_hwlm_replay_batch_mt) while the next chunk is on
 * the wire. *n_records = the scan's count; HSGPU_INSUFFICIENT_SPACE when it exceeds cap (nothing is replayed). */
int hsgpu_hwlm_fetch_replay(const hsgpu_hwlm_t *t, hsgpu_scratch_t *s, const void *d_out, uint64_t cap,
                            const void *d_count, void *stream, hsgpu_hwlm_cb cb, void *const *ctxs, unsigned n_threads,
                            uint64_t groups, size_t *n_records, size_t *n_terminated);
/* hsbench's counting callback (tools/hsbench/engine_hyperscan.cpp:89-97): ctx = uint64_t counter. */
uint64_t hsgpu_hwlm_count_cb(size_t end, uint32_t id, void *ctx);

/* ---- the exchange step of a multi-GPU job (csrc/exchange.hip) ------------------------------------
 * Blocks are independent scans: a job over N GPUs (one process per GPU) shards the corpus by contiguous block ranges, every
 * rank scans its shard with hsgpu_hwlm_scan_dev, and ONE collective per step brings the match records together -- at the
 * rank whose host delivers the callbacks (HSGPU_XCHG_TO_ROOT: what hsbench's threads do with their result table,
 * tools/hsbench/main.cpp:957-963, 990-1030; on xGMI every peer has a link of its own to the root), or at every rank
 * (HSGPU_XCHG_ALL_GATHER). RCCL directly (loaded at run time); no reference counterpart: the reference has no device boundary.
 *   wire record   12 bytes {GLOBAL block index, end, id}
 *   unique_id     rank 0 obtains 128 bytes and hands them to the other ranks by whatever means the job has (MPI, a file, a
 *                 torch.distributed broadcast); world == 1 may pass NULL (no communicator is made)
 *   create        collective over all ranks. rows_per_rank: the records a rank's slot holds (agree on the largest expected
 *                 count + slack); a slot travels whole unless set_counts told every rank the exact rows of every rank
 *                 (steps that repeat a scan: the counts are known from the first one)
 *   step          packs the records of the scan that wrote d_records / d_count (hsgpu_hwlm_scan_dev's buffers, still on the
 *                 device; record_cap = the `cap` that scan was given: a scan that found more wrote nothing usable, its slot
 *                 then travels with no rows and compact reports HSGPU_INSUFFICIENT_SPACE; first_block = the global index of
 *                 this rank's block 0) and posts the transfers, all on `stream`: no host synchronisation, nothing allocated
 *   compact       (synchronises) the records of the last step in rank order = corpus order into d_out (device, hsgpu_wire_t
 *                 [cap]), counts[r] = what rank r's scan found, *total = records delivered; HSGPU_INSUFFICIENT_SPACE when
 *                 a scan found more than its slot (or cap) holds. On ranks that receive nothing (TO_ROOT, not the root)
 *                 *total = 0. */
#define HSGPU_XCHG_ID_BYTES 128
#define HSGPU_XCHG_TO_ROOT 0u
#define HSGPU_XCHG_ALL_GATHER 1u
typedef struct hsgpu_exchange hsgpu_exchange_t;
typedef struct hsgpu_wire {
    uint32_t block, end, id;
} hsgpu_wire_t;
int hsgpu_exchange_unique_id(void *id /* HSGPU_XCHG_ID_BYTES */);
int hsgpu_exchange_create(hsgpu_exchange_t **x, const void *id, int world, int rank, int device, uint64_t rows_per_rank,
                          unsigned mode, int root);
int hsgpu_exchange_set_counts(hsgpu_exchange_t *x, const uint64_t *rows /* [world], NULL = fixed slots again */, int world);
int hsgpu_exchange_step(hsgpu_exchange_t *x, const void *d_records, uint64_t record_cap, const void *d_count, uint64_t first_block,
                        void *stream);
/* An id for VIRTUAL ranks: `world` exchanges created with it in ONE process on one device talk to each other by device copies
 * instead of RCCL (a Send meeting its Recv = one hipMemcpyAsync, ordered by events; the ranks may be driven from one thread in
 * any order, each on a stream of its own or all on one). Everything else -- who sends what to whom, slot sizes, agreed
 * counts, compact -- is the code the RCCL transport runs: the way to run an N > 1 job's exchange on a 1-GPU box. */
int hsgpu_exchange_loopback_id(void *id /* HSGPU_XCHG_ID_BYTES */);
int hsgpu_exchange_compact(hsgpu_exchange_t *x, void *d_out, uint64_t cap, uint64_t *counts /* [world], host */, uint64_t *total,
                           void *stream);
/* bytes this rank sends / receives in one step (the figure a link budget needs) */
int hsgpu_exchange_wire_bytes(const hsgpu_exchange_t *x, uint64_t *sent, uint64_t *received);
void hsgpu_exchange_free(hsgpu_exchange_t *x);

/* ---- character-class scanning ------------------------------------------------------
 * The GPU form of the reference's class accelerators: shuftiExec/rshuftiExec
 * (src/nfa/shufti.h:46-52), truffleExec/rtruffleExec (src/nfa/truffle.h:45-50),
 * vermicelliExec/nvermicelliExec/rvermicelliExec (src/nfa/vermicelli.h:42-518), dispatched
 * by run_accel (src/nfa/accel.c:35-146). Every scheme decodes to a 256-bit class. */
#define HSGPU_CLASS_MAX 8          /* classes per call when first / last are asked for */
#define HSGPU_CLASS_MAX_BITMAPS 16 /* ... for the membership bitmaps alone (d_first == d_last == NULL): one read of the corpus */
#define HSGPU_CLASS_WORK_BYTES 8192 /* device work area for hsgpu_class_scan_dev */

typedef struct hsgpu_class {
    uint8_t bitmap[32]; /* bit v (LSB first) set <=> byte value v is a member */
} hsgpu_class_t;

int hsgpu_class_from_shufti(const uint8_t lo[16], const uint8_t hi[16], hsgpu_class_t *out);
int hsgpu_class_from_truffle(const uint8_t mask1[16], const uint8_t mask2[16], hsgpu_class_t *out);
int hsgpu_class_from_verm(uint8_t c, int nocase, int negate, hsgpu_class_t *out);
int hsgpu_class_to_truffle(const hsgpu_class_t *cls, uint8_t mask1[16], uint8_t mask2[16]);

/* Evaluate n_classes <= 8 classes over a block batch resident in HBM, asynchronously
 * on `stream`. d_bitmaps[c]: device buffer of (total_bytes + 15) / 16 * 2 bytes that
 * receives membership bitmap c (bit i, LSB first <=> corpus[i] in class c).
 * d_first / d_last (optional, uint32 [n_classes][nblocks]): per block the offset of the
 * first / last member -- the accelerators' return value relative to the block start --
 * or the block length / 0xffffffff when there is none (shufti.h:40-52). d_work:
 * HSGPU_CLASS_WORK_BYTES of 16-byte aligned device memory. */
int hsgpu_class_scan_dev(const hsgpu_class_t *classes, unsigned n_classes, const void *d_corpus,
                         uint64_t total_bytes, const void *d_off, uint64_t nblocks, void *const *d_bitmaps,
                         void *d_first, void *d_last, void *d_work, void *stream);

/* ---- class-sequence patterns over the class bitmaps -----------------------------------
 * The consumer of the accelerators' answer. In the reference a literal-less pattern such as
 * [a-z]{3,}\d+ runs in an NFA / DFA engine that idles on run_accel (src/nfa/accel.c:35-146, callers
 * src/nfa/limex_accel.c:49-74, src/nfa/mcclellan.c:92-120) and reports every match end through its
 * callback. With the membership bitmaps of hsgpu_class_scan_dev the pattern A{m,}B{n,} is bit-parallel
 * (csrc/class_seq.hip): the match ends of every pattern, per block, without a host engine.
 *   a, b   indices into the call's bitmap list; m, n in 1 .. HSGPU_SEQ_MAX_REPEAT ("+" is {1,})
 * A pattern matches ending at byte e of a block iff some s <= e exists with bytes s..e all in B,
 * e - s + 1 >= n, and the m bytes before s all in A, everything inside the block: the `to - 1` offsets
 * hs_scan reports for the expression (all matches, no start of match).
 * d_counts: uint64 [n_seqs], matches of every pattern over the whole batch (always). Records
 * {block, end, id, lit = pattern index} are written, in no particular order, only for match ends whose
 * corpus byte lies in [emit_lo, emit_hi): 256 class-heavy patterns report several matches per corpus
 * byte, more than any buffer holds; *d_count = records in that range (may exceed cap: the first cap were
 * written). d_work: hsgpu_class_seq_work_bytes(total_bytes), 16-byte aligned. Asynchronous on `stream`. */
#define HSGPU_SEQ_MAX 1024
#define HSGPU_SEQ_MAX_REPEAT 16
typedef struct hsgpu_class_seq {
    uint8_t a, b; /* class of the repeated prefix, class of the tail */
    uint8_t m, n; /* A{m,} B{n,} */
    uint32_t id;  /* what the records carry as id */
} hsgpu_class_seq_t;
size_t hsgpu_class_seq_work_bytes(uint64_t total_bytes);
int hsgpu_class_seq_scan_dev(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps,
                             unsigned n_classes, uint64_t total_bytes, const void *d_off, uint64_t nblocks,
                             uint64_t emit_lo, uint64_t emit_hi, void *d_counts, void *d_out, uint64_t cap,
                             void *d_count, void *d_work, size_t work_bytes, void *stream);

/* The records of a byte range made of WHOLE blocks only (emit_lo / emit_hi = offsets of two block starts): the parts of the
 * corpus that hold no block of the range are not walked and nothing is counted -- what a caller uses that delivers a
 * match-dense batch range by range through a bounded record buffer. Needs 8-byte aligned bitmaps, at most 32 classes. */
int hsgpu_class_seq_emit_dev(const hsgpu_class_seq_t *seqs, unsigned n_seqs, const void *const *d_bitmaps,
                             unsigned n_classes, uint64_t total_bytes, const void *d_off, uint64_t nblocks,
                             uint64_t emit_lo, uint64_t emit_hi, void *d_out, uint64_t cap, void *d_count, void *d_work,
                             size_t work_bytes, void *stream);

/* The same for a batch in HOST memory (the sibling of hsgpu_hwlm_exec_batch): class bitmaps in passes of <= 8
 * classes, then the sequence kernel; `out` receives every match of the batch in delivery order (block, end,
 * pattern index), *nout the number there is (HSGPU_INSUFFICIENT_SPACE when more than cap: nothing is written
 * then), counts (optional, uint64 [n_seqs]) the matches per pattern. reuse_resident != 0: the batch is the one this
 * scratch uploaded last (hsgpu_hwlm_exec_batch / this function, same base and offsets) and is not copied again. */
int hsgpu_class_seq_exec_batch(const hsgpu_class_t *classes, unsigned n_classes, const hsgpu_class_seq_t *seqs,
                               unsigned n_seqs, hsgpu_scratch_t *s, const uint8_t *base, const uint64_t *off,
                               size_t nblocks, int reuse_resident, uint64_t *counts, hsgpu_match_t *out, size_t cap,
                               size_t *nout);
/* ... for the blocks [block_lo, block_hi) of the batch only, in delivery order: a batch whose patterns match several times per
 * byte is delivered range by range through a buffer of bounded size (HSGPU_INSUFFICIENT_SPACE, *nout = the room needed: ask for
 * fewer blocks). reuse_resident as above; reuse_bitmaps != 0: the class bitmaps of the previous call on this scratch (same batch,
 * same classes) are still there and are not computed again. */
int hsgpu_class_seq_exec_blocks(const hsgpu_class_t *classes, unsigned n_classes, const hsgpu_class_seq_t *seqs,
                                unsigned n_seqs, hsgpu_scratch_t *s, const uint8_t *base, const uint64_t *off, size_t nblocks,
                                int reuse_resident, int reuse_bitmaps, size_t block_lo, size_t block_hi, hsgpu_match_t *out,
                                size_t cap, size_t *nout);

/* ---- choosing an accelerator for a literal set (host only) ---------------------------
 * buildForwardAccel / findForwardAccelScheme (src/rose/rose_build_lit_accel.cpp:372-465): the
 * scheme hwlmExec's pre-skip uses (do_accel_block, src/hwlm/hwlm.c:48-99). `type` takes the
 * values of enum AccelType (src/nfa/accel.h:46-65); `offset` is how far before the
 * accelerator's hit a literal may start. c1 (and c2 for the pair schemes) are upper-cased for
 * the _NOCASE types. mask_lo / mask_hi: shufti lo / hi, or truffle mask1 / mask2. */
#define HSGPU_ACCEL_NONE 0
#define HSGPU_ACCEL_VERM 1
#define HSGPU_ACCEL_VERM_NOCASE 2
#define HSGPU_ACCEL_DVERM 3
#define HSGPU_ACCEL_DVERM_NOCASE 4
#define HSGPU_ACCEL_SHUFTI 13
#define HSGPU_ACCEL_TRUFFLE 15

typedef struct hsgpu_accel {
    uint8_t type, offset, c1, c2;
    uint8_t mask_lo[16], mask_hi[16];
} hsgpu_accel_t;

/* Scheme for the literals whose groups intersect expected_groups (HSGPU_ALL_GROUPS for the
 * reference's accel0). HSGPU_ACCEL_NONE when nothing narrower than 240 byte values exists. */
int hsgpu_accel_forward(const hsgpu_lit_t *lits, size_t n, uint64_t expected_groups, hsgpu_accel_t *out);
/* shuftiBuildMasks (src/nfa/shufticompile.cpp:54-109): number of buckets used, or -1 when
 * the class needs more than 8 (the caller then uses truffle, which represents any class). */
int hsgpu_class_to_shufti(const hsgpu_class_t *cls, uint8_t lo[16], uint8_t hi[16]);

/* ---- two-byte accelerators ---------------------------------------------------------
 * shuftiDoubleExec (src/nfa/shufti.c:319-361), vermicelliDoubleExec /
 * vermicelliDoubleMaskedExec / rvermicelliDoubleExec (src/nfa/vermicelli.h:169-317,464-518).
 * Every one of them tests byte PAIRS against a union of <= 8 "rectangles" (a set of first
 * bytes x a set of second bytes, each a product of a low-nibble set and a high-nibble set),
 * which is exactly what double-shufti masks encode (0-active, one bit per bucket):
 *   t(c) = lo1[c & 15] | hi1[c >> 4],   u(c) = lo2[c & 15] | hi2[c >> 4]
 *   (buf[i], buf[i+1]) matches  <=>  (t(buf[i]) | u(buf[i+1])) != 0xff
 * Forward result per block (the accelerators' return value relative to the block start):
 * the first matching i; else len - 1 when the last byte alone passes t (the "partial match
 * at end" of vermicelli.h:179-186, which the caller re-examines); else len.
 * The reference's shuftiDoubleExec may stop EARLIER than that, at a first-byte-only hit in
 * the last lane of one of its 16-byte vectors (shufti.c:224): an artefact of its vector
 * width that callers tolerate by design (run_accel only promises not to skip a match).
 * Reverse result: offset of the SECOND byte of the last pair, 0xffffffff when none
 * (unit/internal/rvermicelli.cpp:116-140). */
#define HSGPU_PAIR_MAX 8
#define HSGPU_PAIR_WORK_BYTES 8256 /* device work area for hsgpu_pair_scan_dev */

typedef struct hsgpu_pair {
    uint8_t lo1[16], hi1[16], lo2[16], hi2[16];
} hsgpu_pair_t;

int hsgpu_pair_from_dshufti(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                            const uint8_t hi2[16], hsgpu_pair_t *out);
int hsgpu_pair_from_dverm(uint8_t c1, uint8_t c2, int nocase, hsgpu_pair_t *out);
int hsgpu_pair_from_dverm_masked(uint8_t c1, uint8_t c2, uint8_t m1, uint8_t m2, hsgpu_pair_t *out);
/* shuftiBuildDoubleMasks (src/nfa/shufticompile.cpp:135-209): masks for npairs two-byte
 * sequences (pairs[2i], pairs[2i+1]) plus the single bytes of `onechar` (may be NULL) with
 * any second byte. HSGPU_COMPILER_ERROR when more than 8 buckets would be needed. */
int hsgpu_pair_build(const hsgpu_class_t *onechar, const uint8_t *pairs, size_t npairs, hsgpu_pair_t *out);
/* Does the pair (a, b) match? (host-side decode, for tests and callers' own checks) */
int hsgpu_pair_test(const hsgpu_pair_t *p, uint8_t a, uint8_t b);

/* n_pairs <= 8 pair sets over a block batch, asynchronously on `stream`. d_bitmaps[k]:
 * (total_bytes + 15) / 16 * 2 bytes; bit i <=> (corpus[i], corpus[i+1]) matches set k
 * (i + 1 < total_bytes; the bitmap knows no block boundaries). d_first / d_last (optional,
 * uint32 [n_pairs][nblocks]) as described above, pairs never straddle two blocks. */
int hsgpu_pair_scan_dev(const hsgpu_pair_t *pairs, unsigned n_pairs, const void *d_corpus, uint64_t total_bytes,
                        const void *d_off, uint64_t nblocks, void *const *d_bitmaps, void *d_first, void *d_last,
                        void *d_work, void *stream);

/* ---- hwlmExec's pre-skip for a block batch ---------------------------------------------
 * do_accel_block (src/hwlm/hwlm.c:80-99) per block, asynchronously on `stream`: where at least
 * 16 bytes follow the block's start, start' = max(0, hit - aux->offset), hit being what
 * run_hwlm_accel (hwlm.c:48-77) returns for [start, len): the first member of the scheme's class,
 * or for double vermicelli the first pair, else len - 1 when the last byte alone is c1, else len.
 * Other blocks, and HSGPU_ACCEL_NONE, keep their start. d_start_in: uint32 [nblocks] starts, or NULL
 * for `start` everywhere; d_start_out: uint32 [nblocks]. d_bitmap: (total_bytes + 15) / 16 * 2
 * bytes of device scratch (the scheme's membership bitmap; unused for HSGPU_ACCEL_NONE); d_work:
 * HSGPU_PAIR_WORK_BYTES, 16-byte aligned. The literal scan itself takes no pre-skip (it reads every
 * byte once, at constant cost); this entry point is for callers that consume `start`. */
int hsgpu_hwlm_forward_skip_dev(const hsgpu_accel_t *aux, const void *d_corpus, uint64_t total_bytes,
                                const void *d_off, uint64_t nblocks, const void *d_start_in, uint32_t start,
                                void *d_start_out, void *d_bitmap, void *d_work, void *stream);

/* ---- run_accel for a block batch ----------------------------------------------------------
 * hsgpu_accel_aux_t has the layout of the reference's union AccelAux (src/nfa/accel.h:66-113: type, offset, the
 * vermicelli bytes and masks at bytes 2..5, the 16-byte masks at 16, 32, 48, 64; 80 bytes), so an AccelAux taken
 * from a compiled NFA / DFA can be handed over as it is. out[b] (uint32) = run_accel(aux, buf + start, buf + len) - buf
 * for every block (src/nfa/accel.c:35-146): all ten cases it dispatches -- NONE, VERM, VERM_NOCASE, DVERM,
 * DVERM_NOCASE, DVERM_MASKED, SHUFTI, DSHUFTI, TRUFFLE, RED_TAPE -- with their minimum-length rules, the two-byte
 * schemes stopping one byte early, and the offset adjustment max(c + offset, rv) - offset. d_start_in: uint32
 * [nblocks] or NULL for `start` everywhere; d_bitmap: (total_bytes + 15) / 16 * 2 bytes of device scratch and
 * d_work: HSGPU_PAIR_WORK_BYTES, 16-byte aligned (neither is used by NONE / RED_TAPE). For double shufti the
 * result is the exact first pair (see "two-byte accelerators" above: the reference may stop earlier at a
 * first-byte-only hit in the last lane of one of its vectors; callers tolerate either). */
#define HSGPU_ACCEL_DSHUFTI 14
#define HSGPU_ACCEL_RED_TAPE 16
#define HSGPU_ACCEL_DVERM_MASKED 17
typedef struct hsgpu_accel_aux {
    uint8_t accel_type, offset;
    uint8_t c1, c2; /* verm.c = c1; dverm.c1 / c2 */
    uint8_t m1, m2; /* dverm masked variant */
    uint8_t pad[10];
    uint8_t mask[4][16]; /* shufti lo, hi | dshufti lo1, hi1, lo2, hi2 | truffle mask1, mask2 */
} hsgpu_accel_aux_t;
int hsgpu_run_accel_dev(const hsgpu_accel_aux_t *aux, const void *d_corpus, uint64_t total_bytes, const void *d_off,
                        uint64_t nblocks, const void *d_start_in, uint32_t start, void *d_out, void *d_bitmap,
                        void *d_work, void *stream);

const char *hsgpu_last_error(void);
const char *hsgpu_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HSGPU_H */
