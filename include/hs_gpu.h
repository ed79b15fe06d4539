/*
 * hs_gpu.h -- the public Hyperscan block-mode API surface, served by the GPU literal
 * engine (libhsgpu.so) plus a small host-side confirm ("Rose-lite").
 *
 * Names, argument meaning, error codes and callback protocol are the reference's:
 *   hs_compile / hs_compile_multi            src/hs_compile.h:360,443
 *   hs_compile_lit / hs_compile_lit_multi    src/hs_compile.h:608,690
 *   hs_free_compile_error                    src/hs_compile.h:710
 *   hs_alloc_scratch / hs_clone_scratch /
 *   hs_scratch_size / hs_free_scratch        src/hs_runtime.h:555-609
 *   hs_scan                                  src/hs_runtime.h:479-482
 *   match_event_handler                      src/hs_runtime.h:125-129
 *   hs_free_database / hs_database_size /
 *   hs_database_info / hs_serialize_database /
 *   hs_deserialize_database                  src/hs_common.h:84-271
 *   hs_version / hs_valid_platform           src/hs_common.h:450,467
 *   hs_compile_ext_multi / hs_expr_ext_t     src/hs_compile.h:244-310,534
 *   hs_expression_info / hs_expression_ext_info / hs_expr_info_t   src/hs_compile.h:160-236,760-821
 *   hs_populate_platform                     src/hs_compile.h:833
 *   hs_serialized_database_size / _info      src/hs_common.h:196-271
 *   hs_set_allocator / hs_set_{database,misc,scratch,stream}_allocator  src/hs_common.h:273-439
 *   hs_scan_vector (HS_MODE_VECTORED)        src/hs_runtime.h:480-527, src/runtime.c:1106-1174
 *   hs_deserialize_database_at               src/hs_common.h:147-169 (a header in the caller's memory: see there)
 * Not provided: streaming scans (the hs_open_stream family: the literal path this engine
 * replaces is block-shaped).
 * Literal-less class sequences A{m,}B{n,} ([a-z]{3,}\d+ style) are accepted too: they are evaluated on the GPU
 * from class bitmaps (csrc/class_seq.hip), the job the reference gives to an accelerated NFA / DFA.
 *
 * What is behind it: an expression is one or more top-level branches `b1|b2|...`, and every
 * branch must contain a mandatory literal (>= 1 byte) at its top level: R1 LIT R2, where R1 and
 * R2 are regex fragments (either may be empty; LIT is the longest top-level run of plain
 * characters, the one at the front on a tie) and `^` may lead the branch. A branch whose only
 * literals sit inside an unquantified group, X(A|B)Y, is distributed into XAY|XBY first
 * ("\b(foo|bar)\b", "(GET|POST) /"). The literals
 * (their last <= 8 bytes, as Rose truncates them: rose_build_matchers.cpp:717-724)
 * are matched on the GPU through hsgpu_hwlm_exec; the host then checks the full literal
 * (the job of CHECK_MED_LIT / CHECK_LONG_LIT, src/rose/program_runtime.c:2896-2942) and,
 * runs bit-parallel NFAs over the bytes that follow (R2, forwards) and precede (R1, backwards
 * from the literal: the job of Rose's suffix / prefix engines). Supported fragment syntax: literal
 * characters, escapes, `.`, \d \D \w \W \s \S, [...] classes (with [:posix:] names), the
 * quantifiers ? * + {m} {m,} {m,n} and their lazy forms (every end offset is reported, so greed
 * is immaterial), (?ims-ims) options in front, (?is-is) settings and (?i:...) groups inside, groups `( )` / `(?: )` / named / `(?# )` with
 * alternation inside them, nested and
 * quantified (a fragment compiles to a position automaton of <= 4096 positions, run LimEx-style:
 * one shift for the chains, exception rows for the rest). Anchors and
 * assertions at the edges of a branch: `^` / \A in front, `$` / \z / \Z at the back (`$` and \Z
 * also before the data's final newline, reported before the newline as the reference does; with
 * HS_FLAG_MULTILINE `^` / `$` also match after / before any newline). \b / \B anywhere
 * (inside a fragment they become conditional layers of its automaton).
 * HS_FLAG_UTF8 (without HS_FLAG_UCP): `.`, negated classes and \W \D \S take whole code points, a
 * non-ASCII character is one atom, \x{...} names a code point, classes hold code points and
 * code-point ranges, caseless k / s also match U+212A / U+017F; caseless non-ASCII letters and
 * \h \v in classes are refused.
 * Anything else (branches without a mandatory literal, anchors away from the edges of a branch,
 * look-around, back-references, possessive quantifiers, (?m) / (?x) after the start, streaming
 * mode) is
 * rejected with HS_COMPILER_ERROR:
 * the regex compiler proper is out of scope.
 */
#ifndef HS_GPU_H
#define HS_GPU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden and an export list (csrc/Makefile: libhsgpu.map, made from these headers):
 * what is declared between here and the pop is the whole exported surface, as hs.def / hs_runtime.def are the reference's */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef int hs_error_t;
struct hs_database;
typedef struct hs_database hs_database_t;
struct hs_scratch;
typedef struct hs_scratch hs_scratch_t;

#define HS_SUCCESS 0
#define HS_INVALID (-1)
#define HS_NOMEM (-2)
#define HS_SCAN_TERMINATED (-3)
#define HS_COMPILER_ERROR (-4)
#define HS_DB_VERSION_ERROR (-5)
#define HS_DB_PLATFORM_ERROR (-6)
#define HS_DB_MODE_ERROR (-7)
#define HS_BAD_ALIGN (-8)
#define HS_BAD_ALLOC (-9)
#define HS_SCRATCH_IN_USE (-10)
#define HS_ARCH_ERROR (-11)
#define HS_INSUFFICIENT_SPACE (-12)
#define HS_UNKNOWN_ERROR (-13)

#define HS_FLAG_CASELESS 1
#define HS_FLAG_DOTALL 2
#define HS_FLAG_MULTILINE 4
#define HS_FLAG_SINGLEMATCH 8
#define HS_FLAG_ALLOWEMPTY 16
#define HS_FLAG_UTF8 32
#define HS_FLAG_UCP 64
#define HS_FLAG_PREFILTER 128
#define HS_FLAG_SOM_LEFTMOST 256
#define HS_FLAG_COMBINATION 512
#define HS_FLAG_QUIET 1024

#define HS_MODE_BLOCK 1
#define HS_MODE_NOSTREAM 1
#define HS_MODE_STREAM 2
#define HS_MODE_VECTORED 4
#define HS_MODE_SOM_HORIZON_LARGE (1U << 24)
#define HS_MODE_SOM_HORIZON_MEDIUM (1U << 25)
#define HS_MODE_SOM_HORIZON_SMALL (1U << 26)

/* hs_platform_info_t fields (src/hs_compile.h:1010-1134): validated as the reference does,
 * otherwise ignored -- the engine targets gfx950 whatever the host CPU is */
#define HS_CPU_FEATURES_AVX2 (1ULL << 2)
#define HS_CPU_FEATURES_AVX512 (1ULL << 3)
#define HS_CPU_FEATURES_AVX512VBMI (1ULL << 4)
#define HS_TUNE_FAMILY_GENERIC 0
#define HS_TUNE_FAMILY_ICX 10

typedef struct hs_compile_error {
    char *message;
    int expression;
} hs_compile_error_t;

typedef struct hs_platform_info {
    unsigned int tune;
    unsigned long long cpu_features;
    unsigned long long reserved1;
    unsigned long long reserved2;
} hs_platform_info_t;

/* hs_expr_ext_t, src/hs_compile.h:244-310 */
typedef struct hs_expr_ext {
    unsigned long long flags;
    unsigned long long min_offset; /* matches must END at or after this offset */
    unsigned long long max_offset; /* ... and at or before this one */
    unsigned long long min_length; /* to - from must be at least this */
    unsigned edit_distance;        /* not supported here: compile error if flagged */
    unsigned hamming_distance;     /* not supported here */
} hs_expr_ext_t;
#define HS_EXT_FLAG_MIN_OFFSET 1ULL
#define HS_EXT_FLAG_MAX_OFFSET 2ULL
#define HS_EXT_FLAG_MIN_LENGTH 4ULL
#define HS_EXT_FLAG_EDIT_DISTANCE 8ULL
#define HS_EXT_FLAG_HAMMING_DISTANCE 16ULL

/* hs_expr_info_t, src/hs_compile.h:160-236 */
typedef struct hs_expr_info {
    unsigned int min_width;
    unsigned int max_width; /* UINT_MAX: unbounded */
    char unordered_matches;
    char matches_at_eod;
    char matches_only_at_eod;
} hs_expr_info_t;

typedef void *(*hs_alloc_t)(size_t size);
typedef void (*hs_free_t)(void *ptr);

typedef int (*match_event_handler)(unsigned int id, unsigned long long from, unsigned long long to,
                                   unsigned int flags, void *context);

hs_error_t hs_compile(const char *expression, unsigned int flags, unsigned int mode,
                      const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error);
hs_error_t hs_compile_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                            unsigned int elements, unsigned int mode, const hs_platform_info_t *platform,
                            hs_database_t **db, hs_compile_error_t **error);
hs_error_t hs_compile_lit(const char *expression, unsigned flags, const size_t len, unsigned mode,
                          const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error);
hs_error_t hs_compile_lit_multi(const char *const *expressions, const unsigned *flags, const unsigned *ids,
                                const size_t *lens, unsigned elements, unsigned mode,
                                const hs_platform_info_t *platform, hs_database_t **db,
                                hs_compile_error_t **error);
hs_error_t hs_compile_ext_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                                const hs_expr_ext_t *const *ext, unsigned int elements, unsigned int mode,
                                const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error);
hs_error_t hs_free_compile_error(hs_compile_error_t *error);
hs_error_t hs_expression_info(const char *expression, unsigned int flags, hs_expr_info_t **info,
                              hs_compile_error_t **error);
hs_error_t hs_expression_ext_info(const char *expression, unsigned int flags, const hs_expr_ext_t *ext,
                                  hs_expr_info_t **info, hs_compile_error_t **error);
hs_error_t hs_populate_platform(hs_platform_info_t *platform);

hs_error_t hs_free_database(hs_database_t *db);
hs_error_t hs_database_size(const hs_database_t *database, size_t *database_size);
hs_error_t hs_database_info(const hs_database_t *database, char **info);
/* src/hs_runtime.h:294: only streaming databases have a stream size; block and vectored ones
 * (all there are here) answer HS_DB_MODE_ERROR, as the reference does for them */
hs_error_t hs_stream_size(const hs_database_t *database, size_t *stream_size);
hs_error_t hs_serialize_database(const hs_database_t *db, char **bytes, size_t *length);
hs_error_t hs_deserialize_database(const char *bytes, const size_t length, hs_database_t **db);
hs_error_t hs_serialized_database_size(const char *bytes, const size_t length, size_t *deserialized_size);
/* src/hs_common.h:147-169. `db`: 8-byte aligned memory of at least hs_serialized_database_size bytes, owned and
 * freed by the caller. A database of this engine owns device memory (its literal table in HBM), which no caller's
 * buffer can hold: the buffer receives a header that every hs_* entry point follows. Before freeing the buffer the
 * caller releases what lies outside it with hs_free_database(db), which leaves the buffer itself alone. */
hs_error_t hs_deserialize_database_at(const char *bytes, const size_t length, hs_database_t *db);
hs_error_t hs_serialized_database_info(const char *bytes, size_t length, char **info);

/* Allocation hooks (src/hs_common.h:273-439): the objects handed to the caller -- databases,
 * scratch, compile errors, info strings, serialised bytes, expression info -- are allocated
 * with these (NULL = malloc/free) and must come back 8-byte aligned (HS_BAD_ALIGN otherwise);
 * containers inside a database use the C++ heap. */
hs_error_t hs_set_allocator(hs_alloc_t alloc_func, hs_free_t free_func);
hs_error_t hs_set_database_allocator(hs_alloc_t alloc_func, hs_free_t free_func);
hs_error_t hs_set_misc_allocator(hs_alloc_t alloc_func, hs_free_t free_func);
hs_error_t hs_set_scratch_allocator(hs_alloc_t alloc_func, hs_free_t free_func);
hs_error_t hs_set_stream_allocator(hs_alloc_t alloc_func, hs_free_t free_func);

hs_error_t hs_alloc_scratch(const hs_database_t *db, hs_scratch_t **scratch);
hs_error_t hs_clone_scratch(const hs_scratch_t *src, hs_scratch_t **dest);
hs_error_t hs_scratch_size(const hs_scratch_t *scratch, size_t *scratch_size);
hs_error_t hs_free_scratch(hs_scratch_t *scratch);

hs_error_t hs_scan(const hs_database_t *db, const char *data, unsigned int length, unsigned int flags,
                   hs_scratch_t *scratch, match_event_handler onEvent, void *context);

/* Vectored mode: the `count` segments are one logical buffer; offsets run through them. The
 * database must have been compiled with HS_MODE_VECTORED (HS_DB_MODE_ERROR otherwise, and
 * hs_scan on a vectored database likewise). */
hs_error_t hs_scan_vector(const hs_database_t *db, const char *const *data, const unsigned int *length,
                          unsigned int count, unsigned int flags, hs_scratch_t *scratch, match_event_handler onEvent,
                          void *context);

/* Extension (no reference equivalent): scan nblocks independent blocks
 * data[off[i] .. off[i+1]) in one GPU batch; events carry the block index. A non-zero
 * return from the handler stops matching in THAT block only. */
typedef int (*hs_batch_event_handler)(unsigned long long block, unsigned int id, unsigned long long from,
                                      unsigned long long to, unsigned int flags, void *context);
hs_error_t hs_scan_batch(const hs_database_t *db, const char *data, const unsigned long long *off,
                         unsigned long long nblocks, unsigned int flags, hs_scratch_t *scratch,
                         hs_batch_event_handler onEvent, void *context);

/* Extension: hs_scan_batch for a batch that is ALREADY on the device -- the caller keeps a corpus resident in HBM and scans it
 * again and again (hsbench's protocol: the corpus is loaded once, tools/hsbench/main.cpp:502-528), or it arrived there by another
 * route. d_corpus / d_off: the same bytes and offsets as data / off, in device memory, as hsgpu_hwlm_scan_dev takes them (d_corpus
 * 16-byte aligned, d_off relative to d_corpus with d_off[0] == 0); data / off: the host copy the confirm reads the bytes behind
 * every literal hit from (off[0] == 0). Literal scan on the device, its hits to the host, Rose-lite confirm, callbacks in
 * block order on the calling thread: nothing of the corpus crosses the bus. HS_INVALID for a database with literal-less
 * class-sequence patterns (their records are delivered range by range by hs_scan_batch). */
hs_error_t hs_scan_batch_resident(const hs_database_t *db, const char *data, const unsigned long long *off,
                                  unsigned long long nblocks, const void *d_corpus, const void *d_off, hs_scratch_t *scratch,
                                  hs_batch_event_handler onEvent, void *context);

/* Extension: one hs_scan per block is hsbench's block mode (tools/hsbench/engine_hyperscan.cpp:132-145) and the reference serves a
 * packet in about a microsecond; a kernel launch per call costs this engine ~20. With the small-batch server enabled on a scratch
 * (include/hsgpu.h, hsgpu_scratch_enable_server: enable 1 -- requests through the PCIe BAR where the device has a large one --, 2
 * -- through mapped host memory --, 0 off; idle_us: 0 keeps the default of 300) hs_scan / hs_scan_batch calls of up to 16 KiB are
 * served by ONE resident workgroup without a launch: ~8 us per 1 460-byte call. The workgroup ends by itself when no call has come
 * for idle_us and comes back with the next one; hs_free_scratch ends it. *_stats: calls served so far, server launches (either
 * pointer may be NULL). HS_SCRATCH_IN_USE inside a callback. */
hs_error_t hs_scratch_enable_small_batch_server(hs_scratch_t *scratch, int enable, unsigned int idle_us);
hs_error_t hs_scratch_small_batch_server_stats(hs_scratch_t *scratch, unsigned long long *calls, unsigned long long *launches);

/* Extension: only the host-side confirm of hs_scan_batch, over literal hits the caller supplies
 * (hsgpu_match_t records, include/hsgpu.h: sorted by (block, end); id = index of the branch, see
 * hs_database_literal; end = offset of the last byte of that branch's literal). Touches no
 * device. HS_INVALID for records that are out of order or out of range. */
hs_error_t hs_confirm_batch(const hs_database_t *db, const char *data, const unsigned long long *off,
                            unsigned long long nblocks, const void *records, unsigned long long n_records,
                            hs_batch_event_handler onEvent, void *context);

/* Extension: what the GPU matcher of this database is keyed on (the part of the reference's
 * dump output that names Rose's literals). Branch `index` (expressions in compile order, the
 * top-level alternatives of each left to right; HS_INVALID past the last): *bytes / *len = the
 * HWLM literal (the last <= 8 bytes of the branch's literal, not NUL-terminated, owned by the
 * database), *nocase = compared caselessly, *id = the report id of its expression. */
hs_error_t hs_database_literal(const hs_database_t *db, unsigned int index, const char **bytes, size_t *len,
                               int *nocase, unsigned int *id);

/* Ready-made batch handler that counts (hsbench's onMatch, tools/hsbench/engine_hyperscan.cpp:89-97):
 * context = unsigned long long * incremented once per match; never stops the scan. */
int hs_batch_count_handler(unsigned long long block, unsigned int id, unsigned long long from,
                           unsigned long long to, unsigned int flags, void *context);

const char *hs_version(void);
hs_error_t hs_valid_platform(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
