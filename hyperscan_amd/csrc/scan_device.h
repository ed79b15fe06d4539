/*
 * scan_device.h -- the block-mode multi-literal scan kernel for gfx950.
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
 * Mapping. The corpus is the concatenation of all blocks (CSR offsets). Every
 * wavefront streams ONE contiguous share of it (total / wavefronts bytes), 1 KiB tile
 * after 1 KiB tile, each lane one 16-byte chunk per tile. Tiles are read through
 * buffer descriptors built from wave-uniform values (one dwordx4 + one dwordx2 for the
 * 8 bytes in front of the chunk), seven tiles ahead of their use: eight register stages
 * rotate by name, nothing in the steady-state loop waits for HBM. (Contiguous shares
 * stream 8-20 % faster than workgroups walking the corpus side by side, and they make the
 * records of consecutive regions consecutive in the corpus: phase 3 sorts share by share.)
 *
 * Filter (per lookup position, all lanes): hash the 3 bytes ending there with
 * one v_mul_u32_u24, read ONE 32-bit word of the LDS-resident filter, test one
 * bit chosen by the 4th byte (optionally a second bit). With stride 2 only every
 * second byte is a lookup position: the table then also holds every literal
 * keyed one byte early, so a lookup at q catches literals ending at q and q + 1.
 * A lookup is 6.5 (one bit) to 10.5 (two bits) VALU instructions; the stride-1
 * two-bit variant is VALU-issue bound. Two filter layouts:
 *   REPL   small literal sets at stride 1: 32 identical columns, lane l reads
 *          column l & 31 -> every lane of a 32-lane LDS group hits its own bank,
 *          conflict-free by construction;
 *   hashed the default: one 2^k-word table, up to 128 KiB.
 * Candidates are collected as one 16-bit mask per lane and class.
 *
 * Confirm. Candidates are rare (a fraction of a percent of positions) but each
 * one needs a chain of dependent HBM/L2 reads (window -> hash bucket -> literal
 * -> block offsets). Inside the streaming kernel every link of that chain queues
 * behind the wavefront's own prefetches, so the default pipeline is two-phase:
 *   hwlm_filter_kernel  streams the corpus; lanes with a hit append a 32-byte entry
 *                       {chunk index, masks, 8-byte halo, 16-byte chunk} to their
 *                       wavefront's private HBM region (ballot-ranked, no atomics);
 *                       its prologue also writes the per-KiB block hints;
 *   hwlm_confirm_kernel two candidate entries per lane: the key gate (large sets: 64 Kbit
 *                       in LDS, "does any exact-table key have this hash"), exact
 *                       hash-table bucket (16 B: 4 tagged slots), (window & msk) == v of
 *                       the literal the slot names; hits are queued in LDS and resolved
 *                       64 at a time (id/size, block lookup through the hint table, bound
 *                       checks), records stored into the wavefront's region, its fill
 *                       added to the partial sum of its group of regions (one atomic).
 *                       Stride-1 tables without 2-byte keys take confirm_step_fast: the
 *                       window read straight from the entry, conditions as 0 / ~0 masks,
 *                       a third of the general step's vector instructions. The stage is
 *                       bound by the vector L1's outstanding misses;
 *   record_sort_kernel  one workgroup per share: places the share from the partial sums,
 *                       sorts its records by (block, end, literal) into the caller's
 *                       buffer (delivery order), writes *count, zeroes the control block
 *                       the NEXT scan will use (phase 3 below).
 * A fused variant (confirm inside the streaming kernel) is kept as the
 * always-correct fallback for inputs so dense that the candidate buffer
 * overflows (the role of the reference's flood path, flood_runtime.h:86-335);
 * it runs alone where a scratch asks for it (tests, tuning) and behind the other two only
 * where mapped host memory for the "again" note is unavailable.
 *
 * Block boundaries are invisible to the filter; the confirm step resolves the
 * block of a candidate and rejects matches that would start before their block
 * (or before `start` within it).
 */
#ifndef HSGPU_SCAN_DEVICE_H
#define HSGPU_SCAN_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "scan_kernels.h"
#include "table.h"

namespace {


#ifndef HSGPU_STAGES
#define HSGPU_STAGES 8 /* prefetch depth of the filter kernel, in 1 KiB tiles per wavefront: 4, 6 or 8 */
#endif
#ifndef HSGPU_LOAD_AUX
#define HSGPU_LOAD_AUX 0 /* cache policy of the corpus loads (buffer-load aux bits): 0 default, 2 nt (read once) */
#endif
constexpr int WG_THREADS = HSGPU_WG_THREADS; /* largest workgroup: 16 wavefronts; small tables run 8 */
constexpr int CHUNK = 16;                     /* bytes per lane per iteration */
constexpr int WAVE_TILE = 64 * CHUNK;         /* 1 KiB */
/* a workgroup of W wavefronts owns a W KiB super-tile per iteration (16 or 8 KiB) */
constexpr int QCAP = 128;                     /* fused: candidate-queue entries per wavefront */
constexpr int OCAP = 28;                      /* staged match records per wavefront */
constexpr int CONFIRM_THREADS = HSGPU_CONFIRM_THREADS;
constexpr int OFLUSH = 12;                    /* flush the staged records at this fill */

/* per-wavefront LDS area: candidate queue (fused kernel) + staged output records */
struct WaveLds {
    uint2 cand[QCAP];
    uint4 rec[OCAP];
    uint32_t nrec;   /* records staged in rec[] */
    uint32_t nfront; /* records already appended to the front of this wavefront's HBM region */
    uint32_t nback;  /* records spilled to the back of the region (staging full mid-drain) */
    uint32_t nmq;    /* confirm kernel: queued matches (in cand[]) */
    uint32_t nrq;    /* confirm kernel: queued entries with candidate bits left */
    uint32_t pad[11]; /* [0] confirm kernel: a match was resolved in place (emitted out of order); dense scans with run tables (RUN_*):
                       * [1] records of the current region that exist only as run descriptors, [2] the region's runs, [3..5] the last
                       * run's first record / records per lookup / further lookups, [6..7] the corpus offset where it would go on,
                       * [8] its byte value (x 0x01010101), [9] its block, [10] 1 = [3..9] describe a run that may go on */
};
constexpr uint32_t SOLO_LDS_WORDS = 28 * 1024 / 4; /* what solo_tail uses of the (dead) filter image as scratch: flags, sort buffer, prefix array, list of large regions */
constexpr int RUN_VIRT = 1, RUN_N = 2, RUN_PB = 3, RUN_NM = 4, RUN_REPS = 5, RUN_NEXT = 6, RUN_VV = 8, RUN_BLOCK = 9, RUN_LIVE = 10;
static_assert(sizeof(WaveLds) == 1536, "per-wave LDS area is 1.5 KiB: 128 KiB filter + 8 KiB + 16 x 1.5 KiB = 160 KiB");

__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { return __umul24(a, b); }
__device__ __forceinline__ uint32_t bfe1(uint32_t v, uint32_t bit) { return __builtin_amdgcn_ubfe(v, bit, 1); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n) {
    return __builtin_amdgcn_alignbyte(hi, lo, n);
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    return (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32 |
           __builtin_amdgcn_readfirstlane((uint32_t)v);
}

/* 32-bit read at an absolute LDS byte address. The filter kernels have no static
 * __shared__ data, so their dynamic LDS segment starts at LDS address 0 (checked
 * once at kernel entry); addressing it absolutely saves the per-lookup
 * "v_add base" the compiler otherwise emits. */
typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_word(uint32_t byte_addr) { return *(lds_u32_t *)(uintptr_t)byte_addr; }

typedef __attribute__((address_space(3))) const uint64_t lds_u64_t;
__device__ __forceinline__ uint2 lds_pair(uint32_t byte_addr) { /* one ds_read_b64 at an absolute LDS address */
    const uint64_t v = *(lds_u64_t *)(uintptr_t)byte_addr;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

struct Tables {
    const uint8_t *corpus;
    const uint64_t *off;
    uint64_t nblocks, start, total;
    const uint4 *ht_a, *ht_b; /* 16-byte buckets of 4 tagged slots */
    const uint32_t *c2ref, *lists;
    const uint32_t *gate; /* pair tables: 64 Kbit "some 3-byte key has this hash" (hsgpu_gate_bit) */
    const uint32_t *key_gate; /* confirm kernel, HSGPU_F_GATE tables: the 64 Kbit key gate staged in LDS, else nullptr */
    const uint32_t *bloom;    /* confirm kernel, HSGPU_F_BLOOM tables: the full-window Bloom gate (three planes of 2^15 bits) in LDS */
    const HsgpuDevLit *lits;
    uint32_t ht_a_log2, ht_b_log2;
    uint32_t key_mask;    /* 0xdfdfdfdf when the exact-table keys are case-blind, else all ones */
    const uint32_t *hint; /* block containing byte t << HSGPU_HINT_SHIFT, t < n_hint */
    uint64_t n_hint;
    WaveLds *wl;       /* this wavefront's LDS area */
    uint4 *rec_region; /* this wavefront's private region of the staged-record buffer */
    uint32_t rec_cap;  /* its capacity in records */
    /* pair tables: the first corpus byte of the share being confirmed. A literal keyed one byte LATE is found at the lookup
     * position behind its end; at a share's first position that end lies in the share before -- whose records have been (or
     * are being) ordered without it. The share that owns the end looks for it instead (pair_edge_probe). */
    uint64_t share_start;
    /* ... and the folded pipeline's frontier: the first position behind what a wavefront has drained in order. The late-keyed
     * literals that end right in front of it were asked for before that drain (pair_edge_probe); the lookup at the frontier
     * itself leaves them alone. */
    uint64_t late_skip;
    uint32_t mq_redo; /* dense steps: a match the queue has no room for is only counted (the step is done again on fewer positions) */
    /* dense scans: the block of the last drain, [cb_start, cb_end) = block cb (wave-uniform; cb_end = 0: none yet). A dense step's
     * matches are a few dozen consecutive positions: while they stay inside that block nobody looks a block up (resolve_queued) */
    uint64_t cb, cb_start, cb_end;
};

/* the 8 bytes ending at g, little-endian, bytes before the corpus read as 0
 * (conf_key of fdr.c:360 in block mode). Aligned dword loads + funnel shift. */
__device__ __forceinline__ uint64_t window8(const uint8_t *corpus, uint64_t g) {
    if (g < 7) {
        uint64_t w = 0;
        for (uint64_t i = 0; i <= g; i++) w |= (uint64_t)corpus[i] << (8 * (7 - g + i));
        return w;
    }
    uint64_t first = g - 7;
    const uint32_t *p = (const uint32_t *)(corpus + (first & ~3ull));
    uint32_t sh = (uint32_t)(first & 3);
    uint32_t d0 = p[0], d1 = p[1];
    if (sh == 0) return (uint64_t)d1 << 32 | d0;
    uint32_t d2 = p[2];
    uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh * 8);
    uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh * 8);
    return (uint64_t)hi << 32 | lo;
}

/* index of the block containing corpus offset g: last b with off[b] <= g,
 * searched in [lo, hi] (inclusive bounds known to bracket it) */
__device__ __forceinline__ uint64_t find_block(const uint64_t *off, uint64_t lo, uint64_t hi, uint64_t g) {
    hi += 1; /* invariant: off[lo] <= g < off[hi] */
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

/* The same by a whole wavefront for a wave-uniform g: 64 probes per round (every lane one offset of an evenly spaced grid, a
 * ballot picks the segment) -- two dependent rounds for 4 096 blocks instead of twelve. Searches [0, nblocks]: index nblocks
 * stands for "at or behind the last offset", as the block hints use it. */
__device__ __forceinline__ uint64_t find_block_wave(const uint64_t *off, uint64_t nblocks, uint64_t g, uint32_t lane) {
    uint64_t lo = 0, n = nblocks + 1; /* invariant: off[lo] <= g; the answer lies in [lo, lo + n) */
    while (n > 1) {
        const uint64_t step = (n + 63) >> 6;
        const uint64_t at = (uint64_t)lane * step;
        const uint64_t v = off[min(lo + at, nblocks)];
        const uint64_t mask = __ballot(at < n && v <= g); /* monotone: lanes 0 .. k - 1 */
        const uint64_t k = (uint64_t)__popcll(mask);
        const uint64_t adv = (k ? k - 1 : 0) * step;
        lo += adv;
        n = min(step, n - adv);
    }
    return lo;
}

/* Block of corpus offset g and that block's start. Two dependent reads in the
 * common case: the hint pair, then the next three offsets after off[lo] all at
 * once; only tiles cut into more than three blocks fall back to bisection. */
__device__ __forceinline__ uint64_t block_of(const Tables &t, uint64_t g, uint64_t &block_start) {
    if (t.nblocks == 1) { /* (a single block starts at offset 0: hsgpu_hwlm_exec, one packet per call) */
        block_start = 0;
        return 0;
    }
    if (!t.hint) { /* solo scans (small batches, one launch): no hints were written; a few thousand blocks at most */
        const uint64_t b = find_block(t.off, 0, t.nblocks - 1, g);
        block_start = t.off[b];
        return b;
    }
    const uint64_t tile = g >> HSGPU_HINT_SHIFT;
    const uint64_t lo = t.hint[tile];
    const uint64_t hi = (tile + 1 < t.n_hint) ? t.hint[tile + 1] : t.nblocks - 1;
    const uint64_t o0 = t.off[lo];
    const uint64_t o1 = t.off[min(lo + 1, t.nblocks)], o2 = t.off[min(lo + 2, t.nblocks)],
                   o3 = t.off[min(lo + 3, t.nblocks)];
    if (g < o1 || lo + 1 > hi) { block_start = o0; return lo; }
    if (g < o2 || lo + 2 > hi) { block_start = o1; return lo + 1; }
    if (g < o3 || lo + 3 > hi) { block_start = o2; return lo + 2; }
    const uint64_t b = find_block(t.off, lo + 3, hi, g);
    block_start = t.off[b];
    return b;
}

/* Stage a match record in this wavefront's LDS area. No global atomics anywhere:
 * a single counter word sustains only ~90 returning atomics per microsecond,
 * which would cap the whole scan at a few tens of thousands of matches per ms.
 * Staged records are appended to the wavefront's private HBM region at
 * convergent points (flush_records); a compaction pass packs the regions. */
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) { /* set bits below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
__device__ __forceinline__ void stage_record(const Tables &t, const uint4 rec) {
    const uint32_t s = __hip_atomic_fetch_add(&t.wl->nrec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (s < OCAP) {
        t.wl->rec[s] = rec;
    } else { /* staging full inside one drain: spill to the back of the region */
        const uint32_t k = __hip_atomic_fetch_add(&t.wl->nback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (k < t.rec_cap) t.rec_region[t.rec_cap - 1 - k] = rec;
    }
}

/* A literal whose (v, msk) matched the window ending at corpus offset ge: resolve the
 * block, apply the left bound (fdr_confirm_runtime.h:77-88) and `start`
 * (hwlm.h:108-111), stage the record. Three dependent reads (id/size, hints, offsets). */
__device__ __forceinline__ void resolve_match(const Tables &t, uint64_t ge, uint32_t li) {
    const uint4 l1 = ((const uint4 *)(t.lits + li))[1]; /* {groups, id, size|flags} */
    const uint32_t id = l1.z, size = l1.w & 0xff;
    uint64_t bstart;
    const uint64_t b = block_of(t, ge, bstart);
    const uint64_t end = ge - bstart;
    if (end + 1 < size || end + 1 - size < t.start) return;
    stage_record(t, make_uint4((uint32_t)b, (uint32_t)end, id, li));
}

/* Confirm kernel: queue the match in the wavefront's LDS match queue (wl->cand,
 * unused otherwise in that kernel) so that resolving runs with full lanes later;
 * a full queue falls back to resolving in place. */
constexpr uint32_t MQ_CAP = QCAP;
__device__ __forceinline__ void push_match(const Tables &t, uint64_t ge, uint32_t li) {
    const uint32_t s = __hip_atomic_fetch_add(&t.wl->nmq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (s < MQ_CAP) {
        t.wl->cand[s] = make_uint2(li | (uint32_t)ge << 24, (uint32_t)(ge >> 8)); /* = the 64-bit sort key {position, literal}: mq_key */
    } else if (!t.mq_redo) { /* (the folded pipeline emits in order: a queue that overflowed between two of its sync points did not) */
        t.wl->pad[0] = 1u;
        resolve_match(t, ge, li);
    }
}

/* One list entry = literal index | delta << 30: does that literal end at g + delta?
 * w0 / w1 are the 8-byte windows ending at g and g + 1. */
template <bool DEFER = false>
__device__ __forceinline__ void check_lit(const Tables &t, uint32_t ent, uint64_t w0, uint64_t w1, uint64_t g) {
    const uint32_t li = ent & HSGPU_LIST_LIT_MASK;
    const uint32_t delta = (ent >> HSGPU_LIST_DELTA_SHIFT) & 1u;
    const uint64_t w = delta ? w1 : w0;
    const uint4 l0 = *(const uint4 *)(t.lits + li); /* {v, msk} */
    const uint64_t v = (uint64_t)l0.y << 32 | l0.x, msk = (uint64_t)l0.w << 32 | l0.z;
    if ((w & msk) != v) return;
    const uint64_t ge = g + delta;
    if (ge >= t.total) return; /* q + 1 can be one past the corpus */
    if (DEFER) push_match(t, ge, li);
    else resolve_match(t, ge, li);
}

/* Convergent: append the staged records to the front of the wavefront's region. */
__device__ __forceinline__ void flush_records(const Tables &t, uint32_t lane, uint32_t threshold) {
    uint32_t n = __hip_atomic_load(&t.wl->nrec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    n = __builtin_amdgcn_readfirstlane(n);
    if (n < threshold) return;
    if (n > OCAP) n = OCAP;
    const uint32_t f = __builtin_amdgcn_readfirstlane(t.wl->nfront);
    if (lane < n && f + lane < t.rec_cap) t.rec_region[f + lane] = t.wl->rec[lane];
    if (lane == 0) {
        t.wl->nfront = f + n; /* keeps counting past the capacity: the total stays exact */
        __hip_atomic_store(&t.wl->nrec, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

/* End of a wavefront's work on one region: publish how much of the region is in use; returns the fill (wave-uniform).
 * to_supers: the pipeline with a sort kernel of its own (fused scans, dense mode) -- the fills of 2^super_shift consecutive
 * regions are added up here, one atomic per region, so that every sort workgroup can place its share without a scan
 * kernel in between. The folded pipeline (hwlm_confirm_kernel) adds up per SHARE instead, after its workgroup barrier. */
/* RUNS (dense scans with run tables, args.run_tab): the region's count includes the records that exist only as run descriptors
 * (pad[RUN_VIRT]); its run table's head is written here. */
template <bool RUNS = false>
__device__ __forceinline__ uint32_t publish_records(const Tables &t, const HsgpuScanArgs &args, uint32_t lane,
                                                    uint32_t region, bool to_supers = true) {
    flush_records(t, lane, 1);
    uint32_t fill32 = 0;
    if (lane == 0) {
        const uint32_t front = t.wl->nfront,
                       back = __hip_atomic_load(&t.wl->nback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t virt = RUNS ? t.wl->pad[RUN_VIRT] : 0u;
        args.rec_counts[2 * region] = front + virt;
        args.rec_counts[2 * region + 1] = back;
        if (RUNS) args.run_tab[(uint64_t)region * HSGPU_RUN_STRIDE] = make_uint4(t.wl->pad[RUN_N], 0, 0, 0);
        const unsigned long long staged = (unsigned long long)front + back, fill = staged + virt;
        if (to_supers && fill) atomicAdd(&args.rec_super[HSGPU_SUPER(region >> args.super_shift)], fill);
        if (staged > args.rec_cap) atomicAdd(&args.rec_super[HSGPU_SUPER_FLAGS], 1ull); /* the region lost records */
        fill32 = (uint32_t)min(fill, 0x7fffffffull);
    }
    return __builtin_amdgcn_readfirstlane(fill32);
}

template <bool DEFER = false>
__device__ __forceinline__ void walk_ref(const Tables &t, uint32_t ref, uint64_t w0, uint64_t w1, uint64_t g) {
    if (ref & HSGPU_REF_DIRECT) {
        check_lit<DEFER>(t, ref, w0, w1, g); /* check_lit masks the literal index and the delta bit itself */
        return;
    }
    uint32_t i = (ref & HSGPU_LIST_LIT_MASK) - 1, e;
    do {
        e = t.lists[i++];
        check_lit<DEFER>(t, e, w0, w1, g);
    } while (!(e & HSGPU_LIST_END));
}

/* which of a bucket's 4 slots carry this tag (bit i = slot i) */
__device__ __forceinline__ uint32_t bucket_match(const uint4 s, uint32_t tag) {
    auto hit = [&](uint32_t slot) { return slot && ((slot >> HSGPU_SLOT_TAG_SHIFT) & HSGPU_SLOT_TAG_MASK) == tag; };
    return (hit(s.x) ? 1u : 0u) | (hit(s.y) ? 2u : 0u) | (hit(s.z) ? 4u : 0u) | (hit(s.w) ? 8u : 0u);
}

/* one 16-byte bucket (4 tagged slots) per probe; continue to the next bucket only
 * when this one is full (the host inserts with the same rule). A tag match is a hint:
 * the literal compare in check_lit is exact. */
template <bool DEFER = false>
__device__ __forceinline__ void probe(const Tables &t, const uint4 *ht, uint32_t log2, uint32_t key, uint64_t w0,
                                      uint64_t w1, uint64_t g) {
    const uint32_t mask = (1u << log2) - 1, tag = hsgpu_ht_tag(key, log2);
    uint32_t b = hsgpu_ht_bucket(key, log2);
    for (;;) {
        const uint4 s = ht[b];
        const uint32_t m = bucket_match(s, tag);
        if (m & 1) walk_ref<DEFER>(t, s.x, w0, w1, g);
        if (m & 2) walk_ref<DEFER>(t, s.y, w0, w1, g);
        if (m & 4) walk_ref<DEFER>(t, s.z, w0, w1, g);
        if (m & 8) walk_ref<DEFER>(t, s.w, w0, w1, g);
        if (!s.w) return; /* slots fill in order: last slot empty => bucket not full */
        b = (b + 1) & mask;
    }
}

/* confirm one lookup position g given the windows ending at g (w0) and g + 1 (w1) */
template <bool HAS_A, bool HAS_B, bool HAS_C>
__device__ __forceinline__ void confirm_pos(const Tables &t, bool hit_a, bool hit_o, uint64_t w0, uint64_t w1,
                                            uint64_t g) {
    const uint32_t w4 = (uint32_t)(w0 >> 32) & t.key_mask;
    if (HAS_A && hit_a) probe(t, t.ht_a, t.ht_a_log2, w4, w0, w1, g);
    if ((HAS_B || HAS_C) && hit_o) {
        if (HAS_B) probe(t, t.ht_b, t.ht_b_log2, w4 >> 8, w0, w1, g);
        if (HAS_C) {
            const uint32_t ref = t.c2ref[w4 >> 16];
            if (ref) walk_ref(t, ref, w0, w1, g);
        }
    }
}

/* pair tables: the filter does not say which kind of key passed. Probe the 4-byte table with the window
 * ending at g (its entries end at g or g + 1) and, when the gate bitmap knows the hash, the 3-byte table with
 * the 3 bytes ending at g (literals ending at g / g + 1) and with the 3 bytes ending at g - 1 (HSGPU_KEY_M:
 * literals keyed one byte late, which end at g - 1; wm = the window ending there). */
__device__ __forceinline__ bool gate_hit(const Tables &t, uint32_t key25) {
    const uint32_t h = hsgpu_gate_bit(key25);
    return (t.gate[h >> 5] >> (h & 31)) & 1u;
}
template <bool HAS_A, bool HAS_B, bool DEFER>
__device__ __forceinline__ void confirm_pos_pair(const Tables &t, uint64_t wm, uint64_t w0, uint64_t w1, uint64_t g) {
    const uint32_t w4 = (uint32_t)(w0 >> 32) & t.key_mask;
    if (HAS_A) probe<DEFER>(t, t.ht_a, t.ht_a_log2, w4, w0, w1, g);
    if (HAS_B) {
        const uint32_t kb = w4 >> 8, km = (((uint32_t)(wm >> 32) & t.key_mask) >> 8) | HSGPU_KEY_M;
        if (gate_hit(t, kb)) probe<DEFER>(t, t.ht_b, t.ht_b_log2, kb, w0, w1, g);
        if (g && g != t.share_start && g != t.late_skip && gate_hit(t, km)) probe<DEFER>(t, t.ht_b, t.ht_b_log2, km, wm, 0, g - 1);
    }
}

/* One lane, pair tables: the late-keyed literals that end at g - 1, the last byte of a share -- asked for by that share, since
 * the lookup position g that would find them is the NEXT share's first (see Tables::share_start). The exact tables are the
 * authority: no filter bit is needed. */
template <bool DEFER>
__device__ __forceinline__ void pair_edge_probe(const Tables &t, uint64_t g) {
    const uint64_t wm = window8(t.corpus, g - 1);
    const uint32_t km = (((uint32_t)(wm >> 32) & t.key_mask) >> 8) | HSGPU_KEY_M;
    if (gate_hit(t, km)) probe<DEFER>(t, t.ht_b, t.ht_b_log2, km, wm, 0, g - 1);
}

/* fused kernel: {chunk, masks} entry, windows re-read from the corpus (L2 hits) */
template <bool HAS_A, bool HAS_B, bool HAS_C, bool S2, bool PAIR = false>
__device__ __forceinline__ void drain_entry(const Tables &t, uint2 e) {
    const uint32_t m = e.y;
    uint32_t any = (m | m >> 16) & 0xffffu;
    while (any) {
        const uint32_t j = __builtin_ctz(any);
        any &= any - 1;
        const uint64_t g = (uint64_t)e.x * CHUNK + j;
        const uint64_t w0 = window8(t.corpus, g);
        const uint64_t w1 = (S2 && g + 1 < t.total) ? window8(t.corpus, g + 1) : 0;
        if (PAIR) confirm_pos_pair<HAS_A, HAS_B, false>(t, g ? window8(t.corpus, g - 1) : 0, w0, w1, g);
        else confirm_pos<HAS_A, HAS_B, HAS_C>(t, m >> j & 1, m >> (16 + j) & 1, w0, w1, g);
    }
}

/* two-phase: a 32-byte candidate entry carries the lane's chunk and the 8 bytes
 * in front of it, so confirming never touches the corpus again */
__device__ __forceinline__ uint64_t funnel64(uint64_t lo, uint64_t hi, uint32_t nbytes) { /* nbytes 1..8 */
    return nbytes == 8 ? hi : (lo >> (8 * nbytes)) | (hi << (64 - 8 * nbytes));
}

/* ---- the confirm kernel's candidate path ------------------------------------
 * The kernel is bound by how many (lane x load) pairs it issues and by the depth
 * of dependent reads inside divergent code, so the work is kept DENSE:
 *   step:   two entries per lane; ONE candidate bit of each: bucket (16 B) ->
 *           {v, msk} of the literal a uniquely tag-matching slot names (16 B).
 *           A (v, msk) hit is pushed onto the wavefront's LDS match queue; an
 *           entry with further candidate bits goes onto its LDS rest queue.
 *   drain:  64 queued matches at a time with full lanes: id/size, block hints,
 *           offsets, bounds, records stored straight into the wavefront's region.
 * (Measured on fdr10k, 9.4M candidate entries, 0.8M matches: resolving each match
 * where it is found -- three dependent reads under divergence, ~10 matching lanes
 * per 128 entries -- cost 0.16 ms of the kernel's 0.33 ms; fetching block data
 * speculatively for every candidate was slower still.)
 * Several tag matches in one bucket, literal lists and displaced keys (full buckets)
 * go through the general code, which queues its matches the same way. */
#ifndef HSGPU_CONFIRM_BLOOM
#define HSGPU_CONFIRM_BLOOM 1 /* tuning builds: 0 compiles the opt-in Bloom gate out of the fast step (what it costs the default in registers) */
#endif
/* Candidate entries per lane and confirm step, and the wavefronts per SIMD the kernel's registers are capped for -- per
 * instantiation (confirm_shape). The fast step (stride-1 tables without 2-byte keys: the headline's) takes ONE entry per lane at
 * EIGHT wavefronts per SIMD: ~60 registers, 18 KiB of LDS per workgroup, eight workgroups per CU. Measured on one MI355X, the
 * headline workload, alternating libraries (profiles/r05_confirm_shape_ab.txt): two entries at six wavefronts (rounds 3-5) 0.1625-
 * 0.1655 ms, two at five (89 registers, nothing spilled) the same, three at five 0.223, four at four 0.213-0.216 (both spill),
 * one at eight 0.153-0.156. The stage is not short of entries in flight per lane; its SIMDs are ~50 % busy issuing vector
 * instructions (48.9 M per GiB x 4 cycles over 1 024 SIMDs = 0.08 of its 0.165 ms) and wait on dependent reads the rest of the
 * time: more wavefronts cover more of that, more entries per wavefront only lengthen a step. The general step keeps two entries
 * at six wavefronts (its 85-92 registers do not fit eight). */
#ifndef HSGPU_CONFIRM_U
#define HSGPU_CONFIRM_U 1 /* the fast step's entries per lane (tuning builds: 2 with HSGPU_CONFIRM_WAVES=6 is rounds 3-5) */
#endif
#ifndef HSGPU_CONFIRM_WAVES
#define HSGPU_CONFIRM_WAVES 8 /* ... and its wavefronts per SIMD */
#endif
#ifndef HSGPU_CONFIRM_FAST
#define HSGPU_CONFIRM_FAST 1 /* tuning builds: 0 = the general step for every table */
#endif
template <bool HAS_A, bool HAS_C, bool S2, bool PAIR, bool DENSE> struct confirm_shape {
    static constexpr bool FAST = HSGPU_CONFIRM_FAST && HAS_A && !HAS_C && !S2 && !PAIR;
    static constexpr int U = (FAST && !DENSE) ? HSGPU_CONFIRM_U : 2;
#ifndef HSGPU_CONFIRM_DENSE_WAVES
#define HSGPU_CONFIRM_DENSE_WAVES 5 /* the dense kernels' position-parallel path on top of the general step: 96 registers (at 80 they spill 33-73) */
#endif
    static constexpr int WAVES = (FAST && !DENSE) ? HSGPU_CONFIRM_WAVES : DENSE ? HSGPU_CONFIRM_DENSE_WAVES : 6;
    static constexpr uint32_t RQ_CAP = 128u * U; /* rest queue: {entry index, masks still to do}; a step may queue 64 * U more */
};

__device__ __forceinline__ uint32_t pick_slot(const uint4 s, uint32_t m) {
    return (m & 1) ? s.x : (m & 2) ? s.y : (m & 4) ? s.z : s.w;
}

/* (v, msk) of a literal already in registers */
__device__ __forceinline__ void check_lit_loaded(const Tables &t, uint32_t ent, const uint4 l0, uint64_t w0, uint64_t w1,
                                                 uint64_t g) {
    const uint32_t delta = (ent >> HSGPU_LIST_DELTA_SHIFT) & 1u;
    const uint64_t w = delta ? w1 : w0;
    const uint64_t v = (uint64_t)l0.y << 32 | l0.x, msk = (uint64_t)l0.w << 32 | l0.z;
    if ((w & msk) != v) return;
    const uint64_t ge = g + delta;
    if (ge >= t.total) return;
    push_match(t, ge, ent & HSGPU_LIST_LIT_MASK);
}

/* The Bloom gate (table.h, HSGPU_F_BLOOM) for one masked window {hi, lo} of group `salt`: ~0 when all three planes have the
 * key's bit, else 0. Three LDS reads; v_bfe_i32 takes the bit index from the low five bits of its operand. */
__device__ __forceinline__ uint32_t bloom_pass(const uint32_t *bl, uint32_t hi, uint32_t lo, uint32_t salt) {
    uint32_t idx[3];
    hsgpu_bloom_idx(hi, lo, salt, idx);
    constexpr uint32_t PW = 1u << (HSGPU_BLOOM_PLANE_LOG2 - 5);
    const uint32_t w0 = bl[idx[0] >> 5], w1 = bl[PW + (idx[1] >> 5)], w2 = bl[2 * PW + (idx[2] >> 5)];
    uint32_t r0, r1, r2;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r0) : "v"(w0), "v"(idx[0]));
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r1) : "v"(w1), "v"(idx[1]));
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r2) : "v"(w2), "v"(idx[2]));
    return r0 & r1 & r2;
}
/* -> which exact tables a position with this (case-blinded where the table is) window has to be probed in */
template <bool HAS_B>
__device__ __forceinline__ void bloom_gate(const Tables &t, uint32_t whi, uint32_t wlo, uint32_t &pass_a, uint32_t &pass_b) {
    const uint32_t hb = whi & t.key_mask, lb = wlo & t.key_mask & 0xff000000u;
    pass_a = bloom_pass(t.bloom, hb, lb, 5) | bloom_pass(t.bloom, hb, 0, 4);
    pass_b = HAS_B ? bloom_pass(t.bloom, hb & 0xffffff00u, 0, 3) : 0u;
}

/* Convergent. idx[u] / pend[u]: entry index in `region` and its candidate masks still
 * to do (0 = idle lane, reads entry 0). `fresh`: take the masks from the entry itself. */
template <bool HAS_A, bool HAS_B, bool HAS_C, bool S2, bool PAIR>
__device__ __forceinline__ void confirm_step(const Tables &t, const uint4 *region, uint2 *rq, const uint32_t (&idx)[2],
                                             uint32_t (&pend)[2], const bool (&valid)[2], bool fresh) {
    uint4 e0[2], e1[2];
#pragma unroll
    for (int u = 0; u < 2; u++) e0[u] = region[2 * idx[u]], e1[u] = region[2 * idx[u] + 1];
    uint64_t g[2], w0[2], w1[2], wm[2];
    uint32_t w4[2];
    bool do_a[2], do_b[2], do_c[2], cnd[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint32_t m = fresh ? (valid[u] ? e0[u].y : 0) : pend[u];
        const uint32_t any = (m | m >> 16) & 0xffffu;
        const uint32_t j = __builtin_ctz(any | 0x10000u) & 15u;
        const uint64_t A = (uint64_t)e0[u].w << 32 | e0[u].z, B = (uint64_t)e1[u].y << 32 | e1[u].x,
                       C = (uint64_t)e1[u].w << 32 | e1[u].z;
        g[u] = (uint64_t)e0[u].x * CHUNK + j;
        /* window ending at c[j]: j <= 7 -> (A,B) shifted by j+1 bytes, else (B,C) by j-7 */
        w0[u] = j < 8 ? funnel64(A, B, j + 1) : funnel64(B, C, j - 7);
        w1[u] = 0;
        if (S2) w1[u] = (j + 1) < 8 ? funnel64(A, B, j + 2) : funnel64(B, C, j - 6); /* j even: j + 1 <= 15 */
        wm[u] = 0;
        if (PAIR) wm[u] = j == 0 ? A : j <= 8 ? funnel64(A, B, j) : funnel64(B, C, j - 8); /* the window ending at c[j - 1] */
        w4[u] = (uint32_t)(w0[u] >> 32) & t.key_mask;
        cnd[u] = any && (m >> j & 1);
        do_a[u] = HAS_A && any && (m >> j & 1);
        do_b[u] = !PAIR && HAS_B && any && (m >> (16 + j) & 1); /* pair tables: through the gate bitmap, below */
        do_c[u] = HAS_C && any && (m >> (16 + j) & 1);
        if (!PAIR && !S2 && t.bloom) { /* windows that no literal's full key matches need no probe (HSGPU_F_BLOOM) */
            uint32_t pa, pb;
            bloom_gate<HAS_B>(t, (uint32_t)(w0[u] >> 32), (uint32_t)w0[u], pa, pb);
            do_a[u] = do_a[u] && pa;
            do_b[u] = do_b[u] && pb;
        } else if (!PAIR && t.key_gate) { /* keys that no exact table holds need no probe: 14 % of them pass on a 10 000-literal set */
            const uint32_t ga = hsgpu_key_gate_bit(w4[u]), gb = hsgpu_key_gate_bit((w4[u] >> 8) | HSGPU_GATE_B_SALT);
            if (HAS_A) do_a[u] = do_a[u] && ((t.key_gate[ga >> 5] >> (ga & 31)) & 1u);
            if (HAS_B) do_b[u] = do_b[u] && ((t.key_gate[gb >> 5] >> (gb & 31)) & 1u);
        }
        const uint32_t rest = any & (any - 1);
        pend[u] = m & (rest | rest << 16);
    }
    /* entries with candidate bits left: onto the rest queue */
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint64_t mask = __ballot(pend[u] != 0);
        if (mask) {
            const uint32_t base = __builtin_amdgcn_readfirstlane(t.wl->nrq);
            if (pend[u]) rq[base + lane_rank(mask)] = make_uint2(idx[u], pend[u]);
            if (lane_rank(~0ull) == 0) t.wl->nrq = base + (uint32_t)__popcll(mask);
        }
    }
    /* level 1: one 16-byte bucket per key class */
    uint4 sa[2], sb[2];
    uint32_t ref_c[2] = {0, 0};
    /* pair tables: the two 3-byte keys of the position and their gate words (read beside the 4-byte bucket) */
    uint32_t kb[2] = {0, 0}, km[2] = {0, 0}, gb[2] = {0, 0}, gm[2] = {0, 0};
    if (PAIR && HAS_B) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            kb[u] = w4[u] >> 8;
            km[u] = (((uint32_t)(wm[u] >> 32) & t.key_mask) >> 8) | HSGPU_KEY_M;
            gb[u] = t.gate[cnd[u] ? hsgpu_gate_bit(kb[u]) >> 5 : 0];
            gm[u] = t.gate[cnd[u] ? hsgpu_gate_bit(km[u]) >> 5 : 0];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (HAS_A) sa[u] = t.ht_a[do_a[u] ? hsgpu_ht_bucket(w4[u], t.ht_a_log2) : 0];
        if (HAS_B && !PAIR) sb[u] = t.ht_b[do_b[u] ? hsgpu_ht_bucket(w4[u] >> 8, t.ht_b_log2) : 0];
        if (HAS_C) ref_c[u] = t.c2ref[do_c[u] ? (w4[u] >> 16) : 0];
    }
    uint32_t ma[2] = {0, 0}, mb[2] = {0, 0}, ref_a[2] = {0, 0}, ref_b[2] = {0, 0};
    bool fast_a[2], fast_b[2], fast_c[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (HAS_A && do_a[u]) ma[u] = bucket_match(sa[u], hsgpu_ht_tag(w4[u], t.ht_a_log2));
        if (HAS_B && !PAIR && do_b[u]) mb[u] = bucket_match(sb[u], hsgpu_ht_tag(w4[u] >> 8, t.ht_b_log2));
        if (!do_c[u]) ref_c[u] = 0;
        ref_a[u] = ma[u] ? pick_slot(sa[u], ma[u]) : 0;
        ref_b[u] = mb[u] ? pick_slot(sb[u], mb[u]) : 0;
        /* fast = exactly one tag match, naming its literal directly, bucket not full */
        fast_a[u] = __popc(ma[u]) == 1 && (ref_a[u] & HSGPU_REF_DIRECT) && !sa[u].w;
        fast_b[u] = !PAIR && __popc(mb[u]) == 1 && (ref_b[u] & HSGPU_REF_DIRECT) && !sb[u].w;
        fast_c[u] = (ref_c[u] & HSGPU_REF_DIRECT) != 0;
    }
    /* level 2: {v, msk} of the literal the slot names */
    uint4 la[2], lb[2], lc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (HAS_A) la[u] = *(const uint4 *)(t.lits + (fast_a[u] ? (ref_a[u] & HSGPU_LIST_LIT_MASK) : 0));
        if (HAS_B && !PAIR) lb[u] = *(const uint4 *)(t.lits + (fast_b[u] ? (ref_b[u] & HSGPU_LIST_LIT_MASK) : 0));
        if (HAS_C) lc[u] = *(const uint4 *)(t.lits + (fast_c[u] ? (ref_c[u] & HSGPU_LIST_LIT_MASK) : 0));
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (HAS_A && fast_a[u]) check_lit_loaded(t, ref_a[u], la[u], w0[u], w1[u], g[u]);
        if (HAS_B && !PAIR && fast_b[u]) check_lit_loaded(t, ref_b[u], lb[u], w0[u], w1[u], g[u]);
        if (HAS_C && fast_c[u]) check_lit_loaded(t, ref_c[u], lc[u], w0[u], w1[u], g[u]);
        /* everything else: the general path */
        if (HAS_A && do_a[u] && !fast_a[u] && (ma[u] || sa[u].w)) probe<true>(t, t.ht_a, t.ht_a_log2, w4[u], w0[u], w1[u], g[u]);
        if (HAS_B && !PAIR && do_b[u] && !fast_b[u] && (mb[u] || sb[u].w)) probe<true>(t, t.ht_b, t.ht_b_log2, w4[u] >> 8, w0[u], w1[u], g[u]);
        if (HAS_C && ref_c[u] && !fast_c[u]) walk_ref<true>(t, ref_c[u], w0[u], w1[u], g[u]);
        if (PAIR && HAS_B && cnd[u]) { /* rare: a few percent of the candidates have a 3-byte key with their hash */
            if ((gb[u] >> (hsgpu_gate_bit(kb[u]) & 31)) & 1u) probe<true>(t, t.ht_b, t.ht_b_log2, kb[u], w0[u], w1[u], g[u]);
            if (g[u] && g[u] != t.share_start && g[u] != t.late_skip && ((gm[u] >> (hsgpu_gate_bit(km[u]) & 31)) & 1u))
                probe<true>(t, t.ht_b, t.ht_b_log2, km[u], wm[u], 0, g[u] - 1);
        }
    }
}

/* ---- the same step for stride-1 tables without 2-byte keys, in a third of the vector instructions ---------------
 * The general step above is compiled from booleans and 64-bit values: 68 v_cndmask_b32 (most of them a boolean turned
 * into 0 / 1 and compared again), 64-bit funnel shifts for the window, 64-bit address arithmetic for every load --
 * ~470 vector instructions per step of 128 entries. Here every condition is a 0 / ~0 MASK made with plain
 * arithmetic and applied with v_and / v_or; nothing on the common path selects: ~330 instructions, measured 8 %
 * off the stage (profiles/r03_confirm_fast_ab.txt). No more than that, because the stage is bound by the vector L1's
 * outstanding misses: 10.45 M read requests per GiB at 473 cycles against 64 per CU (profiles/r03_confirm_mem.json).
 *   entry     8 bytes {chunk, masks}; then the WINDOW itself is read: the 8 bytes ending at the candidate position
 *             are bytes [9 + j, 16 + j] of the 32-byte entry -- three aligned dwords and two v_alignbit, instead of
 *             six dwords in registers, two 64-bit funnel shifts and six selects
 *   key gate  both keys' bits from LDS, unconditionally (an idle lane reads some word of the gate)
 *   buckets   16 bytes per table at (bucket & mask): a lane with nothing to probe reads bucket 0
 *   slots     hit_i = ~0 when slot i is in use and carries the tag; the literal a lone direct hit names is compared
 *             in place ({v, msk}: 16 bytes); lists, several hits, full buckets: the general code (probe<true>)
 * All loads go through buffer descriptors (32-bit lane offsets against scalar bases; out-of-range offsets read 0). */
struct FastRs {
    __amdgpu_buffer_rsrc_t region, ht_a, ht_b, lits;
};
/* (inline asm: written as C the compiler recognises each of these as a sign-extended compare and goes back to
 * v_cmp + v_cndmask and a boolean per condition) */
__device__ __forceinline__ uint32_t m_zero31(uint32_t x) { /* x < 2^31: ~0 when x == 0 */
    uint32_t r;
    asm("v_add_u32_e32 %0, -1, %1\n\tv_ashrrev_i32_e32 %0, 31, %0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t m_nonzero(uint32_t x) { /* any x: ~0 when x != 0 */
    uint32_t r;
    asm("v_min_u32_e32 %0, 1, %1\n\tv_sub_u32_e32 %0, 0, %0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t m_bit(uint32_t v, uint32_t bit) { /* ~0 when bit (bit & 31) of v is set: a one-bit signed field */
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(v), "v"(bit));
    return r;
}
__device__ __forceinline__ uint32_t m_less(uint32_t a, uint32_t b) { /* ~0 when a < b, both below 2^31 */
    uint32_t r;
    asm("v_sub_u32_e32 %0, %1, %2\n\tv_ashrrev_i32_e32 %0, 31, %0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t slot_hit(uint32_t slot, uint32_t tag) {
    return m_zero31(((slot >> HSGPU_SLOT_TAG_SHIFT) & HSGPU_SLOT_TAG_MASK) ^ tag) & m_nonzero(slot);
}
/* one bucket against one key: fast = a lone, direct, delta-0 hit in a bucket that is not full (ref names the literal);
 * slow = anything else that needs the general code */
__device__ __forceinline__ void bucket_masks(const uint32_t (&s)[4], uint32_t tag, uint32_t doit, uint32_t &ref, uint32_t &fast,
                                             uint32_t &slow) {
    const uint32_t h0 = slot_hit(s[0], tag), h1 = slot_hit(s[1], tag), h2 = slot_hit(s[2], tag), h3 = slot_hit(s[3], tag);
    const uint32_t n = 0u - (h0 + h1 + h2 + h3); /* 0..4 hits */
    ref = (s[0] & h0) | (s[1] & h1) | (s[2] & h2) | (s[3] & h3);
    const uint32_t full = m_nonzero(s[3]);
    fast = doit & m_zero31(n ^ 1u) & m_zero31((ref >> HSGPU_LIST_DELTA_SHIFT) ^ 2u) & ~full;
    slow = doit & ~fast & (m_nonzero(n) | full);
}
template <bool HAS_B, bool FRESH, int U = 2>
__device__ __forceinline__ void confirm_step_fast(const Tables &t, const FastRs &rs, uint2 *rq, const uint32_t (&idx)[U],
                                                  uint32_t (&pend)[U], const uint32_t (&vm)[U]) {
    /* (Requesting the next fresh step's masks one step ahead, behind this step's last loads, was measured: 82 registers
     * instead of 79 cost a wavefront per SIMD and the stage went 0.143 -> 0.158 ms, 0.151 with the registers capped and two
     * spills: profiles/r03_confirm_fast_ab.txt. The stage does not wait on that hop.) */
    uint32_t chunk[U], m[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (FRESH) {
            const auto e = __builtin_amdgcn_raw_buffer_load_b64(rs.region, idx[u] << 5, 0, 0);
            chunk[u] = e[0];
            m[u] = e[1] & vm[u];
        } else { /* an entry off the rest queue: its masks are known, the chunk index is read beside the window */
            chunk[u] = __builtin_amdgcn_raw_buffer_load_b32(rs.region, idx[u] << 5, 0, 0);
            m[u] = pend[u];
        }
    }
    uint32_t j[U], wlo[U], whi[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t any = (m[u] | m[u] >> 16) & 0xffffu;
        j[u] = (uint32_t)__builtin_ctz(any | 0x10000u) & 15u;
        const uint32_t o = 9u + j[u]; /* the window ending at c[j]: entry bytes [o, o + 8) */
        const auto d = __builtin_amdgcn_raw_buffer_load_b96(rs.region, (idx[u] << 5) + (o & ~3u), 0, 0);
        const uint32_t sh = (o & 3u) << 3;
        wlo[u] = __builtin_amdgcn_alignbit(d[1], d[0], sh); /* (o = 24: sh = 0, the dword past the entry is not used) */
        whi[u] = __builtin_amdgcn_alignbit(d[2], d[1], sh);
        const uint32_t rest = any & (any - 1u);
        pend[u] = m[u] & (rest | rest << 16);
    }
    /* entries with candidate bits left: onto the rest queue */
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t mask = __ballot(pend[u] != 0);
        if (mask) {
            const uint32_t base = __builtin_amdgcn_readfirstlane(t.wl->nrq);
            if (pend[u]) rq[base + lane_rank(mask)] = make_uint2(idx[u], pend[u]);
            if (lane_rank(~0ull) == 0) t.wl->nrq = base + (uint32_t)__popcll(mask);
        }
    }
    uint32_t w4[U], pa[U], pb[U], do_a[U], do_b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        w4[u] = whi[u] & t.key_mask;
        pa[u] = w4[u] * HSGPU_HT_MUL;
        pb[u] = (w4[u] >> 8) * HSGPU_HT_MUL;
        do_a[u] = m_bit(m[u], j[u]); /* (no candidate bits at all: j = 0 and bit 0 is clear) */
        do_b[u] = HAS_B ? m_bit(m[u], 16u + j[u]) : 0u;
        if (HSGPU_CONFIRM_BLOOM && t.bloom) { /* windows that no literal's full key matches need no probe (HSGPU_F_BLOOM) */
            uint32_t ga, gb;
            bloom_gate<HAS_B>(t, whi[u], wlo[u], ga, gb);
            do_a[u] &= ga;
            do_b[u] &= gb;
        } else if (t.key_gate) { /* keys that no exact table holds need no probe */
            const uint32_t ga = pa[u] >> 16, gb = (pb[u] + HSGPU_GATE_B_SALT * HSGPU_HT_MUL) >> 16; /* the salt sits above the 24 key bits */
            do_a[u] &= m_bit(t.key_gate[ga >> 5], ga);
            if (HAS_B) do_b[u] &= m_bit(t.key_gate[gb >> 5], gb);
        }
    }
    uint32_t sa[U][4], sb[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const auto a = __builtin_amdgcn_raw_buffer_load_b128(rs.ht_a, ((pa[u] >> (32u - t.ht_a_log2)) & do_a[u]) << 4, 0, 0);
        sa[u][0] = a[0], sa[u][1] = a[1], sa[u][2] = a[2], sa[u][3] = a[3];
        if (HAS_B) {
            const auto b = __builtin_amdgcn_raw_buffer_load_b128(rs.ht_b, ((pb[u] >> (32u - t.ht_b_log2)) & do_b[u]) << 4, 0, 0);
            sb[u][0] = b[0], sb[u][1] = b[1], sb[u][2] = b[2], sb[u][3] = b[3];
        }
    }
    /* (all bucket reads issued before the first is looked at: with one entry per lane and 64 registers the scheduler otherwise
     * waited for table A's bucket and worked on it before it asked for table B's -- two round trips for one) */
    __builtin_amdgcn_sched_barrier(0);
    uint32_t ref_a[U], fast_a[U], slow_a[U], ref_b[U] = {}, fast_b[U] = {}, slow_b[U] = {};
#pragma unroll
    for (int u = 0; u < U; u++) {
        bucket_masks(sa[u], (pa[u] >> (26u - t.ht_a_log2)) & HSGPU_SLOT_TAG_MASK, do_a[u], ref_a[u], fast_a[u], slow_a[u]);
        if (HAS_B) bucket_masks(sb[u], (pb[u] >> (26u - t.ht_b_log2)) & HSGPU_SLOT_TAG_MASK, do_b[u], ref_b[u], fast_b[u], slow_b[u]);
    }
    /* {v, msk} of the literal the slot names (literal 0 for lanes without one). (One literal load per entry, a position
     * whose two keys both name a literal sending its 3-byte key through the general path, brought the kernel from 79 to 69
     * registers -- seven wavefronts per SIMD -- and was slower, 0.150 vs 0.143 ms: with BFOLD tables both keys of every
     * candidate are tried, and the general path then runs in nearly every step.) */
    uint32_t la[U][4], lb[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const auto a = __builtin_amdgcn_raw_buffer_load_b128(rs.lits, (ref_a[u] & HSGPU_LIST_LIT_MASK & fast_a[u]) << 5, 0, 0);
        la[u][0] = a[0], la[u][1] = a[1], la[u][2] = a[2], la[u][3] = a[3];
        if (HAS_B) {
            const auto b = __builtin_amdgcn_raw_buffer_load_b128(rs.lits, (ref_b[u] & HSGPU_LIST_LIT_MASK & fast_b[u]) << 5, 0, 0);
            lb[u][0] = b[0], lb[u][1] = b[1], lb[u][2] = b[2], lb[u][3] = b[3];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        /* (window & msk) == v, 32 bits at a time; a stride-1 table has no delta-1 entries, every g is inside the corpus */
        const uint32_t xa = ((wlo[u] & la[u][2]) ^ la[u][0]) | ((whi[u] & la[u][3]) ^ la[u][1]);
        if (fast_a[u] && xa == 0) push_match(t, (uint64_t)chunk[u] * CHUNK + j[u], ref_a[u] & HSGPU_LIST_LIT_MASK);
        if (HAS_B) {
            const uint32_t xb = ((wlo[u] & lb[u][2]) ^ lb[u][0]) | ((whi[u] & lb[u][3]) ^ lb[u][1]);
            if (fast_b[u] && xb == 0) push_match(t, (uint64_t)chunk[u] * CHUNK + j[u], ref_b[u] & HSGPU_LIST_LIT_MASK);
        }
    }
    /* everything else: the general path (behind the compares: the literals' registers are free again) */
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (slow_a[u] | slow_b[u]) {
            const uint64_t g = (uint64_t)chunk[u] * CHUNK + j[u], w0 = (uint64_t)whi[u] << 32 | wlo[u];
            if (slow_a[u]) probe<true>(t, t.ht_a, t.ht_a_log2, w4[u], w0, 0, g);
            if (HAS_B && slow_b[u]) probe<true>(t, t.ht_b, t.ht_b_log2, w4[u] >> 8, w0, 0, g);
        }
    }
}

/* Convergent: one queued match per lane (`valid`) resolved -- id / size, block through the hints, bounds -- and the records
 * appended to the front of the wavefront's region in lane order. A queue entry is the 64-bit key {position << 24 | literal}. */
template <bool CACHE = false, class TABLES>
__device__ __forceinline__ void resolve_queued(TABLES &t, uint32_t lane, bool valid, uint2 it) {
    const uint32_t li = it.x & HSGPU_LIST_LIT_MASK;
    const uint64_t ge = (uint64_t)it.y << 8 | it.x >> 24;
    const uint4 l1 = ((const uint4 *)(t.lits + li))[1];
    uint64_t bstart, b;
    bool cached = false;
    if constexpr (CACHE) {
        /* every queued match inside the block of the last drain: no lookup (the reference's flood case is one block for millions
         * of matches; block_of is five dependent reads) */
        if (__ballot(valid && !(ge >= t.cb_start && ge < t.cb_end)) == 0) b = t.cb, bstart = t.cb_start, cached = true;
    }
    if (!cached) {
        b = block_of(t, ge, bstart);
        if constexpr (CACHE) { /* the block of the last valid lane (the keys are sorted: the largest position) is the next drain's guess */
            const uint64_t vm = __ballot(valid);
            if (vm) {
                const int last = 63 - __builtin_clzll(vm);
                const uint64_t bend = t.off[min(b + 1, t.nblocks)];
                t.cb = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), last) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, last);
                t.cb_start = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bstart >> 32), last) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bstart, last);
                t.cb_end = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bend >> 32), last) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bend, last);
            }
        }
    }
    const uint32_t id = l1.z, size = l1.w & 0xff;
    const uint64_t end = ge - bstart;
    const bool ok = valid && !(end + 1 < size || end + 1 - size < t.start);
    const uint64_t mask = __ballot(ok);
    const uint32_t f = __builtin_amdgcn_readfirstlane(t.wl->nfront);
    const uint32_t at = f + lane_rank(mask);
    if (ok && at < t.rec_cap) t.rec_region[at] = make_uint4((uint32_t)b, (uint32_t)end, id, li);
    if (lane == 0) t.wl->nfront = f + (uint32_t)__popcll(mask); /* keeps counting past the capacity: the total stays exact */
}

/* Convergent: resolve queued matches, 64 at a time with full lanes, while more than
 * `keep` are queued; records go straight to the front of the wavefront's region. */
__device__ __forceinline__ void drain_matches(const Tables &t, uint32_t lane, uint32_t keep) {
    uint32_t n = __hip_atomic_load(&t.wl->nmq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    n = min(__builtin_amdgcn_readfirstlane(n), MQ_CAP); /* pushes past the capacity resolved in place */
    if (n <= keep) {
        if (lane == 0) __hip_atomic_store(&t.wl->nmq, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        return;
    }
    while (n > keep) {
        const uint32_t k = min(n, 64u);
        n -= k;
        const bool valid = lane < k;
        const uint2 it = t.wl->cand[valid ? n + lane : 0];
        resolve_queued(t, lane, valid, it);
    }
    if (lane == 0) __hip_atomic_store(&t.wl->nmq, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

struct Chunk {
    uint4 d;  /* this lane's 16 bytes */
    uint2 h;  /* the 8 bytes in front of them (h.y = c[-4..-1]) */
};

struct FilterCfg {
    uint32_t shift;  /* prod >> shift = a */
    uint32_t amask;  /* hashed: byte-address mask for a */
    uint32_t lane4;  /* replicated: (lane & 31) * 4 */
    uint32_t c2base; /* LDS byte address of the 2-byte table */
    uint32_t hmask;  /* pair filter: bits of the 3 hashed bytes that enter the hash */
};

/* The filter over one 16-byte chunk: candidate masks, bit q = lookup position q;
 * low half = 4-byte-key hits, high half = 3-/2-byte-key hits. Fully unrolled:
 * every window is a compile-time byte offset into {h.y, d.x, d.y, d.z, d.w}.
 *
 * The kernel is VALU-bound (a wave64 integer instruction occupies its SIMD for 4
 * cycles), so the per-lookup sequence is kept to the minimum the ISA allows:
 *   x      v_alignbyte (aligned offsets: none); case-blinding done once per dword
 *   prod   v_mul_u32_u24
 *   a      v_lshrrev;  addr  v_and;  word  ds_read_b32
 *   4-byte key:  t1 = word >> (a + b3)            v_add_u32_sdwa (b3 = byte select) + v_lshrrev
 *                t2 = word >> (prod.byte1 + b3)   same, both operands byte selects   (K2 only)
 *   3-byte key:  t1 = word >> a, t2 = word >> prod.byte1                            (not BFOLD)
 *   hit = t1 & t2: only bit 0 means anything; v_alignbit shifts exactly that bit into
 *   the top of the accumulator, so no per-lookup mask / shift-into-place is needed.
 * Shifts take their amount mod 32 in hardware, which is the "& 31" of the bit index. */
__device__ __forceinline__ uint32_t shr_lo5(uint32_t v, uint32_t amount) { return v >> (amount & 31u); }
/* a + byte BYTE of src: the byte select is an SDWA operand modifier, not an instruction.
 * (Written out: left to itself the compiler extracts about half of these with shifts.) */
template <int BYTE> __device__ __forceinline__ uint32_t add_byte(uint32_t a, uint32_t src) {
    uint32_t r;
    if (BYTE == 0)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(src));
    else if (BYTE == 1)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(src));
    else if (BYTE == 2)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(src));
    else
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(src));
    return r;
}
/* byte 1 of prod + byte BYTE of src */
template <int BYTE> __device__ __forceinline__ uint32_t add_byte1_byte(uint32_t prod, uint32_t src) {
    uint32_t r;
    if (BYTE == 0)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "=v"(r) : "v"(prod), "v"(src));
    else if (BYTE == 1)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(prod), "v"(src));
    else if (BYTE == 2)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(r) : "v"(prod), "v"(src));
    else
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3" : "=v"(r) : "v"(prod), "v"(src));
    return r;
}
/* word >> byte 1 of prod */
__device__ __forceinline__ uint32_t shr_byte1(uint32_t word, uint32_t prod) {
    uint32_t r;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(prod), "v"(word));
    return r;
}
__device__ __forceinline__ uint32_t push_top(uint32_t hit, uint32_t acc, uint32_t nbits) {
    return __builtin_amdgcn_alignbit(hit, acc, nbits); /* ({hit, acc} >> nbits): low nbits of hit enter at the top */
}

/* Two passes over the chunk's lookup positions: first every hash and every LDS read (8 or 16 reads in
 * flight per lane), then the bit tests. Interleaved, the compiler kept at most two reads ahead of their use
 * and every second lookup waited out the whole LDS latency. */
template <bool REPL, int Q>
__device__ __forceinline__ uint32_t filter_hash(const uint32_t (&arr)[6]) {
    /* 3 bytes ending at c[Q]: byte offset Q + 2 into arr */
    constexpr int O = Q + 2;
    /* byte offset 1 inside a dword: the three bytes are the dword's upper three, a plain shift (twice the issue rate of
     * v_alignbyte); the multiply reads 24 bits, so offset 0 needs nothing */
    const uint32_t x = (O & 3) == 0 ? arr[O >> 2] : (O & 3) == 1 ? arr[O >> 2] >> 8 : alignbyte(arr[(O >> 2) + 1], arr[O >> 2], O & 3);
    return mul_u24(x, HSGPU_FILTER_MUL);
}
template <bool REPL> __device__ __forceinline__ uint32_t filter_addr(uint32_t prod, const FilterCfg &f) {
    const uint32_t a = prod >> f.shift;
    return REPL ? ((a << 7) | f.lane4) : (a & f.amask);
}
/* word >> byte BYTE of src (the hardware reads the low five bits of the selected byte) */
template <int BYTE> __device__ __forceinline__ uint32_t shr_byte(uint32_t word, uint32_t src) {
    uint32_t r;
    if (BYTE == 0) return word >> (src & 31u); /* a plain shift: its amount is the low five bits of the register */
    else if (BYTE == 1)
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(src), "v"(word));
    else if (BYTE == 2)
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(src), "v"(word));
    else
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(src), "v"(word));
    return r;
}
/* HSGPU_F_WIDE: 64-bit entries, the 4-byte key's first bit in lo (index = b3), its second in hi (index = the product): two
 * shifts whose amounts the hardware takes from a byte / from the product as they are, one and, one alignbit */
/* (b3 = c[Q-3] is byte 0 of the three hashed bytes of position Q - 1: `xprev` is that lookup's hash input, still in its register,
 * so the first shift is a plain v_lshrrev too -- as an SDWA byte select it issued at half the rate, 12 times per tile) */
__device__ __forceinline__ void filter_test_wide(uint32_t xprev, uint32_t prod, uint2 word, uint32_t &acc_a) {
    const uint32_t hit = shr_lo5(word.x, xprev) & shr_lo5(word.y, prod);
    acc_a = push_top(hit, acc_a, 1);
}
/* the 3 bytes ending at c[Q] in the low 24 bits (what is above them is ignored by the 24-bit multiply and by a 5-bit shift amount) */
template <int Q> __device__ __forceinline__ uint32_t filter_x(const uint32_t (&arr)[6]) {
    constexpr int O = Q + 2;
    return (O & 3) == 0 ? arr[O >> 2] : (O & 3) == 1 ? arr[O >> 2] >> 8 : alignbyte(arr[(O >> 2) + 1], arr[O >> 2], O & 3);
}
template <bool HAS_A, bool HAS_B, bool HAS_C, bool REPL, bool K2, bool S2, bool BLIND, int Q>
__device__ __forceinline__ void filter_test(const uint32_t (&arr)[6], const FilterCfg &f, uint32_t prod, uint32_t word,
                                            uint32_t &acc_a, uint32_t &acc_o) {
    constexpr uint32_t STEP = S2 ? 2 : 1;
    /* the byte before the hashed ones, c[Q-3]: byte (Q + 1) & 3 of arr[(Q + 1) >> 2] */
    const uint32_t b3src = arr[(Q + 1) >> 2];
    constexpr int B3 = (Q + 1) & 3;
    const uint32_t a = prod >> f.shift;
    if (HAS_A) {
        uint32_t hit = shr_lo5(word, add_byte<B3>(a, b3src));
        if (K2) hit &= shr_lo5(word, add_byte1_byte<B3>(prod, b3src)); /* byte 1 of prod: second bit index */
        acc_a = push_top(hit, acc_a, STEP);
    }
    if (HAS_B || HAS_C) {
        uint32_t hit = 0;
        if (HAS_B) {
            hit = shr_lo5(word, a);
            if (K2) hit &= shr_byte1(word, prod);
        }
        if (HAS_C) {
            constexpr int O = Q + 2;
            const uint32_t x = (O & 3) ? alignbyte(arr[(O >> 2) + 1], arr[O >> 2], O & 3) : arr[O >> 2];
            const uint32_t kc = __builtin_amdgcn_ubfe(x, 8, 16);
            hit |= shr_lo5(lds_word(f.c2base + ((kc >> 5) << 2)), kc);
        }
        acc_o = push_top(hit, acc_o, STEP);
    }
}

template <int BASE, int... I>
__device__ __forceinline__ void filter_positions_wide(const uint32_t (&arr)[6], const FilterCfg &f, uint32_t &acc_a,
                                                      std::integer_sequence<int, I...>) {
    const uint32_t x[sizeof...(I) + 1] = {filter_x<BASE - 1>(arr), filter_x<BASE + I>(arr)...};
    const uint32_t prod[sizeof...(I)] = {mul_u24(x[I + 1], HSGPU_FILTER_MUL)...};
    const uint2 word[sizeof...(I)] = {lds_pair((prod[I] >> f.shift) & f.amask)...};
    __builtin_amdgcn_sched_barrier(0);
    (filter_test_wide(x[I], prod[I], word[I], acc_a), ...);
}
template <bool HAS_A, bool HAS_B, bool HAS_C, bool REPL, bool K2, bool S2, bool BLIND, int BASE, int... I>
__device__ __forceinline__ void filter_positions(const uint32_t (&arr)[6], const FilterCfg &f, uint32_t &acc_a,
                                                 uint32_t &acc_o, std::integer_sequence<int, I...>) {
    constexpr int STEP = S2 ? 2 : 1;
    const uint32_t prod[sizeof...(I)] = {filter_hash<REPL, (BASE + I) * STEP>(arr)...};
    const uint32_t word[sizeof...(I)] = {lds_word(filter_addr<REPL>(prod[I], f))...};
    __builtin_amdgcn_sched_barrier(0);
    (filter_test<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, (BASE + I) * STEP>(arr, f, prod[I], word[I], acc_a, acc_o), ...);
}

template <bool HAS_A, bool HAS_B, bool HAS_C, bool REPL, bool K2, bool S2, bool BLIND, bool WIDE = false>
__device__ __forceinline__ uint32_t filter_chunk(const Chunk &c, const FilterCfg &f) {
    uint32_t arr[6] = {c.h.y, c.d.x, c.d.y, c.d.z, c.d.w, 0u};
    if (BLIND) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            arr[i] &= 0xdfdfdfdfu; /* b3's bit 5 never reaches the 5-bit index either */
            /* (opaque from here: knowing that the multiply reads 24 bits, the compiler otherwise masks the raw dword a second
             * time with 0xdfdfdf for every aligned lookup instead of using this value) */
            asm("" : "+v"(arr[i]));
        }
    }
    uint32_t acc_a = 0, acc_o = 0;
    if (WIDE) { /* stride 1, the 4-byte-key test alone, 64-bit entries */
        filter_positions_wide<0>(arr, f, acc_a, std::make_integer_sequence<int, 16>{});
        return acc_a >> 16;
    }
#ifndef HSGPU_FILTER_BATCHES
#define HSGPU_FILTER_BATCHES 1 /* tuning builds: 2 = the chunk's lookups in two halves (half the registers held across the LDS reads) */
#endif
    constexpr int N_LOOK = S2 ? 8 : 16, PER = N_LOOK / HSGPU_FILTER_BATCHES;
    filter_positions<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, 0>(arr, f, acc_a, acc_o, std::make_integer_sequence<int, PER>{});
    if (HSGPU_FILTER_BATCHES > 1)
        filter_positions<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, PER % N_LOOK>(arr, f, acc_a, acc_o, std::make_integer_sequence<int, PER>{});
    /* 16 / STEP pushes of STEP bits: lookup q sits at bit 16 + q (stride 2: odd bits are noise) */
    constexpr uint32_t KEEP = S2 ? 0x5555u : 0xffffu;
    return ((acc_a >> 16) & KEEP) | (acc_o & (KEEP << 16));
}


/* ---- the pair filter over one 16-byte chunk (table.h "the pair filter") --------------------
 * Lookups at the 8 even positions Q. X(Q) = the dword starting at c[Q-2] = {b2, b1, b0, nx}; the dword
 * starting at c[Q-4] = {b4, b3, b2, b1} is X(Q-2), already there. Plain VOP2 logic (v_and / v_or /
 * v_lshrrev) issues at one wave64 instruction per 2 cycles on gfx950, VOP3 and SDWA forms and v_mul_u32_u24
 * at one per 4 (tools/ubench/valu_rates.hip); byte fields are therefore moved to bit 0 with a plain shift
 * and used as shift amounts (the hardware reads their low 5 bits):
 *   X            aligned dword, or v_alignbit by 16 (every other lookup)
 *   x, prod      v_and (hash mask), v_mul_u32_u24
 *   entry        v_lshrrev, v_and, ds_read_b64 -> {B, A}
 *   tB1, tB2     B >> (W >> 8), B >> (W >> 11)        b3 bits 0..4 and 3..7
 *   tA, tH       A >> (X >> 24), A >> (prod >> 8)     nx bits 0..4; hash bits 8..12
 *   hit          ((tB1 & tB2) | tA) & tH, bit 0 pushed into the accumulator by v_alignbit */
template <int Q> __device__ __forceinline__ uint32_t pair_window(const uint32_t (&arr)[5]) {
    constexpr int O = Q + 2; /* byte offset of c[Q-2] in arr */
    if constexpr ((O & 3) == 0) return arr[O >> 2];
    else return __builtin_amdgcn_alignbit(arr[(O >> 2) + 1], arr[O >> 2], 16);
}
/* all 8 entry reads of a chunk are issued before the first one is used: the LDS latency is paid once per
 * chunk, not once per lookup */
template <int... I>
__device__ __forceinline__ uint32_t pair_chunk_unrolled(const uint32_t (&arr)[5], const FilterCfg &f,
                                                        std::integer_sequence<int, I...>) {
    const uint32_t X[8] = {pair_window<2 * I>(arr)...};
    const uint32_t prod[8] = {mul_u24(X[I] & f.hmask, HSGPU_FILTER_MUL)...};
    const uint2 e[8] = {lds_pair((prod[I] >> f.shift) & f.amask)...};
    __builtin_amdgcn_sched_barrier(0); /* left alone the scheduler keeps only one read ahead of its use */
    uint32_t acc = 0;
    auto test = [&](int i) {
        const uint32_t W = i ? X[i - 1] : arr[0];
        const uint32_t hit = ((shr_lo5(e[i].x, W >> 8) & shr_lo5(e[i].x, W >> 11)) | shr_lo5(e[i].y, X[i] >> 24)) &
                             shr_lo5(e[i].y, prod[i] >> 8);
        acc = push_top(hit, acc, 2);
    };
    (test(I), ...);
    return acc;
}
__device__ __forceinline__ uint32_t pair_filter_chunk(const Chunk &c, const FilterCfg &f) {
    const uint32_t arr[5] = {c.h.y, c.d.x, c.d.y, c.d.z, c.d.w};
    const uint32_t acc = pair_chunk_unrolled(arr, f, std::make_integer_sequence<int, 8>{});
    return (acc >> 16) & 0x5555u; /* lookup Q at bit Q; the spill copies the mask to the other class half */
}

struct SpillState {
    __amdgpu_buffer_rsrc_t rs; /* this wavefront's region of the candidate buffer (32-byte entries), as a buffer descriptor */
    uint32_t cap;      /* entries the region holds */
    uint32_t written;  /* entries appended so far (wave-uniform: kept in a scalar register) */
    uint32_t overflow; /* region exhausted: the scan reports "again" (or falls back to the fused kernel) */
};

/* two-phase: lanes with candidates append {chunk index, masks, 8-byte halo, chunk}
 * straight to this wavefront's private HBM region (ranked by ballot; no atomics,
 * no LDS staging). Everything wave-uniform here -- the fill, the room check, the
 * region -- lives in scalar registers (readfirstlane where the compiler would not see it):
 * as loop-carried vector values the fill and the flag cost two v_cndmask, a vector compare
 * and 64-bit vector address arithmetic (v_lshlrev_b64, v_lshl_add_u64) on every tile of a
 * loop that is bound by vector issue. The stores take a 32-bit lane offset against the
 * region's descriptor. (profiles/r03_scalar_loop_ab.txt) */
__device__ __forceinline__ void spill(SpillState &sp, uint32_t fold_shift, uint32_t chunk_idx, uint32_t acc, const Chunk &c) {
    const bool has = acc != 0;
    const unsigned long long bal = __ballot(has);
    if (!bal) return;
    const uint32_t n = __builtin_amdgcn_readfirstlane(__popcll(bal));
    if (sp.written + n > sp.cap) {
        sp.overflow = 1;
        return;
    }
    if (has) {
        const uint32_t rank =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        /* BFOLD tables: the one class test stands for both (fold_shift = 16, else 0) */
        const v4u e0 = {chunk_idx, acc | acc << fold_shift, c.h.x, c.h.y};
        const v4u e1 = {c.d.x, c.d.y, c.d.z, c.d.w};
        __builtin_amdgcn_raw_buffer_store_b128(e0, sp.rs, rank << 5, sp.written << 5, 0);
        __builtin_amdgcn_raw_buffer_store_b128(e1, sp.rs, (rank << 5) + 16u, sp.written << 5, 0);
    }
    sp.written = __builtin_amdgcn_readfirstlane(sp.written + n);
}

/* fused: push into the wavefront's LDS queue; confirm 64 at a time */
template <bool HAS_A, bool HAS_B, bool HAS_C, bool S2, bool PAIR = false>
__device__ __forceinline__ void enqueue_fused(const Tables &t, uint32_t &qcount, uint32_t lane, uint64_t coff,
                                              uint32_t acc) {
    uint2 *queue = t.wl->cand;
    const bool has = acc != 0;
    const unsigned long long bal = __ballot(has);
    if (bal) {
        if (has) {
            uint32_t idx = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
            queue[idx] = make_uint2((uint32_t)(coff >> 4), acc);
        }
        qcount += __popcll(bal);
        if (qcount >= 64) {
            qcount -= 64;
            drain_entry<HAS_A, HAS_B, HAS_C, S2, PAIR>(t, queue[qcount + lane]);
            flush_records(t, lane, OFLUSH);
        }
    }
}

__device__ __forceinline__ void init_tables(Tables &t, const HsgpuScanArgs &args) {
    t.corpus = args.corpus;
    t.off = args.off;
    t.nblocks = args.nblocks;
    t.start = args.start;
    t.total = args.total;
    t.ht_a = (const uint4 *)(args.blob + args.t_off_ht_a);
    t.ht_b = (const uint4 *)(args.blob + args.t_off_ht_b);
    t.c2ref = (const uint32_t *)(args.blob + args.t_off_c2ref);
    t.gate = (const uint32_t *)(args.blob + args.t_off_c2bits);
    t.key_gate = nullptr;
    t.bloom = nullptr;
    t.lists = (const uint32_t *)(args.blob + args.t_off_lists);
    t.lits = (const HsgpuDevLit *)(args.blob + args.t_off_lits);
    t.ht_a_log2 = args.t_ht_a_log2;
    t.ht_b_log2 = args.t_ht_b_log2;
    t.key_mask = (args.t_flags & HSGPU_F_BLIND) ? 0xdfdfdfdfu : 0xffffffffu;
    t.hint = args.hint;
    t.n_hint = args.n_hint;
    t.wl = nullptr;
    t.rec_region = nullptr;
    t.rec_cap = 0;
    t.share_start = 0;
    t.late_skip = ~0ull;
    t.mq_redo = 0;
    t.cb = t.cb_start = t.cb_end = 0;
}

__device__ __forceinline__ void init_wave_lds(Tables &t, WaveLds *wl, uint32_t lane) {
    t.wl = wl;
    if (lane == 0) {
        /* (an opaque zero: inside the persistent confirm loop the compiler otherwise keeps a quad of zero registers alive
         * across the whole confirm step for these stores -- four registers of a budget that decides the occupancy) */
        uint32_t z = 0;
        asm volatile("" : "+v"(z));
        wl->nrec = z;
        wl->nfront = z;
        wl->nback = z;
        wl->nmq = z;
        wl->nrq = z;
        wl->pad[0] = z; /* "resolved a match in place" (push_match) */
        wl->pad[RUN_VIRT] = z;
        wl->pad[RUN_N] = z;
        wl->pad[RUN_LIVE] = z;
    }
}

/* ---- block hints: hint[t] = block containing corpus byte t * 1024 ------------------ */
/* Convergent (whole wavefront): the lane's block b, spanning corpus bytes [s, e), writes the hint of every
 * KiB boundary inside it (no searching: a per-tile bisection was 20 dependent reads per entry); blocks
 * covering more than 4 boundaries are written by the whole wavefront. Block index nblocks stands for
 * "boundaries at/after the last offset" (s = off[nblocks]). */
__device__ __forceinline__ void write_block_hints_of(uint64_t s, uint64_t e, uint64_t nblocks, uint32_t *hint,
                                                     uint64_t n_hint, uint64_t b, uint32_t lane) {
    uint64_t t0 = 0, t1 = 0;
    uint32_t val = 0;
    if (b < nblocks) {
        t0 = (s + ((1ull << HSGPU_HINT_SHIFT) - 1)) >> HSGPU_HINT_SHIFT;
        t1 = (e + ((1ull << HSGPU_HINT_SHIFT) - 1)) >> HSGPU_HINT_SHIFT;
        val = (uint32_t)b;
    } else if (b == nblocks) {
        t0 = (s + ((1ull << HSGPU_HINT_SHIFT) - 1)) >> HSGPU_HINT_SHIFT;
        t1 = n_hint;
        val = (uint32_t)(nblocks - 1);
    }
    t1 = min(t1, n_hint);
    const uint64_t n = t1 > t0 ? t1 - t0 : 0;
    if (n <= 4) {
        for (uint64_t k = 0; k < n; k++) hint[t0 + k] = val;
    }
    unsigned long long wide = __ballot(n > 4);
    while (wide) {
        const int l = __builtin_ctzll(wide);
        wide &= wide - 1;
        const uint64_t T0 = __shfl(t0, l), T1 = __shfl(t1, l);
        const uint32_t V = __shfl(val, l);
        for (uint64_t t = T0 + lane; t < T1; t += 64) hint[t] = V;
    }
}
__device__ __forceinline__ void write_block_hints(const uint64_t *off, uint64_t nblocks, uint32_t *hint, uint64_t n_hint,
                                                  uint64_t b, uint32_t lane) {
    const uint64_t s = (b && b <= nblocks) ? off[b] : 0, e = b < nblocks ? off[b + 1] : 0;
    write_block_hints_of(s, e, nblocks, hint, n_hint, b, lane);
}
/* K x 64 consecutive blocks per wavefront, every offset read before the first is used: one round trip to
 * memory for the whole batch (with one block per lane and step, a 1 GiB corpus of packets cost every
 * wavefront seven dependent round trips before it could start streaming) */
template <int K>
__device__ __forceinline__ void write_block_hints_batch(const uint64_t *off, uint64_t nblocks, uint32_t *hint,
                                                        uint64_t n_hint, uint64_t b0, uint32_t lane) {
    uint64_t s[K], e[K];
#pragma unroll
    for (int k = 0; k < K; k++) { /* unconditional loads at clamped indices: one wait for all 2K of them */
        const uint64_t b = b0 + (uint64_t)k * 64 + lane;
        s[k] = off[min(b, nblocks)];
        e[k] = off[min(b + 1, nblocks)];
    }
#pragma unroll
    for (int k = 0; k < K; k++)
        if (b0 + (uint64_t)k * 64 + lane == 0) s[k] = 0;
#pragma unroll
    for (int k = 0; k < K; k++) write_block_hints_of(s[k], e[k], nblocks, hint, n_hint, b0 + (uint64_t)k * 64 + lane, lane);
}



__device__ __forceinline__ void solo_tail(const HsgpuScanArgs &args, uint32_t *lds, uint32_t n_reg); /* below, behind sort_share */

/* ---- phase 1: the streaming filter (FUSED: + in-kernel confirm) ----------- */
#ifndef HSGPU_FILTER_MIN_WAVES
#define HSGPU_FILTER_MIN_WAVES 1 /* tuning builds: 8 caps the kernel at 64 VGPRs so that two 16-wavefront workgroups fit a CU */
#endif
__device__ __forceinline__ bool rec_less(const uint4 a, const uint4 b) { /* (block, end, literal index) */
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.w < b.w;
}

/* Workgroup-convergent. A solo scan of ONE workgroup whose wavefronts still hold every record they found in their LDS staging
 * (nothing flushed to a region, nothing spilled: the usual packet): the records ranked inside their wavefront's handful (a
 * wavefront's records are those of one piece of the corpus) and written to where they go -- the wavefronts in front + the rank --
 * and the count: one barrier, LDS reads, one store per record. false (uniform): somebody has flushed; the caller takes solo_tail.
 * (solo_tail reads fills, regions and records back from memory: four dependent round trips of the 12 us a 1 460-byte request took.) */
__device__ __forceinline__ bool solo_place_lds(const HsgpuScanArgs &args, WaveLds *wls, uint32_t W, uint32_t *inl = nullptr) {
    __syncthreads(); /* every wavefront has confirmed its share */
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    /* every wavefront looks at all of them (<= 16; the same LDS words for every lane group: broadcasts) */
    uint32_t my_n = 0, dirty = 0;
    if (lane < W) {
        const WaveLds *w = wls + lane;
        my_n = __hip_atomic_load(&w->nrec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        dirty = (w->nfront | __hip_atomic_load(&w->nback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0 || my_n > (uint32_t)OCAP;
    }
    if (__ballot(dirty)) return false;
    uint32_t incl = my_n;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += v;
    }
    const uint32_t all = __shfl(incl, (int)W - 1);
    if (all > args.cap) return false; /* (solo_tail says "again" the way record_sort_kernel does) */
    /* the small-batch server's answer line (HsgpuServerCtl, scan_kernels.h): <= HSGPU_SRV_INLINE_RECS records stay in LDS, where the
     * server's wavefront 0 picks them up and sends records, count and sequence number in ONE store */
    const bool to_line = inl != nullptr && all <= (uint32_t)HSGPU_SRV_INLINE_RECS;
    uint4 *out = (uint4 *)args.out;
    /* a thread per staging slot: wavefront tid / 32, record tid % 32 (OCAP <= 32; the workgroup has 64 threads per wavefront) */
    static_assert(OCAP <= 32, "a staging slot per half wavefront");
    const uint32_t w = min(tid >> 5, W - 1u), j = tid & 31u;
    const uint32_t n = __shfl(my_n, (int)w), at = __shfl(incl, (int)w) - n; /* (every wavefront holds every fill in its lanes 0 .. W - 1) */
    if (tid < W * 32u && j < n) {
        const uint4 rec = wls[w].rec[j];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; q++) {
            const uint4 o = wls[w].rec[q];
            rank += (rec_less(o, rec) || (q < j && !rec_less(rec, o))) ? 1u : 0u;
        }
        if (to_line) ((uint4 *)inl)[at + rank] = rec;
        else out[at + rank] = rec;
    }
    if (tid == 0) {
        if (to_line) inl[12] = all, inl[15] = 1u;
        else *args.count = all;
        if (args.tstamp) { /* as solo_tail: one kernel, every stage ends here */
            const unsigned long long now = wall_clock64();
            args.tstamp[1] = now, args.tstamp[2] = now, args.tstamp[3] = now;
            if (args.tstamp_next) args.tstamp_next[0] = ~0ull, args.tstamp_next[1] = 0, args.tstamp_next[2] = 0, args.tstamp_next[3] = 0;
        }
    }
    return true;
}

/* The kernel's body as a function (round 6): hwlm_filter_kernel runs it once, hwlm_server_kernel -- a resident workgroup that
 * serves small host batches without a launch per call -- once per request. Every `return` below ends one scan. */
template <bool HAS_A, bool HAS_B, bool HAS_C, bool REPL, bool K2, bool S2, bool BLIND, bool FUSED, bool PAIR = false, bool WIDE = false>
__device__ __forceinline__ bool hwlm_filter_body(const HsgpuScanArgs &args, uint32_t *lds, const uint32_t bdim, const uint32_t gdim) {
    /* bdim / gdim: bdim / gdim, handed in. (They are loads -- of the dispatch packet or the hidden kernel arguments, both
     * in HOST memory -- that the compiler repeats rather than keeps: inside the small-batch server's loop, behind the request's
     * acquire, that was a read over the bus in front of the first tile's loads, ~1.5 us of every request. The server reads them
     * once, in front of its loop.) */
    /* -> true (uniform): solo_tail has used the head of the table image in LDS as its scratch (the server loads it again) */
    /* the fused kernel doubles as the overflow fallback: nothing to do unless the
     * two-phase pipeline ran out of candidate space */
    if (FUSED && args.cand_counts && !args.cand_counts[args.cand_waves]) {
        /* it runs right behind the confirm kernel: its start is the end of the confirm stage */
        if (args.tstamp && blockIdx.x == 0 && threadIdx.x == 0) args.tstamp[2] = wall_clock64();
        return false;
    }
    bool image_used = false;
    if (FUSED && args.cand_counts && args.overflow_note && blockIdx.x == 0 && threadIdx.x == 0) *args.overflow_note = 1u;
    if ((!FUSED || args.solo) && args.tstamp && threadIdx.x == 0) atomicMin(&args.tstamp[0], (unsigned long long)wall_clock64());
    if ((!FUSED || args.solo) && args.wg_stamps && threadIdx.x == 0) args.wg_stamps[4 * blockIdx.x] = wall_clock64();

    const uint32_t flog2 = args.t_filter_log2;
    const uint32_t nw = (PAIR || WIDE) ? (2u << flog2) : REPL ? (32u << flog2) : (1u << flog2);
    uint32_t *filter = lds;
    uint32_t *c2bits = lds + nw;

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t WAVES = bdim >> 6;           /* 16 or 8 */
    const uint32_t n_waves = gdim * WAVES;
    const uint32_t wave_global = blockIdx.x * WAVES + wave;

    const uint8_t *corpus = args.corpus;
    const uint64_t total = args.total;
    /* Every wavefront streams ONE contiguous share of the corpus, 1 KiB tiles [tile, tile_end): measured
     * 8-20% faster than all workgroups walking the corpus together (tile = blockIdx + k * grid), and the
     * records found in a wavefront's candidates are then the records of one corpus range: delivery order
     * costs one small sort per share instead of a global one (phase 3). */
    const uint64_t n_full = total >> 10; /* 1 KiB tiles with every load in bounds */
    /* (the fused body -- solo scans, the small-batch server -- in 32-bit arithmetic: a corpus has at most 2^26 tiles, and the two
     * 64-bit divisions were ~1.3 us of dependent instructions in front of the first load of a 6.5 us request) */
    const uint64_t per_wave = FUSED ? (uint64_t)(((uint32_t)n_full + n_waves - 1) / n_waves) : (n_full + n_waves - 1) / n_waves;
    /* the wavefront whose share would hold tile n_full, the partial last one (the last wavefront when the shares come out even) */
    const uint64_t tail_wave = !per_wave ? 0 : FUSED ? (uint64_t)min(n_waves - 1, (uint32_t)n_full / (uint32_t)per_wave) : min((uint64_t)n_waves - 1, n_full / per_wave);
    /* The share as wave-uniform SCALARS: its first tile (64-bit) and the number of tiles in it (32-bit; a share of
     * 2^32 tiles would be 4 TiB). The 64-bit division above leaves its result in vector registers; with `tile` and
     * `tile_end` taken from there every stage of the loop below paid three 64-bit vector compares and two 64-bit moves
     * for its "in range" and "done" tests. Now they are s_cmp on a 32-bit counter. */
    const uint64_t tile0 = rfl64(min(n_full, (uint64_t)wave_global * per_wave));
    const uint32_t n_own = __builtin_amdgcn_readfirstlane((uint32_t)(min(n_full, tile0 + per_wave) - tile0));
    uint32_t tk = 0; /* tiles of the share done */
    const uint32_t lane_off = lane * CHUNK;

    /* Loads of one tile through a buffer descriptor built from wave-uniform values only
     * (scalar registers): the lane offset is a 32-bit voffset, so a stage costs no vector
     * address arithmetic at all, nothing branches, and the compiler keeps the loads in
     * flight across iterations with counted s_waitcnt. The descriptor starts 8 bytes in
     * front of the tile (halo at voffset, chunk at voffset + 8 via soffset); a tile past
     * the end of the wavefront's share gets an empty descriptor: its loads return zeros and
     * touch no memory, which replaces every bounds check. */
    auto issue = [&](uint32_t k) -> Chunk { /* tile k of the share */
        const uint8_t *base = corpus + ((tile0 + k) << 10) - 8;
        const int records = k < n_own ? (int)0x7ffffff0 : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, records, 0x00020000);
        Chunk c;
        const auto d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, 8, HSGPU_LOAD_AUX);
        c.d = make_uint4(d[0], d[1], d[2], d[3]);
        const auto h = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, 0, HSGPU_LOAD_AUX);
        c.h = make_uint2(h[0], h[1]);
        return c;
    };
    /* tile 0 has nothing in front of it: descriptor at the corpus itself, and the
     * first lane's halo offset wraps out of range, which reads as the zeros it needs */
    auto issue_first = [&]() -> Chunk {
        if (tile0) return issue(0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)corpus, 0, (int)0x7ffffff0, 0x00020000);
        Chunk c;
        const auto d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, 0, HSGPU_LOAD_AUX);
        const auto h = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off - 8u, 0, HSGPU_LOAD_AUX);
        c.d = make_uint4(d[0], d[1], d[2], d[3]);
        c.h = make_uint2(h[0], h[1]);
        return c;
    };

    /* Prologue, ordered by when each load is needed (loads of one wavefront complete in issue order): the
     * filter image for LDS (L2 hits, wanted first), the block offsets of this wavefront's hints, then the
     * first tiles of the corpus, which stay in flight while the image is written to LDS and the hints are
     * stored. (With the LDS copy and the hints in front, one load and one wait at a time, the memory pipeline
     * idled for the first 25-30 us of every scan.) */
#ifndef HSGPU_PROLOGUE_IMG
#define HSGPU_PROLOGUE_IMG 8
#define HSGPU_PROLOGUE_HK 8
#endif
    constexpr int IMG = HSGPU_PROLOGUE_IMG; /* 16-byte pieces of the image per thread in the first batch: 128 KiB / 1024 threads */
    const uint4 *img_src = (const uint4 *)(args.blob + args.t_off_filter);
    /* (the small-batch server from its second request on: the image is in LDS but for its head, which the placement of the request
     * before used as scratch -- solo_tail) */
    const bool img_kept = FUSED && args.img_keep_words != 0;
    const uint32_t nw_load = img_kept ? min(nw, args.img_keep_words) : nw;
    uint4 img[IMG];
#pragma unroll
    for (int u = 0; u < IMG; u++) {
        const uint32_t i = threadIdx.x + u * bdim;
        img[u] = i < nw_load / 4 ? img_src[i] : make_uint4(0, 0, 0, 0);
    }
#ifndef HSGPU_HINTS_LATE
#define HSGPU_HINTS_LATE 1 /* the block hints are written after the wavefront's share, not in front of it (0: in the prologue, as in
                            * round 3: the median workgroup then started streaming after 15.4 us instead of 7.4, the kernel took 9 us
                            * longer: gpurun_out r4a, profiles/r04_filter_wg_stamps.txt) */
#endif
    const bool hints = !FUSED && args.hint_in_filter;
    constexpr int HK = HSGPU_PROLOGUE_HK; /* every lane takes 8 consecutive blocks: 9 offsets, 512 blocks per wavefront and step */
#if !HSGPU_HINTS_LATE
    const uint64_t hb0 = ((uint64_t)wave_global * 64 + lane) * HK;
#endif
#if !HSGPU_HINTS_LATE
    uint64_t ho[HK + 1];
#pragma unroll
    for (int k = 0; k <= HK; k++) ho[k] = hints ? args.off[min(hb0 + k, args.nblocks)] : 0; /* clamped: no branches */
#endif
    const bool streaming = n_own != 0;
    Chunk c0, c1, c2, c3, c4, c5, c6, c7;
    c0.d = c1.d = c2.d = c3.d = c4.d = c5.d = c6.d = c7.d = make_uint4(0, 0, 0, 0);
    c0.h = c1.h = c2.h = c3.h = c4.h = c5.h = c6.h = c7.h = make_uint2(0, 0);
    if (streaming) {
        c0 = issue_first(), c1 = issue(1), c2 = issue(2);
#if HSGPU_STAGES >= 6
        c3 = issue(3), c4 = issue(4);
#endif
#if HSGPU_STAGES >= 8
        c5 = issue(5), c6 = issue(6);
#endif
    }
    /* The small-batch server (its batch ends in 16 zero bytes wherever it lies: runtime.hip, scan_host_small): the partial last tile
     * is asked for HERE, beside the first tiles, with whole 16-byte loads -- behind the streaming loop it was a memory round trip of
     * its own plus a byte-wise loop for the lane that holds the corpus' end (1.3 us of a 6.5 us request: the other wavefronts
     * waited for this one at the placement's barrier). */
    Chunk ct;
    ct.d = make_uint4(0, 0, 0, 0);
    ct.h = make_uint2(0, 0);
    const bool tail_early = FUSED && args.srv_inline_at != 0;
    if (tail_early && (total & 1023) && wave_global == tail_wave) {
        const uint64_t coff = (n_full << 10) + lane_off;
        if (coff < total) {
            ct.d = *(const uint4 *)(corpus + coff);
            if (coff) ct.h = *(const uint2 *)(corpus + coff - 8);
        }
    }
    if (FUSED && args.srv_inline_at && args.wg_stamps && threadIdx.x == 0) args.wg_stamps[4] = wall_clock64(); /* (server: the loads are out) */
    if (FUSED && args.solo) {
        /* solo scans: no kernel ran in front of this one, so every wavefront writes the block hints of its OWN tiles (and of the
         * boundary behind them: block_of reads hint[tile] and hint[tile + 1]) before it needs them -- a wave-wide search each, two
         * dependent rounds, behind the image and corpus loads already in flight. (Resolving matches by plain bisection was 11
         * dependent loads per match: a 1 MiB solo scan took 61 us against the three kernels' 33.) */
        const bool owns_tail = (total & 1023) && wave_global == tail_wave;
        if (n_own || owns_tail) {
            const uint64_t t_last = min(tile0 + n_own, args.n_hint - 1);
            for (uint64_t tt = tile0; tt <= t_last; tt++) {
                /* (one block -- a hwlmExec call --: nothing to search, and nothing to read over the bus when the offsets are the host's) */
                const uint64_t b = args.nblocks == 1 ? ((tt << HSGPU_HINT_SHIFT) >= total ? 1u : 0u)
                                                     : find_block_wave(args.off, args.nblocks, tt << HSGPU_HINT_SHIFT, lane);
                if (lane == 0) ((uint32_t *)args.hint)[tt] = (uint32_t)b;
            }
        }
    }
    /* the filter image into LDS: once per workgroup */
#pragma unroll
    for (int u = 0; u < IMG; u++) {
        const uint32_t i = threadIdx.x + u * bdim;
        if (i < nw_load / 4) ((uint4 *)filter)[i] = img[u];
    }
    for (uint32_t i = threadIdx.x + IMG * bdim; i < nw_load / 4; i += bdim) ((uint4 *)filter)[i] = img_src[i];
    if (HAS_C && !img_kept) {
        const uint4 *src2 = (const uint4 *)(args.blob + args.t_off_c2bits);
        for (uint32_t i = threadIdx.x; i < 512; i += bdim) ((uint4 *)c2bits)[i] = src2[i];
    }
    /* (two-phase: 64 bytes behind the tables hold the workgroup's progress sum, hsgpu_filter_lds_bytes) */
    uint32_t *wg_done = lds + nw + (HAS_C ? 2048 : 0);
    if (!FUSED && threadIdx.x == 0) *wg_done = 0;
    /* two-phase: the confirm kernel's block hints are written here (a hint kernel on a side stream needed a
     * fork/join pair of cross-stream waits around the confirm launch) */
#if !HSGPU_HINTS_LATE
    if (hints) {
#pragma unroll
        for (int k = 0; k < HK; k++)
            write_block_hints_of(hb0 + k ? ho[k] : 0, ho[k + 1], args.nblocks, (uint32_t *)args.hint, args.n_hint, hb0 + k, lane);
        for (uint64_t w0 = (uint64_t)wave_global + n_waves; w0 * (64 * HK) <= args.nblocks; w0 += n_waves)
            write_block_hints_batch<HK>(args.off, args.nblocks, (uint32_t *)args.hint, args.n_hint, w0 * (64 * HK), lane);
    }
#endif
    if (FUSED && args.srv_inline_at && args.wg_stamps && threadIdx.x == 0) args.wg_stamps[5] = wall_clock64(); /* (server: in front of the barrier) */
    __syncthreads();
    if ((!FUSED || args.solo) && args.wg_stamps && threadIdx.x == 0) args.wg_stamps[4 * blockIdx.x + 1] = wall_clock64();

    Tables t;
    uint32_t qcount = 0;
    SpillState sp;
    sp.rs = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    sp.cap = 0;
    sp.written = 0;
    sp.overflow = 0;
    if (FUSED) {
        init_tables(t, args);
        t.share_start = tile0 << 10;
        init_wave_lds(t, (WaveLds *)(lds + nw + (HAS_C ? 2048 : 0)) + wave, lane);
        t.rec_region = args.rec_stage + (uint64_t)wave_global * args.rec_cap;
        t.rec_cap = args.rec_cap;
    } else {
        /* (a region is at most a few hundred MiB: cand_cap entries of 32 bytes; runtime.hip keeps it below 2 GiB) */
        sp.rs = __builtin_amdgcn_make_buffer_rsrc((void *)(args.cand + 2ull * wave_global * args.cand_cap), 0,
                                                  (int)min((uint64_t)args.cand_cap * 32u, (uint64_t)0x7ffffff0), 0x00020000);
        sp.cap = args.cand_cap;
    }

    FilterCfg f;
    f.shift = (PAIR || WIDE) ? 29u - flog2 : REPL ? 32u - flog2 : 30u - flog2;
    f.amask = PAIR ? ((1u << flog2) - 1u) << 3 : WIDE ? ((1u << flog2) - 1u) << 3 : (nw - 1u) << 2;
    f.hmask = args.t_hash_mask;
    f.lane4 = (lane & 31u) << 2;
    f.c2base = nw * 4;

/* (the chunk index of a lane in 32-bit arithmetic, as the entry holds it: corpora below 64 GiB, like everything that reads it) */
#define HSGPU_SPILL(args, sp, coff, acc, cur) spill(sp, args.fold_shift, ((uint32_t)(tile0 + tk) << 6) | lane, acc, cur)
#define HSGPU_HANDLE(CUR, COFF)                                                                   \
    {                                                                                             \
        const uint32_t acc = PAIR ? pair_filter_chunk(CUR, f)                                     \
                                  : filter_chunk<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, WIDE>(CUR, f); \
        /* (WIDE: one test stands for both key classes; the in-kernel confirm looks at a class's own half of the mask) */ \
        if (FUSED) enqueue_fused<HAS_A, HAS_B, HAS_C, S2, PAIR>(t, qcount, lane, (COFF), (WIDE && HAS_B) ? (acc | acc << 16) : acc); \
        else HSGPU_SPILL(args, sp, (COFF), acc, CUR);                                             \
    }

    if (streaming) {
        /* HSGPU_STAGES register stages rotate by name (loop unrolled to match): all but
         * one tile are in flight while that one is filtered, a stage is never copied, and
         * the only wait is for the stage about to be filtered. */
#define HSGPU_STAGE(CUR, NEW)                                            \
    {                                                                    \
        NEW = issue(tk + (HSGPU_STAGES - 1));                            \
        const uint64_t coff = ((tile0 + tk) << 10) | lane_off;           \
        HSGPU_HANDLE(CUR, coff)                                          \
        tk += 1;                                                         \
        if (tk >= n_own) break;                                          \
    }
#ifndef HSGPU_BALANCE
#define HSGPU_BALANCE 1 /* 0: round 3's loop (the same box, back to back: fdr10k filter 0.2998 -> 0.2855 ms, 8 GiB 2.341 -> 2.217 ms,
                         * teddy64 0.2175 -> 0.2114; wavefront 0 done at 253 of 263 us instead of 181 of 273: profiles/r04_filter_wg_stamps.txt) */
#endif
#if HSGPU_STAGES == 8
        for (;;) {
#if HSGPU_BALANCE
            /* The wavefronts of a workgroup do not advance together: the instruction arbiter prefers the oldest, and wavefront
             * 0 has streamed its share after 2/3 of the kernel (profiles/r04_filter_wg_stamps.txt) -- the last third runs with
             * ever fewer wavefronts per SIMD to hide latency behind. Every eight tiles a wavefront adds its progress to the
             * workgroup's sum in LDS and raises its priority while it is behind the mean, lowers it while ahead. */
            if (!FUSED) {
                uint32_t tot = 0;
                if (lane == 0) tot = __hip_atomic_fetch_add(wg_done, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                tot = __builtin_amdgcn_readfirstlane(tot);
                if (tk * WAVES < tot) __builtin_amdgcn_s_setprio(3);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            HSGPU_STAGE(c0, c7)
            HSGPU_STAGE(c1, c0)
            HSGPU_STAGE(c2, c1)
            HSGPU_STAGE(c3, c2)
            HSGPU_STAGE(c4, c3)
            HSGPU_STAGE(c5, c4)
            HSGPU_STAGE(c6, c5)
            HSGPU_STAGE(c7, c6)
        }
#elif HSGPU_STAGES == 6
        for (;;) {
            HSGPU_STAGE(c0, c5)
            HSGPU_STAGE(c1, c0)
            HSGPU_STAGE(c2, c1)
            HSGPU_STAGE(c3, c2)
            HSGPU_STAGE(c4, c3)
            HSGPU_STAGE(c5, c4)
        }
#else
        for (;;) {
            HSGPU_STAGE(c0, c3)
            HSGPU_STAGE(c1, c0)
            HSGPU_STAGE(c2, c1)
            HSGPU_STAGE(c3, c2)
        }
#endif
#undef HSGPU_STAGE
    }

    /* the partial last tile: guarded, byte-wise where needed; the wavefront whose share would hold tile n_full
     * (the last one when the shares come out even). Bytes at/after the end of the corpus read as zero and
     * their lookup positions are masked off. */
    if ((total & 1023) && wave_global == tail_wave) {
        const uint64_t coff = (n_full << 10) + lane_off;
        Chunk c;
        c.d = make_uint4(0, 0, 0, 0);
        c.h = make_uint2(0, 0);
        if (tail_early) {
            c = ct;
        } else if (coff < total) {
            if (coff + CHUNK <= total) {
                c.d = *(const uint4 *)(corpus + coff);
            } else {
                uint32_t tmp[4] = {0, 0, 0, 0};
                const uint32_t n = (uint32_t)(total - coff);
                for (uint32_t i = 0; i < n; i++) tmp[i >> 2] |= (uint32_t)corpus[coff + i] << (8 * (i & 3));
                c.d = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
            if (coff) c.h = *(const uint2 *)(corpus + coff - 8);
        }
        uint32_t acc = PAIR ? pair_filter_chunk(c, f) : filter_chunk<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, WIDE>(c, f);
        uint32_t valid = 0;
        if (coff < total) valid = (coff + CHUNK <= total) ? 0xffffu : ((1u << (uint32_t)(total - coff)) - 1u);
        acc &= valid | valid << 16;
        if (FUSED) enqueue_fused<HAS_A, HAS_B, HAS_C, S2, PAIR>(t, qcount, lane, coff, (WIDE && HAS_B) ? (acc | acc << 16) : acc);
        else spill(sp, args.fold_shift, (uint32_t)(coff >> 4), acc, c);
    }
#undef HSGPU_HANDLE

    if (FUSED) {
        if (lane < qcount) drain_entry<HAS_A, HAS_B, HAS_C, S2, PAIR>(t, t.wl->cand[lane]);
        /* pair tables: the late-keyed literals ending at this share's last byte (the next share's first lookup would find them) */
        /* ... and the corpus' last byte: the lookup position behind it does not exist. (Round 6, found by the guard-page tests: a
         * 3-byte literal keyed late and ending at byte total - 1 was never reported; nor was one ending at the last byte of the
         * last whole tile when the partial tile behind it belonged to the next wavefront.) The wavefront with the partial tile
         * asks at `total` when that is a lookup position, every other one at the first byte behind its tiles. */
        if (PAIR && HAS_B && lane == 0) {
            const bool holds_tail = (total & 1023) && wave_global == tail_wave;
            /* (the lookups sit on even positions: behind an odd total there is none to stand in for -- the end at total - 1 is even) */
            const uint64_t edge = holds_tail ? ((total & 1) ? 0 : total) : n_own ? (tile0 + n_own) << 10 : 0;
            if (edge) pair_edge_probe<false>(t, edge);
        }
        if (args.solo && args.wg_stamps && wave == 0 && lane == 0) args.wg_stamps[4 * blockIdx.x + 2] = wall_clock64(); /* (wavefront 0's share confirmed) */
        /* ONE workgroup (the small-batch server; a solo scan of one super tile) whose wavefronts have staged everything they found
         * in LDS: placed from there (solo_place_lds) -- no region, no count and no control word goes through memory */
        if (!(args.solo && gdim == 1 && solo_place_lds(args, (WaveLds *)(lds + nw + (HAS_C ? 2048 : 0)), WAVES, args.srv_inline_at ? lds + args.srv_inline_at : nullptr))) {
            publish_records(t, args, lane, wave_global);
            if (args.solo) solo_tail(args, lds, n_waves), image_used = true;
        }
        if (args.solo && args.wg_stamps && threadIdx.x == 0) args.wg_stamps[4 * blockIdx.x + 3] = wall_clock64();
    } else if (lane == 0) {
        args.cand_counts[wave_global] = sp.written;
        if (sp.overflow) args.cand_counts[n_waves] = 1;
    }
    if (!FUSED && args.wg_stamps && wave == 0 && lane == 0) args.wg_stamps[4 * blockIdx.x + 2] = wall_clock64(); /* (wavefront 0's share done) */
#if HSGPU_HINTS_LATE
    /* the hints depend on the offsets alone and only the NEXT kernel reads them: written behind the share, they cost the
     * wavefronts that finish early nothing that anybody waits for, instead of sitting in front of every wavefront's first tile */
    if (hints)
        for (uint64_t w0 = (uint64_t)wave_global; w0 * (64 * HK) <= args.nblocks; w0 += n_waves)
            write_block_hints_batch<HK>(args.off, args.nblocks, (uint32_t *)args.hint, args.n_hint, w0 * (64 * HK), lane);
#endif
    if (!FUSED && (args.tstamp || args.wg_stamps)) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long now = wall_clock64();
            if (args.tstamp) atomicMax(&args.tstamp[1], now);
            if (args.wg_stamps) args.wg_stamps[4 * blockIdx.x + 3] = now;
        }
    }
    return image_used;
}

template <bool HAS_A, bool HAS_B, bool HAS_C, bool REPL, bool K2, bool S2, bool BLIND, bool FUSED, bool PAIR = false, bool WIDE = false>
__global__ __launch_bounds__(WG_THREADS, HSGPU_FILTER_MIN_WAVES) void hwlm_filter_kernel(HsgpuScanArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds != 0) __builtin_trap();
    hwlm_filter_body<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, FUSED, PAIR, WIDE>(args, lds, blockDim.x, gridDim.x);
}

/* ---- the small-batch server (round 6) -----------------------------------------------------------------------------------
 * hsbench block mode is one hs_scan per block (tools/hsbench/engine_hyperscan.cpp:132-145) and the reference serves a packet in
 * about a microsecond (src/rose/block.c:382-391, src/runtime.c:401-413). A launch per call costs this engine ~25 us: the launch,
 * the dispatch, a stream synchronisation that sleeps. The server is ONE workgroup that stays resident and polls a request word in
 * mapped page-locked host memory: the host copies the batch (<= one 16 KiB super tile: what one workgroup scans) into the
 * scratch's mapped area -- offsets | corpus, as for a solo scan --, writes the parameters and then the sequence number (release);
 * the workgroup runs the solo scan's body (filter + confirm inline + placement by the one workgroup) with records and count
 * going back to the same mapped area, fences at system scope and writes the done word, which the host spins on. No launch, no
 * interrupt, no copy command.
 *   idle        a server that has seen no request for idle_ticks of the 100 MHz wall clock says so (exited = 1) and ENDS: a
 *               forgotten server cannot hold a CU (or a device synchronisation) for longer than that, and nothing can hang; the
 *               host launches the next one when the next small call comes (runtime.hip, server_call)
 *   stop        the host's way of ending it at once (hsgpu_scratch_free, a scan on the same scratch that needs the buffers)
 * req / done are words of their own cache lines; the parameters are read AFTER the sequence number has been seen to change.
 * `req` (the request lines: stop, req_seq, total, nblocks, start) and src_* may be DEVICE memory that the host writes through the
 * PCIe BAR (runtime.hip, bar_area): the poll and the batch's tiles are then local reads -- a 1.5 KB request's round trip through
 * one wavefront is 4.4 us instead of 10.5 (tools/experiments/bar_mailbox.hip) --, `ctl` (done_seq, exited, stamps), records and
 * count stay in mapped host memory, which the host polls and reads at memory speed. */
#ifndef HSGPU_SRV_ACQUIRE_SCOPE
#define HSGPU_SRV_ACQUIRE_SCOPE "" /* system (tuning builds: "agent" -- the vector L1 alone) */
#endif
#ifndef HSGPU_SRV_POLL_SLEEP
#define HSGPU_SRV_POLL_SLEEP 2
#endif
template <bool HAS_A, bool HAS_B, bool HAS_C, bool REPL, bool K2, bool S2, bool BLIND, bool PAIR = false, bool WIDE = false>
__global__ __launch_bounds__(WG_THREADS, HSGPU_FILTER_MIN_WAVES) void hwlm_server_kernel(HsgpuScanArgs args, HsgpuServerCtl *ctl, HsgpuServerCtl *req,
                                                                                            unsigned long long idle_ticks, const uint4 *src_corpus,
                                                                                            const uint4 *src_off) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds != 0) __builtin_trap();
    /* the mailbox sits behind everything the body uses (runtime.hip sizes the launch: hsgpu_filter_lds_bytes + 192) */
    uint32_t bdim = __builtin_amdgcn_readfirstlane(blockDim.x); /* read ONCE (hwlm_filter_body): opaque to the compiler, so kept, not loaded again per request */
    asm volatile("" : "+s"(bdim));
    const uint32_t words = (uint32_t)(hsgpu_filter_words(args.t_flags, args.t_filter_log2) + ((args.t_flags & HSGPU_F_HAS_C) ? 2048u : 0u)) +
                           (uint32_t)((bdim >> 6) * sizeof(WaveLds) / 4);
    /* (an LDS pointer by TYPE and not volatile: a volatile generic pointer compiled to flat loads at system scope, each with a wait
     * of its own -- the request's nine words alone were ~0.6 us of serialised round trips behind the barrier; the barriers order
     * every access below) */
    typedef __attribute__((address_space(3))) uint32_t lds_rw_u32_t;
    lds_rw_u32_t *mail = (lds_rw_u32_t *)(lds + words); /* [0] seq, [1] command (0 go, 1 end), [2..7] total, nblocks, start, [8] debug; [16..31] the answer line; [32..43] the stage stamps */
    /* Everything around the barriers is WAVE-UNIFORM control flow (scalar branches on values made scalar with readfirstlane):
     * wavefront 0 polls as a whole. (The first version polled in `if (threadIdx.x == 0)`, a thread-divergent loop in front of
     * the barrier: the structurised code ran the loop's barriers a different number of times in wavefront 0 and in the others,
     * and the workgroup hung one barrier apart.) */
    const bool leader = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t last = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl->done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    uint32_t img_keep = 0; /* words at the head of the table image to load again (0: the whole image) */
    for (;;) {
        if (leader) {
            const unsigned long long t0 = wall_clock64();
            uint32_t cmd = 0, seq = last, line;
            for (;;) {
                /* ONE read per poll: the request line, a dword per lane (16 lanes, one 64-byte request): stop, the sequence number
                 * and the parameters, which the host wrote BEFORE the number -- a read that brings the new number brings them */
                line = __hip_atomic_load((uint32_t *)req + (lane & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (__builtin_amdgcn_readlane(line, 1)) { /* stop first: nothing outranks it */
                    cmd = 1;
                    break;
                }
                seq = __builtin_amdgcn_readlane(line, 0);
                if (seq != last) break;
                if (wall_clock64() - t0 > idle_ticks) {
                    cmd = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(HSGPU_SRV_POLL_SLEEP);
            }
            /* (HsgpuServerCtl: dwords 2 .. 7 = total, nblocks, start) */
            if (lane < 9) mail[lane] = lane == 0 ? seq : lane == 1 ? cmd : cmd ? 0u : line; /* ([8]: the request's debug flag) */
            if (lane == 31) mail[lane] = 0; /* the answer line's "records inside" flag (solo_place_lds) */
        }
        __syncthreads();
        const uint32_t seq = __builtin_amdgcn_readfirstlane(mail[0]), cmd = __builtin_amdgcn_readfirstlane(mail[1]);
        if (cmd) break;
        HsgpuScanArgs a = args;
        a.total = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mail[2]) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mail[3]) << 32;
        a.nblocks = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mail[4]) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mail[5]) << 32;
        a.start = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mail[6]) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mail[7]) << 32;
        a.n_hint = (a.total >> HSGPU_HINT_SHIFT) + 1;
        /* a batch of several blocks is copied to device memory first (below); ONE block is scanned where the host put it: its
         * tiles come over the bus beside the table image's staging, and nothing else of it is read (hints and block lookups
         * need no offsets for one block). Measured, 1 460-byte packets: in place 20.6 us per call, through the copy 24.3. */
        const bool stage = a.nblocks != 1;
        if (!stage) a.corpus = (const uint8_t *)src_corpus, a.off = (const uint64_t *)src_off;
        /* what the host wrote into the mapped area since the last request must not come out of this CU's caches */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, HSGPU_SRV_ACQUIRE_SCOPE);
        const bool dbg = __builtin_amdgcn_readfirstlane(mail[8]) != 0; /* stamps only when asked for: a clock read is a scalar memory operation the wavefront waits for */
        const unsigned long long t_seen = dbg ? wall_clock64() : 0;
        /* The batch comes over the bus ONCE: offsets and corpus are copied from the mapped area into device memory by the whole
         * workgroup (16 bytes per lane and pass, every request in flight at once), and the scan runs on the copy -- the block
         * hints' two search rounds, the tiles and every match's offsets were a bus round trip each when read in place. */
        if (stage) {
            const uint32_t n16 = (uint32_t)((a.total + 15) >> 4), o16 = (uint32_t)(((a.nblocks + 1) * 8 + 15) >> 4);
            for (uint32_t i = threadIdx.x; i < n16 + o16; i += bdim) {
                if (i < n16) ((uint4 *)a.corpus)[i] = src_corpus[i];
                else ((uint4 *)a.off)[i - n16] = src_off[i - n16];
            }
            __syncthreads(); /* (the same compute unit wrote it: the barrier's workgroup-scope release / acquire is enough) */
        }
        const unsigned long long t_copied = dbg ? wall_clock64() : 0;
        a.img_keep_words = img_keep;
        a.srv_inline_at = words + 16;
        /* the request's stages: start, image staged, wavefront 0 confirmed, placed (hsgpu_debug_server_stamps) -- stamped into LDS
         * (a generic pointer) and sent with the answer: a store to host memory in the body is a bus write that the next barrier's
         * wait for outstanding memory operations sits behind */
        a.wg_stamps = dbg ? (unsigned long long *)(lds + words + 32) : nullptr;
        const bool image_used = hwlm_filter_body<HAS_A, HAS_B, HAS_C, REPL, K2, S2, BLIND, true, PAIR, WIDE>(a, lds, bdim, 1u);
        img_keep = image_used ? SOLO_LDS_WORDS : 1u; /* the table image stays in LDS between requests; solo_tail's scratch was its head (1: nothing to load) */
        __syncthreads();
        const uint32_t inlined = __builtin_amdgcn_readfirstlane(mail[31]);
        if (!inlined) {
            /* records and count went to the mapped area: every wavefront's out at SYSTEM scope before the answer line. (A
             * workgroup-scope release in front of a relaxed done word was measured: the done word, another address and so another
             * L2 channel, overtook the count -- the host read the count it had put there itself and sent every call down the
             * launch path.) */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __syncthreads();
        }
        if (leader) {
            /* the answer: records (from LDS, <= 3), count (~0: in the mapped area), the request's stamps (100 MHz ticks: copy,
             * body) and the sequence number, the line's LAST dword: four lanes x 16 bytes, one store instruction, one 64-byte
             * write -- the 2.4 us of write-back and wait that a release in front of a separate done word cost are gone */
            const unsigned long long t_end = dbg ? wall_clock64() : 0;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v;
            if (lane < 3) {
                v[0] = mail[16 + 4 * lane], v[1] = mail[17 + 4 * lane], v[2] = mail[18 + 4 * lane], v[3] = mail[19 + 4 * lane];
            } else {
                v[0] = inlined ? mail[28] : ~0u, v[1] = (uint32_t)(t_copied - t_seen), v[2] = (uint32_t)(t_end - t_copied), v[3] = seq;
            }
            if (lane >= 4 && lane < 7) v[0] = mail[32 + 4 * (lane - 4)], v[1] = mail[33 + 4 * (lane - 4)], v[2] = mail[34 + 4 * (lane - 4)], v[3] = mail[35 + 4 * (lane - 4)];
            if (lane < (dbg ? 7u : 4u)) { /* (lanes 4 .. 6: the six stage stamps, a line of their own -- debug, ordered with nothing) */
                u32x4 *dst = lane < 4 ? (u32x4 *)ctl->done_rec + lane : (u32x4 *)ctl->stamps + (lane - 4);
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
            }
        }
        last = seq;
    }
    if (leader) __hip_atomic_store(&ctl->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* ---- phase 3 helpers: the records in delivery order -------------------------------------------
 * hwlmExec delivers callbacks in non-decreasing `end` (src/hwlm/hwlm.h:101-118) and Rose relies on it
 * (src/rose/match.c:396-476); a batch delivers block by block. The output of a scan is therefore sorted by
 * (block, end, literal index). Every filter wavefront streams ONE contiguous share of the corpus (see the
 * tile loop), so the staging regions fed from its candidates -- HSGPU_CONFIRM_SPLIT consecutive region numbers --
 * hold exactly the records of that share, and the shares follow each other in region order: delivery order is one
 * small sort per share plus knowing how many records the shares in front of it hold. No atomic per record and no
 * global sort: a share holds a few hundred records on ordinary input. */
constexpr uint32_t SORT_LDS = 1024; /* largest share sorted in LDS (16 KiB of records) */
constexpr uint32_t SORT_THREADS = 256;


/* normalized bitonic network (every comparator ascending) over n records at x; positions from n up to the
 * next power of two stand for records greater than all: a comparator that touches one is a no-op. One workgroup. */
template <class PTR>
__device__ __forceinline__ void bitonic_sort(PTR x, uint32_t n) {
    uint64_t P = 2;
    while (P < n) P <<= 1;
    for (uint64_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = (uint32_t)(k >> 1); j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t l = (j == (uint32_t)(k >> 1)) ? i ^ (uint32_t)(k - 1) : i ^ j;
                if (l > i && l < n) {
                    const uint4 a = x[i], b = x[l];
                    if (rec_less(b, a)) x[i] = b, x[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

/* Workgroup-convergent: the records of the staging regions [first, last) (one share: <= 64 regions) gathered, sorted by
 * (block, end, literal index) and written to out[base ...). buf: SORT_LDS records of LDS. A share one of whose regions
 * lost records (its fill counter ran past its capacity: the scan reports "again") is left alone. */
__device__ __forceinline__ void sort_share(const HsgpuScanArgs &args, uint4 *buf, uint32_t first, uint32_t last,
                                           unsigned long long base) {
    const uint32_t NT = blockDim.x, tid = threadIdx.x, lane = tid & 63;
    uint4 *out = (uint4 *)args.out;
    const uint2 *counts = (const uint2 *)args.rec_counts;
    /* the fills of all regions of the share at once (lane r = region first + r) and where each region's records
     * start inside the share */
    const uint32_t nreg = last - first; /* <= 64 */
    const uint2 my = lane < nreg ? counts[first + lane] : make_uint2(0, 0);
    /* dense scans with run tables: a region's count includes the records its run descriptors stand for (they never fit; a region
     * that lost staged records has raised the scan's flag, and nobody gets here) */
    const bool runs = args.fold && args.run_tab != nullptr;
    if (!runs && __ballot((unsigned long long)my.x + my.y > args.rec_cap)) return; /* (every wavefront holds the same fills: uniform) */
    uint32_t incl = my.x + my.y;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += v;
    }
    const uint32_t n = __shfl(incl, 63);
    const uint32_t my_at = lane < nreg ? incl - (my.x + my.y) : n;
    if (!n) return;
    /* folded pipeline: the regions are consecutive sorted runs of the corpus -- gathered straight into place */
    const bool in_lds = !args.fold && n <= SORT_LDS;
    if (runs) {
        /* the share's run tables into LDS, once (buf is free in the folded pipeline): read from memory per RECORD -- the count, then
         * descriptor after descriptor, then the record: a chain of dependent round trips in front of every store -- the flood
         * corpus' gather took 0.21 ms for 537 MB, twice what the writes take (uniform: every wavefront holds the same fills) */
        for (uint32_t k = tid; k < nreg * HSGPU_RUN_STRIDE; k += NT) buf[k] = args.run_tab[(uint64_t)first * HSGPU_RUN_STRIDE + k];
        __syncthreads();
    }
    /* gather: every lane walks the share's records: which region, which slot (front records, then the ones
     * spilled to the back) */
    for (uint32_t i = tid; i < n + ((64 - n % 64) % 64); i += NT) { /* whole wavefronts: shuffles below */
        uint32_t r = 0;
        for (uint32_t k = 1; k < nreg; k++) r += __shfl(my_at, k) <= i ? 1u : 0u; /* the last region starting at or before i */
        const uint32_t f = __shfl(my.x, r), at = __shfl(my_at, r), j = i - at;
        if (i < n) {
            const uint4 *region = args.rec_stage + (uint64_t)(first + r) * args.rec_cap;
            if (runs) {
                /* record j of the region: staged at j less the records that the runs in front of it stand for -- or one of those: the
                 * run's lookup 0 has the record, `end` moves on by the run's step per lookup (position-major: delivery order) */
                const uint4 *rt = buf + r * HSGPU_RUN_STRIDE;
                const uint32_t nr = min(rt[0].x, (uint32_t)HSGPU_RUN_MAX);
                uint32_t at_staged = j, skipped = 0, add = 0;
                for (uint32_t k = 1; k <= nr; k++) {
                    const uint4 d = rt[k]; /* {first record, records per lookup, further lookups, step} */
                    const uint32_t ls = d.x + skipped + d.y; /* the first record of the region that is not staged */
                    if (j < ls) break;
                    const uint32_t span = d.y * d.z, v = j - ls;
                    if (v < span) {
                        const uint32_t qv = v / d.y;
                        at_staged = d.x + (v - qv * d.y) + skipped; /* (skipped comes off below) */
                        add = d.w * (qv + 1u);
                        break;
                    }
                    skipped += span;
                }
                uint4 rec = region[min(at_staged - skipped, args.rec_cap - 1u)];
                rec.y += add;
                out[base + i] = rec;
            } else {
                const uint4 rec = j < f ? region[j] : region[args.rec_cap - 1 - (j - f)];
                if (in_lds) buf[i] = rec;
                else out[base + i] = rec;
            }
        }
    }
    if (args.fold) return; /* (uniform) */
    __syncthreads();
    if (n <= 64) {
        /* rank by counting: one record per lane of the first wavefront, every lane walks the share
         * (all lanes read the same LDS address: a broadcast); no barriers, no compare-exchange chains */
        if (tid < n) {
            const uint4 mine = buf[tid];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < n; q++) { /* equal keys cannot occur; if they did, the gather order keeps ranks distinct */
                const uint4 o = buf[q];
                rank += (rec_less(o, mine) || (q < tid && !rec_less(mine, o))) ? 1u : 0u;
            }
            out[base + rank] = mine;
        }
    } else if (in_lds) {
        bitonic_sort(buf, n);
        for (uint32_t i = tid; i < n; i += NT) out[base + i] = buf[i];
    } else {
        /* a dense share (the reference's flood case, src/fdr/flood_runtime.h:86-335: thousands of records from a
         * few KiB of corpus): tiles of SORT_LDS records sorted in LDS, then merge passes between the output and
         * the share's own staging regions (free once gathered; together at least n records long), every thread
         * merging MERGE_SEG outputs from the split point its diagonal gives (merge path). The bitonic network
         * run in place in global memory took 28 ms for 8 192 records. */
        uint4 *a = out + base, *b = args.rec_stage + (uint64_t)first * args.rec_cap;
        for (uint32_t t0 = 0; t0 < n; t0 += SORT_LDS) {
            const uint32_t cnt = min(SORT_LDS, n - t0);
            for (uint32_t i = tid; i < cnt; i += NT) buf[i] = a[t0 + i];
            __syncthreads();
            bitonic_sort(buf, cnt);
            for (uint32_t i = tid; i < cnt; i += NT) a[t0 + i] = buf[i];
            __threadfence_block();
            __syncthreads();
        }
        constexpr uint32_t MERGE_SEG = 8;
        for (uint32_t w = SORT_LDS; w < n; w <<= 1) {
            for (uint32_t o = tid * MERGE_SEG; o < n; o += NT * MERGE_SEG) {
                const uint32_t pair = o / (2 * w) * (2 * w); /* this output segment lies in the merge of runs at pair */
                const uint32_t la = min(w, n - pair), lb = min(w, n - min(n, pair + w));
                const uint4 *ra = a + pair, *rb = a + pair + la;
                const uint32_t d = o - pair; /* diagonal: d outputs come before this segment */
                /* i elements of run A and d - i of run B precede: the smallest i with A[i] > B[d - i - 1] fails */
                uint32_t lo = d > lb ? d - lb : 0, hi = min(d, la);
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (rec_less(rb[d - mid - 1], ra[mid])) hi = mid; /* B's element first: take fewer of A */
                    else lo = mid + 1;
                }
                uint32_t i = lo, j = d - lo;
                const uint32_t stop = min(o + MERGE_SEG, min(n, pair + la + lb));
                for (uint32_t k = o; k < stop; k++) {
                    const bool take_b = i >= la || (j < lb && rec_less(rb[j], ra[i]));
                    b[k] = take_b ? rb[j++] : ra[i++];
                }
            }
            __threadfence_block();
            __syncthreads();
            uint4 *t = a;
            a = b;
            b = t;
        }
        if (a != out + base) { /* an odd number of passes left the result in the staging area */
            for (uint32_t i = tid; i < n; i += NT) out[base + i] = a[i];
        }
    }
}


/* ---- solo scans: the placement inside the (fused) scan kernel -------------------------------------------------
 * hsbench block mode is one hs_scan per block (tools/hsbench/engine_hyperscan.cpp:132-145) and the reference serves short blocks
 * from dedicated small matchers (src/rose/block.c:382-391, src/runtime.c:401-413); three launches per scan -- filter, confirm,
 * placement -- are ~20 us of fixed cost whatever the batch holds (also.batch_sweep, round 4). A batch of up to ~1 MiB takes ONE:
 * the fused kernel (filter + confirm inline, one staging region per wavefront = one contiguous piece of the corpus), and the
 * workgroup that finishes LAST does what record_sort_kernel does for the others. "Last" is a ticket: every workgroup releases
 * its stores at agent scope (L2 write-back: the XCDs' L2s are not coherent with each other inside a kernel) and takes a number;
 * whoever draws grid - 1 acquires and sees everybody's regions. Nobody waits for anybody.
 *   placement   fills -> exclusive prefix (<= 1024 regions, one wavefront); a region of <= 64 records is ranked by counting by
 *               ONE wavefront (its records in that wavefront's 1 KiB of LDS, every lane walks them: broadcast reads), 16
 *               regions at a time; larger regions go through sort_share one after the other (LDS bitonic, merge path beyond)
 *   count       as record_sort_kernel: the exact total; cap + 1 ("again") when a region lost records
 *   control     the scan's own block (fills, sums, ticket) is zeroed by the same workgroup: the next solo scan finds it clean
 * LDS: the filter image is dead by then; its first 28 KiB serve as sort buffer, prefix array and list of large regions. */
constexpr uint32_t SOLO_MAX_REGIONS = 1024;
__device__ __forceinline__ void solo_tail(const HsgpuScanArgs &args, uint32_t *lds, uint32_t n_reg) {
    const uint32_t NT = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* every wavefront releases what it staged and published at agent scope BEFORE the barrier (advisor, round 5: the barrier's own
     * release is workgroup scope; that one thread's agent-scope ticket made the others' stores visible to another XCD too was a
     * property of this compiler and this chip, not of the memory model) */
    if (gridDim.x != 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads(); /* every wavefront of this workgroup has published */
    uint32_t *flagw = lds; /* [0] = last, [1] = total (low), [2] = total (high), [3] = flag, [4] = number of large regions, [5] = a region lost records */
    if (tid == 0) {
        /* (one workgroup -- the small-batch server, a batch of one tile: it is the last by construction, and what it reads was
         * written on this compute unit: no ticket, no agent-scope release / acquire, i.e. no write-back and no invalidation of L2) */
        const uint32_t ticket = gridDim.x == 1 ? 0u : __hip_atomic_fetch_add(args.solo_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        flagw[0] = ticket == gridDim.x - 1;
        flagw[5] = 0;
    }
    __syncthreads();
    if (!flagw[0]) return; /* (uniform) */
    if (gridDim.x != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* every wavefront, before it reads what other workgroups wrote */
    uint4 *buf = (uint4 *)(lds + 64);                       /* SORT_LDS records: 16 KiB */
    uint32_t *start = lds + 64 + SORT_LDS * 4;              /* [SOLO_MAX_REGIONS + 1] fills, then exclusive prefix */
    uint32_t *large = start + SOLO_MAX_REGIONS + 64;        /* regions of more than 64 records */
    static_assert(64 + SORT_LDS * 4 + 2 * (SOLO_MAX_REGIONS + 64) <= SOLO_LDS_WORDS, "solo_tail's scratch is the head of the filter image that the server loads again");
    const uint2 *counts = (const uint2 *)args.rec_counts;
    uint32_t over = 0;
    for (uint32_t i = tid; i < SOLO_MAX_REGIONS; i += NT) {
        uint32_t f = 0;
        if (i < n_reg) {
            const uint2 c = counts[i];
            const unsigned long long fill = (unsigned long long)c.x + c.y;
            over |= fill > args.rec_cap;
            f = (uint32_t)min(fill, 0xffffffffull);
        }
        start[i] = f;
    }
    /* (not __syncthreads_or: the device library's workgroup reduction owns a static LDS word, and a kernel with static LDS no
     * longer has its dynamic LDS at address 0, which every filter kernel's absolute LDS addressing relies on -- they trap) */
    if (over) flagw[5] = 1;
    __syncthreads();
    const uint32_t any_over = flagw[5];
    if (wave == 0) { /* exclusive prefix over 1024 fills: 16 per lane */
        uint32_t v[16];
        unsigned long long sum = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = start[lane * 16 + k], sum += v[k];
        unsigned long long incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += o;
        }
        unsigned long long at = incl - sum;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            start[lane * 16 + k] = (uint32_t)min(at, 0xffffffffull);
            at += v[k];
        }
        if (lane == 63) {
            const unsigned long long all = incl;
            const unsigned long long flag = args.rec_super[HSGPU_SUPER_FLAGS] | any_over;
            flagw[1] = (uint32_t)all, flagw[2] = (uint32_t)(all >> 32), flagw[3] = (flag != 0) || all > args.cap, flagw[4] = 0;
            *args.count = (flag && all <= args.cap) ? args.cap + 1 : all; /* as record_sort_kernel: never a value <= cap for an incomplete output */
            start[SOLO_MAX_REGIONS] = (uint32_t)min(all, 0xffffffffull);
        }
    }
    __syncthreads();
    if (!flagw[3]) {
        uint4 *out = (uint4 *)args.out;
        /* regions of up to 64 records (nearly all): ONE pass over the record slots, a thread per record -- its region by bisection
         * of the prefix array in LDS, its rank by counting against the region's other records (a handful; read from L2), its place
         * start[region] + rank. (A wavefront per region, regions one after the other, was 32 dependent rounds for the 512 regions of
         * a 1 MiB scan: 30 of its 50 us.) Larger regions are listed for sort_share. */
        const uint32_t all = flagw[1]; /* (complete: all <= cap < 2^32) */
        for (uint32_t i0 = 0; i0 < all; i0 += NT) {
            const uint32_t i = i0 + tid;
            if (i < all) {
                uint32_t lo_r = 0, hi_r = n_reg; /* the last region with start <= i */
                while (hi_r - lo_r > 1) {
                    const uint32_t mid = (lo_r + hi_r) >> 1;
                    if (start[mid] <= i) lo_r = mid;
                    else hi_r = mid;
                }
                const uint32_t r = lo_r, j = i - start[r];
                const uint2 c = counts[r];
                const uint32_t n = c.x + c.y;
                if (n > 64) {
                    if (j == 0) large[atomicAdd(&flagw[4], 1u)] = r;
                } else {
                    const uint4 *region = args.rec_stage + (uint64_t)r * args.rec_cap;
                    const uint4 rec = j < c.x ? region[j] : region[args.rec_cap - 1 - (j - c.x)];
                    uint32_t rank = 0;
                    for (uint32_t q = 0; q < n; q++) {
                        const uint4 o = q < c.x ? region[q] : region[args.rec_cap - 1 - (q - c.x)];
                        rank += (rec_less(o, rec) || (q < j && !rec_less(rec, o))) ? 1u : 0u;
                    }
                    out[(uint64_t)start[r] + rank] = rec;
                }
            }
        }
        __syncthreads();
        const uint32_t n_large = flagw[4];
        for (uint32_t k = 0; k < n_large; k++) { /* rare: dense pieces of the corpus */
            const uint32_t r = large[k];
            sort_share(args, buf, r, r + 1, start[r]);
            __syncthreads();
        }
    }
    /* the control block back to zero (nobody else is left); the timing slots */
    for (uint32_t i = tid; i < args.solo_ctl_words; i += NT) args.rec_counts[i] = 0;
    if (tid == 0 && args.tstamp) {
        const unsigned long long now = wall_clock64();
        args.tstamp[1] = now, args.tstamp[2] = now, args.tstamp[3] = now;
        if (args.tstamp_next) args.tstamp_next[0] = ~0ull, args.tstamp_next[1] = 0, args.tstamp_next[2] = 0, args.tstamp_next[3] = 0;
    }
}

/* Last kernel of a scan, every workgroup. The control words of THIS scan are read across workgroups (fills, sums,
 * counts), so nobody can zero them here; instead every workgroup zeroes its slice of the OTHER control block, the one
 * the previous scan used and the next scan will use: no memset in front of any scan. Plus the cumulative statistics
 * for hsgpu_scratch_get_stats (1024 counters per workgroup: only a handful of workgroups touch the statistics word). */
__device__ __forceinline__ void scan_epilogue(const HsgpuScanArgs &args) {
    const uint32_t NT = blockDim.x, tid = threadIdx.x, lane = tid & 63;
    {
        uint4 *other = (uint4 *)args.ctl_other;
        const uint32_t n4 = args.ctl_other_words >> 2, per = (n4 + gridDim.x - 1) / gridDim.x;
        const uint32_t lo = blockIdx.x * per, hi = min(n4, lo + per);
        for (uint32_t i = lo + tid; i < hi; i += NT) other[i] = make_uint4(0, 0, 0, 0);
    }
    if (!args.cand_counts || blockIdx.x * 1024u > args.cand_waves) return; /* (uniform per workgroup) */
    uint32_t v = 0, o = 0;
    for (uint32_t i = blockIdx.x * 1024 + tid; i < min(args.cand_waves + 1, (blockIdx.x + 1) * 1024); i += NT) {
        const uint32_t c = args.cand_counts[i];
        if (i == args.cand_waves) o = c;
        else v += c;
    }
    unsigned long long vs = v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vs += __shfl_xor(vs, d);
    const unsigned long long any_o = __ballot(o != 0);
    if (lane == 0) {
        if (vs) atomicAdd(&args.stats[0], vs);
        if (any_o) atomicAdd(&args.stats[1], 1ull);
    }
}

/* ---- phase 2: confirm, in order -- one resident grid of worker wavefronts ------------------------------------------
 * The grid is what the device holds at once (workgroups per CU x CUs); the 8 KiB key gate is staged once per workgroup, behind
 * the kernel's only barrier. Every wavefront is a WORKER with conf_k consecutive PARTS of the corpus -- a share (= one filter
 * wavefront's candidate region) cut into conf_q pieces of whole batches of 128 entries; runtime.hip picks the two numbers so
 * that the parts go round the workers evenly (4 096 shares x 2 = 1 per worker on the fast step's 8 192; x 3 = 2 per worker on the
 * general step's 6 144). Its records go to ONE staging region -- a region per PART where a worker's parts are not neighbours in
 * the corpus: dense scans (conf_spread: the parts of a worker spread row by row) and the two halves of a share on workers of
 * mirrored dispatch ranks (conf_skew); both at the head of the worker's loop below.
 *   in order  (args.fold; the default) matches wait in the wavefront's LDS queue until a sync point -- more than SYNC_AT queued, or
 *             the part done: the entries with candidate bits left are confirmed first (so that no earlier position is still
 *             pending), then the queue is sorted on 64-bit keys {position, literal} by a bitonic network across the lanes,
 *             resolved 64 at a time IN THAT ORDER and appended to the region. A region is therefore sorted by (block, end,
 *             literal) as it is written, and the concatenation of all regions is the delivery order: record_sort_kernel
 *             behind this kernel only gathers them into place. (A queue that overflows between two sync points resolves in
 *             place, out of order; the scan then reports "again" and sends the scratch to dense mode.)
 *   publish   the wavefront adds its fill to the sum of its region's "super" (2^super_shift consecutive regions, at most 256
 *             supers): one relaxed atomic per worker; record_sort_kernel finds every group's place from <= 256 sums + the fills
 *             in front of it inside its own super.
 * History, all measured (profiles/r04_tail_*.txt, r04_confirm_fixed_cost.txt): round 3 launched one short-lived wavefront per
 * part of a share. Round 4 first made the workgroups persistent with shares handed out by ticket, quarters per wavefront, and
 * the placement done by the wavefronts themselves, deferred by one share so that nobody polled: equal to confirm + sort at
 * 1 GiB, better at 8 GiB -- but a line fitted through 16 MiB .. 1 GiB showed 93 us of the 200 us stage there before the first
 * candidate (2.67 rounds of ticket, barrier, final drain, publish, placement, copy per workgroup). Static parts per worker
 * remove the rounds; placing in the same kernel then means every worker polling the sums until everybody in front has
 * published (0.30 ms: 6 144 wavefronts on a few hundred words), so the placement went back to a kernel of its own, which for
 * sorted regions is a plain gather: 16 MiB 118 -> 49 us, 1 GiB 0.512 -> 0.491 ms, teddy64 0.320 -> 0.281 ms. Round 5: the shape per
 * instantiation (confirm_shape: one entry per lane at eight wavefronts per SIMD for the fast step), the tail measured worker by
 * worker (HSGPU_CONFIRM_STAMPS, profiles/r05_confirm_workers.txt): stage 0.1665 -> 0.1445 ms. */
constexpr uint32_t DENSE_AT = 48;    /* dense scans: candidate positions in a batch of 128 entries from which the batch goes position by position */
constexpr uint32_t DENSE_POS = 128 * 16; /* ... the positions of a batch */
#ifndef HSGPU_SYNC_AT
#define HSGPU_SYNC_AT 48 /* tuning builds */
#endif
constexpr uint32_t SYNC_AT = HSGPU_SYNC_AT;          /* folded: queued matches that end a group of batches (the queue holds MQ_CAP = 128) */
static_assert(CONFIRM_THREADS / 64 == HSGPU_CONFIRM_SPLIT, "one wavefront per part of a share");

/* A queued match IS its sort key: {position (< 2^36) << 24 | literal index (< 2^24)} -- delivery order is (block, end,
 * literal), and (block, end) grows with the position. */
__device__ __forceinline__ uint64_t mq_key(const uint2 it) { return (uint64_t)it.y << 32 | it.x; }
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, uint32_t addr) { /* the value of lane (lane ^ j): addr = (lane ^ j) * 4 */
    return (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)(uint32_t)(v >> 32)) << 32 |
           (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)(uint32_t)v);
}

/* Convergent, folded pipeline: the queued matches (wl->cand[0 .. nmq)) sorted, resolved in order, appended to the front of the
 * region. The sort is a bitonic network ACROSS THE LANES on the 64-bit keys (one key per lane, two above 64 queued; keys that
 * are not there are all ones and sort to the end): 21 compare-exchange stages of two ds_bpermute, one compare and two
 * selects -- ~130 vector instructions. (Ranking by counting, every lane walking the queue, was ~20 instructions per queued
 * match and lane: 1 000 for the usual 50 matches, as much again as the confirm steps that found them; the stage took
 * 0.19 ms instead of 0.15: profiles/r04_tail_ab.txt.) */
template <bool CACHE = false, class TABLES>
__device__ __forceinline__ void drain_matches_sorted(TABLES &t, uint32_t lane) {
    uint32_t n = __hip_atomic_load(&t.wl->nmq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    n = min(__builtin_amdgcn_readfirstlane(n), MQ_CAP);
    if (n == 0) return;
    uint64_t k0 = lane < n ? mq_key(t.wl->cand[lane]) : ~0ull;
    if (n <= 64) {
        if (n > 1) {
#pragma unroll
            for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    const uint64_t o = lane_xor64(k0, (lane ^ j) << 2);
                    const bool up = (lane & k) == 0, low = (lane & j) == 0; /* ascending block; this lane keeps the smaller one */
                    const bool take = (o < k0) == (up == low);
                    k0 = take ? o : k0;
                }
            }
        }
        /* (a lane without a match resolves the smallest key again -- its all-ones filler names no literal and no position) */
        /* (v_readlane of lane 0, not readfirstlane: the compiler sank one half of a readfirstlane into the `lane >= n` side of
         * the select, where the first active lane is a filler) */
        const uint64_t safe = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k0 >> 32), 0) << 32 |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k0, 0);
        if (lane >= n) k0 = safe;
        resolve_queued<CACHE>(t, lane, lane < n, make_uint2((uint32_t)k0, (uint32_t)(k0 >> 32)));
    } else { /* 65 .. 128 queued: two keys per lane, element i = lane + 64 r */
        uint64_t k1 = lane + 64 < n ? mq_key(t.wl->cand[lane + 64]) : ~0ull;
#pragma unroll
        for (uint32_t k = 2; k <= 128; k <<= 1) {
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                if (j == 64) { /* the partner is the lane's own other key; k = 128: ascending */
                    const uint64_t lo = min(k0, k1), hi = max(k0, k1);
                    k0 = lo, k1 = hi;
                } else {
                    const uint64_t o0 = lane_xor64(k0, (lane ^ j) << 2), o1 = lane_xor64(k1, (lane ^ j) << 2);
                    const bool low = (lane & j) == 0;
                    const bool up0 = (lane & k) == 0, up1 = ((lane + 64) & k) == 0;
                    k0 = ((o0 < k0) == (up0 == low)) ? o0 : k0;
                    k1 = ((o1 < k1) == (up1 == low)) ? o1 : k1;
                }
            }
        }
        resolve_queued<CACHE>(t, lane, true, make_uint2((uint32_t)k0, (uint32_t)(k0 >> 32)));
        const uint64_t safe = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k0 >> 32), 0) << 32 |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k0, 0);
        if (lane + 64 >= n) k1 = safe;
        resolve_queued<CACHE>(t, lane, lane + 64 < n, make_uint2((uint32_t)k1, (uint32_t)(k1 >> 32)));
    }
    if (lane == 0) __hip_atomic_store(&t.wl->nmq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

#ifndef HSGPU_CONFIRM_SGPR_ATTR
#define HSGPU_CONFIRM_SGPR_ATTR
#endif
template <bool HAS_A, bool HAS_B, bool HAS_C, bool S2, bool PAIR = false, bool DENSE = false>
__global__ __launch_bounds__(CONFIRM_THREADS)
/* Registers capped for the wavefronts per SIMD that the kernel's LDS admits (confirm_shape: eight workgroups per CU for the fast
 * step, six otherwise). The scheduler has no occupancy target of its own here and takes 94-98 registers for the general worker's
 * loop -- five wavefronts; capped, the 4-byte-key variants keep everything in registers and the others spill a few dwords outside
 * the confirm step (tools/kernel_regs.sh). */
__attribute__((amdgpu_waves_per_eu(confirm_shape<HAS_A, HAS_C, S2, PAIR, DENSE>::WAVES, 8))) HSGPU_CONFIRM_SGPR_ATTR
void hwlm_confirm_kernel(HsgpuScanArgs args) {
    using shape = confirm_shape<HAS_A, HAS_C, S2, PAIR, DENSE>;
    constexpr bool FAST = shape::FAST;
    constexpr int CU = shape::U; /* entries per lane and step */
    constexpr uint32_t RQ_CAP = shape::RQ_CAP;
    constexpr uint32_t W = CONFIRM_THREADS / 64;
    __shared__ WaveLds wave_lds[W];
    __shared__ uint2 rest_q[W][RQ_CAP];
    /* HSGPU_F_GATE: 64 Kbit, "is there an exact-table key with this hash at all"; HSGPU_F_BLOOM (opt-in, stride-1 tables): 96 Kbit,
     * "is there a literal whose full key this window has" -- one or the other, in dynamic LDS (runtime.hip asks for 8 or 12 KiB) */
    extern __shared__ __attribute__((aligned(16))) uint4 key_gate[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fold = args.fold != 0;
    constexpr bool dense = DENSE; /* (an instantiation of its own, scan_inst_dense.hip: the position-parallel path costs the ordinary kernel registers it does not have) */
    const uint32_t n_shares = args.cand_waves;
    if (args.cand_counts[n_shares]) return; /* candidate overflow: nothing is confirmed, the total is unknown; record_sort_kernel reports ("again") */
    const bool gated = !PAIR && (args.t_flags & HSGPU_F_GATE);
    const bool bloomed = !PAIR && !S2 && (args.t_flags & HSGPU_F_BLOOM);
    if (gated || bloomed) { /* once per workgroup */
        const uint4 *src = (const uint4 *)(args.blob + args.t_off_c2bits);
        for (uint32_t i = tid; i < (bloomed ? HSGPU_BLOOM_WORDS / 4 : 512u); i += CONFIRM_THREADS) key_gate[i] = src[i];
    }
    Tables t;
    init_tables(t, args);
    if (gated) t.key_gate = (const uint32_t *)key_gate;
    if (bloomed) t.bloom = (const uint32_t *)key_gate;
    t.rec_cap = args.rec_cap;
    uint2 *rq = rest_q[wave];
    FastRs rs;
    if (FAST) {
        rs.ht_a = __builtin_amdgcn_make_buffer_rsrc((void *)t.ht_a, 0, (int)(16u << min(t.ht_a_log2, 26u)), 0x00020000);
        rs.ht_b = __builtin_amdgcn_make_buffer_rsrc((void *)t.ht_b, 0, (int)(16u << min(t.ht_b_log2, 26u)), 0x00020000);
        rs.lits = __builtin_amdgcn_make_buffer_rsrc((void *)t.lits, 0, (int)0x7ffffff0, 0x00020000);
    }

    __syncthreads(); /* the gate is staged: the only barrier of the kernel */
    /* Every wavefront is a WORKER with K consecutive parts of the corpus: share r = one filter wavefront's candidates, cut into Q
     * parts of whole batches (runtime.hip picks Q and K so that the parts go round the workers the device holds: 4 096 shares x 2
     * = 1 per worker on 8 192; x 3 = 2 per worker on 6 144). No tickets, no barrier per share, ONE publish per worker, and nobody waits for anybody: with shares
     * handed out by ticket to workgroups (quarters per wavefront, a barrier, a publish and a placement per share, 2.67 rounds of
     * them) the stage cost 93 us before the first candidate -- 96 us for 16 MiB, 200 for 1 GiB (profiles/r04_confirm_fixed_cost.txt). */
    const uint32_t worker = blockIdx.x * W + wave;
    const uint32_t Q = args.conf_q, K = args.conf_k, n_parts = n_shares * Q;
    init_wave_lds(t, &wave_lds[wave], lane);
    /* The instruction arbiter prefers the oldest wavefront of a SIMD, and the workgroups of this resident grid reach a CU in index
     * order: with equal work the workers of the first 256 workgroups finish first and those of the last 256 last (two entries per
     * lane, six per SIMD: 137 .. 164 us; one entry, eight per SIMD: 97 .. 121 us; the kernel ends 20 % behind its mean worker:
     * profiles/r05_confirm_workers.txt). Tuning builds, HSGPU_CONFIRM_ROTPRIO=1: every step a worker takes the next of the four
     * priorities, starting from its workgroup's rank on its CU -- the spread narrows (p10 .. p90 of the lifetimes 104 .. 120 us
     * instead of 96 .. 123) and the stage gains 1.5 % (0.1475 vs 0.150 ms); "through all eight ranks" and "the youngest first"
     * gained nothing. Not in the product build: 1.5 % does not pay for a dependence on the dispatch order. */
#ifndef HSGPU_CONFIRM_ROTPRIO
#define HSGPU_CONFIRM_ROTPRIO 0
#endif
#if HSGPU_CONFIRM_ROTPRIO
    const uint32_t prio_rank = __builtin_amdgcn_readfirstlane(blockIdx.x / max(args.conf_cus, 1u));
    uint32_t prio_step = 0;
#endif
#ifndef HSGPU_CONFIRM_STAMPS
#define HSGPU_CONFIRM_STAMPS 0 /* tuning builds: 1 = per-worker stamps and step counts (hsgpu_scratch_get_conf_stamps) */
#endif
#if HSGPU_CONFIRM_STAMPS
    uint32_t st_fresh = 0, st_rest = 0, st_drains = 0, st_entries = 0;
    if (args.conf_stamps && lane == 0) args.conf_stamps[4 * worker] = wall_clock64();
#define HSGPU_ST(x) x
#else
#define HSGPU_ST(x)
#endif
    t.rec_region = args.rec_stage + (uint64_t)worker * args.rec_cap;
    /* Dense scans (args.conf_spread, runtime.hip): a region per PART, and a worker's K parts are not neighbours but spread over the
     * corpus -- its j-th part is part j x workers + (worker + 2 731 j) mod workers: consecutive parts go to consecutive workers, so
     * a dense run of L parts gives every worker L / workers of them, give or take one (a multiplicative shuffle of the parts was
     * measured first: Poisson, the busiest worker had twice the mean); the rotation per row keeps a corpus whose dense runs
     * repeat with a period of `workers` parts from landing on the same workers every time. Dense input
     * comes in runs (the reference's flood case: whole blocks of one byte), a dense part is hundreds of times the work of a quiet
     * one, and with K consecutive parts per worker the bench's flood corpus kept a quarter of the workers busy for 1.65 ms while the
     * others had finished after 0.1 (profiles/r05_flood.txt). A dense step ends in a sorted drain anyway, so small parts cost
     * nothing here (on ordinary input every part end is a sync point: the 10 000-literal stage is 20 % slower with four parts
     * per worker than with one). */
    const bool spread = DENSE && !PAIR && args.conf_spread;
    const bool runs = spread && args.run_tab != nullptr; /* a run of one byte value: one confirm and a descriptor (below) */
    const uint32_t n_workers = gridDim.x * W;
    /* Ordinary scans with one part per worker and two parts per share (args.conf_skew, runtime.hip: the headline's geometry): the
     * instruction arbiter prefers the oldest wavefront of a SIMD and the resident workgroups reach a CU in index order, so with
     * equal work the workers of the first-dispatched workgroups finish 20 % before those of the last (97 .. 121 us by the rank of a
     * workgroup on its CU: profiles/r05_confirm_workers.txt) and the kernel's tail runs on ever fewer wavefronts. The two halves of a
     * share therefore go to workers of MIRRORED ranks on the same CU slot -- rank k and rank R - 1 - k of the R workgroups per CU --,
     * the older one takes the first conf_skew / 2^16 more than half of the share's batches, scaled by how far apart the two ranks
     * are, and a record region belongs to a part, not to a worker. */
    const bool skew = !DENSE && !PAIR && args.conf_skew && K == 1 && Q == 2;
    uint32_t skew_part = 0, skew_num = 0; /* the part of this worker; the older worker's share of the batches in 1 / 2^16 */
    if (skew) {
        const uint32_t cus = max(args.conf_cus, 1u), ranks = gridDim.x / cus, wg = blockIdx.x, rank = wg / cus, slot_cu = wg - rank * cus;
        const bool old = rank < ranks / 2;
        const uint32_t low = old ? rank : ranks - 1u - rank; /* the older rank of the pair */
        skew_part = 2u * ((low * cus + slot_cu) * W + wave) + (old ? 0u : 1u);
        skew_num = 32768u + args.conf_skew * (ranks - 1u - 2u * low) / max(ranks - 1u, 1u);
    }
    uint32_t region_of = worker; /* the record region this worker publishes at the end */
    uint32_t fills64 = 0; /* spread: lane j holds the fill of the share of this worker's part of row (row & ~63) + j */
    for (uint32_t slot = worker * K; slot < (spread ? worker * K + K : min(n_parts, worker * K + K)); slot++) {
        const uint32_t row = slot - worker * K;
        const uint32_t part = spread ? row * n_workers + (worker + 2731u * row) % n_workers : skew ? skew_part : slot;
        if (spread && (row & 63u) == 0) {
            /* the fills of the next 64 rows' shares in ONE round trip (most parts of a flood corpus are empty: a dependent load per
             * part was most of what they cost) */
            const uint32_t rj = row + lane, pj = rj * n_workers + (worker + 2731u * rj) % n_workers;
            fills64 = (rj < K && pj < n_parts) ? args.cand_counts[pj / Q] : 0u;
        }
        if (part >= n_parts) continue; /* (spread: the last row is not full) */
        const uint32_t r = part / Q, q = part - r * Q;
        if (spread || skew) t.rec_region = args.rec_stage + (uint64_t)part * args.rec_cap, region_of = part;
        const uint32_t n = min(spread ? (uint32_t)__shfl((int)fills64, (int)(row & 63u)) : args.cand_counts[r], args.cand_cap); /* never past the region, whatever the counter says */
        /* this part's entries [base, end): the q-th of Q pieces of the share's batches of 128, consecutive pieces of the corpus */
        /* (spread and skewed parts: pieces of whole half batches -- a dense half batch is 1 024 lookup positions; a skewed cut in
         * whole batches would move in steps of 6 % of a share) */
        const uint32_t gshift = (spread || skew) ? 6u : 7u, nb = (n + (1u << gshift) - 1u) >> gshift;
        uint32_t base = (q * nb / Q) << gshift;
        uint32_t end_b = (q + 1) * nb / Q;
        if (skew) { /* the share's batches cut at skew_num / 2^16 */
            const uint32_t cut = min(nb, (uint32_t)(((uint64_t)nb * skew_num + 32768u) >> 16));
            base = (q ? cut : 0u) << gshift;
            end_b = q ? nb : cut;
        }
        const uint32_t end = min(n, end_b << gshift);
        const bool any_entries = base < end;
        const uint32_t stride = 128u;
        uint64_t edge = 0; /* pair tables, the share's last part: the next share's first byte, when that share exists */
        if (PAIR && HAS_B) {
            const uint64_t n_full = args.total >> 10, per = (n_full + n_shares - 1) / n_shares;
            t.share_start = min(n_full, (uint64_t)r * per) << 10;
            const uint64_t next = min(n_full, ((uint64_t)r + 1) * per);
            /* (the part that holds the share's last entries -- the ends in question sort among its records; a share without
             * entries: its last part) */
            const bool owner = n ? (base < end && end == n) : q == Q - 1;
            /* the share with the corpus' partial last tile asks at `total` (no lookup position exists there), every other one at the
             * first byte behind its tiles -- the next share's first lookup, or `total` itself (hwlm_filter_kernel has the same rule) */
            const bool holds_tail = (args.total & 1023) && r == (per ? min((uint64_t)n_shares - 1, n_full / per) : 0);
            if (owner) edge = holds_tail ? ((args.total & 1) ? 0 : args.total) : next > (t.share_start >> 10) ? next << 10 : 0;
            t.late_skip = ~0ull;
        }
        HSGPU_ST(st_entries += end > base ? end - base : 0;)
        if (base < end || edge) { /* else nothing in this part */
            const uint4 *region = args.cand + 2ull * r * args.cand_cap;
            if (PAIR && HAS_B) {
                /* unfolded: queued here, sorted with everything else by record_sort_kernel. Folded: before the drain that the end in
                 * question sorts into (below); a part that does not start the share starts behind somebody's frontier */
                if (!fold && edge && lane == 0) pair_edge_probe<true>(t, edge);
                if (fold && base && base < end) t.late_skip = (uint64_t)((const uint32_t *)region)[8ull * base] * CHUNK;
            }
            /* (+16: the window of a chunk's last position is read as three dwords from entry byte 24; runtime.hip allocates the slack) */
            if (FAST) rs.region = __builtin_amdgcn_make_buffer_rsrc((void *)region, 0, (int)min((uint64_t)args.cand_cap * 32u + 16u, (uint64_t)0x7ffffff0), 0x00020000);
            bool syncing = false; /* folded: the rest queue is being emptied for a sorted drain */
            /* Dense scans (args.fold == 2: the scratch gives every chunk an entry, the reference's flood case): a batch with more
             * candidate positions than a step and the queue can order is confirmed POSITION-parallel -- dP consecutive positions of the
             * batch at a time, one per lane slot, a sorted drain after every such step -- so that the records still leave the wavefront
             * in delivery order and nothing has to be sorted afterwards. dP adapts: a step that found more matches than the queue holds
             * is taken back (the surplus was only counted) and done again on half the positions; one that filled less than half
             * the queue doubles it (flood: 64 positions, two matches each). (Dense mode used to mean record_sort_kernel: 10.4 ms for
             * the bench's 33.5 M flood records, profiles/r04_flood.txt.) */
            uint32_t dq = DENSE_POS, dP = 32, dense_base = 0, dm[2] = {0, 0}; /* the batch's next position, positions per step */
            /* The reference's flood case proper (src/fdr/flood_runtime.h:86-335: one confirm replayed along a run of equal bytes). A
             * dense batch whose chunks are consecutive, all made of ONE byte value (halo included), inside one block and clear of
             * its start has the same eight-byte window at every one of its positions: position 0 is confirmed and drained the
             * ordinary way (uni: dP = 1), and the records of the batch's other positions are the same records with `end` counted
             * on -- written straight behind them, position-major, which IS delivery order (run_replicate below). */
            bool uni = false, uni_wait = false;
            uint32_t uni_f = 0, uni_pos = 0, uni_from = 0, dP_saved = 32;
            for (;;) {
#if HSGPU_CONFIRM_ROTPRIO
                switch ((prio_rank + prio_step++) & 3u) { /* (the argument of s_setprio is an immediate) */
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
#endif
                uint32_t idx[CU] = {}, pend[CU] = {};
                const uint32_t nrq = __builtin_amdgcn_readfirstlane(t.wl->nrq);
                bool again = false; /* a step on (idx, pend) that somebody queued: the rest queue's entries, a dense batch's positions */
                if (DENSE && !PAIR && uni_wait && dq > uni_from) dq = uni_from; /* (a step's stride went past the run's first position: nothing behind it was looked at) */
                if (DENSE && dq < DENSE_POS && !syncing) {
                    if (DENSE && !PAIR && uni_wait && dq >= uni_from) {
                        /* everything in front of the run is out (a dense step ends in a drain): the run's first position, alone */
                        uni_wait = false;
                        uni = true;
                        uni_f = __builtin_amdgcn_readfirstlane(t.wl->nfront);
                        dq = uni_from;
                        dP_saved = dP;
                        dP = 1;
                    }
                    const uint32_t pos_end = (DENSE && !PAIR && uni_wait) ? uni_from : 128u * CHUNK; /* (the run's positions are not confirmed one by one) */
                    /* positions [dq, dq + dP) of the batch, one per lane slot: entry (position >> 4), its candidate bits of that position */
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t sigma = u * 64 + lane, a = dq + sigma, k = a >> 4;
                        const uint32_t lo = (uint32_t)__shfl((int)dm[0], (int)(k & 63u)), hi = (uint32_t)__shfl((int)dm[1], (int)(k & 63u));
                        const uint32_t m_e = k < 64 ? lo : hi;
                        pend[u] = (sigma < dP && a < pos_end) ? m_e & (0x10001u << (a & 15u)) : 0u;
                        idx[u] = pend[u] ? dense_base + k : 0u;
                    }
                    if (__ballot((pend[0] | pend[1]) != 0) == 0) { /* nothing there: on */
                        dq += dP;
                        continue;
                    }
                    t.mq_redo = dP > 1; /* a step that finds more than the queue holds is taken back and done again on half the positions */
                    again = true;
                } else if (!syncing && base < end && nrq <= RQ_CAP - 64 * CU) { /* a fresh batch (the step may queue up to 64 * CU more) */
                    const uint32_t i0 = base + lane, i1 = base + 64 + lane;
                    if (dense) {
                        const uint32_t *rw = (const uint32_t *)region;
                        const uint32_t m0 = i0 < end ? rw[8ull * i0 + 1] : 0u, m1 = i1 < end ? rw[8ull * i1 + 1] : 0u;
                        uint32_t cnt = __popc((m0 | m0 >> 16) & 0xffffu) + __popc((m1 | m1 >> 16) & 0xffffu);
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
                        if (cnt > DENSE_AT) {
                            const uint32_t nmq = __hip_atomic_load(&t.wl->nmq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            if (nrq || __builtin_amdgcn_readfirstlane(nmq)) { /* what is queued lies in front of this batch: out first */
                                syncing = true;
                                continue;
                            }
                            dm[0] = m0, dm[1] = m1, dense_base = base, dq = 0;
                            if (DENSE && !PAIR && fold) { /* a run of one byte value? (all loads and tests wave-uniform in their outcome) */
                                const uint4 *re = (const uint4 *)region;
                                const uint4 z = make_uint4(0, 0, 0, 0);
                                const uint4 a0 = i0 < end ? re[2ull * i0] : z, b0 = i0 < end ? re[2ull * i0 + 1] : z;
                                const uint4 a1 = i1 < end ? re[2ull * i1] : z, b1 = i1 < end ? re[2ull * i1 + 1] : z;
                                /* The batch's LAST chunk names the byte and the chunk numbers; the run is the batch's longest tail of consecutive
                                 * chunks made of that byte (halo included) with every lookup position a candidate, cut to what lies inside the
                                 * block of the last byte, 7 + `start` bytes clear of its first one (every window inside the block, every literal
                                 * -- <= 8 bytes -- clear of `start`). The chunks in front of it (a block's first one: its halo is the block
                                 * before) are confirmed position by position as ever (uni_from: where the run's positions begin). */
                                const uint32_t ne = min(128u, end - base), last = ne - 1u;
                                const uint32_t c_last = (uint32_t)__shfl((int)(last < 64 ? a0.x : a1.x), (int)(last & 63u));
                                const uint32_t vv = (uint32_t)__shfl((int)(last < 64 ? b0.x : b1.x), (int)(last & 63u));
                                constexpr uint32_t ALL = S2 ? 0x5555u : 0xffffu; /* every lookup position of the chunk (stride 2: the even ones) a candidate */
                                const bool ok0 = a0.x + (last - lane) == c_last && ((a0.y | a0.y >> 16) & 0xffffu) == ALL && a0.z == vv && a0.w == vv &&
                                                 b0.x == vv && b0.y == vv && b0.z == vv && b0.w == vv;
                                const bool ok1 = a1.x + (last - 64u - lane) == c_last && ((a1.y | a1.y >> 16) & 0xffffu) == ALL && a1.z == vv && a1.w == vv &&
                                                 b1.x == vv && b1.y == vv && b1.z == vv && b1.w == vv;
                                const uint64_t bad0 = __ballot(lane < ne && !ok0), bad1 = __ballot(64u + lane < ne && !ok1);
                                const uint32_t k_val = bad1 ? 128u - (uint32_t)__builtin_clzll(bad1) : bad0 ? 64u - (uint32_t)__builtin_clzll(bad0) : 0u; /* the tail's first entry */
                                if ((vv & 0xffu) * 0x01010101u == vv && k_val < ne) {
                                    const uint64_t g_last = (uint64_t)c_last * CHUNK + CHUNK - 1, g_tail = (uint64_t)(c_last - (last - k_val)) * CHUNK;
                                    if (!(t.cb_end && g_last >= t.cb_start && g_last < t.cb_end)) { /* the block of the run's last byte (every lane the same lookup) */
                                        uint64_t bs;
                                        const uint64_t b = block_of(t, g_last, bs);
                                        t.cb = rfl64(b), t.cb_start = rfl64(bs), t.cb_end = rfl64(t.off[min(b + 1, t.nblocks)]);
                                    }
                                    const uint64_t clear = (t.cb_start + 7 + t.start + (CHUNK - 1)) & ~(uint64_t)(CHUNK - 1);
                                    const uint64_t g_run = max(g_tail, clear); /* the run's first byte */
                                    if (g_run <= g_last) {
                                        const uint32_t k_run = k_val + (uint32_t)((g_run - g_tail) / CHUNK), n_run = ne - k_run; /* its first entry, its chunks */
                                        const uint32_t f0 = __builtin_amdgcn_readfirstlane(t.wl->nfront);
                                        bool goes_on = false;
                                        if (runs) {
                                            /* the run described last goes on here -- same byte, same block, the next byte of the corpus,
                                             * nothing emitted since: its lookups are this batch's too, nothing to confirm at all */
                                            uint32_t *pd = t.wl->pad;
                                            const uint64_t nxt = (uint64_t)pd[RUN_NEXT + 1] << 32 | pd[RUN_NEXT];
                                            goes_on = __builtin_amdgcn_readfirstlane((int)(k_run == 0 && pd[RUN_LIVE] && nxt == g_run && pd[RUN_VV] == vv &&
                                                                                           pd[RUN_BLOCK] == (uint32_t)t.cb && f0 == pd[RUN_PB] + pd[RUN_NM])) != 0;
                                            if (lane == 0) {
                                                const uint64_t after = g_last + 1;
                                                pd[RUN_NEXT] = (uint32_t)after, pd[RUN_NEXT + 1] = (uint32_t)(after >> 32);
                                                if (goes_on) {
                                                    constexpr uint32_t STEP = S2 ? 2u : 1u;
                                                    const uint32_t more = ne * CHUNK / STEP, nm = pd[RUN_NM];
                                                    pd[RUN_REPS] += more;
                                                    pd[RUN_VIRT] += more * nm;
                                                    if (nm) ((uint32_t *)(args.run_tab + (uint64_t)region_of * HSGPU_RUN_STRIDE + pd[RUN_N]))[2] = pd[RUN_REPS];
                                                } else {
                                                    pd[RUN_LIVE] = 0; /* (what the drain of the run's first position finds decides: below) */
                                                    pd[RUN_VV] = vv, pd[RUN_BLOCK] = (uint32_t)t.cb;
                                                }
                                            }
                                        }
                                        if (goes_on) {
                                            dq = DENSE_POS; /* the batch is done */
                                        } else {
                                            uni_pos = n_run * CHUNK;
                                            uni_from = k_run * CHUNK;
                                            uni_wait = true; /* (the dense step below starts the run when it gets to its first position) */
                                        }
                                    }
                                }
                            }
                            base += stride;
                            continue;
                        }
                    }
                    base += FAST ? 64u * CU : stride;
                    HSGPU_ST(st_fresh++;)
                    if constexpr (FAST) { /* the schedule of the general step with masks for booleans (confirm_step_fast) */
                        uint32_t vm[CU];
#pragma unroll
                        for (int u = 0; u < CU; u++) {
                            const uint32_t iu = i0 + 64u * u;
                            vm[u] = m_less(iu, end); /* ~0 when i < end: past the fill, entry 0 without candidate bits */
                            idx[u] = iu & vm[u];
                        }
                        confirm_step_fast<HAS_B, true, CU>(t, rs, rq, idx, pend, vm);
                    } else {
                        const bool valid[2] = {i0 < end, i1 < end};
                        idx[0] = valid[0] ? i0 : 0, idx[1] = valid[1] ? i1 : 0; /* end > 0 here: entry 0 exists */
                        confirm_step<HAS_A, HAS_B, HAS_C, S2, PAIR>(t, region, rq, idx, pend, valid, true);
                    }
                } else if (nrq) { /* entries with candidate bits left: same path, next bit */
                    const uint32_t k = min(nrq, 64u * CU), first_q = nrq - k;
#pragma unroll
                    for (int u = 0; u < CU; u++) {
                        if (FAST) {
                            const uint32_t vq = m_less(u * 64 + lane, k);
                            const uint2 it = rq[(first_q + u * 64 + lane) & vq];
                            idx[u] = it.x & vq, pend[u] = it.y & vq;
                        } else {
                            const bool v = u * 64 + lane < k;
                            const uint2 it = rq[v ? first_q + u * 64 + lane : 0];
                            idx[u] = v ? it.x : 0, pend[u] = v ? it.y : 0;
                        }
                    }
                    if (lane == 0) t.wl->nrq = first_q;
                    again = true;
                    HSGPU_ST(st_rest++;)
                } else if (fold) { /* a sync point with nothing pending: everything queued is final; in order into the region */
                    if (PAIR && HAS_B) {
                        /* the frontier: the entry that is confirmed next (by this wavefront, or by the one with the next quarter); the
                         * late-keyed literals ending right in front of its chunk belong into THIS drain */
                        const bool mid = DENSE && dq < DENSE_POS && dense_base + (dq >> 4) < n; /* inside a dense batch: position by position */
                        const uint32_t nxt = mid ? dense_base + (dq >> 4) : min(base, end);
                        uint64_t gf = 0;
                        if (nxt < n) gf = (uint64_t)((const uint32_t *)region)[8ull * nxt] * CHUNK + (mid ? dq & 15u : 0u);
                        else if (base >= end) gf = edge;
                        if (gf && gf != t.late_skip && gf != t.share_start) {
                            if (lane == 0) pair_edge_probe<true>(t, gf);
                            t.late_skip = gf;
                        }
                    }
                    drain_matches_sorted<DENSE && !PAIR>(t, lane);
                    syncing = false;
                    HSGPU_ST(st_drains++;)
                    if (DENSE && !PAIR && uni) {
                        /* position 0 of a run has been confirmed and its records are in the region: the run's other positions.
                         * (More than 64 literals on one window, or a wavefront that has already emitted out of order: the ordinary way.) */
                        const uint32_t f1 = __builtin_amdgcn_readfirstlane(t.wl->nfront), nm = f1 - uni_f;
                        const uint32_t ooo = __builtin_amdgcn_readfirstlane(t.wl->pad[0]);
                        uni = false;
                        dP = dP_saved;
                        if (nm <= 64 && !ooo && dq >= uni_from + 1) {
                            /* (stride 2: lookup 0 reported the ends g0 and g0 + 1; every later lookup of the run the same pair, two further on) */
                            constexpr uint32_t STEP = S2 ? 2u : 1u;
                            const uint32_t total_new = (uni_pos / STEP - 1) * nm;
                            /* with run tables: the other positions as a descriptor (record_sort_kernel writes their records where they go);
                             * a region with HSGPU_RUN_MAX runs already, or one that has lost records: staged like everything else */
                            const bool described = runs && f1 <= t.rec_cap && (!nm || __builtin_amdgcn_readfirstlane((int)t.wl->pad[RUN_N]) < HSGPU_RUN_MAX);
                            if (described) {
                                if (lane == 0) {
                                    uint32_t *pd = t.wl->pad;
                                    pd[RUN_PB] = uni_f, pd[RUN_NM] = nm, pd[RUN_REPS] = uni_pos / STEP - 1, pd[RUN_LIVE] = 1;
                                    if (nm) {
                                        const uint32_t k = ++pd[RUN_N];
                                        pd[RUN_VIRT] += total_new;
                                        args.run_tab[(uint64_t)region_of * HSGPU_RUN_STRIDE + k] = make_uint4(uni_f, nm, uni_pos / STEP - 1, STEP);
                                    }
                                }
                            } else if (nm) {
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* (the drain's stores, read back below) */
                                const bool have = lane < nm && uni_f + lane < t.rec_cap;
                                const uint4 br = have ? t.rec_region[uni_f + lane] : make_uint4(0, 0, 0, 0); /* lane r: record r of position 0 */
                                const uint32_t inv = nm > 1 ? (uint32_t)(((1ull << 32) + nm - 1) / nm) : 0u; /* i / nm = mulhi(i, inv) for i * nm < 2^32 (nm = 1: 2^32 does not fit) */
                                for (uint32_t i0r = 0; i0r < total_new; i0r += 64) { /* whole wavefronts: the shuffles read every lane */
                                    const uint32_t i = i0r + lane, q = nm > 1 ? __umulhi(i, inv) : i, r = (i - q * nm) & 63u;
                                    uint4 rec;
                                    rec.x = (uint32_t)__shfl((int)br.x, (int)r), rec.y = (uint32_t)__shfl((int)br.y, (int)r) + STEP * (q + 1u);
                                    rec.z = (uint32_t)__shfl((int)br.z, (int)r), rec.w = (uint32_t)__shfl((int)br.w, (int)r);
                                    const uint64_t at = (uint64_t)f1 + i;
                                    if (i < total_new && at < t.rec_cap) t.rec_region[at] = rec;
                                }
                                if (lane == 0) t.wl->nfront = f1 + total_new; /* (keeps counting past the capacity: the total stays exact) */
                            }
                            dq = DENSE_POS; /* the batch is done */
                        }
                    }
                    if (base >= end && dq >= DENSE_POS) break;
                    continue;
                } else {
                    break;
                }
                if (again) { /* (a lane with nothing to do: entry 0, no candidate bits) */
                    if constexpr (FAST) {
                        const uint32_t vm[CU] = {}; /* (only a fresh step masks with it) */
                        confirm_step_fast<HAS_B, false, CU>(t, rs, rq, idx, pend, vm);
                    } else {
                        const bool valid[2] = {pend[0] != 0, pend[1] != 0};
                        confirm_step<HAS_A, HAS_B, HAS_C, S2, PAIR>(t, region, rq, idx, pend, valid, false);
                    }
                }
                if (fold) {
                    const uint32_t nmq = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&t.wl->nmq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
                    if (DENSE && t.mq_redo) { /* a dense step */
                        t.mq_redo = 0;
                        if (nmq > MQ_CAP) { /* (the matches past the queue were not resolved: nothing of this step has left the wavefront) */
                            if (lane == 0) __hip_atomic_store(&t.wl->nmq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            dP >>= 1;
                            continue;
                        }
                        dq += dP;
                        if (2 * nmq <= MQ_CAP && dP < 128) dP <<= 1;
                        syncing = true;
                    } else if (DENSE && dq < DENSE_POS) { /* (a single position, resolved in place past the queue's capacity) */
                        dq += dP;
                        syncing = true;
                    }
                    syncing = syncing || nmq > SYNC_AT || base >= end;
                } else {
                    drain_matches(t, lane, 63);
                    flush_records(t, lane, OFLUSH);
                }
            }
            if (!fold) drain_matches(t, lane, 0);
        }
        if (spread && any_entries) { /* this part's region is complete: published on its own, the wavefront's counters start over
                                     * (a part without entries: its words of the control block are zero as they are, nobody reads its run table) */
            if (DENSE && !PAIR && runs) publish_records<true>(t, args, lane, part, true);
            else publish_records(t, args, lane, part, true);
            if (fold && lane == 0 && t.wl->pad[0]) atomicAdd(&args.rec_super[HSGPU_SUPER_FLAGS], 1ull << 32);
            init_wave_lds(t, &wave_lds[wave], lane);
        }
    }
    /* the worker's region is complete (in delivery order when folded): its fill, once, added to the sum of its super. Placing
     * the regions is record_sort_kernel's (folded: a gather, the regions are consecutive sorted runs). (Placed by the workers
     * themselves -- every worker polling the sums until everybody in front had published -- the stage took 0.30 ms instead of
     * 0.15: 6 144 wavefronts polling the same few hundred words, the publishing atomics queued behind the polls.) */
#if HSGPU_CONFIRM_STAMPS
    if (args.conf_stamps && lane == 0) {
        args.conf_stamps[4 * worker + 1] = (unsigned long long)st_fresh | (unsigned long long)st_rest << 16 | (unsigned long long)st_drains << 32;
        args.conf_stamps[4 * worker + 2] = st_entries;
        args.conf_stamps[4 * worker + 3] = wall_clock64();
    }
#endif
#undef HSGPU_ST
    if (spread) return;
    publish_records(t, args, lane, region_of, true);
    if (fold && lane == 0 && t.wl->pad[0]) atomicAdd(&args.rec_super[HSGPU_SUPER_FLAGS], 1ull << 32); /* emitted out of order: "again", in dense mode */
}

/* ---- phase 3 as a kernel of its own: fused scans and dense mode -------------------------------
 *   publish      every producing wavefront adds its region's fill to the sum of its "super" (2^super_shift
 *                consecutive regions, at most 256 supers): one atomic per region, not per record
 *   record_sort  one workgroup per share: where its records go = the supers in front of it + the fills in front
 *                of it inside its own super (one round trip of independent loads); sort_share; workgroup 0 writes
 *                *count; then the OTHER control block (the previous scan's) goes back to zero for the next scan.
 * 256 threads per share; 1024 once the scratch has seen dense input (runtime.hip): a dense share holds tens of
 * thousands of records and its sort is this one workgroup's work. */
__global__ __launch_bounds__(1024) void record_sort_kernel(HsgpuScanArgs args) {
    __shared__ uint4 buf[SORT_LDS];
    __shared__ unsigned long long placed[3]; /* records in front of this share, records in all, overflow flag */
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t first = blockIdx.x * args.group_regions, last = min(args.rec_regions, first + args.group_regions);
    const uint2 *counts = (const uint2 *)args.rec_counts;
    if (tid < 64) {
        /* where this share's records go: the supers in front of its own + the regions of its own super in front of
         * it (fewer than 2^super_shift fills, 64 at a time); all loads independent, one round trip */
        const uint32_t n_super = (args.rec_regions + (1u << args.super_shift) - 1) >> args.super_shift; /* <= 256 */
        const uint32_t S = first >> args.super_shift;
        /* (clamped, unconditional loads, all issued before the first is used: as a loop with its bounds check
         * this was five dependent round trips at the head of every workgroup, most of the kernel's 55 us) */
        unsigned long long sv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) sv[k] = args.rec_super[HSGPU_SUPER(min(lane + 64u * k, 255u))]; /* (the flags word is read below) */
        /* a staging region lost records, or (two-phase pipeline) a candidate region overflowed and the confirm kernel did
         * nothing: either way this scan delivers nothing and says so */
        const unsigned long long flag =
            args.rec_super[HSGPU_SUPER_FLAGS] | ((args.cand_counts && args.overflow_note) ? args.cand_counts[args.cand_waves] : 0u);
        const uint32_t c0 = (S << args.super_shift) + lane;
        const uint2 cfirst = counts[min(c0, args.rec_regions - 1)];
        unsigned long long before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = lane + 64u * k;
            if (i < n_super) all += sv[k];
            if (i < S) before += sv[k]; /* S <= n_super */
        }
        if (c0 < first) before += (unsigned long long)cfirst.x + cfirst.y;
        for (uint32_t i = c0 + 64; i < first; i += 64) { /* supers of more than 64 regions: very large grids only */
            const uint2 c = counts[i];
            before += (unsigned long long)c.x + c.y;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += __shfl_xor(before, d);
            all += __shfl_xor(all, d);
        }
        if (lane == 0) {
            placed[0] = before;
            placed[1] = all;
            placed[2] = flag;
            if (blockIdx.x == 0) {
                /* a region that ran out of space lost records; its fill counters kept counting, so the total is
                 * still exact: report it, but never a value <= cap (that would claim the output is complete).
                 * Candidate overflow: nothing was confirmed, the total is unknown: cap + 1 ("again"), and the word in
                 * mapped host memory that makes the next scan on this scratch give every chunk an entry of its own. */
                *args.count = (flag && all <= args.cap) ? args.cap + 1 : all;
                /* ... and so does the folded pipeline when a wavefront had to emit out of order (bits 32+ of the flags word): dense mode next */
                if (args.overflow_note && ((args.cand_counts && args.cand_counts[args.cand_waves]) || (flag >> 32))) *args.overflow_note = 1u;
                if (args.tstamp) args.tstamp[2] = wall_clock64(); /* the stages before the sort are done */
            }
        }
    }
    __syncthreads();
    const unsigned long long base = placed[0];
    const bool complete = !placed[2] && placed[1] <= args.cap; /* no region overflowed, everything fits */
    if (complete) sort_share(args, buf, first, last, base);
    if (blockIdx.x == 0 && tid == 0 && args.tstamp_next) {
        args.tstamp[3] = wall_clock64(); /* the scan's last kernel (its first workgroup: the stages before it are done) */
        args.tstamp_next[0] = ~0ull;
        args.tstamp_next[1] = 0;
        args.tstamp_next[2] = 0;
        args.tstamp_next[3] = 0;
    }
    scan_epilogue(args);
}

/* ---- phase 0 (fused-only pipeline): the hints as a kernel of their own ------------ */
__global__ __launch_bounds__(256) void block_hint_kernel(const uint64_t *off, uint64_t nblocks, uint64_t total,
                                                         uint32_t *hint, uint64_t n_hint) {
    write_block_hints(off, nblocks, hint, n_hint, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, threadIdx.x & 63);
}


} // namespace

#endif
