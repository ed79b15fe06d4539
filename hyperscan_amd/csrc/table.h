/*
 * table.h -- layout of the compiled literal table ("blob") shared by the host
 * compiler (compile.cpp), the runtime (runtime.hip) and the scan kernels.
 *
 * The blob is position independent (offsets from its first byte), so it can be
 * serialised as-is (the reference's bytecode has the same property, tested by
 * unit/internal/fdr.cpp:408-445) and uploaded to HBM with one copy.
 *
 * Roles (reference analogue in brackets):
 *   filter   u32[2^k]  hashed suffix bit-filter, staged in LDS by every workgroup
 *                      [FDR table fdr.c:157-244 / Teddy nibble masks teddy.c:918-969]
 *   c2bits   u32[2048] exact 2-byte-suffix bit table, staged in LDS (only when
 *                      some literal is keyed on <= 2 bytes)
 *   ht_a/b   open-addressed tables of 16-byte buckets (4 tagged 32-bit slots) keyed by
 *                      the 4-/3-byte suffix variant [litIndex hash, fdr_confirm_runtime.h:51-56]
 *   c2ref    u32[65536] list reference per 2-byte suffix
 *   lists    u32[]     literal indices, bit 31 marks the last entry of a list
 *   lits     DevLit[]  per literal (v, msk, groups, id, size, flags)
 *                      [struct LitInfo, fdr_confirm.h:57-83]
 */
#ifndef HSGPU_TABLE_H
#define HSGPU_TABLE_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HSGPU_HD __host__ __device__ __forceinline__
#else
#define HSGPU_HD static inline
#endif

#define HSGPU_TABLE_MAGIC 0x54475348u /* "HSGT" */
#define HSGPU_TABLE_VERSION 12u

#define HSGPU_F_HAS_A 1u /* literals keyed on their last 4 bytes */
#define HSGPU_F_HAS_B 2u /* literals keyed on their last 3 bytes */
#define HSGPU_F_HAS_C 4u /* literals keyed on their last <= 2 bytes */
#define HSGPU_F_REPL 8u  /* filter is bank-replicated (small sets): conflict-free LDS reads */
#define HSGPU_F_K2 16u   /* keys set/test two bits of their filter word */
#define HSGPU_F_STRIDE2 32u /* lookups at even positions only; keys for delta 0 and 1 */
#define HSGPU_F_BLIND 64u   /* hash and exact-table keys ignore bit 5 (the ASCII case bit) of every byte */
#define HSGPU_F_BFOLD 128u  /* the (few) 3-byte keys own their whole filter word: the filter kernel runs the
                             * 4-byte-key test only, and a hit means "probe both exact tables" */
#define HSGPU_F_PAIR 256u   /* pair filter (large sets, short literals included): stride 2, 64-bit entries hashed on the
                             * 3 bytes ending at the lookup position, one 32-bit plane indexed by the byte BEFORE them
                             * and one by the byte AFTER them; see "the pair filter" below */
#define HSGPU_F_GATE 512u   /* the c2bits section holds the key gate: 64 Kbit, bit hsgpu_key_gate_bit(key) set for every
                             * key of the two exact tables (3-byte keys salted). The confirm kernel stages it in LDS
                             * and probes a table (one divergent 16-byte read per lane) only for keys that pass */

#define HSGPU_F_WIDE 1024u  /* hashed stride-1 two-bit filter with 64-bit entries {lo, hi} (2^filter_log2 of them): a 4-byte key's
                             * first bit -- b3 & 31, the byte in front of the hashed three -- lives in lo, its second bit in hi
                             * and depends on the hash alone (prod & 31). A folded 3-byte key (HSGPU_F_BFOLD) then owns only the
                             * lo half of its entry: the hi bit still discriminates. Either test is ONE shift whose amount the
                             * hardware takes from the low five bits of a byte (an SDWA select) or of the product itself --
                             * no index arithmetic at all (tools/sim/b2p.py: these two indices pass fewer candidates than
                             * (a + b3) and byte 1 of the product, which cost two more instructions per lookup) */

#define HSGPU_F_BLOOM 2048u /* opt-in (HSGPU_BUILD_FORCE_BLOOM), stride-1 tables that would carry the key gate: the c2bits section is 12 KiB,
                             * three planes of 2^15 bits -- a Bloom filter over the FULL-window keys of the literals, probed by the
                             * confirm kernel in LDS before it reads an exact-table bucket. The key gate tests the four bytes the
                             * filter kernel already tested, so 3.6 M of 9.5 M candidate positions per GiB of the bench corpus
                             * fetch a bucket from L2 (and 2.8 M of them a literal) only for the 8-byte compare to fail; keyed
                             * on five bytes where the literal has them, 0.7 M do (tools/sim/bloomgate.py). Groups:
                             *   3  every 3-byte key of table B                      hi = key << 8,            lo = 0
                             *   4  the 4-byte keys of literals that are not in 5    hi = key,                 lo = 0
                             *   5  literals whose last FIVE bytes are fully given   hi = last four bytes,     lo = fifth << 24
                             *      (up to the case bit of a case-blind table)       (DevLit.v's alignment: last byte on top)
                             * A position is probed in table A when group 4 or group 5 passes, in table B when group 3 does. */

#define HSGPU_FILTER_MUL 0x9E3779u /* 24-bit odd multiplier (golden ratio) */
#define HSGPU_HT_MUL 0x9E3779B1u
#define HSGPU_LIST_END 0x80000000u
/* exact-table slot (32 bits): DIRECT | delta | 6-bit key tag | 24-bit index.
 * DIRECT: index = ONE literal (ending at q + delta); else index = 1 + start of a list in
 * lists[]. 0 = empty. A tag match is only a hint: the literal's own (v, msk) decides. */
#define HSGPU_REF_DIRECT 0x80000000u
#define HSGPU_LIST_DELTA_SHIFT 30     /* bit 30 (slots and list entries): the literal ends at q + 1, not at q */
#define HSGPU_LIST_LIT_MASK 0x00ffffffu
#define HSGPU_SLOT_TAG_SHIFT 24
#define HSGPU_SLOT_TAG_MASK 0x3fu
#define HSGPU_BUCKET_SLOTS 4u

#define HSGPU_LIT_NORUNS 1u

struct HsgpuTableHeader {
    uint32_t magic;
    uint32_t version;
    uint32_t blob_bytes;
    uint32_t flags;
    uint32_t n_lits;
    uint32_t max_size;
    uint32_t filter_log2; /* hashed: log2(words); replicated: log2(rows), 32 words per row */
    uint32_t filter_entries;
    uint32_t ht_a_log2; /* log2(buckets), HSGPU_BUCKET_SLOTS slots each */
    uint32_t ht_b_log2;
    uint32_t n_a, n_b, n_c;
    uint32_t off_filter;
    uint32_t off_c2bits;
    uint32_t off_ht_a;
    uint32_t off_ht_b;
    uint32_t off_c2ref;
    uint32_t off_lists;
    uint32_t n_lists;
    uint32_t off_lits;
    uint32_t checksum; /* adler-style sum over everything after the header */
    uint32_t hash_mask; /* pair filter: bits of the 3 hashed bytes that enter the hash (byte q-2 in bits 0..7) */
    uint32_t n_m;       /* pair filter: literals keyed one byte late (they end at q - 1) */
    uint32_t reserved[8];
};
static_assert(sizeof(HsgpuTableHeader) == 128, "header is 128 bytes");

typedef uint32_t HsgpuHtSlot; /* see HSGPU_REF_DIRECT above; a bucket = 4 slots = one 16-byte read */

struct HsgpuDevLit {
    uint64_t v;      /* literal bytes, last byte in the most significant byte */
    uint64_t msk;    /* compare mask, same alignment */
    uint64_t groups; /* hwlm_group_t */
    uint32_t id;
    uint8_t size; /* max(len(s), len(msk)) */
    uint8_t flags;
    uint16_t pad;
};
static_assert(sizeof(HsgpuDevLit) == 32, "DevLit is 32 bytes");

/* ---- the filter hash, identical on host (insert) and device (probe) --------
 * For a lookup at byte position q (b0 = buf[q], b1 = buf[q-1], b2 = buf[q-2], b3 = buf[q-3]):
 *   x    = b2 | b1 << 8 | b0 << 16                  (the 3 bytes ending at q; & 0xdfdfdf if BLIND)
 *   prod = (x * HSGPU_FILTER_MUL) mod 2^32          (one v_mul_u32_u24)
 * hashed filter of 2^k words:
 *   a    = prod >> (30 - k);  word = a >> 2         (a keeps two sub-word hash bits)
 * replicated filter of 2^r rows x 32 identical columns (lane l reads column l & 31):
 *   a    = prod >> (32 - r);  word = a * 32 + column
 * bits tested in that word:
 *   bitA  = (a + b3) & 31              4-byte key, first bit   (v_add_u32_sdwa + v_bfe)
 *   bitA2 = ((prod >> 8) + b3) & 31    4-byte key, second bit  (only with HSGPU_F_K2; byte 1 of prod: an SDWA select.
 *                                      Without b3 here -- one v_add less per lookup -- the fdr10k filter passed 27 %
 *                                      more candidates and ran no faster: table version 8, withdrawn)
 *   bitB  = a & 31                     3-byte key, first bit
 *   bitB2 = (prod >> 8) & 31           3-byte key, second bit  (only with HSGPU_F_K2)
 * With HSGPU_F_BFOLD a 3-byte key sets all 32 bits of its word instead (any b3 passes the bitA tests).
 */
HSGPU_HD uint32_t hsgpu_filter_prod(uint32_t x24) { return (x24 & 0xffffffu) * HSGPU_FILTER_MUL; }
HSGPU_HD uint32_t hsgpu_filter_shift(uint32_t flags, uint32_t log2) {
    return (flags & HSGPU_F_REPL) ? 32u - log2 : (flags & HSGPU_F_WIDE) ? 29u - log2 : 30u - log2; /* WIDE: entry = a >> 3 */
}
HSGPU_HD uint32_t hsgpu_filter_bit_a(uint32_t b3, uint32_t a) { return (b3 + a) & 31u; }
HSGPU_HD uint32_t hsgpu_filter_bit_a2(uint32_t b3, uint32_t prod) { return (b3 + (prod >> 8)) & 31u; }
HSGPU_HD uint32_t hsgpu_filter_bit_b(uint32_t a) { return a & 31u; }
HSGPU_HD uint32_t hsgpu_filter_bit_b2(uint32_t prod) { return (prod >> 8) & 31u; }
HSGPU_HD uint32_t hsgpu_filter_bit_lo_wide(uint32_t b3) { return b3 & 31u; } /* WIDE: the first bit, in the lo word */
HSGPU_HD uint32_t hsgpu_filter_bit_hi(uint32_t prod) { return prod & 31u; }       /* WIDE: the second bit, in the hi word */
HSGPU_HD uint32_t hsgpu_filter_words(uint32_t flags, uint32_t log2) {
    return (flags & (HSGPU_F_PAIR | HSGPU_F_WIDE)) ? (2u << log2) : (flags & HSGPU_F_REPL) ? (32u << log2) : (1u << log2);
}

/* ---- the pair filter (HSGPU_F_PAIR) ------------------------------------------------------
 * Lookups at even positions q only. With b0 = buf[q], b1 = buf[q-1], b2 = buf[q-2], b3 = buf[q-3] and
 * nx = buf[q+1]:
 *   x     = (b2 | b1 << 8 | b0 << 16) & hash_mask     (hash_mask drops the case bit when the table is
 *                                                      case-blind and keeps only 5 bits of b0)
 *   prod  = x * HSGPU_FILTER_MUL mod 2^32
 *   entry = prod >> (32 - filter_log2)                 64-bit entries {B, A}, 2^filter_log2 of them
 *   pass  = ((B >> (b3 & 31)) & (B >> ((b3 >> 3) & 31)) | (A >> (nx & 31))) & (A >> ((prod >> 8) & 31)) & 1
 * Every literal is keyed twice, once per parity of its end offset e:
 *   e even: lookup q = e (delta 0);
 *   e odd:  lookup q = e - 1 (delta +1: its last byte is nx) or q = e + 1 (delta -1: b0 is not part of
 *           it and is enumerated), whichever needs fewer entries.
 * A key sets bit (prod >> 8) & 31 of A in each of its entries, and either the two B bits of every b3 value it
 * admits ("B key": all 32 when b3 lies outside the literal) or the A bit of every nx value it admits
 * ("A key": literals of 4 bytes at delta +1, whose b3 is unknown but whose last byte is nx). */
HSGPU_HD uint32_t hsgpu_pair_entry(uint32_t prod, uint32_t log2) { return prod >> (32u - log2); }
HSGPU_HD uint32_t hsgpu_pair_bit_h(uint32_t prod) { return (prod >> 8) & 31u; }
HSGPU_HD uint32_t hsgpu_pair_bit_b1(uint32_t b3) { return b3 & 31u; }
HSGPU_HD uint32_t hsgpu_pair_bit_b2(uint32_t b3) { return (b3 >> 3) & 31u; }
HSGPU_HD uint32_t hsgpu_pair_bit_a(uint32_t nx) { return nx & 31u; }
/* exact-table key of a literal keyed one byte late: the 3 bytes ending at q - 1, told apart from the 3-byte
 * keys ending at q by bit 24 */
#define HSGPU_KEY_M 0x01000000u
/* gate bitmap (64 Kbit, in the c2bits section of a pair table): is there ANY 3-byte key (of either kind)
 * with this hash? The confirm step probes the 3-byte exact table only then. */
HSGPU_HD uint32_t hsgpu_gate_bit(uint32_t key25) { return (key25 * HSGPU_HT_MUL) >> 16; }

/* key gate (HSGPU_F_GATE): the top 16 bits of the product the bucket index is taken from; 3-byte keys carry a salt
 * in their free top byte so that they do not alias the 4-byte key with a zero first byte */
#define HSGPU_GATE_B_SALT 0xB5000000u
HSGPU_HD uint32_t hsgpu_key_gate_bit(uint32_t key) { return (key * HSGPU_HT_MUL) >> 16; }

/* the Bloom gate's three bit indices, one per plane of 2^15 bits, for a masked window {hi, lo} of group `salt` */
#define HSGPU_BLOOM_PLANE_LOG2 15u
#define HSGPU_BLOOM_WORDS (3u << (HSGPU_BLOOM_PLANE_LOG2 - 5)) /* 3072 words = 12 KiB */
#define HSGPU_BLOOM_M1 0x9E3779B1u
#define HSGPU_BLOOM_M2 0x85EBCA6Bu
#define HSGPU_BLOOM_M3 0xC2B2AE35u
#define HSGPU_BLOOM_M4 0x27D4EB2Fu
#define HSGPU_BLOOM_M5 0x632BE5ABu
HSGPU_HD void hsgpu_bloom_idx(uint32_t hi, uint32_t lo, uint32_t salt, uint32_t (&idx)[3]) {
    uint32_t h = (hi * HSGPU_BLOOM_M1) ^ (lo * HSGPU_BLOOM_M2 + salt * HSGPU_BLOOM_M5);
    h ^= h >> 15;
    const uint32_t p1 = h * HSGPU_BLOOM_M3, p2 = h * HSGPU_BLOOM_M4;
    idx[0] = p1 >> (32u - HSGPU_BLOOM_PLANE_LOG2); /* indices from the TOP bits of a product: its low bits depend on few key bits */
    idx[1] = p2 >> (32u - HSGPU_BLOOM_PLANE_LOG2);
    idx[2] = (p1 >> 2) & ((1u << HSGPU_BLOOM_PLANE_LOG2) - 1u);
}

HSGPU_HD uint32_t hsgpu_ht_bucket(uint32_t key, uint32_t log2) { return (key * HSGPU_HT_MUL) >> (32u - log2); }
HSGPU_HD uint32_t hsgpu_ht_tag(uint32_t key, uint32_t log2) { return ((key * HSGPU_HT_MUL) >> (26u - log2)) & HSGPU_SLOT_TAG_MASK; }

#endif
