/*
 * oracle/hwlm_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker).
 *
 * A plain-C CPU restatement of the reference's block-mode literal-matcher
 * contract (hwlmExec and the character-class accelerators).  It is NOT part of
 * the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * the golden vectors of the reference's own unit tests (unit/internal/fdr.cpp,
 * fdr_flood.cpp, noodle.cpp, shufti.cpp, truffle.cpp, vermicelli.cpp) and, where
 * oracle/_ref/libhsref.so (the reference compiled in place) is present, against
 * the reference itself on seeded random inputs.
 *
 * What is restated (reference file:line):
 *   literal normalisation   src/hwlm/hwlm_literal.cpp:82-115 (nocase => upper-case;
 *                           all-zero msk dropped)
 *   (v, msk, size) per lit  src/fdr/fdr_confirm_compile.cpp:73-127 (fillLitInfo)
 *   the exact confirm       src/fdr/fdr_confirm_runtime.h:43-102 (confWithBit):
 *                           (conf_key & msk) == v, NOREPEAT vs last_match, left
 *                           bound, group gate, control = cb(...)
 *   conf_key                src/fdr/fdr.c:360 / teddy_runtime_common.h:395-416:
 *                           the 8 bytes ending at `end`, little-endian, bytes
 *                           before the buffer read as zero (block mode, no history)
 *   start semantics         src/hwlm/hwlm.h:101-118 and SURVEY.md section 8(a): a match is
 *                           reported iff its first byte offset end-size+1 >= start
 *   termination             src/fdr/fdr.c:719-721 (control == 0 => HWLM_TERMINATED)
 *   shufti membership       src/nfa/shufti.c:75-102 (scalar form lo[c&15] & hi[c>>4])
 *   truffle membership      src/nfa/truffle.c:64-81, src/nfa/trufflecompile.cpp:59-94
 *   vermicelli              src/nfa/vermicelli.h:42-104
 *
 * Callback order inside one `end` offset is engine specific in the reference
 * (bucket / hash-chain order); this oracle fixes it to literal-index order, and
 * parity is defined on the sorted multiset of (end, id).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct hso_lit {
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
} hso_lit_t;

typedef uint64_t (*hso_cb_t)(size_t end, uint32_t id, void *ctx);

typedef struct hso_info { /* one per literal: LitInfo, fdr_confirm.h:57-83 */
    uint64_t v, msk, groups;
    uint32_t id;
    uint8_t size, noruns;
} hso_info_t;

typedef struct hso_table {
    hso_info_t *li;
    size_t n;
    /* candidate index keyed by the last two bytes of the window; a pure
     * prefilter -- every candidate still goes through the full check. */
    uint32_t *idx2_off; /* 65537 */
    uint32_t *idx2;     /* literal indices, ascending within each key */
    uint32_t *one;      /* literals of size 1 (checked when end == 0 too) */
    size_t n_one;
    int brute;
} hso_table_t;

static int is_alpha(uint8_t c) { /* ourisalpha, src/util/compare.h */
    return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
}

static uint64_t right_aligned_u64(const uint8_t *p, uint32_t len) {
    /* make_u64a_mask, fdr_confirm_compile.cpp:57-64: last byte of p becomes the
     * most significant byte of the little-endian u64. */
    uint64_t m = 0;
    uint32_t n = len < 8 ? len : 8;
    memcpy((uint8_t *)&m + 8 - n, p + len - n, n);
    return m;
}

static void fill_info(const hso_lit_t *l, hso_info_t *o) {
    uint64_t msk = ~0ULL, val = 0;
    int all_zero = 1;
    for (uint32_t j = 0; j < 8; j++) {
        uint32_t sh = (7 - j) * 8;
        if (j >= l->len) {
            msk &= ~(0xffULL << sh);
        } else {
            uint8_t c = l->s[l->len - j - 1];
            if (l->nocase && is_alpha(c)) {
                msk &= ~(0x20ULL << sh);
                val |= (uint64_t)(c & 0xdf) << sh;
            } else {
                val |= (uint64_t)c << sh;
            }
        }
    }
    uint32_t mlen = l->msk_len;
    for (uint32_t j = 0; j < mlen; j++) {
        if (l->msk[j]) all_zero = 0;
    }
    if (all_zero) mlen = 0; /* hwlm_literal.cpp:110-114 */
    if (mlen) {
        msk |= right_aligned_u64(l->msk, mlen);
        val |= right_aligned_u64(l->cmp, mlen);
    }
    o->v = val;
    o->msk = msk;
    o->groups = l->groups;
    o->id = l->id;
    o->size = (uint8_t)(mlen > l->len ? mlen : l->len);
    o->noruns = l->noruns;
}

void hso_free(hso_table_t *t) {
    if (!t) return;
    free(t->li);
    free(t->idx2_off);
    free(t->idx2);
    free(t->one);
    free(t);
}

hso_table_t *hso_build(const hso_lit_t *lits, size_t n, int brute) {
    hso_table_t *t = (hso_table_t *)calloc(1, sizeof(*t));
    t->li = (hso_info_t *)calloc(n ? n : 1, sizeof(hso_info_t));
    t->n = n;
    t->brute = brute;
    for (size_t i = 0; i < n; i++) {
        if (lits[i].len > 8 || lits[i].msk_len > 8 || (lits[i].len == 0 && lits[i].msk_len == 0)) {
            hso_free(t);
            return NULL;
        }
        fill_info(&lits[i], &t->li[i]);
    }
    /* two-pass counting sort of (key -> literal) over all 2-byte keys each
     * literal is consistent with: (key16 & m16) == v16 on the top two bytes. */
    t->idx2_off = (uint32_t *)calloc(65537, sizeof(uint32_t));
    t->one = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    for (int pass = 0; pass < 2; pass++) {
        uint32_t *fill = NULL;
        if (pass == 1) {
            uint32_t acc = 0;
            for (int k = 0; k < 65536; k++) {
                uint32_t c = t->idx2_off[k];
                t->idx2_off[k] = acc;
                acc += c;
            }
            t->idx2_off[65536] = acc;
            t->idx2 = (uint32_t *)calloc(acc ? acc : 1, sizeof(uint32_t));
            fill = (uint32_t *)calloc(65536, sizeof(uint32_t));
        }
        for (size_t i = 0; i < n; i++) {
            uint32_t m16 = (uint32_t)(t->li[i].msk >> 48), v16 = (uint32_t)(t->li[i].v >> 48);
            uint32_t wild = ~m16 & 0xffff;
            /* enumerate all submasks of the wildcard bits */
            uint32_t sub = 0;
            do {
                uint32_t key = v16 | sub;
                if (pass == 0) t->idx2_off[key]++;
                else t->idx2[t->idx2_off[key] + fill[key]++] = (uint32_t)i;
                sub = (sub - wild) & wild;
            } while (sub != 0);
            if (pass == 0 && t->li[i].size == 1) t->one[t->n_one++] = (uint32_t)i;
        }
        free(fill);
    }
    return t;
}

static uint64_t conf_key(const uint8_t *buf, size_t end) {
    uint64_t w = 0;
    size_t n = end + 1 < 8 ? end + 1 : 8;
    memcpy((uint8_t *)&w + 8 - n, buf + end + 1 - n, n);
    return w;
}

static int lit_matches(const hso_info_t *li, uint64_t w, size_t end, size_t start) {
    if ((w & li->msk) != li->v) return 0;
    if (end + 1 < li->size) return 0;             /* would overhang the block start */
    if (end + 1 - li->size < start) return 0;     /* must start at/after `start` */
    return 1;
}

/* Mirror of hwlmExec. Returns 0 (HWLM_SUCCESS) or 1 (HWLM_TERMINATED). */
int hso_exec(const hso_table_t *t, const uint8_t *buf, size_t len, size_t start,
             hso_cb_t cb, void *ctx, uint64_t groups) {
    if (!groups) return 0; /* hwlm.c:178 */
    uint64_t control = groups;
    uint32_t last_match = 0xffffffffu; /* INVALID_MATCH_ID, fdr.c:766 */
    for (size_t e = start; e < len; e++) {
        uint64_t w = conf_key(buf, e);
        const uint32_t *cand;
        size_t nc;
        if (t->brute) {
            cand = NULL;
            nc = t->n;
        } else if (e == 0) {
            cand = t->one;
            nc = t->n_one;
        } else {
            uint32_t key = (uint32_t)(w >> 48);
            cand = t->idx2 + t->idx2_off[key];
            nc = t->idx2_off[key + 1] - t->idx2_off[key];
        }
        for (size_t k = 0; k < nc; k++) {
            const hso_info_t *li = &t->li[cand ? cand[k] : k];
            if (!lit_matches(li, w, e, start)) continue;
            if (li->noruns && last_match == li->id) continue;
            if (!(li->groups & control)) continue;
            last_match = li->id;
            control = cb(e, li->id, ctx);
            if (!control) return 1;
        }
    }
    return 0;
}

typedef struct collect_ctx {
    uint64_t *end;
    uint32_t *id;
    size_t cap, n;
} collect_ctx_t;

static uint64_t collect_cb(size_t end, uint32_t id, void *c_) {
    collect_ctx_t *c = (collect_ctx_t *)c_;
    if (c->n < c->cap) {
        c->end[c->n] = end;
        c->id[c->n] = id;
    }
    c->n++;
    return ~0ULL;
}

/* Convenience: all matches of one block, in delivery order. Returns the total
 * count (may exceed cap; only the first cap are stored). */
size_t hso_collect(const hso_table_t *t, const uint8_t *buf, size_t len, size_t start,
                   uint64_t groups, uint64_t *out_end, uint32_t *out_id, size_t cap) {
    collect_ctx_t c = {out_end, out_id, cap, 0};
    hso_exec(t, buf, len, start, collect_cb, &c, groups);
    return c.n;
}

static uint64_t count_cb(size_t end, uint32_t id, void *c) {
    (void)end;
    (void)id;
    ++*(uint64_t *)c;
    return ~0ULL;
}

/* hsbench-style block loop with a counting callback. */
uint64_t hso_count_blocks(const hso_table_t *t, const uint8_t *base, const uint64_t *off,
                          size_t nblocks, size_t start, uint64_t groups) {
    uint64_t n = 0;
    for (size_t i = 0; i < nblocks; i++) {
        hso_exec(t, base + off[i], (size_t)(off[i + 1] - off[i]), start, count_cb, &n, groups);
    }
    return n;
}

/* Batched form: records (block, end, id) for all blocks, delivery order. */
size_t hso_collect_blocks(const hso_table_t *t, const uint8_t *base, const uint64_t *off,
                          size_t nblocks, size_t start, uint64_t groups, uint32_t *out_block,
                          uint32_t *out_end, uint32_t *out_id, size_t cap) {
    size_t total = 0;
    size_t tmp_cap = 1 << 16;
    uint64_t *te = (uint64_t *)malloc(tmp_cap * sizeof(uint64_t));
    uint32_t *ti = (uint32_t *)malloc(tmp_cap * sizeof(uint32_t));
    for (size_t b = 0; b < nblocks; b++) {
        size_t blen = (size_t)(off[b + 1] - off[b]);
        size_t n = hso_collect(t, base + off[b], blen, start, groups, te, ti, tmp_cap);
        if (n > tmp_cap) {
            tmp_cap = n;
            te = (uint64_t *)realloc(te, tmp_cap * sizeof(uint64_t));
            ti = (uint32_t *)realloc(ti, tmp_cap * sizeof(uint32_t));
            n = hso_collect(t, base + off[b], blen, start, groups, te, ti, tmp_cap);
        }
        for (size_t k = 0; k < n; k++) {
            if (total < cap) {
                out_block[total] = (uint32_t)b;
                out_end[total] = (uint32_t)te[k];
                out_id[total] = ti[k];
            }
            total++;
        }
    }
    free(te);
    free(ti);
    return total;
}

/* ------------------------------------------------------------------------
 * Character-class accelerators (scalar "first byte in class" semantics).
 * All forward scans return len when nothing is found; reverse scans return -1.
 * ------------------------------------------------------------------------ */

/* shufti: member iff lo[c & 15] & hi[c >> 4] != 0 (shufti.c:75-87) */
int64_t hso_shufti_fwd(const uint8_t lo[16], const uint8_t hi[16], const uint8_t *buf, size_t len) {
    for (size_t i = 0; i < len; i++) {
        if (lo[buf[i] & 15] & hi[buf[i] >> 4]) return (int64_t)i;
    }
    return (int64_t)len;
}
int64_t hso_shufti_rev(const uint8_t lo[16], const uint8_t hi[16], const uint8_t *buf, size_t len) {
    for (size_t i = len; i-- > 0;) {
        if (lo[buf[i] & 15] & hi[buf[i] >> 4]) return (int64_t)i;
    }
    return -1;
}

/* truffle: mask1 covers bytes 0x00-0x7f, mask2 bytes 0x80-0xff; bit (c>>4)&7 of
 * mask[c & 15] set <=> member (trufflecompile.cpp:59-72, truffle2cr :77-94) */
static int truffle_member(const uint8_t m1[16], const uint8_t m2[16], uint8_t c) {
    const uint8_t *m = (c & 0x80) ? m2 : m1;
    return (m[c & 15] >> ((c >> 4) & 7)) & 1;
}
void hso_truffle_build(const uint8_t bitmap[32], uint8_t m1[16], uint8_t m2[16]) {
    memset(m1, 0, 16);
    memset(m2, 0, 16);
    for (unsigned c = 0; c < 256; c++) {
        if (bitmap[c / 8] & (1u << (c % 8))) {
            uint8_t *m = (c & 0x80) ? m2 : m1;
            m[c & 15] |= (uint8_t)(1u << ((c >> 4) & 7));
        }
    }
}
int64_t hso_truffle_fwd(const uint8_t m1[16], const uint8_t m2[16], const uint8_t *buf, size_t len) {
    for (size_t i = 0; i < len; i++) {
        if (truffle_member(m1, m2, buf[i])) return (int64_t)i;
    }
    return (int64_t)len;
}
int64_t hso_truffle_rev(const uint8_t m1[16], const uint8_t m2[16], const uint8_t *buf, size_t len) {
    for (size_t i = len; i-- > 0;) {
        if (truffle_member(m1, m2, buf[i])) return (int64_t)i;
    }
    return -1;
}

/* generic: membership by 256-bit class bitmap (what truffle2cr/shufti2cr decode to) */
int64_t hso_class_fwd(const uint8_t bitmap[32], const uint8_t *buf, size_t len) {
    for (size_t i = 0; i < len; i++) {
        if (bitmap[buf[i] >> 3] & (1u << (buf[i] & 7))) return (int64_t)i;
    }
    return (int64_t)len;
}
int64_t hso_class_rev(const uint8_t bitmap[32], const uint8_t *buf, size_t len) {
    for (size_t i = len; i-- > 0;) {
        if (bitmap[buf[i] >> 3] & (1u << (buf[i] & 7))) return (int64_t)i;
    }
    return -1;
}
/* membership bitmap: bit i of out (LSB-first within each byte) = buf[i] in class */
void hso_class_bitmap(const uint8_t bitmap[32], const uint8_t *buf, size_t len, uint8_t *out) {
    memset(out, 0, (len + 7) / 8);
    for (size_t i = 0; i < len; i++) {
        if (bitmap[buf[i] >> 3] & (1u << (buf[i] & 7))) out[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}

/* vermicelli (vermicelli.h:42-104): first occurrence of c (caseless: compare
 * with 0xdf mask when c is alpha); nverm = first byte that is NOT c. */
int64_t hso_verm_fwd(uint8_t c, int nocase, int negate, const uint8_t *buf, size_t len) {
    uint8_t mask = (nocase && is_alpha(c)) ? 0xdf : 0xff;
    uint8_t cv = c & mask;
    for (size_t i = 0; i < len; i++) {
        int eq = (buf[i] & mask) == cv;
        if (eq != negate) return (int64_t)i;
    }
    return (int64_t)len;
}
int64_t hso_verm_rev(uint8_t c, int nocase, int negate, const uint8_t *buf, size_t len) {
    uint8_t mask = (nocase && is_alpha(c)) ? 0xdf : 0xff;
    uint8_t cv = c & mask;
    for (size_t i = len; i-- > 0;) {
        int eq = (buf[i] & mask) == cv;
        if (eq != negate) return (int64_t)i;
    }
    return -1;
}
/* double vermicelli (vermicelli.h:106-170): first i with buf[i]==c1 && buf[i+1]==c2;
 * if the final byte equals c1 the reference returns its position (partial match
 * that the caller must re-examine), else len. */
int64_t hso_dverm_fwd(uint8_t c1, uint8_t c2, int nocase, const uint8_t *buf, size_t len) {
    uint8_t m1 = (nocase && is_alpha(c1)) ? 0xdf : 0xff, m2 = (nocase && is_alpha(c2)) ? 0xdf : 0xff;
    for (size_t i = 0; i + 1 < len; i++) {
        if ((buf[i] & m1) == (c1 & m1) && (buf[i + 1] & m2) == (c2 & m2)) return (int64_t)i;
    }
    if (len && (buf[len - 1] & m1) == (c1 & m1)) return (int64_t)len - 1;
    return (int64_t)len;
}

/* masked double vermicelli (vermicelli.h:241-317): first i with (buf[i] & m1) == c1 and
 * (buf[i+1] & m2) == c2; a final byte with (b & m1) == c1 is reported as a partial match. */
int64_t hso_dverm_masked_fwd(uint8_t c1, uint8_t c2, uint8_t m1, uint8_t m2, const uint8_t *buf, size_t len) {
    for (size_t i = 0; i + 1 < len; i++) {
        if ((buf[i] & m1) == c1 && (buf[i + 1] & m2) == c2) return (int64_t)i;
    }
    if (len && (buf[len - 1] & m1) == c1) return (int64_t)len - 1;
    return (int64_t)len;
}
/* reverse double vermicelli (vermicelli.h:464-518, vermicelli_sse.h:325-341): the position of
 * the SECOND byte of the last c1 c2 pair (unit/internal/rvermicelli.cpp:116-140), -1 if none;
 * no partial-match rule in this direction. */
int64_t hso_rdverm(uint8_t c1, uint8_t c2, int nocase, const uint8_t *buf, size_t len) {
    uint8_t m1 = (nocase && is_alpha(c1)) ? 0xdf : 0xff, m2 = (nocase && is_alpha(c2)) ? 0xdf : 0xff;
    for (size_t i = len; i-- > 1;) {
        if ((buf[i - 1] & m1) == (c1 & m1) && (buf[i] & m2) == (c2 & m2)) return (int64_t)i;
    }
    return -1;
}
/* The reference's own return value for the reverse double scan. It is a reverse
 * ACCELERATOR: the result r promises "no pair ends in (r, len)"; the head of the buffer
 * below the last 16-byte boundary it visits is simply not examined and that boundary is
 * returned (rdvermSearchAligned's final `return buf_end`, vermicelli_sse.h:325-341), so r is
 * never below hso_rdverm and equals it whenever the pair lies in the examined part.
 * `align` = buf mod 16. Needs len >= 16 (vermicelli.h:488). */
int64_t hso_rdverm_ref_model(uint8_t c1, uint8_t c2, int nocase, const uint8_t *buf, size_t len, unsigned align) {
    /* "nonalphas and nocase having interesting behaviour": caseless compares BOTH bytes with 0xdf */
    const uint8_t m = nocase ? 0xdf : 0xff;
    int64_t end = (int64_t)len;
    const unsigned min = (unsigned)((align + len) & 15);
    if (min) { /* unaligned tail vector [end-16, end): pairs inside it only (vermicelli_sse.h:367-378) */
        for (int64_t i = end - 1; i >= end - 15; i--)
            if ((buf[i] & m) == c2 && (buf[i - 1] & m) == c1) return i;
        end -= min;
        if (end <= 0) return end;
    }
    for (; 16 < end; end -= 16) { /* aligned vectors, plus the pair straddling the vector's start */
        for (int64_t i = end - 1; i >= end - 16; i--)
            if ((buf[i] & m) == c2 && (buf[i - 1] & m) == c1) return i;
    }
    return end;
}

/* double shufti (shufti.c:205-236 fwdBlock2, :286-361): masks are 0-active, one bit per
 * bucket (<= 8 buckets of byte-pair "rectangles", shufticompile.cpp:135-209):
 *   t(c) = lo1[c & 15] | hi1[c >> 4],  u(c) = lo2[c & 15] | hi2[c >> 4]
 * the pair (buf[i], buf[i+1]) matches iff (t(buf[i]) | u(buf[i+1])) != 0xff.
 * EXACT form: the first matching i, else len - 1 if the last byte alone passes t (a partial
 * match the caller re-examines, as in double vermicelli), else len. */
static uint8_t dsh_t(const uint8_t lo[16], const uint8_t hi[16], uint8_t c) { return lo[c & 15] | hi[c >> 4]; }
int64_t hso_dshufti_fwd(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                        const uint8_t hi2[16], const uint8_t *buf, size_t len) {
    for (size_t i = 0; i + 1 < len; i++) {
        if ((dsh_t(lo1, hi1, buf[i]) | dsh_t(lo2, hi2, buf[i + 1])) != 0xff) return (int64_t)i;
    }
    if (len && dsh_t(lo1, hi1, buf[len - 1]) != 0xff) return (int64_t)len - 1;
    return (int64_t)len;
}
/* The reference's own return value, which also depends on where its vectors end: the last
 * byte of every 128-bit lane it loads is tested with t alone (rshiftbyte_m128 /
 * rshift128_m256 shift a zero = "all buckets alive" into lane byte 15, shufti.c:224,647), so
 * a first-byte-only hit there ends the scan early. Vectors of `width` bytes (16: SSE build,
 * shufti.c:319-361; 32: AVX2 build, :696-752): [buf, buf+width), then width-aligned ones below
 * buf_end - width, then [buf_end-width, buf_end); the AVX2 build scans buffers shorter than 32
 * as two 16-byte vectors (shuftiDoubleShort, :676-694). `align` = buf mod 64.
 * Conservative either way: never later than hso_dshufti_fwd. Needs len >= 16. */
static int64_t dsh_vec(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16], const uint8_t hi2[16],
                       const uint8_t *buf, size_t v, size_t width) {
    for (size_t j = 0; j < width; j++) {
        const size_t i = v + j;
        const uint8_t t = dsh_t(lo1, hi1, buf[i]);
        const uint8_t u = (j & 15) < 15 ? dsh_t(lo2, hi2, buf[i + 1]) : 0;
        if ((t | u) != 0xff) return (int64_t)i;
    }
    return -1;
}
int64_t hso_dshufti_ref_model(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                              const uint8_t hi2[16], const uint8_t *buf, size_t len, unsigned align,
                              unsigned width) {
    int64_t r;
    if (width == 32 && len < 32) {
        if ((r = dsh_vec(lo1, hi1, lo2, hi2, buf, 0, 16)) >= 0) return r;
        if ((r = dsh_vec(lo1, hi1, lo2, hi2, buf, len - 16, 16)) >= 0) return r;
        return (int64_t)len;
    }
    if ((r = dsh_vec(lo1, hi1, lo2, hi2, buf, 0, width)) >= 0) return r;
    size_t pos = width - (align % width);
    while (pos + width < len) { /* buf < last_block */
        if ((r = dsh_vec(lo1, hi1, lo2, hi2, buf, pos, width)) >= 0) return r;
        pos += width;
    }
    if ((r = dsh_vec(lo1, hi1, lo2, hi2, buf, len - width, width)) >= 0) return r;
    return (int64_t)len;
}
/* pair-membership bitmap: bit i <=> (buf[i], buf[i+1]) matches, i + 1 < len */
void hso_dshufti_bitmap(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                        const uint8_t hi2[16], const uint8_t *buf, size_t len, uint8_t *out) {
    memset(out, 0, (len + 7) / 8);
    for (size_t i = 0; i + 1 < len; i++) {
        if ((dsh_t(lo1, hi1, buf[i]) | dsh_t(lo2, hi2, buf[i + 1])) != 0xff) out[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}
/* last pair in the block: position of its second byte (the reverse double-vermicelli
 * convention), -1 if none */
int64_t hso_dshufti_rev(const uint8_t lo1[16], const uint8_t hi1[16], const uint8_t lo2[16],
                        const uint8_t hi2[16], const uint8_t *buf, size_t len) {
    for (size_t i = len; i-- > 1;) {
        if ((dsh_t(lo1, hi1, buf[i - 1]) | dsh_t(lo2, hi2, buf[i])) != 0xff) return (int64_t)i;
    }
    return -1;
}
