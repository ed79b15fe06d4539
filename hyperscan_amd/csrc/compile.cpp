/*
 * compile.cpp -- host-side literal compiler for the GPU scan engine.
 *
 * Plays the role of the reference's hwlmBuildProto/hwlmBuild
 * (src/hwlm/hwlm_build.cpp:121-215) and of the FDR/Teddy table and confirm
 * compilers behind it (src/fdr/fdr_compile.cpp:589-632 setupTab,
 * src/fdr/teddy_compile.cpp:439-509 fillNibbleMasks,
 * src/fdr/fdr_confirm_compile.cpp:73-290), but produces this engine's own
 * tables (table.h) -- the reference's bucket/domain/stride machinery is an x86
 * cache-and-PSHUFB design and is deliberately not reproduced.
 *
 * What IS reproduced exactly is the match predicate: each literal becomes the
 * (v, msk, size) triple of the reference's LitInfo (fillLitInfo,
 * fdr_confirm_compile.cpp:73-127), so that "(conf_key & msk) == v" on the eight
 * bytes ending at `end` decides a match in both implementations.
 *
 * Filter design: every literal is keyed on its last 4 (class A), 3 (class B) or
 * <= 2 (class C) bytes, whichever keeps the number of concrete key variants
 * (case bits / mask wildcards enumerated) small. Each variant sets one bit in
 * the hashed LDS filter and owns an entry in an exact hash table that lists the
 * literals to confirm.
 */
#include "internal.h"
#include "../../include/hsgpu_tuning.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <unordered_map>

static thread_local std::string g_last_error;

void hsgpu_set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *hsgpu_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *hsgpu_version(void) { return "hsgpu 0.6 (gfx950)"; }

namespace {

bool is_alpha(uint8_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

uint64_t right_aligned_u64(const uint8_t *p, uint32_t len) {
    uint64_t m = 0;
    uint32_t n = std::min<uint32_t>(len, 8);
    memcpy((uint8_t *)&m + 8 - n, p + len - n, n);
    return m;
}

/* hwlmLiteral constructor checks (hwlm_literal.cpp:57-115) + LitInfo fill. */
int normalise(const hsgpu_lit_t &l, size_t index, HsgpuDevLit &o) {
    if (l.len > HSGPU_LITERAL_MAX_LEN || l.msk_len > HSGPU_MASKLEN) {
        hsgpu_set_error("literal %zu: length %u / mask length %u exceeds 8", index, l.len, l.msk_len);
        return HSGPU_COMPILER_ERROR;
    }
    if ((l.len && !l.s) || (l.msk_len && (!l.msk || !l.cmp))) {
        hsgpu_set_error("literal %zu: null pointer", index);
        return HSGPU_COMPILER_ERROR;
    }
    if (l.id == 0xffffffffu) { /* reserved, hwlm_build.cpp:190 */
        hsgpu_set_error("literal %zu: id 0xffffffff is reserved", index);
        return HSGPU_COMPILER_ERROR;
    }
    uint32_t mlen = l.msk_len;
    bool all_zero = true;
    for (uint32_t j = 0; j < mlen; j++) all_zero &= (l.msk[j] == 0);
    if (all_zero) mlen = 0;
    if (l.len == 0 && mlen == 0) {
        hsgpu_set_error("literal %zu: empty literal", index);
        return HSGPU_COMPILER_ERROR;
    }
    uint64_t msk = ~0ULL, val = 0;
    for (uint32_t j = 0; j < 8; j++) {
        uint32_t sh = (7 - j) * 8;
        if (j >= l.len) {
            msk &= ~(0xffULL << sh);
        } else {
            uint8_t c = l.s[l.len - j - 1];
            if (l.nocase && is_alpha(c)) {
                msk &= ~(0x20ULL << sh);
                val |= (uint64_t)(c & 0xdf) << sh;
            } else {
                val |= (uint64_t)c << sh;
            }
        }
    }
    if (mlen) {
        uint64_t lm = right_aligned_u64(l.msk, mlen), lc = right_aligned_u64(l.cmp, mlen);
        /* maskIsConsistent (hwlm_literal.cpp:57-79): where both constrain a bit
         * they must agree; cmp may not set bits outside msk. */
        uint64_t both = lm & msk;
        if ((val & both) != (lc & both) || (lc & ~lm)) {
            hsgpu_set_error("literal %zu: msk/cmp inconsistent with the literal", index);
            return HSGPU_COMPILER_ERROR;
        }
        msk |= lm;
        val |= lc;
    }
    o.v = val;
    o.msk = msk;
    o.groups = l.groups;
    o.id = l.id;
    o.size = (uint8_t)std::max(mlen, l.len);
    o.flags = l.noruns ? HSGPU_LIT_NORUNS : 0;
    o.pad = 0;
    return HSGPU_SUCCESS;
}

struct KeyChoice {
    int cls; /* 0 = A (4 bytes), 1 = B (3 bytes), 2 = C (2 bytes) */
};

inline uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

KeyChoice choose_class(const HsgpuDevLit &l) {
    /* the longest key that needs at most 16 concrete variants (4 wildcard bits);
     * failing that, the key with the fewest variants */
    uint32_t m4 = (uint32_t)(l.msk >> 32), m3 = m4 >> 8, m2 = m4 >> 16;
    uint32_t wa = popc(~m4), wb = popc(~m3 & 0xffffffu), wc = popc(~m2 & 0xffffu);
    if (wa <= 4) return {0};
    if (wb <= 4) return {1};
    if (wc <= 4) return {2};
    if (wa <= wb && wa <= wc) return {0};
    if (wb <= wc) return {1};
    return {2};
}

/* call f(key) for every value of (v | sub) over all submasks sub of wild */
template <class F> void for_each_variant(uint32_t v, uint32_t wild, F f) {
    uint32_t sub = 0;
    do {
        f(v | sub);
        sub = (sub - wild) & wild;
    } while (sub != 0);
}

uint32_t ceil_log2(uint64_t x) {
    uint32_t k = 0;
    while ((1ULL << k) < x) k++;
    return k;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

/* ---- the pair filter (table.h, HSGPU_F_PAIR) ------------------------------------------------ */

constexpr uint32_t PAIR_MAX_VARIANTS = 1u << 12; /* filter entries one literal may need per parity */

/* byte of the literal at distance p from its end (p = 0: last byte) and its compare mask */
inline void lit_byte(const HsgpuDevLit &l, int p, uint32_t &v, uint32_t &m) {
    if (p < 0 || p > 7) {
        v = m = 0;
        return;
    }
    v = (uint32_t)(l.v >> (8 * (7 - p))) & 0xffu;
    m = (uint32_t)(l.msk >> (8 * (7 - p))) & 0xffu;
}

/* One literal at one delta (it ends at q + delta): the 24-bit hash input and its unknown bits, the B-plane
 * bits of every admissible byte before it, the A-plane bits of every admissible byte after it. */
struct PairKey {
    uint32_t val, unk; /* hash input under hash_mask; bits to enumerate */
    uint32_t bbits, abits;
    bool use_a;
    uint32_t variants() const { return 1u << popc(unk); }
};

PairKey pair_key(const HsgpuDevLit &l, int delta, uint32_t hash_mask) {
    PairKey k;
    k.val = k.unk = 0;
    for (int i = 0; i < 3; i++) { /* b2 (i = 2) is the low byte of the hash input */
        uint32_t v, m;
        lit_byte(l, delta + i, v, m);
        const uint32_t rel = (hash_mask >> (8 * (2 - i))) & 0xffu;
        k.val |= (v & m & rel) << (8 * (2 - i));
        k.unk |= (rel & ~m) << (8 * (2 - i));
    }
    uint32_t v3, m3, vn, mn;
    lit_byte(l, delta + 3, v3, m3);
    lit_byte(l, delta - 1, vn, mn);
    k.bbits = k.abits = 0;
    for (uint32_t b = 0; b < 256; b++) {
        if ((b & m3) == (v3 & m3)) k.bbits |= 1u << hsgpu_pair_bit_b1(b) | 1u << hsgpu_pair_bit_b2(b);
        if ((b & mn) == (vn & mn)) k.abits |= 1u << hsgpu_pair_bit_a(b);
    }
    /* which plane says more about this key: two probes of B against one of A */
    const uint32_t pb = popc(k.bbits), pa = popc(k.abits);
    k.use_a = pa * 32 < pb * pb;
    return k;
}

} // namespace

/* Adler-32 over the whole image, the header included (its checksum field read as zero): a
 * damaged geometry field must not get as far as the device */
uint32_t hsgpu_blob_checksum(const uint8_t *blob, size_t len) {
    uint32_t a = 1, b = 0;
    const size_t c0 = offsetof(HsgpuTableHeader, checksum), c1 = c0 + sizeof(uint32_t);
    for (size_t i = 0; i < len; i++) {
        a = (a + ((i >= c0 && i < c1) ? 0u : blob[i])) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}

int hsgpu_compile_table(const hsgpu_lit_t *lits, size_t n, unsigned flags, std::vector<uint8_t> &blob) {
    if (!lits || n == 0) {
        hsgpu_set_error("no literals");
        return HSGPU_COMPILER_ERROR;
    }
    if (n > (1u << 24)) {
        hsgpu_set_error("too many literals (%zu)", n);
        return HSGPU_COMPILER_ERROR;
    }
    std::vector<HsgpuDevLit> dl(n);
    for (size_t i = 0; i < n; i++) {
        int rv = normalise(lits[i], i, dl[i]);
        if (rv != HSGPU_SUCCESS) return rv;
    }

    /* Stride. With stride 2 the kernel looks up every second byte position q only
     * (FDR has the same idea, fdr.c:246-327): a lookup at q must then catch
     * literals ending at q (delta 0, keyed on their last bytes) AND at q + 1
     * (delta 1, keyed on the bytes before their last one). Needs every literal to
     * have at least one byte left at delta 1. */
    uint32_t min_size = 8;
    for (size_t i = 0; i < n; i++) min_size = std::min<uint32_t>(min_size, dl[i].size);

    /* Case-blind keys: when any literal is caseless, hash and exact-table keys drop
     * bit 5 of every byte (one v_and per lookup), so a caseless literal needs ONE
     * key instead of one per case variant; the exact compare still decides. */
    bool blind = (flags & HSGPU_BUILD_FORCE_BLIND) != 0;
    for (size_t i = 0; i < n && !blind; i++) blind = lits[i].nocase != 0;
    const uint32_t blind4 = blind ? 0x20202020u : 0u;

    /* The pair filter (table.h): a stride-2 layout that holds short literals too. Every literal is keyed at
     * delta 0 and at delta +1 or -1, whichever needs fewer filter entries. Sets it cannot hold (a literal
     * that needs thousands of entries, or one keyed on two bytes) are refused under FORCE_PAIR.
     * Opt-in, not the default: on the 10 000-literal snort-like set the 3-byte literals (each owns whole filter
     * planes at one parity and 32 entries at the other) leave it with about twice the candidates of the
     * stride-1 layout, which costs more in the spill and the confirm step than the halved lookups save. */
    bool pair = (flags & HSGPU_BUILD_FORCE_PAIR) != 0;
    std::vector<int8_t> odd_delta(n, 1);
    uint32_t pair_hash_mask = 0;
    if (pair) {
        const uint32_t hm_late = blind ? 0x1fdfdfu : 0x1fffffu; /* 5 bits of b0: a key at delta -1 enumerates them */
        bool any_late = false;
        for (size_t i = 0; i < n && pair; i++) {
            const uint32_t c0 = pair_key(dl[i], 0, hm_late).variants(), cp = pair_key(dl[i], 1, hm_late).variants(),
                           cm = pair_key(dl[i], -1, hm_late).variants();
            odd_delta[i] = cm < cp ? -1 : 1;
            any_late |= odd_delta[i] < 0;
            if (c0 > PAIR_MAX_VARIANTS || std::min(cp, cm) > PAIR_MAX_VARIANTS) pair = false;
            /* its exact-table keys must be 3- or 4-byte ones (class A / B) */
            HsgpuDevLit l = dl[i];
            l.msk |= (uint64_t)blind4 << 32;
            if (choose_class(l).cls == 2) pair = false;
            if (odd_delta[i] > 0) {
                l = dl[i];
                l.v <<= 8;
                l.msk = l.msk << 8 | (uint64_t)blind4 << 32;
                if (choose_class(l).cls == 2) pair = false;
            } else if (popc(~(uint32_t)(dl[i].msk >> 40) & 0xffffffu & ~(blind4 >> 8)) > 4) {
                pair = false; /* a late key with more than 16 variants */
            }
        }
        if (!pair && (flags & HSGPU_BUILD_FORCE_PAIR)) {
            hsgpu_set_error("the pair filter cannot hold this literal set");
            return HSGPU_COMPILER_ERROR;
        }
        pair_hash_mask = any_late ? hm_late : (blind ? 0xffdfdfu : 0xffffffu);
    }

    /* below 4 bytes the delta-1 key would be a bare 2-gram: too many candidates */
    const bool stride2 = pair || (!(flags & HSGPU_BUILD_FORCE_STRIDE1) && (min_size >= 4 || ((flags & HSGPU_BUILD_FORCE_STRIDE2) && min_size >= 2)));
    const uint32_t n_delta = stride2 ? 2 : 1;

    /* key -> (literal index | delta << 30), per class (literal order preserved). Exact-table keys: the last
     * 4 (A) / 3 (B) / 2 (C) bytes of the window ending at q; a literal keyed one byte late (pair filter,
     * delta -1) has a 3-byte key of the window ending at q - 1, marked HSGPU_KEY_M, and no delta bit. */
    std::unordered_map<uint32_t, std::vector<uint32_t>> keys[3];
    uint32_t n_cls[3] = {0, 0, 0}, max_size = 0, n_late = 0;
    for (size_t i = 0; i < n; i++) {
        max_size = std::max<uint32_t>(max_size, dl[i].size);
        for (uint32_t k = 0; k < n_delta; k++) {
            const int delta = k == 0 ? 0 : (pair ? odd_delta[i] : 1);
            HsgpuDevLit l = dl[i];
            if (delta > 0) {
                l.v <<= 8;
                l.msk <<= 8;
            }
            /* blind: bit 5 is neither a constraint nor a wildcard to enumerate */
            l.msk |= (uint64_t)blind4 << 32;
            l.v &= ~((uint64_t)blind4 << 32);
            uint32_t m4 = (uint32_t)(l.msk >> 32), v4 = (uint32_t)(l.v >> 32);
            if (delta < 0) {
                const uint32_t wild = ~(m4 >> 8) & 0xffffffu; /* <= 4 bits: checked when the layout was chosen */
                n_late++;
                for_each_variant(v4 >> 8, wild, [&](uint32_t key) { keys[1][key | HSGPU_KEY_M].push_back((uint32_t)i); });
                continue;
            }
            int cls = choose_class(l).cls;
            if (delta == 0) n_cls[cls]++;
            uint32_t v, wild;
            if (cls == 0) {
                v = v4;
                wild = ~m4;
            } else if (cls == 1) {
                v = v4 >> 8;
                wild = ~(m4 >> 8) & 0xffffffu;
            } else {
                v = v4 >> 16;
                wild = ~(m4 >> 16) & 0xffffu;
            }
            const uint32_t ent = (uint32_t)i | (uint32_t)delta << HSGPU_LIST_DELTA_SHIFT;
            for_each_variant(v, wild, [&](uint32_t key) { keys[cls][key].push_back(ent); });
        }
    }
    const uint32_t entries = (uint32_t)(keys[0].size() + keys[1].size());
    uint32_t tflags = (keys[0].size() ? HSGPU_F_HAS_A : 0) | (keys[1].size() ? HSGPU_F_HAS_B : 0) |
                      (keys[2].size() ? HSGPU_F_HAS_C : 0) | (stride2 ? HSGPU_F_STRIDE2 : 0) |
                      (blind ? HSGPU_F_BLIND : 0) | (pair ? HSGPU_F_PAIR : 0);
    /* Filter layout.
     *  - small sets ("Teddy class", <= 1024 key variants): bank-replicated rows; every
     *    lane reads its own LDS bank, so lookups are conflict-free. 128 bits per
     *    entry per copy, between 256 rows (32 KiB) and 1024 rows (128 KiB).
     *  - large sets ("FDR class"): one hashed 2^k-word filter, k <= 15 (128 KiB),
     *    two bits per class-A key once the filter is more than ~0.4% full. */
    uint32_t k;
    /* replicated only pays at stride 1, where the lookup rate is high enough for LDS
     * bank conflicts to matter; at stride 2 the hashed table's 32x more bits win */
    if (pair) {
        /* 2^k entries of 8 bytes: 128 KiB at k = 14 */
        k = (flags & HSGPU_BUILD_FORCE_SMALL) ? 12 : (flags & HSGPU_BUILD_FORCE_MEDIUM) ? 13 : 14;
    } else if (!(flags & HSGPU_BUILD_FORCE_HASHED) && ((entries <= 1024 && !stride2) || (flags & HSGPU_BUILD_FORCE_REPL))) {
        tflags |= HSGPU_F_REPL;
        /* 160 KiB of LDS = 128 KiB filter + 8 KiB 2-byte table + 24 KiB per-wavefront areas (fused kernel) */
        k = 10;
        if ((uint64_t)entries * 128 > ((uint64_t)32 << k)) tflags |= HSGPU_F_K2; /* > 0.8% of bits set */
    } else {
        /* the full 128 KiB that one 16-wavefront workgroup per CU can hold: a smaller
         * table with three 8-wavefront workgroups per CU measured slower */
        k = (flags & HSGPU_BUILD_FORCE_SMALL) ? 13 : (flags & HSGPU_BUILD_FORCE_MEDIUM) ? 14 : 15;
        if ((uint64_t)entries * 128 > ((uint64_t)32 << k)) tflags |= HSGPU_F_K2;
    }
    if (flags & HSGPU_BUILD_FORCE_K2) tflags |= HSGPU_F_K2;
    if ((flags & HSGPU_BUILD_FORCE_K1) || pair) tflags &= ~HSGPU_F_K2;
    const uint32_t fwords = hsgpu_filter_words(tflags, k);
    /* Few 3-byte keys beside 4-byte ones at stride 1 (a handful of short literals in a large
     * set): give each its whole filter word. The filter kernel then runs ONE class test per
     * lookup -- at stride 1 it is VALU-bound and the 3-byte test was ~30% of its instructions --
     * and the price is a false-positive rate of keys / words <= 1% of lookups. At stride 2 only a
     * handful of such keys are folded (<= 0.1% of the words: the 4-byte literals of a 64-literal set at
     * their odd alignment); with hundreds of them the extra candidates cost more than the test
     * (measured, 1000 literals: 0.30 ms folded vs 0.27 ms not). */
    if (!pair && !(tflags & (HSGPU_F_REPL | HSGPU_F_HAS_C)) && (tflags & HSGPU_F_HAS_A) && (tflags & HSGPU_F_HAS_B) && /* (a pair filter is filled from the literals: nothing to fold) */
        !(flags & HSGPU_BUILD_NO_FOLD) && (uint64_t)keys[1].size() * ((tflags & HSGPU_F_STRIDE2) ? 1000 : 100) <= fwords)
        tflags |= HSGPU_F_BFOLD;
    /* The key gate: while no 2-byte table and no pair gate needs the 64 Kbit section, and the set leaves it mostly
     * empty, it says which exact-table keys exist at all (see HSGPU_F_GATE). Measured on the 10 000-literal set: 82 %
     * fewer table probes, confirm stage 0.201 -> 0.173 ms. Small sets have few candidates and nothing to gain from it
     * (64 literals: the 8 KiB staged per workgroup cost 3 us of a 17 us kernel). */
    if (!pair && !(tflags & HSGPU_F_HAS_C) && !(flags & HSGPU_BUILD_NO_GATE) && entries >= 2048 && (uint64_t)entries * 2 <= 65536)
        tflags |= HSGPU_F_GATE;
    /* Round 5, opt-in: at stride 1 (every key is a delta-0 key) the gate can be a Bloom filter over the FULL-window keys instead
     * -- table.h, HSGPU_F_BLOOM; it takes the gate's place (12 KiB of LDS instead of 8). Built because the verdict of round 4 asked
     * for it and the simulator promised 70 % fewer bucket reads (tools/sim/bloomgate.py: A probes 3.64 M -> 0.69 M per GiB, which
     * the device confirms by delivering identical records); not the default because the stage is no faster for it: 0.1747 ms against
     * 0.1667 with the key gate, three alternating runs on one box (profiles/r05_bloom_gate_ab.txt). The stage is bound by the
     * length of a step's dependent chain at six wavefronts per SIMD, not by its L2 requests; three more LDS round trips per
     * candidate lengthen the chain. */
    if ((tflags & HSGPU_F_GATE) && !stride2 && (flags & HSGPU_BUILD_FORCE_BLOOM)) tflags = (tflags & ~HSGPU_F_GATE) | HSGPU_F_BLOOM;
    const size_t c2words = (tflags & HSGPU_F_BLOOM) ? HSGPU_BLOOM_WORDS : 2048;
    /* 64-bit entries for the large stride-1 two-bit sets whose kernel runs the 4-byte-key test alone (no 3-byte keys, or
     * folded ones): same 128 KiB, the second bit in a word of its own. Simulated on the bench's 10 000-literal set
     * (tools/sim/b2p.py): 11.5 M -> 8.8 M candidate lanes per GiB (a folded 3-byte key no longer passes every position
     * that hashes into its word), and the second test is one instruction (a shift by a byte of the hash) instead of two. */
    if (!pair && !(flags & HSGPU_BUILD_NO_WIDE) && !(tflags & (HSGPU_F_REPL | HSGPU_F_HAS_C | HSGPU_F_STRIDE2)) && (tflags & HSGPU_F_K2) &&
        (!(tflags & HSGPU_F_HAS_B) || (tflags & HSGPU_F_BFOLD)) && k == 15) {
        tflags |= HSGPU_F_WIDE;
        k = 14; /* entries; the same 2^15 words */
    }
    const uint32_t fshift = hsgpu_filter_shift(tflags, k);
    uint32_t ht_log2[2];
    for (int c = 0; c < 2; c++)
        ht_log2[c] = std::min<uint32_t>(24, std::max<uint32_t>(2, ceil_log2((uint64_t)keys[c].size() * 2 + 1))); /* <= 0.5 keys per 4-slot bucket: a full bucket (probe continues) is a < 0.2% event */

    /* lists */
    std::vector<uint32_t> lists;
    auto add_list = [&](const std::vector<uint32_t> &v) -> uint32_t {
        if (v.size() == 1) return v[0] | HSGPU_REF_DIRECT; /* literal | delta << 30 */
        uint32_t ref = (uint32_t)lists.size() + 1;
        if (ref + v.size() > HSGPU_LIST_LIT_MASK) throw std::bad_alloc();
        for (size_t j = 0; j < v.size(); j++) lists.push_back(v[j] | (j + 1 == v.size() ? HSGPU_LIST_END : 0));
        return ref;
    };

    /* layout */
    size_t off = sizeof(HsgpuTableHeader);
    HsgpuTableHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = HSGPU_TABLE_MAGIC;
    h.version = HSGPU_TABLE_VERSION;
    h.flags = tflags;
    h.n_lits = (uint32_t)n;
    h.max_size = max_size;
    h.filter_log2 = k;
    h.filter_entries = entries;
    h.ht_a_log2 = ht_log2[0];
    h.ht_b_log2 = ht_log2[1];
    h.n_a = n_cls[0];
    h.n_b = n_cls[1];
    h.n_c = n_cls[2];
    h.hash_mask = pair_hash_mask;
    h.n_m = n_late;
    h.off_filter = (uint32_t)off;
    off += (size_t)4 * fwords;
    h.off_c2bits = (uint32_t)off;
    off += c2words * 4;
    h.off_ht_a = (uint32_t)off;
    off += (sizeof(HsgpuHtSlot) * HSGPU_BUCKET_SLOTS) << ht_log2[0];
    h.off_ht_b = (uint32_t)off;
    off += (sizeof(HsgpuHtSlot) * HSGPU_BUCKET_SLOTS) << ht_log2[1];
    h.off_c2ref = (uint32_t)off;
    off += (tflags & HSGPU_F_HAS_C) ? 65536 * 4 : 16;

    std::vector<uint32_t> filter(fwords, 0), c2bits(c2words, 0);
    auto bloom_add = [&](uint32_t hi, uint32_t lo, uint32_t group) {
        uint32_t idx[3];
        hsgpu_bloom_idx(hi, lo, group, idx);
        for (uint32_t j = 0; j < 3; j++) c2bits[(j << (HSGPU_BLOOM_PLANE_LOG2 - 5)) + (idx[j] >> 5)] |= 1u << (idx[j] & 31);
    };
    if (tflags & HSGPU_F_BLOOM) {
        /* group 3: the keys of table B; groups 4 / 5: every class-A literal by its five last bytes when they are all given (no
         * wildcard beyond the case bit of a case-blind table), else by each of its 4-byte key variants */
        for (auto &kv : keys[1]) bloom_add(kv.first << 8, 0, 3);
        const uint32_t blind1 = blind ? 0x20u : 0u;
        for (size_t i = 0; i < n; i++) {
            HsgpuDevLit l = dl[i];
            l.msk |= (uint64_t)blind4 << 32;
            l.v &= ~((uint64_t)blind4 << 32);
            if (choose_class(l).cls != 0) continue;
            const uint32_t m4 = (uint32_t)(l.msk >> 32), v4 = (uint32_t)(l.v >> 32);
            const uint32_t m5 = (uint32_t)(dl[i].msk >> 24) & 0xffu, v5 = (uint32_t)(dl[i].v >> 24) & 0xffu;
            if (m4 == 0xffffffffu && (m5 | blind1) == 0xffu) bloom_add(v4, (v5 & ~blind1) << 24, 5);
            else for_each_variant(v4, ~m4, [&](uint32_t key) { bloom_add(key, 0, 4); });
        }
    }
    std::vector<HsgpuHtSlot> ht[2];
    for (int c = 0; c < 2; c++) ht[c].assign((size_t)HSGPU_BUCKET_SLOTS << ht_log2[c], 0u);
    std::vector<uint32_t> c2ref((tflags & HSGPU_F_HAS_C) ? 65536 : 4, 0);

    auto set_bit = [&](uint32_t a, uint32_t bit) {
        if (tflags & HSGPU_F_REPL) {
            for (uint32_t col = 0; col < 32; col++) filter[a * 32 + col] |= 1u << bit;
        } else {
            filter[a >> 2] |= 1u << bit;
        }
    };

    for (int c = 0; c < 2; c++) {
        /* deterministic order: sort keys */
        std::vector<uint32_t> ks;
        ks.reserve(keys[c].size());
        for (auto &kv : keys[c]) ks.push_back(kv.first);
        std::sort(ks.begin(), ks.end());
        for (uint32_t key : ks) {
            uint32_t x1 = (c == 0) ? (key >> 8) : key; /* 3-byte suffix */
            uint32_t prod1 = hsgpu_filter_prod(x1);
            uint32_t a1 = prod1 >> fshift;
            if (tflags & HSGPU_F_GATE) { /* the key gate of the confirm kernel (no 2-byte table needs the section) */
                const uint32_t gbit = hsgpu_key_gate_bit(c == 0 ? key : (key | HSGPU_GATE_B_SALT));
                c2bits[gbit >> 5] |= 1u << (gbit & 31);
            }
            if (pair) {
                /* the pair filter is filled from the literals themselves, below; a 3-byte key of either kind
                 * marks the gate bitmap the confirm step consults before it probes this table */
                if (c == 1) c2bits[hsgpu_gate_bit(key) >> 5] |= 1u << (hsgpu_gate_bit(key) & 31);
            } else if (tflags & HSGPU_F_WIDE) { /* {lo, hi} at entry a1 >> 3 */
                filter[2 * (a1 >> 3)] |= c == 0 ? 1u << hsgpu_filter_bit_lo_wide(key & 0xff) : ~0u; /* (3-byte keys are folded: any b3) */
                filter[2 * (a1 >> 3) + 1] |= 1u << hsgpu_filter_bit_hi(prod1);
            } else if (c == 0) {
                set_bit(a1, hsgpu_filter_bit_a(key & 0xff, a1));
                if (tflags & HSGPU_F_K2) set_bit(a1, hsgpu_filter_bit_a2(key & 0xff, prod1));
            } else if (tflags & HSGPU_F_BFOLD) {
                filter[a1 >> 2] = ~0u;
            } else {
                set_bit(a1, hsgpu_filter_bit_b(a1));
                if (tflags & HSGPU_F_K2) set_bit(a1, hsgpu_filter_bit_b2(prod1));
            }
            uint32_t bmask = ((uint32_t)1 << ht_log2[c]) - 1;
            uint32_t bkt = hsgpu_ht_bucket(key, ht_log2[c]);
            for (;;) {
                HsgpuHtSlot *sl = &ht[c][(size_t)bkt * HSGPU_BUCKET_SLOTS];
                uint32_t j = 0;
                while (j < HSGPU_BUCKET_SLOTS && sl[j]) j++;
                if (j < HSGPU_BUCKET_SLOTS) {
                    sl[j] = add_list(keys[c][key]) | hsgpu_ht_tag(key, ht_log2[c]) << HSGPU_SLOT_TAG_SHIFT;
                    break;
                }
                bkt = (bkt + 1) & bmask;
            }
        }
    }
    {
        std::vector<uint32_t> ks;
        for (auto &kv : keys[2]) ks.push_back(kv.first);
        std::sort(ks.begin(), ks.end());
        for (uint32_t key : ks) {
            c2bits[key >> 5] |= 1u << (key & 31);
            c2ref[key] = add_list(keys[2][key]);
        }
    }
    if (pair) {
        for (size_t i = 0; i < n; i++) {
            const int deltas[2] = {0, odd_delta[i]};
            for (int delta : deltas) {
                const PairKey pk = pair_key(dl[i], delta, pair_hash_mask);
                for_each_variant(pk.val, pk.unk, [&](uint32_t x) {
                    const uint32_t prod = hsgpu_filter_prod(x), e = hsgpu_pair_entry(prod, k);
                    filter[2 * e + 1] |= 1u << hsgpu_pair_bit_h(prod);
                    if (pk.use_a) filter[2 * e + 1] |= pk.abits;
                    else filter[2 * e] |= pk.bbits;
                });
            }
        }
    }
    if (lists.empty()) lists.push_back(HSGPU_LIST_END);

    h.off_lists = (uint32_t)off;
    h.n_lists = (uint32_t)lists.size();
    off += align_up(lists.size() * 4, 16);
    h.off_lits = (uint32_t)off;
    off += n * sizeof(HsgpuDevLit);
    off = align_up(off, 16);
    if (off > 0xfffffff0ull) {
        hsgpu_set_error("compiled table too large");
        return HSGPU_COMPILER_ERROR;
    }
    h.blob_bytes = (uint32_t)off;

    blob.assign(off, 0);
    memcpy(blob.data() + h.off_filter, filter.data(), filter.size() * 4);
    memcpy(blob.data() + h.off_c2bits, c2bits.data(), c2bits.size() * 4);
    memcpy(blob.data() + h.off_ht_a, ht[0].data(), ht[0].size() * sizeof(HsgpuHtSlot));
    memcpy(blob.data() + h.off_ht_b, ht[1].data(), ht[1].size() * sizeof(HsgpuHtSlot));
    memcpy(blob.data() + h.off_c2ref, c2ref.data(), c2ref.size() * 4);
    memcpy(blob.data() + h.off_lists, lists.data(), lists.size() * 4);
    memcpy(blob.data() + h.off_lits, dl.data(), n * sizeof(HsgpuDevLit));
    h.checksum = 0;
    memcpy(blob.data(), &h, sizeof(h));
    h.checksum = hsgpu_blob_checksum(blob.data(), blob.size());
    memcpy(blob.data(), &h, sizeof(h));
    return HSGPU_SUCCESS;
}

/* Was this table compiled from exactly these literals (same order)? A loader that rebuilds its own
 * view of the literals (hs_deserialize_database) asks before trusting a stored table. */
int hsgpu_table_agrees(const hsgpu_hwlm *t, const hsgpu_lit_t *lits, size_t n) {
    if (!t || !lits || t->hdr()->n_lits != n) return 0;
    const HsgpuDevLit *have = t->lits();
    for (size_t i = 0; i < n; i++) {
        HsgpuDevLit want;
        if (normalise(lits[i], i, want) != HSGPU_SUCCESS) return 0;
        if (want.v != have[i].v || want.msk != have[i].msk || want.groups != have[i].groups || want.id != have[i].id ||
            want.size != have[i].size || want.flags != have[i].flags)
            return 0;
    }
    return 1;
}

int hsgpu_validate_blob(const void *buf, size_t len) {
    if (!buf || len < sizeof(HsgpuTableHeader)) return HSGPU_INVALID;
    HsgpuTableHeader h;
    memcpy(&h, buf, sizeof(h));
    if (h.magic != HSGPU_TABLE_MAGIC) return HSGPU_INVALID;
    if (h.version != HSGPU_TABLE_VERSION) return HSGPU_DB_VERSION_ERROR;
    if (h.blob_bytes != len) return HSGPU_INVALID;
    if (h.filter_log2 < 4 || h.filter_log2 > 15 || ((h.flags & HSGPU_F_REPL) && h.filter_log2 > 10) ||
        ((h.flags & HSGPU_F_WIDE) && (h.filter_log2 > 14 || !(h.flags & HSGPU_F_K2) ||
                                      (h.flags & (HSGPU_F_REPL | HSGPU_F_HAS_C | HSGPU_F_STRIDE2 | HSGPU_F_PAIR)) ||
                                      ((h.flags & HSGPU_F_HAS_B) && !(h.flags & HSGPU_F_BFOLD)))) ||
        ((h.flags & HSGPU_F_PAIR) && (h.filter_log2 > 14 || (h.flags & (HSGPU_F_REPL | HSGPU_F_K2 | HSGPU_F_HAS_C | HSGPU_F_BFOLD)) ||
                                      !(h.flags & HSGPU_F_STRIDE2) || !(h.hash_mask & 0xff0000u) || h.hash_mask > 0xffffffu)) ||
        h.ht_a_log2 < 2 || h.ht_a_log2 > 26 || h.ht_b_log2 < 2 || h.ht_b_log2 > 26)
        return HSGPU_INVALID;
    auto in = [&](uint64_t off, uint64_t bytes) { return off >= sizeof(h) && off + bytes <= len; };
    if (((h.flags & HSGPU_F_BLOOM) && (h.flags & (HSGPU_F_GATE | HSGPU_F_STRIDE2 | HSGPU_F_PAIR | HSGPU_F_HAS_C))) ||
        !in(h.off_filter, 4ull * hsgpu_filter_words(h.flags, h.filter_log2)) ||
        !in(h.off_c2bits, (h.flags & HSGPU_F_BLOOM) ? 4ull * HSGPU_BLOOM_WORDS : 8192) ||
        !in(h.off_ht_a, 16ull << h.ht_a_log2) || !in(h.off_ht_b, 16ull << h.ht_b_log2) ||
        !in(h.off_c2ref, (h.flags & HSGPU_F_HAS_C) ? 262144 : 16) || !in(h.off_lists, 4ull * h.n_lists) ||
        !in(h.off_lits, 32ull * h.n_lits))
        return HSGPU_INVALID;
    if (hsgpu_blob_checksum((const uint8_t *)buf, len) != h.checksum) return HSGPU_INVALID;
    return HSGPU_SUCCESS;
}

/* ---- C ABI: build side --------------------------------------------------- */

extern "C" int hsgpu_hwlm_build(const hsgpu_lit_t *lits, size_t n, unsigned flags, hsgpu_hwlm_t **out) {
    if (!out) return HSGPU_INVALID;
    *out = nullptr;
    try {
        std::vector<uint8_t> blob;
        int rv = hsgpu_compile_table(lits, n, flags, blob);
        if (rv != HSGPU_SUCCESS) return rv;
        hsgpu_hwlm *t = new hsgpu_hwlm;
        t->blob.swap(blob);
        *out = t;
        return HSGPU_SUCCESS;
    } catch (const std::bad_alloc &) {
        return HSGPU_NOMEM;
    } catch (...) {
        return HSGPU_UNKNOWN_ERROR;
    }
}

extern "C" void hsgpu_hwlm_free(hsgpu_hwlm_t *t) {
    if (!t) return;
    hsgpu_release_device_copies(t);
    delete t;
}

extern "C" size_t hsgpu_hwlm_size(const hsgpu_hwlm_t *t) { return t ? t->blob.size() : 0; }

extern "C" int hsgpu_hwlm_get_info(const hsgpu_hwlm_t *t, hsgpu_hwlm_info_t *info) {
    if (!t || !info) return HSGPU_INVALID;
    const HsgpuTableHeader *h = t->hdr();
    info->n_lits = h->n_lits;
    info->n_class_a = h->n_a;
    info->n_class_b = h->n_b;
    info->n_class_c = h->n_c;
    info->filter_words = hsgpu_filter_words(h->flags, h->filter_log2);
    info->flags = h->flags;
    info->filter_entries = h->filter_entries;
    info->ht_a_slots = HSGPU_BUCKET_SLOTS << h->ht_a_log2;
    info->ht_b_slots = HSGPU_BUCKET_SLOTS << h->ht_b_log2;
    info->max_size = h->max_size;
    info->blob_bytes = h->blob_bytes;
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_hwlm_serialize(const hsgpu_hwlm_t *t, void *buf, size_t cap, size_t *len) {
    if (!t || !len) return HSGPU_INVALID;
    *len = t->blob.size();
    if (!buf) return HSGPU_SUCCESS;
    if (cap < t->blob.size()) return HSGPU_INSUFFICIENT_SPACE;
    memcpy(buf, t->blob.data(), t->blob.size());
    return HSGPU_SUCCESS;
}

extern "C" int hsgpu_hwlm_deserialize(const void *buf, size_t len, hsgpu_hwlm_t **out) {
    if (!out) return HSGPU_INVALID;
    *out = nullptr;
    int rv = hsgpu_validate_blob(buf, len);
    if (rv != HSGPU_SUCCESS) return rv;
    try {
        hsgpu_hwlm *t = new hsgpu_hwlm;
        t->blob.assign((const uint8_t *)buf, (const uint8_t *)buf + len);
        *out = t;
        return HSGPU_SUCCESS;
    } catch (const std::bad_alloc &) {
        return HSGPU_NOMEM;
    }
}
