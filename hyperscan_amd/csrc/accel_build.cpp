/*
 * accel_build.cpp -- compile side of the accelerators (host only):
 *   hsgpu_accel_forward     the pre-skip scheme Rose attaches to a literal matcher:
 *                           findForwardAccelScheme / buildForwardAccel,
 *                           src/rose/rose_build_lit_accel.cpp:372-465
 *   hsgpu_class_to_shufti   shuftiBuildMasks, src/nfa/shufticompile.cpp:54-109
 * The decisions are the reference's (which scheme, which byte / pair / class, which offset);
 * the code is not: candidates are scored by a rank tuple instead of a comparison cascade,
 * and classes are plain 256-bit sets.
 *
 * On the GPU the chosen scheme is not needed to make the literal scan fast (the filter reads
 * every byte at constant cost); it exists for callers that keep the reference's structure --
 * hwlmExec's do_accel_block (src/hwlm/hwlm.c:48-99) asks "where could the first literal
 * start?" per block, which hsgpu_class_scan_dev / hsgpu_pair_scan_dev answer for a whole
 * batch at once (hyperscan_amd/accel.py: forward_skip).
 */
#include "../../include/hsgpu.h"
#include "internal.h"

#include <algorithm>
#include <bitset>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace {

constexpr unsigned kMaxAccelOffset = 16; /* MAX_ACCEL_OFFSET */
constexpr unsigned kMaxShuftiWidth = 240; /* MAX_SHUFTI_WIDTH */
constexpr uint8_t kCaseClear = 0xdf;

typedef std::bitset<256> ByteSet;

bool is_alpha(uint8_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
uint8_t lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

struct Lit { /* AccelString, src/rose/rose_build_lit_accel.h:43-53 */
    std::string s; /* upper-cased when nocase, as hwlmLiteral stores it */
    bool nocase;
    std::vector<uint8_t> msk, cmp;
    uint64_t groups;
    /* how far a mask reaches in front of the literal's own bytes (mask_overhang) */
    size_t overhang() const {
        size_t live = msk.size();
        for (uint8_t m : msk) {
            if (m) break;
            live--;
        }
        return live > s.size() ? live - s.size() : 0;
    }
    size_t span() const { return std::min<size_t>(kMaxAccelOffset, s.size()); }
};

/* Does `probe` (one byte, or two with WIDTH == 2) occur in lit within the first 16 bytes,
 * case-exactly / ignoring bit 5? first position or -1. */
template <int WIDTH> int find_in(const Lit &l, const uint8_t *probe, bool blind) {
    const uint8_t m = blind ? kCaseClear : 0xff;
    const size_t n = l.span();
    for (size_t j = 0; j + WIDTH <= n; j++) {
        bool eq = true;
        for (int k = 0; k < WIDTH; k++) eq &= ((uint8_t)l.s[j + k] & m) == (probe[k] & m);
        if (eq) return (int)j;
    }
    return -1;
}

/* One candidate byte (WIDTH 1) or byte pair (WIDTH 2) taken from the first literal must occur
 * in EVERY literal; it turns case-blind as soon as a caseless literal has an alphabetic byte
 * there, or some literal only has it in the other case. */
template <int WIDTH> struct Candidate {
    uint8_t c[2] = {0, 0};
    bool blind = false, valid = false;
    unsigned max_offset = 0;
    /* the reference's preference order (candidate::operator>): distinct bytes (pairs only),
     * case-exact, smaller offset; later candidates win ties */
    std::tuple<int, int, int> rank() const {
        const uint8_t m = blind ? kCaseClear : 0xff;
        const int differs = WIDTH == 2 ? ((c[0] & m) != (c[1] & m)) : 0;
        return std::make_tuple(differs, blind ? 0 : 1, -(int)max_offset);
    }
};

template <int WIDTH> Candidate<WIDTH> best_candidate(const std::vector<const Lit *> &lits) {
    Candidate<WIDTH> best;
    const Lit &first = *lits.front();
    for (size_t i = 0; i + WIDTH <= first.span(); i++) {
        Candidate<WIDTH> cur;
        cur.valid = true;
        for (int k = 0; k < WIDTH; k++) cur.c[k] = (uint8_t)first.s[i + k];
        bool everywhere = true;
        for (const Lit *l : lits) {
            bool alpha = false;
            for (int k = 0; k < WIDTH; k++) alpha |= is_alpha(cur.c[k]);
            if (l->nocase && alpha) cur.blind = true;
            if (find_in<WIDTH>(*l, cur.c, cur.blind) >= 0) continue;
            if (!cur.blind && find_in<WIDTH>(*l, cur.c, true) >= 0) {
                cur.blind = true;
                continue;
            }
            everywhere = false;
            break;
        }
        if (!everywhere) continue;
        for (const Lit *l : lits) {
            /* pairs: the first occurrence; single bytes: the last one (the reference's loops
             * differ in exactly that `break`, rose_build_lit_accel.cpp:177-179 vs :291-294) */
            const uint8_t m = cur.blind ? kCaseClear : 0xff;
            for (size_t j = 0; j + WIDTH <= l->span(); j++) {
                bool eq = true;
                for (int k = 0; k < WIDTH; k++) eq &= ((uint8_t)l->s[j + k] & m) == (cur.c[k] & m);
                if (!eq) continue;
                cur.max_offset = std::max<unsigned>(cur.max_offset, (unsigned)(j + l->overhang()));
                if (WIDTH == 2) break;
            }
        }
        if (!best.valid || cur.rank() >= best.rank()) best = cur;
    }
    return best;
}

int class_to_shufti(const ByteSet &cls, uint8_t lo[16], uint8_t hi[16]) {
    /* high nibbles that accept the same set of low nibbles share a bucket */
    std::map<uint16_t, uint16_t> buckets; /* low-nibble set -> high-nibble set */
    for (unsigned h = 0; h < 16; h++) {
        uint16_t lows = 0;
        for (unsigned l = 0; l < 16; l++)
            if (cls[h << 4 | l]) lows |= (uint16_t)(1u << l);
        if (lows) buckets[lows] |= (uint16_t)(1u << h);
    }
    if (buckets.size() > 8) return -1;
    memset(lo, 0, 16);
    memset(hi, 0, 16);
    unsigned bit = 0;
    for (const auto &b : buckets) {
        for (unsigned n = 0; n < 16; n++) {
            if (b.first >> n & 1) lo[n] |= (uint8_t)(1u << bit);
            if (b.second >> n & 1) hi[n] |= (uint8_t)(1u << bit);
        }
        bit++;
    }
    return (int)bit;
}

} // namespace

extern "C" int hsgpu_class_to_shufti(const hsgpu_class_t *cls, uint8_t lo[16], uint8_t hi[16]) {
    if (!cls || !lo || !hi) return HSGPU_INVALID;
    ByteSet s;
    for (unsigned c = 0; c < 256; c++)
        if (cls->bitmap[c >> 3] >> (c & 7) & 1) s.set(c);
    if (s.none() || s.all()) return -1; /* the reference asserts on both */
    return class_to_shufti(s, lo, hi);
}

extern "C" int hsgpu_accel_forward(const hsgpu_lit_t *lits, size_t n, uint64_t expected_groups, hsgpu_accel_t *out) {
    if (!out || (n && !lits)) return HSGPU_INVALID;
    memset(out, 0, sizeof(*out));
    out->type = HSGPU_ACCEL_NONE;
    std::vector<Lit> all;
    try {
        for (size_t i = 0; i < n; i++) {
            if (!lits[i].s || lits[i].len == 0 || (lits[i].msk_len && (!lits[i].msk || !lits[i].cmp)) || lits[i].msk_len > 8) {
                hsgpu_set_error("literal %zu is malformed", i);
                return HSGPU_COMPILER_ERROR;
            }
            Lit l;
            l.s.assign((const char *)lits[i].s, lits[i].len);
            l.nocase = lits[i].nocase != 0;
            if (l.nocase)
                for (char &c : l.s) c = (char)upper((uint8_t)c);
            l.msk.assign(lits[i].msk, lits[i].msk + lits[i].msk_len);
            l.cmp.assign(lits[i].cmp, lits[i].cmp + lits[i].msk_len);
            l.groups = lits[i].groups;
            all.push_back(l);
        }
        std::vector<const Lit *> sel;
        for (const Lit &l : all)
            if (l.groups & expected_groups) sel.push_back(&l);
        if (sel.empty()) return HSGPU_SUCCESS;

        const Candidate<2> two = best_candidate<2>(sel);
        if (two.valid) {
            out->type = two.blind ? HSGPU_ACCEL_DVERM_NOCASE : HSGPU_ACCEL_DVERM;
            out->offset = (uint8_t)two.max_offset;
            out->c1 = two.blind ? (two.c[0] & kCaseClear) : two.c[0];
            out->c2 = two.blind ? (two.c[1] & kCaseClear) : two.c[1];
            return HSGPU_SUCCESS;
        }
        const Candidate<1> one = best_candidate<1>(sel);
        if (one.valid) {
            out->type = one.blind ? HSGPU_ACCEL_VERM_NOCASE : HSGPU_ACCEL_VERM;
            out->offset = (uint8_t)one.max_offset;
            out->c1 = one.blind ? (one.c[0] & kCaseClear) : one.c[0];
            return HSGPU_SUCCESS;
        }

        /* a byte class per offset 0..15: which byte could stand there if some literal starts
         * at offset 0; a literal adds nothing at an offset once one of its earlier bytes is
         * already in that offset's class (litGuardedByCharReach) */
        std::vector<ByteSet> reach(kMaxAccelOffset);
        for (const Lit *lp : sel) {
            const Lit &l = *lp;
            const size_t oh = l.overhang();
            for (size_t i = 0; i < oh && i < kMaxAccelOffset; i++)
                for (unsigned v = 0; v < 256; v++)
                    if ((v & l.msk[i]) == l.cmp[i]) reach[i].set(v);
            for (size_t i = oh; i < kMaxAccelOffset; i++) {
                const size_t e = i - oh;
                bool guarded = false;
                for (size_t k = 0; k <= e && k < l.s.size() && !guarded; k++) {
                    const uint8_t c = (uint8_t)l.s[k];
                    guarded = l.nocase ? (reach[i][upper(c)] && reach[i][lower(c)]) : (bool)reach[i][c];
                }
                if (guarded) continue;
                const uint8_t c = (uint8_t)(e < l.s.size() ? l.s[e] : l.s.back());
                if (l.nocase) {
                    reach[i].set(upper(c));
                    reach[i].set(lower(c));
                } else {
                    reach[i].set(c);
                }
            }
        }
        size_t best_i = 0;
        for (size_t i = 1; i < kMaxAccelOffset; i++)
            if (reach[i].count() < reach[best_i].count()) best_i = i;
        if (reach[best_i].count() > kMaxShuftiWidth) return HSGPU_SUCCESS; /* too wide: no acceleration */
        out->offset = (uint8_t)best_i;
        if (class_to_shufti(reach[best_i], out->mask_lo, out->mask_hi) != -1) {
            out->type = HSGPU_ACCEL_SHUFTI;
            return HSGPU_SUCCESS;
        }
        hsgpu_class_t cls;
        memset(&cls, 0, sizeof(cls));
        for (unsigned v = 0; v < 256; v++)
            if (reach[best_i][v]) cls.bitmap[v >> 3] |= (uint8_t)(1u << (v & 7));
        hsgpu_class_to_truffle(&cls, out->mask_lo, out->mask_hi);
        out->type = HSGPU_ACCEL_TRUFFLE;
        return HSGPU_SUCCESS;
    } catch (const std::bad_alloc &) {
        return HSGPU_NOMEM;
    }
}
