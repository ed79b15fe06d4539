/*
 * hs_pattern.h -- the pattern side of the hs_* facade: what a compiled branch looks like and how
 * its fragments run on the host (test and product infrastructure of hs_facade.cpp; not part of
 * the C ABI).
 *
 * A branch is R1 LIT R2 (hs_pattern.cpp compiles it). LIT goes to the GPU literal matcher; on a
 * hit the host runs R2 forwards from the literal's end and R1 backwards from its start:
 *   TailNfa::run64        linear tails of <= 63 units, one shift-and word
 *   TailNfa::run_general  Auto forwards  (Rose's suffix engines, nfaQueueExec)
 *   TailNfa::run_reverse  Auto backwards (Rose's prefix engines)
 */
#ifndef HS_PATTERN_H
#define HS_PATTERN_H

#include <algorithm>
#include <bitset>
#include <string>
#include <vector>

namespace hsf {

typedef std::bitset<256> ByteSet;
constexpr unsigned kInf = ~0u;

struct Unit { /* one character class; `star` = may repeat, `optional` = may be skipped */
    ByteSet cls;
    bool optional = false, star = false;
};

constexpr unsigned long long kInf64 = ~0ull;
constexpr size_t kMaxPositions = 4096;
typedef std::vector<unsigned long long> Bits;

/* A regex fragment as a Glushkov position automaton (one position per character-class
 * occurrence; the construction of src/nfagraph's position NFA in miniature) in the shape of the
 * reference's LimEx engines (src/nfa/limex_*): most positions are followed by the next one, so
 *     next = ((cur & shift_ok) << 1) | OR of exc_row[p] for p in cur & exc
 * -- one shift for the chains, a row per "exceptional" position for everything else. */
struct Auto {
    size_t npos = 0, W = 0;       /* positions, 64-bit words per position set */
    Bits first, last;             /* [W] */
    Bits reach;                   /* [256][W]: positions accepting byte c */
    Bits shift_ok, exc;           /* [W] */
    std::vector<Bits> exc_row;    /* [npos]: follow[p] minus {p+1}; empty unless p is in exc */
    bool nullable = true;
    unsigned long long wmin = 0, wmax = 0; /* width bounds; kInf64 = unbounded */
    /* \b / \B inside the fragment: a second and third layer of the same sets, valid only across
     * a boundary where the assertion holds ([0] = \b, [1] = \B). Entering a position, following
     * an edge, ending and matching the empty string can each be conditional. */
    bool has_cond = false;
    Bits cfirst[2], clast[2], cshift_ok[2], cexc[2];
    std::vector<Bits> cexc_row[2];
    bool cnullable[2] = {false, false};
    size_t bytes() const {
        size_t n = (first.size() + last.size() + reach.size() + shift_ok.size() + exc.size()) * 8;
        for (const Bits &r : exc_row) n += r.size() * 8;
        for (int k = 0; k < 2; k++) {
            n += (cfirst[k].size() + clast[k].size() + cshift_ok[k].size() + cexc[k].size()) * 8;
            for (const Bits &r : cexc_row[k]) n += r.size() * 8;
        }
        return n;
    }
};

struct Pattern {
    std::string lit;          /* literal prefix, as written (upper-cased compare if nocase) */
    bool nocase = false, single = false, som = false, quiet = false;
    /* `^` in front / `$` at the back of the branch (multiline: the HS_FLAG_MULTILINE reading) */
    bool bol = false, eol = false;
    bool bol_ml = false; /* `^` under HS_FLAG_MULTILINE: also after any newline (never for \A) */
    bool eol_ml = false; /* `$` under HS_FLAG_MULTILINE: also before any newline */
    bool eol_nl = false; /* `$` and \Z: also before the data's final newline (never for \z) */
    /* \b (1) / \B (2) at the four places they are supported: the start of the match, just before
     * and just after the literal, the end of the match */
    unsigned char as_start = 0, as_lit_pre = 0, as_lit_post = 0, as_end = 0;
    unsigned id = 0;
    unsigned expr = 0;        /* index of the expression this branch belongs to */
    std::vector<Unit> tail;   /* empty: pure literal */
    bool tail_nullable = true;
    /* hs_expr_ext_t (src/hs_compile.h:244-310): bounds on `to` and on the match length */
    unsigned long long ext_flags = 0, min_offset = 0, max_offset = 0, min_length = 0;
    /* shift-and form of the tail for <= 63 units (built once by finish_pattern): bit i of
     * reach[c] = unit i accepts byte c; star / optional unit masks */
    bool fast = false;
    std::vector<unsigned long long> reach; /* [256] */
    /* set by the database once its patterns are final: the table of the FIRST pattern with the same one (patterns that share a
     * tail share its 2 KiB table -- config 5's 1 000 patterns have three: the confirm's tables stay in L1 instead of 2 MB of L3) */
    const unsigned long long *reach_shared = nullptr;
    const unsigned long long *reach_tab() const { return reach_shared ? reach_shared : reach.data(); }
    unsigned long long star_mask = 0, opt_mask = 0;
    /* fragments with groups / alternation / long repeats, and every R1 in front of a literal:
     * position automata (see Auto). `general`: R2 forwards from the literal's end; `has_pre`: R1
     * REVERSED, run backwards from the literal's first byte */
    bool general = false, has_pre = false;
    Auto g, pre;
};

struct ParseError {
    std::string msg;
};
struct NeverMatch : ParseError { /* well-formed, but its language is empty: the branch is dropped */
    NeverMatch() : ParseError{"Pattern can never match."} {}
};
struct NoLiteral : ParseError { /* the branch is well-formed but offers no top-level literal */
    NoLiteral() : ParseError{"Pattern has no mandatory literal at its top level (every branch needs one)."} {}
};

/* A literal-less class sequence A{m,}B{n,} ("+" is {1,}): no literal for the GPU literal matcher to find, so it is
 * evaluated on the GPU from the class bitmaps (csrc/class_seq.hip) -- the job the reference gives to an accelerated
 * NFA / DFA engine (src/nfa/limex_accel.c:49-74). */
struct ClassSeq {
    ByteSet a, b;
    unsigned m = 1, n = 1;
    unsigned id = 0, expr = 0;
    bool single = false, quiet = false;
};

/* hs_pattern.cpp */
/* true and `out` filled when the expression is exactly two classes with open repeats (classes: [...], \d \w \s
 * and their complements, `.`; repeats: + or {k,} with k <= 16) under flags the form supports; false otherwise
 * (the general compiler then has its say) */
bool parse_class_seq(const std::string &expr, unsigned flags, ClassSeq &out);
bool is_word_char(unsigned char c);
bool is_alpha(unsigned char c);
/* the reference's own flag rules, in its order (src/compiler/compiler.cpp:286-294,166-196) */
void check_flags(unsigned flags, bool literal_api);
/* expression -> its branches (all reporting `id`); throws ParseError */
std::vector<Pattern> parse_pattern(const std::string &expr, unsigned flags, unsigned id);
/* derived tables of a branch (after ext parameters are applied) */
void finish_pattern(Pattern &p);
/* width of one branch as written (no ext parameters): [lo, hi], hi meaningless when inf */
void raw_widths(const Pattern &p, unsigned long long &lo, unsigned long long &hi, bool &inf);

/* \b / \B between buf[pos - 1] and buf[pos]; outside the block counts as a non-word byte */
inline bool is_word_byte(unsigned char c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
}
inline bool assert_ok(unsigned char kind, const unsigned char *buf, size_t len, size_t pos) {
    if (!kind) return true;
    const bool before = pos > 0 && is_word_byte(buf[pos - 1]), after = pos < len && is_word_byte(buf[pos]);
    return (before != after) == (kind == 1);
}

/* simulation of the linear NFA: state i = "units 0..i-1 consumed" */
struct TailNfa {
    /* the linear tail, bit-parallel (shift-and) in one 64-bit word: state i = "units 0..i-1
     * consumed"; per input byte one table read, a shift, and the optional-unit closure (skipping
     * unit i = bit i -> i+1). Calls report(to) for every offset after which the tail has matched. */
    static unsigned long long closure64(unsigned long long s, unsigned long long opt) {
        for (unsigned long long add; (add = ((s & opt) << 1) & ~s) != 0;) s |= add;
        return s;
    }
    template <class F> static void run64(const Pattern &p, const unsigned char *buf, size_t len, size_t pos, F report) {
        const unsigned long long accept = 1ull << p.tail.size();
        unsigned long long cur = closure64(1ull, p.opt_mask);
        const unsigned long long *reach = p.reach_tab();
        if (cur & accept) { if (!report(pos)) return; }
        while (pos < len && (cur & (accept - 1))) {
            const unsigned long long live = cur & reach[buf[pos++]];
            cur = closure64((live << 1) | (live & p.star_mask), p.opt_mask);
            if (cur & accept) { if (!report(pos)) return; }
        }
    }

    /* one step of a position automaton (Auto): cur = positions that consumed the last byte */
    static constexpr size_t kMaxW = kMaxPositions / 64;
    static inline unsigned long long step1(const Auto &a, unsigned long long cur) {
        unsigned long long next = (cur & a.shift_ok[0]) << 1;
        for (unsigned long long e = cur & a.exc[0]; e; e &= e - 1) next |= a.exc_row[__builtin_ctzll(e)][0];
        return next;
    }
    static inline void step(const Auto &a, const unsigned long long *cur, unsigned long long *next) {
        const size_t W = a.W;
        unsigned long long carry = 0;
        for (size_t w = 0; w < W; w++) {
            const unsigned long long v = cur[w] & a.shift_ok[w];
            next[w] = (v << 1) | carry;
            carry = v >> 63;
        }
        for (size_t w = 0; w < W; w++)
            for (unsigned long long e = cur[w] & a.exc[w]; e; e &= e - 1) {
                const Bits &row = a.exc_row[w * 64 + __builtin_ctzll(e)];
                for (size_t k = 0; k < W; k++) next[k] |= row[k];
            }
    }
    /* cur = next & reach[c]; returns whether anything is live, *acc whether `last` was met */
    static inline bool advance(const Auto &a, unsigned char c, const unsigned long long *next, unsigned long long *cur, bool *acc) {
        const unsigned long long *r = &a.reach[(size_t)c * a.W];
        unsigned long long any = 0, hit = 0;
        for (size_t w = 0; w < a.W; w++) {
            cur[w] = next[w] & r[w];
            any |= cur[w];
            hit |= cur[w] & a.last[w];
        }
        *acc = hit != 0;
        return any != 0;
    }

    /* ---- fragments with \b / \B inside: three layers of every set. Whether a layer applies is
     * known once the bytes on both sides of a boundary are: N[0] always, N[1] across a word
     * boundary, N[2] across a non-boundary. `dir` = +1 forwards from pos, -1 backwards. ---- */
    static inline void step_layer(size_t W, const Bits &shift_ok, const Bits &exc, const std::vector<Bits> &exc_row,
                                  const unsigned long long *cur, unsigned long long *next) {
        unsigned long long carry = 0;
        for (size_t w = 0; w < W; w++) {
            const unsigned long long v = cur[w] & shift_ok[w];
            next[w] = (v << 1) | carry;
            carry = v >> 63;
        }
        for (size_t w = 0; w < W; w++)
            for (unsigned long long e = cur[w] & exc[w]; e; e &= e - 1) {
                const Bits &row = exc_row[w * 64 + __builtin_ctzll(e)];
                for (size_t k = 0; k < W; k++) next[k] |= row[k];
            }
    }
    /* on_accept(pos) for every boundary at which the fragment can end, nullable included; it returns
     * false to stop. Forwards: pos runs up from `pos`; backwards: down. */
    template <class F> static void run_cond(const Auto &a, const unsigned char *buf, size_t len, size_t pos, int dir, F on_accept) {
        const size_t W = a.W;
        auto boundary = [&](size_t at) {
            return (at > 0 && is_word_byte(buf[at - 1])) != (at < len && is_word_byte(buf[at]));
        };
        bool bd = boundary(pos);
        if (a.nullable || a.cnullable[bd ? 0 : 1]) { if (!on_accept(pos)) return; }
        if (!a.npos) return;
        unsigned long long cur[kMaxW], N0[kMaxW], N1[kMaxW], N2[kMaxW];
        std::copy(a.first.begin(), a.first.end(), N0);
        std::copy(a.cfirst[0].begin(), a.cfirst[0].end(), N1);
        std::copy(a.cfirst[1].begin(), a.cfirst[1].end(), N2);
        while (dir > 0 ? pos < len : pos > 0) {
            const unsigned char c = dir > 0 ? buf[pos] : buf[pos - 1];
            const unsigned long long *r = &a.reach[(size_t)c * W], *cond = bd ? N1 : N2;
            unsigned long long any = 0;
            for (size_t w = 0; w < W; w++) any |= cur[w] = (N0[w] | cond[w]) & r[w];
            if (!any) return;
            pos = dir > 0 ? pos + 1 : pos - 1;
            bd = boundary(pos);
            const Bits &cl = a.clast[bd ? 0 : 1];
            unsigned long long hit = 0;
            for (size_t w = 0; w < W; w++) hit |= cur[w] & (a.last[w] | cl[w]);
            if (hit) { if (!on_accept(pos)) return; }
            step_layer(W, a.shift_ok, a.exc, a.exc_row, cur, N0);
            step_layer(W, a.cshift_ok[0], a.cexc[0], a.cexc_row[0], cur, N1);
            step_layer(W, a.cshift_ok[1], a.cexc[1], a.cexc_row[1], cur, N2);
        }
    }

    /* R2 forwards from `pos`: the active set after byte c is (first | follow[active]) & reach[c];
     * a match ends wherever the set meets `last` */
    template <class F> static void run_general(const Auto &a, const unsigned char *buf, size_t len, size_t pos, F report) {
        if (a.has_cond) {
            run_cond(a, buf, len, pos, +1, report);
            return;
        }
        if (a.nullable) { if (!report(pos)) return; }
        if (a.W == 1) {
            unsigned long long next = a.first[0];
            while (pos < len && next) {
                const unsigned long long cur = next & a.reach[buf[pos++]];
                if (cur & a.last[0]) { if (!report(pos)) return; }
                next = step1(a, cur);
            }
            return;
        }
        unsigned long long cur[kMaxW], next[kMaxW];
        std::copy(a.first.begin(), a.first.end(), next);
        bool acc;
        while (pos < len && advance(a, buf[pos++], next, cur, &acc)) {
            if (acc) { if (!report(pos)) return; }
            step(a, cur, next);
        }
    }
    /* The same from SEVERAL start offsets at once (ascending, duplicates allowed): the automaton is
     * run once over the block and `first` is injected at every start, so k hits of one literal
     * cost one pass instead of k. Reports the union of the separate runs' offsets (a position may be
     * reported twice). Not for automata with \b / \B layers (has_cond). */
    template <class F>
    static void run_general_multi(const Auto &a, const unsigned char *buf, size_t len, const size_t *starts, size_t n, F report) {
        if (!n) return;
        const size_t W = a.W;
        unsigned long long cur[kMaxW], next[kMaxW];
        std::fill(next, next + W, 0ull);
        size_t pos = starts[0], k = 0;
        for (;;) {
            for (; k < n && starts[k] == pos; k++) {
                for (size_t w = 0; w < W; w++) next[w] |= a.first[w];
                if (a.nullable) { if (!report(pos)) return; }
            }
            if (pos >= len) return;
            unsigned long long any = 0;
            for (size_t w = 0; w < W; w++) any |= next[w];
            if (!any) { /* dead until the next start */
                if (k >= n) return;
                pos = starts[k];
                continue;
            }
            bool acc;
            const bool live = advance(a, buf[pos++], next, cur, &acc);
            if (acc) { if (!report(pos)) return; }
            if (live) step(a, cur, next);
            else std::fill(next, next + W, 0ull);
        }
    }
    template <class F>
    static void run64_multi(const Pattern &p, const unsigned char *buf, size_t len, const size_t *starts, size_t n, F report) {
        if (!n) return;
        const unsigned long long accept = 1ull << p.tail.size(), init = closure64(1ull, p.opt_mask);
        unsigned long long cur = 0;
        const unsigned long long *reach = p.reach_tab();
        size_t pos = starts[0], k = 0;
        for (;;) {
            bool injected = false;
            for (; k < n && starts[k] == pos; k++) injected = true;
            if (injected) {
                cur |= init;
                if (init & accept) { if (!report(pos)) return; }
            }
            if (pos >= len) return;
            if (!(cur & (accept - 1))) {
                if (k >= n) return;
                cur = 0;
                pos = starts[k];
                continue;
            }
            const unsigned long long live = cur & reach[buf[pos++]];
            cur = closure64((live << 1) | (live & p.star_mask), p.opt_mask);
            if (cur & accept) { if (!report(pos)) return; }
        }
    }

    /* R1 backwards from the literal's first byte (the automaton is the reversed one): is there a
     * `from` with buf[from, start) in R1 (and, for `^`, a line start at `from`)? leftmost = keep
     * going for the smallest one */
    static bool run_reverse(const Pattern &p, const unsigned char *buf, size_t len, size_t start, bool leftmost,
                            size_t &from) {
        const Auto &a = p.pre;
        auto at_bol = [&](size_t pos) {
            return (!p.bol || pos == 0 || (p.bol_ml && buf[pos - 1] == '\n')) && assert_ok(p.as_start, buf, len, pos);
        };
        bool found = false;
        if (a.has_cond) {
            run_cond(a, buf, len, start, -1, [&](size_t pos) {
                if (!at_bol(pos)) return true;
                found = true;
                from = pos;
                return leftmost;
            });
            return found;
        }
        if (a.nullable && at_bol(start)) {
            found = true;
            from = start;
            if (!leftmost) return true;
        }
        size_t pos = start;
        if (a.W == 1) {
            unsigned long long next = a.first[0];
            while (pos > 0 && next) {
                const unsigned long long cur = next & a.reach[buf[--pos]];
                if ((cur & a.last[0]) && at_bol(pos)) {
                    found = true;
                    from = pos;
                    if (!leftmost) return true;
                }
                next = step1(a, cur);
            }
            return found;
        }
        unsigned long long cur[kMaxW], next[kMaxW];
        std::copy(a.first.begin(), a.first.end(), next);
        bool acc;
        while (pos > 0 && advance(a, buf[--pos], next, cur, &acc)) {
            if (acc && at_bol(pos)) {
                found = true;
                from = pos;
                if (!leftmost) return true;
            }
            step(a, cur, next);
        }
        return found;
    }
};


} // namespace hsf

#endif
