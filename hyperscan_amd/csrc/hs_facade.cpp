/*
 * hs_facade.cpp -- the public hs_* block-mode API (include/hs_gpu.h) on top of the
 * GPU literal engine, with a host-side "Rose-lite" confirm.
 *
 * What each piece stands in for in the reference:
 *   parse_pattern        the subset of src/parser/ needed for "literal prefix + simple
 *                        tail" patterns (the full Ragel parser is out of scope)
 *   build_database       rose_build_matchers.cpp:701-744: one HWLM literal per pattern =
 *                        the last <= 8 bytes of its literal prefix, id = pattern index
 *   on_literal           roseCallback -> roseRunProgram (src/rose/match.c:479-523,
 *                        program_runtime.c): CHECK_MED/LONG_LIT = full-literal compare,
 *                        then the tail engine; REPORT -> the user's match_event_handler;
 *                        SINGLEMATCH = the exhaustion vector (src/report.h)
 *   TailNfa::run64       the NFA engines Rose triggers after a literal (nfaQueueExec):
 *                        linear tails of <= 63 units as one shift-and word
 *   TailBuilder / Auto   every other fragment (groups, alternation, long repeats, and R1 in
 *   run_general/_reverse front of the literal): a Glushkov position automaton of <= 4096
 *                        positions in LimEx shape (shift + exception rows)
 * Matches are reported as the reference does: `to` = offset after the last byte,
 * `from` = 0 unless HS_FLAG_SOM_LEFTMOST, every distinct (id, to) once, in
 * non-decreasing `to`.
 */
#include "../../include/hs_gpu.h"
#include "internal.h"

#include <algorithm>
#include <bitset>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

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
    std::vector<Unit> tail;   /* empty: pure literal */
    bool tail_nullable = true;
    /* hs_expr_ext_t (src/hs_compile.h:244-310): bounds on `to` and on the match length */
    unsigned long long ext_flags = 0, min_offset = 0, max_offset = 0, min_length = 0;
    /* shift-and form of the tail for <= 63 units (built once by finish_pattern): bit i of
     * reach[c] = unit i accepts byte c; star / optional unit masks */
    bool fast = false;
    std::vector<unsigned long long> reach; /* [256] */
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

bool is_word_char(unsigned char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
bool is_alpha(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

void add_range(ByteSet &s, unsigned lo, unsigned hi) {
    for (unsigned c = lo; c <= hi && c < 256; c++) s.set(c);
}

ByteSet class_escape(char e, bool &ok) {
    ByteSet s;
    ok = true;
    switch (e) {
    case 'd': add_range(s, '0', '9'); break;
    case 'w': add_range(s, '0', '9'); add_range(s, 'a', 'z'); add_range(s, 'A', 'Z'); s.set('_'); break;
    case 's': s.set(' '); s.set('\t'); s.set('\n'); s.set('\r'); s.set('\f'); s.set('\v'); break;
    case 'D': s = ~class_escape('d', ok); break;
    case 'W': s = ~class_escape('w', ok); break;
    case 'S': s = ~class_escape('s', ok); break;
    /* src/parser/ComponentClass.cpp:87-88,114-115: horizontal / vertical white space */
    case 'h': s.set(0x09); s.set(0x20); s.set(0xa0); break;
    case 'H': s = ~class_escape('h', ok); break;
    case 'v': s.set(0x0a); s.set(0x0b); s.set(0x0c); s.set(0x0d); s.set(0x85); break;
    case 'V': s = ~class_escape('v', ok); break;
    default: ok = false;
    }
    return s;
}

/* a single escaped literal character: \n \t \r \f \v \a \e \0 \xHH \\ \. etc. */
bool char_escape(const std::string &p, size_t &i, unsigned char &out) {
    char e = p[i];
    switch (e) {
    case 'n': out = '\n'; i++; return true;
    case 't': out = '\t'; i++; return true;
    case 'r': out = '\r'; i++; return true;
    case 'f': out = '\f'; i++; return true;
    case 'a': out = 7; i++; return true;
    case 'e': out = 27; i++; return true;
    case '0': { /* \0 and up to two more octal digits (Parser.rl:503) */
        unsigned v = 0;
        size_t k = i + 1;
        for (int d = 0; d < 2 && k < p.size() && p[k] >= '0' && p[k] <= '7'; d++, k++) v = v * 8 + (p[k] - '0');
        out = (unsigned char)v;
        i = k;
        return true;
    }
    case 'x': {
        unsigned v = 0;
        for (int k = 1; k <= 2; k++) {
            if (i + k >= p.size()) return false;
            char h = p[i + k];
            unsigned d = (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10
                         : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : 99;
            if (d == 99) return false;
            v = v * 16 + d;
        }
        out = (unsigned char)v;
        i += 3;
        return true;
    }
    default:
        if ((e >= 'a' && e <= 'z') || (e >= 'A' && e <= 'Z') || (e >= '1' && e <= '9')) return false;
        out = (unsigned char)e;
        i++;
        return true;
    }
}

/* an escaped single character inside a bracket class: everything char_escape knows, plus \b
 * (backspace) and \ddd octal, which mean something else outside a class */
bool class_char_escape(const std::string &p, size_t &i, unsigned char &out) {
    if (i < p.size() && p[i] == 'b') { out = 8; i++; return true; }
    if (i < p.size() && p[i] >= '1' && p[i] <= '7') {
        unsigned v = 0;
        for (int d = 0; d < 3 && i < p.size() && p[i] >= '0' && p[i] <= '7'; d++, i++) v = v * 8 + (p[i] - '0');
        out = (unsigned char)v;
        return true;
    }
    return char_escape(p, i, out);
}

ByteSet fold_case(const ByteSet &s) {
    ByteSet o = s;
    for (unsigned c = 'a'; c <= 'z'; c++)
        if (s[c] || s[c - 32]) {
            o.set(c);
            o.set(c - 32);
        }
    return o;
}

/* [:name:] inside a bracket class (PCRE's set, ASCII semantics) */
bool posix_class(const std::string &name, ByteSet &s) {
    auto fill = [&](int (*fn)(int)) {
        for (unsigned c = 0; c < 128; c++)
            if (fn((int)c)) s.set(c);
        return true;
    };
    if (name == "alpha") return fill(isalpha);
    if (name == "digit") return fill(isdigit);
    if (name == "alnum") return fill(isalnum);
    if (name == "upper") return fill(isupper);
    if (name == "lower") return fill(islower);
    if (name == "space") return fill(isspace);
    if (name == "blank") { s.set(' '); s.set('\t'); return true; }
    if (name == "punct") return fill(ispunct);
    if (name == "print") return fill(isprint);
    if (name == "graph") return fill(isgraph);
    if (name == "cntrl") return fill(iscntrl);
    if (name == "xdigit") return fill(isxdigit);
    if (name == "ascii") { add_range(s, 0, 127); return true; }
    if (name == "word") { fill(isalnum); s.set('_'); return true; }
    return false;
}

/* "[...]" at p[i]: the class, with i moved past the closing bracket */
ByteSet parse_bracket_class(const std::string &p, size_t &i) {
    ByteSet cls;
    size_t j = i + 1;
    bool neg = false;
    if (j < p.size() && p[j] == '^') {
        neg = true;
        j++;
    }
    bool first = true;
    for (;;) {
        if (j >= p.size()) throw ParseError{"Unterminated character class."};
        if (p[j] == ']' && !first) break;
        first = false;
        ByteSet item;
        unsigned lo;
        bool is_class = false;
        if (p[j] == '[' && j + 1 < p.size() && (p[j + 1] == '.' || p[j + 1] == '=') &&
            p.find(std::string(1, p[j + 1]) + "]", j + 2) != std::string::npos)
            throw ParseError{"Unsupported POSIX collating element."};
        if (p[j] == '[' && j + 1 < p.size() && p[j + 1] == ':') { /* [:alpha:] and friends */
            const size_t e = p.find(":]", j + 2);
            if (e == std::string::npos) throw ParseError{"Unterminated POSIX class."};
            std::string name = p.substr(j + 2, e - j - 2);
            const bool inv = !name.empty() && name[0] == '^';
            if (inv) name.erase(0, 1);
            if (!posix_class(name, item)) throw ParseError{"Unknown POSIX character class."};
            cls |= inv ? ~item : item;
            j = e + 2;
            continue;
        }
        if (p[j] == '\\') {
            if (j + 1 >= p.size()) throw ParseError{"Trailing backslash."};
            bool ok;
            item = class_escape(p[j + 1], ok);
            if (ok) {
                is_class = true;
                j += 2;
            } else {
                size_t k = j + 1;
                unsigned char lit;
                if (!class_char_escape(p, k, lit)) throw ParseError{"Unsupported escape sequence."};
                lo = lit;
                j = k;
            }
        } else {
            lo = (unsigned char)p[j++];
        }
        if (is_class) {
            cls |= item;
            continue;
        }
        unsigned hi = lo;
        if (j + 1 < p.size() && p[j] == '-' && p[j + 1] != ']') {
            j++;
            if (p[j] == '[' && j + 1 < p.size() && p[j + 1] == ':') throw ParseError{"Invalid range in character class."};
            if (p[j] == '\\') {
                size_t k = j + 1;
                unsigned char lit;
                if (k >= p.size() || !class_char_escape(p, k, lit)) throw ParseError{"Unsupported escape sequence."};
                hi = lit;
                j = k;
            } else {
                hi = (unsigned char)p[j++];
            }
            if (hi < lo) throw ParseError{"Range out of order in character class."};
        }
        add_range(cls, lo, hi);
    }
    i = j + 1;
    return neg ? ~cls : cls;
}

/* ---- general tails: groups and alternation ---------------------------------------
 * tail := alt ; alt := cat ('|' cat)* ; cat := rep* ; rep := atom ('?' | '*' | '+' | {m[,[n]]})?
 * atom := '(' ['?:'] alt ')' | '[' class ']' | '\\' escape | '.' | character
 * Built directly as a Glushkov automaton: every character-class occurrence is a position. */
/* position sets that grow with the automaton */
inline void bits_or(Bits &a, const Bits &b) {
    if (a.size() < b.size()) a.resize(b.size(), 0);
    for (size_t i = 0; i < b.size(); i++) a[i] |= b[i];
}
inline void bits_set(Bits &a, size_t i) {
    if (a.size() <= i / 64) a.resize(i / 64 + 1, 0);
    a[i / 64] |= 1ull << (i % 64);
}
inline bool bits_test(const Bits &a, size_t i) { return i / 64 < a.size() && (a[i / 64] >> (i % 64) & 1); }

/* conditions on a boundary between two bytes: 0 = none, 1 = \b, 2 = \B; -1 = contradictory */
inline int cond_and(int a, int b) { return a == 0 ? b : (b == 0 || a == b) ? a : -1; }

struct Frag {
    Bits first[3], last[3];     /* per condition: to enter a first position / after a last one */
    unsigned char nullmask = 1; /* bit c: the empty string is in the language under condition c */
    unsigned long long wmin = 0, wmax = 0;
    bool nullable() const { return nullmask != 0; }
};

struct TailBuilder {
    const std::string &p;
    bool nocase, dotall;
    std::vector<ByteSet> cls;    /* per position */
    std::vector<Bits> follow[3]; /* per condition, per position */
    bool has_cond = false;

    static unsigned long long add_w(unsigned long long a, unsigned long long b) {
        return (a == kInf64 || b == kInf64) ? kInf64 : a + b;
    }
    /* can an edge x -> y under condition c (1 = \b, 2 = \B) ever be taken? not when both classes
     * sit wholly on one side of the word / non-word divide in the wrong way */
    bool edge_possible(int c, size_t x, size_t y) const {
        if (c == 0) return true;
        static const ByteSet word = [] {
            ByteSet w;
            for (unsigned b = 0; b < 256; b++)
                if (is_word_char((unsigned char)b)) w.set(b);
            return w;
        }();
        const bool xw = (cls[x] & word).any(), xn = (cls[x] & ~word).any();
        const bool yw = (cls[y] & word).any(), yn = (cls[y] & ~word).any();
        const bool can_differ = (xw && yn) || (xn && yw), can_agree = (xw && yw) || (xn && yn);
        return c == 1 ? can_differ : can_agree;
    }
    unsigned new_pos(const ByteSet &c) {
        if (cls.size() >= kMaxPositions) throw ParseError{"Pattern too large."};
        cls.push_back(nocase ? fold_case(c) : c);
        for (int k = 0; k < 3; k++) follow[k].emplace_back();
        return (unsigned)cls.size() - 1;
    }
    /* every last position of a, then every first position of b: the conditions on the two sides
     * speak about the same boundary, so they combine */
    void link(const Frag &a, const Frag &b) {
        for (int ca = 0; ca < 3; ca++)
            for (int cb = 0; cb < 3; cb++) {
                const int c = cond_and(ca, cb);
                if (c < 0) continue;
                const Bits &from = a.last[ca];
                for (size_t w = 0; w < from.size(); w++)
                    for (unsigned long long m = from[w]; m; m &= m - 1)
                        bits_or(follow[c][w * 64 + __builtin_ctzll(m)], b.first[cb]);
            }
    }
    Frag cat(const Frag &a, const Frag &b) {
        link(a, b);
        Frag r;
        r.nullmask = 0;
        for (int c = 0; c < 3; c++) {
            r.first[c] = a.first[c];
            r.last[c] = b.last[c];
        }
        for (int n = 0; n < 3; n++)
            for (int c = 0; c < 3; c++) {
                const int k = cond_and(n, c);
                if (k < 0) continue;
                if (a.nullmask >> n & 1) bits_or(r.first[k], b.first[c]);
                if (b.nullmask >> n & 1) bits_or(r.last[k], a.last[c]);
                if ((a.nullmask >> n & 1) && (b.nullmask >> c & 1)) r.nullmask |= 1u << k;
            }
        r.wmin = add_w(a.wmin, b.wmin);
        r.wmax = add_w(a.wmax, b.wmax);
        return r;
    }
    static bool is_repeat_at(const std::string &p, size_t k) {
        if (k >= p.size() || p[k] != '{') return false;
        size_t j = k + 1, d = 0;
        while (j < p.size() && p[j] >= '0' && p[j] <= '9') j++, d++;
        if (d == 0) return false;
        if (j < p.size() && p[j] == ',') {
            j++;
            while (j < p.size() && p[j] >= '0' && p[j] <= '9') j++;
        }
        return j < p.size() && p[j] == '}';
    }

    Frag parse_alt(size_t &i, int depth) {
        Frag r = parse_cat(i, depth);
        while (i < p.size() && p[i] == '|') {
            i++;
            const Frag b = parse_cat(i, depth);
            for (int c = 0; c < 3; c++) {
                bits_or(r.first[c], b.first[c]);
                bits_or(r.last[c], b.last[c]);
            }
            r.nullmask |= b.nullmask;
            r.wmin = std::min(r.wmin, b.wmin);
            r.wmax = std::max(r.wmax, b.wmax);
        }
        return r;
    }
    Frag parse_cat(size_t &i, int depth) {
        Frag r; /* the empty string */
        while (i < p.size() && p[i] != '|' && p[i] != ')') r = cat(r, parse_rep(i, depth));
        return r;
    }
    /* one atom, then its quantifier; a counted repeat re-parses the atom's source for every copy
     * (fresh positions), optional copies being x? x? ... (same language as the nested form) */
    Frag parse_rep(size_t &i, int depth) {
        const size_t a0 = i;
        Frag f = parse_atom(i, depth);
        const size_t a1 = i;
        if (i >= p.size()) return f;
        unsigned lo = 1, hi = 1;
        const char q = p[i];
        if (q == '?') { lo = 0; hi = 1; i++; }
        else if (q == '*') { lo = 0; hi = kInf; i++; }
        else if (q == '+') { lo = 1; hi = kInf; i++; }
        else if (q == '{' && is_repeat_at(p, i)) {
            size_t j = i + 1;
            auto num = [&](unsigned &v) {
                v = 0;
                while (j < p.size() && p[j] >= '0' && p[j] <= '9') v = v * 10 + (p[j++] - '0');
                return v <= 1000;
            };
            if (!num(lo)) throw ParseError{"Malformed repeat."};
            hi = lo;
            if (p[j] == ',') {
                j++;
                if (p[j] == '}') hi = kInf;
                else if (!num(hi) || hi < lo) throw ParseError{"Malformed repeat."};
            }
            i = j + 1;
        } else {
            return f;
        }
        if (i < p.size() && p[i] == '?') i++; /* lazy: every end offset is reported anyway, greed is immaterial */
        else if (i < p.size() && p[i] == '+') throw ParseError{"Possessive quantifiers are not supported."};
        auto again = [&]() { /* a fresh copy of the atom */
            size_t k = a0;
            Frag c = parse_atom(k, depth);
            (void)a1;
            return c;
        };
        auto star_of = [&](Frag c) { /* c* */
            link(c, c);
            c.nullmask |= 1;
            c.wmin = 0;
            c.wmax = c.wmax ? kInf64 : 0;
            return c;
        };
        auto opt_of = [&](Frag c) {
            c.nullmask |= 1;
            c.wmin = 0;
            return c;
        };
        /* copies: the already parsed one is copy #1 */
        Frag r;
        bool used_first = false;
        auto next_copy = [&]() {
            if (!used_first) { used_first = true; return f; }
            return again();
        };
        if (lo == 0 && hi == kInf) return star_of(next_copy());
        for (unsigned k = 0; k < lo; k++) {
            Frag c = next_copy();
            if (hi == kInf && k + 1 == lo) { /* last mandatory copy loops: c+ */
                link(c, c);
                c.wmax = c.wmax ? kInf64 : 0;
            }
            r = cat(r, c);
        }
        if (hi != kInf)
            for (unsigned k = lo; k < hi; k++) r = cat(r, opt_of(next_copy()));
        if (!used_first) { /* {0} or {0,0}: the atom is parsed but contributes nothing */
            Frag none;
            return none;
        }
        return r;
    }
    Frag parse_atom(size_t &i, int depth) {
        if (i >= p.size()) throw ParseError{"Unexpected end of pattern."};
        const unsigned char c = (unsigned char)p[i];
        if (c == '(') {
            if (depth > 20) throw ParseError{"Groups nested too deeply."};
            i++;
            if (i + 1 < p.size() && p[i] == '?') {
                const char k = p[i + 1];
                if (k == ':') {
                    i += 2;
                } else if (k == '#') { /* (?# comment ) */
                    const size_t e = p.find(')', i);
                    if (e == std::string::npos) throw ParseError{"Missing closing parenthesis."};
                    i = e + 1;
                    return Frag();
                } else if ((k == '<' && i + 2 < p.size() && p[i + 2] != '=' && p[i + 2] != '!') || k == '\'' ||
                           (k == 'P' && i + 2 < p.size() && p[i + 2] == '<')) { /* named group: a plain group here */
                    const char close = k == '\'' ? '\'' : '>';
                    const size_t e = p.find(close, i + (k == 'P' ? 3 : 2));
                    if (e == std::string::npos) throw ParseError{"Unterminated group name."};
                    i = e + 1;
                } else {
                    throw ParseError{"Only plain, named and (?:...) groups are supported."};
                }
            }
            Frag f = parse_alt(i, depth + 1);
            if (i >= p.size() || p[i] != ')') throw ParseError{"Missing closing parenthesis."};
            i++;
            return f;
        }
        if (c == '\\' && i + 1 < p.size() && (p[i + 1] == 'b' || p[i + 1] == 'B')) {
            Frag f; /* zero width: the empty string, under a condition */
            f.nullmask = p[i + 1] == 'b' ? 2 : 4;
            has_cond = true;
            i += 2;
            return f;
        }
        ByteSet set;
        if (c == '\\') {
            if (i + 1 >= p.size()) throw ParseError{"Trailing backslash."};
            bool ok;
            set = class_escape(p[i + 1], ok);
            if (ok) {
                i += 2;
            } else {
                size_t j = i + 1;
                unsigned char lit;
                if (!char_escape(p, j, lit)) throw ParseError{"Unsupported escape sequence."};
                set.set(lit);
                i = j;
            }
        } else if (c == '.') {
            set.set();
            if (!dotall) set.reset('\n');
            i++;
        } else if (c == '[') {
            set = parse_bracket_class(p, i);
        } else if (strchr(")|^$*+?", c) || (c == '{' && is_repeat_at(p, i))) {
            throw ParseError{std::string("Unsupported regex construct '") + (char)c + "'."};
        } else {
            set.set(c);
            i++;
        }
        const unsigned pos = new_pos(set);
        Frag f;
        bits_set(f.first[0], pos);
        bits_set(f.last[0], pos);
        f.nullmask = 0;
        f.wmin = f.wmax = 1;
        return f;
    }
};

/* builder output -> the runtime form; `reversed`: the automaton of the reversed language, with
 * the positions renumbered back to front so that its chains are left shifts again */
Auto finish_auto(const TailBuilder &tb, const Frag &f, bool reversed) {
    Auto a;
    const size_t n = tb.cls.size(), W = (n + 63) / 64;
    a.npos = n;
    a.W = W;
    a.nullable = f.nullmask & 1;
    a.cnullable[0] = f.nullmask & 2;
    a.cnullable[1] = f.nullmask & 4;
    a.wmin = f.wmin;
    a.wmax = f.wmax;
    a.has_cond = tb.has_cond;
    if (!n) return a;
    auto idx = [&](size_t p) { return reversed ? n - 1 - p : p; };
    a.reach.assign(256 * W, 0);
    for (size_t x = 0; x < n; x++)
        for (unsigned c = 0; c < 256; c++)
            if (tb.cls[x][c]) a.reach[c * W + idx(x) / 64] |= 1ull << (idx(x) % 64);
    for (int layer = 0; layer < 3; layer++) { /* 0 = unconditional, 1 = \b, 2 = \B */
        if (layer && !tb.has_cond) break;
        Bits &first = layer ? a.cfirst[layer - 1] : a.first, &last = layer ? a.clast[layer - 1] : a.last;
        Bits &shift_ok = layer ? a.cshift_ok[layer - 1] : a.shift_ok, &exc = layer ? a.cexc[layer - 1] : a.exc;
        std::vector<Bits> &exc_row = layer ? a.cexc_row[layer - 1] : a.exc_row;
        std::vector<Bits> follow(n, Bits(W, 0));
        for (size_t x = 0; x < n; x++)
            for (size_t y = 0; y < n; y++)
                if (bits_test(tb.follow[layer][x], y) && tb.edge_possible(layer, x, y)) {
                    if (reversed) bits_set(follow[idx(y)], idx(x));
                    else bits_set(follow[x], y);
                }
        first.assign(W, 0);
        last.assign(W, 0);
        for (size_t x = 0; x < n; x++) {
            if (bits_test(f.first[layer], x)) bits_set(reversed ? last : first, idx(x));
            if (bits_test(f.last[layer], x)) bits_set(reversed ? first : last, idx(x));
        }
        shift_ok.assign(W, 0);
        exc.assign(W, 0);
        exc_row.assign(n, Bits());
        for (size_t x = 0; x < n; x++) {
            Bits row = follow[x];
            if (x + 1 < n && bits_test(row, x + 1)) {
                bits_set(shift_ok, x);
                row[(x + 1) / 64] &= ~(1ull << ((x + 1) % 64));
            }
            bool any = false;
            for (unsigned long long w : row) any |= w != 0;
            if (any) {
                bits_set(exc, x);
                exc_row[x] = row;
            }
        }
    }
    return a;
}

/* is anything in the fragment's language? (an empty class, or stacked contradictory assertions,
 * can leave none: the reference refuses such patterns, "Pattern can never match.") */
bool language_nonempty(const TailBuilder &tb, const Frag &f) {
    if (f.nullmask) return true;
    const size_t n = tb.cls.size();
    std::vector<char> seen(n, 0);
    std::vector<size_t> todo;
    auto visit = [&](size_t x) {
        if (!seen[x] && tb.cls[x].any()) {
            seen[x] = 1;
            todo.push_back(x);
        }
    };
    for (int c = 0; c < 3; c++)
        for (size_t x = 0; x < n; x++)
            if (bits_test(f.first[c], x)) visit(x);
    while (!todo.empty()) {
        const size_t x = todo.back();
        todo.pop_back();
        for (int c = 0; c < 3; c++) {
            if (bits_test(f.last[c], x)) return true;
            for (size_t y = 0; y < n; y++)
                if (bits_test(tb.follow[c][x], y) && tb.edge_possible(c, x, y)) visit(y);
        }
    }
    return false;
}

Auto compile_auto(const std::string &src, bool nocase, bool dotall, bool reversed = false) {
    TailBuilder tb{src, nocase, dotall, {}, {}, false};
    size_t i = 0;
    const Frag f = tb.parse_cat(i, 0);
    if (i < src.size()) throw ParseError{"Unmatched closing parenthesis."};
    if (!language_nonempty(tb, f)) throw NeverMatch();
    return finish_auto(tb, f, reversed);
}

/* the longest run of plain characters at the top level of a branch (not inside a group or a
 * class, not quantified): [begin, end) in the source and the bytes; the earliest of equals */
struct LitRun {
    size_t begin = 0, end = 0;
    std::string bytes;
};

LitRun longest_literal_run(const std::string &p) {
    LitRun best, cur;
    auto close = [&]() {
        if (cur.bytes.size() > best.bytes.size()) best = cur;
        cur = LitRun();
    };
    auto skip_quant = [&](size_t &k) {
        const size_t k0 = k;
        if (k < p.size() && (p[k] == '?' || p[k] == '*' || p[k] == '+')) k++;
        else if (TailBuilder::is_repeat_at(p, k)) k = p.find('}', k) + 1;
        if (k != k0 && k < p.size() && (p[k] == '?' || p[k] == '+')) k++; /* lazy / possessive marker */
    };
    size_t i = 0;
    while (i < p.size()) {
        const unsigned char c = (unsigned char)p[i];
        size_t j = i;
        unsigned char lit = 0;
        bool is_lit = false;
        if (c == '\\') {
            if (i + 1 >= p.size()) throw ParseError{"Trailing backslash."};
            bool ok;
            class_escape(p[i + 1], ok);
            if (ok || strchr("bBAzZ", p[i + 1])) { /* a class, or a zero-width assertion: ends the run */
                j = i + 2;
            } else {
                j = i + 1;
                if (!char_escape(p, j, lit)) throw ParseError{"Unsupported escape sequence."};
                is_lit = true;
            }
        } else if (c == '[') {
            parse_bracket_class(p, j);
        } else if (c == '(') {
            int depth = 0;
            for (;; j++) {
                if (j >= p.size()) throw ParseError{"Missing closing parenthesis."};
                if (p[j] == '\\') { j++; continue; }
                if (p[j] == '[') { size_t e = j; parse_bracket_class(p, e); j = e - 1; continue; }
                if (p[j] == '(') depth++;
                if (p[j] == ')' && --depth == 0) break;
            }
            j++;
        } else if (c == '.' || strchr(")|^$*+?", c) || (c == '{' && TailBuilder::is_repeat_at(p, i))) {
            j = i + 1; /* not a literal; a misplaced operator is reported by the fragment compiler */
        } else {
            lit = c;
            j = i + 1;
            is_lit = true;
        }
        const size_t after = j;
        skip_quant(j);
        if (is_lit && j == after) {
            if (cur.bytes.empty()) cur.begin = i;
            cur.bytes.push_back((char)lit);
            cur.end = j;
        } else {
            close();
        }
        i = j;
    }
    close();
    return best;
}

constexpr unsigned kAllFlags = 0x7ff; /* HS_FLAG_ALL: the eleven flags of src/hs_compile.h */

/* the reference's own flag rules, in its order (src/compiler/compiler.cpp:286-294,166-196) */
void check_flags(unsigned flags, bool literal_api) {
    if (!literal_api && (flags & HS_FLAG_COMBINATION)) {
        if (flags & ~(HS_FLAG_COMBINATION | HS_FLAG_QUIET | HS_FLAG_SINGLEMATCH))
            throw ParseError{"only HS_FLAG_QUIET and HS_FLAG_SINGLEMATCH are supported in combination with "
                             "HS_FLAG_COMBINATION."};
        throw ParseError{"Logical combinations are not supported by the GPU literal engine."};
    }
    if (!literal_api && (flags & HS_FLAG_QUIET) && (flags & HS_FLAG_SOM_LEFTMOST))
        throw ParseError{"HS_FLAG_QUIET is not supported in combination with HS_FLAG_SOM_LEFTMOST."};
    if (flags & ~kAllFlags) throw ParseError{"Unrecognised flag."};
    if ((flags & HS_FLAG_SINGLEMATCH) && (flags & HS_FLAG_SOM_LEFTMOST))
        throw ParseError{"HS_FLAG_SINGLEMATCH is not supported in combination with HS_FLAG_SOM_LEFTMOST."};
    if (!literal_api && (flags & HS_FLAG_PREFILTER) && (flags & HS_FLAG_SOM_LEFTMOST))
        throw ParseError{"HS_FLAG_PREFILTER is not supported in combination with HS_FLAG_SOM_LEFTMOST."};
}

/* branch := '^'? literal-prefix tail '$'? ; tail := (atom quantifier?)* */
Pattern parse_branch(const std::string &src, unsigned flags, unsigned id) {
    Pattern pat;
    std::string p = src;
    const bool multiline = flags & HS_FLAG_MULTILINE;
    /* is the two-character escape "\\<c>" at p[k], with its backslash not itself escaped? */
    auto escape_at = [&](size_t k, const char *cs) {
        if (k + 1 >= p.size() || p[k] != '\\' || !strchr(cs, p[k + 1])) return false;
        size_t bs = 0;
        while (bs < k && p[k - 1 - bs] == '\\') bs++;
        return bs % 2 == 0;
    };
    auto assertion = [](char c) { return (unsigned char)(c == 'b' ? 1 : 2); };
    /* front: ^ or \A, then \b / \B */
    if (!p.empty() && p[0] == '^') {
        pat.bol = true;
        pat.bol_ml = multiline;
        p.erase(0, 1);
    } else if (escape_at(0, "A")) {
        pat.bol = true;
        p.erase(0, 2);
    }
    if (escape_at(0, "bB")) {
        pat.as_start = assertion(p[1]);
        p.erase(0, 2);
    }
    if (!p.empty() && (strchr("*+?", p[0]) || TailBuilder::is_repeat_at(p, 0)))
        throw ParseError{"Invalid repeat."}; /* a quantifier needs something that consumes bytes before it */
    /* back: $, \z or \Z, before it \b / \B */
    if (!p.empty() && p.back() == '$' && !escape_at(p.size() - 2, "$")) {
        pat.eol = pat.eol_nl = true;
        pat.eol_ml = multiline;
        p.pop_back();
    } else if (p.size() >= 2 && escape_at(p.size() - 2, "zZ")) {
        pat.eol = true;
        pat.eol_nl = p.back() == 'Z';
        p.erase(p.size() - 2);
    }
    if (p.size() >= 2 && escape_at(p.size() - 2, "bB")) {
        pat.as_end = assertion(p.back());
        p.erase(p.size() - 2);
    }
    pat.nocase = flags & HS_FLAG_CASELESS;
    pat.single = flags & HS_FLAG_SINGLEMATCH;
    pat.som = flags & HS_FLAG_SOM_LEFTMOST;
    pat.id = id;
    const bool dotall = flags & HS_FLAG_DOTALL;
    /* `{` opens a repeat only when a well-formed {m}, {m,} or {m,n} follows; otherwise it (and
     * a lone `}`) is an ordinary character, as in PCRE ("foo.{,10}bar" is twelve literal-ish
     * positions: unit/hyperscan/expr_info.cpp:211) */
    auto is_repeat = [&](size_t k) { return TailBuilder::is_repeat_at(p, k); };
    /* the branch is R1 LIT R2 around its longest top-level literal run (the front one on a tie);
     * \b / \B may hug the literal on either side */
    LitRun run = longest_literal_run(p);
    if (run.bytes.empty()) throw NoLiteral();
    size_t r1_end = run.begin, r2_begin = run.end;
    if (run.begin >= 2 && escape_at(run.begin - 2, "bB")) {
        pat.as_lit_pre = assertion(p[run.begin - 1]);
        r1_end -= 2;
    }
    if (escape_at(run.end, "bB")) {
        pat.as_lit_post = assertion(p[run.end + 1]);
        r2_begin += 2;
        if (r2_begin < p.size() && (strchr("*+?", p[r2_begin]) || TailBuilder::is_repeat_at(p, r2_begin)))
            throw ParseError{"Invalid repeat."};
    }
    pat.lit = run.bytes;
    if (pat.bol && !pat.bol_ml && r1_end == 0) {
        /* the match starts at offset 0: what is in front is not a word byte */
        const unsigned char k = pat.as_start ? pat.as_start : pat.as_lit_pre;
        if (k && (is_word_char((unsigned char)pat.lit[0]) != (k == 1))) throw NeverMatch();
    }
    if (pat.as_start && pat.as_lit_pre && r1_end == 0 && pat.as_start != pat.as_lit_pre)
        throw NeverMatch();
    if (pat.as_end && pat.as_lit_post && r2_begin == p.size() && pat.as_end != pat.as_lit_post)
        throw NeverMatch();
    if (pat.as_lit_post && r2_begin < p.size() && !strchr("\\.[]()|^$*+?{", p[r2_begin]) &&
        !(r2_begin + 1 < p.size() && (strchr("*+?", p[r2_begin + 1]) || TailBuilder::is_repeat_at(p, r2_begin + 1)))) {
        /* literal, assertion, plain character: both neighbours of the boundary are known */
        unsigned char nx = (unsigned char)p[r2_begin];
        const bool differ = is_word_char((unsigned char)pat.lit.back()) != is_word_char(nx);
        if (differ != (pat.as_lit_post == 1)) throw NeverMatch();
    }
    size_t i = r2_begin;
    if (r1_end != 0) {
        /* the literal is not at the front: R1 backwards, R2 as a position automaton */
        pat.pre = compile_auto(p.substr(0, r1_end), pat.nocase, dotall, true);
        pat.g = compile_auto(p.substr(r2_begin), pat.nocase, dotall);
        pat.has_pre = pat.pre.npos != 0 || pat.pre.has_cond;
        pat.general = pat.g.npos != 0 || pat.g.has_cond;
        pat.tail_nullable = pat.g.nullable || pat.g.cnullable[0] || pat.g.cnullable[1];
        return pat;
    }
    /* a tail with a group in it goes to the position automaton; the linear form below stays the
     * path for everything it can express */
    bool grouped = false;
    for (size_t k = i; k < p.size() && !grouped; k++) {
        if (p[k] == '\\') { grouped = k + 1 < p.size() && (p[k + 1] == 'b' || p[k + 1] == 'B'); k++; }
        else if (p[k] == '[') { size_t e = k; parse_bracket_class(p, e); k = e - 1; }
        else if (p[k] == '(') grouped = true;
    }
    if (grouped) {
        pat.g = compile_auto(p.substr(i), pat.nocase, dotall);
        pat.general = pat.g.npos != 0 || pat.g.has_cond;
        pat.tail_nullable = pat.g.nullable || pat.g.cnullable[0] || pat.g.cnullable[1];
        return pat;
    }
    const size_t tail_begin = i;
    /* tail */
    while (i < p.size()) {
        ByteSet cls;
        unsigned char c = (unsigned char)p[i];
        if (c == '\\') {
            if (i + 1 >= p.size()) throw ParseError{"Trailing backslash."};
            bool ok;
            cls = class_escape(p[i + 1], ok);
            if (ok) {
                i += 2;
            } else {
                size_t j = i + 1;
                unsigned char lit;
                if (!char_escape(p, j, lit)) throw ParseError{"Unsupported escape sequence."};
                cls.set(lit);
                i = j;
            }
        } else if (c == '.') {
            cls.set();
            if (!dotall) cls.reset('\n');
            i++;
        } else if (c == '[') {
            cls = parse_bracket_class(p, i);
        } else if (strchr("()|^$*+?]", c) || (c == '{' && is_repeat(i))) {
            throw ParseError{std::string("Unsupported regex construct '") + (char)c + "'."};
        } else {
            cls.set(c);
            i++;
        }
        if (pat.nocase) cls = fold_case(cls);
        /* quantifier */
        unsigned lo = 1, hi = 1;
        if (i < p.size()) {
            char q = p[i];
            if (q == '?') { lo = 0; hi = 1; i++; }
            else if (q == '*') { lo = 0; hi = kInf; i++; }
            else if (q == '+') { lo = 1; hi = kInf; i++; }
            else if (q == '{' && is_repeat(i)) {
                size_t j = i + 1;
                auto num = [&](unsigned &v) {
                    if (j >= p.size() || p[j] < '0' || p[j] > '9') return false;
                    v = 0;
                    while (j < p.size() && p[j] >= '0' && p[j] <= '9') v = v * 10 + (p[j++] - '0');
                    return v <= 1000;
                };
                if (!num(lo)) throw ParseError{"Malformed repeat."};
                hi = lo;
                if (j < p.size() && p[j] == ',') {
                    j++;
                    if (j < p.size() && p[j] == '}') hi = kInf;
                    else if (!num(hi) || hi < lo) throw ParseError{"Malformed repeat."};
                }
                if (j >= p.size() || p[j] != '}') throw ParseError{"Malformed repeat."};
                i = j + 1;
            }
            if (i < p.size() && p[i] == '?') i++; /* lazy: immaterial, as above */
            else if (i < p.size() && p[i] == '+') throw ParseError{"Possessive quantifiers are not supported."};
        }
        if (lo > 0 && cls.none()) throw NeverMatch();
        for (unsigned k = 0; k < lo; k++) pat.tail.push_back(Unit{cls, false, false});
        if (hi == kInf) {
            if (lo == 0) pat.tail.push_back(Unit{cls, true, true});
            else pat.tail.back().star = true;
        } else {
            for (unsigned k = lo; k < hi; k++) pat.tail.push_back(Unit{cls, true, false});
        }
        if (pat.tail.size() > 63) { /* too long for one shift-and word: the position automaton takes it */
            pat.tail.clear();
            pat.g = compile_auto(p.substr(tail_begin), pat.nocase, dotall);
            pat.general = true;
            pat.tail_nullable = pat.g.nullable || pat.g.cnullable[0] || pat.g.cnullable[1];
            return pat;
        }
    }
    pat.tail_nullable = true;
    for (const Unit &u : pat.tail) pat.tail_nullable &= u.optional;
    return pat;
}

/* A branch without a top-level literal may still hold one inside an alternation: X(A|B)Y is
 * XAY|XBY, so the first unquantified top-level group is distributed over its alternatives and
 * every product is tried again (Rose gets the same literals by cutting the graph at the
 * alternation). "\\b(foo|bar)\\b" and "(GET|POST) /" are the everyday cases. */
constexpr size_t kMaxBranches = 256;
void distribute_group(const std::string &b, unsigned flags, unsigned id, std::vector<Pattern> &out, int depth);

void expand_branch(const std::string &b, unsigned flags, unsigned id, std::vector<Pattern> &out, int depth) {
    /* as written, if it has a usable literal; with only a 1-2 byte one, distributing a group is
     * tried as well and kept when every product gets a longer literal ("(GET|POST) /x": " /" ->
     * "GET /", "POST /") */
    bool have = false;
    Pattern whole;
    try {
        whole = parse_branch(b, flags, id);
        have = true;
    } catch (const NeverMatch &) {
        throw;
    } catch (const NoLiteral &) {
        if (depth >= 8) throw;
    } catch (const ParseError &first) {
        /* an anchor or assertion inside a group, e.g. "(^|\n)foo": fine once the group is distributed */
        if (depth >= 8) throw;
        try {
            distribute_group(b, flags, id, out, depth);
        } catch (const ParseError &) {
            throw first;
        }
        return;
    }
    if (have && (whole.lit.size() > 2 || depth >= 8)) {
        out.push_back(std::move(whole));
        return;
    }
    if (have) {
        std::vector<Pattern> alt;
        bool better = false;
        try {
            distribute_group(b, flags, id, alt, depth);
            better = !alt.empty() && out.size() + alt.size() <= kMaxBranches;
            for (const Pattern &a : alt) better = better && a.lit.size() > whole.lit.size();
        } catch (const ParseError &) {
            better = false;
        }
        if (better) {
            for (Pattern &a : alt) out.push_back(std::move(a));
        } else {
            out.push_back(std::move(whole));
        }
        return;
    }
    distribute_group(b, flags, id, out, depth);
}

void distribute_group(const std::string &b, unsigned flags, unsigned id, std::vector<Pattern> &out, int depth) {
    /* the first top-level group that is not quantified, not special, and has its alternatives */
    size_t i = 0;
    while (i < b.size()) {
        const char c = b[i];
        if (c == '\\') { i += 2; continue; }
        if (c == '[') { size_t e = i; parse_bracket_class(b, e); i = e; continue; }
        if (c != '(') { i++; continue; }
        size_t body = i + 1;
        bool plain = true;
        if (body < b.size() && b[body] == '?') {
            if (body + 1 < b.size() && b[body + 1] == ':') body += 2;
            else plain = false; /* named / comment / look-around: leave it alone */
        }
        std::vector<std::string> alts;
        size_t j = i, last = body;
        for (int d = 0;; j++) {
            if (j >= b.size()) throw ParseError{"Missing closing parenthesis."};
            if (b[j] == '\\') { j++; continue; }
            if (b[j] == '[') { size_t e = j; parse_bracket_class(b, e); j = e - 1; continue; }
            if (b[j] == '(') d++;
            else if (b[j] == '|' && d == 1) { alts.push_back(b.substr(last, j - last)); last = j + 1; }
            else if (b[j] == ')' && --d == 0) { alts.push_back(b.substr(last, j - last)); break; }
        }
        size_t end = j + 1;
        if (plain && end < b.size() && b[end] == '?' && !(end + 1 < b.size() && b[end + 1] == '+')) {
            alts.push_back(std::string()); /* (X)? is (X|): distributable too */
            end += (end + 1 < b.size() && b[end + 1] == '?') ? 2 : 1;
        }
        const bool quantified = end < b.size() && (b[end] == '?' || b[end] == '*' || b[end] == '+' || TailBuilder::is_repeat_at(b, end));
        if (!plain || quantified) { i = end; continue; }
        for (const std::string &a : alts) /* "(*VERB)", "(+x)": not ours to rearrange */
            if (!a.empty() && (strchr("*+?", a[0]) || TailBuilder::is_repeat_at(a, 0))) throw NoLiteral();
        if (out.size() + alts.size() > kMaxBranches) throw ParseError{"Pattern too large."};
        const size_t before = out.size();
        for (const std::string &a : alts) {
            try {
                expand_branch(b.substr(0, i) + a + b.substr(end), flags, id, out, depth + 1);
            } catch (const NeverMatch &) { /* this product contributes nothing */
            }
        }
        if (out.size() == before) throw NeverMatch();
        return;
    }
    throw NoLiteral();
}

/* expression := branch ('|' branch)* at the top level: every branch is its own literal-prefixed
 * pattern reporting the same id (the reference builds one graph; the reports are the same) */
std::vector<Pattern> parse_pattern(const std::string &expr, unsigned flags, unsigned id) {
    check_flags(flags, false);
    /* leading inline options: (?i) (?s) (?m), combined and negated forms */
    std::string p = expr;
    while (p.size() >= 4 && p[0] == '(' && p[1] == '?') {
        size_t k = 2;
        bool on = true, any = false;
        unsigned set = 0, clear = 0;
        for (; k < p.size() && strchr("ims-", p[k]); k++) {
            if (p[k] == '-') { on = false; continue; }
            const unsigned f = p[k] == 'i' ? HS_FLAG_CASELESS : p[k] == 's' ? HS_FLAG_DOTALL : HS_FLAG_MULTILINE;
            (on ? set : clear) |= f;
            any = true;
        }
        if (!any || k >= p.size() || p[k] != ')') break;
        flags = (flags | set) & ~clear;
        p.erase(0, k + 1);
    }
    /* HS_FLAG_PREFILTER allows a superset of the matches: the exact set is one. HS_FLAG_ALLOWEMPTY
     * permits patterns that can match the empty string: none here can (a mandatory literal).
     * HS_FLAG_QUIET: the expression reports nothing (src/hs_compile.h:328-330 "ignore match reporting"). */
    const unsigned unsupported = HS_FLAG_UTF8 | HS_FLAG_UCP | HS_FLAG_COMBINATION;
    if (flags & unsupported) throw ParseError{"Unsupported flag for the GPU literal engine."};
    std::vector<Pattern> out;
    size_t from = 0;
    int depth = 0;
    for (size_t k = 0; k <= p.size(); k++) {
        if (k < p.size()) {
            const char c = p[k];
            if (c == '\\') { k++; continue; }
            if (c == '[') { /* skip the class: "]" first in a class is a member */
                size_t j = k + 1;
                if (j < p.size() && p[j] == '^') j++;
                if (j < p.size() && p[j] == ']') j++;
                while (j < p.size() && p[j] != ']') j += p[j] == '\\' ? 2 : 1;
                k = j;
                continue;
            }
            if (c == '(') depth++;
            if (c == ')') depth--;
            if (c != '|' || depth != 0) continue;
        }
        try {
            expand_branch(p.substr(from, k - from), flags, id, out, 0);
        } catch (const NeverMatch &) { /* an alternative that cannot match is dropped; all of them: an error */
        }
        from = k + 1;
    }
    if (out.empty()) throw NeverMatch();
    for (Pattern &b : out) b.quiet = flags & HS_FLAG_QUIET;
    return out;
}

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
        if (cur & accept) { if (!report(pos)) return; }
        while (pos < len && (cur & (accept - 1))) {
            const unsigned long long live = cur & p.reach[buf[pos++]];
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

void finish_pattern(Pattern &p) {
    if (p.general) return;
    p.fast = !p.tail.empty(); /* parse_branch keeps linear tails to <= 63 units */
    if (!p.fast) return;
    p.reach.assign(256, 0);
    for (size_t i = 0; i < p.tail.size(); i++) {
        for (unsigned c = 0; c < 256; c++)
            if (p.tail[i].cls[c]) p.reach[c] |= 1ull << i;
        if (p.tail[i].star) p.star_mask |= 1ull << i;
        if (p.tail[i].optional) p.opt_mask |= 1ull << i;
    }
}

} // namespace

struct hs_database {
    unsigned magic = 0x48534744; /* "HSGD" */
    unsigned mode = HS_MODE_BLOCK;
    std::vector<Pattern> pats;
    hsgpu_hwlm_t *hwlm = nullptr;
    size_t min_width = 0;
    std::vector<std::string> sources; /* for serialisation: original expressions + flags */
    std::vector<unsigned> src_flags, src_ids;
    std::vector<unsigned char> src_is_lit;
    std::vector<hs_expr_ext_t> src_ext; /* flags == 0: none */
    std::set<unsigned> single_ids;      /* report ids carrying HS_FLAG_SINGLEMATCH (built once) */
};

struct hs_scratch {
    unsigned magic = 0x48534753; /* "HSGS" */
    hsgpu_scratch_t *gpu = nullptr;
    bool in_use = false;
    /* record buffer: grown on demand, never value-initialised (a std::vector::resize of the
     * worst-case capacity cost more than the scan) */
    hsgpu_match_t *recs = nullptr;
    size_t recs_cap = 0;
    ~hs_scratch() { free(recs); }
    bool reserve(size_t n) {
        if (n <= recs_cap) return true;
        free(recs);
        recs = (hsgpu_match_t *)malloc(n * sizeof(hsgpu_match_t));
        recs_cap = recs ? n : 0;
        return recs != nullptr;
    }
};

namespace {

/* allocation hooks, src/alloc.c / src/hs_common.h:273-439 */
struct Hooks {
    hs_alloc_t alloc = nullptr;
    hs_free_t free = nullptr;
};
Hooks g_db, g_misc, g_scratch, g_stream;
void *hook_alloc(const Hooks &h, size_t n) { return h.alloc ? h.alloc(n) : malloc(n); }
void hook_free(const Hooks &h, void *p) {
    if (!p) return;
    if (h.free) h.free(p);
    else free(p);
}
/* hs_check_alloc, src/alloc.c: NULL -> HS_NOMEM, misaligned -> HS_BAD_ALIGN */
hs_error_t check_alloc(const void *p) {
    if (!p) return HS_NOMEM;
    return ((uintptr_t)p & 7) ? HS_BAD_ALIGN : HS_SUCCESS;
}
char *misc_strdup(const std::string &sv) {
    char *m = (char *)hook_alloc(g_misc, sv.size() + 1);
    if (m) memcpy(m, sv.c_str(), sv.size() + 1);
    return m;
}

hs_compile_error_t *make_error(const std::string &msg, int expr) {
    hs_compile_error_t *e = (hs_compile_error_t *)hook_alloc(g_misc, sizeof(*e));
    if (!e) return nullptr;
    e->message = misc_strdup(msg);
    e->expression = expr;
    return e;
}

void destroy_db(hs_database *d);

/* hs_expr_ext_t validation as in src/compiler/compiler.cpp:97-130 (flags known, bounds
 * consistent); approximate matching belongs to the graph compiler and is refused */
void apply_ext(Pattern &p, const hs_expr_ext_t &e) {
    const unsigned long long known = HS_EXT_FLAG_MIN_OFFSET | HS_EXT_FLAG_MAX_OFFSET | HS_EXT_FLAG_MIN_LENGTH |
                                     HS_EXT_FLAG_EDIT_DISTANCE | HS_EXT_FLAG_HAMMING_DISTANCE;
    if (e.flags & ~known) throw ParseError{"Invalid hs_expr_ext flag set."};
    if (e.flags & (HS_EXT_FLAG_EDIT_DISTANCE | HS_EXT_FLAG_HAMMING_DISTANCE))
        throw ParseError{"Approximate matching (edit/Hamming distance) is not supported by the GPU literal engine."};
    if ((e.flags & HS_EXT_FLAG_MIN_OFFSET) && (e.flags & HS_EXT_FLAG_MAX_OFFSET) && e.min_offset > e.max_offset)
        throw ParseError{"In hs_expr_ext, min_offset must be less than or equal to max_offset."};
    if ((e.flags & HS_EXT_FLAG_MIN_LENGTH) && (e.flags & HS_EXT_FLAG_MAX_OFFSET) && e.min_length > e.max_offset)
        throw ParseError{"In hs_expr_ext, min_length must be less than or equal to max_offset."};
    p.ext_flags = e.flags;
    p.min_offset = e.min_offset;
    p.max_offset = e.max_offset;
    p.min_length = e.min_length;
}

/* width of one branch as written (no ext parameters): [lo, hi], hi meaningless when inf */
void raw_widths(const Pattern &p, unsigned long long &lo, unsigned long long &hi, bool &inf) {
    lo = hi = p.lit.size();
    inf = false;
    for (const Unit &u : p.tail) {
        lo += u.optional ? 0 : 1;
        hi += 1;
        inf |= u.star;
    }
    if (p.general) {
        lo += p.g.wmin;
        inf |= p.g.wmax == kInf64;
        if (p.g.wmax != kInf64) hi += p.g.wmax;
    }
    if (p.has_pre) {
        lo += p.pre.wmin;
        inf |= p.pre.wmax == kInf64;
        if (p.pre.wmax != kInf64) hi += p.pre.wmax;
    }
}

/* ext parameters no match could satisfy, with the reference's messages
 * (propagateExtendedParams, src/nfagraph/ng_extparam.cpp:871-905) */
void check_ext_widths(const Pattern *b, size_t n, const hs_expr_ext_t &e) {
    unsigned long long minw = kInf64, maxw = 0;
    bool unbounded = false, anchored = true;
    for (size_t k = 0; k < n; k++) {
        unsigned long long lo, hi;
        bool inf;
        raw_widths(b[k], lo, hi, inf);
        minw = std::min(minw, lo);
        maxw = std::max(maxw, hi);
        unbounded |= inf;
        anchored &= b[k].bol && !b[k].bol_ml;
    }
    char msg[200];
    if ((e.flags & HS_EXT_FLAG_MIN_OFFSET) && anchored && !unbounded && e.min_offset > maxw) {
        snprintf(msg, sizeof(msg), "Expression is anchored and cannot satisfy min_offset=%llu as it can only produce matches "
                 "of length %llu bytes at most.", e.min_offset, maxw);
        throw ParseError{msg};
    }
    if ((e.flags & HS_EXT_FLAG_MAX_OFFSET) && minw > e.max_offset) {
        snprintf(msg, sizeof(msg), "Expression has max_offset=%llu but requires %llu bytes to match.", e.max_offset, minw);
        throw ParseError{msg};
    }
    if ((e.flags & HS_EXT_FLAG_MIN_LENGTH) && !unbounded && maxw < e.min_length) {
        snprintf(msg, sizeof(msg), "Expression has min_length=%llu but can only produce matches of length %llu bytes at most.",
                 e.min_length, maxw);
        throw ParseError{msg};
    }
}

hs_error_t build_database(const std::vector<std::string> &exprs, const std::vector<unsigned char> &is_lit,
                          const unsigned *flags, const unsigned *ids, const hs_expr_ext_t *const *ext, unsigned mode,
                          hs_database_t **db, hs_compile_error_t **error,
                          const std::vector<unsigned char> *gpu_table = nullptr) {
    /* checkMode, src/hs.cpp:78-118: the reference's rules and messages first, then ours */
    const unsigned som_modes = HS_MODE_SOM_HORIZON_LARGE | HS_MODE_SOM_HORIZON_MEDIUM | HS_MODE_SOM_HORIZON_SMALL;
    const unsigned scan_modes = mode & (HS_MODE_BLOCK | HS_MODE_STREAM | HS_MODE_VECTORED);
    const char *mode_err = nullptr;
    if (mode & ~(HS_MODE_BLOCK | HS_MODE_STREAM | HS_MODE_VECTORED | som_modes))
        mode_err = "Invalid parameter: unrecognised mode flags.";
    else if (scan_modes == 0 || (scan_modes & (scan_modes - 1)))
        mode_err = "Invalid parameter: mode must have one (and only one) of HS_MODE_BLOCK, HS_MODE_STREAM or "
                   "HS_MODE_VECTORED set.";
    else if ((mode & som_modes) && !(mode & HS_MODE_STREAM))
        mode_err = "Invalid parameter: the HS_MODE_SOM_HORIZON_ mode flags may only be set in streaming mode.";
    else if ((mode & som_modes) & ((mode & som_modes) - 1))
        mode_err = "Invalid parameter: only one HS_MODE_SOM_HORIZON_ mode flag can be set.";
    else if (mode != HS_MODE_BLOCK && mode != HS_MODE_VECTORED)
        mode_err = "Only HS_MODE_BLOCK and HS_MODE_VECTORED are supported by the GPU literal engine.";
    if (mode_err) {
        *error = make_error(mode_err, -1);
        return HS_COMPILER_ERROR;
    }
    void *mem = hook_alloc(g_db, sizeof(hs_database));
    if (hs_error_t ae = check_alloc(mem)) {
        hook_free(g_db, mem);
        *error = make_error(ae == HS_BAD_ALIGN ? "Database allocator returned misaligned memory."
                                               : "Unable to allocate memory.", -1);
        return HS_COMPILER_ERROR;
    }
    hs_database *d = new (mem) hs_database;
    std::vector<std::string> hw_s;
    try {
        for (size_t i = 0; i < exprs.size(); i++) {
            const unsigned f = flags ? flags[i] : 0, id = ids ? ids[i] : 0;
            const size_t first_of_expr = d->pats.size();
            try {
                if (is_lit[i]) {
                    check_flags(f, true);
                    /* src/compiler/compiler.cpp:405-419: flags the pure-literal API refuses */
                    const unsigned bad = HS_FLAG_DOTALL | HS_FLAG_ALLOWEMPTY | HS_FLAG_UTF8 | HS_FLAG_UCP |
                                         HS_FLAG_PREFILTER | HS_FLAG_COMBINATION | HS_FLAG_QUIET | HS_FLAG_MULTILINE;
                    if (f & bad)
                        throw ParseError{"Only HS_FLAG_CASELESS, HS_FLAG_SINGLEMATCH and HS_FLAG_SOM_LEFTMOST are "
                                         "supported in literal API."};
                    if (exprs[i].empty()) throw ParseError{"Pure literal API doesn't support empty string."};
                    Pattern p;
                    p.lit = exprs[i];
                    p.nocase = f & HS_FLAG_CASELESS;
                    p.single = f & HS_FLAG_SINGLEMATCH;
                    p.som = f & HS_FLAG_SOM_LEFTMOST;
                    p.id = id;
                    d->pats.push_back(p);
                } else {
                    for (Pattern &b : parse_pattern(exprs[i], f, id)) d->pats.push_back(std::move(b));
                }
                for (size_t k = first_of_expr; k < d->pats.size(); k++) {
                    if (ext && ext[i]) apply_ext(d->pats[k], *ext[i]);
                    finish_pattern(d->pats[k]);
                }
                if (ext && ext[i] && !is_lit[i])
                    check_ext_widths(&d->pats[first_of_expr], d->pats.size() - first_of_expr, *ext[i]);
            } catch (const ParseError &pe) {
                *error = make_error(pe.msg, (int)i);
                destroy_db(d);
                return HS_COMPILER_ERROR;
            }
        }
        /* one HWLM literal per pattern: the last <= 8 bytes of the literal prefix */
        std::vector<hsgpu_lit_t> lits(d->pats.size());
        hw_s.resize(d->pats.size());
        d->min_width = ~(size_t)0;
        for (size_t i = 0; i < d->pats.size(); i++) {
            const Pattern &p = d->pats[i];
            hw_s[i] = p.lit.size() > 8 ? p.lit.substr(p.lit.size() - 8) : p.lit;
            memset(&lits[i], 0, sizeof(lits[i]));
            lits[i].s = (const uint8_t *)hw_s[i].data();
            lits[i].len = (uint32_t)hw_s[i].size();
            lits[i].id = (uint32_t)i; /* the "Rose program" of this literal = the pattern index */
            lits[i].nocase = p.nocase;
            lits[i].groups = HSGPU_ALL_GROUPS;
            size_t w = p.lit.size();
            for (const Unit &u : p.tail) w += u.optional ? 0 : 1;
            if (p.general) w += p.g.wmin;
            if (p.has_pre) w += p.pre.wmin;
            d->min_width = std::min(d->min_width, w);
        }
        /* a deserialised database brings its GPU table along: no literal compile on load */
        int rv = gpu_table && !gpu_table->empty() ? hsgpu_hwlm_deserialize(gpu_table->data(), gpu_table->size(), &d->hwlm)
                                                  : hsgpu_hwlm_build(lits.data(), lits.size(), 0, &d->hwlm);
        if (rv != HSGPU_SUCCESS) {
            *error = make_error(hsgpu_last_error(), -1);
            destroy_db(d);
            return HS_COMPILER_ERROR;
        }
    } catch (const std::bad_alloc &) {
        destroy_db(d);
        *error = make_error("Unable to allocate memory.", -1);
        return HS_COMPILER_ERROR;
    }
    d->mode = mode;
    d->sources = exprs;
    d->src_is_lit = is_lit;
    for (size_t i = 0; i < exprs.size(); i++) {
        d->src_flags.push_back(flags ? flags[i] : 0);
        d->src_ids.push_back(ids ? ids[i] : 0);
        hs_expr_ext_t none;
        memset(&none, 0, sizeof(none));
        d->src_ext.push_back(ext && ext[i] ? *ext[i] : none);
    }
    for (const Pattern &p : d->pats) /* (an expression may have several branches: walk the branches) */
        if (p.single) d->single_ids.insert(p.id);
    *db = d;
    *error = nullptr;
    return HS_SUCCESS;
}

void destroy_db(hs_database *d) {
    if (!d) return;
    hsgpu_hwlm_free(d->hwlm);
    d->~hs_database();
    hook_free(g_db, d);
}

bool lit_matches_at(const Pattern &p, const unsigned char *buf, size_t end /* offset after the literal */) {
    const size_t n = p.lit.size();
    if (end < n) return false;
    const unsigned char *b = buf + end - n;
    if (!p.nocase) return memcmp(b, p.lit.data(), n) == 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char x = b[i], y = (unsigned char)p.lit[i];
        if (is_alpha(x)) x &= 0xdf;
        if (is_alpha(y)) y &= 0xdf;
        if (x != y) return false;
    }
    return true;
}

struct Event {
    unsigned long long to, from;
    unsigned id;
    bool operator<(const Event &o) const { return to != o.to ? to < o.to : (id != o.id ? id < o.id : from < o.from); }
    bool operator==(const Event &o) const { return to == o.to && id == o.id; }
};

/* turn the HWLM hits of ONE block into the user events it owes, in delivery order:
 * appended to `out` (sorted by to, one per (id, to), SINGLEMATCH ids once) */
void collect_block_events(const hs_database *db, const unsigned char *buf, size_t len, const hsgpu_match_t *recs,
                          size_t n, std::vector<Event> &out) {
    const size_t base = out.size();
    for (size_t k = 0; k < n; k++) {
        const Pattern &p = db->pats[recs[k].id];
        if (p.quiet) continue;
        const size_t lit_end = (size_t)recs[k].end + 1;
        if (!lit_matches_at(p, buf, lit_end)) continue; /* long-literal check */
        unsigned long long start = lit_end - p.lit.size();
        if (!assert_ok(p.as_lit_pre, buf, len, start) || !assert_ok(p.as_lit_post, buf, len, lit_end)) continue;
        if (p.has_pre) { /* the part in front of the literal, backwards; `start` becomes the match start */
            size_t f = 0;
            if (!TailNfa::run_reverse(p, buf, len, start, p.som || (p.ext_flags & HS_EXT_FLAG_MIN_LENGTH), f)) continue;
            start = f;
        } else if ((p.bol && start != 0 && !(p.bol_ml && buf[start - 1] == '\n')) || !assert_ok(p.as_start, buf, len, start)) {
            continue;
        }
        const unsigned long long from = p.som ? start : 0;
        /* hs_expr_ext_t bounds: the job of the reference's CHECK_BOUNDS / CHECK_MIN_LENGTH
         * program instructions (src/rose/program_runtime.c) */
        auto in_bounds = [&](unsigned long long to) {
            /* `$`: at the end of the data or before its final newline; multiline: before any newline */
            if (p.eol && to != len && !(buf[to] == '\n' && (p.eol_ml || (p.eol_nl && to + 1 == len)))) return false;
            if (!assert_ok(p.as_end, buf, len, to)) return false;
            if ((p.ext_flags & HS_EXT_FLAG_MIN_OFFSET) && to < p.min_offset) return false;
            if ((p.ext_flags & HS_EXT_FLAG_MAX_OFFSET) && to > p.max_offset) return false;
            if ((p.ext_flags & HS_EXT_FLAG_MIN_LENGTH) && to - start < p.min_length) return false;
            return true;
        };
        if (p.general) {
            TailNfa::run_general(p.g, buf, len, lit_end, [&](size_t to) {
                if (in_bounds(to)) out.push_back(Event{to, from, p.id});
                return true;
            });
        } else if (p.tail.empty()) {
            if (in_bounds(lit_end)) out.push_back(Event{lit_end, from, p.id});
        } else {
            auto on_to = [&](size_t to) {
                if (in_bounds(to)) out.push_back(Event{to, from, p.id});
                return true;
            };
            TailNfa::run64(p, buf, len, lit_end, on_to);
        }
    }
    std::sort(out.begin() + base, out.end());
    out.erase(std::unique(out.begin() + base, out.end()), out.end()); /* one report per (id, to) */
    if (!db->single_ids.empty()) {
        std::set<unsigned> exhausted; /* SINGLEMATCH ids already reported in this block */
        size_t w = base;
        for (size_t r = base; r < out.size(); r++) {
            if (db->single_ids.count(out[r].id) && !exhausted.insert(out[r].id).second) continue;
            out[w++] = out[r];
        }
        out.resize(w);
    }
}

/* one contiguous slice of the record array, whole blocks only: events of every block in it */
struct BlockRun {
    unsigned long long block;
    size_t ev_begin, ev_end;
};
void collect_slice(const hs_database *db, const unsigned char *data, const unsigned long long *off,
                   const hsgpu_match_t *recs, size_t lo, size_t hi, std::vector<Event> &events,
                   std::vector<BlockRun> &runs) {
    size_t k = lo;
    while (k < hi) { /* records are sorted by (block, end): one run per block */
        const unsigned long long b = recs[k].block;
        size_t e = k;
        while (e < hi && recs[e].block == b) e++;
        const size_t len = (size_t)(off[b + 1] - off[b]);
        const size_t ev0 = events.size();
        if (len >= db->min_width) collect_block_events(db, data + off[b], len, recs + k, e - k, events);
        if (events.size() > ev0) runs.push_back(BlockRun{b, ev0, events.size()});
        k = e;
    }
}

} // namespace

extern "C" {

hs_error_t hs_compile_ext_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                                const hs_expr_ext_t *const *ext, unsigned int elements, unsigned int mode,
                                const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
    if (platform) { /* checkPlatform, src/hs.cpp:122-146 */
        const char *perr = nullptr;
        if (platform->cpu_features & ~(HS_CPU_FEATURES_AVX2 | HS_CPU_FEATURES_AVX512 | HS_CPU_FEATURES_AVX512VBMI))
            perr = "Invalid cpu features specified in the platform information.";
        else if (platform->tune > HS_TUNE_FAMILY_ICX)
            perr = "Invalid tuning value specified in the platform information.";
        if (perr) {
            if (db) *db = nullptr;
            if (error) *error = make_error(perr, -1);
            return HS_COMPILER_ERROR;
        }
    }
    if (!error) {
        if (db) *db = nullptr;
        return HS_COMPILER_ERROR;
    }
    if (!db) { *error = make_error("Invalid parameter: db is NULL", -1); return HS_COMPILER_ERROR; }
    *db = nullptr;
    if (!expressions) { *error = make_error("Invalid parameter: expressions is NULL", -1); return HS_COMPILER_ERROR; }
    if (elements == 0) { *error = make_error("Invalid parameter: elements is zero", -1); return HS_COMPILER_ERROR; }
    std::vector<std::string> ex;
    for (unsigned i = 0; i < elements; i++) {
        if (!expressions[i]) { *error = make_error("Invalid parameter: expression is NULL", (int)i); return HS_COMPILER_ERROR; }
        ex.push_back(expressions[i]);
    }
    return build_database(ex, std::vector<unsigned char>(elements, 0), flags, ids, ext, mode, db, error);
}

hs_error_t hs_compile_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                            unsigned int elements, unsigned int mode, const hs_platform_info_t *platform,
                            hs_database_t **db, hs_compile_error_t **error) {
    return hs_compile_ext_multi(expressions, flags, ids, nullptr, elements, mode, platform, db, error);
}

hs_error_t hs_compile(const char *expression, unsigned int flags, unsigned int mode, const hs_platform_info_t *platform,
                      hs_database_t **db, hs_compile_error_t **error) {
    if (!expression) {
        if (db) *db = nullptr;
        if (error) *error = make_error("Invalid parameter: expression is NULL", -1);
        return HS_COMPILER_ERROR;
    }
    unsigned id = 0;
    return hs_compile_multi(&expression, &flags, &id, 1, mode, platform, db, error);
}

hs_error_t hs_compile_lit_multi(const char *const *expressions, const unsigned *flags, const unsigned *ids,
                                const size_t *lens, unsigned elements, unsigned mode,
                                const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
    if (platform) { /* checkPlatform, src/hs.cpp:122-146 */
        const char *perr = nullptr;
        if (platform->cpu_features & ~(HS_CPU_FEATURES_AVX2 | HS_CPU_FEATURES_AVX512 | HS_CPU_FEATURES_AVX512VBMI))
            perr = "Invalid cpu features specified in the platform information.";
        else if (platform->tune > HS_TUNE_FAMILY_ICX)
            perr = "Invalid tuning value specified in the platform information.";
        if (perr) {
            if (db) *db = nullptr;
            if (error) *error = make_error(perr, -1);
            return HS_COMPILER_ERROR;
        }
    }
    if (!error) {
        if (db) *db = nullptr;
        return HS_COMPILER_ERROR;
    }
    if (!db) { *error = make_error("Invalid parameter: db is NULL", -1); return HS_COMPILER_ERROR; }
    *db = nullptr;
    if (!expressions) { *error = make_error("Invalid parameter: expressions is NULL", -1); return HS_COMPILER_ERROR; }
    if (!lens) { *error = make_error("Invalid parameter: len is NULL", -1); return HS_COMPILER_ERROR; }
    if (elements == 0) { *error = make_error("Invalid parameter: elements is zero", -1); return HS_COMPILER_ERROR; }
    std::vector<std::string> ex;
    for (unsigned i = 0; i < elements; i++) ex.emplace_back(expressions[i] ? expressions[i] : "", expressions[i] ? lens[i] : 0);
    return build_database(ex, std::vector<unsigned char>(elements, 1), flags, ids, nullptr, mode, db, error);
}

hs_error_t hs_compile_lit(const char *expression, unsigned flags, const size_t len, unsigned mode,
                          const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
    if (!expression) {
        if (db) *db = nullptr;
        if (error) *error = make_error("Invalid parameter: expression is NULL", -1);
        return HS_COMPILER_ERROR;
    }
    unsigned id = 0;
    return hs_compile_lit_multi(&expression, &flags, &id, &len, 1, mode, platform, db, error);
}

hs_error_t hs_free_compile_error(hs_compile_error_t *error) {
    if (!error) return HS_SUCCESS;
    hook_free(g_misc, error->message);
    hook_free(g_misc, error);
    return HS_SUCCESS;
}

hs_error_t hs_free_database(hs_database_t *db) {
    if (!db) return HS_SUCCESS;
    if (db->magic != 0x48534744) return HS_INVALID;
    destroy_db(db);
    return HS_SUCCESS;
}

hs_error_t hs_database_size(const hs_database_t *db, size_t *size) {
    if (!db || !size || db->magic != 0x48534744) return HS_INVALID;
    size_t s = sizeof(*db) + hsgpu_hwlm_size(db->hwlm);
    for (const Pattern &p : db->pats) s += sizeof(p) + p.lit.size() + p.tail.size() * sizeof(Unit) + p.reach.size() * 8 + p.g.bytes() + p.pre.bytes();
    *size = s;
    return HS_SUCCESS;
}

hs_error_t hs_database_info(const hs_database_t *db, char **info) {
    if (!db || !info || db->magic != 0x48534744) return HS_INVALID;
    char buf[160];
    snprintf(buf, sizeof(buf), "Version: %s Features: gfx950 Mode: %s", hs_version(),
             db->mode == HS_MODE_VECTORED ? "VECTORED" : "BLOCK");
    *info = misc_strdup(buf);
    if (hs_error_t ae = check_alloc(*info)) {
        hook_free(g_misc, *info);
        *info = nullptr;
        return ae;
    }
    return HS_SUCCESS;
}

/* serialised form: magic "HSGF", CRC-32 of everything after it, count, then per pattern
 * (top bit: vectored mode) {is_lit, flags, id, len, ext flags, min_offset, max_offset, min_length, bytes},
 * then the GPU table section {u64 length, the hsgpu_hwlm_serialize image}. On load the host
 * automata are rebuilt from the sources (microseconds each) and the GPU table is taken from its
 * section as it is. The reference guards its bytecode with a CRC too
 * (src/database.c:119-168): a damaged blob is HS_INVALID, never a different database. */
static const unsigned kSerialMagic = 0x48534747; /* "HSGG": version 2 = sources + GPU table section */
static const unsigned kSerialVectored = 0x80000000u; /* top bit of the count word: HS_MODE_VECTORED */

static unsigned crc32_of(const unsigned char *p, size_t n) {
    static unsigned table[256];
    static bool ready = false;
    if (!ready) {
        for (unsigned i = 0; i < 256; i++) {
            unsigned c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    unsigned c = ~0u;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}

hs_error_t hs_serialize_database(const hs_database_t *db, char **bytes, size_t *length) {
    if (!db || !bytes || !length || db->magic != 0x48534744) return HS_INVALID;
    std::string out;
    auto put32 = [&](unsigned v) { out.append((const char *)&v, 4); };
    auto put64 = [&](unsigned long long v) { out.append((const char *)&v, 8); };
    put32(kSerialMagic);
    put32(0); /* CRC, filled in below */
    put32((unsigned)db->sources.size() | (db->mode == HS_MODE_VECTORED ? kSerialVectored : 0));
    for (size_t i = 0; i < db->sources.size(); i++) {
        put32(db->src_is_lit[i]);
        put32(db->src_flags[i]);
        put32(db->src_ids[i]);
        put32((unsigned)db->sources[i].size());
        put64(db->src_ext[i].flags);
        put64(db->src_ext[i].min_offset);
        put64(db->src_ext[i].max_offset);
        put64(db->src_ext[i].min_length);
        out += db->sources[i];
    }
    /* the GPU literal table, as hsgpu_hwlm_serialize writes it: a deserialised database scans
     * without compiling its literals again */
    size_t tlen = 0;
    if (hsgpu_hwlm_serialize(db->hwlm, nullptr, 0, &tlen) != HSGPU_SUCCESS) return HS_UNKNOWN_ERROR;
    std::string table(tlen, '\0');
    if (hsgpu_hwlm_serialize(db->hwlm, &table[0], table.size(), &tlen) != HSGPU_SUCCESS) return HS_UNKNOWN_ERROR;
    put64(tlen);
    out += table;
    const unsigned crc = crc32_of((const unsigned char *)out.data() + 8, out.size() - 8);
    memcpy(&out[4], &crc, 4);
    *bytes = (char *)hook_alloc(g_misc, out.size());
    if (hs_error_t ae = check_alloc(*bytes)) {
        hook_free(g_misc, *bytes);
        *bytes = nullptr;
        return ae;
    }
    memcpy(*bytes, out.data(), out.size());
    *length = out.size();
    return HS_SUCCESS;
}

namespace {
struct Serial {
    std::vector<std::string> ex;
    std::vector<unsigned char> is_lit;
    std::vector<unsigned> flags, ids;
    std::vector<hs_expr_ext_t> ext;
    unsigned mode = HS_MODE_BLOCK;
    std::vector<unsigned char> table; /* the GPU table section */
};
hs_error_t parse_serial(const char *bytes, size_t length, Serial &out) {
    if (!bytes) return HS_INVALID;
    size_t off = 0;
    auto get32 = [&](unsigned &v) {
        if (off + 4 > length) return false;
        memcpy(&v, bytes + off, 4);
        off += 4;
        return true;
    };
    auto get64 = [&](unsigned long long &v) {
        if (off + 8 > length) return false;
        memcpy(&v, bytes + off, 8);
        off += 8;
        return true;
    };
    unsigned magic, n;
    if (!get32(magic)) return HS_INVALID;
    if ((magic & 0xffffff00u) == (kSerialMagic & 0xffffff00u) && magic != kSerialMagic) return HS_DB_VERSION_ERROR;
    unsigned crc = 0;
    if (magic != kSerialMagic || !get32(crc) || length < 12 ||
        crc != crc32_of((const unsigned char *)bytes + 8, length - 8))
        return HS_INVALID;
    if (!get32(n)) return HS_INVALID;
    if (n & kSerialVectored) out.mode = HS_MODE_VECTORED;
    n &= ~kSerialVectored;
    if (n == 0) return HS_INVALID;
    for (unsigned i = 0; i < n; i++) {
        unsigned l, f, id, len;
        hs_expr_ext_t e;
        memset(&e, 0, sizeof(e));
        if (!get32(l) || !get32(f) || !get32(id) || !get32(len) || !get64(e.flags) || !get64(e.min_offset) ||
            !get64(e.max_offset) || !get64(e.min_length) || off + len > length)
            return HS_INVALID;
        out.ex.emplace_back(bytes + off, len);
        off += len;
        out.is_lit.push_back((unsigned char)l);
        out.flags.push_back(f);
        out.ids.push_back(id);
        out.ext.push_back(e);
    }
    unsigned long long tlen = 0;
    if (!get64(tlen) || tlen > length - off) return HS_INVALID;
    out.table.assign((const unsigned char *)bytes + off, (const unsigned char *)bytes + off + tlen);
    off += tlen;
    return off == length ? HS_SUCCESS : HS_INVALID;
}
} // namespace

hs_error_t hs_deserialize_database(const char *bytes, const size_t length, hs_database_t **db) {
    if (!bytes || !db) return HS_INVALID;
    *db = nullptr;
    Serial sr;
    if (hs_error_t rv = parse_serial(bytes, length, sr)) return rv;
    std::vector<const hs_expr_ext_t *> ext;
    for (const hs_expr_ext_t &e : sr.ext) ext.push_back(e.flags ? &e : nullptr);
    hs_compile_error_t *err = nullptr;
    hs_error_t rv = build_database(sr.ex, sr.is_lit, sr.flags.data(), sr.ids.data(), ext.data(), sr.mode, db, &err, &sr.table);
    hs_free_compile_error(err);
    return rv == HS_SUCCESS ? HS_SUCCESS : HS_INVALID;
}

/* src/hs_common.h:196-271: how much a deserialised database will occupy / what it is,
 * without building it */
hs_error_t hs_serialized_database_size(const char *bytes, const size_t length, size_t *size) {
    if (!size) return HS_INVALID;
    Serial sr;
    if (hs_error_t rv = parse_serial(bytes, length, sr)) return rv;
    hs_database_t *db = nullptr;
    if (hs_error_t rv = hs_deserialize_database(bytes, length, &db)) return rv;
    hs_error_t rv = hs_database_size(db, size);
    hs_free_database(db);
    return rv;
}

hs_error_t hs_serialized_database_info(const char *bytes, size_t length, char **info) {
    if (!info) return HS_INVALID;
    *info = nullptr;
    Serial sr;
    if (hs_error_t rv = parse_serial(bytes, length, sr)) return rv;
    char buf[160];
    snprintf(buf, sizeof(buf), "Version: %s Features: gfx950 Mode: %s", hs_version(),
             sr.mode == HS_MODE_VECTORED ? "VECTORED" : "BLOCK");
    *info = misc_strdup(buf);
    if (hs_error_t ae = check_alloc(*info)) {
        hook_free(g_misc, *info);
        *info = nullptr;
        return ae;
    }
    return HS_SUCCESS;
}

/* ---- allocators (src/hs_common.h:273-439, src/alloc.c) ---- */
hs_error_t hs_set_database_allocator(hs_alloc_t a, hs_free_t f) { g_db.alloc = a; g_db.free = f; return HS_SUCCESS; }
hs_error_t hs_set_misc_allocator(hs_alloc_t a, hs_free_t f) { g_misc.alloc = a; g_misc.free = f; return HS_SUCCESS; }
hs_error_t hs_set_scratch_allocator(hs_alloc_t a, hs_free_t f) { g_scratch.alloc = a; g_scratch.free = f; return HS_SUCCESS; }
hs_error_t hs_set_stream_allocator(hs_alloc_t a, hs_free_t f) { g_stream.alloc = a; g_stream.free = f; return HS_SUCCESS; }
hs_error_t hs_set_allocator(hs_alloc_t a, hs_free_t f) {
    hs_set_database_allocator(a, f);
    hs_set_misc_allocator(a, f);
    hs_set_scratch_allocator(a, f);
    hs_set_stream_allocator(a, f);
    return HS_SUCCESS;
}

hs_error_t hs_populate_platform(hs_platform_info_t *platform) {
    if (!platform) return HS_INVALID;
    memset(platform, 0, sizeof(*platform)); /* tune = HS_TUNE_FAMILY_GENERIC, no CPU features: the engine is a GPU */
    return HS_SUCCESS;
}

/* hs_expression_info / _ext_info (src/hs.cpp:413-516): widths of the supported pattern
 * subset; never unordered, never EOD-anchored */
hs_error_t hs_expression_ext_info(const char *expression, unsigned int flags, const hs_expr_ext_t *ext,
                                  hs_expr_info_t **info, hs_compile_error_t **error) {
    if (!error) return HS_COMPILER_ERROR;
    *error = nullptr;
    if (!info) { *error = make_error("Invalid parameter: info is NULL", -1); return HS_COMPILER_ERROR; }
    *info = nullptr;
    if (!expression) { *error = make_error("Invalid parameter: expression is NULL", -1); return HS_COMPILER_ERROR; }
    std::vector<Pattern> branches;
    try {
        branches = parse_pattern(expression, flags, 0);
        if (ext) {
            for (Pattern &b : branches) apply_ext(b, *ext);
            check_ext_widths(branches.data(), branches.size(), *ext);
        }
    } catch (const ParseError &pe) {
        *error = make_error(pe.msg, 0);
        return HS_COMPILER_ERROR;
    }
    unsigned long long minw = kInf64, maxw = 0;
    bool unbounded = false, any_eol = false, all_eol = true, eol_multiline = false, unordered = false, at_eod = false;
    for (const Pattern &p : branches) {
        unsigned long long lo, hi;
        bool inf;
        raw_widths(p, lo, hi, inf);
        if (p.ext_flags & HS_EXT_FLAG_MIN_LENGTH) lo = std::max(lo, p.min_length);
        if (p.ext_flags & HS_EXT_FLAG_MAX_OFFSET) {
            hi = inf ? p.max_offset : std::min(hi, p.max_offset);
            inf = false;
        }
        minw = std::min(minw, lo);
        maxw = std::max(maxw, hi);
        unbounded |= inf;
        any_eol |= p.eol;
        all_eol &= p.eol;
        eol_multiline |= p.eol && p.eol_ml;
        unordered |= (p.eol && p.eol_nl) || p.as_end;
        at_eod |= p.eol || p.as_end;
    }
    hs_expr_info_t *out = (hs_expr_info_t *)hook_alloc(g_misc, sizeof(*out));
    if (hs_error_t ae = check_alloc(out)) {
        hook_free(g_misc, out);
        *error = make_error(ae == HS_BAD_ALIGN ? "Allocator returned misaligned memory." : "Unable to allocate memory.", -1);
        return HS_COMPILER_ERROR;
    }
    out->min_width = (unsigned)std::min<unsigned long long>(minw, 0xffffffffu);
    out->max_width = unbounded ? 0xffffffffu : (unsigned)std::min<unsigned long long>(maxw, 0xffffffffu);
    /* a branch may be completed by the end of the data (expr_info.cpp:190-206): "foobar$" and \Z
     * are unordered / at EOD / only at EOD (the multiline `$` also matches before inner newlines),
     * \z is ordered and only at EOD, a closing \b is unordered and at EOD but not only there */
    out->unordered_matches = unordered;
    out->matches_at_eod = at_eod;
    out->matches_only_at_eod = all_eol && !eol_multiline;
    *info = out;
    return HS_SUCCESS;
}

hs_error_t hs_expression_info(const char *expression, unsigned int flags, hs_expr_info_t **info,
                              hs_compile_error_t **error) {
    return hs_expression_ext_info(expression, flags, nullptr, info, error);
}

hs_error_t hs_alloc_scratch(const hs_database_t *db, hs_scratch_t **scratch) {
    if (!db || !scratch || db->magic != 0x48534744) return HS_INVALID;
    if (*scratch) return (*scratch)->magic == 0x48534753 ? ((*scratch)->in_use ? HS_SCRATCH_IN_USE : HS_SUCCESS) : HS_INVALID;
    void *mem = hook_alloc(g_scratch, sizeof(hs_scratch));
    if (hs_error_t ae = check_alloc(mem)) {
        hook_free(g_scratch, mem);
        return ae;
    }
    hs_scratch *s = new (mem) hs_scratch;
    int rv = hsgpu_scratch_alloc(&s->gpu, -1);
    if (rv != HSGPU_SUCCESS) {
        s->~hs_scratch();
        hook_free(g_scratch, mem);
        return rv == HSGPU_NOMEM ? HS_NOMEM : HS_UNKNOWN_ERROR;
    }
    *scratch = s;
    return HS_SUCCESS;
}

hs_error_t hs_clone_scratch(const hs_scratch_t *src, hs_scratch_t **dest) {
    if (!src || !dest || src->magic != 0x48534753) return HS_INVALID;
    *dest = nullptr;
    void *mem = hook_alloc(g_scratch, sizeof(hs_scratch));
    if (hs_error_t ae = check_alloc(mem)) {
        hook_free(g_scratch, mem);
        return ae;
    }
    hs_scratch *s = new (mem) hs_scratch;
    if (hsgpu_scratch_alloc(&s->gpu, -1) != HSGPU_SUCCESS) {
        s->~hs_scratch();
        hook_free(g_scratch, mem);
        return HS_UNKNOWN_ERROR;
    }
    *dest = s;
    return HS_SUCCESS;
}

hs_error_t hs_scratch_size(const hs_scratch_t *scratch, size_t *size) {
    if (!scratch || !size || scratch->magic != 0x48534753) return HS_INVALID;
    *size = sizeof(*scratch);
    return HS_SUCCESS;
}

hs_error_t hs_free_scratch(hs_scratch_t *scratch) {
    if (!scratch) return HS_SUCCESS;
    if (scratch->magic != 0x48534753) return HS_INVALID;
    if (scratch->in_use) return HS_SCRATCH_IN_USE;
    hsgpu_scratch_free(scratch->gpu);
    scratch->~hs_scratch();
    hook_free(g_scratch, scratch);
    return HS_SUCCESS;
}

/* The host confirm of a batch ("Rose-lite"): literal hits -> events, delivered in block order
 * on the calling thread. recs: sorted by (block, end), id = pattern index, as the literal
 * engine emits them. Returns true if some callback asked to stop (its block only). */
static bool confirm_and_deliver(const hs_database *db, const char *data, const unsigned long long *off,
                                const hsgpu_match_t *recs, size_t n, hs_batch_event_handler onEvent, void *context) {
    /* host confirm: the events of different blocks are independent, so large batches are cut
     * into slices of whole blocks handled by worker threads; delivery stays on the calling
     * thread, in block order, as the callback contract requires */
    unsigned n_thr = 1;
    if (onEvent && n >= 8192) n_thr = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::vector<Event>> ev(n_thr);
    std::vector<std::vector<BlockRun>> runs(n_thr);
    if (onEvent) {
        std::vector<size_t> cut(n_thr + 1, n);
        cut[0] = 0;
        for (unsigned t = 1; t < n_thr; t++) {
            size_t c = std::max(cut[t - 1], n * t / n_thr);
            while (c < n && c > 0 && recs[c].block == recs[c - 1].block) c++; /* snap to a block boundary */
            cut[t] = c;
        }
        static const bool timing = getenv("HSGPU_FACADE_TIMING") != nullptr;
        auto work = [&](unsigned t) {
            const auto t0 = std::chrono::steady_clock::now();
            collect_slice(db, (const unsigned char *)data, off, recs, cut[t], cut[t + 1], ev[t], runs[t]);
            if (timing)
                fprintf(stderr, "  confirm worker %u: %zu hits in %.2f ms\n", t, cut[t + 1] - cut[t],
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        };
        if (n_thr == 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            try {
                for (unsigned t = 1; t < n_thr; t++) pool.emplace_back(work, t);
            } catch (...) { /* could not start every worker: the caller's thread does the rest */
                for (unsigned t = (unsigned)pool.size() + 1; t < n_thr; t++) work(t);
            }
            work(0);
            for (std::thread &th : pool) th.join();
        }
    }
    bool any_terminated = false;
    for (unsigned t = 0; t < n_thr; t++)
        for (const BlockRun &r : runs[t])
            for (size_t i = r.ev_begin; i < r.ev_end; i++) {
                const Event &e = ev[t][i];
                if (onEvent(r.block, e.id, e.from, e.to, 0, context) != 0) { /* stops THIS block only */
                    any_terminated = true;
                    break;
                }
            }
    return any_terminated;
}

static hs_error_t scan_blocks(const hs_database_t *db, const char *data, const unsigned long long *off,
                              unsigned long long nblocks, hs_scratch_t *scratch, hs_batch_event_handler onEvent,
                              void *context);

hs_error_t hs_scan_batch(const hs_database_t *db, const char *data, const unsigned long long *off,
                         unsigned long long nblocks, unsigned int flags, hs_scratch_t *scratch,
                         hs_batch_event_handler onEvent, void *context) {
    (void)flags;
    if (!scratch || !data || !off) return HS_INVALID;
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_BLOCK) return HS_DB_MODE_ERROR;
    return scan_blocks(db, data, off, nblocks, scratch, onEvent, context);
}

static hs_error_t scan_blocks(const hs_database_t *db, const char *data, const unsigned long long *off,
                              unsigned long long nblocks, hs_scratch_t *scratch, hs_batch_event_handler onEvent,
                              void *context) {
    if (scratch->magic != 0x48534753) return HS_INVALID;
    if (scratch->in_use) return HS_SCRATCH_IN_USE;
    scratch->in_use = true;
    struct Guard { hs_scratch *s; ~Guard() { s->in_use = false; } } guard{scratch};
    if (nblocks == 0) return HS_SUCCESS;
    const auto t_begin = std::chrono::steady_clock::now();
    size_t cap = std::max<size_t>(std::max<size_t>(4096, scratch->recs_cap), (size_t)(off[nblocks] - off[0]) / 1024), n = 0;
    for (int attempt = 0; attempt < 8; attempt++) {
        if (!scratch->reserve(cap)) return HS_NOMEM;
        int rv = hsgpu_hwlm_exec_batch(db->hwlm, scratch->gpu, (const uint8_t *)data, (const uint64_t *)off,
                                       (size_t)nblocks, 0, scratch->recs, cap, &n);
        if (rv == HSGPU_SUCCESS) break;
        if (rv != HSGPU_INSUFFICIENT_SPACE) return rv == HSGPU_NOMEM ? HS_NOMEM : HS_UNKNOWN_ERROR;
        cap = n + n / 4; /* n = the exact total */
        if (attempt == 7) return HS_UNKNOWN_ERROR;
    }
    static const bool timing = getenv("HSGPU_FACADE_TIMING") != nullptr; /* diagnostic: where does a batch go? */
    const auto t_scan = std::chrono::steady_clock::now();
    const bool any_terminated = confirm_and_deliver(db, data, off, scratch->recs, n, onEvent, context);
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "hs_scan_batch: %zu literal hits; GPU literal scan incl. copies %.2f ms, host confirm %.2f ms\n", n,
                std::chrono::duration<double, std::milli>(t_scan - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_scan).count());
    }
    return any_terminated ? HS_SCAN_TERMINATED : HS_SUCCESS;
}

/* the literal the GPU matcher holds for one branch (hs_gpu.h) */
hs_error_t hs_database_literal(const hs_database_t *db, unsigned int index, const char **bytes, size_t *len,
                               int *nocase, unsigned int *id) {
    if (!db || db->magic != 0x48534744 || index >= db->pats.size()) return HS_INVALID;
    const Pattern &p = db->pats[index];
    const size_t n = std::min<size_t>(p.lit.size(), 8);
    if (bytes) *bytes = p.lit.data() + p.lit.size() - n;
    if (len) *len = n;
    if (nocase) *nocase = p.nocase;
    if (id) *id = p.id;
    return HS_SUCCESS;
}

/* Extension: the host confirm alone, for callers that bring their own literal hits (another
 * literal engine, a replayed capture, a test): recs as hsgpu_hwlm_exec_batch would return them
 * for this database's literals -- sorted by (block, end), id = index of the pattern in
 * compile order. No device is touched. */
hs_error_t hs_confirm_batch(const hs_database_t *db, const char *data, const unsigned long long *off,
                            unsigned long long nblocks, const void *records, unsigned long long n_records,
                            hs_batch_event_handler onEvent, void *context) {
    if (!db || db->magic != 0x48534744 || !data || !off || (n_records && !records)) return HS_INVALID;
    const hsgpu_match_t *recs = (const hsgpu_match_t *)records;
    for (unsigned long long i = 0; i < n_records; i++) {
        if (recs[i].block >= nblocks || recs[i].id >= db->pats.size()) return HS_INVALID;
        if (i && (recs[i].block < recs[i - 1].block ||
                  (recs[i].block == recs[i - 1].block && recs[i].end < recs[i - 1].end)))
            return HS_INVALID;
        if (recs[i].end >= off[recs[i].block + 1] - off[recs[i].block]) return HS_INVALID;
    }
    return confirm_and_deliver(db, data, off, recs, (size_t)n_records, onEvent, context) ? HS_SCAN_TERMINATED : HS_SUCCESS;
}

hs_error_t hs_scan(const hs_database_t *db, const char *data, unsigned int length, unsigned int flags,
                   hs_scratch_t *scratch, match_event_handler onEvent, void *context) {
    if (!scratch || !data) return HS_INVALID; /* src/runtime.c:320-322 */
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_BLOCK) return HS_DB_MODE_ERROR; /* src/runtime.c:334-336 */
    struct Ctx { match_event_handler cb; void *user; } c{onEvent, context};
    const unsigned long long off[2] = {0, length};
    if (length < db->min_width) { /* src/runtime.c:346-350 */
        if (scratch->magic != 0x48534753) return HS_INVALID;
        return scratch->in_use ? HS_SCRATCH_IN_USE : HS_SUCCESS;
    }
    return hs_scan_batch(db, data, off, 1, flags, scratch,
                         onEvent ? +[](unsigned long long, unsigned id, unsigned long long from, unsigned long long to,
                                       unsigned fl, void *cc) { return ((Ctx *)cc)->cb(id, from, to, fl, ((Ctx *)cc)->user); }
                                 : (hs_batch_event_handler) nullptr,
                         &c);
}

/* src/runtime.c:1106-1174 hs_scan_vector: the segments are one logical buffer (offsets run
 * through them). Here they are gathered into one block and scanned as such -- on this engine a
 * vectored scan IS a block scan of the concatenation, so the semantics hold by construction. */
hs_error_t hs_scan_vector(const hs_database_t *db, const char *const *data, const unsigned int *length,
                          unsigned int count, unsigned int flags, hs_scratch_t *scratch, match_event_handler onEvent,
                          void *context) {
    (void)flags;
    if (!scratch || !data || !length) return HS_INVALID; /* src/runtime.c:1113-1115 */
    if (!db || db->magic != 0x48534744) return HS_INVALID;
    if (db->mode != HS_MODE_VECTORED) return HS_DB_MODE_ERROR;
    if (scratch->magic != 0x48534753) return HS_INVALID;
    unsigned long long total = 0;
    for (unsigned i = 0; i < count; i++) {
        if (length[i] && !data[i]) return HS_INVALID;
        total += length[i];
    }
    if (total < db->min_width) return scratch->in_use ? HS_SCRATCH_IN_USE : HS_SUCCESS;
    std::string joined;
    joined.reserve((size_t)total);
    for (unsigned i = 0; i < count; i++) joined.append(data[i], length[i]);
    struct Ctx { match_event_handler cb; void *user; } c{onEvent, context};
    const unsigned long long off[2] = {0, total};
    return scan_blocks(db, joined.data(), off, 1, scratch,
                       onEvent ? +[](unsigned long long, unsigned id, unsigned long long from, unsigned long long to,
                                     unsigned fl, void *cc) { return ((Ctx *)cc)->cb(id, from, to, fl, ((Ctx *)cc)->user); }
                               : (hs_batch_event_handler) nullptr,
                       &c);
}

int hs_batch_count_handler(unsigned long long, unsigned int, unsigned long long, unsigned long long, unsigned int,
                           void *context) {
    if (context) ++*(unsigned long long *)context;
    return 0;
}

const char *hs_version(void) { return "5.4.2-hsgpu-gfx950"; }

hs_error_t hs_valid_platform(void) {
    hsgpu_scratch_t *s = nullptr;
    if (hsgpu_scratch_alloc(&s, -1) != HSGPU_SUCCESS) return HS_ARCH_ERROR;
    hsgpu_scratch_free(s);
    return HS_SUCCESS;
}

} // extern "C"
