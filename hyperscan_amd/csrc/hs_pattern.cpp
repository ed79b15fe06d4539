/*
 * hs_pattern.cpp -- pattern compiler of the hs_* facade: an expression becomes branches
 * R1 LIT R2 around a mandatory literal (the subset of src/parser/ + src/nfagraph/ this engine
 * needs; the Ragel parser and the graph compiler proper are out of scope, SURVEY.md section 2
 * rows 11-15).
 *
 *   parse_pattern        top-level alternation, leading (?ims) options, never-matching branches
 *   expand_branch /      literals inside unquantified groups: X(A|B)Y -> XAY|XBY (what Rose gets
 *   distribute_group     by cutting the graph at the alternation)
 *   parse_branch         anchors and edge assertions, the longest top-level literal run,
 *                        R1 / R2 as fragments
 *   TailBuilder          fragment -> Glushkov position automaton, with \b / \B as conditions
 *   finish_auto          -> LimEx shape (shift + exception rows), reversed for R1
 */
#include "../../include/hs_gpu.h"
#include "hs_pattern.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace hsf {

/* HS_FLAG_UTF8 for the expression being compiled (set by parse_pattern): positions are code
 * points. Literal text is its UTF-8 bytes either way; what changes is that `.`, negated classes
 * and \W \D \S take a whole multi-byte code point, that a non-ASCII character is one atom for a
 * quantifier, and that caseless k and s also match U+212A KELVIN SIGN and U+017F LONG S (the
 * reference folds those without HS_FLAG_UCP too: tools/hscollider/test_cases/pcre/utf8.txt). */
thread_local bool g_utf8 = false;
bool utf8_hex_escape(const std::string &p, size_t i, std::string &bytes, size_t &end);

inline size_t utf8_len(unsigned char lead) { return lead < 0x80 ? 1 : lead < 0xc2 ? 0 : lead < 0xe0 ? 2 : lead < 0xf0 ? 3 : lead < 0xf5 ? 4 : 0; }

bool utf8_valid(const std::string &s) {
    for (size_t i = 0; i < s.size();) {
        const size_t n = utf8_len((unsigned char)s[i]);
        if (!n || i + n > s.size()) return false;
        for (size_t k = 1; k < n; k++)
            if (((unsigned char)s[i + k] & 0xc0) != 0x80) return false;
        i += n;
    }
    return true;
}

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
    case 'x': { /* \xH, \xHH (PCRE: zero to two hex digits), \x{H...} with a value that fits a byte */
        auto hex = [](char h) {
            return (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1;
        };
        unsigned v = 0;
        size_t k = i + 1;
        if (k < p.size() && p[k] == '{') {
            size_t j = k + 1, digits = 0;
            for (; j < p.size() && hex(p[j]) >= 0; j++, digits++) {
                v = v * 16 + (unsigned)hex(p[j]);
                if (v > 0xff) return false; /* beyond one byte: a UTF-8 mode matter */
            }
            if (j >= p.size() || p[j] != '}' || digits == 0) return false;
            out = (unsigned char)v;
            i = j + 1;
            return true;
        }
        for (int d = 0; d < 2 && k < p.size() && hex(p[k]) >= 0; d++, k++) v = v * 16 + (unsigned)hex(p[k]);
        out = (unsigned char)v;
        i = k;
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
/* "[:digit:]", "[.ch.]", "[=ch=]" where a class would start: the reference refuses them by name
 * (src/parser/Parser.rl:1280-1290: '[' d ( '\\]' | [^\]] )* d ']' for d in ": . =") */
void refuse_posix_outside_class(const std::string &p, size_t i) {
    if (i + 1 >= p.size() || !strchr(":.=", p[i + 1])) return;
    const char d = p[i + 1];
    for (size_t k = i + 2; k < p.size(); k++) {
        if (p[k] == d && k + 1 < p.size() && p[k + 1] == ']')
            throw ParseError{d == ':' ? "POSIX named classes are only supported inside a class." : "Unsupported POSIX collating element."};
        if (p[k] == ']' && p[k - 1] != '\\') return;
    }
}

ByteSet parse_bracket_class(const std::string &p, size_t &i, bool nocase = false) {
    refuse_posix_outside_class(p, i);
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
                std::string useq;
                if (utf8_hex_escape(p, j, useq, k)) lit = 0x80; /* (UTF-8 mode only skips with this function) */
                else if (!class_char_escape(p, k, lit)) throw ParseError{"Unsupported escape sequence."};
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
                std::string useq;
                if (k < p.size() && utf8_hex_escape(p, j, useq, k)) lit = 0xff;
                else if (k >= p.size() || !class_char_escape(p, k, lit)) throw ParseError{"Unsupported escape sequence."};
                hi = lit;
                j = k;
            } else {
                hi = (unsigned char)p[j++];
            }
            if (hi < lo && !g_utf8) throw ParseError{"Range out of order in character class."};
        }
        add_range(cls, lo, hi);
    }
    i = j + 1;
    if (nocase) cls = fold_case(cls); /* before the negation: caseless [^a] excludes both a and A */
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

/* UTF-8 mode: "\xHH" / "\x{H...}" at p[i] (the backslash) naming a code point above 0x7f ->
 * its UTF-8 bytes; false for anything else (the byte-valued paths handle those) */
bool utf8_hex_escape(const std::string &p, size_t i, std::string &bytes, size_t &end) {
    if (!g_utf8 || i + 1 >= p.size() || p[i] != '\\' || p[i + 1] != 'x') return false;
    auto hex = [](char h) {
        return (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1;
    };
    unsigned long cp = 0;
    size_t k = i + 2;
    if (k < p.size() && p[k] == '{') {
        size_t j = k + 1, digits = 0;
        for (; j < p.size() && hex(p[j]) >= 0; j++, digits++) {
            cp = cp * 16 + (unsigned)hex(p[j]);
            if (cp > 0x10ffff) throw ParseError{"Code point beyond U+10FFFF."};
        }
        if (j >= p.size() || p[j] != '}' || digits == 0) return false;
        k = j + 1;
    } else {
        for (int d = 0; d < 2 && k < p.size() && hex(p[k]) >= 0; d++, k++) cp = cp * 16 + (unsigned)hex(p[k]);
    }
    if (cp < 0x80) return false;
    if (cp >= 0xd800 && cp <= 0xdfff) throw ParseError{"Surrogate code points are not characters."};
    bytes.clear();
    if (cp < 0x800) {
        bytes.push_back((char)(0xc0 | cp >> 6));
    } else if (cp < 0x10000) {
        bytes.push_back((char)(0xe0 | cp >> 12));
        bytes.push_back((char)(0x80 | (cp >> 6 & 0x3f)));
    } else {
        bytes.push_back((char)(0xf0 | cp >> 18));
        bytes.push_back((char)(0x80 | (cp >> 12 & 0x3f)));
        bytes.push_back((char)(0x80 | (cp >> 6 & 0x3f)));
    }
    bytes.push_back((char)(0x80 | (cp & 0x3f)));
    end = k;
    return true;
}

/* "(?i)" "(?-s)" "(?is-m:" ... at p[i] == '(': which flags it switches and whether it opens a
 * scoped group. Only i and s may change inside a pattern here (m would move the meaning of the
 * edge anchors, x is not supported). Returns 0 = not an option group, 1 = "(?..)" setting, 2 =
 * "(?..:" scoped group; `end` = index after the ")" or ":". */
int option_group_at(const std::string &p, size_t i, bool &nocase, bool &dotall, size_t &end) {
    if (i + 2 >= p.size() || p[i] != '(' || p[i + 1] != '?') return 0;
    size_t k = i + 2;
    bool on = true, any = false, nc = nocase, ds = dotall;
    for (; k < p.size(); k++) {
        const char c = p[k];
        if (c == '-') { if (!on) return 0; on = false; continue; }
        if (c == 'i') nc = on;
        else if (c == 's') ds = on;
        else if (c == 'm' || c == 'x') throw ParseError{"Only the i and s options may change inside a pattern."};
        else break;
        any = true;
    }
    if (!any || k >= p.size() || (p[k] != ')' && p[k] != ':')) return 0;
    nocase = nc;
    dotall = ds;
    end = k + 1;
    return p[k] == ')' ? 1 : 2;
}

/* ---- UTF-8 mode: a bracket class as a set of code points ------------------------------ */
typedef std::vector<std::pair<unsigned, unsigned>> CpRanges;

unsigned utf8_decode(const std::string &p, size_t &i) {
    const unsigned char c = (unsigned char)p[i];
    const size_t n = utf8_len(c);
    if (!n || i + n > p.size()) throw ParseError{"Expression is not valid UTF-8."};
    unsigned cp = n == 1 ? c : c & (0xff >> (n + 1));
    for (size_t k = 1; k < n; k++) cp = cp << 6 | ((unsigned char)p[i + k] & 0x3f);
    i += n;
    return cp;
}

void cp_normalise(CpRanges &r) {
    std::sort(r.begin(), r.end());
    CpRanges out;
    for (const auto &x : r) {
        if (!out.empty() && x.first <= out.back().second + 1) out.back().second = std::max(out.back().second, x.second);
        else out.push_back(x);
    }
    r.swap(out);
}

/* "[...]" at p[i] in UTF-8 mode -> sorted disjoint code-point ranges (surrogates never included) */
CpRanges parse_bracket_class_cp(const std::string &p, size_t &i, bool nocase) {
    refuse_posix_outside_class(p, i);
    CpRanges r;
    size_t j = i + 1;
    bool neg = false;
    if (j < p.size() && p[j] == '^') {
        neg = true;
        j++;
    }
    auto add_byteset = [&](const ByteSet &item) { /* an ASCII-defined class; its complement forms reach every code point above */
        size_t high = 0;
        for (unsigned b = 0x80; b < 0x100; b++) high += item[b];
        if (high != 0 && high != 128) throw ParseError{"\\h \\v and their complements are not supported inside a UTF-8 class."};
        for (unsigned b = 0; b < 0x80; b++)
            if (item[b]) r.push_back({b, b});
        if (high == 128) r.push_back({0x80, 0x10ffff});
    };
    auto one = [&](size_t &k) -> unsigned { /* one member character at p[k] */
        if (p[k] == '\\') {
            std::string useq;
            size_t e = k;
            if (utf8_hex_escape(p, k, useq, e)) {
                size_t z = 0;
                k = e;
                return utf8_decode(useq, z);
            }
            size_t q = k + 1;
            unsigned char lit;
            if (q >= p.size() || !class_char_escape(p, q, lit)) throw ParseError{"Unsupported escape sequence."};
            k = q;
            return lit;
        }
        return utf8_decode(p, k);
    };
    bool first = true;
    for (;;) {
        if (j >= p.size()) throw ParseError{"Unterminated character class."};
        if (p[j] == ']' && !first) break;
        first = false;
        if (p[j] == '[' && j + 1 < p.size() && (p[j + 1] == '.' || p[j + 1] == '=') &&
            p.find(std::string(1, p[j + 1]) + "]", j + 2) != std::string::npos)
            throw ParseError{"Unsupported POSIX collating element."};
        if (p[j] == '[' && j + 1 < p.size() && p[j + 1] == ':') {
            const size_t e = p.find(":]", j + 2);
            if (e == std::string::npos) throw ParseError{"Unterminated POSIX class."};
            std::string name = p.substr(j + 2, e - j - 2);
            const bool inv = !name.empty() && name[0] == '^';
            if (inv) name.erase(0, 1);
            ByteSet item;
            if (!posix_class(name, item)) throw ParseError{"Unknown POSIX character class."};
            add_byteset(inv ? ~item : item);
            j = e + 2;
            continue;
        }
        if (p[j] == '\\' && j + 1 < p.size()) {
            bool ok;
            const ByteSet item = class_escape(p[j + 1], ok);
            if (ok) {
                add_byteset(item);
                j += 2;
                continue;
            }
        }
        const unsigned lo = one(j);
        unsigned hi = lo;
        if (j + 1 < p.size() && p[j] == '-' && p[j + 1] != ']') {
            j++;
            if (p[j] == '[' && j + 1 < p.size() && p[j + 1] == ':') throw ParseError{"Invalid range in character class."};
            hi = one(j);
            if (hi < lo) throw ParseError{"Range out of order in character class."};
        }
        r.push_back({lo, hi});
    }
    i = j + 1;
    if (nocase) { /* ASCII letters, and the two non-ASCII partners the reference folds without UCP */
        CpRanges extra;
        for (const auto &x : r)
            for (unsigned c = std::max(x.first, 0x41u); c <= std::min(x.second, 0x7au); c++)
                if (is_alpha((unsigned char)c)) extra.push_back({c ^ 0x20, c ^ 0x20});
        auto has = [&](unsigned c) { for (const auto &x : r) if (x.first <= c && c <= x.second) return true; return false; };
        if (has('k') || has('K')) extra.push_back({0x212a, 0x212a});
        if (has('s') || has('S')) extra.push_back({0x17f, 0x17f});
        if (has(0x212a)) { extra.push_back({'k', 'k'}); extra.push_back({'K', 'K'}); }
        if (has(0x17f)) { extra.push_back({'s', 's'}); extra.push_back({'S', 'S'}); }
        r.insert(r.end(), extra.begin(), extra.end());
    }
    cp_normalise(r);
    if (neg) {
        CpRanges c;
        unsigned next = 0;
        for (const auto &x : r) {
            if (x.first > next) c.push_back({next, x.first - 1});
            next = x.second + 1;
        }
        if (next <= 0x10ffff) c.push_back({next, 0x10ffff});
        r.swap(c);
    }
    /* cut the surrogate block out */
    CpRanges out;
    for (const auto &x : r) {
        if (x.second < 0xd800 || x.first > 0xdfff) { out.push_back(x); continue; }
        if (x.first < 0xd800) out.push_back({x.first, 0xd7ff});
        if (x.second > 0xdfff) out.push_back({0xe000, x.second});
    }
    return out;
}

/* [lo, hi] -> sequences of byte ranges whose concatenations are exactly the UTF-8 encodings of
 * the code points in it (the classic split: by encoded length, then until every continuation
 * level is a full rectangle) */
void utf8_split(unsigned lo, unsigned hi, std::vector<std::vector<std::pair<unsigned, unsigned>>> &out) {
    if (lo > hi) return;
    for (unsigned b : {0x7fu, 0x7ffu, 0xffffu})
        if (lo <= b && b < hi) {
            utf8_split(lo, b, out);
            utf8_split(b + 1, hi, out);
            return;
        }
    if (hi < 0x80) {
        out.push_back({{lo, hi}});
        return;
    }
    const int n = lo < 0x800 ? 2 : lo < 0x10000 ? 3 : 4;
    for (int k = 1; k < n; k++) {
        const unsigned m = (1u << (6 * k)) - 1;
        if ((lo & ~m) != (hi & ~m)) {
            if (lo & m) {
                utf8_split(lo, lo | m, out);
                utf8_split((lo | m) + 1, hi, out);
                return;
            }
            if ((hi & m) != m) {
                utf8_split(lo, (hi & ~m) - 1, out);
                utf8_split(hi & ~m, hi, out);
                return;
            }
        }
    }
    auto enc = [&](unsigned cp, unsigned char *b) {
        if (n == 2) { b[0] = 0xc0 | cp >> 6; b[1] = 0x80 | (cp & 0x3f); }
        else if (n == 3) { b[0] = 0xe0 | cp >> 12; b[1] = 0x80 | (cp >> 6 & 0x3f); b[2] = 0x80 | (cp & 0x3f); }
        else { b[0] = 0xf0 | cp >> 18; b[1] = 0x80 | (cp >> 12 & 0x3f); b[2] = 0x80 | (cp >> 6 & 0x3f); b[3] = 0x80 | (cp & 0x3f); }
    };
    unsigned char a[4], z[4];
    enc(lo, a);
    enc(hi, z);
    std::vector<std::pair<unsigned, unsigned>> seq;
    for (int k = 0; k < n; k++) seq.push_back({a[k], z[k]});
    out.push_back(seq);
}

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
            { /* option settings: "(?i)" lasts to the end of the enclosing group, "(?i:...)" is its own group */
                bool nc = nocase, ds = dotall;
                size_t e = i;
                const int kind = option_group_at(p, i, nc, ds, e);
                if (kind == 1) {
                    nocase = nc;
                    dotall = ds;
                    i = e;
                    return Frag();
                }
                if (kind == 2) {
                    const bool nc0 = nocase, ds0 = dotall;
                    nocase = nc;
                    dotall = ds;
                    i = e;
                    Frag f = parse_alt(i, depth + 1);
                    if (i >= p.size() || p[i] != ')') throw ParseError{"Missing closing parenthesis."};
                    i++;
                    nocase = nc0;
                    dotall = ds0;
                    return f;
                }
            }
            const bool nc_outer = nocase, ds_outer = dotall; /* settings made inside end with the group */
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
            nocase = nc_outer;
            dotall = ds_outer;
            return f;
        }
        if (c == '\\' && i + 1 < p.size() && (p[i + 1] == 'b' || p[i + 1] == 'B')) {
            Frag f; /* zero width: the empty string, under a condition */
            f.nullmask = p[i + 1] == 'b' ? 2 : 4;
            has_cond = true;
            i += 2;
            return f;
        }
        {
            std::string seq; /* a non-ASCII character, written raw or as \x..: one atom made of its bytes */
            size_t end = i;
            if (g_utf8 && c >= 0x80) {
                const size_t n = utf8_len(c);
                if (!n || i + n > p.size()) throw ParseError{"Expression is not valid UTF-8."};
                seq = p.substr(i, n);
                end = i + n;
            }
            if (!seq.empty() || utf8_hex_escape(p, i, seq, end)) {
                if (nocase) throw ParseError{"Caseless non-ASCII characters need HS_FLAG_UCP tables: not supported."};
                Frag f;
                for (char b : seq) {
                    ByteSet one;
                    one.set((unsigned char)b);
                    f = cat(f, leaf(one));
                }
                i = end;
                return f;
            }
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
                if (g_utf8 && lit >= 0x80) throw ParseError{"Escapes above \\x7f are code points in UTF-8 mode: not supported."};
                set.set(lit);
                i = j;
            }
        } else if (c == '.') {
            set.set();
            if (!dotall) set.reset('\n');
            i++;
        } else if (c == '[' && g_utf8) { /* a set of code points: one branch per UTF-8 byte-range sequence */
            const CpRanges r = parse_bracket_class_cp(p, i, nocase);
            std::vector<std::vector<std::pair<unsigned, unsigned>>> seqs;
            for (const auto &x : r) utf8_split(x.first, x.second, seqs);
            if (seqs.empty()) throw NeverMatch();
            ByteSet ascii; /* all one-byte sequences share a position */
            Frag f;
            bool have = false;
            auto add = [&](const Frag &x) {
                if (!have) { f = x; have = true; return; }
                for (int k = 0; k < 3; k++) {
                    bits_or(f.first[k], x.first[k]);
                    bits_or(f.last[k], x.last[k]);
                }
                f.wmin = std::min(f.wmin, x.wmin);
                f.wmax = std::max(f.wmax, x.wmax);
            };
            for (const auto &sq : seqs) {
                if (sq.size() == 1) { add_range(ascii, sq[0].first, sq[0].second); continue; }
                Frag x;
                for (const auto &br : sq) {
                    ByteSet one;
                    add_range(one, br.first, br.second);
                    x = cat(x, leaf(one, false));
                }
                add(x);
            }
            if (ascii.any()) add(leaf(ascii, false));
            return f;
        } else if (c == '[') {
            set = parse_bracket_class(p, i, nocase);
        } else if (strchr(")|^$*+?", c) || (c == '{' && is_repeat_at(p, i))) {
            throw ParseError{std::string("Unsupported regex construct '") + (char)c + "'."};
        } else {
            set.set(c);
            i++;
        }
        if (!g_utf8) return leaf(set);
        /* UTF-8: the ASCII members as one position; "everything else" as the multi-byte code
         * points; caseless k / s bring their non-ASCII partners */
        size_t high = 0;
        for (unsigned b = 0x80; b < 0x100; b++) high += set[b];
        if (high != 0 && high != 128) throw ParseError{"Classes with some non-ASCII members are not supported in UTF-8 mode."};
        ByteSet ascii = nocase ? fold_case(set) : set;
        for (unsigned b = 0x80; b < 0x100; b++) ascii.reset(b);
        Frag f;
        bool have = false;
        auto add = [&](const Frag &x) {
            if (!have) { f = x; have = true; return; }
            for (int k = 0; k < 3; k++) {
                bits_or(f.first[k], x.first[k]);
                bits_or(f.last[k], x.last[k]);
            }
            f.wmin = std::min(f.wmin, x.wmin);
            f.wmax = std::max(f.wmax, x.wmax);
        };
        auto seq = [&](std::initializer_list<std::pair<unsigned, unsigned>> ranges) {
            Frag x;
            for (const auto &r : ranges) {
                ByteSet one;
                add_range(one, r.first, r.second);
                x = cat(x, leaf(one, false));
            }
            return x;
        };
        if (ascii.any()) add(leaf(ascii, false));
        if (high == 128) { /* any multi-byte code point (input is valid UTF-8 by contract) */
            add(seq({{0xc2, 0xdf}, {0x80, 0xbf}}));
            add(seq({{0xe0, 0xef}, {0x80, 0xbf}, {0x80, 0xbf}}));
            add(seq({{0xf0, 0xf4}, {0x80, 0xbf}, {0x80, 0xbf}, {0x80, 0xbf}}));
        } else if (nocase) {
            if (ascii['k']) add(seq({{0xe2, 0xe2}, {0x84, 0x84}, {0xaa, 0xaa}})); /* U+212A */
            if (ascii['s']) add(seq({{0xc5, 0xc5}, {0xbf, 0xbf}}));               /* U+017F */
        }
        if (!have) throw NeverMatch();
        return f;
    }
    /* one position */
    Frag leaf(const ByteSet &set, bool fold = true) {
        const bool keep = nocase;
        if (!fold) nocase = false;
        const unsigned pos = new_pos(set);
        nocase = keep;
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
    bool nocase = false, dotall = false; /* the option state where the run sits ("ab(?i)cd": cd is caseless) */
};

LitRun longest_literal_run(const std::string &p, bool nocase, bool dotall) {
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
        std::string useq;
        if (utf8_hex_escape(p, i, useq, j)) { /* \x.. naming a non-ASCII character: its bytes */
            is_lit = !nocase;
        } else if (c == '\\') {
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
        } else if (c == '(' && [&] { /* "(?i)" at the top level: the state changes for the rest of the branch */
                       bool nc = nocase, ds = dotall;
                       size_t e = i;
                       if (option_group_at(p, i, nc, ds, e) != 1) return false;
                       nocase = nc;
                       dotall = ds;
                       j = e;
                       return true;
                   }()) {
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
        } else if (g_utf8 && c >= 0x80) { /* a non-ASCII character: all its bytes, or none */
            const size_t n = utf8_len(c);
            if (!n || i + n > p.size()) throw ParseError{"Expression is not valid UTF-8."};
            j = i + n;
            is_lit = !nocase;
        } else {
            lit = c;
            j = i + 1;
            is_lit = true;
        }
        /* caseless k / s have non-ASCII partners in UTF-8 mode: a class, not a literal byte */
        if (is_lit && g_utf8 && nocase && j == i + 1 && strchr("kKsS", (char)lit)) is_lit = false;
        const size_t after = j;
        skip_quant(j);
        if (is_lit && j == after) {
            if (cur.bytes.empty()) {
                cur.begin = i;
                cur.nocase = nocase;
                cur.dotall = dotall;
            }
            if (!useq.empty()) cur.bytes += useq;
            else if (g_utf8 && c >= 0x80) cur.bytes.append(p, i, after - i);
            else cur.bytes.push_back((char)lit);
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
    const bool base_nocase = flags & HS_FLAG_CASELESS, base_dotall = flags & HS_FLAG_DOTALL;
    pat.single = flags & HS_FLAG_SINGLEMATCH;
    pat.som = flags & HS_FLAG_SOM_LEFTMOST;
    pat.id = id;
    /* `{` opens a repeat only when a well-formed {m}, {m,} or {m,n} follows; otherwise it (and
     * a lone `}`) is an ordinary character, as in PCRE ("foo.{,10}bar" is twelve literal-ish
     * positions: unit/hyperscan/expr_info.cpp:211) */
    auto is_repeat = [&](size_t k) { return TailBuilder::is_repeat_at(p, k); };
    /* the branch is R1 LIT R2 around its longest top-level literal run (the front one on a tie);
     * \b / \B may hug the literal on either side */
    LitRun run = longest_literal_run(p, base_nocase, base_dotall);
    if (run.bytes.empty()) throw NoLiteral();
    /* i / s settings in front of the literal ("ab(?i)cdef") reach it and what follows it */
    pat.nocase = run.nocase;
    const bool dotall = run.dotall;
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
        pat.pre = compile_auto(p.substr(0, r1_end), base_nocase, base_dotall, true);
        pat.g = compile_auto(p.substr(r2_begin), pat.nocase, dotall);
        pat.has_pre = pat.pre.npos != 0 || pat.pre.has_cond;
        pat.general = pat.g.npos != 0 || pat.g.has_cond;
        pat.tail_nullable = pat.g.nullable || pat.g.cnullable[0] || pat.g.cnullable[1];
        return pat;
    }
    /* a tail with a group in it goes to the position automaton; the linear form below stays the
     * path for everything it can express */
    bool grouped = g_utf8 && i < p.size(); /* (UTF-8 mode lives in the builder only) */
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
            cls = parse_bracket_class(p, i, pat.nocase);
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
bool rewrite_for_literal(const std::string &b, unsigned flags, unsigned id, std::vector<Pattern> &out, int depth);

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
    const size_t before = out.size();
    try {
        distribute_group(b, flags, id, out, depth);
    } catch (const NeverMatch &) {
        throw;
    } catch (const NoLiteral &) {
        out.erase(out.begin() + before, out.end());
        if (!rewrite_for_literal(b, flags, id, out, depth)) throw;
    }
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
        for (const std::string &a : alts) { /* "(*VERB)", "(+x)": not ours to rearrange */
            if (!a.empty() && (strchr("*+?", a[0]) || TailBuilder::is_repeat_at(a, 0))) throw NoLiteral();
            for (size_t k = 0; k + 2 < a.size(); k++) { /* an option set inside would outlive its group */
                bool nc = false, ds = false;
                size_t e = 0;
                if (a[k] == '\\') { k++; continue; }
                if (a[k] == '(' && option_group_at(a, k, nc, ds, e) == 1) throw NoLiteral();
            }
        }
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

/* Two more identities reach a literal where neither the branch nor a group offers one (all end
 * offsets are reported and nothing is captured, so they hold exactly):
 *   - a repeat that must run at least once gives up its first turn:  A+ = A A*,
 *     A{n,m} = A A{n-1,m-1}  ("(foobar)+", "((foo){2}){3}", "[a-z]{3,7}");
 *   - a small class standing alone is the alternation of its members:  [pqr] = (?:p|q|r)
 *     ("[pqr]", "\\s"; one-byte literals, so every member byte in the data becomes a candidate
 *     for the host automaton: correct, and as slow as that sounds on text full of them).
 * Each candidate rewrite goes back through expand_branch; the first that compiles is kept.
 * The reference has no such limit (unit/hyperscan/single.cpp:320-345 lists these forms): its
 * Rose build falls back to NFA/DFA engines over the whole buffer where there is no literal. */
constexpr size_t kMaxClassMembers = 40;
/* rewrites tried per expression: a pattern with many literal-less repeats would otherwise try
 * every order of unrolling them (reset by parse_pattern) */
constexpr int kRewriteBudget = 48;
thread_local int g_rewrites_left = kRewriteBudget;

bool rewrite_for_literal(const std::string &b, unsigned flags, unsigned id, std::vector<Pattern> &out, int depth) {
    if (depth >= 8) return false;
    struct Atom {
        size_t begin, end, qend; /* [begin,end) the atom, [end,qend) its quantifier */
        bool is_class, quantified;
        unsigned long long lo, hi; /* hi == ~0ull: unbounded */
        ByteSet set;
    };
    const bool has_options = b.find("(?") != std::string::npos && [&] { /* "(?:" alone is harmless */
        for (size_t k = b.find("(?"); k != std::string::npos; k = b.find("(?", k + 1))
            if (k + 2 >= b.size() || b[k + 2] != ':') return true;
        return false;
    }();
    const bool nocase = flags & HS_FLAG_CASELESS;
    std::vector<Atom> atoms;
    size_t i = 0;
    while (i < b.size()) {
        Atom a{i, i, i, false, false, 1, 1, ByteSet()};
        const unsigned char c = (unsigned char)b[i];
        size_t j = i;
        bool candidate = false;
        std::string useq;
        if (utf8_hex_escape(b, i, useq, j)) {
        } else if (c == '\\') {
            if (i + 1 >= b.size()) return false;
            bool ok;
            const ByteSet cs = class_escape(b[i + 1], ok);
            if (ok) {
                j = i + 2;
                a.is_class = candidate = !g_utf8 && !has_options;
                a.set = nocase ? fold_case(cs) : cs;
            } else if (strchr("bBAzZ", b[i + 1])) {
                j = i + 2;
            } else {
                unsigned char lit;
                j = i + 1;
                if (!char_escape(b, j, lit)) return false;
            }
        } else if (c == '[') {
            a.set = parse_bracket_class(b, j, nocase);
            a.is_class = candidate = !g_utf8 && !has_options;
        } else if (c == '(') {
            int d = 0;
            for (;; j++) {
                if (j >= b.size()) return false;
                if (b[j] == '\\') { j++; continue; }
                if (b[j] == '[') { size_t e = j; parse_bracket_class(b, e); j = e - 1; continue; }
                if (b[j] == '(') d++;
                if (b[j] == ')' && --d == 0) break;
            }
            j++;
            candidate = !(i + 1 < b.size() && b[i + 1] == '?') || (i + 2 < b.size() && b[i + 2] == ':');
            if (i + 1 < b.size() && (b[i + 1] == '*' || b[i + 1] == '+')) candidate = false; /* "(*VERB)" */
        } else {
            j = i + 1;
        }
        a.end = j;
        if (j < b.size() && (b[j] == '?' || b[j] == '*' || b[j] == '+')) {
            a.quantified = true;
            a.lo = b[j] == '+' ? 1 : 0;
            a.hi = b[j] == '?' ? 1 : ~0ull;
            j++;
        } else if (TailBuilder::is_repeat_at(b, j)) {
            a.quantified = true;
            const size_t close = b.find('}', j), comma = b.find(',', j);
            a.lo = strtoull(b.c_str() + j + 1, nullptr, 10);
            a.hi = comma == std::string::npos || comma > close ? a.lo : comma + 1 == close ? ~0ull : strtoull(b.c_str() + comma + 1, nullptr, 10);
            j = close + 1;
        }
        if (a.quantified && j < b.size() && b[j] == '+') candidate = false; /* possessive: not ours to rearrange */
        if (a.quantified && j < b.size() && (b[j] == '?' || b[j] == '+')) j++;
        a.qend = j;
        if (a.quantified && (a.lo == 0 || a.lo > 32767 || (a.hi != ~0ull && a.hi < a.lo))) candidate = false;
        if (a.is_class && nocase) /* caseless: one literal serves both letters of a pair */
            for (unsigned v = 'a'; v <= 'z'; v++)
                if (a.set[v] && a.set[v - 32]) a.set.reset(v);
        if (a.is_class && (a.set.none() || a.set.count() > kMaxClassMembers)) candidate = false;
        if (candidate && (a.quantified || a.is_class)) atoms.push_back(a);
        i = j;
    }
    /* groups before classes (longer literals), small classes before large ones */
    std::stable_sort(atoms.begin(), atoms.end(), [](const Atom &x, const Atom &y) {
        if (x.is_class != y.is_class) return !x.is_class;
        return x.is_class && x.set.count() < y.set.count();
    });
    for (const Atom &a : atoms) {
        const std::string atom = b.substr(a.begin, a.end - a.begin);
        std::string first = atom, rest;
        if (a.is_class) { /* the alternation of its members */
            first = "(?:";
            bool any = false;
            for (unsigned v = 0; v < 256; v++) {
                if (!a.set[v]) continue;
                char hex[8];
                snprintf(hex, sizeof hex, "%s\\x%02x", any ? "|" : "", v);
                first += hex;
                any = true;
            }
            first += ")";
        }
        if (a.quantified) {
            const unsigned long long lo = a.lo - 1, hi = a.hi == ~0ull ? a.hi : a.hi - 1;
            if (hi == ~0ull) rest = atom + (lo == 0 ? std::string("*") : "{" + std::to_string(lo) + ",}");
            else if (hi > 0) rest = atom + "{" + std::to_string(lo) + "," + std::to_string(hi) + "}";
        }
        const size_t before = out.size();
        if (g_rewrites_left <= 0) return false;
        g_rewrites_left--;
        try {
            expand_branch(b.substr(0, a.begin) + first + rest + b.substr(a.qend), flags, id, out, depth + 1);
            return true;
        } catch (const NeverMatch &) {
            throw;
        } catch (const ParseError &) {
            out.erase(out.begin() + before, out.end());
        }
    }
    return false;
}

/* expression := branch ('|' branch)* at the top level: every branch is its own literal-prefixed
 * pattern reporting the same id (the reference builds one graph; the reports are the same) */
bool parse_class_seq(const std::string &expr, unsigned flags, ClassSeq &out) {
    const unsigned ok_flags = HS_FLAG_CASELESS | HS_FLAG_DOTALL | HS_FLAG_MULTILINE | HS_FLAG_SINGLEMATCH | HS_FLAG_PREFILTER |
                              HS_FLAG_QUIET | HS_FLAG_ALLOWEMPTY;
    if (flags & ~ok_flags) return false;
    const bool nocase = flags & HS_FLAG_CASELESS;
    size_t i = 0;
    ByteSet cls[2];
    unsigned rep[2] = {0, 0};
    try {
        for (int t = 0; t < 2; t++) {
            if (i >= expr.size()) return false;
            if (expr[i] == '[') {
                if (expr.find("[:", i) != std::string::npos) return false; /* POSIX names: the general parser's business */
                cls[t] = parse_bracket_class(expr, i, nocase); /* leaves i behind the closing bracket */
            } else if (expr[i] == '\\' && i + 1 < expr.size()) {
                bool ok = false;
                cls[t] = class_escape(expr[i + 1], ok);
                if (!ok) return false;
                i += 2;
            } else if (expr[i] == '.') {
                cls[t].set();
                if (!(flags & HS_FLAG_DOTALL)) cls[t].reset('\n');
                i++;
            } else {
                return false;
            }
            if (nocase) cls[t] = fold_case(cls[t]);
            if (i < expr.size() && expr[i] == '+') {
                rep[t] = 1;
                i++;
            } else if (i < expr.size() && expr[i] == '{') {
                size_t j = i + 1;
                unsigned v = 0, nd = 0;
                for (; j < expr.size() && isdigit((unsigned char)expr[j]) && nd < 3; j++, nd++) v = v * 10 + (unsigned)(expr[j] - '0');
                if (!nd || j + 1 >= expr.size() || expr[j] != ',' || expr[j + 1] != '}') return false;
                rep[t] = v;
                i = j + 2;
            } else {
                return false;
            }
            if (rep[t] < 1 || rep[t] > 16) return false;
            if (i < expr.size() && (expr[i] == '?' || expr[i] == '+')) return false; /* lazy / possessive forms */
        }
    } catch (const ParseError &) {
        return false; /* the general parser reports what is wrong with it */
    }
    if (i != expr.size() || cls[0].none() || cls[1].none()) return false;
    out.a = cls[0], out.b = cls[1];
    out.m = rep[0], out.n = rep[1];
    out.single = flags & HS_FLAG_SINGLEMATCH;
    out.quiet = flags & HS_FLAG_QUIET;
    return true;
}

std::vector<Pattern> parse_pattern(const std::string &expr, unsigned flags, unsigned id) {
    check_flags(flags, false);
    /* \Q...\E quotes: rewritten as escaped characters first (also inside classes) */
    std::string p;
    for (size_t k = 0; k < expr.size(); k++) {
        if (expr[k] == '\\' && k + 1 < expr.size() && expr[k + 1] == 'Q') {
            size_t e = expr.find("\\E", k + 2);
            const size_t stop = e == std::string::npos ? expr.size() : e;
            for (size_t q = k + 2; q < stop; q++) {
                const unsigned char c = (unsigned char)expr[q];
                if (!isalnum(c) && c < 0x80) p.push_back('\\');
                p.push_back((char)c);
            }
            k = e == std::string::npos ? expr.size() : e + 1;
            continue;
        }
        p.push_back(expr[k]);
        if (expr[k] == '\\' && k + 1 < expr.size()) p.push_back(expr[++k]);
    }
    /* leading inline options: (?i) (?s) (?m), combined and negated forms */
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
    const unsigned unsupported = HS_FLAG_UCP | HS_FLAG_COMBINATION;
    if (flags & unsupported) throw ParseError{"Unsupported flag for the GPU literal engine."};
    struct Utf8Mode { /* the builder's mode for this expression */
        explicit Utf8Mode(bool on) { g_utf8 = on; }
        ~Utf8Mode() { g_utf8 = false; }
    } utf8_mode((flags & HS_FLAG_UTF8) != 0);
    if ((flags & HS_FLAG_UTF8) && !utf8_valid(expr)) throw ParseError{"Expression is not valid UTF-8."};
    std::vector<Pattern> out;
    size_t from = 0;
    int depth = 0;
    g_rewrites_left = kRewriteBudget;
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
        const std::string branch = p.substr(from, k - from);
        try {
            expand_branch(branch, flags, id, out, 0);
        } catch (const NeverMatch &) { /* an alternative that cannot match is dropped; all of them: an error */
        }
        { /* "a(?i)b|c": a setting at the top level also governs the alternatives after it */
            bool nc = flags & HS_FLAG_CASELESS, ds = flags & HS_FLAG_DOTALL;
            int d = 0;
            for (size_t q = 0; q < branch.size(); q++) {
                size_t e = q;
                if (branch[q] == '\\') { q++; continue; }
                if (branch[q] == '[') { parse_bracket_class(branch, e); q = e - 1; continue; }
                if (branch[q] == '(' && d == 0) { /* (a scoped "(?i:" group keeps its setting to itself) */
                    bool n2 = nc, s2 = ds;
                    if (option_group_at(branch, q, n2, s2, e) == 1) {
                        nc = n2;
                        ds = s2;
                        q = e - 1;
                        continue;
                    }
                }
                if (branch[q] == '(') d++;
                if (branch[q] == ')') d--;
            }
            flags = (flags & ~(HS_FLAG_CASELESS | HS_FLAG_DOTALL)) | (nc ? HS_FLAG_CASELESS : 0) | (ds ? HS_FLAG_DOTALL : 0);
        }
        from = k + 1;
    }
    if (out.empty()) throw NeverMatch();
    for (Pattern &b : out) b.quiet = flags & HS_FLAG_QUIET;
    return out;
}


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

} // namespace hsf
