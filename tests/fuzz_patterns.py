#!/usr/bin/env python3
"""Differential fuzz of the hs_* pattern compiler + host confirm against Python's re, no GPU:
random expressions from a small grammar (literals, classes, groups, alternation, quantifiers incl.
lazy and counted, \\b \\B, ^ $ with and without MULTILINE, CASELESS / DOTALL / SOM_LEFTMOST); for
every expression the facade accepts, the events over random blocks (literal hits from the HWLM
oracle -> hs_confirm_batch) must equal the brute-force model with the whole block visible.
  python tests/fuzz_patterns.py [--seed S] [--n N] [--multi]
(lives under tests/: it drives the oracle, which only test code may do)
Exits non-zero on the first disagreement, printing the expression, flags and block."""
import argparse
import os
import random
import re
import signal
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperscan_amd import hs  # noqa: E402
from tests.test_hs_confirm_cpu import brute_context, run_exprs_auto  # noqa: E402

ALPHA = "abcABCXYxy01 _-\n"
WORDS = ["ab", "abc", "cab", "X1", "Y0", "a-b", "01", "_a", "ba", "XY", "AB", "aBc", "xy", "Cab"]


def gen_atom(r, depth):
    k = r.random()
    if k < 0.40:
        return re.escape(r.choice(WORDS)).replace("\\ ", " ").replace("\\-", "-").replace("\\_", "_")
    if k < 0.55:
        return r.choice(["[a-c]", "[^a\\n]", "\\d", "\\w", "\\s", ".", "[XY01]", "[ab_-]", "\\W", "[^\\w]", "[^XY]", "[A-c]", "[a-cX]",
                         "[^\\d\\s]", "\\S", "\\D", "[\\w-]", "[]a]", "[^]a]", ".{0,3}", "\\w{2,4}", "[^b]{1,2}"])
    if k < 0.62:
        return r.choice(["\\b", "\\B"])
    if k < 0.85 and depth < 3:
        n = r.randint(1, 3)
        alts = [gen_cat(r, depth + 1, r.randint(0, 3)) for _ in range(n)]
        g = r.choice(["(", "(?:", "(?:", "(?i:", "(?s:", "(?-i:"]) + "|".join(alts) + ")"
        if r.random() < 0.25:
            g += r.choice(["?", "*", "+", "{2}", "{1,2}", "{0,3}", "{2,}"])
        return g
    return r.choice(list("abcXY01"))


def gen_cat(r, depth, n):
    out = []
    for _ in range(n):
        a = gen_atom(r, depth)
        if a not in ("\\b", "\\B") and r.random() < 0.3:
            a += r.choice(["?", "*", "+", "{2}", "{1,3}", "{0,2}", "{2,}", "??", "*?", "+?"])
        out.append(a)
    return "".join(out)


def gen_expr(r):
    branches = []
    for _ in range(r.choice([1, 1, 1, 2, 3])):
        b = gen_cat(r, 0, r.randint(1, 5))
        k = r.random()
        if k < 0.15:
            b = "^" + b
        elif k < 0.20:
            b = "\\A" + b
        k = r.random()
        if k < 0.15:
            b = b + "$"
        elif k < 0.20:
            b = b + "\\z"
        elif k < 0.25:
            b = b + "\\Z"
        branches.append(b)
    return "|".join(branches)


MAX_PARTS = 10  # --long: 60 (many hits of one literal per block: the host confirm's shared automaton pass)


def gen_block(r):
    parts = []
    for _ in range(r.randint(0, MAX_PARTS)):
        parts.append(r.choice(WORDS) if r.random() < 0.6 else "".join(r.choice(ALPHA) for _ in range(r.randint(1, 4))))
    return "".join(parts).encode()


class Timeout(Exception):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--multi", action="store_true", help="databases of 2-6 expressions with ext bounds and SINGLEMATCH mixed in")
    ap.add_argument("--utf8", action="store_true", help="HS_FLAG_UTF8: non-ASCII characters in expressions and blocks, code-point model")
    ap.add_argument("--long", action="store_true", help="blocks of up to 60 parts: many hits of the same literal in one block")
    a = ap.parse_args()
    if a.long:
        global MAX_PARTS
        MAX_PARTS = 60
    if a.multi:
        return main_multi(a)
    if a.utf8:
        return main_utf8(a)
    r = random.Random(a.seed)
    signal.signal(signal.SIGALRM, lambda *_: (_ for _ in ()).throw(Timeout()))
    tried = accepted = compared = 0
    flag_choices = [0, 0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL, hs.HS_FLAG_MULTILINE, hs.HS_FLAG_SOM_LEFTMOST,
                    hs.HS_FLAG_CASELESS | hs.HS_FLAG_MULTILINE, hs.HS_FLAG_SINGLEMATCH]
    for _ in range(a.n):
        expr, fl = gen_expr(r), r.choice(flag_choices)
        tried += 1
        try:
            re.compile(expr.replace("\\Z", "(?=\\n?\\Z)").replace("\\z", "\\Z").encode())
        except re.error:
            continue
        try:
            hs.Database.compile([expr], [fl], [1])
        except hs.HsError:
            continue
        accepted += 1
        blocks = [gen_block(r) for _ in range(6)]
        exprs = [(expr, fl, 1)]
        try:
            signal.alarm(10)
            want = brute_context(exprs, blocks)
            signal.alarm(0)
        except Timeout:
            continue
        if fl & hs.HS_FLAG_SINGLEMATCH:  # only the first event of each block
            first, seen = [], set()
            for e in sorted(want, key=lambda e: (e[0], e[3])):
                if e[0] not in seen:
                    seen.add(e[0])
                    first.append(e)
            want = first
        got = run_exprs_auto(exprs, blocks)
        compared += 1
        if sorted(got) != sorted(want):
            bad = sorted(set(got) ^ set(want))[0][0]
            print("MISMATCH", repr(expr), "flags", fl, "block", blocks[bad])
            print("  got ", sorted(e for e in got if e[0] == bad))
            print("  want", sorted(e for e in want if e[0] == bad))
            sys.exit(1)
    print(f"{tried} expressions, {accepted} accepted by the facade, {compared} compared: all equal")


UNI = ["é", "€", "ß", "\U0001f600", "ü", "ı", "α", "ω", "ÿ"]


def main_utf8(a):
    """the byte grammar with non-ASCII characters spliced into expressions and blocks, under
    HS_FLAG_UTF8, against the code-point model (no caseless flags: re.ASCII keeps the model's
    classes ASCII but also its case folding)"""
    from tests.test_hs_confirm_cpu import brute_utf8

    r = random.Random(a.seed)
    signal.signal(signal.SIGALRM, lambda *_: (_ for _ in ()).throw(Timeout()))
    flag_choices = [0, 0, hs.HS_FLAG_DOTALL, hs.HS_FLAG_MULTILINE, hs.HS_FLAG_SOM_LEFTMOST]
    accepted = compared = 0
    for _ in range(a.n):
        expr, fl = gen_expr(r), r.choice(flag_choices) | hs.HS_FLAG_UTF8
        if "(?i" in expr or "\\z" in expr or "\\Z" in expr:
            continue
        if r.random() < 0.5:  # a class of code points somewhere at the top level
            cls = r.choice(["[é€a]", "[^éb]", "[α-ω]", "[ü-ÿX]", "[^\\dß]", "[\\x{e9}-\\x{fc}_]"]) + r.choice(["", "+", "?", "{1,2}"])
            expr = r.choice([cls + expr, expr + cls]) if "|" not in expr else expr
        pieces = list(expr)
        for _k in range(r.randint(0, 2)):  # splice whole characters in where they stay atoms
            pos = r.randrange(len(pieces) + 1)
            if pos > 0 and pieces[pos - 1] == "\\":
                continue
            pieces.insert(pos, r.choice(UNI))
        expr = "".join(pieces)
        pyexpr = re.sub(r"\\x\{([0-9a-f]+)\}", lambda m: chr(int(m.group(1), 16)), expr)  # Python has no \x{..}
        try:
            re.compile(pyexpr)
            hs.Database.compile([expr.encode("utf-8")], [fl], [1])
        except (re.error, hs.HsError):
            continue
        accepted += 1
        blocks = []
        for _b in range(5):
            t = gen_block(r).decode("latin-1")
            t = "".join(ch + (r.choice(UNI) if r.random() < 0.15 else "") for ch in t)
            blocks.append(t.encode("utf-8"))
        try:
            signal.alarm(10)
            want = brute_utf8([(pyexpr, fl, 1)], blocks)
            signal.alarm(0)
        except Timeout:
            continue
        got = run_exprs_auto([(expr.encode("utf-8"), fl, 1)], blocks)
        compared += 1
        if sorted(got) != sorted(want):
            bad = sorted(set(got) ^ set(want))[0][0]
            print("MISMATCH", repr(expr), "flags", fl, "block", blocks[bad])
            print("  got ", sorted(e for e in got if e[0] == bad))
            print("  want", sorted(e for e in want if e[0] == bad))
            sys.exit(1)
    print(f"utf8: {accepted} accepted, {compared} compared: all equal")


def singlematch_filter(want, single_ids):
    """under SINGLEMATCH only the first event of an id in a block is owed"""
    out, seen = [], set()
    for e in sorted(want, key=lambda e: (e[0], e[3], e[1])):
        if e[1] in single_ids:
            if (e[0], e[1]) in seen:
                continue
            seen.add((e[0], e[1]))
        out.append(e)
    return out


def main_multi(a):
    r = random.Random(a.seed)
    signal.signal(signal.SIGALRM, lambda *_: (_ for _ in ()).throw(Timeout()))
    flag_choices = [0, 0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL, hs.HS_FLAG_MULTILINE, hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_SINGLEMATCH]
    dbs = compared = 0
    for _ in range(a.n):
        exprs = []
        for pid in range(1, r.randint(2, 6) + 1):
            for _try in range(20):
                expr, fl = gen_expr(r), r.choice(flag_choices)
                ext = {}
                if r.random() < 0.3:
                    ext["min_offset"] = r.randint(0, 12)
                if r.random() < 0.3:
                    ext["max_offset"] = ext.get("min_offset", 0) + r.randint(0, 25)
                if r.random() < 0.2:
                    ext["min_length"] = r.randint(1, 6)
                if "min_length" in ext and "max_offset" in ext and ext["min_length"] > ext["max_offset"]:
                    del ext["min_length"]
                try:
                    re.compile(expr.replace("\\Z", "(?=\\n?\\Z)").replace("\\z", "\\Z").encode())
                    hs.Database.compile_ext([expr], [fl], [pid], [hs.ExprExt.make(**ext) if ext else None])
                except (re.error, hs.HsError):
                    continue
                exprs.append((expr, fl, pid, ext))
                break
        if len(exprs) < 2:
            continue
        dbs += 1
        blocks = [gen_block(r) for _ in range(5)]
        try:
            signal.alarm(20)
            want = brute_context(exprs, blocks)
            signal.alarm(0)
        except Timeout:
            continue
        want = singlematch_filter(want, {e[2] for e in exprs if e[1] & hs.HS_FLAG_SINGLEMATCH})
        got = run_exprs_auto(exprs, blocks)
        compared += 1
        if sorted(got) != sorted(want):
            bad = sorted(set(got) ^ set(want))[0][0]
            print("MISMATCH", exprs, "block", blocks[bad])
            print("  got ", sorted(e for e in got if e[0] == bad))
            print("  want", sorted(e for e in want if e[0] == bad))
            sys.exit(1)
        for b in range(len(blocks)):  # delivery order inside a block
            tos = [e[3] for e in got if e[0] == b]
            assert tos == sorted(tos), ("order", exprs, blocks[b])
    print(f"{dbs} databases, {compared} compared: all equal")


if __name__ == "__main__":
    main()
