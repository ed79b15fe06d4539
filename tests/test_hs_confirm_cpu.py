"""The host-side "Rose-lite" confirm of the hs_* facade, on the CPU: literal hits come from the
HWLM oracle instead of the GPU (hs_confirm_batch takes the records hsgpu_hwlm_exec_batch would
return), events are checked against a brute-force model built on Python's `re`. Covers the
long-literal check, the shift-and tail automaton, SINGLEMATCH, SOM_LEFTMOST, ext bounds, per-block
termination and the threaded path (>= 8192 hits) with its ordered delivery."""
import ctypes as C
import re

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import hs
from hyperscan_amd.hwlm import MATCH_DTYPE
from tests import oracle_binding as ob


def literal_hits(parts, corpus, off):
    """what the literal engine reports for the facade's HWLM literals: the last <= 8 bytes of each
    pattern's literal prefix, id = pattern index (hs_facade.cpp build_database)"""
    lits = [H.HwlmLiteral(lit[-8:], nocase=bool(fl & hs.HS_FLAG_CASELESS), id=i) for i, (lit, _t, fl, _pid, _e, *_a) in enumerate(parts)]
    got = ob.Oracle(lits).collect_blocks(corpus, off)
    recs = np.zeros(len(got), dtype=MATCH_DTYPE)
    recs["block"], recs["end"], recs["id"], recs["lit"] = got["block"], got["end"], got["id"], got["id"]
    order = np.lexsort((recs["id"], recs["end"], recs["block"]))
    return np.ascontiguousarray(recs[order])


def confirm(db, corpus, off, recs, on_event=None):
    lib = hs._lib()
    lib.hs_confirm_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, hs.BATCH_CB,
                                     C.c_void_p]
    ev = []

    def default(b, i, f, t):
        ev.append((b, i, f, t))
        return False

    fn = on_event or default
    cb = hs.BATCH_CB(lambda b, i, f, t, _fl, _c: 1 if fn(b, i, f, t) else 0)
    rv = lib.hs_confirm_batch(db._h, corpus.ctypes.data, off.ctypes.data, off.size - 1, recs.ctypes.data, recs.size, cb, None)
    return rv, ev


def brute(parts, corpus, off):
    """events per block: (block, id, from, to), one per (id, to), in delivery order"""
    out = []
    for b in range(off.size - 1):
        data = bytes(corpus[int(off[b]):int(off[b + 1])])
        evs = set()
        for lit, tail, fl, pid, ext, *anch in parts:
            anch = anch[0] if anch else ""  # "^" and/or "$" around the branch
            ml = bool(fl & hs.HS_FLAG_MULTILINE)
            rf = (re.I if fl & hs.HS_FLAG_CASELESS else 0) | (re.S if fl & hs.HS_FLAG_DOTALL else 0)
            tre = re.compile(tail.encode("latin-1"), rf) if tail else None
            hay, needle = (data.upper(), lit.upper()) if fl & hs.HS_FLAG_CASELESS else (data, lit)
            k = hay.find(needle)
            while k >= 0:
                s = k + len(lit)
                tos = [s] if tre is None else [to for to in range(s, len(data) + 1) if tre.fullmatch(data, s, to)]
                if "^" in anch and k != 0 and not (ml and data[k - 1:k] == b"\n"):
                    tos = []
                if "$" in anch:  # the end of the data, or just before its last newline (multiline: any newline)
                    tos = [to for to in tos if to == len(data) or (data[to:to + 1] == b"\n" and (ml or to + 1 == len(data)))]
                for to in tos:
                    if "min_offset" in ext and to < ext["min_offset"]:
                        continue
                    if "max_offset" in ext and to > ext["max_offset"]:
                        continue
                    if "min_length" in ext and to - k < ext["min_length"]:
                        continue
                    evs.add((to, pid, k if fl & hs.HS_FLAG_SOM_LEFTMOST else 0))
                k = hay.find(needle, k + 1)
        seen_single, per = set(), {}
        for to, pid, frm in sorted(evs):  # one report per (id, to): the smallest `from` wins the sort
            if (pid, to) in per:
                continue
            per[(pid, to)] = frm
            single = any(p[3] == pid and p[2] & hs.HS_FLAG_SINGLEMATCH for p in parts)
            if single:
                if pid in seen_single:
                    continue
                seen_single.add(pid)
            out.append((b, pid, frm, to))
    return out


def make_db(parts):
    pats = [re.escape(l.decode("latin-1")).replace("\\ ", " ").replace("\\-", "-").replace("\\=", "=").replace("\\/", "/") + t
            for l, t, _f, _p, _e in parts]
    ext = [hs.ExprExt.make(**e) if e else None for _l, _t, _f, _p, e in parts]
    return hs.Database.compile_ext(pats, [p[2] for p in parts], [p[3] for p in parts], ext)


PARTS = [(b"GET /", r"[a-z]+\d", 0, 100, {}), (b"user=", r"\s+\w{2,8}=", hs.HS_FLAG_CASELESS, 101, {}),
         (b"Content-Length", r".{0,16}END", hs.HS_FLAG_DOTALL, 102, {}), (b"abcdefghijkl", r"x?y*z", 0, 103, {}),
         (b"Zq", r"[^\n]{3}", hs.HS_FLAG_CASELESS, 104, {}), (b"key", "", hs.HS_FLAG_SINGLEMATCH, 105, {}),
         (b"0042", r"\d{2,}", hs.HS_FLAG_SOM_LEFTMOST, 106, {}), (b"BEEF", r"[A-F0-9]{4}:", 0, 107, dict(min_offset=40)),
         # (the literal prefix runs through escaped literal characters: "html\." is all prefix)
         (b"html.", r"\w+", 0, 108, dict(max_offset=200, min_length=6))]
WORDS = [b"GET /", b"get /", b"user=", b"USER=", b"Content-Length", b"abcdefghijkl", b"abcdefghijkX", b"Zq", b"zQ", b"abc12",
         b"  key=", b" END", b"xyyz", b"z", b"\n", b"0042", b"004277", b"BEEF", b"BEEF:", b"C0DE:", b"html", b".html.x", b"   ",
         b"q9", b"END", b"key"]


def corpus_of(rng, n_words, n_blocks):
    data = b"".join(WORDS[int(i)] for i in rng.integers(0, len(WORDS), n_words))
    corpus = np.frombuffer(data, dtype=np.uint8).copy()
    cuts = np.sort(rng.integers(0, corpus.size + 1, n_blocks - 1))
    off = np.concatenate([[0], cuts, [corpus.size]]).astype(np.uint64)
    return corpus, off


def test_confirm_matches_brute_force_small():
    rng = np.random.default_rng(5)
    db = make_db(PARTS)
    corpus, off = corpus_of(rng, 1500, 12)
    recs = literal_hits(PARTS, corpus, off)
    rv, ev = confirm(db, corpus, off, recs)
    assert rv == hs.HS_SUCCESS
    want = brute(PARTS, corpus, off)
    assert sorted(ev) == sorted(want) and len(want) > 100
    # delivery: block order, then non-decreasing `to`, one event per (block, id, to)
    assert [e[0] for e in ev] == sorted(e[0] for e in ev)
    for b in set(e[0] for e in ev):
        tos = [e[3] for e in ev if e[0] == b]
        assert tos == sorted(tos)
    assert len({(e[0], e[1], e[3]) for e in ev}) == len(ev)


def test_confirm_threaded_path_equals_serial_and_brute_force():
    rng = np.random.default_rng(6)
    parts = PARTS[:5] + PARTS[6:]  # without the SINGLEMATCH pattern: more events
    db = make_db(parts)
    corpus, off = corpus_of(rng, 60_000, 700)
    recs = literal_hits(parts, corpus, off)
    assert recs.size >= 8192, "needs enough hits for the worker threads to be used"
    rv, ev = confirm(db, corpus, off, recs)
    assert rv == hs.HS_SUCCESS
    # serial reference: the same records fed block by block (each call below the thread threshold)
    serial = []
    bounds = np.searchsorted(recs["block"], np.arange(off.size))
    for b in range(off.size - 1):
        sl = recs[bounds[b]:bounds[b + 1]]
        if sl.size:
            _rv, e1 = confirm(db, corpus, off, np.ascontiguousarray(sl))
            serial += e1
    assert ev == serial
    sample = rng.choice(off.size - 1, 40, replace=False)
    want = {e for e in brute(parts, corpus, off) if e[0] in set(sample.tolist())}
    assert {e for e in ev if e[0] in set(sample.tolist())} == want


def test_confirm_termination_and_argument_checks():
    rng = np.random.default_rng(7)
    db = make_db(PARTS)
    corpus, off = corpus_of(rng, 800, 6)
    recs = literal_hits(PARTS, corpus, off)
    _rv, all_ev = confirm(db, corpus, off, recs)
    stop_block = all_ev[len(all_ev) // 2][0]
    seen = []

    def stopper(b, i, f, t):
        seen.append((b, i, f, t))
        return b == stop_block  # stop that block at its first event

    rv, _ = confirm(db, corpus, off, recs, stopper)
    assert rv == hs.HS_SCAN_TERMINATED
    assert sum(1 for e in seen if e[0] == stop_block) == 1  # a non-zero return ends THAT block only
    assert [e for e in seen if e[0] != stop_block] == [e for e in all_ev if e[0] != stop_block]
    # malformed record arrays are refused
    bad = recs.copy()
    bad[0], bad[-1] = recs[-1], recs[0]
    assert confirm(db, corpus, off, bad)[0] == hs.HS_INVALID  # out of order
    bad = recs.copy()
    bad["id"][3] = 999
    assert confirm(db, corpus, off, bad)[0] == hs.HS_INVALID  # no such pattern
    bad = recs.copy()
    bad["end"][0] = 1 << 30
    assert confirm(db, corpus, off, bad)[0] == hs.HS_INVALID  # beyond its block
    # a hit whose long literal does not really end there produces nothing (CHECK_LONG_LIT's job)
    fake = np.zeros(1, dtype=MATCH_DTYPE)
    pos = bytes(corpus).find(b"abcdefghijkX")
    if pos >= 0:
        b = int(np.searchsorted(off, pos, side="right")) - 1
        end = pos + 11 - int(off[b])
        if end < int(off[b + 1] - off[b]):
            fake["block"], fake["end"], fake["id"], fake["lit"] = b, end, 3, 3
            assert confirm(db, corpus, off, fake) == (hs.HS_SUCCESS, [])


GROUPED = [(b"GET /", r"(index|home|a+b)\.html?", 0, 200, {}), (b"user=", r"(?:[a-z]+|\d{2,4})&", hs.HS_FLAG_CASELESS, 201, {}),
           (b"key", r"(=|: ?)(true|false|[0-9]+)", 0, 202, {}), (b"BEEF", r"((ab|c)*d){1,3}", 0, 203, {}),
           (b"Zq", r"(x(y|z)?){2,}w", hs.HS_FLAG_SOM_LEFTMOST, 204, {}), (b"0042", r"(7|77)*", 0, 205, {}),
           (b"html", r"(\.[a-z]{1,3}){0,2}(;|$)".replace("|$", ""), 0, 206, dict(min_length=5)),
           (b"END", r"(a|)(b|)c", 0, 207, {}), (b"abc", r"(?:1(2(3)?)?)?.", hs.HS_FLAG_DOTALL, 208, {})]
GWORDS = [b"GET /", b"index", b"home", b"aab", b"ab", b".htm", b".html", b"l", b"user=", b"USER=", b"bob", b"123", b"12345", b"&",
          b"key", b"=", b": ", b":", b"true", b"false", b"77", b"BEEF", b"abd", b"cd", b"d", b"Zq", b"xy", b"xz", b"x", b"w", b"0042",
          b"7", b"html", b".x", b".abc", b";", b"END", b"a", b"b", b"c", b"abc", b"1", b"2", b"3", b"\n", b" "]


def test_grouped_tails_match_brute_force():
    """tails with groups and alternation run on the position automaton; same brute-force model"""
    global WORDS
    rng = np.random.default_rng(9)
    db = make_db(GROUPED)
    saved, WORDS = WORDS, GWORDS
    try:
        corpus, off = corpus_of(rng, 6000, 25)
    finally:
        WORDS = saved
    # one directed block per pattern family on top of the random ones
    directed = b"GET /aaab.html GET /home.htm Zqxyxzxw Zqxxxw Zqxw user=Bob& user=123& END" b"c ENDabc"
    corpus = np.concatenate([corpus, np.frombuffer(directed, dtype=np.uint8)])
    off = np.concatenate([off, [corpus.size]]).astype(np.uint64)
    recs = literal_hits(GROUPED, corpus, off)
    rv, ev = confirm(db, corpus, off, recs)
    assert rv == hs.HS_SUCCESS
    want = brute(GROUPED, corpus, off)
    assert sorted(ev) == sorted(want)
    hit_ids = {e[1] for e in ev}
    assert hit_ids == {p[3] for p in GROUPED}, hit_ids


def test_grouped_tail_compile_errors_and_info():
    import pytest
    for bad in ["foo(bar", "foo(a))", "foo(a)|", "foo(?=a)", "foo(a*+)", "foo(" + "a?" * 4097 + ")", "foo.{4097}x"]:
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [0], [1])
    # widths through groups (hs_expression_info): min over alternatives, max = longest, unbounded loops
    assert hs.expression_info("foo(a|bcd)x") == (5, 7)
    assert hs.expression_info("foo(ab)*") == (3, 0xffffffff)
    assert hs.expression_info("foo(ab|c){2,3}") == (5, 9)
    assert hs.expression_info("foo(a|)") == (3, 4) and hs.expression_info("foo()") == (3, 3)


def test_top_level_alternation_and_anchors():
    """every top-level branch is its own literal-prefixed pattern under the same id; `^` / `$`
    (plain and HS_FLAG_MULTILINE) restrict where a branch may start / end. `$` reports the
    offset before an optional final newline (the reference's -1 offset adjust,
    src/parser/buildstate.cpp:234-241)."""
    ML, SOM = hs.HS_FLAG_MULTILINE, hs.HS_FLAG_SOM_LEFTMOST
    exprs = [("abc|defghi\\d|abd+", 0, 300), ("^GET /x", 0, 301), ("end$", 0, 302), ("^key=\\w+$", ML, 303),
             ("line$", ML, 304), ("^abc|xyz$", SOM, 305), ("cost\\$|bc$", 0, 306)]
    parts = [(b"abc", "", 0, 300, {}), (b"defghi", r"\d", 0, 300, {}), (b"ab", r"d+", 0, 300, {}),
             (b"GET /x", "", 0, 301, {}, "^"), (b"end", "", 0, 302, {}, "$"), (b"key=", r"\w+", ML, 303, {}, "^$"),
             (b"line", "", ML, 304, {}, "$"), (b"abc", "", SOM, 305, {}, "^"), (b"xyz", "", SOM, 305, {}, "$"),
             (b"cost$", "", 0, 306, {}), (b"bc", "", 0, 306, {}, "$")]
    import pytest
    with pytest.raises(hs.HsError):  # one branch without a mandatory literal refuses the whole expression
        hs.Database.compile(["cost\\$|\\w+$"], [0], [1])
    with pytest.raises(hs.HsError):
        hs.Database.compile(["abc|"], [0], [1])
    with pytest.raises(hs.HsError):
        hs.Database.compile(["ab^c"], [0], [1])
    with pytest.raises(hs.HsError):
        hs.Database.compile(["ab$c"], [0], [1])
    db = hs.Database.compile([e[0] for e in exprs], [e[1] for e in exprs], [e[2] for e in exprs])
    words = [b"abc", b"defghi", b"7", b"abdd", b"GET /x", b"end", b"key=", b"v1", b"line", b"xyz", b"cost$", b"bc", b"\n", b"\n", b" "]
    rng = np.random.default_rng(12)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 14)))) for _ in range(400)]
    blocks += [b"GET /x end", b"GET /x end\n", b"end\n\n", b"key=v1", b"x\nkey=v1\nline\nxyz", b"abcxyz", b"abc xyz\n", b"cost$bc\n", b""]
    corpus = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
    off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
    recs = literal_hits(parts, corpus, off)
    rv, ev = confirm(db, corpus, off, recs)
    assert rv == hs.HS_SUCCESS
    want = brute(parts, corpus, off)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == {e[2] for e in exprs}
    # expression info: reference rows with anchors and alternation (unit/hyperscan/expr_info.cpp:182-228)
    lib = hs._lib()
    for pat, fl, row in [("abc|defghi", 0, (3, 6, 0, 0, 0)), ("^foo", 0, (3, 3, 0, 0, 0)), ("^foo.*bar", 0, (6, 0xffffffff, 0, 0, 0)),
                         ("^foo.*bar?", 0, (5, 0xffffffff, 0, 0, 0)), ("^foo.*bar$", 0, (6, 0xffffffff, 1, 1, 1)),
                         ("^foobar$", 0, (6, 6, 1, 1, 1)), ("foobar$", 0, (6, 6, 1, 1, 1)), ("foobar$", ML, (6, 6, 1, 1, 0)),
                         ("foo\\b", 0, (3, 3, 1, 1, 0)), ("\\bfoo", 0, (3, 3, 0, 0, 0)), ("^\\bfoo", 0, (3, 3, 0, 0, 0)),
                         ("\\Bfoo", 0, (3, 3, 0, 0, 0)),
                         # \z / \Z flags as in the table's bare "\\z" (0, 1, 1) and "\\Z" (1, 1, 1) rows
                         ("eod\\z", 0, (3, 3, 0, 1, 1)), ("eod\\Z", 0, (3, 3, 1, 1, 1)),
                         # groups distributed into branches: expr_info.cpp:212-213,218-219
                         ("(^|\n)foo", 0, (3, 4, 0, 0, 0)), ("(^\n|)foo", 0, (3, 4, 0, 0, 0)),
                         ("(foo|bar\\z)", 0, (3, 3, 0, 1, 0)), ("(foo|bar)\\z", 0, (3, 3, 0, 1, 1))]:
        info, err = C.POINTER(hs.ExprInfo)(), C.POINTER(hs.CompileErrorStruct)()
        assert lib.hs_expression_ext_info(pat.encode(), fl, None, C.byref(info), C.byref(err)) == hs.HS_SUCCESS, pat
        i = info.contents
        assert (i.min_width, i.max_width, ord(i.unordered_matches), ord(i.matches_at_eod), ord(i.matches_only_at_eod)) == row, pat
        C.CDLL(None).free(info)


def brute_full(exprs, blocks):
    """events from the whole expression on Python's re: for every end offset, is there a start
    with a full match (the smallest one is the SOM_LEFTMOST `from`)"""
    out = []
    for b, data in enumerate(blocks):
        for pat, fl, pid in exprs:
            rf = (re.I if fl & hs.HS_FLAG_CASELESS else 0) | (re.S if fl & hs.HS_FLAG_DOTALL else 0) | \
                 (re.M if fl & hs.HS_FLAG_MULTILINE else 0)
            rx = re.compile(pat.encode("latin-1"), rf)
            for to in range(len(data) + 1):
                froms = [f for f in range(to) if rx.fullmatch(data, f, to)]
                if froms:
                    out.append((b, pid, min(froms) if fl & hs.HS_FLAG_SOM_LEFTMOST else 0, to))
    return out


def run_exprs(exprs, lits, blocks, ext=None):
    """compile `exprs`, take the literal hits for `lits` (the literal each branch is expected to be
    keyed on, in branch order) from the oracle, and run the facade's confirm"""
    db = hs.Database.compile_ext([e[0] for e in exprs], [e[1] for e in exprs], [e[2] for e in exprs], ext or [None] * len(exprs))
    parts = [(l, "", fl, 0, {}) for l, fl in lits]
    corpus = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
    off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
    rv, ev = confirm(db, corpus, off, literal_hits(parts, corpus, off))
    assert rv == hs.HS_SUCCESS
    return ev


def test_order_cpp_patterns():
    # unit/hyperscan/order.cpp:65-96 (ordering1): counts per id on 32 x 'a'
    D = hs.HS_FLAG_DOTALL
    exprs = [("aa", D, 1), ("aa.", D, 2), ("aa..", D, 3), ("^.{0,4}aa..", D, 4), ("^.{0,4}aa", D, 5)]
    ev = run_exprs(exprs, [(b"aa", D)] * 5, [b"a" * 32])
    cnt = {i: sum(1 for e in ev if e[1] == i) for i in range(1, 6)}
    assert cnt == {1: 31, 2: 30, 3: 29, 4: 5, 5: 5}
    assert [e[3] for e in ev] == sorted(e[3] for e in ev)
    assert sorted(ev) == sorted(brute_full(exprs, [b"a" * 32]))


def test_literal_in_the_middle_matches_brute_force():
    """branches whose literal is not at the front (R1 LIT R2): the part in front runs backwards
    from the literal as a reversed position automaton"""
    I, S, SOM, ML = hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL, hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_MULTILINE
    exprs = [(r"[a-z]+@example\.(com|org)", 0, 1), (r"\d{1,3}\.\d{1,3}:8080", 0, 2), (r"(GET|POST) /index", 0, 3),
             (r"^\s*key\s*=", ML, 4), (r"x?y*zfoo[0-9]", I, 5), (r"[ab]+cd[ab]+", SOM, 6), (r"^.{2,5}END", S, 7),
             (r"(ab|c)+def(g|hi)*j", SOM, 8), (r"a.cab", 0, 9), (r"\w+\.txt|\d+\.dat", 0, 10)]
    lits = [(b"@example.", 0), (b":8080", 0), (b" /index", 0), (b"key", ML), (b"zfoo", I), (b"cd", SOM), (b"END", S), (b"def", SOM),
            (b"cab", 0), (b".txt", 0), (b".dat", 0)]
    words = [b"bob", b"@example.", b"com", b"org", b"10", b".", b"255", b":8080", b"GET", b"POST", b" /index", b"  ", b"key", b"=",
             b"\n", b"xy", b"yyZFOO7", b"zfoo", b"ab", b"ba", b"cd", b"END", b"c", b"def", b"g", b"hi", b"j", b"a", b"cab", b"acab",
             b"f1.txt", b"77.dat", b" "]
    rng = np.random.default_rng(31)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 22)))) for _ in range(120)]
    blocks += [b"key=", b"\n key =", b"xkey=", b"abEND", b"aEND", b"abcdefEND", b"ababcdab", b"abcdefghij", b"cdefj", b"a\ncab", b"",
               b"mail bob@example.com, ba@example.org!", b"at 10.255:8080 and 1234.5:8080"]
    ev = run_exprs(exprs, lits, blocks)
    want = brute_full(exprs, blocks)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == set(range(1, 11))
    # min_length counts from the leftmost start of the match
    ext = [None] * len(exprs)
    ext[5] = hs.ExprExt.make(min_length=6)
    ev2 = run_exprs(exprs, lits, blocks, ext)
    want2 = [e for e in want if e[1] != 6 or e[3] - e[2] >= 6]
    assert sorted(ev2) == sorted(want2) and len(want2) < len(want)
    # widths (hs_expression_info) through both sides
    assert hs.expression_info(r"^.{0,4}aa..") == (4, 8)
    assert hs.expression_info(r"[a-z]+@example\.(com|org)") == (13, 0xffffffff)
    import pytest
    for bad in [r"\w+", r"(foo|\w)z?", r"a*", r"(abc)?"]:  # no mandatory literal (and no small class to stand in)
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [0], [1])


def brute_context(exprs, blocks):
    """like brute_full, but every candidate end offset is tested with the whole block visible
    (a lookahead pins the end), so \\b, \\B and the end anchors see the real neighbours.
    Dialect: the reference's \\z is Python's \\Z; its \\Z is Python's (?=\\n?\\Z).
    exprs: (pattern, flags, id[, ext dict]); ext bounds as hs_expr_ext_t (min_length from the
    leftmost start)."""
    out = []
    for b, data in enumerate(blocks):
        for pat, fl, pid, *rest in exprs:
            ext = rest[0] if rest else {}
            rf = (re.I if fl & hs.HS_FLAG_CASELESS else 0) | (re.S if fl & hs.HS_FLAG_DOTALL else 0) | \
                 (re.M if fl & hs.HS_FLAG_MULTILINE else 0)
            py = pat.replace("\\Z", "(?=\\n?\\Z)").replace("\\z", "\\Z")
            for to in range(len(data) + 1):
                rx = re.compile(("(?:%s)(?=[\\s\\S]{%d}\\Z)" % (py, len(data) - to)).encode("latin-1"), rf)
                m = rx.search(data)  # the leftmost start that can end at `to`
                if m and m.start() == to:  # (an empty match there is no match: look further left is moot)
                    m = None
                if not m:
                    continue
                if "min_offset" in ext and to < ext["min_offset"]:
                    continue
                if "max_offset" in ext and to > ext["max_offset"]:
                    continue
                if "min_length" in ext and to - m.start() < ext["min_length"]:
                    continue
                out.append((b, pid, m.start() if fl & hs.HS_FLAG_SOM_LEFTMOST else 0, to))
    return out


def run_exprs_auto(exprs, blocks):
    """run_exprs with the literals taken from the database itself (hs_database_literal);
    exprs: (pattern, flags, id[, ext dict])"""
    ext = [hs.ExprExt.make(**e[3]) if len(e) > 3 and e[3] else None for e in exprs]
    db = hs.Database.compile_ext([e[0] for e in exprs], [e[1] for e in exprs], [e[2] for e in exprs], ext)
    return run_exprs(exprs, [(b, hs.HS_FLAG_CASELESS if nc else 0) for b, nc, _rid in db.literals()], blocks, ext)


def test_word_boundaries_and_absolute_anchors():
    """\\b / \\B at the edges of a branch and hugging its literal, \\A, \\z, \\Z"""
    I, SOM, ML = hs.HS_FLAG_CASELESS, hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_MULTILINE
    exprs = [(r"\bcat\b", 0, 1), (r"\Bcat", 0, 2), (r"cat\B", I, 3), (r"\bdog\b\s+\w+", SOM, 4), (r"\s+\bfish", 0, 5),
             (r"\Astart", ML, 6), (r"end\z", 0, 7), (r"end\Z", 0, 8), (r"\bend\b$", ML, 9), (r"[a-z]+\B7up", SOM, 10),
             (r"\b\d+ cats?\b", 0, 11), (r"x\W*\bcat", 0, 12), (r"^\bstart|\bfish\b$", 0, 13)]
    words = [b"cat", b"CAT", b"cats", b"dog", b" ", b"  ", b"fish", b"start", b"end", b"\n", b"7up", b"ab", b"12", b" cat", b"x",
             b"-", b"_"]
    rng = np.random.default_rng(44)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 12)))) for _ in range(150)]
    blocks += [b"cat", b"concat cat", b"cat\n", b"start\nstart", b"end", b"end\n", b"end\n\n", b"x end\nend", b"ab7up 7up", b"12 cats",
               b"dog  fish", b"a dog fish-", b"xcat", b"x-cat", b"start fish", b"fish\n"]
    ev = run_exprs_auto(exprs, blocks)
    want = brute_context(exprs, blocks)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == set(range(1, 14))
    import pytest
    for bad in [r"a\zb", r"ca(\z)*t", r"\b", r"cat\b+"]:  # anchors inside, assertions alone or quantified at the literal
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [0], [1])


def test_lazy_quantifiers_inline_flags_named_groups_posix_classes():
    """syntax that changes nothing about WHICH end offsets match: lazy quantifiers (every end is
    reported anyway), leading (?ims-ims) options, named / commented groups, [:posix:] classes"""
    SOM = hs.HS_FLAG_SOM_LEFTMOST
    I, S, M = hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL, hs.HS_FLAG_MULTILINE
    # (expression, flags) as compiled here | (expression, flags) of the same language for Python's re
    pairs = [((r"foo.+?bar", 0), (r"foo.+bar", 0)), ((r"ab??c*?d{1,2}?e", SOM), (r"ab?c*d{1,2}e", SOM)),
             ((r"(?i)select\s+\w+", 0), (r"select\s+\w+", I)), ((r"(?is)begin.{0,6}end", 0), (r"begin.{0,6}end", I | S)),
             ((r"(?-i)Case[a-z]", I), (r"Case[a-z]", 0)), ((r"id=(?<num>\d+);", 0), (r"id=(\d+);", 0)),
             ((r"id=(?P<n>[a-f]+)(?#hex);", 0), (r"id=([a-f]+);", 0)),
             ((r"tag[[:digit:][:upper:]]+[[:^alnum:]]", 0), (r"tag[0-9A-Z]+[^0-9A-Za-z]", 0)),
             ((r"[[:space:]]+key[[:punct:]]", 0), (r"\s+key[!-/:-@\[-`{-~]", 0)), ((r"(?m)^row\d$", 0), (r"^row\d$", M)),
             # \h \v are classes (src/parser/ComponentClass.cpp:87-88,114-115), [\b] is a backspace, octal escapes
             ((r"foo\h+bar\v", 0), (r"foo[\x09\x20\xa0]+bar[\x0a-\x0d\x85]", 0)), ((r"tag\H\V[\b\41-\43]\041\0", 0), (r"tag[^\x09\x20\xa0][^\x0a-\x0d\x85][\x08!-#]!\x00", 0))]
    exprs = [(h[0], h[1], i + 1) for i, (h, _p) in enumerate(pairs)]
    pyexprs = [(p[0], p[1], i + 1) for i, (_h, p) in enumerate(pairs)]
    words = [b"foo", b"bar", b"x", b"ab", b"a", b"b", b"c", b"d", b"e", b"SELECT", b"select", b" ", b"name", b"begin", b"\n", b"END",
             b"end", b"Case", b"case", b"id=", b"42", b"cafe", b";", b"tag", b"7", b"Q", b"-", b"key", b"=", b"row", b"3"]
    rng = np.random.default_rng(52)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 14)))) for _ in range(160)]
    blocks += [b"fooxbarxbar", b"abccdde abe ade", b"Select  x", b"BEGIN\n\nEnd", b"Casex CASEx", b"id=42;id=cafe;", b"tag7Q-", b" \tkey=",
               b"row3\nrow4\nrow55\n", b"foo \t\xa0bar\x85 foo bar\n foo  bar\x0b", b"tagxy\x08!\x00 tagzz#!\x00 tag z!!\x00 tagz\n!!\x00"]
    ev = run_exprs_auto(exprs, blocks)
    want = brute_context(pyexprs, blocks)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == set(range(1, len(pairs) + 1))
    import pytest
    for bad in [r"foo.*+bar", r"foo(?=bar)", r"foo(?<!x)bar", r"(?x)foo", r"foo[[:nope:]]", r"foo(?m)bar", r"foo(?x)bar"]:
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [0], [1])


def test_wide_position_automata():
    """fragments of more than 63 positions (several 64-bit words per position set; chains by
    shift, the rest through exception rows), forwards and reversed"""
    S, SOM = hs.HS_FLAG_DOTALL, hs.HS_FLAG_SOM_LEFTMOST
    exprs = [(r"^foo.{64}b(a?)r", S, 1), (r"foo.{100,120}bar", S, 2), (r"[a-z]{70}END", SOM, 3), (r"(ab|cd){40}x", 0, 4),
             (r"\d{70,}:8080", SOM, 5), (r"(x|yz){35}end(ing)?", 0, 6), (r"key(=[a-f]{2}){33}", 0, 7), (r"a.{200}b", S, 8),
             (r"q[ab]{0,70}q", 0, 9)]
    rng = np.random.default_rng(61)

    def rnd(alpha, n):
        return bytes(rng.choice(np.frombuffer(alpha, np.uint8), n))

    blocks = [b"foo" + rnd(b"xyz\n", 64) + b"bar" + b"foo" + rnd(b"xy", 64) + b"br", b"xfoo" + rnd(b"xyz", 64) + b"bar",
              b"foo" + rnd(b"abr", 130) + b"bar", b"foo" + rnd(b"a", 99) + b"bar", b"foo" + rnd(b"a", 121) + b"bar",
              rnd(b"abcxyz", 90) + b"END", rnd(b"abc", 69) + b"END", rnd(b"abc", 40) + b"1" + rnd(b"abc", 70) + b"END",
              b"".join([b"ab", b"cd"][int(i)] for i in rng.integers(0, 2, 47)) + b"x", b"ab" * 39 + b"x", b"ab" * 40 + b"y" + b"cd" * 40 + b"x",
              rnd(b"0123456789", 85) + b":8080", rnd(b"0123456789", 69) + b":8080",
              b"".join([b"x", b"yz"][int(i)] for i in rng.integers(0, 2, 44)) + b"ending", b"yz" * 34 + b"end",
              b"key" + b"".join(b"=" + rnd(b"abcdef", 2) for _ in range(36)), b"key" + b"=ab" * 32 + b"=a",
              b"a" * 150 + rnd(b"ab\n", 120) + b"b" * 5, b"q" + rnd(b"ab", 70) + b"q" + rnd(b"ab", 30) + b"q", b"q" + b"a" * 71 + b"q", b"qq"]
    ev = run_exprs_auto(exprs, blocks)
    want = brute_context(exprs, blocks)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == set(range(1, 10))
    assert hs.expression_info(r"^foo.{600}bar") == (606, 606) and hs.expression_info(r"[a-z]{70,}END") == (73, 0xffffffff)


def test_literals_inside_an_alternation_group():
    """X(A|B)Y without a top-level literal is distributed into XAY|XBY (nested groups too)"""
    SOM = hs.HS_FLAG_SOM_LEFTMOST
    exprs = [(r"\b(foo|bar|baz)\b", 0, 1), (r"(GET|POST|HEAD) /[a-z]*", 0, 2), (r"(a|b|b$)", 0, 3), (r"(foo|bar).*\z", hs.HS_FLAG_DOTALL, 4),
             (r"[0-9]+(px|em|(r|v)em)\b", SOM, 5), (r"(?:x|yy)(?:1|22)\d", 0, 6), (r"(^GET|^PUT)\s", hs.HS_FLAG_MULTILINE, 7)]
    db = hs.Database.compile([e[0] for e in exprs], [e[1] for e in exprs], [e[2] for e in exprs])
    assert [b for b, _nc, _id in db.literals()] == [b"foo", b"bar", b"baz", b"GET /", b"POST /", b"HEAD /", b"a", b"b", b"b", b"foo", b"bar",
                                                   b"px", b"em", b"rem", b"vem", b"x1", b"x22", b"yy1", b"yy22", b"GET", b"PUT"]
    words = [b"foo", b"bar", b"baz", b" ", b"GET", b"POST", b"HEAD", b" /", b"idx", b"a", b"b", b"\n", b"12", b"px", b"em", b"rem", b"vem",
             b"x", b"yy", b"1", b"22", b"7", b"PUT", b"-"]
    rng = np.random.default_rng(71)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 12)))) for _ in range(200)]
    blocks += [b"foo bar,baz", b"foobar", b"GET /idx POST /", b"ab", b"b", b"xfoo\nbar", b"12px 3rem 4vemx 5em", b"x17 yy229 x2 yy1", b"GET \nPUT\t",
               b"xGET "]
    ev = run_exprs_auto(exprs, blocks)
    want = brute_context(exprs, blocks)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == set(range(1, 8))
    import pytest
    for bad in [r"(foo|\w)x?", r"(foo|bar)?", r"(\w|.)+", r"(?=foo|bar)"]:
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [0], [1])


def test_word_boundaries_inside_fragments():
    """\\b / \\B anywhere in R1 or R2: conditional layers of the position automaton"""
    SOM, S = hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_DOTALL
    exprs = [(r"foo.*\bbar", 0, 1), (r"foo\b.*\bbar\b", S, 2), (r"\w+\b\s+\bneedle", SOM, 3), (r"(\bcat|x)dog\B.", 0, 4),
             (r"a(\b|c)d!", 0, 5), (r"key\b\b=\B\B-", 0, 6), (r"(?:\w\b.)+end", SOM, 7), (r"one(\b.|\B-)*two", S, 8),
             (r"\d+\b\W*[a-z]*\Bzz9", SOM, 9)]
    words = [b"foo", b"bar", b" ", b"x", b"needle", b"cat", b"dog", b"a", b"c", b"d!", b"key", b"=", b"v", b"-", b"end", b"one", b"two", b"12",
             b"ab", b"zz9", b"\n", b"q", b"r", b"s", b"t", b"u", b"_", b"."]
    rng = np.random.default_rng(83)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 13)))) for _ in range(300)]
    blocks += [b"foo bar", b"foobar", b"foo xbar bar.", b"ab  needle", b"ab needlex", b"catdogs xdogs dog.", b"xcatdogs", b"ad! acd! a d!",
               b"key=-v key =-", b"12 abzz9 7-azz9", b"a.b-end", b"ab.end", b"one - two", b"one--two", b"one-.-two", b"12abzz9", b"12 zz9", b"12azz9", b"qr stu st u"]
    ev = run_exprs_auto(exprs, blocks)
    want = brute_context(exprs, blocks)
    assert sorted(ev) == sorted(want)
    hit = {e[1] for e in ev}
    assert hit >= {1, 2, 3, 4, 5, 6, 7, 8, 9}, hit


def test_caseless_negated_class():
    # found by tests/fuzz_patterns.py: caseless [^a] excludes a AND A (fold before negating)
    exprs = [(r"XY??[^a\n]+?", hs.HS_FLAG_CASELESS | hs.HS_FLAG_MULTILINE, 1), (r"ab[^b-c]x", hs.HS_FLAG_CASELESS, 2)]
    blocks = [b"XYa-b_aa-b", b"xyAz", b"abBx abCx abdx ABDX abax"]
    assert sorted(run_exprs_auto(exprs, blocks)) == sorted(brute_context(exprs, blocks))


def test_differential_fuzz_against_re():
    """a fixed slice of tests/fuzz_patterns.py: random expressions from its grammar, every accepted
    one compared with the brute-force model on random blocks"""
    import random
    import sys
    import os

    from tests import fuzz_patterns as fz

    r = random.Random(5)
    flags = [0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL, hs.HS_FLAG_MULTILINE, hs.HS_FLAG_SOM_LEFTMOST]
    compared = 0
    for _ in range(500):
        expr, fl = fz.gen_expr(r), r.choice(flags)
        try:
            re.compile(expr.replace("\\Z", "(?=\\n?\\Z)").replace("\\z", "\\Z").encode())
            hs.Database.compile([expr], [fl], [1])
        except (re.error, hs.HsError):
            continue
        blocks = [fz.gen_block(r) for _ in range(4)]
        assert sorted(run_exprs_auto([(expr, fl, 1)], blocks)) == sorted(brute_context([(expr, fl, 1)], blocks)), (expr, fl, blocks)
        compared += 1
    assert compared >= 250


def test_options_inside_a_pattern():
    """(?i) / (?s) set inside a pattern hold to the end of the enclosing group (and through its
    later alternatives); (?i:...) holds inside its own group. The scoped form means the same in
    Python; the unscoped form is modelled by writing its scope out."""
    S = hs.HS_FLAG_DOTALL
    # (expression here, the same language for Python)
    pairs = [(r"foo(?i:bar)baz", r"foo(?i:bar)baz"), (r"nested(?i:less(?-i:ful)less)lit", r"nested(?i:less(?-i:ful)less)lit"),
             (r"(?i:hat|kettle)s", r"(?i:hat|kettle)s"), (r"foo.*(?i-s:bar.*baz).*bing", r"foo.*(?i-s:bar.*baz).*bing"),
             (r"ab(?i)cdef(?-i)ghi", r"ab(?i:cdef)ghi"), (r"g(?i)odzilla", r"g(?i:odzilla)"), (r"(?i)god(?-i)z(?i)illa", r"(?i:god)z(?i:illa)"),
             (r"x(a(?i)b|c)d", r"x(a(?i:b)|(?i:c))d"), (r"k(?s).y|m.n", r"k(?s:.)y|m(?s:.)n"), (r"[a-c]+(?i)lit\d", r"[a-c]+(?i:lit)\d")]
    exprs = [(h, S if "bing" in h else 0, i + 1) for i, (h, _p) in enumerate(pairs)]
    pyexprs = [(p, S if "bing" in p else 0, i + 1) for i, (_h, p) in enumerate(pairs)]
    words = [b"foo", b"bar", b"BAR", b"baz", b"BAZ", b"nested", b"less", b"LESS", b"ful", b"FUL", b"lit", b"LIT", b"hat", b"HAT", b"kettle",
             b"s", b"bing", b"\n", b"ab", b"cdef", b"CDEF", b"ghi", b"GHI", b"g", b"G", b"odzilla", b"ODZILLA", b"god", b"GOD", b"z", b"Z", b"illa",
             b"x", b"a", b"b", b"B", b"c", b"C", b"d", b"k", b"y", b"m", b"n", b"7", b" "]
    rng = np.random.default_rng(93)
    blocks = [b"".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 9)))) for _ in range(400)]
    blocks += [b"fooBARbaz fooBARBAZ", b"nestedLESSfulLESSlit nestedlessFULlesslit", b"HATs kettleS", b"foo BAR\nbaz", b"foo bar baz bing",
               b"abCDEFghi ABcdefghi abcdefGHI", b"gODZILLA Godzilla", b"GODzILLA godZilla", b"xaBd xCd xAbd", b"k\ny m\nn", b"abcLIT7 ABlit7"]
    ev = run_exprs_auto(exprs, blocks)
    want = brute_context(pyexprs, blocks)
    assert sorted(ev) == sorted(want)
    assert {e[1] for e in ev} == set(range(1, len(pairs) + 1))


def brute_utf8(exprs, blocks, ascii_classes=True):
    """the model in the code-point domain: blocks are valid UTF-8, Python matches the decoded
    string (re.ASCII keeps \\w \\d \\s \\b ASCII-only, as HS_FLAG_UTF8 without HS_FLAG_UCP does) and
    offsets are mapped back to bytes"""
    out = []
    for b, raw in enumerate(blocks):
        text = raw.decode("utf-8")
        at = [0]
        for ch in text:
            at.append(at[-1] + len(ch.encode("utf-8")))
        for pat, fl, pid in exprs:
            rf = (re.I if fl & hs.HS_FLAG_CASELESS else 0) | (re.S if fl & hs.HS_FLAG_DOTALL else 0) | \
                 (re.M if fl & hs.HS_FLAG_MULTILINE else 0) | (re.ASCII if ascii_classes else 0)
            for to in range(len(text) + 1):
                rx = re.compile("(?:%s)(?=[\\s\\S]{%d}\\Z)" % (pat, len(text) - to), rf)
                m = rx.search(text)
                if m and m.start() < to:
                    out.append((b, pid, at[m.start()] if fl & hs.HS_FLAG_SOM_LEFTMOST else 0, at[to]))
    return out


def test_utf8_mode():
    """HS_FLAG_UTF8: `.`, negated classes and \\W \\D \\S take whole code points, a non-ASCII character
    is one atom, \\x{...} names code points, offsets stay byte offsets; caseless k / s also match
    KELVIN SIGN / LONG S (tools/hscollider/test_cases/pcre/utf8.txt has the reference doing so)"""
    U, SOM, I = hs.HS_FLAG_UTF8, hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_CASELESS
    exprs = [("foo.bar", U, 1), ("café+x", U, 2), ("[^a]end", U | SOM, 3), (r"x\W{2}y", U, 4), (r"€\d+(\.\d\d)?", U, 5),
             (r"naïve|über\b", U, 6), (r".{2}\bzip", U | SOM, 7), (r"q[^\n\d]*ß", U, 8), (r"tag\S+\s", U, 9),
             (r"(é|ab)+c", U | SOM, 10), (r"x[éa€]+y", U, 11), (r"[α-ω]{2,}s", U | SOM, 12), (r"k[^é\d]z", U, 13),
             (r"<[é-ü\w]*>", U, 14)]
    # written with \x{...} instead of raw characters: the same expressions
    hexed = [(p.encode("ascii", "backslashreplace").decode().replace("\\u", "\\x{").replace("\\x{00e9", "\\x{e9}").replace(
        "\\x{20ac", "\\x{20ac}").replace("\\x{00ef", "\\x{ef}").replace("\\x{00fc", "\\x{fc}").replace("\\x{00df", "\\x{df}").replace("\\x{03b1", "\\x{3b1}").replace("\\x{03c9", "\\x{3c9}"), f, i)
        for p, f, i in exprs]
    words = ["foo", "bar", "é", "€", "\U0001f600", "x", "caf", "end", "a", "b", "y", "..", "12", ".50", "naïve", "über",
             " ", "zip", "q", "ß", "tag", "\n", "ab", "c", "_", "-", "αβ", "ω", "s", "k", "z", "<", ">", "ü", "ÿ"]
    rng = np.random.default_rng(101)
    blocks = ["".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 12)))).encode() for _ in range(250)]
    blocks += [s.encode() for s in ["fooébar foo\U0001f600bar fooxxbar", "caféééx cafex", "éend aend", "xé€y x..y xaby",
                                    "€12.50 €x", "naïve über übers", "éézip a-zip abzip", "qabß q1ß qéß",
                                    "tagé€ x", "éabéc abc éc", "xéa€y x€üy", "αβγs ωs aαs", "kéz k5z k€z kaz", "<éü_a> <ÿ> <>"]]
    for variant in (exprs, hexed):
        ev = run_exprs_auto([(p.encode("utf-8"), f, i) for p, f, i in variant], blocks)
        want = brute_utf8(exprs, blocks)
        assert sorted(ev) == sorted(want)
        assert {e[1] for e in ev} == set(range(1, 15))
    # caseless k and s: partners U+212A and U+017F (the model folds them too, without re.ASCII)
    cexprs = [("mask", U | I, 1), ("KS[a-s]x", U | I, 2), ("[^ks]:[ſ]!", U | I, 3)]
    cblocks = [s.encode() for s in ["mask MASK maſk maſK", "ksſx Kſkx KSSX kstx", "a:s! K:s! ſ:S! b:ſ!"]]
    ev = run_exprs_auto([(p.encode(), f, i) for p, f, i in cexprs], cblocks)
    assert sorted(ev) == sorted(brute_utf8(cexprs, cblocks, ascii_classes=False)) and len(ev) == 9
    import pytest
    for bad, fl in [(b"caf\xe9", U), ("é".encode(), U | I), (rb"a[\h]", U), (rb"a[\x{fc}-\x{e9}]", U), (b"x", U | hs.HS_FLAG_UCP)]:
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [fl], [1])


def test_repeats_unrolled_and_small_classes():
    """patterns whose literal is only reachable through a rewrite (rewrite_for_literal in
    csrc/hs_pattern.cpp): a repeat that must run at least once gives up its first turn
    (A+ = A A*, A{n,m} = A A{n-1,m-1}), a class of <= 40 members stands as the alternation of
    its members. Against Python's re with the whole block visible, under four flag settings."""
    import random

    r = random.Random(3)
    pats = ["[pqr]", r"\s", "(foobar)+", "(foo){2,5}", "((foo){2}){3}", "[a-z]{3,7}", r"\d+x", "[ab]+c?", "(a|ab)+", "(ab|b)+a",
            "(?:fo|o)+b", "x?[ab]{2,}", "[a-c]{2}", "(a|ab)(b|)c", "[aA]b+", "(?:ab){1,}c?", "^(ab)+", "(ab)+$", r"\b[ab]+\b",
            "(a[bc]){2,3}d?", "[^\\x00-\\xfd]+", "(fo+){2}", "((a|b)c)+", r"\d{2,3}", "[f-h]+?o", "[]a]", "[.][$]", r"[\]]x*"]
    alphabet = b"abcfoqprxd 01\n\tAB].$\xfe\xff"
    fixed = [b"foofoofoofoofoofoofoo", b"foobarfoobar foobarfoobarfoobar", b"abababab ab", b""]
    n = 0
    for p in pats:
        for fl in (0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_MULTILINE):
            blocks = [bytes(r.choice(alphabet) for _ in range(r.randint(0, 40))) for _ in range(4)] + fixed
            got, want = sorted(run_exprs_auto([(p, fl, 1)], blocks)), sorted(brute_context([(p, fl, 1)], blocks))
            assert got == want, (p, fl, sorted(set(got) ^ set(want))[:4])
            n += len(want)
    assert n > 800
    # the members of a caseless class pair up: 26 literals for [a-zA-Z], not 52
    assert len(hs.Database.compile(["[a-zA-Z]+"], [hs.HS_FLAG_CASELESS], [1]).literals()) == 26
    # what stays outside: classes above 40 members, possessive repeats, repeats that may run zero times
    import pytest
    for bad in [r"\w+", r"[^a]{2}", r"(ab)*", r"[ab]?", r"[ab]++", r"."]:
        with pytest.raises(hs.HsError):
            hs.Database.compile([bad], [0], [1])


def test_many_hits_of_one_pattern_share_one_pass():
    """k hits of `foo` in front of an unbounded tail (`foo.*bar`): the confirm runs the tail automaton
    ONCE over the block with the start state injected at every hit, so a block of 20k hits takes
    milliseconds (it was quadratic: 80 KB took 10 s), and the events are still exactly those of the
    brute-force reading -- every `bar` end after the first `foo`."""
    import time

    for pat, flags in (("foo.*bar", hs.HS_FLAG_DOTALL), ("foo[^x]*ba?r", 0), ("foo(?:a|.)*bar", hs.HS_FLAG_DOTALL)):
        db = hs.Database.compile_ext([pat], [flags], [7], [None])
        n = 20_000
        corpus = np.frombuffer(b"foo bar " * n, dtype=np.uint8).copy()
        off = np.array([0, corpus.size], dtype=np.uint64)
        parts = [(b"foo", "", 0, 7, {})]
        recs = literal_hits(parts, corpus, off)
        assert recs.size == n
        t0 = time.time()
        rv, ev = confirm(db, corpus, off, recs)
        dt = time.time() - t0
        assert rv == hs.HS_SUCCESS
        assert [e[3] for e in ev] == [8 * k + 7 for k in range(n)], pat
        assert all(e[1] == 7 and e[2] == 0 for e in ev)
        assert dt < 5.0, "the confirm is no longer linear in the block length: %.1f s" % dt
    # the short form on a small block, against re, with hits that overlap each other's tails
    db = hs.Database.compile_ext(["ab.{0,5}c", "ab[a-c]*c"], [hs.HS_FLAG_DOTALL, 0], [1, 2], [None, None])
    rng = np.random.default_rng(8)
    corpus = rng.choice(np.frombuffer(b"abc.", dtype=np.uint8), 4000).copy()
    off = np.array([0, 1000, 1003, 4000], dtype=np.uint64)
    parts = [(b"ab", ".{0,5}c", hs.HS_FLAG_DOTALL, 1, {}), (b"ab", "[a-c]*c", 0, 2, {})]
    recs = literal_hits(parts, corpus, off)
    rv, ev = confirm(db, corpus, off, recs)
    assert rv == hs.HS_SUCCESS and sorted(ev) == sorted(brute(parts, corpus, off)) and len(ev) > 300


def test_expressions_sharing_an_id_share_its_exhaustion_key_and_keep_their_som():
    """SINGLEMATCH exhaustion belongs to the report id ("all patterns with the same report id share an ekey",
    src/util/report_manager.cpp:254-257): two SINGLEMATCH expressions with one id deliver it once; a SINGLEMATCH
    and a plain expression sharing an id are refused at compile time with the reference's message
    (registerExtReport, :212-234); a SOM and a non-SOM report at the same offset are both delivered"""
    db = hs.Database.compile_ext(["key", "lock"], [hs.HS_FLAG_SINGLEMATCH, hs.HS_FLAG_SINGLEMATCH], [5, 5], [None, None])
    corpus = np.frombuffer(b"key lock key lock lock", dtype=np.uint8).copy()
    off = np.array([0, corpus.size], dtype=np.uint64)
    parts = [(b"key", "", 0, 5, {}), (b"lock", "", 0, 5, {})]
    rv, ev = confirm(db, corpus, off, literal_hits(parts, corpus, off))
    assert rv == hs.HS_SUCCESS
    assert [(e[1], e[3]) for e in ev] == [(5, 3)]  # id 5 once, at its first match
    db = hs.Database.compile_ext(["xab", "ab"], [hs.HS_FLAG_SINGLEMATCH, hs.HS_FLAG_SINGLEMATCH], [9, 9], [None, None])
    c2 = np.frombuffer(b"..xab..", dtype=np.uint8).copy()
    o2 = np.array([0, c2.size], dtype=np.uint64)
    rv, ev = confirm(db, c2, o2, literal_hits([(b"xab", "", 0, 9, {}), (b"ab", "", 0, 9, {})], c2, o2))
    assert rv == hs.HS_SUCCESS and [(e[1], e[3]) for e in ev] == [(9, 5)]  # one event for (to = 5, id = 9)
    with pytest.raises(hs.HsError) as ei:
        hs.Database.compile_ext(["key", "lock"], [hs.HS_FLAG_SINGLEMATCH, 0], [5, 5], [None, None])
    assert ei.value.expression == 1
    assert "did not specify HS_FLAG_SINGLEMATCH whereas previous expression (index 0) with the same match ID did." in str(ei.value)
    with pytest.raises(hs.HsError) as ei:
        hs.Database.compile_ext(["key", "lock"], [0, hs.HS_FLAG_SINGLEMATCH], [5, 5], [None, None])
    assert "specified HS_FLAG_SINGLEMATCH whereas previous expression (index 0) with the same match ID did not." in str(ei.value)
    db = hs.Database.compile_ext(["xab", "ab"], [hs.HS_FLAG_SOM_LEFTMOST, 0], [9, 9], [None, None])
    corpus = np.frombuffer(b"..xab..", dtype=np.uint8).copy()
    off = np.array([0, corpus.size], dtype=np.uint64)
    parts = [(b"xab", "", 0, 9, {}), (b"ab", "", 0, 9, {})]
    rv, ev = confirm(db, corpus, off, literal_hits(parts, corpus, off))
    assert rv == hs.HS_SUCCESS and sorted((e[2], e[3]) for e in ev) == [(0, 5), (2, 5)]
    # two plain expressions with one id are one report per offset, as in the reference
    db = hs.Database.compile_ext(["xab", "ab"], [0, 0], [9, 9], [None, None])
    rv, ev = confirm(db, corpus, off, literal_hits(parts, corpus, off))
    assert rv == hs.HS_SUCCESS and [(e[2], e[3]) for e in ev] == [(0, 5)]


def test_deserialize_does_not_trust_a_table_that_is_not_its_patterns():
    """a blob whose GPU table section belongs to another database (spliced in, CRC recomputed) still
    loads, but the table is compiled afresh from the sources: the loaded database has its own
    literals, and records with an id outside it are refused"""
    import struct
    import zlib

    small = hs.Database.compile_ext(["alpha"], [0], [1], [None])
    big = hs.Database.compile_ext(["alpha", "bravo", "charlie"], [0, 0, 0], [1, 2, 3], [None, None, None])

    def split(blob):  # -> (head up to and including the table length, table)
        n = struct.unpack_from("<I", blob, 8)[0] & 0x7FFFFFFF
        o = 16
        for _ in range(n):
            ln = struct.unpack_from("<I", blob, o + 12)[0]
            o += 16 + 32 + ln
        tlen = struct.unpack_from("<Q", blob, o)[0]
        assert o + 8 + tlen == len(blob)
        return blob[:o], blob[o + 8:]

    head_s, _tab_s = split(small.serialize())
    _head_b, tab_b = split(big.serialize())
    forged = bytearray(head_s + struct.pack("<Q", len(tab_b)) + tab_b)
    struct.pack_into("<I", forged, 4, zlib.crc32(bytes(forged[8:])) & 0xFFFFFFFF)
    db = hs.Database.deserialize(bytes(forged))
    lib = hs._lib()
    lib.hs_database_literal.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.hs_database_literal(db._h, 0, None, None, None, None) == hs.HS_SUCCESS
    assert lib.hs_database_literal(db._h, 1, None, None, None, None) == hs.HS_INVALID  # one pattern, one literal
    # its table is the one-literal table again: serialising it gives the small database's blob
    assert db.serialize() == small.serialize()
    # a blob of another format version is refused as such
    old = bytearray(small.serialize())
    old[0] = old[0] - 1
    try:
        hs.Database.deserialize(bytes(old))
        assert False, "an older format loaded"
    except hs.HsError as e:
        assert e.code == hs.HS_DB_VERSION_ERROR
