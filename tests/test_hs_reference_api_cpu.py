"""The reference's API-level unit tests (unit/hyperscan/*.cpp) that lie inside the supported
pattern subset, on the CPU: the database is compiled by the facade, the literal hits come from
the HWLM oracle for the literals the database reports it is keyed on (hs_database_literal), and
the host confirm (hs_confirm_batch) turns them into events. The same cases run through the GPU
in tests/test_gpu_hs_api.py / test_zz_gpu_late_additions.py; here every expected (to, id) list
is the reference's own."""
import ctypes as C

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import hs
from hyperscan_amd.hwlm import MATCH_DTYPE
from tests import oracle_binding as ob


def cpu_scan(db, data, halt_after=None):
    """hs_scan with the oracle in the GPU's place -> [(to, id)] in delivery order, return code"""
    lits = [H.HwlmLiteral(b, nocase=nc, id=i) for i, (b, nc, _rid) in enumerate(db.literals())]
    corpus = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(0, np.uint8)
    off = np.array([0, corpus.size], dtype=np.uint64)
    got = ob.Oracle(lits).collect_blocks(corpus, off)
    recs = np.zeros(len(got), dtype=MATCH_DTYPE)
    recs["block"], recs["end"], recs["id"], recs["lit"] = got["block"], got["end"], got["id"], got["id"]
    recs = np.ascontiguousarray(recs[np.lexsort((recs["id"], recs["end"], recs["block"]))])
    lib = hs._lib()
    lib.hs_confirm_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, hs.BATCH_CB,
                                     C.c_void_p]
    ev = []

    def on(_b, i, f, t, _fl, _c):
        ev.append((t, i, f))
        return 1 if halt_after is not None and len(ev) >= halt_after else 0

    cb = hs.BATCH_CB(on)
    buf = corpus if corpus.size else np.zeros(1, np.uint8)
    rv = lib.hs_confirm_batch(db._h, buf.ctypes.data, off.ctypes.data, 1, recs.ctypes.data, recs.size, cb, None)
    return ev, rv


def to_id(ev):
    return [(t, i) for t, i, _f in ev]


def test_multi_cpp():
    # unit/hyperscan/multi.cpp:40-290 (MMAdaptor): two patterns, with and without SINGLEMATCH / halting
    exprs, ids = ["aoo[A-K]", "bar[L-Z]"], [30, 31]
    for data in (b"aooAaooAbarZ", b"aooAaooAbarZ" + b" " * 22):
        db = hs.Database.compile(exprs, [0, 0], ids)
        assert to_id(cpu_scan(db, data)[0]) == [(4, 30), (8, 30), (12, 31)]            # norm_cont
        assert to_id(cpu_scan(db, data, halt_after=1)[0]) == [(4, 30)]                 # norm_halt
        db = hs.Database.compile(exprs, [hs.HS_FLAG_SINGLEMATCH, 0], ids)
        assert sorted(to_id(cpu_scan(db, data)[0])) == [(4, 30), (12, 31)]             # high_cont
        assert to_id(cpu_scan(db, data, halt_after=1)[0]) == [(4, 30)]                 # high_halt
    # :331-366 MMRoseLiteralPath.issue_141: ids default to 0
    S = hs.HS_FLAG_DOTALL | hs.HS_FLAG_SINGLEMATCH
    db = hs.Database.compile(["/odezhda-dlya-bega/", "kurtki-i-vetrovki-dlya-bega", "futbolki-i-mayki-dlya-bega"], [S] * 3, [0] * 3)
    assert to_id(cpu_scan(db, b"/odezhda-dlya-bega/")[0]) == [(19, 0)]


def test_multi_cpp_dot_runs():
    # unit/hyperscan/multi.cpp:294-331: "aaa" (id 3) fires at every offset 3..300 of 300 x 'a'
    # (ids 1 and 2, "^.{200}" and ".{40,}", have no literal: outside the subset)
    db = hs.Database.compile(["aaa"], [hs.HS_FLAG_DOTALL], [3])
    ev, _ = cpu_scan(db, b"a" * 300)
    assert to_id(ev) == [(t, 3) for t in range(3, 301)]
    for pat in ("^.{200}", ".{40,}"):
        with pytest.raises(hs.HsError):
            hs.Database.compile([pat], [hs.HS_FLAG_DOTALL], [1])


def test_order_cpp():
    # unit/hyperscan/order.cpp:65-~330: per-id counts over 32 x 'a', matches in offset order
    D = hs.HS_FLAG_DOTALL
    pats = {1: "aa", 2: "aa.", 3: "aa..", 4: "^.{0,4}aa..", 5: "^.{0,4}aa"}
    expect = {1: 31, 2: 30, 3: 29, 4: 5, 5: 5}
    for subset in ([1, 2, 3, 4, 5], [2, 3, 4, 5], [1, 3, 4, 5], [1, 2, 4, 5], [1, 2, 3, 5], [1, 2, 3, 4]):  # ordering1..6
        db = hs.Database.compile([pats[i] for i in subset], [D] * len(subset), subset)
        ev, rv = cpu_scan(db, b"a" * 32)
        assert rv == hs.HS_SUCCESS
        for i in range(1, 6):
            assert sum(1 for e in ev if e[1] == i) == (expect[i] if i in subset else 0)
        assert [e[0] for e in ev] == sorted(e[0] for e in ev)


def test_behaviour_cpp_cases():
    I, ONE, SOM = hs.HS_FLAG_CASELESS, hs.HS_FLAG_SINGLEMATCH, hs.HS_FLAG_SOM_LEFTMOST
    # behaviour.cpp:361-389 BlockThereCanBeOnlyOne
    data = (b"hackhackHACKhackHACkhAcKzofunvuynlslijlnikshvb ar yhtkubq45ytvb iuyh "
            b"ackeruniou viytdfjhg nvldkrjgnal")
    db = hs.Database.compile([".ck"], [I | ONE], [0])
    assert len(cpu_scan(db, data)[0]) == 1
    db = hs.Database.compile([".ck"], [I], [0])
    assert len(cpu_scan(db, data)[0]) == 7
    # behaviour.cpp:1426-1465 UE_2763 (the scan is one chunk: block mode owes the same matches;
    # HS_FLAG_UTF8 dropped from pattern 2, the data is ASCII)
    db = hs.Database.compile(["aaa.a+$", "aaa.a"], [I, I | ONE], [1, 2])
    assert sorted(to_id(cpu_scan(db, b"aAasaA")[0])) == [(5, 2), (6, 1)]
    # behaviour.cpp:1467-1511 UE_2798 (pattern 1's literals come out of its group: "[ab]b$|aab+$")
    db = hs.Database.compile(["([ab]b|aab+)$", "ab+", "a(b.)?ba+b"], [hs.HS_FLAG_DOTALL, SOM, 0], [1, 2, 3])
    ev, _ = cpu_scan(db, b"ab_baab\n")
    assert sorted(to_id(ev)) == [(2, 2), (7, 1), (7, 2), (7, 3)]
    assert {(t, f) for t, i, f in ev if i == 2} == {(2, 0), (7, 5)}
    # behaviour.cpp:400-476 HyperscanLiteralLengthTest: floating and anchored literals of every length
    for n in (1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 32, 33, 100, 255):
        lit = "".join(chr(ord("a") + (k % 26)) for k in range(n))
        for pat, data, want in ((lit, b"x" * 50 + lit.encode() + b"yy", [(50 + n, 0)]),
                                ("^" + lit, lit.encode() + b"yy" + lit.encode(), [(n, 0)])):
            db = hs.Database.compile([pat], [0], [0])
            assert to_id(cpu_scan(db, data)[0]) == want, (n, pat)


def test_callback_return_stop():
    # behaviour.cpp:490-518 CallbackReturnStop.Block: a non-zero return ends the scan after one match
    for pat, fl in (("foo", 0), ("foo.*bar", hs.HS_FLAG_DOTALL), ("fo+", hs.HS_FLAG_SOM_LEFTMOST)):
        db = hs.Database.compile([pat], [fl], [0])
        ev, rv = cpu_scan(db, b"foo bar foo barfoooo" * 4, halt_after=1)
        assert len(ev) == 1


def test_identical_cpp():
    # unit/hyperscan/identical.cpp:46-82,160-188: 100 copies of one pattern under ids 0..99: one
    # match per id, all at the table's offset (every row of the table)
    ONE, SOM = hs.HS_FLAG_SINGLEMATCH, hs.HS_FLAG_SOM_LEFTMOST
    rows = [("a", 0, b"a", 1), ("a", ONE, b"a", 1), ("handbasket", 0, b"__handbasket__", 12),
            ("handbasket", ONE, b"__handbasket__", 12), ("handbasket", SOM, b"__handbasket__", 12),
            ("foo.*bar", 0, b"a foolish embarrassment", 15), ("foo.*bar", ONE, b"a foolish embarrassment", 15),
            ("foo.*bar", SOM, b"a foolish embarrassment", 15),
            ("\\bword\\b(..)+\\d{3,7}", 0, b"    word    012", 15), ("\\bword\\b(..)+\\d{3,7}", ONE, b"    word    012", 15),
            ("\\bword\\b(..)+\\d{3,7}", SOM, b"    word    012", 15),
            ("eod\\z", 0, b"eod", 3), ("eod\\z", ONE, b"eod", 3), ("eod\\z", SOM, b"eod", 3)]
    for pat, fl, corpus, match in rows:
        db = hs.Database.compile([pat] * 100, [fl] * 100, list(range(100)))
        ev, rv = cpu_scan(db, corpus)
        assert rv == hs.HS_SUCCESS and len(ev) == 100 and all(t == match for t, _i, _f in ev), (pat, fl)
        assert sorted(i for _t, i, _f in ev) == list(range(100))


def test_quiet_prefilter_allowempty_flags():
    """HS_FLAG_QUIET reports nothing (src/hs_compile.h:328-330); HS_FLAG_PREFILTER may over-report,
    and the exact matches are a legal answer; HS_FLAG_ALLOWEMPTY has nothing to allow when every
    pattern holds a mandatory literal. The reference's combination rules still apply
    (src/compiler/compiler.cpp:286-294)."""
    db = hs.Database.compile(["foo", "bar", "ba[rz]"], [hs.HS_FLAG_QUIET, hs.HS_FLAG_PREFILTER, hs.HS_FLAG_ALLOWEMPTY], [1, 2, 3])
    assert to_id(cpu_scan(db, b"foo bar baz")[0]) == [(7, 2), (7, 3), (11, 3)]
    for fl in (hs.HS_FLAG_UCP, hs.HS_FLAG_UTF8 | hs.HS_FLAG_UCP, hs.HS_FLAG_QUIET | hs.HS_FLAG_SOM_LEFTMOST,
               hs.HS_FLAG_PREFILTER | hs.HS_FLAG_SOM_LEFTMOST):
        with pytest.raises(hs.HsError):
            hs.Database.compile(["foo"], [fl], [1])


def test_bad_patterns_txt_are_all_refused():
    """unit/hyperscan/bad_patterns.txt (tests/golden/bad_patterns.json): every pattern the reference
    refuses to compile is refused here too; where the reason is one this engine reasons about the
    same way (ext parameters against the pattern's widths, patterns that can never match), with
    the reference's message."""
    import json
    import os

    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bad_patterns.json")))["rows"]
    assert len(rows) >= 150
    letters = {"i": hs.HS_FLAG_CASELESS, "s": hs.HS_FLAG_DOTALL, "m": hs.HS_FLAG_MULTILINE, "H": hs.HS_FLAG_SINGLEMATCH,
               "L": hs.HS_FLAG_SOM_LEFTMOST, "V": hs.HS_FLAG_ALLOWEMPTY, "8": hs.HS_FLAG_UTF8, "W": hs.HS_FLAG_UCP,
               "P": hs.HS_FLAG_PREFILTER, "C": hs.HS_FLAG_COMBINATION, "Q": hs.HS_FLAG_QUIET, "O": 0}
    same_message = 0
    for r in rows:
        flags = 0
        for ch in r["flags"]:
            flags |= letters[ch]
        ext = None
        if r["ext"]:
            known = {k: v for k, v in r["ext"].items() if k in ("min_offset", "max_offset", "min_length")}
            if len(known) != len(r["ext"]):
                continue  # edit_distance / hamming_distance rows: refused wholesale, nothing to compare
            ext = hs.ExprExt.make(**known)
        with pytest.raises(hs.HsError) as e:
            hs.Database.compile_ext([bytes.fromhex(r["pattern_hex"])], [flags], [r["id"]], [ext])
        assert e.value.code == hs.HS_COMPILER_ERROR
        if r["message"].startswith(("Expression has m", "Expression is anchored", "Pattern can never match")):
            same_message += e.value.message == r["message"]
    assert same_message >= 12, same_message


def test_extparam_cpp():
    # unit/hyperscan/extparam.cpp:40-160: large min_offset, exact offset window, large min_length
    def scan(ext, corpus):
        db = hs.Database.compile_ext(["hatstand.*teakettle"], [0], [0], [hs.ExprExt.make(**ext)])
        return to_id(cpu_scan(db, corpus)[0])

    u = lambda n: b"_" * n  # noqa: E731
    assert scan(dict(min_offset=100000), b"hatstand" + u(80000) + b"teakettle") == []                      # LargeMinOffset
    assert scan(dict(min_offset=100000), b"hatstand" + u(99983) + b"teakettle") == [(100000, 0)]
    exact = dict(min_offset=200000, max_offset=200000)                                                         # LargeExactOffset
    assert scan(exact, b"hatstand" + u(199982) + b"teakettle") == []
    assert scan(exact, b"hatstand" + u(199983) + b"teakettle") == [(200000, 0)]
    assert scan(exact, b"hatstand" + u(199984) + b"teakettle") == []
    assert scan(dict(min_length=100000), u(10000) + b"hatstand" + u(80000) + b"teakettle") == []            # LargeMinLength
    assert scan(dict(min_length=100000), u(10000) + b"hatstand" + u(99983) + b"teakettle") == [(110000, 0)]


def test_serialised_database_carries_the_gpu_table():
    """hs_serialize_database = sources + the GPU literal table image: loading takes
    the table from its section instead of compiling the literals again, and the result scans alike"""
    import struct
    import zlib

    pats = ["alpha\\d+", "bet(a|o)x", "\\bgamma\\b|delta$"]
    db = hs.Database.compile(pats, [0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_SINGLEMATCH], [7, 8, 9])
    blob = db.serialize()
    lits = [H.HwlmLiteral(b, nocase=nc, id=i) for i, (b, nc, _r) in enumerate(db.literals())]
    image = H.hwlm_build(lits).serialize()
    at = blob.find(image)
    assert at > 0 and struct.unpack_from("<Q", blob, at - 8)[0] == len(image) and at + len(image) == len(blob)
    db2 = hs.Database.deserialize(blob)
    data = b"alpha12 BETAX betox gamma gammas delta\n delta"
    assert cpu_scan(db2, data) == cpu_scan(db, data) and len(cpu_scan(db, data)[0]) == 5
    assert db2.literals() == db.literals() and db2.serialize() == blob
    # a damaged table section with a CRC made to fit: the table's own validation refuses it
    bad = bytearray(blob)
    bad[at + 40] ^= 0xFF  # a geometry field of the table header
    struct.pack_into("<I", bad, 4, zlib.crc32(bytes(bad[8:])) & 0xFFFFFFFF)
    with pytest.raises(hs.HsError):
        hs.Database.deserialize(bytes(bad))
    # a table section written by another table version (the version word of its header) is not an error: the
    # literals are compiled again from the sources the blob carries, and the database serialises as this library's
    other = bytearray(blob)
    ver = struct.unpack_from("<I", other, at + 4)[0]
    struct.pack_into("<I", other, at + 4, ver + 1)
    struct.pack_into("<I", other, 4, zlib.crc32(bytes(other[8:])) & 0xFFFFFFFF)
    db3 = hs.Database.deserialize(bytes(other))
    assert cpu_scan(db3, data) == cpu_scan(db, data) and db3.serialize() == blob


def test_singlematch_with_branches_in_other_expressions():
    # the exhaustion set is keyed by report id over all branches, wherever their expression sits
    ONE = hs.HS_FLAG_SINGLEMATCH
    db = hs.Database.compile(["cat|dog", "emu", "fox|gnu|ibis"], [0, ONE, ONE], [1, 2, 3])
    ev = to_id(cpu_scan(db, b"cat emu dog emu fox gnu ibis emu cat")[0])
    assert ev == [(3, 1), (7, 2), (11, 1), (19, 3), (36, 1)]


def test_literal_api_random_sets_on_cpu():
    """hs_compile_lit_multi through the host confirm: binary literals of 1..40 bytes (NULs and
    bytes >= 0x80 included), caseless ones, planted in random data; the long-literal check must
    leave exactly the naive search's matches (unit/hyperscan/literals.cpp's shape on the CPU)"""
    rng = np.random.default_rng(17)
    for trial in range(6):
        n = int(rng.integers(1, 60))
        lits, flags = [], []
        for _ in range(n):
            ln = int(rng.integers(1, 41))
            alpha = rng.choice([4, 26, 256])
            b = bytes(rng.integers(0, alpha, ln, dtype=np.uint8) + (97 if alpha < 256 else 0) & 0xFF)
            lits.append(b)
            flags.append(hs.HS_FLAG_CASELESS if rng.random() < 0.4 else 0)
        keep = {}
        for b, f in zip(lits, flags):
            keep.setdefault(b, f)
        lits, flags = list(keep), list(keep.values())
        db = hs.Database.compile_lit(lits, flags, list(range(len(lits))))
        parts = []
        for _ in range(300):
            if rng.random() < 0.5:
                b = lits[int(rng.integers(0, len(lits)))]
                parts.append(b.upper() if rng.random() < 0.3 else b)
            else:
                parts.append(bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)))
        data = b"".join(parts)
        want = set()
        for i, (b, f) in enumerate(zip(lits, flags)):
            up = lambda x: bytes(c - 32 if 97 <= c <= 122 else c for c in x)  # noqa: E731 (ASCII letters only)
            hay, needle = (up(data), up(b)) if f else (data, b)
            k = hay.find(needle)
            while k >= 0:
                want.add((k + len(b), i))
                k = hay.find(needle, k + 1)
        got = to_id(cpu_scan(db, data)[0])
        assert set(got) == want and len(got) == len(want), trial
        assert [t for t, _i in got] == sorted(t for t, _i in got)


def test_quoted_sequences():
    # \Q...\E: everything between is literal (also inside a class; an unclosed \Q runs to the end)
    db = hs.Database.compile([r"a\Q.*+?(x)[y]\Eb+", r"[\Q^]\E-]{2}end", r"tail\Qopen"], [0, 0, 0], [1, 2, 3])
    assert [b for b, _n, _i in db.literals()] == [b"+?(x)[y]", b"end", b"tailopen"]
    assert to_id(cpu_scan(db, b"a.*+?(x)[y]bb  ^]end -^end tailopen")[0]) == [(12, 1), (13, 1), (20, 2), (26, 2), (35, 3)]


# unit/hyperscan/single.cpp:320-345 (the HyperscanTestRuntime / Compile parameter list) and :610-627
# (TerminateMatchData). True = inside the pattern subset; the rest have no literal and no small
# class to stand in for one, and must be refused cleanly.
SINGLE_CPP_PATTERNS = [
    ("foobar", True), ("abd.*def", True), ("abc[123]def", True), ("[pqr]", True), (".", False), ("\\s", True),
    ("hatstand.*(teakettle|badgerbrush)", True), ("abc{1,3}", True), ("abc", True), ("^.{1,10}flibble", True),
    ("(foobar)+", True), ("(foo){2,5}", True), ("((foo){2}){3}", True), ("^.*test[a-f]{3}pattern.*$", True),
    ("(([^u]|.){16}|x){1,2}", False),
    ("fooa?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?", True),
    ("fooa?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?a?", True),
    ("((aaa|[aaa]aa|aa[^aaaa]a|a.aaa.|a)){27}", True),
    ("^.{0,40}((e{0,9}|b){16}a.A|[da]..ecbcbcc[^e]de{10,20}[Bb]{3}[dEe]*){2}", True),
]
SINGLE_CPP_TERMINATE = [
    ("foobar", 0, b"foobarfoobarfoobar"), ("a", 0, b"a a a a a a a a a a a"), (".", 0, b"zzzzzzzzzzzzaaaaaaaaaaaaa"),
    ("...", 0, b"zzzzzzzzzzzzaaaaaaaaaaaaa"), ("[a-z]{3,7}", 0, b"zzzzzzzzzzzzaaaaaaaaaaaaa"),
    ("a", hs.HS_FLAG_CASELESS, b"   xAaAa"), ("xyzzy", hs.HS_FLAG_CASELESS, b"abcdef XYZZy xyzzy XyZzY"),
    ("abc.*def", 0, b"abc   abc   abc   def def"), ("abc.*def", hs.HS_FLAG_DOTALL, b"abc   abc   abc   def def"),
    ("(01234|abcde).*(foo|bar)", 0, b"abcde  xxxx   bar foo abcde foo"),
    ("(01234|abcde).*(foo|bar)", hs.HS_FLAG_DOTALL, b"abcde  xxxx   bar foo abcde foo"),
    ("[0-9a-f]{4,10}.*(foobar|bazbaz)", 0, b"0123456789abcdef  bazbaz foobar"),
    ("[0-9a-f]{4,10}.*(foobar|bazbaz)", hs.HS_FLAG_DOTALL, b"0123456789abcdef  bazbaz foobar"),
    ("^foobar[^z]{20,}", 0, b"foobarxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxx"),
    ("hatstand|teakettle|badgerbrush|mnemosyne", 0, b"hatstand teakettle badgerbrush mnemosyne"),
    ("a|b|c|d", 0, b"a b c d a b c d a b c d"),
    ("bat|cat|mat|rat|fat|sat|pat|hat|vat", hs.HS_FLAG_CASELESS, b"VAt hat pat sat fat rat mat caT BAT"),
]


def _re_ends(pat, fl, data):
    import re

    rf = (re.I if fl & hs.HS_FLAG_CASELESS else 0) | (re.S if fl & hs.HS_FLAG_DOTALL else 0)
    rx = re.compile(pat.encode(), rf)
    return [to for to in range(1, len(data) + 1) if any(rx.fullmatch(data, f, to) for f in range(to))]


def test_single_cpp_patterns_compile_and_scan():
    """single.cpp HyperscanTestRuntime: every pattern compiles to a non-empty database whose info
    starts with "Version:", hs_stream_size says HS_DB_MODE_ERROR for a block database
    (:59-84), and 2048 bytes of 'X' scan cleanly (:117-160). 17 of the 19 patterns are inside
    the subset (repeats unrolled to reach their literal, small classes as alternations)."""
    for flags in (0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL, hs.HS_FLAG_SINGLEMATCH):
        for pat, inside in SINGLE_CPP_PATTERNS:
            if not inside:
                with pytest.raises(hs.HsError) as e:
                    hs.Database.compile([pat], [flags], [0])
                assert e.value.code == hs.HS_COMPILER_ERROR
                continue
            db = hs.Database.compile([pat], [flags], [0])
            assert db.size() > 0 and db.info().startswith("Version:") and db.stream_size() == hs.HS_DB_MODE_ERROR
            ev, rv = cpu_scan(db, b"X" * 2048)
            assert rv == hs.HS_SUCCESS and ev == []


def test_single_cpp_match_terminate():
    """single.cpp HyperscanTestMatchTerminate.MoreThanOne / .Block (:520-552): more than one match
    with a handler that continues; exactly one and HS_SCAN_TERMINATED with one that stops. On top
    of the reference's assertions, the end offsets are compared with Python's re."""
    checked = 0
    for pat, fl, corpus in SINGLE_CPP_TERMINATE:
        if pat in (".", "..."):  # no literal anywhere: outside the subset
            with pytest.raises(hs.HsError):
                hs.Database.compile([pat], [fl], [0])
            continue
        db = hs.Database.compile([pat], [fl], [0])
        ev, rv = cpu_scan(db, corpus)
        assert rv == hs.HS_SUCCESS and len(ev) > 1, pat
        assert [t for t, _i, _f in ev] == _re_ends(pat, fl, corpus), pat
        ev1, rv1 = cpu_scan(db, corpus, halt_after=1)
        assert rv1 == hs.HS_SCAN_TERMINATED and len(ev1) == 1 and ev1[0] == ev[0], pat
        checked += 1
    assert checked == 15


def test_literal_less_patterns_are_refused_quickly():
    """the rewrites that look for a literal (repeat unrolling, class expansion) work to a budget:
    a pattern made of many literal-less repeats is refused in milliseconds, not after trying
    every order of unrolling them"""
    import time

    hs.Database.compile(["warm"])
    for pat in ["(.)+" * 6, "(.)+" * 20, "(\\w|.){2}" * 10, "([^a]x?)+" * 15, "(.)+" * 40 + "|" + "(.)+" * 40]:
        t = time.time()
        with pytest.raises(hs.HsError) as e:
            hs.Database.compile([pat])
        assert e.value.code == hs.HS_COMPILER_ERROR and time.time() - t < 5.0, pat
    for pat in ["[ab]+" * 30, "(a|b)+" * 30]:
        t = time.time()
        hs.Database.compile([pat])
        assert time.time() - t < 5.0, pat
