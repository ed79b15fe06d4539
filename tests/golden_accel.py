"""Golden vectors restated from the reference's unit tests for the character-class and
two-byte accelerators (file:line cited per case). Data only, so that the same vectors pin
the C oracle, the compiled reference and the GPU kernels.

A case: (name, kind, params, text, lo_trim, hi_trim, expect_abs)
  the scan covers text[lo_trim : len(text) - hi_trim]; expect_abs is the absolute index in
  `text` the reference test asserts (None = "not found": len for forward kinds, -1 for
  reverse kinds, relative to the scanned slice). Kinds:
    verm(c, nocase)  nverm(c, nocase)  rverm(c, nocase)
    dverm(c1, c2, nocase)  dverm_masked(c1, c2, m1, m2)  rdverm(c1, c2, nocase)
    dshufti(pairs)   : shuftiBuildDoubleMasks(CharReach(), pairs)
    shufti(members) rshufti(members) : shuftiBuildMasks(CharReach of members) + shuftiExec / rshuftiExec
    truffle(members) rtruffle(members): truffleBuildMasks + truffleExec / rtruffleExec
                       (unit/internal/truffle.cpp; most texts are those of shufti.cpp)
"""
CASE_CLEAR = 0xDF


def _b(s):
    return s.encode("latin-1")


def _sweep(name, kind, params, text, expect_abs, lo=True, hi=False, n=16):
    """the tests' `for i in 0..15` loops: trim i bytes in front (lo) and/or at the back (hi)"""
    out = []
    for i in range(n):
        out.append((f"{name}[{i}]", kind, params, _b(text), i if lo else 0, i if hi else 0, expect_abs))
    return out


def cases():
    c = []
    B61 = "b" * 61
    # unit/internal/vermicelli.cpp:57-71 Vermicelli.Exec1
    t = "bbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbbbbbbbbbbbabbbbbbbbbbbb"
    c += _sweep("Verm.Exec1", "verm", (ord("a"), 0), t, 17)
    c += _sweep("Verm.Exec1nc", "verm", (ord("A"), 1), t, 17)
    # :73-87 Exec2
    t = "bbbbbbbbbbbbbbbbbaaaaaaaaaaaaaaaaaaaaaaabbbbbbbbabbbbbbbbbbbb"
    c += _sweep("Verm.Exec2", "verm", (ord("a"), 0), t, 17)
    c += _sweep("Verm.Exec2nc", "verm", (ord("A"), 1), t, 17)
    # :89-103 Exec3
    t = "bbbbbbbbbbbbbbbbbAaaaaaaaaaaaaaaaaaaaaaabbbbbbbbabbbbbbbbbbbb"
    c += _sweep("Verm.Exec3", "verm", (ord("a"), 0), t, 18)
    c += _sweep("Verm.Exec3nc", "verm", (ord("A"), 1), t, 17)
    # :27-55 ExecNoMatch1 (i in front, j at the back)
    for j in range(0, 16, 5):
        for i in range(0, 16, 3):
            for ch, nc in ((ord("a"), 0), (ord("B"), 0), (ord("A"), 1)):
                c.append((f"Verm.NoMatch[{i},{j},{ch},{nc}]", "verm", (ch, nc), _b(B61), i, j, None))
    # :156-180 DoubleVermicelli.Exec1
    t = "bbbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbbbbbbbbbbbabbbbbbbbbbb"
    c += _sweep("DVerm.Exec1ab", "dverm", (ord("a"), ord("b"), 0), t, 18)
    c += _sweep("DVerm.Exec1AB", "dverm", (ord("A"), ord("B"), 1), t, 18)
    c += _sweep("DVerm.Exec1ba", "dverm", (ord("b"), ord("a"), 0), t, 17)
    c += _sweep("DVerm.Exec1BA", "dverm", (ord("B"), ord("A"), 1), t, 17)
    # :182-196 Exec2
    t = "bbbbbbbbbbbbbbbbbaaaaaaaaaaaaaaaaaaaaaaaabbbbbbbaaaaabbbbbbbb"
    c += _sweep("DVerm.Exec2aa", "dverm", (ord("a"), ord("a"), 0), t, 17)
    c += _sweep("DVerm.Exec2AA", "dverm", (ord("A"), ord("A"), 1), t, 17)
    # :198-223 Exec3
    t = "bbbbbbbbbbbbbbbbbaAaaAAaaaaaaaaaaaaaaaaaabbbbbbbaaaaabbbbbbbb"
    c += _sweep("DVerm.Exec3Aa", "dverm", (ord("A"), ord("a"), 0), t, 18)
    c += _sweep("DVerm.Exec3AAnc", "dverm", (ord("A"), ord("A"), 1), t, 17)
    c += _sweep("DVerm.Exec3AA", "dverm", (ord("A"), ord("A"), 0), t, 21)
    c += _sweep("DVerm.Exec3aA", "dverm", (ord("a"), ord("A"), 0), t, 17)
    # :112-154 DoubleVermicelli.ExecNoMatch1 incl. the partial match at the end
    for j in range(0, 16, 5):
        for i in range(0, 16, 3):
            n = len(B61)
            c.append((f"DVerm.NoMatch[{i},{j}]", "dverm", (ord("a"), ord("b"), 0), _b(B61), i, j, None))
            c.append((f"DVerm.NoMatchBb[{i},{j}]", "dverm", (ord("B"), ord("b"), 0), _b(B61), i, j, None))
            c.append((f"DVerm.Partial[{i},{j}]", "dverm", (ord("b"), ord("B"), 0), _b(B61), i, j, n - j - 1))
            c.append((f"DVerm.PartialNc[{i},{j}]", "dverm", (ord("B"), ord("A"), 1), _b(B61), i, j, n - j - 1))
    # :388-429 DoubleVermicelliMasked.Exec1 (front AND back trimmed by i)
    t = "bbbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbbbbbbbbbbbabbbbbbbbbbb"
    for i in range(16):
        for nm, p, e in (("ab", (ord("a"), ord("b"), 0xFF, 0xFF), 18), ("AB", (ord("A"), ord("B"), CASE_CLEAR, CASE_CLEAR), 18),
                         ("aB", (ord("a"), ord("B"), 0xFF, CASE_CLEAR), 18), ("Ab", (ord("A"), ord("b"), CASE_CLEAR, 0xFF), 18),
                         ("ba", (ord("b"), ord("a"), 0xFF, 0xFF), 17), ("BA", (ord("B"), ord("A"), CASE_CLEAR, CASE_CLEAR), 17)):
            c.append((f"DVermMasked.Exec1{nm}[{i}]", "dverm_masked", p, _b(t), i, i, e))
    # unit/internal/rvermicelli.cpp:56-69 RVermicelli.Exec1
    t = "bbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbb"
    c += _sweep("RVerm.Exec1", "rverm", (ord("a"), 0), t, 48, lo=False, hi=True)
    c += _sweep("RVerm.Exec1nc", "rverm", (ord("A"), 1), t, 48)
    # :86-99 Exec3
    t = "bbbbbbbbbbbbbbbbbabbbbbbbbaaaaaaaaaaaaaaaaaaaaaaAbbbbbbbbbbbbbbbbbbbbbb"
    c += _sweep("RVerm.Exec3", "rverm", (ord("a"), 0), t, 47, lo=False, hi=True)
    c += _sweep("RVerm.Exec3nc", "rverm", (ord("A"), 1), t, 48, lo=False, hi=True)
    # :36-54 ExecNoMatch1
    for j in range(0, 16, 5):
        for i in range(0, 16, 3):
            c.append((f"RVerm.NoMatch[{i},{j}]", "rverm", (ord("a"), 0), _b(B61), i, j, None))
    # :116-140 RDoubleVermicelli.Exec1: the SECOND byte of the last pair
    t = "bbbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbb"
    c += _sweep("RDVerm.Exec1ab", "rdverm", (ord("a"), ord("b"), 0), t, 50, lo=False, hi=True)
    c += _sweep("RDVerm.Exec1AB", "rdverm", (ord("A"), ord("B"), 1), t, 50)
    c += _sweep("RDVerm.Exec1ba", "rdverm", (ord("b"), ord("a"), 0), t, 49)
    c += _sweep("RDVerm.Exec1BA", "rdverm", (ord("B"), ord("A"), 1), t, 49)
    # :142-156 Exec2
    t = "bbbbbbbbbbbbbbbbbaaaaaaaaaaaaaaaaaaaaaaaabbbbbbbaaaaabbbbbbbbbbbbbbbbbb"
    c += _sweep("RDVerm.Exec2aa", "rdverm", (ord("a"), ord("a"), 0), t, 52, lo=False, hi=True)
    c += _sweep("RDVerm.Exec2AA", "rdverm", (ord("A"), ord("A"), 1), t, 52, lo=False, hi=True)
    # unit/internal/shufti.cpp:604-629 DoubleShufti.ExecMatchShort1 / :631-650 ExecMatch1
    P = lambda *ps: tuple((ord(a), ord(b)) for a, b in ps)
    c += _sweep("DShufti.MatchShort1", "dshufti", P("ab"), "bbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbb", 17)
    c += _sweep("DShufti.Match1", "dshufti", P("ab"),
                "bbbbbbbbbbbbbbbbbabbbbbbbbbbbbbbbbbbbbbbbbbbbbbbabbbbbbbbbbbb", 17)
    # :652-671 ExecMatch2, :673-693 ExecMatch3
    c += _sweep("DShufti.Match2", "dshufti", P("aa"),
                "bbbbbbbbbbbbbbbbbaaaaaaaaaaaaaaaabbbbbbbbbbbbbbbabbbbbbbbbbbb", 17)
    c += _sweep("DShufti.Match3", "dshufti", P("Ba", "aa"),
                "bbbbbbbbbbbbbbbbbBaaaaaaaaaaaaaaaabbbbbbbbbbbbbbbabbbbbbbbbbbb", 17)
    # :695-736 ExecMatch4, :738-779 ExecMatch4b
    for k, first in enumerate("ACca"):
        c += _sweep(f"DShufti.Match4[{k}]", "dshufti", P("Aa", "aa", "Ca", "ca"),
                    "bbbbbbbbbbbbbbbbb" + first + "aaaaaaaaaaaaaaabbbbbbbbbbbbbbbabbbbbbbbbbbb", 17)
    for k, second in enumerate("ACca"):
        c += _sweep(f"DShufti.Match4b[{k}]", "dshufti", P("aA", "aa", "aC", "ac"),
                    "bbbbbbbbbbbbbbbbba" + second + "aaaaaaaaaaaaaabbbbbbbbbbbbbbbabbbbbbbbbbbb", 17)
    # unit/internal/shufti.cpp:159-175 Shufti.ExecMatch1 (32 start offsets), :177-193 ExecMatch2,
    # :195-212 ExecMatch3, :214-258 ExecMatch4, :110-157 ExecNoMatch1-3 (their bound
    # `rv >= end & ~15` is met by "not found" = end)
    M = lambda s_: tuple(sorted(s_.encode("latin-1")))
    c += _sweep("shufti.Match1", "shufti", M("a"), "b" * 33 + "a" + "b" * 14 + "a" + "b" * 12, 33, n=32)
    # unit/internal/truffle.cpp:230-247 Truffle.ExecMatch1; :249-323 ExecMatch2-4 and :94-149
    # ExecNoMatch1-3 use the texts of their shufti namesakes
    c += _sweep("truffle.Match1", "truffle", M("a"), "b" * 17 + "a" + "b" * 30 + "a" + "b" * 12, 17)
    for kind in ("shufti", "truffle"):
        c += _sweep(f"{kind}.Match2", kind, M("a"), "b" * 17 + "a" * 16 + "b" * 15 + "a" + "b" * 12, 17)
        c += _sweep(f"{kind}.Match3", kind, M("aB"), "b" * 17 + "B" + "a" * 15 + "b" * 15 + "a" + "b" * 12, 17)
        for k, first in enumerate("ACca"):
            c += _sweep(f"{kind}.Match4[{k}]", kind, M("aCAc"), "b" * 17 + first + "a" * 15 + "b" * 15 + "a" + "b" * 12, 17)
        c += _sweep(f"{kind}.NoMatch1", kind, M("a"), "b" * 61, None)
        c += _sweep(f"{kind}.NoMatch2", kind, M("aB"), "b" * 61, None)
        c += _sweep(f"{kind}.NoMatch3", kind, M("V"), "e" * 61, None)
    # truffle.cpp:151-209 ExecMiniMatch0-3: buffers shorter than one vector
    c.append(("truffle.Mini0", "truffle", M("a"), b"a", 0, 0, 0))
    c.append(("truffle.Mini1", "truffle", M("a"), b"bbbbbbbabbb", 0, 0, 7))
    c.append(("truffle.Mini2", "truffle", (0,), b"bbbbbbb\0bbb", 0, 0, 7))
    c.append(("truffle.Mini3", "truffle", M("a"), b"\0" * 7 + b"a" + b"\0" * 3, 0, 0, 7))
    # shufti.cpp:962-981 ReverseShufti.ExecMatch1 (the end moves back), :983-1002 ExecMatch2, :1004-1034
    # ExecMatch3, :1036-1072 ExecMatch4; ReverseTruffle.ExecMatch1-4 (truffle.cpp:494-600) use the same texts
    for kind in ("rshufti", "rtruffle"):
        c += _sweep(f"{kind}.Match1", kind, M("a"), "bbbbbbabbbbbbbbbba" + "b" * 43, 17, lo=False, hi=True)
        c += _sweep(f"{kind}.Match2", kind, M("a"), "bbbbabbbbbbbbbbbb" + "a" * 16 + "b" * 28, 32, lo=False, hi=True)
        c += _sweep(f"{kind}.Match3", kind, M("aB"), "b" * 17 + "a" * 15 + "B" + "b" * 28, 32, lo=False, hi=True)
        c += _sweep(f"{kind}.Match4", kind, M("aCAc"), "b" * 17 + "a" * 15 + "A" + "b" * 28, 32, lo=False, hi=True)
        c += _sweep(f"{kind}.NoMatch", kind, M("a"), "b" * 61, None, lo=False, hi=True)
    return c


# DoubleShufti cases whose expectation in the reference is an artefact of its 16-byte vectors
# (unit/internal/shufti.cpp:545-564 ExecNoMatch2b, :587-602 ExecNoMatch3b: expects start + 15):
# (pairs, text) -- the tests check the vector model and the "never later than exact" property.
DSHUFTI_EDGE = [
    ((("b", "a"), ("b", "B")), "b" * 61),
    ((("e", "V"),), "e" * 61),
]
# and the ones that only bound the result from below (:523-543 ExecNoMatch2, :566-585 ExecNoMatch3)
DSHUFTI_NOMATCH = [
    ((("a", "b"), ("B", "b")), "b" * 61),
    ((("V", "e"),), "e" * 61),
]
