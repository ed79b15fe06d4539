"""Golden vectors restated from the reference's own unit tests for the HWLM path
(file:line cited per case). Each case is data only: literals, buffer, expected
matches -- so the same vectors pin the C oracle, the compiled reference and the
GPU engine.

A case is a dict:
  name      str
  lits      [HwlmLiteral]
  buf       bytes
  expect    sorted [(end, id)]   (exact delivery list when `ordered` is set)
  ordered   bool: `expect` is the exact callback sequence (noruns cases)
"""
import itertools

import hyperscan_amd as H

L = H.HwlmLiteral

ALPHA = b"mnopqrabcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ12345678901234567890"


def simple_cases():
    out = []
    # unit/internal/fdr.cpp:167-190 FDRp.Simple (buffer includes the trailing NUL: sizeof(data))
    out.append(dict(name="Simple", lits=[L("mnopqr", False, 0)], buf=ALPHA + b"mnopqr\0",
                    expect=[(5, 0), (23, 0), (83, 0)], ordered=True))
    # :192-216 FDRp.SimpleSingle
    out.append(dict(name="SimpleSingle", lits=[L("m", False, 0)], buf=ALPHA + b"m0m\0",
                    expect=[(0, 0), (18, 0), (78, 0), (80, 0)], ordered=True))
    # :246-266 NoRepeat1: noruns literal "m": every later match repeats the last id
    out.append(dict(name="NoRepeat1", lits=[L("m", False, 0, noruns=True)], buf=ALPHA + b"m0m",
                    expect=[(0, 0)], ordered=True))
    # :268-292 NoRepeat2: "A" (id 42) in between re-arms "m"
    out.append(dict(name="NoRepeat2", lits=[L("m", False, 0, noruns=True), L("A", False, 42)], buf=ALPHA + b"m0m",
                    expect=[(0, 0), (32, 42), (78, 0)], ordered=True))
    # :294-316 NoRepeat3: two noruns literals sharing id 0
    out.append(dict(name="NoRepeat3", lits=[L("90m", False, 0, noruns=True), L("zA", False, 0, noruns=True)],
                    buf=ALPHA + b"m0m", expect=[(32, 0)], ordered=True))
    return out


def multi_location_cases():
    # :218-244 FDRp.MultiLocation: "abc" slid over a 128-byte zero buffer, exactly one match at i + 2
    out = []
    for i in range(128 - 3):
        buf = bytearray(128)
        buf[i:i + 3] = b"abc"
        out.append(dict(name=f"MultiLocation[{i}]", lits=[L("abc", False, 1)], buf=bytes(buf),
                        expect=[(i + 2, 1)], ordered=True))
    return out


ALIGN_PATTERNS = [(b"abaabaaa", ord("x")), (b"zzzyyzyz", 0x99), (b"abcdef l", 0)]  # fdr.cpp:563-567


def align_too_early_cases(alignments=range(32)):
    """:496-561 FDRpp.AlignAndTooEarly. Literal = pattern[:litLen] for litLen 1..8;
    buffer of `alien` bytes with the literal at both ends; scanning the window
    shrunk by j bytes on both sides must give 2 matches for j == 0, none else.
    Alignment i only shifts where the scanned window starts in memory (and here,
    where the block starts in the GPU corpus: tests scan them as one batch)."""
    out = []
    for pat, alien in ALIGN_PATTERNS:
        for lit_len in range(1, len(pat) + 1):
            lit = pat[:lit_len]
            for i in alignments:
                base = bytearray([alien]) * (5 * 32)
                base[i:i + lit_len] = lit
                base[i + 128 - lit_len:i + 128] = lit
                for j in range(lit_len + 1):
                    buf = bytes(base[i + j:i + j + 128 - 2 * j])
                    exp = [(lit_len - 1, 0), (127, 0)] if j == 0 else []
                    out.append(dict(name=f"Align[{pat!r},{lit_len},{i},{j}]", lits=[L(lit, False, 0)], buf=buf,
                                    expect=exp, ordered=False))
    return out


def _fib(n):
    a = b = c = 1
    for _ in range(n):
        c = a + b
        a, b = b, c
    return c


SHORT_ALPHABETS = [b"abx", b"xyz", b"\0A ", b"a \x99"]  # fdr.cpp:686-691


def short_writings(alphabet):
    """:594-685 FDRpa.ShortWritings: -> (buffers, literal groups of 32 [(bytes, id)]).
    Expected matches are whatever naive substring search finds (the test's own oracle)."""
    bufs = []
    for ln in range(1, 7):
        for j in range(3 ** ln):
            bufs.append(bytes(alphabet[(j // 3 ** k) % 3] for k in range(ln)))
    buflen = len(bufs)
    for ln in range(7, 64):
        for i in range(10):
            s = b""
            j = 0
            while len(s) < ln:
                s += bufs[_fib(i * 5 + j + (ln - 6) * 10) % buflen]
                j += 1
            bufs.append(s)
    pats = []
    for ln in range(1, 9):
        for j in range(2 ** ln):
            pats.append(bytes(alphabet[(j >> k) & 1] for k in range(ln)))
    groups = []
    for g in range(0, len(pats), 32):
        groups.append([(p, g + n) for n, p in enumerate(pats[g:g + 32])])
    return bufs, groups


def naive_matches(buf, pats):
    out = []
    for p, pid in pats:
        start = 0
        while True:
            k = buf.find(p, start)
            if k < 0:
                break
            out.append((k + len(p) - 1, pid))
            start = k + 1
    return sorted(out)


def flood_literals(c):
    """unit/internal/fdr_flood.cpp:159-176: 32 literals around byte c / cAlt."""
    bit = 1 << (c & 7)
    c_alt = c ^ bit
    lits = []
    for i in range(4):
        n = 1 << i
        s = bytearray([c]) * n
        lits.append(L(bytes(s), False, i * 8 + 0))
        s[0] = c_alt
        lits.append(L(bytes(s), False, i * 8 + 1))
        lits.append(L(bytes(s), True, i * 8 + 2))
        s[0] = c
        s[-1] = c_alt
        lits.append(L(bytes(s), False, i * 8 + 3))
        lits.append(L(bytes(s), True, i * 8 + 4))
        s_alt = bytearray([c_alt]) * n
        lits.append(L(bytes(s_alt), True, i * 8 + 5))
        s_alt[0] = c
        lits.append(L(bytes(s_alt), True, i * 8 + 6))
        lits.append(L(bytes(s_alt), False, i * 8 + 7))
    return lits, c_alt, bit


def flood_expected_counts(c, data_size=1024):
    """fdr_flood.cpp:186-237: expected per-id counts for a buffer of c, then of cAlt."""
    bit = 1 << (c & 7)
    is_case = (chr(c).isalpha() and c < 128) and bit == 0x20
    first, second = {}, {}
    for i in range(4):
        cnt = data_size - (1 << i) + 1
        z = cnt if i == 0 else 0
        first[i * 8 + 0] = cnt
        first[i * 8 + 1] = 0
        first[i * 8 + 3] = 0
        first[i * 8 + 7] = z
        if is_case:
            first[i * 8 + 2] = first[i * 8 + 4] = first[i * 8 + 5] = first[i * 8 + 6] = cnt
        else:
            first[i * 8 + 2] = first[i * 8 + 4] = first[i * 8 + 5] = 0
            first[i * 8 + 6] = z
        second[i * 8 + 0] = 0
        second[i * 8 + 1] = z
        second[i * 8 + 3] = z
        second[i * 8 + 5] = cnt
        second[i * 8 + 7] = 0
        if is_case:
            second[i * 8 + 2] = second[i * 8 + 4] = second[i * 8 + 6] = cnt
        else:
            second[i * 8 + 2] = second[i * 8 + 4] = z
            second[i * 8 + 6] = 0
    return first, second


def noodle_cases():
    """unit/internal/noodle.cpp:81-261 (nood1/nood2/noodLong/cutover): 1 KiB of 'a',
    literals "a", "aa", "aaaa" caseful and caseless, every start alignment and a
    range of lengths; expected: a match at every end >= len(lit) - 1."""
    out = []
    for lit in (b"a", b"aa", b"aaaa", b"A", b"AaaA"):
        nocase = lit != lit.lower()
        for ln in list(range(0, 40)) + [63, 64, 65, 127]:
            buf = b"a" * ln
            n = len(lit)
            exp = [(e, 0) for e in range(n - 1, ln)] if nocase or lit == lit.lower() else []
            out.append(dict(name=f"nood[{lit!r},{ln}]", lits=[L(lit, nocase, 0)], buf=buf, expect=exp, ordered=False))
    return out


def flood_mask_literals(c):
    """unit/internal/fdr_flood.cpp:242-316 FDRFloodp.WithMask: literals "cccc" / "CCCC" (c and its
    one-bit neighbour cAlt) carrying masks of length 1, 2, 4, 8 -- the mask reaches in front of
    the literal for length 8 -- case-sensitive and caseless. msk / cmp evolve in place inside
    one mask length exactly as in the test. -> (lits, cAlt, bit)"""
    bit = 1 << (c & 7)
    c_alt = c ^ bit
    is_case = bit == 0x20 and (chr(c).isalpha() and c < 128)
    s4, s4_alt = bytes([c]) * 4, bytes([c_alt]) * 4
    lits = []
    for i in range(4):
        n = 1 << i
        msk, cmp = bytearray(n), bytearray(n)

        def add(s, nocase, k):
            lits.append(L(s, nocase, i * 12 + k, msk=bytes(msk), cmp=bytes(cmp)))

        cmp[0], msk[0] = c_alt, 0xFF
        if n > 4:
            add(s4, False, 0)
            add(s4, True, 1)
        if is_case:
            add(s4, True, 2)
        if (c_alt & bit) == 0:
            msk[0] = ~bit & 0xFF
            add(s4, False, 3)
            add(s4, True, 4)
        cmp[0], msk[0] = c, 0xFF
        add(s4, False, 5)
        add(s4, True, 6)
        if n > 4:
            add(s4_alt, False, 7)
            add(s4_alt, True, 8)
        if is_case:
            add(s4_alt, True, 9)
            cmp[n - 1], msk[n - 1] = c_alt, 0xFF
            add(s4, True, 10)
            cmp[0] = c_alt
            add(s4, True, 11)
    return lits, c_alt, bit


def flood_mask_expected_counts(c, data_size=1024):
    """fdr_flood.cpp:324-398: the per-id counts the test asserts, for a buffer of c then of cAlt
    (only the ids it makes a statement about)."""
    bit = 1 << (c & 7)
    c_alt = c ^ bit
    is_case = bit == 0x20 and (chr(c).isalpha() and c < 128)
    cnt4 = data_size - 4 + 1
    first, second = {}, {}
    for i in range(4):
        n = 1 << i
        cm = min(cnt4, data_size - n + 1)
        first[i * 12 + 0] = first[i * 12 + 1] = first[i * 12 + 2] = 0
        if (c_alt & bit) == 0:
            first[i * 12 + 3] = first[i * 12 + 4] = cm
        if n > 4:
            first[i * 12 + 5] = first[i * 12 + 6] = cm
            first[i * 12 + 7] = 0
            first[i * 12 + 8] = cm if is_case else 0
        else:
            first[i * 12 + 5] = first[i * 12 + 6] = cnt4
        if is_case:
            first[i * 12 + 9] = cm
            first[i * 12 + 10] = first[i * 12 + 11] = 0
        for k in (0, 3, 5, 6, 7, 8, 9):
            second[i * 12 + k] = 0
        if is_case:
            second[i * 12 + 1] = cm if n > 4 else 0
            second[i * 12 + 2] = cm
            second[i * 12 + 4] = cm if chr(c).islower() else 0
            second[i * 12 + 10] = cnt4 if n == 1 else 0
            second[i * 12 + 11] = cm
        else:
            for k in (1, 2, 4, 10, 11):
                second[i * 12 + k] = 0
    return first, second
