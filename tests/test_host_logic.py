"""Host-side logic without a GPU: the C-ABI library loads and exports every symbol
include/hsgpu.h declares, literal validation mirrors the reference's, compiled
tables satisfy their invariants (every key variant of every literal is present
in the LDS filter and reachable through the exact hash table), serialisation
round-trips and rejects corruption, and the sequential replay applies the
reference's group / noruns / termination rules."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import hyperscan_amd as H
from hyperscan_amd import _native
from hyperscan_amd import hwlm as hw
from tests.util import random_literals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

F_A, F_B, F_C, F_REPL, F_K2, F_S2, F_BLIND, F_BFOLD, F_PAIR = 1, 2, 4, 8, 16, 32, 64, 128, 256
F_GATE, F_WIDE, F_BLOOM = 512, 1024, 2048
NO_GATE, FORCE_BLOOM = 4096, 8192
FORCE_REPL, FORCE_HASHED, FORCE_K2, FORCE_K1, FORCE_S1, FORCE_BLIND, FORCE_S2, NO_FOLD = 1, 2, 4, 8, 16, 32, 64, 512
FORCE_PAIR, KEY_M = 1024, 0x01000000
MUL, HT_MUL = 0x9E3779, 0x9E3779B1


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    hdr = open(os.path.join(ROOT, "include", "hsgpu.h")).read() + open(os.path.join(ROOT, "include", "hsgpu_tuning.h")).read()
    declared = set(re.findall(r"\b(hsgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hsgpu_hwlm_cb", "hsgpu_chunk_cb"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/hsgpu.h but not exported"
    assert set(_native.exported_symbols()) <= declared | {"hsgpu_last_error", "hsgpu_version"}
    assert b"gfx950" in lib.hsgpu_version()
    # the public hs_* facade (include/hs_gpu.h)
    hdr2 = open(os.path.join(ROOT, "include", "hs_gpu.h")).read()
    declared2 = set(re.findall(r"\b(hs_[a-z_]+)\s*\(", hdr2)) - {"hs_batch_event_handler"}
    assert {"hs_compile", "hs_compile_lit_multi", "hs_scan", "hs_alloc_scratch", "hs_free_database"} <= declared2
    for name in sorted(declared2):
        assert hasattr(lib, name), f"{name} declared in include/hs_gpu.h but not exported"


def test_confirm_partition_covers_every_part_and_fits_the_device():
    """hsgpu_confirm_partition (the confirm kernel's static parts per worker, csrc/runtime.hip): every part has a worker, no
    more workers than the device holds, and the common geometries get what DESIGN 4.7 says."""
    import ctypes as C
    lib = _native.load_library()
    f = lib.hsgpu_confirm_partition
    f.restype = C.c_uint
    f.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    q, k = C.c_uint(), C.c_uint()
    for shares in [1, 3, 16, 255, 1024, 4096, 6144, 8192, 16384, 65536, 100003]:
        for w_max in [4, 1024, 4096, 6144, 8192]:
            workers = f(shares, w_max, C.byref(q), C.byref(k))
            parts = shares * q.value
            assert 1 <= q.value <= 8 and k.value >= 1
            assert workers <= w_max, (shares, w_max, workers)
            assert workers * k.value >= parts > (workers - 1) * k.value, (shares, w_max, q.value, k.value, workers)
    assert (f(4096, 6144, C.byref(q), C.byref(k)), q.value, k.value) == (6144, 3, 2)
    assert (f(4096, 8192, C.byref(q), C.byref(k)), q.value, k.value) == (8192, 2, 1)  # the fast step's eight workgroups per CU
    assert (f(6144, 6144, C.byref(q), C.byref(k)), q.value, k.value) == (6144, 1, 1)
    assert f(0, 6144, C.byref(q), C.byref(k)) == 0


def test_hs_compile_errors_without_gpu():
    """Compile-side argument and syntax checks of the hs_* facade need no GPU."""
    from hyperscan_amd import hs

    with pytest.raises(hs.HsError) as e:
        hs.Database.compile(["foo"], mode=hs.HS_MODE_STREAM)
    assert e.value.code == hs.HS_COMPILER_ERROR
    with pytest.raises(hs.HsError) as e:
        hs.Database.compile(["ok", "a(b|c"])  # unbalanced group
    assert e.value.expression == 1
    with pytest.raises(hs.HsError):
        hs.Database.compile(["\\w+[^abc]"])  # no mandatory literal, and no class small enough to stand in for one
    assert len(hs.Database.compile(["\\d+[abc]"]).literals()) == 3  # [abc] as a|b|c: one-byte literals
    db = hs.Database.compile(["needle[a-z]{2,5}\\d", "x\\.y"], [hs.HS_FLAG_CASELESS, 0], [3, 4])
    assert hs.Database.deserialize(db.serialize()).size() == db.size()


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(H.HsgpuError):
        H.Scratch(0)


@pytest.mark.parametrize("bad", [
    dict(s=b"123456789"),                      # > HWLM_LITERAL_MAX_LEN (hwlm.h:75)
    dict(s=b""),                               # empty
    dict(s=b"ab", id=0xFFFFFFFF),              # reserved id (hwlm_build.cpp:190)
    dict(s=b"ab", msk=b"\xff" * 9, cmp=b"a" * 9),  # > HWLM_MASKLEN
    dict(s=b"ab", msk=b"\xff\xff", cmp=b"xb"),     # msk/cmp contradict the literal (maskIsConsistent)
])
def test_build_rejects_invalid_literals(bad):
    with pytest.raises(H.HsgpuError) as e:
        H.hwlm_build([H.HwlmLiteral(**bad)])
    assert e.value.code == -4  # HS_COMPILER_ERROR


def parse(blob):
    f = struct.unpack_from("<32I", blob, 0)
    names = ["magic", "version", "blob_bytes", "flags", "n_lits", "max_size", "filter_log2", "filter_entries",
             "ht_a_log2", "ht_b_log2", "n_a", "n_b", "n_c", "off_filter", "off_c2bits", "off_ht_a", "off_ht_b",
             "off_c2ref", "off_lists", "n_lists", "off_lits", "checksum"]
    h = dict(zip(names, f))
    fl = h["flags"]
    nw = (2 << h["filter_log2"]) if fl & (F_PAIR | F_WIDE) else (32 << h["filter_log2"]) if fl & F_REPL else (1 << h["filter_log2"])
    h["filter"] = np.frombuffer(blob, "<u4", nw, h["off_filter"])
    h["c2bits"] = np.frombuffer(blob, "<u4", 3072 if fl & F_BLOOM else 2048, h["off_c2bits"])
    # 16-byte buckets of 4 tagged slots: DIRECT(31) | delta(30) | tag6(29..24) | index24
    h["ht_a"] = np.frombuffer(blob, "<u4", 4 << h["ht_a_log2"], h["off_ht_a"]).reshape(-1, 4)
    h["ht_b"] = np.frombuffer(blob, "<u4", 4 << h["ht_b_log2"], h["off_ht_b"]).reshape(-1, 4)
    h["c2ref"] = np.frombuffer(blob, "<u4", 65536 if fl & F_C else 4, h["off_c2ref"])
    h["lists"] = np.frombuffer(blob, "<u4", h["n_lists"], h["off_lists"])
    h["lits"] = np.frombuffer(blob, np.dtype([("v", "<u8"), ("msk", "<u8"), ("groups", "<u8"), ("id", "<u4"),
                                              ("size", "u1"), ("flags", "u1"), ("pad", "<u2")]), h["n_lits"],
                              h["off_lits"])
    return h


def list_entries(h, ref):
    if ref & 0x80000000:
        return [ref & 0x40FFFFFF]
    out, i = [], (ref & 0xFFFFFF) - 1
    while True:
        e = int(h["lists"][i])
        out.append(e & 0x7FFFFFFF)
        i += 1
        if e & 0x80000000:
            return out


def ht_lookup(h, tab, log2, key):
    """All literals the device would compare for this key: the entries of every slot whose
    tag matches (a tag match is a hint; (v, msk) decides), walking on while buckets are full."""
    prod = (key * HT_MUL) & 0xFFFFFFFF
    b, tag = prod >> (32 - log2), (prod >> (26 - log2)) & 63
    out = []
    while True:
        for slot in tab[b]:
            slot = int(slot)
            if slot and (slot >> 24) & 63 == tag:
                out += list_entries(h, slot)
        if not tab[b][3]:
            return out or None
        b = (b + 1) & ((1 << log2) - 1)


def bloom_idx(hi, lo, salt):
    """table.h hsgpu_bloom_idx: one bit index per plane of 2^15 bits"""
    M = 0xFFFFFFFF
    h = ((hi * 0x9E3779B1) & M) ^ ((lo * 0x85EBCA6B + salt * 0x632BE5AB) & M)
    h ^= h >> 15
    p1, p2 = (h * 0xC2B2AE35) & M, (h * 0x27D4EB2F) & M
    return [p1 >> 17, p2 >> 17, (p1 >> 2) & 0x7FFF]


def bloom_pass(h, hi, lo, salt):
    return all(int(h["c2bits"][j * 1024 + (i >> 5)]) >> (i & 31) & 1 for j, i in enumerate(bloom_idx(hi, lo, salt)))


def filter_word_and_bits(h, x24, b3):
    fl, k = h["flags"], h["filter_log2"]
    prod = (x24 * MUL) & 0xFFFFFFFF
    if fl & F_REPL:
        a = prod >> (32 - k)
        words = [int(h["filter"][a * 32 + c]) for c in range(32)]
    else:
        a = prod >> (30 - k)
        words = [int(h["filter"][a >> 2])]
    bit_a = [(a + b3) & 31] + ([((prod >> 8) + b3) & 31] if fl & F_K2 else [])
    bit_b = [a & 31] + ([(prod >> 8) & 31] if fl & F_K2 else [])
    if fl & F_BFOLD:  # the filter kernel runs the 4-byte-key test only; a hit probes both exact tables
        bit_b = bit_a
    return words, bit_a, bit_b


def check_table_covers(lits, flags):
    """Every text window a literal can match must pass the filter and reach the
    literal through the exact tables: checked on concrete windows built from each
    literal (both case variants, both stride-2 alignments)."""
    t = H.hwlm_build(lits, flags)
    h = parse(t.serialize())
    fl = h["flags"]
    assert h["magic"] == 0x54475348 and h["n_lits"] == len(lits)
    key_mask = 0xDFDFDFDF if fl & F_BLIND else 0xFFFFFFFF
    rng = np.random.default_rng(5)
    deltas = (0, 1) if fl & F_S2 else (0,)
    for li, lit in enumerate(lits):
        for variant in range(4):
            s = bytearray(lit.s)
            if lit.nocase:
                for i, c in enumerate(s):
                    if chr(c).isalpha() and c < 128 and (variant >> (i & 1)) & 1:
                        s[i] = c ^ 0x20
            ctx = bytearray(bytes(rng.integers(0, 256, 8, dtype=np.uint8)) + bytes(s))  # random bytes in front ...
            for k in range(1, len(lit.msk) + 1):  # ... that satisfy the literal's msk / cmp where it reaches in front of the string
                if k > len(s):
                    ctx[-k] = (ctx[-k] & ~lit.msk[-k] & 0xFF) | (lit.cmp[-k] & lit.msk[-k])
            ctx = bytes(ctx)
            for d in deltas:
                if len(s) - d < 1:
                    continue
                # lookup position q = end - d; window bytes b3 b2 b1 b0 end at q
                w = ctx[: len(ctx) - d][-4:]
                b3, x24 = w[0], w[1] | w[2] << 8 | w[3] << 16
                xh = x24 & (0xDFDFDF if fl & F_BLIND else 0xFFFFFF)
                words, bit_a, bit_b = filter_word_and_bits(h, xh, b3)
                w4 = (w[0] | w[1] << 8 | w[2] << 16 | w[3] << 24) & key_mask
                want = li | d << 30
                found = False

                def gate(key):  # HSGPU_F_GATE: the confirm kernel probes a table only for keys whose gate bit is set
                    if fl & F_BLOOM:  # ... HSGPU_F_BLOOM: only for windows some literal's FULL key matches (groups 3, 4, 5 of table.h)
                        assert d == 0
                        if key >> 24 == 0xB5:
                            return bloom_pass(h, (w4 & 0xFFFFFF00), 0, 3)
                        w5 = (ctx[-5] & (key_mask & 0xFF)) << 24
                        return bloom_pass(h, w4, w5, 5) or bloom_pass(h, w4, 0, 4)
                    if not fl & F_GATE:
                        return True
                    g = ((key * HT_MUL) & 0xFFFFFFFF) >> 16
                    return bool(h["c2bits"][g >> 5] >> (g & 31) & 1)

                if fl & F_A and all(all(wd >> b & 1 for b in bit_a) for wd in words) and gate(w4):
                    ents = ht_lookup(h, h["ht_a"], h["ht_a_log2"], w4)
                    found |= ents is not None and want in ents
                if not found and fl & F_B and all(all(wd >> b & 1 for b in bit_b) for wd in words) and gate((w4 >> 8) | 0xB5000000):
                    ents = ht_lookup(h, h["ht_b"], h["ht_b_log2"], w4 >> 8)
                    found |= ents is not None and want in ents
                if not found and fl & F_C:
                    kc = w4 >> 16
                    if h["c2bits"][kc >> 5] >> (kc & 31) & 1 and h["c2ref"][kc]:
                        found |= want in list_entries(h, int(h["c2ref"][kc]))
                assert found, (lit, d, variant, flags)
    return h


@pytest.mark.parametrize("flags", [0, FORCE_REPL, FORCE_HASHED, FORCE_HASHED | FORCE_K2, FORCE_REPL | FORCE_K2,
                                   FORCE_S1, FORCE_S1 | FORCE_HASHED | FORCE_K1, FORCE_BLIND, FORCE_S2,
                                   FORCE_S2 | FORCE_HASHED | FORCE_BLIND | FORCE_K2, NO_FOLD, NO_FOLD | FORCE_S2])
def test_table_covers_every_literal(flags):
    rng = np.random.default_rng(flags + 1)
    lits = random_literals(rng, 150, 1, 8, nocase_frac=0.4)
    h = check_table_covers(lits, flags)
    if flags & FORCE_REPL:
        assert h["flags"] & F_REPL
    if flags & FORCE_HASHED:
        assert not h["flags"] & F_REPL
    if flags & FORCE_S1:
        assert not h["flags"] & F_S2
    if flags & NO_FOLD or h["flags"] & (F_REPL | F_C):
        assert not h["flags"] & F_BFOLD


def test_few_three_byte_keys_fold_into_the_four_byte_test():
    """Few 3-byte keys beside 4-byte ones: each owns its whole filter word (HSGPU_F_BFOLD)."""
    rng = np.random.default_rng(11)
    lits = random_literals(rng, 400, 4, 8, nocase_frac=0.3) + random_literals(rng, 20, 3, 3, nocase_frac=0)
    for i, l in enumerate(lits):
        l.id = i
    h = check_table_covers(lits, FORCE_HASHED)
    assert h["flags"] & F_BFOLD and h["flags"] & F_B
    assert (h["filter"] == 0xFFFFFFFF).sum() >= 1
    h2 = check_table_covers(lits, FORCE_HASHED | NO_FOLD)
    assert not h2["flags"] & F_BFOLD
    # many 3-byte keys: folding would flood the filter, so it is not chosen
    many = random_literals(rng, 400, 4, 8, nocase_frac=0) + random_literals(rng, 2000, 3, 3, nocase_frac=0)
    assert not H.hwlm_build(many, FORCE_HASHED).info()["flags"] & F_BFOLD


def check_pair_table_covers(lits, flags=FORCE_PAIR):
    """The pair filter (csrc/table.h, HSGPU_F_PAIR): for every literal, in both parities of its end offset
    and every case variant, the lookup that is responsible for it passes the filter rule (evaluated by
    tests/pair_model.py on the serialised table) and reaches the literal through the exact tables -- the
    4-byte table, the 3-byte table, or the 3-byte table under a late key (HSGPU_KEY_M)."""
    from tests import pair_model as pm

    t = H.hwlm_build(lits, flags)
    blob = t.serialize()
    h = parse(blob)
    fl = h["flags"]
    assert fl & F_PAIR and fl & F_S2 and not fl & (F_REPL | F_K2 | F_C | F_BFOLD)
    key_mask = 0xDFDFDFDF if fl & F_BLIND else 0xFFFFFFFF
    gate = h["c2bits"]
    rng = np.random.default_rng(6)
    n_late = 0
    for li, lit in enumerate(lits):
        for variant in range(4):
            s = bytearray(lit.s)
            if lit.nocase:
                for i, c in enumerate(s):
                    if chr(c).isalpha() and c < 128 and (variant >> (i & 1)) & 1:
                        s[i] = c ^ 0x20
            for parity in (0, 1):
                pre = 8 + ((8 + len(s) - 1 - parity) & 1)  # the literal ends at an offset of this parity
                buf = np.concatenate([rng.integers(0, 256, pre, dtype=np.uint8), np.frombuffer(bytes(s), dtype=np.uint8),
                                      rng.integers(0, 256, 6, dtype=np.uint8)])
                e = pre + len(s) - 1
                assert e & 1 == parity
                cand, _ = pm.candidates(blob, buf)
                w = lambda q: int(buf[q - 3]) | int(buf[q - 2]) << 8 | int(buf[q - 1]) << 16 | int(buf[q]) << 24

                def reaches(q, delta):
                    """lookup q passes the filter and the exact tables lead to (li, delta)"""
                    if not cand[q // 2]:
                        return False
                    w4 = w(q) & key_mask
                    if delta >= 0:
                        want = li | delta << 30
                        ents = ht_lookup(h, h["ht_a"], h["ht_a_log2"], w4) if fl & F_A else None
                        if ents and want in ents:
                            return True
                        kb = w4 >> 8
                    else:
                        want = li
                        kb = ((w(q - 1) & key_mask) >> 8) | KEY_M
                    g = ((kb * HT_MUL) & 0xFFFFFFFF) >> 16
                    if not gate[g >> 5] >> (g & 31) & 1:
                        return False
                    ents = ht_lookup(h, h["ht_b"], h["ht_b_log2"], kb)
                    return bool(ents) and want in ents

                if parity == 0:
                    assert reaches(e, 0), (lit, variant)
                else:
                    late = reaches(e + 1, -1)
                    n_late += late
                    assert reaches(e - 1, 1) or late, (lit, variant)
    return h, n_late


def test_pair_table_covers_every_literal():
    rng = np.random.default_rng(21)
    lits = random_literals(rng, 400, 3, 8, nocase_frac=0.4)
    h, n_late = check_pair_table_covers(lits)
    n3 = sum(len(l.s) == 3 for l in lits)
    assert n3 and n_late >= 4 * n3  # 3-byte literals are keyed one byte late at odd ends
    assert struct.unpack_from("<2I", H.hwlm_build(lits, FORCE_PAIR).serialize(), 88)[1] == n3
    # literals of 4 to 8 bytes only: nothing is keyed late, every byte of b0 enters the hash
    h2, n_late2 = check_pair_table_covers(random_literals(rng, 300, 4, 8, nocase_frac=0))
    assert n_late2 == 0 and not h2["flags"] & F_BLIND
    # msk / cmp literals, wildcards in front of and inside the string
    extra = [H.HwlmLiteral("bcd", False, 900, msk=b"\xf0\xff\xff\xff", cmp=b"\x30bcd"),
             H.HwlmLiteral("wxyz", True, 901, msk=b"\xff\x00\x00\x00\x00", cmp=b"Q\x00\x00\x00\x00"),
             H.HwlmLiteral("klm", False, 902, msk=b"\xdf\x00\xff\xff\xff", cmp=b"A\x00klm")]
    t = H.hwlm_build(random_literals(rng, 50, 3, 8) + extra, FORCE_PAIR)
    assert t.info()["flags"] & F_PAIR
    # a set the pair filter cannot hold (a 1-byte literal needs 2^14 entries): refused, like a bad engine hint
    with pytest.raises(H.HsgpuError):
        H.hwlm_build([H.HwlmLiteral("a", False, 1), H.HwlmLiteral("abcd", False, 2)], FORCE_PAIR)


def test_table_modes_auto():
    rng = np.random.default_rng(3)
    small = H.hwlm_build(random_literals(rng, 64, 4, 8, nocase_frac=0)).info()
    assert small["flags"] & F_S2 and not small["flags"] & F_REPL and not small["flags"] & F_BLIND
    short = H.hwlm_build(random_literals(rng, 64, 1, 3, nocase_frac=0)).info()
    assert short["flags"] & F_REPL and not short["flags"] & F_S2  # stride 1: bank-replicated filter
    big = H.hwlm_build(random_literals(rng, 5000, 3, 8, nocase_frac=0.3)).info()
    assert not big["flags"] & F_REPL and not big["flags"] & F_S2 and big["flags"] & F_BLIND
    # case-blind keys: a caseless literal costs one filter entry, not one per case variant
    assert big["filter_entries"] <= 5000


def test_msk_cmp_literals_in_table():
    lits = [H.HwlmLiteral("bc", False, 7, msk=b"\xf0\xff\xff", cmp=b"\x30bc"),
            H.HwlmLiteral("xyz", True, 8, msk=b"\xff\x00\x00\x00", cmp=b"Q\x00\x00\x00")]
    t = H.hwlm_build(lits)
    h = parse(t.serialize())
    assert h["lits"]["size"].tolist() == [3, 4]
    assert h["max_size"] == 4


def test_serialize_roundtrip_and_corruption():
    rng = np.random.default_rng(9)
    t = H.hwlm_build(random_literals(rng, 300, 2, 8))
    blob = t.serialize()
    assert len(blob) == t.size == H.hwlm_size(t)
    t2 = H.HwlmTable.deserialize(blob)
    assert t2.serialize() == blob and t2.info() == t.info()
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x40
    with pytest.raises(H.HsgpuError):
        H.HwlmTable.deserialize(bytes(bad))
    with pytest.raises(H.HsgpuError):
        H.HwlmTable.deserialize(blob[:-16])
    with pytest.raises(H.HsgpuError):
        H.HwlmTable.deserialize(b"\0" * 64)


def replay(table, recs, groups, cb):
    arr = np.zeros(len(recs), dtype=hw.MATCH_DTYPE)
    for i, (e, lit) in enumerate(recs):
        arr[i] = (0, e, 0, lit)
    ccb = _native.HWLM_CB(lambda e, i, _c: cb(e, i))
    return table._lib.hsgpu_hwlm_replay(table._h, arr.ctypes.data, len(recs), ccb, None, groups)


def test_replay_rules():
    """hsgpu_hwlm_replay = what the reference does inside confWithBit
    (fdr_confirm_runtime.h:69-96): noruns vs last delivered id, group gate against
    the callback's last return value, stop on 0."""
    lits = [H.HwlmLiteral("m", False, 0, noruns=True), H.HwlmLiteral("A", False, 42),
            H.HwlmLiteral("q", False, 5, groups=4)]
    t = H.hwlm_build(lits)
    out = []
    recs = [(0, 0), (18, 0), (32, 1), (78, 0), (80, 0)]
    assert replay(t, recs, H.HWLM_ALL_GROUPS, lambda e, i: (out.append((e, i)), H.HWLM_ALL_GROUPS)[1]) == 0
    assert out == [(0, 0), (32, 42), (78, 0)]          # NoRepeat2, unit/internal/fdr.cpp:268-292
    out = []
    assert replay(t, recs, H.HWLM_ALL_GROUPS, lambda e, i: (out.append((e, i)), 0)[1]) == 1
    assert out == [(0, 0)]                             # HWLM_TERMINATED after exactly one match
    out = []
    assert replay(t, [(1, 2), (2, 1), (3, 2)], 3, lambda e, i: (out.append((e, i)), 4)[1]) == 0
    assert out == [(2, 42), (3, 5)]                    # group 4 only live after the first callback


def test_hs_facade_compile_side_extras():
    """hs_compile_ext_multi validation, hs_expression_(ext_)info, hs_serialized_database_*,
    hs_populate_platform and the allocator hooks: all host-only."""
    from hyperscan_amd import hs

    lib = hs._lib()
    # ext validation (src/compiler/compiler.cpp:97-130)
    with pytest.raises(hs.HsError) as e:
        hs.Database.compile_ext(["abc"], ext=[hs.ExprExt.make(min_offset=10, max_offset=5)])
    assert e.value.code == hs.HS_COMPILER_ERROR and e.value.expression == 0
    with pytest.raises(hs.HsError):
        hs.Database.compile_ext(["abc"], ext=[hs.ExprExt.make(edit_distance=1)])  # graph compiler territory
    db = hs.Database.compile_ext(["abc\\d+", "xyz"], ids=[1, 2], ext=[hs.ExprExt.make(min_offset=4, max_offset=90), None])
    blob = db.serialize()
    db2 = hs.Database.deserialize(blob)  # the ext parameters travel with the database
    assert db2.serialize() == blob
    n = C.c_size_t()
    assert lib.hs_serialized_database_size(blob, len(blob), C.byref(n)) == 0 and n.value == db.size()
    info = C.c_char_p()
    assert lib.hs_serialized_database_info(blob, len(blob), C.byref(info)) == 0
    assert b"BLOCK" in info.value and b"gfx950" in info.value
    assert lib.hs_serialized_database_size(b"junkjunk", 8, C.byref(n)) == hs.HS_INVALID
    bad_version = bytes([blob[0] ^ 1]) + blob[1:]
    assert lib.hs_serialized_database_size(bad_version, len(blob), C.byref(n)) == -5  # HS_DB_VERSION_ERROR
    # expression info: widths of literal prefix + tail
    assert hs.expression_info("abc") == (3, 3)
    assert hs.expression_info("abc\\d{2,4}x?") == (5, 8)
    assert hs.expression_info("abc[a-z]+") == (4, 0xFFFFFFFF)
    assert hs.expression_info("abc[a-z]+", ext=hs.ExprExt.make(min_length=10, max_offset=64)) == (10, 64)
    with pytest.raises(hs.HsError):
        hs.expression_info("a|(bc")
    plat = (C.c_ulonglong * 4)(1, 2, 3, 4)
    assert lib.hs_populate_platform(plat) == 0 and list(plat) == [0, 0, 0, 0]


def test_hs_allocator_hooks():
    from hyperscan_amd import hs

    lib = hs._lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live, calls = set(), [0, 0]
    ALLOC, FREE = C.CFUNCTYPE(C.c_void_p, C.c_size_t), C.CFUNCTYPE(None, C.c_void_p)

    @ALLOC
    def my_alloc(n):
        p = libc.malloc(n)
        live.add(p)
        calls[0] += 1
        return p

    @FREE
    def my_free(p):
        if p:
            live.discard(p)
            calls[1] += 1
            libc.free(p)

    @ALLOC
    def misaligned(n):
        return libc.malloc(n + 8) + 4  # leaked on purpose: the library must refuse it ...

    @FREE
    def drop(p):  # ... and hands it straight back to the matching free hook
        calls[1] += 1

    lib.hs_set_allocator.argtypes = [ALLOC, FREE]
    lib.hs_set_database_allocator.argtypes = [ALLOC, FREE]
    try:
        assert lib.hs_set_allocator(my_alloc, my_free) == 0
        db = hs.Database.compile(["hook", "lit\\d"])
        blob_p, blob_n = C.c_void_p(), C.c_size_t()
        assert lib.hs_serialize_database(db._h, C.byref(blob_p), C.byref(blob_n)) == 0
        assert blob_p.value in live  # serialised bytes: misc allocator
        my_free(blob_p.value)
        assert calls[0] >= 2
        with pytest.raises(hs.HsError):
            hs.Database.compile(["a(b"])  # the error object came from the misc allocator and went back
        db.close()
        assert not live, "everything handed out through the hooks was returned through them"
        assert lib.hs_set_database_allocator(misaligned, drop) == 0
        before = calls[1]
        with pytest.raises(hs.HsError):
            hs.Database.compile(["x1"])
        assert calls[1] > before
    finally:
        lib.hs_set_allocator(ALLOC(0), FREE(0))
    hs.Database.compile(["back", "to", "malloc"]).close()


def test_hs_compile_arg_checks_like_the_reference():
    """unit/hyperscan/arg_checks.cpp:102-337, the compile-side half (no device needed): same
    return codes, and the reference's own messages where the test pins them."""
    from hyperscan_amd import hs

    lib = hs._lib()
    P = C.POINTER
    lib.hs_compile.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_void_p, P(C.c_void_p), P(P(hs.CompileErrorStruct))]

    def compile_one(expr, flags, mode, platform=None, want_db=True):
        db, err = C.c_void_p(), P(hs.CompileErrorStruct)()
        rv = lib.hs_compile(expr, flags, mode, platform, C.byref(db) if want_db else None, C.byref(err))
        msg = err.contents.message.decode() if err else None
        if err:
            lib.hs_free_compile_error(err)
        if rv == 0:
            lib.hs_free_database(db)
        return rv, msg, db.value

    ERR = hs.HS_COMPILER_ERROR
    # SingleCompileBlockNoPattern / StreamingNoPattern: NULL pattern
    for mode in (hs.HS_MODE_BLOCK, hs.HS_MODE_STREAM):
        rv, msg, dbv = compile_one(None, 0, mode)
        assert rv == ERR and msg and dbv is None
    # SingleCompileBlockNoDatabase: NULL database pointer still yields an error object
    rv, msg, _ = compile_one(b"foobar", 0, hs.HS_MODE_BLOCK, want_db=False)
    assert rv == ERR and msg
    # SingleCompileNoMode / SeveralModes1
    one_mode = "Invalid parameter: mode must have one (and only one) of HS_MODE_BLOCK, HS_MODE_STREAM or HS_MODE_VECTORED set."
    assert compile_one(b"foobar", 0, 0)[:2] == (ERR, one_mode)
    assert compile_one(b"foobar", 0, hs.HS_MODE_STREAM | hs.HS_MODE_BLOCK)[:2] == (ERR, one_mode)
    # SingleCompileBogusFlags: 0xdeadbeef has HS_FLAG_COMBINATION set
    assert compile_one(b"foobar", 0xDEADBEEF, hs.HS_MODE_BLOCK)[:2] == (
        ERR, "only HS_FLAG_QUIET and HS_FLAG_SINGLEMATCH are supported in combination with HS_FLAG_COMBINATION.")
    assert compile_one(b"foobar", 1 << 20, hs.HS_MODE_BLOCK)[:2] == (ERR, "Unrecognised flag.")
    assert compile_one(b"foobar", hs.HS_FLAG_SINGLEMATCH | hs.HS_FLAG_SOM_LEFTMOST, hs.HS_MODE_BLOCK)[:2] == (
        ERR, "HS_FLAG_SINGLEMATCH is not supported in combination with HS_FLAG_SOM_LEFTMOST.")
    # SingleCompileBogusMode1
    assert compile_one(b"foobar", 0, hs.HS_MODE_STREAM | (1 << 30))[:2] == (ERR, "Invalid parameter: unrecognised mode flags.")
    # SOM horizon flags belong to streaming mode (hs.cpp:100-115)
    assert compile_one(b"foobar", 0, hs.HS_MODE_BLOCK | (1 << 24))[:2] == (
        ERR, "Invalid parameter: the HS_MODE_SOM_HORIZON_ mode flags may only be set in streaming mode.")
    assert compile_one(b"foobar", 0, hs.HS_MODE_STREAM | (1 << 24) | (1 << 25))[:2] == (
        ERR, "Invalid parameter: only one HS_MODE_SOM_HORIZON_ mode flag can be set.")
    # SingleCompileBadTune / BadFeatures: hs_platform_info_t {tune u32, cpu_features u64, reserved x2}
    class Plat(C.Structure):
        _fields_ = [("tune", C.c_uint), ("cpu_features", C.c_ulonglong), ("r1", C.c_ulonglong), ("r2", C.c_ulonglong)]
    assert compile_one(b"foobar", 0, hs.HS_MODE_BLOCK, C.byref(Plat(42, 0, 0, 0)))[:2] == (
        ERR, "Invalid tuning value specified in the platform information.")
    assert compile_one(b"foobar", 0, hs.HS_MODE_BLOCK, C.byref(Plat(0, 42, 0, 0)))[:2] == (
        ERR, "Invalid cpu features specified in the platform information.")
    assert compile_one(b"foobar", 0, hs.HS_MODE_BLOCK, C.byref(Plat(10, 1 << 2, 0, 0)))[0] == 0  # ICX + AVX2: accepted, ignored
    # streaming is a well-formed request this engine declines
    assert compile_one(b"foobar", 0, hs.HS_MODE_STREAM)[:2] == (
        ERR, "Only HS_MODE_BLOCK and HS_MODE_VECTORED are supported by the GPU literal engine.")
    assert compile_one(b"foobar", 0, hs.HS_MODE_VECTORED)[0] == 0
    # MultiCompileZeroPatterns / NoPattern
    lib.hs_compile_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, P(C.c_void_p),
                                     P(P(hs.CompileErrorStruct))]
    for exprs, n in ((None, 1), ((C.c_char_p * 1)(b"foobar"), 0)):
        db, err = C.c_void_p(), P(hs.CompileErrorStruct)()
        assert lib.hs_compile_multi(exprs, None, None, n, hs.HS_MODE_BLOCK, None, C.byref(db), C.byref(err)) == ERR
        assert err and db.value is None
        lib.hs_free_compile_error(err)


def test_hs_runtime_arg_checks_without_a_device():
    """unit/hyperscan/arg_checks.cpp:875-1545, the checks that come before any device work:
    NULL / corrupt handles are HS_INVALID, never a crash."""
    from hyperscan_amd import hs

    lib = hs._lib()
    db = hs.Database.compile(["foobar"])
    cb = hs.MATCH_CB(lambda *a: 0)
    junk = C.create_string_buffer(4096)  # a "database" / "scratch" whose magic is wrong
    # ScanBlockNoDatabase / BrokenDatabaseMagic / NoScratch / NoData
    assert lib.hs_scan(None, b"data", 4, 0, None, cb, None) == hs.HS_INVALID
    assert lib.hs_scan(db._h, b"data", 4, 0, None, cb, None) == hs.HS_INVALID
    assert lib.hs_scan(C.cast(junk, C.c_void_p), b"data", 4, 0, C.cast(junk, C.c_void_p), cb, None) == hs.HS_INVALID
    assert lib.hs_scan(db._h, None, 4, 0, C.cast(junk, C.c_void_p), cb, None) == hs.HS_INVALID
    # AllocScratchNoDatabase / NullScratchPtr / BogusScratch / BadDatabaseMagic
    s = C.c_void_p()
    assert lib.hs_alloc_scratch(None, C.byref(s)) == hs.HS_INVALID
    assert lib.hs_alloc_scratch(db._h, None) == hs.HS_INVALID
    bogus = C.cast(junk, C.c_void_p)
    assert lib.hs_alloc_scratch(db._h, C.byref(bogus)) == hs.HS_INVALID
    assert lib.hs_alloc_scratch(C.cast(junk, C.c_void_p), C.byref(s)) == hs.HS_INVALID
    # CloneScratchNoSource, scratch size / free of junk
    lib.hs_clone_scratch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.hs_clone_scratch(None, C.byref(s)) == hs.HS_INVALID
    assert lib.hs_free_scratch(None) == 0 and lib.hs_free_scratch(C.cast(junk, C.c_void_p)) == hs.HS_INVALID
    # SerializeNoDatabase / NoBuffer / NoLength; database_size / info with junk
    p, n = C.c_void_p(), C.c_size_t()
    assert lib.hs_serialize_database(None, C.byref(p), C.byref(n)) == hs.HS_INVALID
    assert lib.hs_serialize_database(db._h, None, C.byref(n)) == hs.HS_INVALID
    assert lib.hs_serialize_database(db._h, C.byref(p), None) == hs.HS_INVALID
    assert lib.hs_database_size(C.cast(junk, C.c_void_p), C.byref(n)) == hs.HS_INVALID
    assert lib.hs_free_database(None) == 0 and lib.hs_free_database(C.cast(junk, C.c_void_p)) == hs.HS_INVALID
    lib.hs_deserialize_database.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    assert lib.hs_deserialize_database(None, 10, C.byref(p)) == hs.HS_INVALID
    assert lib.hs_deserialize_database(b"x" * 10, 10, None) == hs.HS_INVALID


def test_hs_serialized_database_is_guarded_by_a_crc():
    """unit/hyperscan/serialize.cpp: a damaged blob must not deserialise (to anything)."""
    from hyperscan_amd import hs

    db = hs.Database.compile(["alpha\\d+", "beta"], [hs.HS_FLAG_CASELESS, 0], [1, 2])
    blob = db.serialize()
    assert hs.Database.deserialize(blob).serialize() == blob
    for pos in (4, 9, len(blob) // 2, len(blob) - 1):
        bad = bytearray(blob)
        bad[pos] ^= 0x01
        with pytest.raises(hs.HsError) as e:
            hs.Database.deserialize(bytes(bad))
        assert e.value.code == hs.HS_INVALID, pos
    with pytest.raises(hs.HsError):
        hs.Database.deserialize(blob[:-1])  # truncated
    with pytest.raises(hs.HsError):
        hs.Database.deserialize(blob + b"\\0")  # trailing junk


def test_expression_info_reference_table_subset():
    """unit/hyperscan/expr_info.cpp:182-228: the rows of ei_test[] that lie inside the supported
    pattern subset (anchors, alternation of literal-prefixed branches, classes / quantifiers / groups), with and without ext parameters."""
    from hyperscan_amd import hs

    U = 0xFFFFFFFF
    rows = [("abc", None, 3, 3), ("abc.*def", None, 6, U), ("foo.{1,13}bar", None, 7, 19), ("foo.{10,}bar", None, 16, U),
            ("foo.{0,10}bar", None, 6, 16), ("foo.{,10}bar", None, 12, 12), ("foo.{10}bar", None, 16, 16),
            ("abc.*def", dict(max_offset=10), 6, 10), ("abc.*def", dict(min_length=100), 100, U),
            ("abc.*def", dict(min_length=5), 6, U),
            ("abc(def)?", None, 3, 6), ("abc(def){0,3}", None, 3, 12), ("abc(def){1,4}", None, 6, 15),
            ("abc|defghi", None, 3, 6), ("^foo", None, 3, 3), ("^foo.*bar", None, 6, U), ("^foo.*bar?", None, 5, U),
            ("^foo.*bar$", None, 6, U), ("^foobar$", None, 6, 6), ("foobar$", None, 6, 6), ("^.*foo", None, 3, U),
            ("foo\\b", None, 3, 3), ("\\bfoo", None, 3, 3), ("^\\bfoo", None, 3, 3), ("\\Bfoo", None, 3, 3), ("(?m)^foo", None, 3, 3), ("(?m)^\\bfoo", None, 3, 3), ("(^|\n)foo", None, 3, 4), ("(^\n|)foo", None, 3, 4),
            ("(foo|bar\\z)", None, 3, 3), ("(foo|bar)\\z", None, 3, 3),
            ("^abc.*def", dict(max_offset=10), 6, 10), ("^abc.*def", dict(min_length=100), 100, U)]
    for pat, ext, mn, mx in rows:
        assert hs.expression_info(pat, 0, hs.ExprExt.make(**ext) if ext else None) == (mn, mx), pat
    # rows outside the subset are refused, not mis-measured
    for pat in ("(?m)\\b$", "", "^", "$", "\\b$", "\\A", "\\z", "\\Z"):
        with pytest.raises(hs.HsError):
            hs.expression_info(pat)


def test_headers_are_valid_c_and_the_example_links(tmp_path):
    """include/*.h must be usable from plain C (the drop-in boundary is a C ABI), and
    examples/simplegrep.c must build and link against libhsgpu.so with nothing but gcc."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text('#include "hsgpu.h"\n#include "hs_gpu.h"\n'
                   "int main(void) { hsgpu_lit_t l; hsgpu_match_t m; hsgpu_accel_t a; hsgpu_pair_t p; hs_expr_ext_t e;\n"
                   "  (void)l; (void)m; (void)a; (void)p; (void)e; return hs_version() == 0 || hsgpu_version() == 0; }\n")
    lib_dir = os.path.join(ROOT, "hyperscan_amd", "lib")
    inc = os.path.join(ROOT, "include")
    for std in ("-std=c99", "-std=c11"):
        r = subprocess.run(["gcc", std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(tmp_path / "abi"),
                            "-L", lib_dir, "-lhsgpu", f"-Wl,-rpath,{lib_dir}"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I", inc, os.path.join(ROOT, "examples", "simplegrep.c"), "-o",
                        str(tmp_path / "simplegrep"), "-L", lib_dir, "-lhsgpu", f"-Wl,-rpath,{lib_dir}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # without a pattern argument the example prints its usage and fails before touching a device
    r = subprocess.run([str(tmp_path / "simplegrep")], capture_output=True, text=True)
    assert r.returncode != 0 and "sage" in (r.stdout + r.stderr)


def test_vectored_mode_without_gpu():
    """HS_MODE_VECTORED compiles, survives serialisation, names itself in the info string, and the
    scan entry points insist on the matching mode before touching a device
    (src/runtime.c:343-345,1127-1129)."""
    from hyperscan_amd import hs

    lib = hs._lib()
    vdb = hs.Database.compile(["^foo.*bar"], [hs.HS_FLAG_DOTALL], [0], mode=hs.HS_MODE_VECTORED)
    bdb = hs.Database.compile(["^foo.*bar"], [hs.HS_FLAG_DOTALL], [0])
    info = C.c_char_p()
    for db, word in ((vdb, b"VECTORED"), (bdb, b"BLOCK"), (hs.Database.deserialize(vdb.serialize()), b"VECTORED")):
        assert lib.hs_database_info(db._h, C.byref(info)) == 0 and word in info.value
    blob = vdb.serialize()
    assert lib.hs_serialized_database_info(blob, len(blob), C.byref(info)) == 0 and b"VECTORED" in info.value
    fake_scratch = C.create_string_buffer(256)  # never dereferenced: the mode check comes first
    cb = hs.MATCH_CB(lambda *a: 0)
    seg = (C.c_char_p * 1)(b"foobar")
    ln = (C.c_uint * 1)(6)
    lib.hs_scan_vector.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_void_p, hs.MATCH_CB,
                                   C.c_void_p]
    assert lib.hs_scan_vector(bdb._h, seg, ln, 1, 0, fake_scratch, cb, None) == hs.HS_DB_MODE_ERROR
    assert lib.hs_scan(vdb._h, b"foobar", 6, 0, fake_scratch, cb, None) == hs.HS_DB_MODE_ERROR
    assert lib.hs_scan_vector(vdb._h, None, ln, 1, 0, fake_scratch, cb, None) == hs.HS_INVALID
    assert lib.hs_scan_vector(vdb._h, seg, None, 1, 0, fake_scratch, cb, None) == hs.HS_INVALID
    assert lib.hs_scan_vector(vdb._h, seg, ln, 1, 0, None, cb, None) == hs.HS_INVALID
    with pytest.raises(hs.HsError):
        hs.Database.compile(["foo"], mode=hs.HS_MODE_STREAM)


def test_host_record_sort_matches_lexsort():
    """hsgpu_match_sort_host (the delivery order of hsgpu_hwlm_exec_batch): serial below 64 Ki
    records, chunked over threads and merged above; equal to a lexicographic sort at every size."""
    from hyperscan_amd.hwlm import MATCH_DTYPE

    lib = _native.load_library()
    lib.hsgpu_match_sort_host.argtypes = [C.c_void_p, C.c_size_t]
    lib.hsgpu_match_sort_host.restype = None
    rng = np.random.default_rng(8)
    for n in (0, 1, 2, 1000, 65535, 65536, 65537, 100_003, 300_000, 1_000_000):
        r = np.zeros(n, dtype=MATCH_DTYPE)
        r["block"] = rng.integers(0, max(1, n // 50), n)
        r["end"] = rng.integers(0, 1500, n)
        r["lit"] = rng.integers(0, 1000, n)
        r["id"] = r["lit"] * 7 + 1
        if n > 4:  # nearly sorted input and runs of equal keys, as the device hands them over
            r[: n // 2] = np.sort(r[: n // 2], order=["block", "end", "lit"])
            r[n // 2: n // 2 + 3] = r[0]
        want = r[np.lexsort((r["lit"], r["end"], r["block"]))]
        lib.hsgpu_match_sort_host(r.ctypes.data, n)
        assert np.array_equal(r["block"], want["block"]) and np.array_equal(r["end"], want["end"])
        assert np.array_equal(r["lit"], want["lit"]) and np.array_equal(r["id"], want["id"])


def test_key_gate_of_large_sets_admits_every_key():
    """HSGPU_F_GATE (sets of >= 2048 keys without a 2-byte table): the confirm kernel probes an exact table only
    for keys whose bit is set in the 64 Kbit gate, so every key of every literal, case variants included, must have
    its bit -- check_table_covers models the gate -- and the gate must be selective (mostly empty)."""
    rng = np.random.default_rng(77)
    lits = random_literals(rng, 3000, 3, 8, nocase_frac=0.3)
    h = check_table_covers(lits, 0)
    assert h["flags"] & F_GATE and not h["flags"] & (F_C | F_BLOOM)
    fill = sum(bin(int(w)).count("1") for w in h["c2bits"]) / 65536.0
    assert 0.01 < fill < 0.2, fill
    # NO_GATE: the section stays empty and the flag off
    h2 = parse(H.hwlm_build(lits, NO_GATE).serialize())
    assert not h2["flags"] & (F_GATE | F_BLOOM) and not any(int(w) for w in h2["c2bits"])


def test_bloom_gate_of_large_stride_1_sets_admits_every_window():
    """HSGPU_F_BLOOM (round 5, opt-in: HSGPU_BUILD_FORCE_BLOOM; stride-1 sets of >= 2048 keys without a 2-byte table): in place of the key gate the confirm kernel
    probes a Bloom filter over the literals' FULL-window keys -- five bytes where a literal has them -- so every window a literal
    can match, case variants and msk / cmp wildcards included, must pass its group (check_table_covers models the three groups),
    the planes must be selective, and stride-2 tables keep the key gate."""
    rng = np.random.default_rng(78)
    lits = random_literals(rng, 3000, 3, 8, nocase_frac=0.3)
    n0 = len(lits)
    # wildcards inside the last five bytes and in front of them; a mask longer than the string
    lits += [H.HwlmLiteral("abcde", False, n0, msk=b"\xff\xf0\xff\xff\xff", cmp=b"a\x60cde"),
             H.HwlmLiteral("qrstuv", True, n0 + 1, msk=b"\x00\xff\xff\xff\xff\xff", cmp=b"\x00RSTUV"),
             H.HwlmLiteral("wxyz", False, n0 + 2, msk=b"\x0f\xff\xff\xff\xff", cmp=b"\x01wxyz"),
             H.HwlmLiteral("mnop", False, n0 + 3, msk=b"\xff\xff\xff\xff\xff", cmp=b"Zmnop")]
    h = check_table_covers(lits, FORCE_BLOOM)
    assert h["flags"] & F_BLOOM and not h["flags"] & (F_GATE | F_C | F_S2)
    fills = [sum(bin(int(w)).count("1") for w in h["c2bits"][j * 1024:(j + 1) * 1024]) / 32768.0 for j in range(3)]
    assert all(0.02 < f < 0.3 for f in fills), fills
    # case-blind and not: the same with no caseless literal in the set
    caseful = random_literals(rng, 2500, 3, 8, nocase_frac=0)
    h2 = check_table_covers(caseful, FORCE_BLOOM)
    assert h2["flags"] & F_BLOOM and not h2["flags"] & F_BLIND
    # literals of >= 4 bytes: stride 2, and the key gate stays
    h3 = check_table_covers(random_literals(rng, 3000, 4, 8, nocase_frac=0.3), FORCE_BLOOM)
    assert h3["flags"] & F_S2 and h3["flags"] & F_GATE and not h3["flags"] & F_BLOOM


def test_every_built_table_reloads():
    """hsgpu_hwlm_deserialize(hsgpu_hwlm_serialize(t)) for every layout the compiler can choose, the opt-in pair
    filter on small sets included (found by tools/asan_table_harness.cpp: a pair table of nine literals carried
    the folded-keys flag, which its own validation refuses)."""
    rng = np.random.default_rng(146)
    FORCE_PAIR = 1024
    built = 0
    for flags in (0, FORCE_REPL, FORCE_HASHED | FORCE_K2, FORCE_S1, FORCE_S2, NO_FOLD, FORCE_PAIR, FORCE_PAIR | FORCE_BLIND, NO_GATE, FORCE_BLOOM):
        for n in (1, 9, 60, 700, 2500):
            lits = random_literals(rng, n, 3 if flags & FORCE_PAIR else 1, 8, nocase_frac=0.3)
            try:
                t = H.hwlm_build(lits, flags)
            except H.HsgpuError:
                continue  # (a set the forced layout cannot hold)
            blob = t.serialize()
            t2 = H.HwlmTable.deserialize(blob)
            assert t2.serialize() == blob
            built += 1
    assert built >= 30


def test_replay_rules_equal_the_reference_under_changing_group_masks():
    """The sequential half of hwlmExec -- group gate against the callback's last return value, noruns against the
    last delivered id, stop on 0 (fdr_confirm_runtime.h:69-96, fdr.c:719-721) -- runs on the host in
    hsgpu_hwlm_replay over the group-independent superset the device emits. Here the superset comes from the oracle
    (the same literals without groups / noruns) and the result is compared with the COMPILED REFERENCE driven by the
    same callback: masks that change at thresholds of `end`, and termination. No GPU involved."""
    from tests import oracle_binding as ob
    from tests.util import random_corpus, random_literals

    ob.require_ref()
    rng = np.random.default_rng(61)
    base = random_literals(rng, 90, 2, 8, nocase_frac=0.3)
    lits, raw = [], []
    for i, l in enumerate(base):  # ids = literal index, so that a record's `lit` is its id
        groups = [H.HWLM_ALL_GROUPS, 0x1, 0x2, 0x4, 0x6][int(rng.integers(0, 5))]
        noruns = bool(rng.random() < 0.25)
        lits.append(H.HwlmLiteral(l.s, l.nocase, i, noruns=noruns, groups=groups))
        raw.append(H.HwlmLiteral(l.s, l.nocase, i))
    corpus = random_corpus(rng, 120_000, lits, plant_every=60)
    corpus[3000:3300] = np.frombuffer(bytes(lits[1].s) * 300, dtype=np.uint8)[:300]  # a run: noruns matters
    superset = ob.Oracle(raw).collect(corpus)
    assert len(superset) > 1500
    ends = sorted(e for e, _ in superset)
    unique_ends = [e for k, e in enumerate(ends) if (k == 0 or ends[k - 1] != e) and (k + 1 == len(ends) or ends[k + 1] != e)]
    t = H.hwlm_build(lits)
    ref = ob.Reference(lits)
    recs = np.zeros(len(superset), dtype=hw.MATCH_DTYPE)
    recs["end"] = [e for e, _ in superset]
    recs["id"] = [i for _, i in superset]
    recs["lit"] = recs["id"]
    order = np.lexsort((recs["lit"], recs["end"]))
    recs = np.ascontiguousarray(recs[order])

    def run_ours(policy, groups):
        out = []

        def cb(e, i, _c):
            out.append((e, i))
            return policy(e)

        rv = t._lib.hsgpu_hwlm_replay(t._h, recs.ctypes.data, recs.size, _native.HWLM_CB(cb), None, groups)
        return rv, out

    def run_ref(policy, groups):
        out = []

        def cb(e, i):
            out.append((e, i))
            return policy(e)

        return ref.exec(corpus, 0, cb, groups), out

    # thresholds at ends that carry exactly one match: the order inside one `end` (engine specific) cannot matter
    t1, t2, t3 = (unique_ends[len(unique_ends) * k // 4] for k in (1, 2, 3))
    policies = [lambda e: H.HWLM_ALL_GROUPS,
                lambda e: 0x1 if e < t1 else (0x6 if e < t2 else H.HWLM_ALL_GROUPS),
                lambda e: 0x2 if e < t2 else 0x4,
                lambda e: H.HWLM_ALL_GROUPS if e < t3 else 0,   # terminate at the first match from t3 on
                lambda e: 0x4 if e < t1 else 0]
    for pi, policy in enumerate(policies):
        for groups in (H.HWLM_ALL_GROUPS, 0x3, 0x4):
            rv_o, got = run_ours(policy, groups)
            rv_r, want = run_ref(policy, groups)
            assert rv_o == rv_r, (pi, hex(groups))
            assert sorted(got) == sorted(want), (pi, hex(groups), len(got), len(want))
            assert [e for e, _ in got] == sorted(e for e, _ in got)


def test_replay_batch_restarts_the_rules_in_every_block_like_the_reference():
    """hsgpu_hwlm_replay_batch over the records of a whole batch: the sequential rules start afresh in every block (a
    block is one hwlmExec call), a callback that returns 0 ends its own block only. Compared block by block with the
    compiled reference driven by the same callback; the superset comes from the oracle. No GPU involved."""
    from tests import oracle_binding as ob
    from tests.util import random_blocks, random_corpus, random_literals

    ob.require_ref()
    rng = np.random.default_rng(62)
    base = random_literals(rng, 60, 2, 8, nocase_frac=0.3)
    lits = [H.HwlmLiteral(l.s, l.nocase, i, noruns=bool(i % 4 == 0), groups=[H.HWLM_ALL_GROUPS, 0x1, 0x2][i % 3]) for i, l in enumerate(base)]
    raw = [H.HwlmLiteral(l.s, l.nocase, i) for i, l in enumerate(base)]
    corpus = random_corpus(rng, 60_000, lits, plant_every=50)
    off = random_blocks(rng, corpus.size, mean_len=900)
    sup = ob.Oracle(raw).collect_blocks(corpus, off)
    recs = np.zeros(sup.size, dtype=hw.MATCH_DTYPE)
    recs["block"], recs["end"], recs["id"], recs["lit"] = sup["block"], sup["end"], sup["id"], sup["id"]
    recs = np.ascontiguousarray(recs[np.lexsort((recs["lit"], recs["end"], recs["block"]))])
    t = H.hwlm_build(lits)
    ref = ob.Reference(lits)
    # per block: switch the mask at the block's median match end if that end carries one match only, stop at 3/4
    by_block = {}
    for b, e in zip(recs["block"].tolist(), recs["end"].tolist()):
        by_block.setdefault(b, []).append(e)
    cuts = {}
    for b, es in by_block.items():
        uniq = [e for e in es if es.count(e) == 1]
        if len(uniq) >= 4:
            cuts[b] = (uniq[len(uniq) // 2], uniq[3 * len(uniq) // 4])
    assert len(cuts) > 20
    cur = [0]

    def policy(e):
        lo, hi = cuts.get(cur[0], (1 << 62, 1 << 62))
        return 0x1 if e < lo else (H.HWLM_ALL_GROUPS if e < hi else 0)

    # ours: one call over all records; the callback learns its block from a cursor advanced by the record order
    delivered = []
    blocks_in_order = recs["block"].tolist()
    state = {"k": 0}

    def cb_ours(e, i, _c):
        # find the record this callback belongs to: the next one at or after the cursor with this (end, id)
        k = state["k"]
        while not (int(recs["end"][k]) == e and int(recs["id"][k]) == i):
            k += 1
        state["k"] = k + 1
        cur[0] = blocks_in_order[k]
        delivered.append((blocks_in_order[k], e, i))
        return policy(e)

    n_term = C.c_size_t(0)
    rv = t._lib.hsgpu_hwlm_replay_batch(t._h, recs.ctypes.data, recs.size, C.cast(_native.HWLM_CB(cb_ours), C.c_void_p), None,
                                        H.HWLM_ALL_GROUPS, C.byref(n_term))
    assert rv == 0
    want, want_term = [], 0
    for b in range(off.size - 1):
        cur[0] = b
        blk = np.ascontiguousarray(corpus[int(off[b]):int(off[b + 1])])
        out = []

        def cb_ref(e, i):
            out.append((b, e, i))
            return policy(e)

        want_term += 1 if ref.exec(blk, 0, cb_ref, H.HWLM_ALL_GROUPS) == 1 else 0
        want += out
    assert sorted(delivered) == sorted(want) and len(want) > 500
    assert n_term.value == want_term and want_term > 10


def test_replay_batch_mt_delivers_what_the_single_threaded_walk_delivers():
    """hsgpu_hwlm_replay_batch_mt: contiguous block ranges on several threads, a context per thread. The callbacks of
    one block stay on one thread in order (each thread checks that its blocks and, inside a block, its ends never go
    backwards), the multiset delivered and the number of terminated blocks equal the single-threaded walk's; with
    noruns literals, group masks that the callback changes and a stop rule inside blocks. No GPU involved."""
    import threading

    from tests import oracle_binding as ob
    from tests.util import random_blocks, random_corpus, random_literals

    rng = np.random.default_rng(63)
    base = random_literals(rng, 40, 2, 8, nocase_frac=0.3)
    lits = [H.HwlmLiteral(l.s, l.nocase, i, noruns=bool(i % 4 == 0), groups=[H.HWLM_ALL_GROUPS, 0x1, 0x2][i % 3]) for i, l in enumerate(base)]
    raw = [H.HwlmLiteral(l.s, l.nocase, i) for i, l in enumerate(base)]
    corpus = random_corpus(rng, 80_000, lits, plant_every=40)
    off = random_blocks(rng, corpus.size, mean_len=300)
    sup = ob.Oracle(raw).collect_blocks(corpus, off)
    recs = np.zeros(sup.size, dtype=hw.MATCH_DTYPE)
    recs["block"], recs["end"], recs["id"], recs["lit"] = sup["block"], sup["end"], sup["id"], sup["id"]
    recs = np.ascontiguousarray(recs[np.lexsort((recs["lit"], recs["end"], recs["block"]))])
    t = H.hwlm_build(lits)

    def policy(e):  # depends on the end offset only: the same decision whichever thread asks
        return 0 if e % 97 == 0 else (0x1 if e % 5 == 0 else H.HWLM_ALL_GROUPS)

    single = []
    n1 = C.c_size_t(0)
    cb1 = _native.HWLM_CB(lambda e, i, _c: (single.append((e, i)), policy(e))[1])
    assert t._lib.hsgpu_hwlm_replay_batch(t._h, recs.ctypes.data, recs.size, C.cast(cb1, C.c_void_p), None, H.HWLM_ALL_GROUPS,
                                          C.byref(n1)) == 0
    for threads in (1, 3, 8):
        per = {}
        lock = threading.Lock()

        def cb(e, i, ctx):
            with lock:
                per.setdefault(ctx, []).append((e, i))
            return policy(e)

        ccb = _native.HWLM_CB(cb)
        ctxs = (C.c_void_p * threads)(*[1000 + k for k in range(threads)])
        nt = C.c_size_t(0)
        t._lib.hsgpu_hwlm_replay_batch_mt.restype = C.c_int
        t._lib.hsgpu_hwlm_replay_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint64,
                                                      C.c_void_p]
        assert t._lib.hsgpu_hwlm_replay_batch_mt(t._h, recs.ctypes.data, recs.size, C.cast(ccb, C.c_void_p), ctxs, threads,
                                                 H.HWLM_ALL_GROUPS, C.byref(nt)) == 0
        # contexts 1000, 1001, ... hold consecutive pieces of the single-threaded delivery
        joined = [x for k in range(threads) for x in per.get(1000 + k, [])]
        assert joined == single and nt.value == n1.value and n1.value > 5
        assert len(per) == min(threads, len(per)) and (threads == 1 or len(per) > 1)
    assert hw.hwlm_replay_count_mt(t, recs, 4) > 0


def test_confirm_part_mappings_are_permutations():
    """The two worker -> part mappings of round 5's confirm kernel, restated (csrc/scan_device.h, hwlm_confirm_kernel): dense scans
    spread a worker's K parts over the corpus row by row (part = row * workers + (worker + 2731 * row) % workers, parts past the
    end skipped); ordinary scans with two parts per share and one per worker give the halves of a share to workers of mirrored
    dispatch ranks on the same CU slot. Either way every part must be taken exactly once -- the first build of the dense mapping
    used a multiplier that shared a factor with the part count and visited a third of the parts three times (profiles/
    r05_flood.txt); the GPU tests caught it, this one would have caught it here."""
    import numpy as np

    for n_parts, workers in [(18432, 6144), (8192, 4096), (36864, 5120), (34816, 5120), (20480, 5120), (7, 4), (12289, 6144), (5120, 5120), (3, 1024)]:
        k = -(-n_parts // workers)
        w = np.arange(workers, dtype=np.int64)
        taken = np.concatenate([(row * workers + (w + 2731 * row) % workers) for row in range(k)])
        taken = taken[taken < n_parts]
        assert taken.size == n_parts and np.array_equal(np.sort(taken), np.arange(n_parts)), (n_parts, workers)
    for ranks, cus, waves in [(8, 256, 4), (6, 256, 4), (2, 304, 4), (8, 64, 8)]:
        wg = np.arange(ranks * cus)
        rank, slot = wg // cus, wg % cus
        old = rank < ranks // 2
        low = np.where(old, rank, ranks - 1 - rank)
        parts = []
        for wave in range(waves):
            parts.append(2 * ((low * cus + slot) * waves + wave) + np.where(old, 0, 1))
        parts = np.concatenate(parts)
        assert np.array_equal(np.sort(parts), np.arange(ranks * cus * waves)), (ranks, cus, waves)
        # the older worker's share of the half batches: above one half, at most 1/2 + 0.053, the same for both halves of a share
        num = 32768 + 3473 * (ranks - 1 - 2 * low) // max(ranks - 1, 1)
        assert num.min() > 32768 and num.max() <= 32768 + 3473
