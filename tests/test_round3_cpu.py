"""Round-3 host logic that needs no GPU: which expressions the class-sequence front end takes, what it reports about
them, serialisation of databases without a literal table, hs_deserialize_database_at's argument checks, and the bench's
shard / CPU-quota helpers."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from hyperscan_amd import hs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("expr,width", [(r"[a-z]{3,}\d+", 4), (r"\s+[A-Z]{2,}", 3), (r".{3,}[\n]+", 4), (r"[^a]{16,}\w{1,}", 17),
                                         (r"[a-f0-9]+[g-z]+", 2)])
def test_class_sequences_compile_and_report_their_widths(expr, width):
    db = hs.Database.compile([expr], [hs.HS_FLAG_DOTALL], [1])
    assert db.literals() == []  # nothing for the literal matcher: the expression lives on the class-sequence path
    assert hs.expression_info(expr, hs.HS_FLAG_DOTALL) == (width, 0xffffffff)
    blob = db.serialize()
    again = hs.Database.deserialize(blob)
    assert again.serialize() == blob and again.literals() == []


@pytest.mark.parametrize("expr", [r"[a-z]{3,}", r"[a-z]{3,}\d+x*", r"[a-z]{3,5}\d+", r"[a-z]{17,}\d+", r"[a-z]{0,}\d+", r"[a-z]+?\d+",
                                  r"(?i)[a-z]+\d+", r"^[a-z]+\d+", r"[a-z]+\d+$", r"[a-z]*\d+"])
def test_what_is_not_a_plain_class_sequence_goes_to_the_general_compiler(expr):
    """one class with an open repeat, then another, and nothing else: any other expression is the general compiler's to
    accept (then it is keyed on a literal, possibly a small class standing in for one) or to refuse"""
    try:
        db = hs.Database.compile([expr], [0], [1])
    except hs.HsError as e:
        assert e.code == hs.HS_COMPILER_ERROR
    else:
        assert db.literals(), "compiled without a literal and without being a class sequence?"


def test_class_sequences_refuse_flags_and_ext_parameters_they_do_not_implement():
    def not_a_class_sequence(compile_):
        try:
            db = compile_()
        except hs.HsError as e:
            assert e.code == hs.HS_COMPILER_ERROR
        else:
            assert db.literals()  # the general compiler took it (a small class standing in for the literal)

    for fl in (hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_UTF8):
        not_a_class_sequence(lambda: hs.Database.compile([r"[a-z]{3,}\d+"], [fl], [1]))
    not_a_class_sequence(lambda: hs.Database.compile_ext([r"[a-z]{3,}\d+"], [0], [1], [hs.ExprExt.make(min_offset=10)]))
    # mixed with literal patterns: both kinds in one database, ids shared under the reference's SINGLEMATCH rule
    db = hs.Database.compile([r"[a-z]{3,}\d+", "needle"], [hs.HS_FLAG_SINGLEMATCH, hs.HS_FLAG_SINGLEMATCH], [7, 7])
    assert [l[0] for l in db.literals()] == [b"needle"]
    with pytest.raises(hs.HsError) as ei:
        hs.Database.compile([r"[a-z]{3,}\d+", "needle"], [hs.HS_FLAG_SINGLEMATCH, 0], [7, 7])
    assert "HS_FLAG_SINGLEMATCH whereas previous expression" in str(ei.value)


def test_deserialize_database_at_checks_its_arguments():
    lib = hs._lib()
    db = hs.Database.compile_lit([b"needle"], [0], [1])
    blob = db.serialize()
    mem = (C.c_uint64 * 64)()
    lib.hs_deserialize_database_at.restype = C.c_int
    lib.hs_deserialize_database_at.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    assert lib.hs_deserialize_database_at(None, len(blob), C.addressof(mem)) == hs.HS_INVALID
    assert lib.hs_deserialize_database_at(blob, len(blob), None) == hs.HS_INVALID
    assert lib.hs_deserialize_database_at(blob, len(blob), C.addressof(mem) + 2) == -8  # HS_BAD_ALIGN
    assert lib.hs_deserialize_database_at(blob[:-3], len(blob) - 3, C.addressof(mem)) != 0  # truncated: refused, nothing placed
    assert mem[0] == 0
    assert lib.hs_deserialize_database_at(blob, len(blob), C.addressof(mem)) == 0
    placed = hs.Database(C.c_void_p(C.addressof(mem)))
    assert placed.serialize() == blob and placed.size() == db.size() and [l[0] for l in placed.literals()] == [b"needle"]
    placed.close()
    assert mem[0] & 0xffffffff == 0 and lib.hs_free_database(C.c_void_p(C.addressof(mem))) == hs.HS_INVALID  # freed twice: refused


def test_bench_helpers_shards_and_quota(tmp_path):
    sys.path.insert(0, ROOT)
    import bench

    q, src = bench.cgroup_cpu_quota()
    assert q is None or q > 0
    _l, c, o = bench.build_shards("teddy64", 1 << 20, [0, 1, 2])
    assert c.size == 3 << 20 and int(o[-1]) == 3 << 20 and np.all(np.diff(o.astype(np.int64)) > 0)
    _l, c1, o1 = bench.build_workload("teddy64", 1 << 20, 1)
    k = int(np.searchsorted(o, 1 << 20))
    assert np.array_equal(c[1 << 20: 2 << 20], c1) and np.array_equal(o[k:k + o1.size] - np.uint64(1 << 20), o1)


def test_wide_tables_build_reload_and_hold_every_key():
    """HSGPU_F_WIDE (csrc/table.h): chosen for hashed stride-1 two-bit tables whose kernel runs the 4-byte-key test alone; the
    image reloads, a damaged one is refused, NO_WIDE keeps the 32-bit layout -- and, the filter restated in numpy from table.h,
    every literal's own bytes pass it (a filter may pass too much, never too little)."""
    import hyperscan_amd as H
    from tests.util import random_literals

    FORCE_HASHED, FORCE_K2, FORCE_S1, NO_WIDE, F_WIDE, F_BFOLD, F_BLIND = 2, 4, 16, 2048, 1024, 128, 64
    MUL = 0x9E3779
    rng = np.random.default_rng(5)
    for short in (0, 12):
        base = random_literals(rng, 900, 4, 8, nocase_frac=0.3) + random_literals(rng, short, 3, 3, nocase_frac=0.3)
        lits = [H.HwlmLiteral(l.s, nocase=l.nocase, id=i) for i, l in enumerate(base)]
        t = H.hwlm_build(lits, FORCE_HASHED | FORCE_K2 | FORCE_S1)
        info = t.info()
        assert info["flags"] & F_WIDE and bool(info["flags"] & F_BFOLD) == bool(short) and info["filter_words"] == 32768
        assert not H.hwlm_build(lits, FORCE_HASHED | FORCE_K2 | FORCE_S1 | NO_WIDE).info()["flags"] & F_WIDE
        blob = bytes(t.serialize())
        assert bytes(H.HwlmTable.deserialize(blob).serialize()) == blob
        bad = bytearray(blob)
        bad[len(bad) // 2] ^= 0x40
        with pytest.raises(H.HsgpuError):
            H.HwlmTable.deserialize(bytes(bad))
        # the filter image: header field off_filter (word 13), 2^14 entries {lo, hi}
        off_filter = int(np.frombuffer(blob, dtype="<u4", count=32)[13])
        ent = np.frombuffer(blob, dtype="<u4", count=32768, offset=off_filter).reshape(-1, 2)
        blind = 0xdf if info["flags"] & F_BLIND else 0xff
        for l in lits:
            s = l.s
            b = [c & blind for c in s[-4:]] if len(s) >= 4 else [None] + [c & blind for c in s[-3:]]
            x = b[1] | b[2] << 8 | b[3] << 16
            prod = (x * MUL) & 0xffffffff
            lo, hi = int(ent[prod >> 18, 0]), int(ent[prod >> 18, 1])
            assert (hi >> (prod & 31)) & 1, l.s
            if b[0] is None:
                assert lo == 0xffffffff, l.s  # a folded 3-byte key: any byte in front of it
            else:
                for c in ({b[0], b[0] | 0x20} if l.nocase and blind == 0xdf else {b[0]}):  # every admissible b3 after blinding
                    assert (lo >> ((c & blind) & 31)) & 1, l.s
