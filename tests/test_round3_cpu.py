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
