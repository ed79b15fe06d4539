"""The public hs_* surface (include/hs_gpu.h) on the GPU engine, in the manner of the
reference's API-level tests: unit/hyperscan/literals.cpp (seeded random literal sets,
every planted corpus must match), single.cpp / multi.cpp, order.cpp (offsets
non-decreasing), arg_checks.cpp (error codes), serialize.cpp; plus the Rose-lite
confirm (literal prefix + regex tail) against a brute-force oracle built on Python's re."""
import os
import re
import subprocess

import numpy as np
import pytest

from hyperscan_amd import hs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(db, data, scratch):
    out = []
    rv = hs.scan(db, data, scratch, lambda i, f, t: out.append((i, f, t)) and False)
    assert rv == hs.HS_SUCCESS
    return out


def brute_literal(data, lits, flags, ids):
    want = set()
    for s, fl, i in zip(lits, flags, ids):
        hay, needle = (data.upper(), s.upper()) if fl & hs.HS_FLAG_CASELESS else (data, s)
        k = hay.find(needle)
        first = True
        while k >= 0:
            if not (fl & hs.HS_FLAG_SINGLEMATCH) or first:
                want.add((i, k if fl & hs.HS_FLAG_SOM_LEFTMOST else 0, k + len(s)))
            first = False
            k = hay.find(needle, k + 1)
    return sorted(want, key=lambda e: (e[2], e[0]))


@pytest.mark.parametrize("count,lo,hi", [(1, 3, 10), (10, 3, 10), (100, 3, 10), (500, 10, 100), (2000, 3, 10)])
def test_random_literal_sets(count, lo, hi):
    # unit/hyperscan/literals.cpp:160-245 (rng.seed(29785643), sets of {1,10,100,500,10000} x len ranges)
    rng = np.random.default_rng(29785643 + count)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    lits = list({bytes(rng.choice(alpha, int(rng.integers(lo, hi + 1)))) for _ in range(count)})
    flags = [int(rng.choice([0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_SINGLEMATCH, hs.HS_FLAG_SOM_LEFTMOST])) for _ in lits]
    ids = list(range(len(lits)))
    db = hs.Database.compile_lit(lits, flags, ids)
    scratch = hs.HsScratch(db)
    data = bytearray(rng.choice(alpha, 20000).tobytes())
    for s in lits[:: max(1, len(lits) // 60)]:
        p = int(rng.integers(0, len(data) - len(s)))
        data[p:p + len(s)] = s
    data = bytes(data)
    got = collect(db, data, scratch)
    want = brute_literal(data, lits, flags, ids)
    assert sorted(got, key=lambda e: (e[2], e[0])) == want
    tos = [t for _i, _f, t in got]
    assert tos == sorted(tos)  # unit/hyperscan/order.cpp:43-53
    assert len(got) >= min(20, len(lits))  # planted literals may overwrite each other


def test_single_pattern_and_termination():
    db = hs.Database.compile(["hyperscan"], [hs.HS_FLAG_DOTALL])
    scratch = hs.HsScratch(db)
    data = b"xx hyperscan yy hyperscan zz"
    assert collect(db, data, scratch) == [(0, 0, 12), (0, 0, 25)]
    seen = []
    rv = hs.scan(db, data, scratch, lambda i, f, t: seen.append(t) or True)
    assert rv == hs.HS_SCAN_TERMINATED and seen == [12]
    # block shorter than the pattern: nothing to do (src/runtime.c:346)
    assert collect(db, b"hyper", scratch) == []


def test_arg_checks():
    # unit/hyperscan/arg_checks.cpp
    with pytest.raises(hs.HsError) as e:
        hs.Database.compile(["foo"], mode=hs.HS_MODE_STREAM)
    assert e.value.code == hs.HS_COMPILER_ERROR
    with pytest.raises(hs.HsError) as e:
        hs.Database.compile(["(\\w|.)+"])  # no mandatory literal at the top level, no small class either
    assert e.value.code == hs.HS_COMPILER_ERROR and e.value.expression == 0
    with pytest.raises(hs.HsError) as e:
        hs.Database.compile(["good", "[a-z]+(tail"])
    assert e.value.expression == 1
    with pytest.raises(hs.HsError) as e:
        hs.Database.compile_lit([b""])
    assert "empty" in e.value.message
    with pytest.raises(hs.HsError):
        hs.Database.compile_lit([b"abc"], [hs.HS_FLAG_DOTALL])  # compiler.cpp:405-419
    db = hs.Database.compile_lit([b"abc"])
    scratch = hs.HsScratch(db)
    lib = hs._lib()
    cb = hs.MATCH_CB(lambda *a: 0)
    assert lib.hs_scan(db._h, None, 3, 0, scratch._h, cb, None) == hs.HS_INVALID
    assert lib.hs_scan(db._h, b"abc", 3, 0, None, cb, None) == hs.HS_INVALID
    assert lib.hs_scan(None, b"abc", 3, 0, scratch._h, cb, None) == hs.HS_INVALID
    assert b"hsgpu" in lib.hs_version()


def test_serialize_roundtrip():
    db = hs.Database.compile(["needle\\d+", "hay"], [0, hs.HS_FLAG_CASELESS], [7, 9])
    blob = db.serialize()
    db2 = hs.Database.deserialize(blob)
    s1, s2 = hs.HsScratch(db), hs.HsScratch(db2)
    data = b"HAY needle123 hay needle9"
    assert collect(db, data, s1) == collect(db2, data, s2)
    assert db.size() > 0
    with pytest.raises(hs.HsError):
        hs.Database.deserialize(b"garbage!")


def brute_tail(data, lit, tail, flags, pid):
    """All (id, 0, to): `to` such that data[k:to] fullmatches lit + tail for an occurrence k of lit."""
    rf = (re.I if flags & hs.HS_FLAG_CASELESS else 0) | (re.S if flags & hs.HS_FLAG_DOTALL else 0)
    tre = re.compile(tail.encode("latin-1"), rf)
    hay, needle = (data.upper(), lit.upper()) if flags & hs.HS_FLAG_CASELESS else (data, lit)
    out = set()
    k = hay.find(needle)
    while k >= 0:
        s = k + len(lit)
        for to in range(s, len(data) + 1):
            if tre.fullmatch(data, s, to):
                out.add((pid, 0, to))
        k = hay.find(needle, k + 1)
    return out


TAILS = [r"[a-z]+\d", r"\s+\w{2,8}=", r".{0,16}END", r"x?y*z", r"[^\n]{3}", r"\d{2,}", r"[A-F0-9]{4}:", r"\.\w+"]


def test_literal_prefix_plus_tail_patterns():
    # config 5 shape: LIT_k<tail>, GPU literal hits feeding the host confirm
    rng = np.random.default_rng(77)
    lits = [b"GET /", b"user=", b"Content-Length", b"abcdefghijkl", b"Zq"]
    pats, flags, ids, parts = [], [], [], []
    for n, tail in enumerate(TAILS):
        lit = lits[n % len(lits)]
        fl = [0, hs.HS_FLAG_CASELESS, hs.HS_FLAG_DOTALL][n % 3]
        pats.append(re.escape(lit.decode()).replace("\\ ", " ").replace("\\-", "-").replace("\\=", "=") + tail)
        flags.append(fl)
        ids.append(100 + n)
        parts.append((lit, tail, fl, 100 + n))
    db = hs.Database.compile(pats, flags, ids)
    scratch = hs.HsScratch(db)
    words = [b"GET /", b"get /", b"user=", b"USER=", b"Content-Length", b"abcdefghijkl", b"Zq", b"abc12", b"  key=",
             b" END", b"xyyz", b"z", b"\n", b"0042", b"BEEF:", b".html", b"   ", b"q9", b"END"]
    data = b"".join(words[int(i)] for i in rng.integers(0, len(words), 900))
    want = set()
    for lit, tail, fl, pid in parts:
        want |= brute_tail(data, lit, tail, fl, pid)
    got = collect(db, data, scratch)
    assert set(got) == want and len(got) == len(want) and len(want) > 50
    # batched form agrees with per-block hs_scan
    off = np.array([0, 500, 500, 1800, len(data)], dtype=np.uint64)
    ev = []
    assert hs.scan_batch(db, data, off, scratch, lambda b, i, f, t: ev.append((b, i, f, t)) and False) == 0
    for b in range(4):
        blk = data[int(off[b]):int(off[b + 1])]
        assert sorted(e[1:] for e in ev if e[0] == b) == sorted(collect(db, blk, scratch))


def test_ext_offset_and_length_bounds():
    """hs_compile_ext_multi: min_offset / max_offset bound `to`, min_length bounds to - start
    (unit/hyperscan/extparam.cpp's shape, on the literal + tail subset)."""
    data = b"..foo1.....foo22....foo333...foo4444"
    base = hs.Database.compile(["foo\\d+"], [0], [5])
    sb = hs.HsScratch(base)
    every = collect(base, data, sb)
    assert len(every) == 10
    for kw in (dict(min_offset=15), dict(max_offset=20), dict(min_offset=10, max_offset=30), dict(min_length=6),
               dict(min_length=5, max_offset=28)):
        db = hs.Database.compile_ext(["foo\\d+"], [0], [5], [hs.ExprExt.make(**kw)])
        sc = hs.HsScratch(db)
        want = []
        for (i, f, t) in every:
            start = data.rfind(b"foo", 0, t)
            if "min_offset" in kw and t < kw["min_offset"]:
                continue
            if "max_offset" in kw and t > kw["max_offset"]:
                continue
            if "min_length" in kw and t - start < kw["min_length"]:
                continue
            want.append((i, f, t))
        assert collect(db, data, sc) == want and want, kw
        db2 = hs.Database.deserialize(db.serialize())
        assert collect(db2, data, hs.HsScratch(db2)) == want


def test_simplegrep_example(tmp_path):
    """Config 1 plumbing: the simplegrep use case (examples/simplegrep.c of the reference)
    through the public API, on a 64 MiB printable-ASCII buffer with 64 planted literals."""
    exe = tmp_path / "simplegrep"
    lib_dir = os.path.join(ROOT, "hyperscan_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-o", str(exe), os.path.join(ROOT, "examples", "simplegrep.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-lhsgpu",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(1)
    data = rng.integers(0x20, 0x7F, 64 << 20, dtype=np.uint8)
    lit = np.frombuffer(b"hyperscan", dtype=np.uint8)
    pos = np.sort(rng.choice((64 << 20) - 64, 64, replace=False))
    for p in pos:
        data[p:p + lit.size] = lit
    f = tmp_path / "corpus.bin"
    data.tofile(f)
    out = subprocess.run([str(exe), "hyperscan", str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    offs = [int(l.rsplit(" ", 1)[1]) for l in out.stdout.splitlines() if l.startswith("Match for pattern")]
    want = sorted(set((pos + lit.size).tolist()) | {m.end() for m in re.finditer(b"hyperscan", data.tobytes())})
    assert offs == want


def test_deserialize_database_at_places_a_header_the_api_follows():
    """hs_deserialize_database_at (src/hs_common.h:147-169, unit/hyperscan/serialize.cpp): caller memory of
    hs_serialized_database_size bytes, 8-byte aligned; a misaligned pointer is HS_BAD_ALIGN; the database scans, sizes,
    serialises again and is released with hs_free_database without the buffer being freed."""
    import ctypes as C

    lits = [b"needle", b"hay"]
    db = hs.Database.compile_lit(lits, [0, hs.HS_FLAG_CASELESS], [7, 8])
    blob = db.serialize()
    lib = hs._lib()
    size = C.c_size_t(0)
    assert lib.hs_serialized_database_size(blob, len(blob), C.byref(size)) == 0 and size.value >= 16
    mem = (C.c_uint64 * ((size.value + 15) // 8))()
    lib.hs_deserialize_database_at.restype = C.c_int
    lib.hs_deserialize_database_at.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    assert lib.hs_deserialize_database_at(blob, len(blob), C.addressof(mem) + 4) == -8  # HS_BAD_ALIGN
    assert lib.hs_deserialize_database_at(blob, len(blob), C.addressof(mem)) == 0
    placed = hs.Database(C.c_void_p(C.addressof(mem)))
    scratch = hs.HsScratch(placed)
    got = collect(placed, b"a needle in the HAY, hay", scratch)
    assert got == collect(db, b"a needle in the HAY, hay", hs.HsScratch(db)) and len(got) == 3
    assert placed.serialize() == blob and placed.size() == db.size()
    placed.close()  # hs_free_database: releases the database proper, leaves the caller's buffer alone
    assert mem[0] & 0xffffffff == 0


@pytest.mark.parametrize("mailbox", [1, 2])
def test_hs_scan_per_packet_through_the_small_batch_server(mailbox):
    """hs_scratch_enable_small_batch_server (include/hs_gpu.h): one hs_scan per packet -- hsbench's block mode,
    tools/hsbench/engine_hyperscan.cpp:132-145 -- served by the scratch's resident workgroup: the same events as without it and as
    one hs_scan_batch over all packets (literals, literal + regex tail, caseless, SOM), termination by the callback, packets above
    the server's 16 KiB (a launch), arguments checked, hs_free_scratch with a server live."""
    rng = np.random.default_rng(404 + mailbox)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789 ._-", dtype=np.uint8)
    pats = [b"needle", b"hay[a-z]{2}ack", b"foo\\d+bar", b"zq7x", b"GET /[a-z]+", b"abc"]
    flags = [0, hs.HS_FLAG_CASELESS, 0, hs.HS_FLAG_SOM_LEFTMOST, hs.HS_FLAG_CASELESS, hs.HS_FLAG_SINGLEMATCH]
    ids = [10, 11, 12, 13, 14, 15]
    plants = [b"needle", b"HAYstACK", b"foo123bar", b"zq7x", b"get /index", b"abcabc", b"hayzzack"]
    db = hs.Database.compile(pats, flags, ids)
    plain, served = hs.HsScratch(db), hs.HsScratch(db)
    served.enable_server(mailbox, idle_us=2000)
    packets = []
    for k in range(160):
        n = int(rng.integers(20, 1500)) if k % 40 else 20000  # (every 40th is too large for the server: a launch)
        p = bytearray(rng.choice(alpha, n).tobytes())
        for j in range(int(rng.integers(0, 5))):
            w = plants[int(rng.integers(0, len(plants)))]
            at = int(rng.integers(0, n - len(w)))
            p[at:at + len(w)] = w
        packets.append(bytes(p))
    total_events = 0
    for k, p in enumerate(packets):
        a, b = collect(db, p, plain), collect(db, p, served)
        assert a == b, (k, len(p))
        total_events += len(a)
    assert total_events > 100
    calls, launches = served.server_stats()
    assert calls >= 150 and launches <= 6, (calls, launches)
    # ... and as one batch
    off = np.concatenate([[0], np.cumsum([len(p) for p in packets])]).astype(np.uint64)
    got = []
    assert hs.scan_batch(db, b"".join(packets), off, plain, lambda blk, i, f, t: got.append((blk, i, f, t)) and False) == hs.HS_SUCCESS
    per_call = [(k, i, f, t) for k, p in enumerate(packets) for i, f, t in collect(db, p, served)]
    assert sorted(got) == sorted(per_call)
    # termination: the callback's stop ends the scan (src/runtime.c:177-185)
    hit = next(p for p in packets if len(collect(db, p, plain)) >= 2 and len(p) < 4000)
    seen = []
    assert hs.scan(db, hit, served, lambda i, f, t: seen.append(i) or True) == hs.HS_SCAN_TERMINATED
    assert len(seen) == 1
    lib = hs._lib()
    assert lib.hs_scratch_enable_small_batch_server(None, 1, 0) == hs.HS_INVALID
    assert lib.hs_scratch_enable_small_batch_server(served._h, 3, 0) == hs.HS_INVALID
    served.enable_server(False)
    assert collect(db, packets[1], served) == collect(db, packets[1], plain)
    served.enable_server(mailbox)
    assert collect(db, packets[2], served) == collect(db, packets[2], plain)
    served.close()  # (a server is live: hs_free_scratch ends it)
    plain.close()
