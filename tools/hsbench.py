#!/usr/bin/env python3
"""tools/hsbench.py -- hsbench for the GPU engine: same inputs, same protocol, same
output lines as the reference's benchmarker (tools/hsbench/main.cpp), block mode.

  inputs   -e FILE   pattern file, one `ID:/regex/flags{ext}` per line, `#` comments
                     (util/ExpressionParser.rl: flags i s m H V W 8 P L C Q; ext
                     min_offset/max_offset/min_length/edit_distance/hamming_distance)
           -c FILE   SQLite corpus: table chunk(id, stream_id, data), read `ORDER BY id`
                     (tools/hsbench/data_corpus.cpp:69-110, scripts/CorpusBuilder.py:19-25);
                     in block mode (-N) every chunk is one hs_scan call = one block here
           -n N      repeats (default 20), -N block mode (the only mode of this engine),
           --literal-on  patterns are pure literals -> hs_compile_lit_multi (main.cpp:226,479)
  protocol corpus loaded into memory first, then N timed repeats of "scan every block",
           counting callback (engine_hyperscan.cpp:89-97), per-repeat match counts must agree
           (main.cpp:502-528,720-724,778-787)
  output   displayResults (main.cpp:771-856) and printStats (engine_hyperscan.cpp:246-277)

What is timed: the default is the whole hs_scan_batch call per repeat (H2D of the corpus,
GPU literal scan, D2H of records, host confirm, counting callback). `--resident` keeps the
corpus in HBM and times the device pipeline alone (pure-literal pattern sets only) -- the
figure bench.py reports. `--one-scan-per-block` is the reference's own loop (one hs_scan per block), `--server 1|2` with the
scratch's small-batch server (include/hs_gpu.h) serving those calls without a launch each. Also: `--make-corpus KIND -o FILE --mib M` writes a synthetic corpus
in the same SQLite format (packets | lines), usable by the reference's hsbench as well."""
import argparse
import ctypes as C
import os
import re
import sqlite3
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FLAG_CHARS = {"i": 1, "s": 2, "m": 4, "H": 8, "V": 16, "8": 32, "W": 64, "P": 128, "L": 256, "C": 512, "Q": 1024}
EXT_KEYS = {"min_offset": 1, "max_offset": 2, "min_length": 4, "edit_distance": 8, "hamming_distance": 16}


class ParseError(ValueError):
    pass


def parse_expression_line(line):
    """`ID:/regex/flags{key=val,...}` -> (id, regex bytes, flags, ext dict) or None for blanks/comments."""
    line = line.rstrip("\r\n")
    if not line.strip() or line.lstrip().startswith("#"):
        return None
    m = re.match(r"^\s*(\d+):(.*)$", line)
    if not m:
        raise ParseError(f"no 'ID:' prefix: {line!r}")
    pid, rest = int(m.group(1)), m.group(2)
    if not rest.startswith("/"):
        raise ParseError(f"pattern must be delimited by '/': {line!r}")
    end = rest.rfind("/")
    if end <= 0:
        raise ParseError(f"missing closing '/': {line!r}")
    regex, tail = rest[1:end], rest[end + 1:]
    ext = {}
    em = re.search(r"\{([^}]*)\}\s*$", tail)
    if em:
        for kv in filter(None, (x.strip() for x in em.group(1).split(","))):
            k, _, v = kv.partition("=")
            if k.strip() not in EXT_KEYS or not v.strip().isdigit():
                raise ParseError(f"bad extended parameter {kv!r}")
            ext[k.strip()] = int(v)
        tail = tail[: em.start()]
    flags = 0
    for ch in tail.strip():
        if ch == "O":
            continue  # the reference's "must_be_ordered" marker: a test hint, no compile flag
        if ch not in FLAG_CHARS:
            raise ParseError(f"unknown flag {ch!r} in {line!r}")
        flags |= FLAG_CHARS[ch]
    return pid, regex.encode("latin-1"), flags, ext


def read_expressions(path):
    out = []
    with open(path, encoding="latin-1") as f:
        for n, line in enumerate(f, 1):
            try:
                e = parse_expression_line(line)
            except ParseError as pe:
                raise SystemExit(f"{path}:{n}: {pe}")
            if e:
                out.append(e)
    if not out:
        raise SystemExit(f"{path}: no expressions")
    return out


def read_corpus(path):
    """-> (corpus uint8, off uint64[nblocks+1], n_streams); blocks in chunk-id order"""
    if not os.path.exists(path):
        raise SystemExit(f"Corpus data error: Unable to open database '{path}'")
    db = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
    try:
        rows = db.execute("SELECT id, stream_id, data FROM chunk ORDER BY id;").fetchall()
    except sqlite3.Error as e:
        raise SystemExit(f"Corpus data error: Query failed: {e}")
    finally:
        db.close()
    if not rows:
        raise SystemExit("Corpus data error: Database contains no blocks.")
    lens = np.fromiter((len(r[2]) for r in rows), dtype=np.uint64, count=len(rows))
    off = np.zeros(len(rows) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    corpus = np.frombuffer(b"".join(bytes(r[2]) for r in rows), dtype=np.uint8)
    return corpus, off, len({r[1] for r in rows})


def write_corpus(path, corpus, off, stream_of_block=None):
    """CorpusBuilder.py's schema and post-processing (index, vacuum, analyze)"""
    if os.path.exists(path):
        raise SystemExit(f"Database '{path}' already exists")
    db = sqlite3.connect(path)
    db.executescript("CREATE TABLE chunk (id integer primary key, stream_id integer not null, data blob);")
    nb = off.size - 1
    raw = corpus.tobytes()
    db.executemany("insert into chunk (id, stream_id, data) values (?, ?, ?)",
                   ((i, int(stream_of_block[i]) if stream_of_block is not None else i,
                     sqlite3.Binary(raw[int(off[i]):int(off[i + 1])])) for i in range(nb) if off[i + 1] > off[i]))
    db.commit()
    db.execute("create index chunk_stream_id_idx on chunk(stream_id)")
    db.commit()
    db.execute("vacuum")
    db.execute("analyze")
    db.commit()
    db.close()


def calc_mbps(seconds, nbytes):  # main.cpp:721-724
    return nbytes / (seconds * 125000.0)


def main(argv=None):
    ap = argparse.ArgumentParser(description="hsbench-compatible driver for the MI355X literal engine", add_help=True)
    ap.add_argument("-e", dest="expr")
    ap.add_argument("-c", dest="corpus")
    ap.add_argument("-n", dest="repeats", type=int, default=20)
    ap.add_argument("-N", dest="block", action="store_true", help="block mode (the only mode; implied)")
    ap.add_argument("-T", dest="threads", default=None, help="accepted for compatibility; the scan runs on the GPU")
    ap.add_argument("--literal-on", action="store_true")
    ap.add_argument("--per-scan", action="store_true")
    ap.add_argument("--echo-matches", action="store_true")
    ap.add_argument("--resident", action="store_true", help="corpus resident in HBM, device pipeline only")
    ap.add_argument("--one-scan-per-block", action="store_true",
                    help="the reference's own call pattern in block mode: one hs_scan per block (engine_hyperscan.cpp:132-145) instead of one hs_scan_batch per repeat")
    ap.add_argument("--server", type=int, default=0, choices=[0, 1, 2],
                    help="hs_scratch_enable_small_batch_server on the scratch (1: requests through the PCIe BAR, 2: through mapped host memory)")
    ap.add_argument("--make-corpus", choices=["packets", "lines"])
    ap.add_argument("-o", dest="out")
    ap.add_argument("--mib", type=float, default=64.0)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args(argv)

    if a.make_corpus:
        from hyperscan_amd import corpus as cp

        if not a.out:
            raise SystemExit("--make-corpus needs -o FILE")
        total = int(a.mib * (1 << 20))
        lits = []
        if a.expr:
            class L:  # plant the literal parts of the patterns
                def __init__(self, s):
                    self.s = s
            lits = [L(re.split(rb"[\\\[\.\(\{\*\+\?\|\^\$]", e[1])[0]) for e in read_expressions(a.expr)]
            lits = [l for l in lits if l.s]
        if a.make_corpus == "packets":
            corpus, off = cp.packet_corpus(total, lits or cp.teddy_literals(64, seed=2), seed=a.seed)
        else:
            corpus, off = cp.line_corpus(total, seed=a.seed)
        write_corpus(a.out, corpus, off)
        print(f"wrote {a.out}: {corpus.size} bytes in {off.size - 1} blocks")
        return 0

    if not a.expr or not a.corpus:
        ap.error("-e FILE and -c FILE are required")
    from hyperscan_amd import hs

    exprs = read_expressions(a.expr)
    t0 = time.perf_counter()
    corpus, off, _n_streams = read_corpus(a.corpus)
    nblocks = off.size - 1

    ids = [e[0] for e in exprs]
    flags = [e[2] for e in exprs]
    pats = [e[1] for e in exprs]
    t0 = time.perf_counter()
    try:
        if a.literal_on:
            db = hs.Database.compile_lit(pats, flags, ids)
        elif any(e[3] for e in exprs):
            db = hs.Database.compile_ext(pats, flags, ids, [hs.ExprExt.make(**e[3]) if e[3] else None for e in exprs])
        else:
            db = hs.Database.compile(pats, flags, ids)
    except hs.HsError as e:
        print(f"Error: expressions failed to compile.\n  {e.message} (expression index {e.expression})")
        return 1
    compile_secs = time.perf_counter() - t0
    scratch = hs.HsScratch(db)
    lib = hs._lib()
    info = C.c_char_p()
    lib.hs_database_info(db._h, C.byref(info))
    blob = db.serialize()

    # printStats, engine_hyperscan.cpp:246-277
    print(f"Signatures:        {a.expr}")
    print(f"Hyperscan info:    {info.value.decode()}")
    print(f"Expression count:  {len(exprs)}")
    print(f"Bytecode size:     {db.size()} bytes")
    print(f"Database CRC:      0x{zlib.crc32(blob) & 0xffffffff:x}")
    ssz = C.c_size_t()
    lib.hs_scratch_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.hs_scratch_size(scratch._h, C.byref(ssz))
    print(f"Scratch size:      {ssz.value} bytes")
    print(f"Compile time:      {compile_secs:0.3f} seconds")
    print("Peak heap usage:   0 bytes")
    print()

    results = []  # (seconds, matches) per repeat
    if a.resident:
        import torch

        import hyperscan_amd as H
        from hyperscan_amd import hwlm as hw

        if not a.literal_on or any(len(p) > 8 or f & 8 for p, f in zip(pats, flags)) or len(set(ids)) != len(ids):
            raise SystemExit("--resident times the device literal pipeline alone: it needs --literal-on, literals of "
                             "<= 8 bytes, distinct ids and no SINGLEMATCH (anything else needs the host confirm)")
        table = H.hwlm_build([H.HwlmLiteral(p, nocase=bool(f & 1), id=i) for p, f, i in zip(pats, flags, ids)])
        gs = H.Scratch(0)
        dev = torch.device("cuda", 0)
        d_corpus = torch.from_numpy(corpus.copy()).to(dev)
        d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
        cap = max(1 << 16, corpus.size // 64)
        d_out = torch.zeros(cap * 4, dtype=torch.int32, device=dev)
        d_count = torch.zeros(1, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def one():
            hw.hwlm_scan_dev(table, gs, d_corpus.data_ptr(), corpus.size, d_off.data_ptr(), nblocks, d_out.data_ptr(), cap,
                             d_count.data_ptr(), 0, stream)
        one()
        torch.cuda.synchronize()
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            one()
            torch.cuda.synchronize()
            results.append((time.perf_counter() - t0, int(d_count.item())))
    elif a.one_scan_per_block:
        # the reference's loop (engine_hyperscan.cpp:132-145): hs_scan per block, a counting callback. (Driven from Python:
        # ~1.5 us of ctypes per call on top of the call itself.)
        if a.server:
            scratch.enable_server(a.server)
        cnt = [0]

        def on_match(_id, _from, _to, _flags, _ctx):
            cnt[0] += 1
            return 0
        cb = hs.MATCH_CB(on_match)
        base = corpus.ctypes.data
        starts = [base + int(o) for o in off[:-1]]
        lens = [int(n) for n in np.diff(off.astype(np.int64))]
        scan = lib.hs_scan
        scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, hs.MATCH_CB, C.c_void_p]
        for _ in range(a.repeats):
            cnt[0] = 0
            t0 = time.perf_counter()
            for p, n in zip(starts, lens):
                rv = scan(db._h, p, n, 0, scratch._h, cb, None)
                if rv != 0:
                    print(f"Fatal error: hs_scan returned error {rv}")
                    return 1
            results.append((time.perf_counter() - t0, cnt[0]))
        if a.server:
            calls, launches = scratch.server_stats()
            print(f"Small-batch server:        {calls} calls served, {launches} server launches")
    else:
        handler = C.cast(lib.hs_batch_count_handler, hs.BATCH_CB)
        echo = None
        if a.echo_matches:
            echo = hs.BATCH_CB(lambda b, i, f, t, _fl, ctx: (print(f"Match @{b}:{t} for {i}"), 0)[1])
        for _ in range(a.repeats):
            cnt = C.c_ulonglong(0)
            t0 = time.perf_counter()
            rv = lib.hs_scan_batch(db._h, corpus.ctypes.data, off.ctypes.data, nblocks, 0, scratch._h,
                                   echo if echo else handler, C.byref(cnt))
            dt = time.perf_counter() - t0
            if rv != 0:
                print(f"Fatal error: hs_scan returned error {rv}")
                return 1
            if echo:  # count separately: the echo handler has no counter
                cnt = C.c_ulonglong(0)
                lib.hs_scan_batch(db._h, corpus.ctypes.data, off.ctypes.data, nblocks, 0, scratch._h, handler, C.byref(cnt))
            results.append((dt, cnt.value))

    # displayResults, main.cpp:771-856
    total_secs = sum(r[0] for r in results)
    bytes_per_run = int(corpus.size)
    matches_per_run = results[0][1]
    if any(r[1] != matches_per_run for r in results):
        print("\nWARNING: PER-SCAN MATCH COUNTS ARE INCONSISTENT!\n")
    print(f"Time spent scanning:       {total_secs:0.3f} seconds")
    print(f"Corpus size:               {bytes_per_run} bytes ({nblocks} blocks)")
    print(f"Matches per iteration:     {matches_per_run} ({matches_per_run * 1024 / bytes_per_run:0.3f} matches/kilobyte)")
    print(f"Overall block rate:        {nblocks * a.repeats / total_secs:0.2f} blocks/sec")
    print(f"Mean throughput (overall): {calc_mbps(total_secs, bytes_per_run * a.repeats):0.2f} Mbit/sec")
    print(f"Max throughput (per core): {calc_mbps(min(r[0] for r in results), bytes_per_run):0.2f} Mbit/sec")
    print()
    if a.per_scan:
        for j, r in enumerate(results):
            print(f"T  0 Scan {j:2d}: {calc_mbps(r[0], bytes_per_run):0.2f} Mbit/sec")
        print()
    return 0


if __name__ == "__main__":
    sys.exit(main())
