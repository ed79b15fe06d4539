#!/usr/bin/env python3
"""tools/bench_extra.py -- the other configs of BASELINE.json on one GPU (not part of the
driver's bench contract; results are recorded in profiles/ and DESIGN.md):

  class8    config 4: character-class scanning (shufti/truffle/vermicelli semantics): the 8
            distinct classes behind the 256 `[a-z]{3,}\\d+`-style patterns, evaluated in ONE
            pass over a line-based corpus (one block per line): membership bitmaps +
            per-line first/last hit.
  rose1000  config 5: 1000 patterns "LIT_k<tail>" through the public hs_* API: GPU literal
            hits feeding the host-side Rose-lite confirm (hs_scan_batch).
Prints one JSON line per workload."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def class8(gib, iters):
    import torch

    from hyperscan_amd import accel
    from hyperscan_amd import corpus as cp
    total = int(gib * (1 << 30))
    corpus, off = cp.line_corpus(total, seed=5)
    classes = [accel.CharClass(range(ord("a"), ord("z") + 1)), accel.CharClass(range(ord("A"), ord("Z") + 1)),
               accel.CharClass(range(ord("0"), ord("9") + 1)), accel.CharClass(b"0123456789abcdef"),
               accel.CharClass(b" \t\r\n\x0b\x0c"), accel.CharClass(range(128, 256)), accel.CharClass(b"aeiou"),
               accel.CharClass(b"\n")]
    dev = torch.device("cuda", 0)
    d_corpus = torch.from_numpy(corpus).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    nb = off.size - 1
    for _ in range(2):
        bm, first, last = accel.class_scan(classes, d_corpus, total, d_off, nb, True, True)
    # sanity on a slice (the parity tests proper live in tests/test_gpu_class_scan.py): bit i <=> corpus[i] in class
    n = 1 << 20
    for ci, cls in enumerate(classes):
        want = np.packbits(np.isin(corpus[:n], np.array(cls.members(), dtype=np.uint8)), bitorder="little")
        assert np.array_equal(bm[ci][: n // 8].cpu().numpy(), want)
    work = torch.zeros(accel.WORK_BYTES, dtype=torch.uint8, device=dev)
    bufs = (bm, first, last, work)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters):
        ev[i][0].record()
        accel.class_scan(classes, d_corpus, total, d_off, nb, True, True, buffers=bufs)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) / 1e3 for a, b in ev]
    t = float(np.median(ts))
    alg = total * (1 + len(classes) / 8) + nb * len(classes) * 8
    print(json.dumps({"workload": f"class8: 8 classes, {gib:g} GiB line corpus, {nb} blocks", "GBps_corpus": round(total / t / 1e9, 1),
                      "ms": round(t * 1e3, 3), "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1),
                                                              "peak": 8000.0, "frac": round(alg / t / 1e9 / 8000, 4),
                                                              "note": "HIP events around table upload + class_bitmap_kernel + class_first_last_kernel"}}))


def rose1000(gib, iters):
    from hyperscan_amd import corpus as cp
    from hyperscan_amd import hs

    rng = np.random.default_rng(6)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
    tails = [r"[a-z]+\d", r"\s+\w{2,8}=", r".{0,16}END"]
    lits = sorted({bytes(rng.choice(alpha, int(rng.integers(6, 13)))) for _ in range(1000)})
    pats = [l.decode() + tails[i % 3] for i, l in enumerate(lits)]
    db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))
    scratch = hs.HsScratch(db)
    total = int(gib * (1 << 30))

    class L:  # corpus generator wants objects with .s
        def __init__(self, s):
            self.s = s
    follow = [b"abc7", b"  key=", b"....END"]
    plant = [L(l + follow[i % 3]) if i % 2 == 0 else L(l) for i, l in enumerate(lits)]
    corpus, off = cp.packet_corpus(total, plant, seed=6, match_every=4096)
    import ctypes as C

    lib = hs._lib()
    handler = C.cast(lib.hs_batch_count_handler, hs.BATCH_CB)  # hsbench's counting callback, native
    buf = np.ascontiguousarray(corpus)
    offs = np.ascontiguousarray(off, dtype=np.uint64)
    ts, n_ev = [], 0
    for _ in range(iters):
        cnt = C.c_ulonglong(0)
        t0 = time.perf_counter()
        rv = lib.hs_scan_batch(db._h, buf.ctypes.data, offs.ctypes.data, offs.size - 1, 0, scratch._h, handler, C.byref(cnt))
        ts.append(time.perf_counter() - t0)
        assert rv == 0
        assert n_ev in (0, cnt.value), "match count changed between repeats"
        n_ev = cnt.value
    t = float(np.median(ts))
    print(json.dumps({"workload": f"rose1000: 1000 literal-prefix+tail patterns, {gib:g} GiB packets, hs_scan_batch "
                                  "(H2D of the corpus + GPU literal scan + D2H of records + host confirm + counting callback)",
                      "GBps_end_to_end": round(total / t / 1e9, 2), "ms": round(t * 1e3, 1), "matches": n_ev,
                      "matches_per_s": round(n_ev / t, 1)}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["class8", "rose1000"])
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    {"class8": class8, "rose1000": rose1000}[a.workload](a.gib, a.iters)
