#!/usr/bin/env python3
"""tools/confirm_prof.py [MiB] -- CPU only: config 5's host confirm (hs_confirm_batch) alone, over literal hits taken from the
compiled reference's hwlmExec (oracle/_ref) on the bench's rose1000 corpus: hits/s, the figure also.rose1000.host_confirm reports."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperscan_amd import corpus as cp, hs  # noqa: E402
from hyperscan_amd.hwlm import MATCH_DTYPE, HwlmLiteral  # noqa: E402
from tests import oracle_binding as ob, rose_model as RM  # noqa: E402


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(6)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
    lits = sorted({bytes(rng.choice(alpha, int(rng.integers(6, 13)))) for _ in range(1000)})
    pats = [l.decode() + RM.TAILS[i % 3] for i, l in enumerate(lits)]
    db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))

    class L:
        def __init__(self, s):
            self.s = s
    follow = [b"abc7", b"  key=", b"....END"]
    plant = [L(l + follow[i % 3]) if i % 2 == 0 else L(l) for i, l in enumerate(lits)]
    os.makedirs("/tmp/c5", exist_ok=True)
    cache = f"/tmp/c5/rose_{mib}.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        corpus, off, recs = z["corpus"], z["off"], z["recs"]
    else:
        corpus, off = cp.packet_corpus(mib << 20, plant, seed=6, match_every=4096)
        keyed = db.literals()
        hl = [HwlmLiteral(k[0], k[1], i) for i, k in enumerate(keyed)]
        hits = ob.Reference(hl, variant=ob.ref_variants()[-1]).collect_blocks(corpus, off)
        order = np.lexsort((hits["id"], hits["end"], hits["block"]))
        recs = np.zeros(len(hits), dtype=MATCH_DTYPE)
        recs["block"], recs["end"], recs["id"], recs["lit"] = hits["block"][order], hits["end"][order], hits["id"][order], hits["id"][order]
        np.savez(cache, corpus=corpus, off=off, recs=recs)
    lib = hs._lib()
    handler = C.cast(lib.hs_batch_count_handler, hs.BATCH_CB)
    lib.hs_confirm_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_ulonglong, hs.BATCH_CB, C.c_void_p]
    offs = np.ascontiguousarray(off, dtype=np.uint64)
    ts = []
    for _ in range(5):
        cnt = C.c_ulonglong(0)
        t0 = time.perf_counter()
        rv = lib.hs_confirm_batch(db._h, corpus.ctypes.data, offs.ctypes.data, offs.size - 1, recs.ctypes.data, len(recs), handler, C.byref(cnt))
        ts.append(time.perf_counter() - t0)
        assert rv == 0
    t = float(np.median(ts[1:]))
    tm = (C.c_double * 5)()
    lib.hsgpu_debug_confirm_timing(tm)
    print("last call: setup %.2f ms, parallel part %.2f ms (slowest worker %.2f, fastest %.2f), delivery %.2f ms" % tuple(x * 1e3 for x in tm))
    print(f"{mib} MiB, {len(recs)} hits, {cnt.value} events: {t * 1e3:.2f} ms = {len(recs) / t / 1e6:.2f} M hits/s on {os.cpu_count()} CPUs (runs: {[round(x * 1e3, 2) for x in ts]})")


if __name__ == "__main__":
    main()
