#!/usr/bin/env python3
"""tools/ab_tail.py -- on the GPU box: one workload resident, the step timed under several pipelines / library variants
back to back in ONE process per variant (box-to-box spread is +-5 %, so only same-box numbers compare).
  python tools/ab_tail.py [fdr10k|teddy64] [--gib 1] [--iters 30] [--modes folded,unfolded] [--wg-stamps]
Prints per mode: wall ms per step (serial launches), filter ms (HIP events / device clock), confirm-stage ms,
device-clock pipeline ms, matches; with --wg-stamps the distribution of the filter workgroups' prologue and end times.
Library variants: HSGPU_LIB_VARIANT=_x python tools/ab_tail.py ... (csrc/Makefile VARIANT=_x)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="fdr10k")
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--modes", default="folded,unfolded")
    ap.add_argument("--wg-stamps", action="store_true")
    a = ap.parse_args()
    import torch

    import bench

    lits, corpus, off = bench.build_workload(a.workload, int(a.gib * (1 << 30)), 0)
    job = bench.GpuJob(lits, corpus, off, 0)
    tag = os.environ.get("HSGPU_LIB_VARIANT", "") or "default"
    ref = None
    for mode in a.modes.split(","):
        job.scratch.set_tuning({"folded": 0, "fused": 1, "unfolded": 2}[mode])
        job.scratch.enable_timing(2 if a.wg_stamps else 1)
        for _ in range(3):
            job.launch()
        torch.cuda.synchronize()
        n = job.count()
        recs = job.records()
        if ref is None:
            ref = recs
        same = recs.shape == ref.shape and bool(np.array_equal(recs, ref))
        job.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            job.launch()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters * 1e3
        fm, cm, tm, sm = [], [], [], []
        for back in range(min(a.iters, 30)):
            f, c, t = job.scratch.timing(back)
            fm.append(f); cm.append(c); tm.append(t); sm.append(job.scratch.kernel_span(back))
        print(f"{tag} {a.workload} {a.gib:g}GiB {mode}: step {wall:.4f} ms ({job.total / wall / 1e6:.1f} GB/s); filter ev {np.mean(fm):.4f} "
              f"clk {np.mean(sm):.4f} ms; filter end -> stage end {np.mean(cm):.4f}; pipeline clk {np.mean(tm):.4f}; matches {n}; "
              f"records identical to first mode: {same}; candidates {job.scratch.stats()}", flush=True)
        if a.wg_stamps:
            st = job.scratch.wg_stamps()
            if len(st):
                pro, own, end = st[:, 1] - st[:, 0], st[:, 2], st[:, 3]
                q = lambda x: " ".join(f"{v * 1e3:.1f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))
                print(f"   wg stamps ({len(st)} workgroups, us; min p10 p50 p90 max): start {q(st[:, 0])} | prologue {q(pro)} | "
                      f"wave0 share done {q(own)} | end {q(end)}", flush=True)


if __name__ == "__main__":
    main()
