#!/usr/bin/env python3
"""tools/kbench.py -- kernel-only timing loop for profiling / tuning (GPU box).
Usage: python tools/kbench.py [teddy64|fdr10k] [--gib 1] [--iters 20]
Prints kernel ms (avg/best), GB/s and the match count. No CPU baseline."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="teddy64")
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import torch

    import bench

    lits, corpus, off = bench.build_workload(a.workload, int(a.gib * (1 << 30)), 0)
    job = bench.GpuJob(lits, corpus, off, 0)
    for _ in range(3):
        job.launch()
    torch.cuda.synchronize()
    n = job.count()
    fm, cm, tm, sm = [], [], [], []
    for i in range(a.iters):
        job.launch()
    torch.cuda.synchronize()
    for back in range(min(a.iters, 32)):
        f, c, t = job.scratch.timing(back)
        fm.append(f); cm.append(c); tm.append(t); sm.append(job.scratch.kernel_span(back))
    ms = np.array(tm)
    gb = (job.total + 16 * n) / 1e9
    print(f"{a.workload}: kernel avg {ms.mean():.4f} ms best {ms.min():.4f} ms -> {gb / ms.mean() * 1e3:.1f} GB/s avg, "
          f"{gb / ms.min() * 1e3:.1f} best; filter {np.mean(fm):.4f} ms (device clock {np.mean(sm):.4f}) confirm {np.mean(cm):.4f} ms; matches {n}; "
          f"candidates {job.scratch.stats()}; table {job.table.info()}")


if __name__ == "__main__":
    main()
