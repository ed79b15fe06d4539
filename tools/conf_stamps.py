#!/usr/bin/env python3
"""tools/conf_stamps.py -- on the GPU box, with a library built -DHSGPU_CONFIRM_STAMPS=1 (HSGPU_LIB_VARIANT): the confirm kernel's
per-worker timeline of one scan of a bench workload: when every worker wavefront started and ended, its steps and drains.
Usage: HSGPU_LIB_VARIANT=_st python tools/conf_stamps.py [fdr10k] [--gib 1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="fdr10k")
    ap.add_argument("--gib", type=float, default=1.0)
    a = ap.parse_args()
    import torch

    import bench

    lits, corpus, off = bench.build_workload(a.workload, int(a.gib * (1 << 30)), 0)
    job = bench.GpuJob(lits, corpus, off, 0)
    job.scratch.enable_timing(2)
    for _ in range(4):
        job.launch()
    torch.cuda.synchronize()
    f, c, t = job.scratch.timing(0)
    st = job.scratch.conf_stamps()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "conf_stamps%s.npy" % os.environ.get("HSGPU_LIB_VARIANT", ""))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.save(out, st)
    st = st[st[:, 0] >= 0]
    q = lambda x: " ".join(f"{v:.1f}" for v in np.percentile(x, [0, 1, 10, 50, 90, 99, 100]))
    us = 1e3
    print(f"{a.workload} {a.gib:g} GiB: filter {f:.4f} ms, confirm stage {c:.4f} ms, pipeline {t:.4f}; {len(st)} workers")
    print(f"  (min p1 p10 p50 p90 p99 max)  start us: {q(st[:, 0] * us)} | end us: {q(st[:, 1] * us)} | life us: {q((st[:, 1] - st[:, 0]) * us)}")
    print(f"  fresh steps: {q(st[:, 2])} | rest steps: {q(st[:, 3])} | sorted drains: {q(st[:, 4])} | entries: {q(st[:, 5])}")
    life = (st[:, 1] - st[:, 0]) * us
    steps = st[:, 2] + st[:, 3]
    ok = steps > 0
    print(f"  us per step (life / (fresh + rest)): {q(life[ok] / steps[ok])}; corr(life, entries) = {np.corrcoef(life, st[:, 5])[0, 1]:.3f}")
    span = st[:, 1].max() - st[:, 0].min()
    print(f"  kernel span {span * us:.1f} us; mean life {life.mean():.1f} us = {life.mean() / (span * us):.2f} of the span; workers alive at 50/75/90 % of the span: "
          + " ".join(str(int(((st[:, 0] <= st[:, 0].min() + p * span) & (st[:, 1] > st[:, 0].min() + p * span)).sum())) for p in (0.5, 0.75, 0.9)))
    # by workgroup position: are late finishers clustered (XCD / CU)?
    wg = np.arange(len(st)) // 4
    by_xcd = [life[(wg % 8) == x].mean() for x in range(8)]
    print("  mean life by workgroup index mod 8 (XCD round robin): " + " ".join(f"{v:.1f}" for v in by_xcd))


if __name__ == "__main__":
    main()
