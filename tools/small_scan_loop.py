#!/usr/bin/env python3
"""tools/small_scan_loop.py [bytes] [iters] [mode] -- resident scans of a small batch in a loop (for a kernel trace):
mode 0 the runtime's choice, 3 three kernels, 4 one launch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lits, corpus, off = bench.build_workload("fdr10k", max(size, 1 << 16), 0)
# cut the corpus at a block boundary at or below `size`
nb = int(np.searchsorted(off, size, side="right")) - 1
nb = max(nb, 1)
off = off[: nb + 1].copy(); corpus = corpus[: int(off[-1])].copy()
job = bench.GpuJob(lits, corpus, off, 0)
job.scratch.set_tuning(mode)
for _ in range(20):
    job.launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    job.launch()
torch.cuda.synchronize()
print(f"{int(off[-1])} bytes, {nb} blocks, mode {mode}: {(time.perf_counter() - t0) / iters * 1e6:.2f} us per scan, {job.count()} matches")
