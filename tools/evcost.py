#!/usr/bin/env python3
"""tools/evcost.py -- what do the library's per-scan HIP events cost? (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for name in sys.argv[1:] or ["teddy64"]:
    lits, corpus, off = bench.build_workload(name, 1 << 30, 0)
    job = bench.GpuJob(lits, corpus, off, 0)
    for timing in (True, False, True, False):
        job.scratch.enable_timing(timing)
        for _ in range(3): job.launch()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): job.launch()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"{name} timing={timing}: {dt*1e3:.4f} ms/scan", flush=True)
