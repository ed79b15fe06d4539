import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
name = sys.argv[1]
lits, corpus, off = bench.build_workload(name, 1 << 30, 0)
job = bench.GpuJob(lits, corpus, off, 0)
for _ in range(3): job.launch()
torch.cuda.synchronize()
def timed(tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): job.launch()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name} {tag}: host launch {1e3*(t1-t0)/20:.3f} ms/scan, total {1e3*(t2-t0)/20:.3f} ms/scan", flush=True)
timed("fresh")
recs = job.records(); timed("after records()")
cpu, _ = bench.cpu_baseline(lits, corpus, off, want_seconds=3.0); timed("after cpu_baseline")
time.sleep(3); timed("after 3 s idle")
