"""Times the scan pipeline of the bench's 10 000-literal workload without the parity gate (tuning experiments whose kernels
are deliberately incomplete): prints the average step and the count it produced."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

lits, corpus, off = bench.build_workload("fdr10k", 1 << 30, 0)
jb = bench.GpuJob(lits, corpus, off, 0)
for _ in range(5):
    jb.launch()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); jb.launch(); b.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
span = [jb.scratch.kernel_span(k) for k in range(8)]
print("step ms avg %.4f best %.4f | first-kernel span ms %s | count %d" % (sum(ms) / len(ms), min(ms), ["%.4f" % s[0] if isinstance(s, tuple) else "%.4f" % s for s in span][:4], jb.count()))
