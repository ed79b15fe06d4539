import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hyperscan_amd import corpus as cp
from hyperscan_amd.hwlm import HwlmLiteral
nb = int(sys.argv[1]); blk = 1 << 20
corpus = np.repeat((np.arange(nb) % 16 + ord("a")).astype(np.uint8), blk)
off = (np.arange(nb + 1, dtype=np.uint64) * np.uint64(blk))
lits = []
for c in b"abcd":
    lits += [HwlmLiteral(bytes([c]) * 4, False, len(lits)), HwlmLiteral(bytes([c]) * 8, False, len(lits) + 1), HwlmLiteral(bytes([c]) * 3 + b"x", False, len(lits) + 2)]
lits += [HwlmLiteral(l.s, l.nocase, len(lits) + i) for i, l in enumerate(cp.teddy_literals(100, seed=12))]
want = sum(4 if False else 0 for _ in ())
want = sum((blk - 3) + (blk - 7) for b in range(nb) if b % 16 < 4)
cap = 2 * want + (1 << 20)
job = bench.GpuJob(lits, corpus, off, torch.cuda.current_device(), cap=cap)
job.scratch.enable_timing(2)
for attempt in range(4):
    job.launch(); torch.cuda.synchronize()
    n = job.count()
    st = job.scratch.conf_stamps()
    act = st[st[:, 0] >= 0]
    print(f"attempt {attempt}: count {n} want {want} cap {cap}; workers stamped {len(act)} of {len(st)}; entries sum {act[:,5].sum():.0f}; fresh {act[:,2].sum():.0f} rest {act[:,3].sum():.0f} drains {act[:,4].sum():.0f}; life max {((act[:,1]-act[:,0])*1e3).max() if len(act) else 0:.1f} us", flush=True)
