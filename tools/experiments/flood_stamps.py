"""tools/experiments/flood_stamps.py [blocks] -- the bench's flood corpus, the confirm kernel's per-worker timeline of one dense scan
(a library built -DHSGPU_CONFIRM_STAMPS=1): HSGPU_LIB_VARIANT=_st python tools/experiments/flood_stamps.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from hyperscan_amd import corpus as cp  # noqa: E402
from hyperscan_amd.hwlm import HwlmLiteral  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
blk = 1 << 20
corpus = np.repeat((np.arange(nb) % 16 + ord("a")).astype(np.uint8), blk)
off = (np.arange(nb + 1, dtype=np.uint64) * np.uint64(blk))
lits = []
for c in b"abcd":
    lits += [HwlmLiteral(bytes([c]) * 4, False, len(lits)), HwlmLiteral(bytes([c]) * 8, False, len(lits) + 1), HwlmLiteral(bytes([c]) * 3 + b"x", False, len(lits) + 2)]
lits += [HwlmLiteral(l.s, l.nocase, len(lits) + i) for i, l in enumerate(cp.teddy_literals(100, seed=12))]
want = sum((blk - 3) + (blk - 7) for b in range(nb) if b % 16 < 4)
cap = want + (1 << 20)
job = bench.GpuJob(lits, corpus, off, torch.cuda.current_device(), cap=cap)
mode = int(os.environ.get("FLOOD_TUNE", "0"))
if mode:
    job.scratch.set_tuning(mode)
job.scratch.enable_timing(2)
q = lambda x: " ".join(f"{v:.1f}" for v in np.percentile(x, [0, 1, 10, 50, 90, 99, 100]))
for attempt in range(5):  # (with the timed steps below fewer than 16: the scratch leaves dense mode after that many quiet scans)
    job.launch()
    torch.cuda.synchronize()
    n = job.count()
    if n > cap:  # "again with more room" (a staging region, sized from cap for an even spread, overflowed)
        cap *= 2
        job.cap = cap
        job.d_out = None
        torch.cuda.empty_cache()
        job.d_out = torch.zeros(cap * 4, dtype=torch.int32, device=job.dev)
    f, c, t = job.scratch.timing(0)
    st = job.scratch.conf_stamps()
    act = st[st[:, 1] > 0]
    print(f"scan {attempt}: count {n} want {want}; filter {f:.3f} ms, confirm stage {c:.3f}, pipeline {t:.3f}; workers stamped {len(act)} of {len(st)}", flush=True)
    if len(act):
        life = (act[:, 1] - act[:, 0]) * 1e3
        print(f"   (min p1 p10 p50 p90 p99 max) start us {q(act[:, 0] * 1e3)} | end us {q(act[:, 1] * 1e3)} | life us {q(life)}")
        print(f"   fresh {q(act[:, 2])} | rest {q(act[:, 3])} | drains {q(act[:, 4])} | entries {q(act[:, 5])}; sum of lives {life.sum() / 1e3:.1f} ms over {len(act)} workers = {life.mean():.1f} us each")
import time  # noqa: E402

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    job.launch()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"steps: {dt * 1e3:.3f} ms each = {job.total / dt / 1e9:.1f} GB/s (count {job.count()})")
