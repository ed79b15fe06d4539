"""tools/flood_prof.py [blocks] [dense_every] -- the bench's flood case on its own (for a kernel trace):
rocprofv3 --kernel-trace --stats -- python tools/flood_prof.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from hyperscan_amd import corpus as cp
    from hyperscan_amd.hwlm import HwlmLiteral

    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    blk = 1 << 20
    corpus = np.repeat((np.arange(nb) % 16 + ord("a")).astype(np.uint8), blk)
    off = (np.arange(nb + 1, dtype=np.uint64) * np.uint64(blk))
    lits = []
    for c in b"abcd":
        lits += [HwlmLiteral(bytes([c]) * 4, False, len(lits)), HwlmLiteral(bytes([c]) * 8, False, len(lits) + 1),
                 HwlmLiteral(bytes([c]) * 3 + b"x", False, len(lits) + 2)]
    lits += [HwlmLiteral(l.s, l.nocase, len(lits) + i) for i, l in enumerate(cp.teddy_literals(100, seed=12))]
    want_total = (nb // 16) * 4 * ((blk - 3) + (blk - 7))
    cap = want_total + (1 << 20)
    job = bench.GpuJob(lits, corpus, off, torch.cuda.current_device(), cap=cap)
    for attempt in range(5):
        job.launch()
        torch.cuda.synchronize()
        n = job.count()
        print(f"attempt {attempt}: count {n} cap {cap} stats {job.scratch.stats()}", flush=True)
        if n <= cap:
            break
        cap *= 2
        job.cap = cap
        job.d_out = None
        torch.cuda.empty_cache()
        job.d_out = torch.zeros(cap * 4, dtype=torch.int32, device=job.dev)
    assert n == want_total, (n, want_total)
    steps = 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    f_ms, c_ms, p_ms = job.scratch.timing(0)
    print(f"flood_prof: {job.total} bytes, {n} matches: {dt * 1e3:.3f} ms/step = {job.total / dt / 1e9:.2f} GB/s; filter {f_ms:.3f} ms, "
          f"filter end -> next stage {c_ms:.3f}, filter start -> sort start {p_ms:.3f}; table {job.table.info()}")


main()
