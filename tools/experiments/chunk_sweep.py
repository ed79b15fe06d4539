#!/usr/bin/env python3
"""tools/chunk_sweep.py -- on the GPU box: hsgpu_hwlm_exec_batch_cb (copy / scan / deliver pipeline) from PINNED host memory at
several chunk sizes, against the bus itself (one pinned H2D copy of the same bytes): where config 5's end-to-end rate goes.
Prints one line per chunk size: ms, GB/s, fraction of the bus."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import hyperscan_amd as H
    from hyperscan_amd import corpus as cp
    from hyperscan_amd import hwlm as hw

    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    lits = cp.teddy_literals(64, seed=2)
    corpus, off = cp.packet_corpus(int(gib * (1 << 30)), lits, seed=6, match_every=4096)
    pinned = torch.from_numpy(corpus).pin_memory()
    buf = pinned.numpy()
    dev = torch.device("cuda", 0)
    d = torch.empty(corpus.size, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bus = []
    for _ in range(3):
        e0.record()
        d.copy_(pinned, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        bus.append(e0.elapsed_time(e1))
    bus_ms = float(np.median(bus))
    print(f"bus: one pinned H2D copy of {corpus.size} bytes: {bus_ms:.2f} ms = {corpus.size / bus_ms / 1e6:.1f} GB/s")
    del d
    table, scratch = H.hwlm_build(lits), H.Scratch(0)
    n_chunks = [0]

    def on_chunk(recs):
        n_chunks[0] += 1
        return 0
    for mib in (0, 16, 32, 64, 128, 256, 512):
        ts = []
        for _ in range(4):
            n_chunks[0] = 0
            t0 = time.perf_counter()
            hw.hwlm_exec_batch_pipelined(table, scratch, buf, off, chunk_bytes=mib << 20, on_chunk=on_chunk)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[1:])) * 1e3
        print(f"chunk {'ramp 8/16/32/64' if mib == 0 else str(mib) + ' MiB':>16}: {n_chunks[0]:3d} chunks with records, {t:7.2f} ms = "
              f"{corpus.size / t / 1e6:5.1f} GB/s = {bus_ms / t:.3f} of the bus")


if __name__ == "__main__":
    main()
