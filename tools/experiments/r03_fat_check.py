"""tools/experiments/r03_fat_check.py -- the prepared fat-bucket tables (r03_fat_buckets.patch, built as
hyperscan_amd/lib/libhsgpu_fat.so) against the plain ones on the fdr10k workload: identical records in identical
order, and the per-stage times of both. Run with HSGPU_LIB_VARIANT=_fat."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch

    import bench

    lits, corpus, off = bench.build_workload("fdr10k", 1 << 30, 0)
    recs = {}
    for fl in (0, 8192, 0, 8192):
        os.environ["HSGPU_BUILD_FLAGS"] = str(fl)
        job = bench.GpuJob(lits, corpus, off, 0)
        for _ in range(3):
            job.launch()
        torch.cuda.synchronize()
        n = job.count()
        for _ in range(20):
            job.launch()
        torch.cuda.synchronize()
        f, c, t = zip(*[job.scratch.timing(b) for b in range(20)])
        print(f"flags={fl} table flags {job.table.info()['flags']}: matches {n}; filter {np.mean(f):.4f} ms confirm {np.mean(c):.4f} ms "
              f"kernels {np.mean(t):.4f} ms", flush=True)
        if fl not in recs:
            recs[fl] = job.records()
        del job
    same = recs[0].shape == recs[8192].shape and bool(np.array_equal(recs[0], recs[8192]))
    print("records identical and in the same order:", same, recs[0].shape)


if __name__ == "__main__":
    main()
