"""tools/class_prof.py [gib] -- the class256 work item of bench.py on its own (8 classes, first + last per
line) for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/class_prof.py 1"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from hyperscan_amd import accel
    from hyperscan_amd import corpus as cp

    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    n_cls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    fl = not (len(sys.argv) > 3 and sys.argv[3] == "nofl")
    unit, uoff = cp.line_corpus(int(gib * (1 << 30)), seed=5)
    total, nb = int(unit.size), int(uoff.size - 1)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(unit).to(dev)
    d_off = torch.from_numpy(uoff.view(np.int64)).to(dev)
    pool = ["abcdefghijklmnopqrstuvwxyz", "0123456789", " \t", "ABCDEFGHIJKLMNOPQRSTUVWXYZ", "aeiou", "/.:-_", "{}[]()<>", "\"'=&?%"]
    classes = [accel.CharClass(pool[i % len(pool)]) for i in range(n_cls)]
    bm, first, last = accel.class_scan(classes, d, total, d_off, nb, fl, fl)
    bufs = (bm, first, last, torch.zeros(accel.WORK_BYTES, dtype=torch.uint8, device=dev))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        accel.class_scan(classes, d, total, d_off, nb, fl, fl, buffers=bufs)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    alg = total * (1 + n_cls / 8) + (nb * (n_cls * 8 + 8) if fl else 0)
    print(f"class_prof{'' if fl else ' (bitmaps only)'}: {total} bytes, {nb} lines, {n_cls} classes: {ms:.3f} ms/pass, {total / ms / 1e6:.1f} GB/s of corpus, "
          f"algorithmic {alg / ms / 1e6:.1f} GB/s ({alg / ms / 1e6 / 8000:.3f} of 8 TB/s)")


if __name__ == "__main__":
    main()
