"""tools/seq_prof.py [mib] -- config 4's sequence stage on its own (the bench's 256 patterns over the line corpus, bitmaps
of the 12 classes first) for a kernel trace / PMC pass: rocprofv3 --kernel-trace --stats -- python tools/seq_prof.py 256"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from hyperscan_amd import accel
    from hyperscan_amd import corpus as cp

    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(5)
    abm, used = [], set()
    for k in range(256):
        a, b = rng.choice(len(bench.CLASS_POOL), 2, replace=False)
        abm.append((int(a), int(b), int(rng.integers(3, 9))))
        used |= {int(a), int(b)}
    classes = [accel.CharClass(bench.CLASS_POOL[i][1]) for i in sorted(used)]
    idx = {i: k for k, i in enumerate(sorted(used))}
    seqs = [(idx[a], idx[b], m, 1, k) for k, (a, b, m) in enumerate(abm)]
    corpus, off = cp.line_corpus(mib << 20, seed=5)
    total, nb = int(corpus.size), int(off.size - 1)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(corpus).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    bm, _f, _l = accel.class_scan(classes, d, total, d_off, nb, False, False)
    bitmaps = [bm[i] for i in range(len(classes))]
    buf = accel.class_seq_buffers(len(seqs), total, 0, dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record()
        accel.class_seq_scan(seqs, bitmaps, total, d_off, nb, (0, 0), 0, buffers=buf)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    print(f"seq_prof: {total} bytes, {nb} lines, 256 patterns over {len(classes)} classes: {ms:.3f} ms = {ms * 1024 / mib:.2f} ms/GiB, "
          f"{int(buf[1].sum().item())} match ends")


if __name__ == "__main__":
    main()
