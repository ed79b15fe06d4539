"""One rank of the N > 1 step, for tests/test_gpu_exchange.py (run as a script, one process per GPU):
scan this rank's shard on the device with hsgpu_hwlm_scan_dev, exchange the records with
hyperscan_amd.dist.RecordExchange over the given backend (nccl = RCCL), and check on rank 0 that the
rows of all ranks, global block indices included, equal the oracle's scan of the whole corpus.

    python tests/exchange_worker.py <backend> <rank> <world> <port> [exact]

With `exact` the step goes through hyperscan_amd.dist.ExactExchange (counts agreed once, unpadded broadcasts).
"""
import os
import sys

import numpy as np


def main():
    backend, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist

    import hyperscan_amd as H
    from hyperscan_amd import corpus as cp
    from hyperscan_amd import dist as hd
    from hyperscan_amd import hwlm as hw
    from tests import oracle_binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    lits = cp.teddy_literals(32, seed=5)
    corpus, off = cp.packet_corpus(2 << 20, lits, seed=9, match_every=1024)
    my_corpus, my_off, base = hd.local_shard(corpus, off, rank, world)
    base += 7  # a non-zero first global block even at world size 1: the add must happen
    t = H.hwlm_build(lits)
    s = H.Scratch(dev.index)
    d_corpus = torch.from_numpy(np.ascontiguousarray(my_corpus)).to(dev)
    d_off = torch.from_numpy(my_off.view(np.int64)).to(dev)
    rows = 1 << 16
    d_out = torch.zeros((rows, 4), dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    exact = len(sys.argv) > 5 and sys.argv[5] == "exact"
    stream = torch.cuda.current_stream().cuda_stream
    if exact:  # the counts and bases of all ranks, once, from a warm-up scan
        hw.hwlm_scan_dev(t, s, d_corpus.data_ptr(), int(my_off[-1]), d_off.data_ptr(), my_off.size - 1,
                         d_out.data_ptr(), rows, d_count.data_ptr(), stream=stream)
        mine = torch.tensor([int(d_count.item()), base], dtype=torch.int64, device=dev)
        allc = torch.empty(world * 2, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, mine)
        allc = allc.view(world, 2).cpu()
        ex = hd.ExactExchange(dist, world, rank, dev, allc[:, 0].tolist(), allc[:, 1].tolist())
    else:
        ex = hd.RecordExchange(dist, world, rank, dev, rows, base)
    for _ in range(3):  # the same buffers step after step, nothing allocated, no host sync inside
        hw.hwlm_scan_dev(t, s, d_corpus.data_ptr(), int(my_off[-1]), d_off.data_ptr(), my_off.size - 1,
                         d_out.data_ptr(), rows, d_count.data_ptr(), stream=stream)
        ex.step(d_out, d_count)
    got, counts = ex.compact()
    got = got.cpu().numpy()
    if rank == 0:
        want = ob.Oracle(lits).collect_blocks(corpus, off)
        g = list(zip((got[:, 0].astype(np.int64) & 0xFFFFFFFF).tolist(), got[:, 1].tolist(), got[:, 2].tolist()))
        g = [(b - 7, e, i) for b, e, i in g]  # every rank added 7 to its own base
        w = sorted(zip(want["block"].tolist(), want["end"].tolist(), want["id"].tolist()))
        assert len(w) > 100 and sum(counts) == len(g), (len(w), counts)
        assert g == w, "rows of all ranks in rank order are not the whole-corpus scan in delivery order"
        print("EXCHANGE_OK", backend, world, len(g))
    dist.barrier()
    dist.destroy_process_group()
    s.close()


if __name__ == "__main__":
    main()
