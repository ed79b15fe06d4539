#!/usr/bin/env python3
"""bench.py -- hsbench-style block-mode throughput of the GPU literal scan engine.

Metric (BASELINE.json): GB/s scanned in block mode (+ matches/s), whole job over
N GPUs, inputs resident in HBM when the timed region starts.

A "step" is one pass of the hot path over one batch: every rank scans its own
shard (all blocks of the corpus, one kernel launch) and, for N > 1, the match
records are all-gathered over RCCL. hsbench protocol (tools/hsbench/main.cpp:
502-528, 720-724, 823-839): corpus pre-loaded, barrier, K repeats, bytes*K/secs.

Workloads:
  teddy64   config 2: 64 literals len 4-8, 1 GiB synthetic packet corpus per GPU
  fdr10k    config 3: 10 000 snort-like literals, 1 GiB packet shard per GPU
The default is teddy64 (configs[1]); the fdr10k line is measured in the same run
and attached as "also".

One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
REC_BYTES = 16


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(name, total_bytes, seed_shift):
    from hyperscan_amd import corpus as cp

    if name == "teddy64":
        lits = cp.teddy_literals(64, seed=2)
        corpus, off = cp.packet_corpus(total_bytes, lits, seed=3 + seed_shift)
    elif name == "fdr10k":
        lits, _full = cp.snort_like_literals(10000, seed=4)
        corpus, off = cp.packet_corpus(total_bytes, lits, seed=10 + seed_shift)
    elif name.startswith("lits") and name[4:].isdigit():  # tuning: N literals of len 4-8 (stride-2 tables)
        lits = cp.teddy_literals(int(name[4:]), seed=2)
        corpus, off = cp.packet_corpus(total_bytes, lits[:256], seed=3 + seed_shift)
    else:
        raise SystemExit(f"unknown workload {name}")
    return lits, corpus, off


def pmc_traffic(table_flags):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
    command (profiles/r01_bench_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE
    in separate runs). Units are KB; on gfx950 FETCH_SIZE counts a wide coalesced read stream at
    half its bytes (MI355X_MICROARCH.md, HBM section), hence the factor 2 on the read side."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_summary.json")
    if not os.path.exists(path):
        return None
    b = lambda bit: "true" if table_flags & bit else "false"
    # HSGPU_F_BFOLD (128): the two-phase filter kernel of such a table is the 4-byte-key-only variant
    cls = ("true, true, true" if table_flags & 4 else
           "true, true, false" if table_flags & 2 and not table_flags & 128 else "true, false, false")
    name = f"hwlm_filter_kernel<{cls}, {b(8)}, {b(16)}, {b(32)}, {b(64)}, false>"
    try:
        e = json.load(open(path)).get(name)
        return int(e["FETCH_SIZE"]["avg_KB"] * 1024 * 2 + e["WRITE_SIZE"]["avg_KB"] * 1024) if e else None
    except Exception:
        return None


class GpuJob:
    """One rank's resident state: table, corpus/offsets/records in HBM."""

    def __init__(self, lits, corpus, off, device, sibling=None):
        import torch

        import hyperscan_amd as H

        self.torch = torch
        self.H = H
        self.dev = torch.device("cuda", device)
        self.scratch = H.Scratch(device)
        if sibling is not None:  # a second scan context over the SAME table and resident corpus
            self.table, self.total, self.nblocks = sibling.table, sibling.total, sibling.nblocks
            self.d_corpus, self.d_off = sibling.d_corpus, sibling.d_off
        else:
            self.table = H.hwlm_build(lits, int(os.environ.get("HSGPU_BUILD_FLAGS", "0")))  # tuning knob
            self.total = int(corpus.size)
            self.nblocks = int(off.size - 1)
            self.d_corpus = torch.from_numpy(corpus).to(self.dev)
            self.d_off = torch.from_numpy(off.view(np.int64)).to(self.dev)
        self.cap = max(1 << 16, self.total // 512)
        self.scratch.enable_timing(True)
        self.d_out = torch.zeros(self.cap * 4, dtype=torch.int32, device=self.dev)
        self.d_count = torch.zeros(1, dtype=torch.int64, device=self.dev)
        assert self.d_corpus.data_ptr() % 16 == 0

    def launch(self):
        """One batch scan (the library's kernel pipeline) on torch's current stream."""
        from hyperscan_amd import hwlm as hw

        stream = self.torch.cuda.current_stream().cuda_stream
        hw.hwlm_scan_dev(self.table, self.scratch, self.d_corpus.data_ptr(), self.total, self.d_off.data_ptr(),
                         self.nblocks, self.d_out.data_ptr(), self.cap, self.d_count.data_ptr(), 0, stream)

    def count(self):
        return int(self.d_count.item())

    def records(self):
        n = min(self.count(), self.cap)
        return self.d_out[: n * 4].view(n, 4).cpu().numpy().astype(np.uint32)


def cpu_baseline(lits, corpus, off, want_seconds=6.0, sample_bytes=64 << 20):
    """The reference's own hwlmExec (oracle/_ref, compiled from /root/reference) -- or
    the C restatement when that library is absent -- timed on this box's host cores
    over a bounded sample of the same workload, hsbench style: T threads, each
    scanning its own slice of the sample, repeated until ~want_seconds elapsed."""
    from tests import oracle_binding as ob

    k = int(np.searchsorted(off, min(sample_bytes, int(off[-1])), side="right")) - 1
    k = max(k, 1)
    s_off = off[: k + 1].copy()
    s_bytes = int(s_off[-1])
    sample = corpus[:s_bytes]
    kind = "reference" if ob.ref_available() else "port"
    threads = max(1, min(os.cpu_count() or 1, 64))
    eng = [ob.Reference(lits) if kind == "reference" else ob.Oracle(lits) for _ in range(threads)]
    info = eng[0].info() if kind == "reference" else "oracle/hwlm_oracle.c"
    # slice the sample's blocks evenly (by block count) over the threads
    bounds = np.linspace(0, k, threads + 1).astype(np.int64)
    counts = [0] * threads
    passes = [0] * threads

    def work(i, deadline):
        lo, hi = int(bounds[i]), int(bounds[i + 1])
        if hi <= lo:
            return
        o = s_off[lo:hi + 1]
        while True:
            counts[i] = eng[i].count_blocks(sample, o)
            passes[i] += 1
            if time.perf_counter() >= deadline:
                break

    # single-thread number first (T=1 over the whole sample), then all cores
    t0 = time.perf_counter()
    n1 = eng[0].count_blocks(sample, s_off)
    t1 = time.perf_counter() - t0
    reps1 = 1
    while t1 < want_seconds / 3:
        eng[0].count_blocks(sample, s_off)
        reps1 += 1
        t1 = time.perf_counter() - t0
    single = s_bytes * reps1 / t1 / 1e9

    t0 = time.perf_counter()
    deadline = t0 + want_seconds * 2 / 3
    ths = [threading.Thread(target=work, args=(i, deadline)) for i in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    scanned = sum(int(s_off[int(bounds[i + 1])] - s_off[int(bounds[i])]) * passes[i] for i in range(threads))
    multi = scanned / dt / 1e9
    assert sum(counts) == n1, "CPU baseline: per-thread counts do not add up"
    return {
        "value": round(multi, 4), "unit": "GB/s", "cores": threads, "kind": kind,
        "sample": f"first {s_bytes} bytes / {k} blocks of the same corpus, {threads} threads x own slice, "
                  f"~{want_seconds * 2 / 3:.0f}s; engine: {info}",
        "single_thread_GBps": round(single, 4), "matches_in_sample": int(n1),
    }, (k, int(n1))


def run_workload(name, args, rank, world, dist, do_cpu):
    import torch

    total = int(args.gib * (1 << 30))
    t0 = time.perf_counter()
    lits, corpus, off = build_workload(name, total, seed_shift=rank)
    log(f"[rank {rank}] {name}: generated {corpus.size} bytes / {off.size - 1} blocks in {time.perf_counter() - t0:.1f}s")
    job = GpuJob(lits, corpus, off, torch.cuda.current_device())
    info = job.table.info()

    # Optional software pipelining (--pipeline-depth 2): step i+1's scan is launched on a
    # second stream (its own scratch and record buffer, same table and resident corpus)
    # while step i's confirm / pack kernels and -- for N > 1 -- its record all-gather
    # finish on the first. Every step still does all of its work. Default: serial steps.
    depth = max(1, min(2, args.pipeline_depth))
    jobs = [job] + [GpuJob(None, None, None, torch.cuda.current_device(), sibling=job) for _ in range(depth - 1)]
    streams = [torch.cuda.Stream(device=job.dev) for _ in range(depth)]

    def gather(jb):
        """The path's one exchange step: RCCL all-gather of match records over xGMI
        (counts, then records padded to the largest count) -- hyperscan_amd/dist.py."""
        from hyperscan_amd import dist as hd

        n = min(int(jb.d_count.item()), jb.cap)  # waits for THIS job's stream only
        recs, base = jb.d_out.view(-1, 4), rank * jb.nblocks
        if args.exchange == "exact":  # no padding to the largest count (skewed shards)
            return hd.all_gather_records_exact(recs, n, base, dist, world, rank, jb.dev)
        if args.exchange == "root":  # only rank 0's host would deliver the callbacks
            return hd.gather_records_to_root(recs, n, base, dist, world, rank, 0, jb.dev)
        return hd.all_gather_records(recs, n, base, dist, world, jb.dev)

    def run_steps(n):
        for i in range(n):
            j = i % depth
            with torch.cuda.stream(streams[j]):
                jobs[j].launch()
            if world > 1:
                p = (i - (depth - 1)) % depth  # the step launched depth-1 steps ago
                if i >= depth - 1:
                    with torch.cuda.stream(streams[p]):
                        gather(jobs[p])
        if world > 1:  # drain: the last depth-1 steps still owe their gather
            for i in range(max(0, n - (depth - 1)), n):
                if depth > 1:
                    with torch.cuda.stream(streams[i % depth]):
                        gather(jobs[i % depth])
        for st in streams:
            st.synchronize()

    run_steps(max(args.warmup, depth))
    torch.cuda.synchronize()
    n_matches = job.count()
    assert n_matches <= job.cap, "record buffer too small"
    assert all(jb.count() == n_matches for jb in jobs)

    # parity gate on a sample (bounded CPU time): GPU count over the first k blocks == CPU count
    cpu = None
    if do_cpu:
        # records first: on this stack the scans that directly follow a large D2H copy run ~3x
        # slower for tens of ms (measured: 1.9 vs 0.66 ms per scan, host launch time unchanged);
        # the CPU baseline's seconds in between and one untimed scan keep that out of the timed region
        recs = job.records()
        cpu, (k, n_cpu) = cpu_baseline(lits, corpus, off)
        run_steps(1)
        torch.cuda.synchronize()
        n_gpu = int((recs[:, 0] < k).sum())
        assert n_gpu == n_cpu, f"PARITY FAILURE on the sample: GPU {n_gpu} vs CPU {n_cpu}"
        cpu["parity"] = f"GPU == CPU match count on the sample ({n_cpu})"

    # timed region: barrier + sync on both sides, exactly K steps
    filt_ms, conf_ms, pipe_ms = [], [], []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert all(jb.count() == n_matches for jb in jobs), "match count changed between repeats"  # hsbench main.cpp:778-787
    # HIP events the library recorded on the launch stream around its kernels during
    # the timed steps (ring of the last 32 scans); read only now, so the timed loop
    # itself never waited on them
    span_ms = []
    for jb in jobs:
        for back in range(min(args.steps // depth, 32)):
            f, c, t = jb.scratch.timing(back)
            filt_ms.append(f)
            conf_ms.append(c)
            pipe_ms.append(t)
            span_ms.append(jb.scratch.kernel_span(back))
    kern_avg_s = float(np.mean(filt_ms)) / 1e3

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=job.dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([job.total, n_matches], dtype=torch.int64, device=job.dev)
        dist.all_reduce(tot)
        all_bytes, all_matches = int(tot[0].item()), int(tot[1].item())
    else:
        all_bytes, all_matches = job.total, n_matches

    alg_bytes = job.total + REC_BYTES * n_matches
    traffic = pmc_traffic(info["flags"])
    achieved = alg_bytes / kern_avg_s / 1e9
    res = {
        "value": round(all_bytes * args.steps / dt / 1e9, 3),
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "matches_per_s": round(all_matches * args.steps / dt, 1),
        "matches_per_step": all_matches,
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": "hwlm_filter_kernel", "kernel_ms_avg": round(kern_avg_s * 1e3, 4),
            "kernel_ms_best": round(float(np.min(filt_ms)), 4),
            "algorithmic_bytes_per_launch": alg_bytes,
            # the kernel's own execution span from the device wall clock (what rocprofv3's kernel
            # trace reports); the HIP-event interval above also contains the dispatch gaps
            "kernel_ms_device_clock_avg": round(float(np.mean(span_ms)), 4),
            "achieved_device_clock": round(alg_bytes / (float(np.mean(span_ms)) / 1e3) / 1e9, 2),
            "confirm_stage_ms_avg": round(float(np.mean(conf_ms)), 4),
            "pipeline_ms_avg": round(float(np.mean(pipe_ms)), 4),
            "pipeline_GBps": round(alg_bytes / (float(np.mean(pipe_ms)) / 1e3) / 1e9, 2),
        },
        "table": info,
    }
    res["pipeline_depth"] = depth
    if do_cpu:
        # (never `value`) the same engine fed from HOST memory: hsgpu_hwlm_exec_batch = H2D of the
        # corpus + the pipeline above + D2H of the sorted records, on a bounded sample
        from hyperscan_amd import hwlm as hw

        k = int(np.searchsorted(off, min(256 << 20, int(off[-1])), side="right")) - 1
        s_off = np.ascontiguousarray(off[: k + 1])
        sample = np.ascontiguousarray(corpus[: int(s_off[-1])])
        hw.hwlm_exec_batch(job.table, job.scratch, sample, s_off, cap=job.cap)  # warm: buffers sized
        t0 = time.perf_counter()
        recs_h = hw.hwlm_exec_batch(job.table, job.scratch, sample, s_off, cap=job.cap)
        dt_h = time.perf_counter() - t0
        res["host_buffers"] = {"GBps": round(sample.size / dt_h / 1e9, 2), "sample_bytes": int(sample.size),
                               "matches": int(recs_h.size),
                               "what": "hsgpu_hwlm_exec_batch from pageable host memory: H2D of the corpus + scan + "
                                       "D2H of the sorted records; PCIe bound, reported for completeness only"}
    if cpu:
        res["cpu_baseline"] = cpu
    del job, jobs
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gib", type=float, default=1.0, help="corpus GiB per GPU")
    ap.add_argument("--workload", default="teddy64", choices=["teddy64", "fdr10k"])
    ap.add_argument("--no-also", action="store_true", help="skip the second workload line")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "exact", "root"],
                    help="N>1 record exchange: padded all-gather (default, what BASELINE names), exact-size "
                         "all-gather, or gather to rank 0 (hyperscan_amd/dist.py)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--pipeline-depth", type=int, default=1, choices=[1, 2],
                    help="scans in flight: 2 overlaps a step's confirm/pack/gather with the next step's filter "
                         "(+8..11%% throughput measured, but the per-kernel event timing then includes the overlap; "
                         "the default keeps steps serial so that the roofline figures are those of the kernel alone)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    do_cpu = (rank == 0 and world == 1 and not args.no_cpu)
    main_res = run_workload(args.workload, args, rank, world, dist, do_cpu)
    also = None
    if not args.no_also:
        other = "fdr10k" if args.workload == "teddy64" else "teddy64"
        also = run_workload(other, args, rank, world, dist, do_cpu)

    if rank == 0:
        blocks_desc = "synthetic packets {64,128,256,576,1024,1460} B, 70% HTTP-like text / 30% random"
        out = {
            "metric": "GB/s scanned (hsbench block mode)", "value": main_res["value"], "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: " + (
                "64 literals len 4-8" if args.workload == "teddy64" else "10000 snort-like literals (8-byte suffixes)")
                + f", {args.gib:g} GiB per GPU, block mode, {blocks_desc}",
                "records": "16 B (block,end,id,lit)", "pipeline_depth": main_res["pipeline_depth"],
                "sharding": f"{world} x independent shards"
                + (", RCCL all-gather of records per step" if world > 1 else "")},
            "matches_per_s": main_res["matches_per_s"], "matches_per_step": main_res["matches_per_step"],
            "roofline": main_res["roofline"], "table": main_res["table"],
        }
        if "cpu_baseline" in main_res:
            out["cpu_baseline"] = main_res["cpu_baseline"]
        if "host_buffers" in main_res:
            out["host_buffers"] = main_res["host_buffers"]
        if also:
            other = "fdr10k" if args.workload == "teddy64" else "teddy64"
            out["also"] = {other: {k: also[k] for k in also}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
