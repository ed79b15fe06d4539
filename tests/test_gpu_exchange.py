"""The N > 1 step on a real device: RCCL (backend "nccl") at world size 1 -- the one multi-GPU
configuration a 1-GPU box can run. The collectives, the device-resident counts and the global block
base all take the same code path as at N = 8 (tests/test_dist_cpu.py covers world size 2 over gloo)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(400)
def test_record_exchange_over_rccl_world_size_1():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "exchange_worker.py"), "nccl", "0", "1", str(_free_port())],
                       cwd=root, env=env, capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "EXCHANGE_OK nccl 1" in r.stdout


@pytest.mark.timeout(400)
def test_exact_exchange_over_rccl_world_size_1():
    """the unpadded step (hyperscan_amd.dist.ExactExchange: broadcasts of exactly counts[r] rows) over RCCL"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "exchange_worker.py"), "nccl", "0", "1", str(_free_port()), "exact"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "EXCHANGE_OK nccl 1" in r.stdout


@pytest.mark.timeout(500)
def test_bench_multi_gpu_path_at_world_size_1():
    """bench.py's N > 1 code path end to end (process group, fixed-size exchange inside the timed steps, two
    alternating streams, the all-reduced totals, the `exchange` object of the JSON line) over RCCL at world
    size 1: HSGPU_BENCH_FORCE_DIST takes that path without a second GPU."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HSGPU_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--gib", "0.0625", "--steps", "6",
                        "--warmup", "2", "--no-cpu", "--no-also"], cwd=root, env=env, capture_output=True, text=True, timeout=450)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert rows and rows[-1].startswith("{"), "the JSON line is not the last line of stdout: " + repr(r.stdout[-1500:])
    assert sum(ln.startswith("{") for ln in rows) == 1, "more than one JSON line"
    line = json.loads(rows[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["value"] > 0
    assert line["config"]["pipeline_depth"] == 2 and "RCCL" in line["config"]["sharding"]
    ex = line["exchange"]  # (default: the C ABI's exchange, hsgpu_exchange_step; round 5: `value` over the all-gather, to-root timed beside it)
    assert "hsgpu_exchange_step" in ex["collective"] and ex["value_uses"] == "all_gather" and "all-gather" in line["config"]["sharding"]
    assert ex["all_gather"]["gather_ms_avg_rank0"] > 0 and ex["all_gather"]["wire_bytes_received_rank0"] == 0
    assert ex["to_root"]["ms_per_step"] > 0
    mg = line["multi_gpu"]
    assert mg["measured"] is True and mg["n_gpus"] == 1 and len(mg["per_rank_GBps"]) == 1 and mg["per_rank_GBps"][0] > 0
    assert line["env_knobs"] == {"HSGPU_BENCH_FORCE_DIST": "1"}
    # and the torch.distributed form it falls back to
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--gib", "0.0625", "--steps", "4", "--warmup", "2",
                        "--no-cpu", "--no-also", "--exchange", "padded"], cwd=root, env=dict(env, MASTER_PORT=str(_free_port())),
                       capture_output=True, text=True, timeout=450)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["exchange"]["rows_per_rank"] >= line["matches_per_step"] and line["exchange"]["gather_ms_avg_rank0"] > 0


@pytest.mark.timeout(500)
def test_bench_strong_scaling_mode_and_exact_exchange_at_world_size_1():
    """bench.py --scaling strong (the rank scans its run of the fixed set of shards) with the exact-size exchange, over
    RCCL at world size 1; and the strong corpus is the concatenation of the weak shards: twice the shard's matches."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HSGPU_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--gib", "0.03125", "--steps", "4", "--warmup", "2",
                        "--no-cpu", "--no-also", "--scaling", "strong", "--shards", "2", "--exchange", "exact"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=450)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and "2 shards" in line["config"]["sharding"]
    assert "exactly counts[r] rows" in line["exchange"]["collective"]
    from bench import build_shards, build_workload

    _l, c0, o0 = build_workload("fdr10k", 1 << 25, 0)
    _l, c1, o1 = build_workload("fdr10k", 1 << 25, 1)
    _l, c, o = build_shards("fdr10k", 1 << 25, [0, 1])
    import numpy as np

    assert np.array_equal(c, np.concatenate([c0, c1])) and o.size == o0.size + o1.size - 1 and int(o[-1]) == 1 << 26
    assert np.array_equal(o[o0.size - 1:], o1 + np.uint64(1 << 25))


@pytest.mark.timeout(300)
def test_bench_refuses_to_run_fewer_gpus_than_asked_for():
    """`python bench.py --gpus 2` with no launcher starts its ranks itself (torch.distributed.run); on a box with one GPU it
    must fail loudly -- round 4's bench ran ONE GPU there and printed n_gpus: 1. The same under a launcher whose WORLD_SIZE
    does not match --gpus."""
    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 1), "--gib", "0.03125", "--steps", "2", "--no-cpu",
                        "--no-also"], cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--gib", "0.03125", "--steps", "2", "--no-cpu", "--no-also"],
                       cwd=root, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
