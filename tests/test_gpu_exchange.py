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
