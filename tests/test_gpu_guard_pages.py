"""No kernel reads or writes outside the buffers it was given (round 6).

The reference goes to great lengths never to touch a byte outside [buf, buf + len): FDR's zones copy heads and tails into a
padded stack buffer (src/fdr/fdr.c:392-690), Teddy's vectoredLoad* do the same (src/fdr/teddy_runtime_common.h:126-391), and
unit/internal/fdr.cpp:496-561 scans at every alignment. Here in-bounds access is by construction -- and hipMalloc's 2 MiB
granules would hide a violation. These tests take the hiding place away: every buffer the C ABI takes (corpus, offsets,
records, count, class bitmaps, first / last arrays, work areas, start arrays, exchange record buffers) is placed by
hsgpu_debug_guard_malloc so that it STARTS behind an unmapped page ("front") or ENDS in front of one ("back"), and with
hsgpu_debug_guard_mode every buffer the library allocates for itself (candidate regions, staging regions, control blocks,
hints, table images, exchange slots) is placed the same way and sized exactly. One byte too far is a page fault, and a page
fault kills the process -- so every family runs in a child process (tests/guard_worker.py), which also checks every result
against the oracle. The first three tests prove that the mechanism bites on this box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(case, mode, timeout=900):
    env = dict(os.environ)
    env.pop("HSGPU_MODE", None)  # the worker chooses its pipelines itself
    r = subprocess.run([sys.executable, "-m", "tests.guard_worker", case, mode], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = r.stdout.strip().splitlines()
    try:  # the whole story of a child that died, where gpurun brings it back from
        d = os.path.join(ROOT, "gpurun_out", "guard")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"{case.replace(':', '_')}_{mode}.log"), "w") as f:
            f.write(f"rc={r.returncode}\n--- stdout (last 40 lines)\n" + "\n".join(lines[-40:]) + "\n--- stderr (tail)\n" + r.stderr[-6000:])
    except OSError:
        pass
    return r.returncode, lines, r.stderr[-3000:]


@pytest.mark.parametrize("which", ["read_past_end", "write_past_end", "read_before_start"])
def test_guard_mechanism_faults_on_one_byte(which):
    """a kernel touching ONE byte outside a guard buffer must not get away with it on this box: the child dies (GPU memory
    access fault) or the runtime reports an error -- it must never return success"""
    mode = "front" if which == "read_before_start" else "back"
    rc, lines, err = run_worker("probe:" + which, mode, timeout=300)
    assert rc != 7 and "GUARD-OK" not in lines, f"the out-of-bounds access went unnoticed: rc={rc}\n{lines}\n{err}"
    assert rc != 0 or any("probe returned" in l for l in lines), (rc, lines, err)


def test_guard_mechanism_allows_in_bounds_access():
    rc, lines, err = run_worker("probe:inside", "back", timeout=300)
    assert rc == 0 and lines and lines[-1] == "GUARD-OK", (rc, lines[-3:], err)


CASES = ["literal:default", "literal:fused", "literal:unfolded", "literal:nosolo", "literal:solo", "literal_dense", "host_entry",
         "class_scan", "pair_scan", "class_seq", "accel", "exchange"]


@pytest.mark.parametrize("mode", ["back", "front"])
@pytest.mark.parametrize("case", CASES)
def test_every_buffer_against_an_unmapped_page(case, mode):
    rc, lines, err = run_worker(case, mode)
    last = lines[-1] if lines else "(no output)"
    assert rc == 0 and last == "GUARD-OK", (f"{case} [{mode}] died or failed (rc {rc}) after {len(lines)} sub-cases; last: {last!r}\n"
                                            f"stderr tail:\n{err}")
