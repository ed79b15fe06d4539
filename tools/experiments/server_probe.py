"""tools/experiments/server_probe.py -- the small-batch server, step by step with a watchdog (GPU box)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(40, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import hyperscan_amd as H
from tests import oracle_binding as ob
from tests.util import random_corpus, random_literals
rng = np.random.default_rng(1)
lits = random_literals(rng, 64, 4, 8)
t = H.hwlm_build(lits)
s = H.Scratch(0)
s.enable_server(True, idle_us=int(sys.argv[1]) if len(sys.argv) > 1 else 300)
for total in (1460, 100, 16384, 1460):
    pkt = random_corpus(rng, total, lits, plant_every=97)
    want = sorted(ob.Oracle(lits).collect(pkt))
    for rep in range(3):
        g = []
        t0 = time.perf_counter()
        rv = H.hwlm_exec(t, pkt, 0, lambda e, i, c: g.append((e, i)) or H.HWLM_CONTINUE_MATCHING, s)
        print(total, rep, "rv", rv, "ok" if sorted(g) == want else f"MISMATCH {len(g)} vs {len(want)}", f"{(time.perf_counter() - t0) * 1e6:.1f} us", s.server_stats(), flush=True)
import ctypes as C
lib = t._lib
cu, su = C.c_float(), C.c_float()
lib.hsgpu_scratch_server_last_us.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.hsgpu_debug_server_stamping.argtypes = [C.c_void_p, C.c_int]
lib.hsgpu_debug_server_stamping(s._h, 1)  # (the device-side times are taken only for requests made while stamping is on)
H.hwlm_exec(t, pkt, 0, lambda e, i, c: H.HWLM_CONTINUE_MATCHING, s)
lib.hsgpu_debug_server_stamping(s._h, 0)
lib.hsgpu_scratch_server_last_us(s._h, C.byref(cu), C.byref(su))
print("device copy us", cu.value, "scan us", su.value, flush=True)
print("closing", flush=True)
s.close()
print("done", flush=True)
