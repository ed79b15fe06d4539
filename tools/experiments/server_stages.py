"""tools/experiments/server_stages.py -- the small-batch server on the bench's fdr10k table and a 1 460-byte packet: microseconds per
hsgpu_hwlm_exec call with the request mailbox in device memory (1) and in mapped host memory (2), and the last request's stages on
the device (hsgpu_debug_server_stamps). GPU box."""
import ctypes as C
import faulthandler
import os
import sys
import time

faulthandler.dump_traceback_later(120, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import hyperscan_amd as H  # noqa: E402
from hyperscan_amd import corpus as cp  # noqa: E402
from hyperscan_amd import hwlm as hw  # noqa: E402

lits = cp.snort_like_literals(10000, seed=4)[0]
t = H.hwlm_build(lits)
corpus, off = cp.packet_corpus(1 << 22, lits, seed=3)
k = int(np.argmax(np.diff(off.astype(np.int64)) == 1460))
pkt = np.ascontiguousarray(corpus[int(off[k]):int(off[k + 1])])
lib = t._lib
s = H.Scratch(0)
ncb = C.c_uint64(0)
count_cb = C.cast(lib.hsgpu_hwlm_count_cb, hw.HWLM_CB)
lib.hsgpu_scratch_set_context(s._h, C.addressof(ncb))
lib.hsgpu_debug_server_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
lib.hsgpu_debug_server_head_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
lib.hsgpu_debug_server_stamping.argtypes = [C.c_void_p, C.c_int]
lib.hsgpu_scratch_server_last_us.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
p, n = pkt.ctypes.data, pkt.size


def per_call(calls=3000):
    for _ in range(20):
        assert lib.hsgpu_hwlm_exec(t._h, p, n, 0, count_cb, s._h, hw.HWLM_ALL_GROUPS) == 0
    t0 = time.perf_counter()
    for _ in range(calls):
        lib.hsgpu_hwlm_exec(t._h, p, n, 0, count_cb, s._h, hw.HWLM_ALL_GROUPS)
    return (time.perf_counter() - t0) / calls * 1e6


print(f"launch per call: {per_call(500):.2f} us; matches per call {ncb.value / 520:.1f}")
for kind in (2, 1, 2, 1):
    s.enable_server(kind)
    ncb.value = 0
    us = per_call()
    lib.hsgpu_debug_server_stamping(s._h, 1)  # (the stamps cost: taken on a few calls of their own)
    us_stamped = per_call(300)
    lib.hsgpu_debug_server_stamping(s._h, 0)
    st = (C.c_float * 3)()
    lib.hsgpu_debug_server_stamps(s._h, st)
    cu, su = C.c_float(), C.c_float()
    lib.hsgpu_scratch_server_last_us(s._h, C.byref(cu), C.byref(su))
    hd = (C.c_float * 2)()
    lib.hsgpu_debug_server_head_stamps(s._h, hd)
    print(f"  head: loads out at {hd[0]:.2f} us, in front of the barrier at {hd[1]:.2f} us")
    print(f"server, mailbox {'in device memory (BAR)' if kind == 1 else 'in mapped host memory'}: {us:.2f} us per call ({us_stamped:.2f} with stamps; {ncb.value / 3340:.1f} matches); on the device "
          f"{su.value:.2f} us: image + first tiles {st[0]:.2f}, filter + confirm {st[1]:.2f}, placement {st[2]:.2f}; stats {s.server_stats()}")
s.enable_server(False)
s.close()
