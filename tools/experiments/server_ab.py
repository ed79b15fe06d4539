"""tools/experiments/server_ab.py -- microseconds per hsgpu_hwlm_exec call (1 460-byte packet, fdr10k table, a native loop:
hsgpu_debug_exec_repeat) through the small-batch server, for the library HSGPU_LIB_VARIANT names: the A/B of two builds on one box
(`for v in "" _prev; do HSGPU_LIB_VARIANT=$v python tools/experiments/server_ab.py; done`). GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import hyperscan_amd as H  # noqa: E402
from hyperscan_amd import corpus as cp  # noqa: E402
from hyperscan_amd import hwlm as hw  # noqa: E402

lits = cp.snort_like_literals(10000, seed=4)[0]
t = H.hwlm_build(lits)
corpus, off = cp.packet_corpus(1 << 22, lits, seed=3)
sizes = np.diff(off.astype(np.int64))
lib = t._lib
lib.hsgpu_debug_exec_repeat.restype = C.c_int
lib.hsgpu_debug_exec_repeat.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, hw.HWLM_CB, C.c_void_p, C.c_uint64, C.c_uint, C.POINTER(C.c_double)]
s = H.Scratch(0)
ncb = C.c_uint64(0)
count_cb = C.cast(lib.hsgpu_hwlm_count_cb, hw.HWLM_CB)
lib.hsgpu_scratch_set_context(s._h, C.addressof(ncb))
out = []
for want, label in ((1460, "1460 B"), (int(sizes[sizes < 600].max()), "small")):
    k = int(np.argmax(sizes == want))
    pkt = np.ascontiguousarray(corpus[int(off[k]):int(off[k + 1])])
    for kind in (1, 2):
        s.enable_server(kind)
        us = C.c_double(0)
        best = 1e9
        for _ in range(3):
            assert lib.hsgpu_debug_exec_repeat(t._h, pkt.ctypes.data, pkt.size, 0, count_cb, s._h, hw.HWLM_ALL_GROUPS, 3000, C.byref(us)) == 0
            best = min(best, us.value)
        out.append(f"{label} {'bar' if kind == 1 else 'host'} {best:.2f}")
s.enable_server(False)
print(f"variant[{os.environ.get('HSGPU_LIB_VARIANT', '')}] us per call: " + "; ".join(out) + f"; matches seen {ncb.value}")
s.close()
