"""tools/experiments/rose_resident_probe.py -- where hs_scan_batch_resident's time goes on config 5 (GPU box)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hyperscan_amd import corpus as cp, hs
from tests import rose_model as RM

rng = np.random.default_rng(6)
alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
lits = sorted({bytes(rng.choice(alpha, int(rng.integers(6, 13)))) for _ in range(1000)})
pats = [l.decode() + RM.TAILS[i % 3] for i, l in enumerate(lits)]
db = hs.Database.compile(pats, [0] * len(pats), list(range(len(pats))))
scratch = hs.HsScratch(db)
class L:
    def __init__(self, s): self.s = s
follow = [b"abc7", b"  key=", b"....END"]
plant = [L(l + follow[i % 3]) if i % 2 == 0 else L(l) for i, l in enumerate(lits)]
corpus, off = cp.packet_corpus(2 << 30, plant, seed=6, match_every=4096)
offs = np.ascontiguousarray(off, dtype=np.uint64)
pinned = torch.from_numpy(corpus).pin_memory(); buf = pinned.numpy()
d_corpus = torch.from_numpy(corpus).to("cuda:0"); d_off = torch.from_numpy(offs.view(np.int64)).to("cuda:0")
lib = hs._lib()
handler = C.cast(lib.hs_batch_count_handler, hs.BATCH_CB)
lib.hs_scan_batch_resident.restype = C.c_int
lib.hs_scan_batch_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p, hs.BATCH_CB, C.c_void_p]
tm = (C.c_double * 5)()
for which, data, thr in (("pinned", buf, 8), ("pinned", buf, 16), ("pinned", buf, 32), ("pinned", buf, 64), ("pinned", buf, 128), ("pinned", buf, 0), ("pageable", corpus, 0)):
    lib.hsgpu_debug_confirm_threads(thr)
    which = f"{which} host copy, {thr or 'default'} threads"
    for rep in range(4):
        cnt = C.c_ulonglong(0)
        t0 = time.perf_counter()
        rv = lib.hs_scan_batch_resident(db._h, data.ctypes.data, offs.ctypes.data, offs.size - 1, d_corpus.data_ptr(), d_off.data_ptr(), scratch._h, handler, C.byref(cnt))
        dt = time.perf_counter() - t0
        lib.hsgpu_debug_confirm_timing(tm)
        print(f"{which}: {dt * 1e3:.2f} ms rc {rv} events {cnt.value}; confirm: setup %.2f parallel %.2f (slowest slice %.2f fastest %.2f) delivery %.2f ms" % tuple(x * 1e3 for x in tm), flush=True)
