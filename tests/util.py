"""Shared generators for the parity tests (seeded, deterministic)."""
import numpy as np

import hyperscan_amd as H

ALNUM = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)


def random_literals(rng, n, lo=1, hi=8, alphabet=ALNUM, nocase_frac=0.3, dup_ids=False):
    lits, seen = [], set()
    while len(lits) < n:
        ln = int(rng.integers(lo, hi + 1))
        s = bytes(rng.choice(alphabet, ln))
        nocase = bool(rng.random() < nocase_frac)
        key = (s.upper() if nocase else s, nocase)
        if key in seen:
            continue
        seen.add(key)
        lid = len(lits) // 2 if dup_ids else len(lits)
        lits.append(H.HwlmLiteral(s, nocase=nocase, id=lid))
    return lits


def random_corpus(rng, nbytes, lits, alphabet=ALNUM, plant_every=4096):
    corpus = rng.choice(alphabet, nbytes).astype(np.uint8)
    nplant = max(1, nbytes // plant_every)
    for _ in range(nplant):
        l = lits[int(rng.integers(0, len(lits)))].s
        if len(l) >= nbytes:
            continue
        p = int(rng.integers(0, nbytes - len(l)))
        b = np.frombuffer(l, dtype=np.uint8).copy()
        if rng.random() < 0.5:  # flip case to exercise nocase
            up = (b >= 97) & (b <= 122)
            b[up] -= 32
        corpus[p:p + len(l)] = b
    return corpus


def random_blocks(rng, total, mean_len=600, allow_empty=True):
    cuts = [0]
    while cuts[-1] < total:
        ln = int(rng.integers(0 if allow_empty else 1, 2 * mean_len))
        cuts.append(min(total, cuts[-1] + ln))
    return np.asarray(cuts, dtype=np.uint64)


def as_set(recs):
    return sorted(zip(recs["block"].tolist(), recs["end"].tolist(), recs["id"].tolist()))


def do_accel_block_model(L, kind, sc, offset, blk, start):
    """do_accel_block (src/hwlm/hwlm.c:80-99) restated with the oracle's accelerators as run_hwlm_accel (`L` = the
    oracle library, `kind` / `sc` = ForwardAccel.scanner()): the expectation of the GPU test of
    hsgpu_hwlm_forward_skip_dev. Pinned to the reference's own do_accel_block in tests/test_accel_build.py."""
    n = blk.size
    if n - start < 16:
        return start
    tail = np.ascontiguousarray(blk[start:])
    if kind == "class":
        hit = L.hso_class_fwd(sc.bitmap.ctypes.data, tail.ctypes.data, tail.size)
    else:
        hit = L.hso_dshufti_fwd(*sc.masks, tail.ctypes.data, tail.size)
    return max(0, start + hit - offset)
