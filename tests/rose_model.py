"""CPU model of the bench's config-5 patterns LIT_k + tail (bench.py run_rose1000): test infrastructure.

expected_events() says which (block, id, to) hs_scan_batch must report for the pattern set `lits[i] + TAILS[i % 3]`
(id = i) over a CSR batch: the literal occurrences come from the HWLM oracle (oracle/hwlm_oracle.c, the last <= 8
bytes of each literal, then the whole literal compared), the three tails are restated directly:
    [a-z]+\\d        one end: the digit that follows the maximal run of lower-case letters
    \\s+\\w{2,8}=     one end: '=' after a white-space run and a word of 2..8 characters (\\s, \\w and '=' are disjoint)
    .{0,16}END      every END that starts 0..16 bytes after the literal without crossing a newline
tests/test_rose_model_cpu.py pins it to Python `re` (all-ends semantics by brute force) and to hs_confirm_batch."""
import numpy as np

TAILS = [r"[a-z]+\d", r"\s+\w{2,8}=", r".{0,16}END"]
_WS = frozenset(b" \t\n\r\x0b\x0c")
_WORD = frozenset(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_")
_LOWER = frozenset(b"abcdefghijklmnopqrstuvwxyz")
_DIGIT = frozenset(b"0123456789")


def tail_ends(blk, s, kind):
    """ends `to` (exclusive) of TAILS[kind] matched from offset s of the block (bytes)"""
    L = len(blk)
    if kind == 0:
        j = s
        while j < L and blk[j] in _LOWER:
            j += 1
        return [j + 1] if j > s and j < L and blk[j] in _DIGIT else []
    if kind == 1:
        j = s
        while j < L and blk[j] in _WS:
            j += 1
        if j == s:
            return []
        k = j
        while k < L and blk[k] in _WORD:
            k += 1
        return [k + 1] if 2 <= k - j <= 8 and k < L and blk[k] == 0x3D else []
    out = []
    for k in range(0, 17):
        if s + k + 3 > L:
            break
        if k and blk[s + k - 1] == 0x0A:
            break
        if blk[s + k:s + k + 3] == b"END":
            out.append(s + k + 3)
    return out


def expected_events(corpus, off, lits, hits):
    """hits: the oracle's records (fields block, end, id) for the literals' HWLM suffixes over this batch.
    -> sorted list of (block, id, to)"""
    ev = set()
    c = corpus.tobytes() if isinstance(corpus, np.ndarray) else bytes(corpus)
    for b, e, i in zip(hits["block"].tolist(), hits["end"].tolist(), hits["id"].tolist()):
        lo, hi = int(off[b]), int(off[b + 1])
        lit = lits[i]
        s = e + 1
        if s < len(lit) or c[lo + s - len(lit):lo + s] != lit:
            continue
        for to in tail_ends(c[lo:hi], s, i % 3):
            ev.add((b, i, to))
    return sorted(ev)
