"""CPU models of the class-sequence patterns A{m,}B{n,} (hyperscan_amd/csrc/class_seq.hip): test infrastructure.

  ends_re     Python `re`, one block at a time: the pattern reversed, a zero-width look-ahead at every start of the
              reversed match = every end of the forward match (all-matches semantics, as hs_scan reports `to`)
  ends_numpy  an independent restatement with run lengths (no bit tricks): per block, per position, the length of
              the run of A ending before it and the B-run reaching it
  ends_vec    the same statement without a Python loop over bytes (prefix sums over the whole batch): what bench.py gates
              megabytes with; pinned to the two above in tests/test_class_seq_cpu.py
All return, per pattern, a sorted array of (block, end) with `end` the offset of the last byte of the match."""
import re

import numpy as np


def _cls_re(members):
    return "[" + "".join("\\x%02x" % c for c in sorted(set(members))) + "]"


def ends_re(block, a_members, b_members, m, n):
    """end offsets (last byte) of every match of A{m,}B{n,} in one block (bytes)"""
    rev = block[::-1]
    pat = re.compile(("(?=%s{%d,}%s{%d})" % (_cls_re(b_members), n, _cls_re(a_members), m)).encode("latin-1"), re.S)
    L = len(block)
    # a reversed match starting at r covers the forward byte L-1-r as its last byte -- but B{n,} is greedy-free
    # here: the look-ahead asks for "at least n of B, then m of A" starting at r, and {n,} may stop early, so
    # every r that is the end of SOME match is found
    return np.array(sorted(L - 1 - mt.start() for mt in pat.finditer(rev)), dtype=np.int64)


def ends_numpy(corpus, off, a_members, b_members, m, n):
    """(block, end) of every match end over a CSR batch, by run lengths"""
    isa = np.isin(corpus, np.array(sorted(set(a_members)), dtype=np.uint8))
    isb = np.isin(corpus, np.array(sorted(set(b_members)), dtype=np.uint8))
    out = []
    for blk in range(len(off) - 1):
        lo, hi = int(off[blk]), int(off[blk + 1])
        run_a = 0      # members of A immediately before the current position
        best = 0       # B-run length from the EARLIEST admissible split reaching the previous position (0 = none)
        for i in range(lo, hi):
            if isb[i]:
                # extend an admissible run, or start one here if >= m of A precede
                best = best + 1 if best else (1 if run_a >= m else 0)
                if best >= n:
                    out.append((blk, i - lo))
            else:
                best = 0
            # a later split can only shorten the run; the earliest admissible one dominates -- but when the
            # earliest run broke (not B) a new one may start at a position that is both B and preceded by A's
            run_a = run_a + 1 if isa[i] else 0
    return np.array(out, dtype=np.int64).reshape(-1, 2)


class VecModel:
    """ends_vec over one CSR batch for many patterns: the per-class run lengths are computed once."""

    def __init__(self, corpus, off):
        self.corpus = np.ascontiguousarray(corpus)
        self.off = np.asarray(off, dtype=np.int64)
        N = int(self.corpus.size)
        lens = np.diff(self.off)
        self.blk = np.repeat(np.arange(lens.size, dtype=np.int64), lens)          # block of every byte
        self.bstart = np.repeat(self.off[:-1], lens)                              # its block's first byte
        self.idx = np.arange(N, dtype=np.int64)
        self._runs = {}

    def run(self, members):
        """length of the run of class members ending at every byte, inside its block (0 where the byte is no member)"""
        key = bytes(sorted(set(members)))
        if key not in self._runs:
            mem = np.isin(self.corpus, np.frombuffer(key, dtype=np.uint8))
            last_non = np.maximum.accumulate(np.where(mem, -1, self.idx))          # last non-member at or before i (-1: none)
            r = np.where(mem, np.minimum(self.idx - last_non, self.idx - self.bstart + 1), 0)
            self._runs[key] = r
        return self._runs[key]

    def ends(self, a_members, b_members, m, n):
        N = int(self.corpus.size)
        if N == 0:
            return np.zeros((0, 2), dtype=np.int64)
        ra, rb = self.run(a_members), self.run(b_members)
        # G[s]: a split may sit in front of byte s -- at least m members of A end at s - 1, in the same block as s
        G = np.zeros(N + 1, dtype=np.int64)
        G[1:N] = (ra[: N - 1] >= m) & (self.bstart[1:] != self.idx[1:])
        P = np.concatenate([[0], np.cumsum(G)])                                    # P[k] = splits in front of bytes < k
        e = np.flatnonzero(rb >= n)
        lo, hi = e - rb[e] + 1, e - n + 1                                          # the B-run may start at any s in [lo, hi]
        ok = P[hi + 1] - P[lo] > 0
        e = e[ok]
        return np.stack([self.blk[e], e - self.bstart[e]], axis=1)


def ends_vec(corpus, off, a_members, b_members, m, n):
    return VecModel(corpus, off).ends(a_members, b_members, m, n)
