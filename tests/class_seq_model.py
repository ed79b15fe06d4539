"""CPU models of the class-sequence patterns A{m,}B{n,} (hyperscan_amd/csrc/class_seq.hip): test infrastructure.

  ends_re     Python `re`, one block at a time: the pattern reversed, a zero-width look-ahead at every start of the
              reversed match = every end of the forward match (all-matches semantics, as hs_scan reports `to`)
  ends_numpy  an independent restatement with run lengths (no bit tricks): per block, per position, the length of
              the run of A ending before it and the B-run reaching it
Both return, per pattern, a sorted array of (block, end) with `end` the offset of the last byte of the match."""
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
