"""tools/sim/pk16.py -- verdict r3 item 1(d): price a packed-16-bit (VOP3P, v_pk_mul_lo_u16) hash for the fdr10k filter.

Two questions: (1) what does a hash built from 16-bit pair products pass (candidates per GiB, against the shipped 24-bit
multiply's 8.5 M), (2) how many vector instructions does it take per lookup position. The second is counted in the docstring
of `instruction_count`; the first is simulated here on 64 MiB of the bench corpus with the bench's 10 000 literals.

  H(q) = (P1(q - 1) + P2(q)) mod 2^16,  P1(i) = pair(i) * M1, P2(i) = pair(i) * M2 (low 16 bits), pair(i) = c[i-1] | c[i] << 8
  entry  = H >> 2 (14 bits, the shipped table size), lo bit = b3 & 31 (as shipped), hi bit = a second 16-bit sum with other multipliers
"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo/tools/sim'); sys.path.insert(0, '/root/repo')
import base
from base import Lit


def instruction_count():
    """Per 16-byte chunk (16 lookup positions), shipped loop vs packed-16:
    shipped : 12 x (alignbyte | shift) to bring the 3 bytes to bit 0, 16 v_mul_u32_u24, 16 v_lshrrev (entry), 16 v_and (address)   = 60
              + per position 2 shifts (lo >> b3, hi >> prod), 1 and, 1 alignbit                                                    = 64
    packed  : pairs at even byte offsets are the dword's halves, pairs at odd offsets need the dword shifted by 8: 4 v_alignbyte;
              P1, P2 of 8 dwords: 16 v_pk_mul_lo_u16; the odd pairs' P1 / P2 must meet the even pairs' in the SAME half: 8 more
              v_alignbit / v_perm to shift one of them by 16; 8 v_pk_add_u16 for H                                                 = 36
              the second index (hi bit) from other multipliers: + 16 v_pk_mul + 8 v_pk_add                                         = 24
              and then EVERY position needs its hash as a 32-bit LDS byte address of its own: v_and (low half) or v_lshrrev (high
              half) + the << 1 that makes a 14-bit index an 8-byte offset, which the 32-bit form folds into its one shift: 32      = 32
              + per position the same 4 test instructions                                                                          = 64
    60 + 64 = 124 (shipped) vs 36 + 24 + 32 + 64 = 156 (packed).  The multiply is 16 of the shipped loop's 138 instructions per
    tile; halving THEM cannot pay for moving 16-bit halves into address registers. (VOP3P issues at the v_mul_u32_u24 rate:
    profiles/r02_valu_lds_rates.txt.)"""


def main():
    N = 64 << 20
    lits, corpus, off = base.load(N)
    true_e = base.true_ends(lits, corpus, off)
    L = [Lit(l) for l in lits]
    pad = base.padded(corpus).astype(np.uint32)
    n = corpus.size
    M1, M2, M3, M4 = 0x9E37, 0x85EB, 0xC2B3, 0x27D5
    lo_t = np.zeros(1 << 14, dtype=np.uint32)
    hi_t = np.zeros(1 << 14, dtype=np.uint32)

    def H(p_prev, p_cur, Ma, Mb):
        return ((p_prev * Ma) + (p_cur * Mb)) & 0xffff
    for li in L:
        v = [li.val[p] & 0xdf for p in range(4)]
        pair_cur = v[1] | v[0] << 8   # c[q-1] | c[q] << 8
        pair_prev = v[2] | v[1] << 8  # c[q-2] | c[q-1] << 8
        h, h2 = H(pair_prev, pair_cur, M1, M2), H(pair_prev, pair_cur, M3, M4)
        if li.len >= 4:
            lo_t[h >> 2] |= np.uint32(1 << (v[3] & 31))
            hi_t[h >> 2] |= np.uint32(1 << (h2 & 31))
        else:  # folded 3-byte key: owns the lo half
            lo_t[h >> 2] = 0xffffffff
            hi_t[h >> 2] |= np.uint32(1 << (h2 & 31))
    pos = np.arange(n)
    b = [pad[8 + pos - k] & 0xdf for k in range(4)]
    pair_cur = b[1] | b[0] << 8
    pair_prev = b[2] | b[1] << 8
    h, h2 = H(pair_prev, pair_cur, M1, M2), H(pair_prev, pair_cur, M3, M4)
    hit = (((lo_t[h >> 2] >> (b[3] & 31)) & (hi_t[h >> 2] >> (h2 & 31))) & 1).astype(bool)
    assert hit[true_e].all(), "recall"
    lanes = np.zeros(n // 16 + 1, dtype=bool)
    lanes[pos[hit] >> 4] = True
    print(f"packed-16 pair-product hash: {hit.sum() * 16 / 1e6:.2f} M candidate positions / GiB, {lanes.sum() * 16 / 1e6:.2f} M candidate entries / GiB "
          f"(shipped WIDE filter: 8.52 M entries / GiB on the device)")
    print(instruction_count.__doc__)


if __name__ == "__main__":
    t = time.time()
    main()
    print(f"{time.time() - t:.0f} s")
