"""Host-side model of the pair filter (hyperscan_amd/csrc/table.h, HSGPU_F_PAIR): reads a serialised table
and evaluates the filter kernel's pass rule with numpy. Test infrastructure: lets the CPU suite check that the
compiled filter never drops a position the oracle reports (no GPU needed), and counts its candidates."""
import struct

import numpy as np

F_PAIR = 256
FILTER_MUL = 0x9E3779
HEADER = struct.Struct("<22I")  # magic .. checksum (csrc/table.h HsgpuTableHeader), then hash_mask, n_m


def parse(blob):
    f = HEADER.unpack_from(blob, 0)
    h = dict(zip(("magic", "version", "blob_bytes", "flags", "n_lits", "max_size", "filter_log2", "filter_entries",
                  "ht_a_log2", "ht_b_log2", "n_a", "n_b", "n_c", "off_filter", "off_c2bits", "off_ht_a", "off_ht_b",
                  "off_c2ref", "off_lists", "n_lists", "off_lits", "checksum"), f))
    h["hash_mask"], h["n_m"] = struct.unpack_from("<2I", blob, HEADER.size)
    return h


def candidates(blob, corpus):
    """Boolean array over the even positions q = 0, 2, 4, ...: does the pair filter pass there?
    Bytes outside the corpus read as zero, as in the kernel."""
    h = parse(blob)
    assert h["flags"] & F_PAIR, "not a pair table"
    k = h["filter_log2"]
    ent = np.frombuffer(blob, dtype="<u4", count=2 << k, offset=h["off_filter"]).reshape(-1, 2)
    B, A = ent[:, 0], ent[:, 1]
    n = corpus.size
    pad = np.zeros(n + 8, dtype=np.uint32)
    pad[4:4 + n] = corpus
    q = np.arange(0, n, 2) + 4
    b0, b1, b2, b3, nx = pad[q], pad[q - 1], pad[q - 2], pad[q - 3], pad[q + 1]
    x = (b2 | b1 << 8 | b0 << 16) & np.uint32(h["hash_mask"])
    prod = ((x.astype(np.uint64) * FILTER_MUL) & 0xFFFFFFFF).astype(np.uint32)
    e = prod >> np.uint32(32 - k)
    tb = (B[e] >> (b3 & 31)) & (B[e] >> ((b3 >> 3) & 31))
    ta = A[e] >> (nx & 31)
    th = A[e] >> ((prod >> 8) & 31)
    return ((tb | ta) & th & 1).astype(bool), h
