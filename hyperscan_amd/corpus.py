"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d): literal sets and block
corpora, seeded and deterministic, shared byte-for-byte by the GPU run and the
CPU baseline. Pure numpy; nothing here is on the scan path.

hsbench reads its corpora from SQLite (tools/hsbench/data_corpus.cpp:69-136; reader and
writer for that format: tools/hsbench.py); what it yields -- an ordered list of independent
blocks -- is what these generators produce in CSR form (one contiguous byte array +
nblocks+1 offsets).
"""
import numpy as np

from .hwlm import HwlmLiteral

ALNUM = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
PRINTABLE = np.arange(0x20, 0x7F, dtype=np.uint8)

# packet-length mix of SURVEY section 8(d) config 2
PACKET_LENS = np.array([64, 128, 256, 576, 1024, 1460])
PACKET_PROB = np.array([0.30, 0.10, 0.10, 0.15, 0.10, 0.25])


def teddy_literals(n=64, seed=2):
    """Config 2: n literals, lengths uniform 4-8 over [a-z0-9], caseful, ids 0..n-1."""
    rng = np.random.default_rng(seed)
    lits, seen = [], set()
    while len(lits) < n:
        s = bytes(rng.choice(ALNUM, int(rng.integers(4, 9))))
        if s in seen:
            continue
        seen.add(s)
        lits.append(HwlmLiteral(s, nocase=False, id=len(lits)))
    return lits


def snort_like_literals(n=10000, seed=4):
    """Config 3: 'snort-like' set. Lengths {3:2%, 4:8%, 5-8:40%, 9-16:35%, 17-32:15%},
    30% caseless, mostly printable with 5% binary bytes. HWLM sees only the final
    8 bytes of a long literal (rose_build_matchers.cpp:717-724); the full strings
    are returned as well for the host-side long-literal check."""
    rng = np.random.default_rng(seed)
    word_chars = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789/._-=%&?: ",
                               dtype=np.uint8)
    lits, full, seen = [], [], set()
    while len(lits) < n:
        r = rng.random()
        if r < 0.02:
            ln = 3
        elif r < 0.10:
            ln = 4
        elif r < 0.50:
            ln = int(rng.integers(5, 9))
        elif r < 0.85:
            ln = int(rng.integers(9, 17))
        else:
            ln = int(rng.integers(17, 33))
        b = rng.choice(word_chars, ln)
        binary = rng.random(ln) < 0.05
        b = np.where(binary, rng.integers(0, 256, ln), b).astype(np.uint8)
        s = bytes(b)
        nocase = bool(rng.random() < 0.30)
        tail = s[-8:]
        key = (tail.upper() if nocase else tail, nocase)
        if key in seen:
            continue
        seen.add(key)
        lits.append(HwlmLiteral(tail, nocase=nocase, id=len(lits)))
        full.append(s)
    return lits, full


def _dictionary(rng, nwords=4096):
    lens = rng.integers(2, 12, nwords)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    p = np.arange(len(letters), 0, -1, dtype=np.float64)
    p /= p.sum()
    words = [bytes(rng.choice(letters, int(n), p=p)) for n in lens]
    words[:16] = [b"GET", b"POST", b"HTTP/1.1", b"Host:", b"User-Agent:", b"Accept:", b"Content-Length:",
                  b"Cookie:", b"200", b"OK", b"text/html", b"keep-alive", b"gzip", b"Mozilla/5.0", b"index.html",
                  b"charset=utf-8"]
    return words


def packet_corpus(total_bytes, lits, seed=3, match_every=16384, text_frac=0.70):
    """Config 2/3 corpus: packets with the length mix above, scaled to total_bytes;
    70% of the bytes are HTTP-like text built from a 4096-word dictionary, 30% are
    uniform random bytes (whole packets of either kind); every literal planted so
    that there is about one planted occurrence per match_every bytes.
    Returns (corpus uint8[total], off uint64[nblocks+1])."""
    rng = np.random.default_rng(seed)
    mean = float((PACKET_LENS * PACKET_PROB).sum())
    nblocks = max(1, int(total_bytes / mean))
    lens = rng.choice(PACKET_LENS, nblocks, p=PACKET_PROB).astype(np.int64)
    # scale to the exact total: trim / extend the tail
    csum = np.cumsum(lens)
    k = int(np.searchsorted(csum, total_bytes))
    if k >= nblocks:
        extra = []
        rem = total_bytes - int(csum[-1])
        while rem > 0:
            ln = int(min(rem, rng.choice(PACKET_LENS, p=PACKET_PROB)))
            extra.append(ln)
            rem -= ln
        lens = np.concatenate([lens, np.asarray(extra, dtype=np.int64)])
    else:
        lens = lens[: k + 1].copy()
        lens[k] -= int(csum[k]) - total_bytes
        if lens[k] == 0:
            lens = lens[:k]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert int(off[-1]) == total_bytes
    nblocks = lens.size

    # text stream: an 8 MiB pool of dictionary words joined by separators, then the
    # corpus is stitched from random 64 KiB windows of the pool (memcpy speed)
    words = _dictionary(rng)
    pool_tokens = (8 << 20) // 7
    ids = rng.integers(0, len(words), pool_tokens)
    seps = [b" ", b" ", b" ", b" ", b" ", b"\r\n", b"/", b"=", b"&", b"; ", b": "]
    sep_ids = rng.integers(0, len(seps), pool_tokens)
    pool = np.frombuffer(b"".join(words[i] + seps[j] for i, j in zip(ids.tolist(), sep_ids.tolist())),
                         dtype=np.uint8)
    win = 64 << 10
    nwin = (total_bytes + win - 1) // win
    starts = rng.integers(0, pool.size - win, nwin)
    corpus = np.empty(nwin * win, dtype=np.uint8)
    for i, st in enumerate(starts.tolist()):
        corpus[i * win:(i + 1) * win] = pool[st:st + win]
    corpus = corpus[:total_bytes]

    binary_block = rng.random(nblocks) >= text_frac
    is_bin = np.repeat(binary_block, lens)
    nb = int(is_bin.sum())
    if nb:
        corpus[is_bin] = rng.integers(0, 256, nb, dtype=np.uint8)
    del is_bin

    nplant = max(1, total_bytes // match_every)
    which = rng.integers(0, len(lits), nplant)
    pos = rng.integers(0, max(1, total_bytes - 40), nplant)
    for w, p in zip(which.tolist(), pos.tolist()):
        s = lits[w].s
        corpus[p:p + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return corpus, off


def line_corpus(total_bytes, seed=5, lo=40, hi=200):
    """Config 4 corpus: newline-terminated text lines of 40-200 bytes, one block
    per line (tools/hsbench/scripts/linebasedCorpus.py:29-35)."""
    rng = np.random.default_rng(seed)
    n = int(total_bytes / ((lo + hi) / 2) * 1.1) + 64
    lens = rng.integers(lo, hi + 1, n).astype(np.int64)
    csum = np.cumsum(lens)
    assert int(csum[-1]) >= total_bytes
    k = int(np.searchsorted(csum, total_bytes))
    lens = lens[: k + 1].copy()
    lens[k] -= int(csum[k]) - total_bytes
    if lens[k] == 0:
        lens = lens[:k]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    chars = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz     ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789,.;:-_()[]\t",
                          dtype=np.uint8)
    # random bytes through a 256-entry table of the alphabet (repeated to 256 entries: the first 25 characters are 4/3 as
    # likely as the rest), 64 MiB at a time: bounded 64-bit indices for a whole GiB were an 8 GiB temporary and a minute
    table = np.resize(chars, 256)
    corpus = np.empty(total_bytes, dtype=np.uint8)
    step = 64 << 20
    for lo_ in range(0, total_bytes, step):
        n_ = min(step, total_bytes - lo_)
        corpus[lo_:lo_ + n_] = table[np.frombuffer(rng.bytes(n_), dtype=np.uint8)]
    corpus[off[1:].astype(np.int64) - 1] = 0x0A
    return corpus, off
