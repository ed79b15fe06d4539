import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
print("shapes", a.shape, b.shape, a.dtype)
ka = a[:, 0].astype(np.uint64) << np.uint64(32) | a[:, 1].astype(np.uint64)
kb = b[:, 0].astype(np.uint64) << np.uint64(32) | b[:, 1].astype(np.uint64)
# per block counts
for blkno in np.unique(a[:, 0]):
    ea = a[a[:, 0] == blkno][:, 1]; eb = b[b[:, 0] == blkno][:, 1]
    print("block", blkno, "head", len(ea), "new", len(eb))
    if len(ea) != len(eb):
        missing = np.setdiff1d(ea, eb)
        print("  missing ends: n", len(missing), "first", missing[:8], "last", missing[-8:])
        d = np.diff(missing); brk = np.nonzero(d > 1)[0]
        runs = np.split(missing, brk + 1)
        print("  runs:", len(runs), [(int(r[0]), int(r[-1]), len(r)) for r in runs[:12]])
        break
