#!/bin/bash
# tools/small_trace.sh [bytes] -- on the GPU box: kernel trace of resident small scans: per-kernel durations and the gaps between them
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-1048576}
OUT=$R/gpurun_out/small_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 3 4; do
python $R/tools/small_scan_loop.py $B 500 $mode 2>&1 | grep -v amdgpu.ids
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/m$mode -- python $R/tools/small_scan_loop.py $B 300 $mode > $OUT/m$mode.log 2>&1
python - <<PY
import csv, glob, collections
f=sorted(glob.glob("$OUT/m$mode/**/*kernel_trace.csv", recursive=True))[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-600:]  # steady state
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
prev=None
for r in rows:
    import re
    m=re.search(r"(hwlm_\w+|record_\w+|block_hint\w+|__amd\w+)", r["Kernel_Name"]); n=m.group(1) if m else r["Kernel_Name"][:40]
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    dur[n].append(e-s)
    if prev: gap[(prev[0],n)].append(s-prev[1])
    prev=(n,e)
print("mode $mode")
for n,v in dur.items(): print("  %-62s n %4d avg %.2f us" % (n,len(v),sum(v)/len(v)/1e3))
for k,v in gap.items(): print("  gap %-28s -> %-28s avg %.2f us" % (k[0][-28:],k[1][-28:],sum(v)/len(v)/1e3))
PY
done
