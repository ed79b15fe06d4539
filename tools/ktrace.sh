#!/bin/bash
# tools/ktrace.sh <tag> <workload> -- per-kernel durations (rocprofv3 --kernel-trace --stats)
TAG=${1:-kt}; WL=${2:-teddy64}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/kbench.py $WL --iters 10 > $OUT/trace.log 2>&1
python - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:10.1f} min_us={float(r["MinNs"])/1e3:10.1f}')
PY
