#!/bin/bash
# per-dispatch timeline of the fdr10k step (rocprofv3 --kernel-trace): kernel durations and the gaps between them
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/tools/kbench.py fdr10k --iters 12 > $OUT/trace.log 2>&1
python - <<PY
import csv, glob
rows=[]
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 3 steps: find filter kernels
idx=[i for i,r in enumerate(rows) if "hwlm_filter_kernel" in r["Kernel_Name"] and "true, false, false, false, true, false, true, false, false" in r["Kernel_Name"]]
out=[]
for i in idx[-4:-1]:
    t0=int(rows[i]["Start_Timestamp"])
    prev_end=t0
    for r in rows[i:i+5]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        name=r["Kernel_Name"].split("(")[0][:60]
        out.append(f"{name:60s} start+{(s-t0)/1e3:8.1f}us dur {(e-s)/1e3:8.1f}us gap_before {(s-prev_end)/1e3:6.1f}us wg={r.get('Workgroup_Size','?')} grid={r.get('Grid_Size','?')}")
        prev_end=e
    out.append("")
open("$OUT/timeline.txt","w").write("\n".join(out))
print("\n".join(out))
PY
tail -2 $OUT/trace.log
