#!/bin/bash
# tools/pmc_filter.sh <workload> -- PMC passes for the filter / confirm kernels of one workload
# (run on the GPU box via gpurun). Prints per-kernel counter averages and the implied shader clock.
set -u
WL=${1:-teddy64}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/kbench.py $WL --iters 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
out="$OUT"
dur={}
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "hwlm" in r["Name"]:
            print(f'{r["Name"][:110]:110s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:10.1f}')
            dur[r["Name"].split("(Hsgpu")[0]]=float(r["AverageNs"])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "hwlm" in k:
            agg[k.split("(Hsgpu")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in sorted(v.items())}
    print("==", k, "n=", len(next(iter(v.values()))))
    print("   ", {c:round(x,1) for c,x in m.items()})
    if k in dur and "GRBM_GUI_ACTIVE" in m:
        print("    implied clock GHz (GRBM_GUI_ACTIVE / trace duration):", round(m["GRBM_GUI_ACTIVE"]/dur[k],3))
PY
