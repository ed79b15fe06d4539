#!/bin/bash
# tools/profile.sh <tag> <workload> -- run on the GPU box (via gpurun): kernel trace
# + stats and separate PMC passes; per-kernel averages printed, CSVs in gpurun_out/<tag>/.
set -u
TAG=${1:-prof}; WL=${2:-teddy64}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/kbench.py $WL --iters 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
out="$OUT"
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:96]:96s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:10.1f}')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "hwlm" in k or "record_" in k or "block_hint" in k:
            short=k.split("(")[0].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("==", k)
    print("   ", {c:round(sum(x)/len(x),1) for c,x in sorted(v.items())})
PY
