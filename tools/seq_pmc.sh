#!/bin/bash
# tools/seq_pmc.sh <tag> [mib] -- SQ counters of the class-sequence kernel (two passes) -> gpurun_out/<tag>/seq_sq.json
TAG=${1:-seqpmc}; MIB=${2:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/seq_prof.py $MIB"
$CMD 2>&1 | grep seq_prof | tee $OUT/seq_prof.txt
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/sq1 -- $CMD > /dev/null 2> $OUT/sq1.err
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
python - <<PY
import csv, glob, collections, json
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1","sq2"):
    for f in sorted(glob.glob(out+"/"+d+"/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","")
            if "class_seq" in k: agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{c:round(sum(x)/len(x),1) for c,x in sorted(v.items())} for k,v in agg.items()}
for k,v in res.items():
    if v.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY"):
            if c in v: v[c+"_share_of_wave_cycles"]=round(v[c]/v["SQ_WAVE_CYCLES"],3)
json.dump(res, open(out+"/seq_sq.json","w"), indent=1)
print(json.dumps(res, indent=1))
PY
