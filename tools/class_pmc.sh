#!/bin/bash
# tools/class_pmc.sh <tag> [class_prof args] -- SQ counters of class_tile_kernel (two passes)
TAG=${1:-clspmc}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/class_prof.py $*"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sq1 -- $CMD > /dev/null 2> $OUT/sq1.err
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
timeout 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $OUT/mem -- $CMD > /dev/null 2> $OUT/mem.err
python - <<PY
import csv, glob, collections, json
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1","sq2","mem"):
    for f in sorted(glob.glob(out+"/"+d+"/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","")
            if "class_" in k or "pair_" in k: agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{c:round(sum(x)/len(x),1) for c,x in sorted(v.items())} for k,v in agg.items()}
json.dump(res, open(out+"/class_sq.json","w"), indent=1)
print(json.dumps(res, indent=1))
PY
