#!/bin/bash
# tools/confirm_mem_pmc.sh -- memory-side counters (vector L1 <-> L2) of the fdr10k confirm and filter kernels, two PMC passes of
# the headline bench command -> gpurun_out/confirm_mem/confirm_mem.json (what the confirm stage waits for: profiles/r03_confirm_mem.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/confirm_mem
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-also"
timeout 100 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_READ_TAGCONFLICT_STALL_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > /dev/null 2> $OUT/p1.err
timeout 100 rocprofv3 --pmc TCC_EA0_RDREQ TCC_HIT TCC_MISS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > /dev/null 2> $OUT/p2.err
python - <<PY
import csv, glob, collections, json, re
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("p1","p2"):
    for f in sorted(glob.glob(out+"/"+d+"/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            m=re.search(r"(hwlm_\w+<[^>]*>|record_sort_kernel)", r.get("Kernel_Name",""))
            if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{c:round(sum(x)/len(x),1) for c,x in sorted(v.items())} for k,v in agg.items()}
for k,v in res.items():
    if v.get("TCP_TCC_READ_REQ"): v["avg_read_latency_cycles"]=round(v.get("TCP_TCC_READ_REQ_LATENCY",0)/v["TCP_TCC_READ_REQ"],1)
    if v.get("TCC_HIT") is not None and v.get("TCC_MISS") is not None and v["TCC_HIT"]+v["TCC_MISS"]: v["l2_hit_share"]=round(v["TCC_HIT"]/(v["TCC_HIT"]+v["TCC_MISS"]),3)
json.dump(res, open(out+"/confirm_mem.json","w"), indent=1)
print(json.dumps(res, indent=1)[:3500])
PY
