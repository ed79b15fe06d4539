#!/bin/bash
# tools/teddy_mem_pmc.sh -- memory-side counters of the teddy64 filter kernel (the kernel that is neither vector- nor
# LDS-bound): which of the TCP / TCC counters this rocprofv3 knows, then one pass with those -> gpurun_out/teddy_mem/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/teddy_mem
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || timeout 120 rocprofv3 -L > $OUT/avail.txt 2>&1
WANT="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_READ_TAGCONFLICT_STALL_CYCLES TCC_EA0_RDREQ TCC_EA_RDREQ TCC_EA0_RDREQ_32B TCC_EA_RDREQ_32B TCC_HIT TCC_MISS TCC_EA0_RD_LATENCY TCC_EA_RDREQ_LEVEL TCP_GATE_EN1 TCP_TA_TCP_STATE_READ"
HAVE=""
for c in $WANT; do grep -qw "$c" $OUT/avail.txt && HAVE="$HAVE $c"; done
echo "counters available:$HAVE" | tee $OUT/chosen.txt
set -- $HAVE
P1="${@:1:4}"; P2="${@:5:4}"
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu --workload teddy64 --no-also"
[ -n "$P1" ] && timeout 250 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > /dev/null 2> $OUT/p1.err
[ -n "$P2" ] && timeout 250 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > /dev/null 2> $OUT/p2.err
python - <<PY
import csv, glob, collections, json
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("p1","p2"):
    for f in sorted(glob.glob(out+"/"+d+"/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","")
            if "hwlm_filter_kernel" in k: agg[k[k.index("hwlm_filter_kernel"):][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{c:round(sum(x)/len(x),1) for c,x in sorted(v.items())} for k,v in agg.items()}
json.dump(res, open(out+"/teddy64_mem.json","w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
