#!/bin/bash
# tools/round_profile.sh <round-tag> -- on the GPU box: the driver's bench command under
# rocprofv3 (--kernel-trace --stats), then separate PMC passes (FETCH_SIZE / WRITE_SIZE) of
# the same command; summaries are written under gpurun_out/<tag>/ for copying to profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
python - <<PY
import csv, glob, collections, re, json
out="$OUT"
lines=[]
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    lines.append(open(f).read())
open(out+"/kernel_stats.csv","w").write("".join(lines))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch","pmc_write"):
    for f in sorted(glob.glob(out+"/"+d+"/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            m=re.search(r"(hwlm_\w+<[^>]*>|record_\w+|block_hint_kernel|control_reset_kernel)", r.get("Kernel_Name",""))
            if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ={k:{c:{"avg_KB":sum(x)/len(x),"n":len(x)} for c,x in v.items()} for k,v in agg.items()}
json.dump(summ, open(out+"/pmc_summary.json","w"), indent=1)
print(open(out+"/kernel_stats.csv").read()[:3000])
print(json.dumps(summ, indent=1)[:3000])
PY
