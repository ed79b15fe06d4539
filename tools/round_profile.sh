#!/bin/bash
# tools/round_profile.sh <round-tag> -- on the GPU box: the driver's bench command (headline workload, no CPU
# leg) under rocprofv3 (--kernel-trace --stats), then separate PMC passes of the same command: FETCH_SIZE,
# WRITE_SIZE, and two sets of SQ counters for the filter kernels; summaries are written under
# gpurun_out/<tag>/ for copying to profiles/.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (serial launches only, no sustained run: the averages are those of the K timed steps' kernels)
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-overlap-probe --sustain-seconds 0 --also teddy64,class256 --class-gib 1"
# 1. the headline workload alone (the driver's command without the other workloads and the CPU leg): kernel_stats.csv is
#    what roofline.achieved's launch duration has to agree with
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-also --no-overlap-probe --sustain-seconds 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
# 2. the same with teddy64 and class256 (1 GiB) behind it: their kernels are other instantiations (kernel_stats_also.csv);
#    flood has a trace of its own (tools/flood_prof.py), rose1000's GPU stage is teddy64's kernel, batch_sweep is 1 000 small launches
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_also -- $CMD > $OUT/bench_also_under_rocprof.json 2> $OUT/trace_also.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $CMD > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
python - <<PY
import csv, glob, collections, re, json
out="$OUT"
lines=[]
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    lines.append(open(f).read())
open(out+"/kernel_stats.csv","w").write("".join(lines))
open(out+"/kernel_stats_also.csv","w").write("".join(open(f).read() for f in sorted(glob.glob(out+"/trace_also/**/*kernel_stats.csv", recursive=True))))
def collect(dirs):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in sorted(glob.glob(out+"/"+d+"/**/*counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                m=re.search(r"(hwlm_\w+<[^>]*>|record_\w+|block_hint_kernel|class_\w+(?:<[^>]*>)?|pair_\w+|seq_\w+|run_accel_\w+)", r.get("Kernel_Name",""))
                if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
summ={k:{c:{"avg_KB":sum(x)/len(x),"n":len(x)} for c,x in v.items()} for k,v in collect(("pmc_fetch","pmc_write")).items()}
json.dump(summ, open(out+"/pmc_summary.json","w"), indent=1)
sq={k:{c:round(sum(x)/len(x),1) for c,x in sorted(v.items())} for k,v in collect(("pmc_sq1","pmc_sq2")).items() if "filter" in k or "confirm" in k or "class_" in k}
for k,v in sq.items():
    if v.get("SQ_LDS_IDX_ACTIVE"): v["lds_conflict_share_of_lds_cycles"]=round(v.get("SQ_LDS_BANK_CONFLICT",0)/v["SQ_LDS_IDX_ACTIVE"],3)
    if v.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_ANY","SQ_WAIT_ANY"):
            if c in v: v[c+"_share_of_wave_cycles"]=round(v[c]/v["SQ_WAVE_CYCLES"],3)
json.dump(sq, open(out+"/filter_sq.json","w"), indent=1)
# what size the passes ran at, on which box: bench.py shows traffic / kernel_ms_trace / issue bounds beside a run of THAT size only
meta={}
try:
    line=[l for l in open(out+"/bench_under_rocprof.json") if l.startswith("{")][-1]
    b=json.loads(line)
    meta={"corpus_bytes": b["roofline"]["algorithmic_bytes_per_launch"], "workload": b["config"]["workload"], "ms_per_step_under_rocprof": b["ms_per_step"]}
except Exception as e:
    meta={"error": str(e)}
try:
    meta["gpu_uuid"]=[l.split(":",1)[1].strip() for l in open("$R/gpurun_out/box_info.txt") if "Uuid" in l and "GPU-" in l][0]
except Exception:
    pass
json.dump(meta, open(out+"/profile_meta.json","w"), indent=1)
print(open(out+"/kernel_stats.csv").read()[:2500])
print(json.dumps(summ, indent=1)[:2500])
print(json.dumps(sq, indent=1)[:3000])
PY
