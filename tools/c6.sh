#!/bin/bash
# GPU call 6: suite, bench workloads per kernel, PMC counters of the teddy64 filter kernel against the harness
O=$PWD/gpurun_out/c6; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
HSGPU_MODE=fused timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pair or delivery or forced or dense" > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$?"; tail -2 $O/pytest_fused.log
for w in teddy64 fdr10k; do
  timeout 300 python tools/kbench.py $w --iters 12 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-250
done | tee $O/kbench.txt
cd /tmp && export TMPDIR=/tmp
for w in teddy64 fdr10k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $R/tools/kbench.py $w --iters 12 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rocclr" in r["Name"] or "at::" in r["Name"]: continue
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done | tee $O/kernel_times.txt
cd $R
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1))
  (cd /tmp; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_t$i -- python $R/tools/kbench.py teddy64 --iters 4 > $O/pmc_t$i.log 2>&1)
  (cd /tmp; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_h$i -- $R/tools/ubench/stream_exp one > $O/pmc_h$i.log 2>&1)
done
python - <<PY | tee $O/pmc_summary.txt
import csv, glob, collections
out="$O"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "hwlm_filter" in k or "exp_kernel" in k:
            agg[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("==", k, "n=", len(next(iter(v.values()))))
    print("   ", {c:round(sum(x)/len(x),1) for c,x in sorted(v.items())})
PY
