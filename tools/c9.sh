#!/bin/bash
O=$PWD/gpurun_out/c9; mkdir -p $O; R=$PWD
for wg in 512 1024; do for w in teddy64 fdr10k; do
  echo "wg=$wg"; HSGPU_WG_THREADS=$wg timeout 120 python tools/kbench.py $w --iters 12 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done; done | tee $O/kbench.txt
timeout 500 python -m pytest tests -m gpu -x -q --timeout 100 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for w in teddy64 fdr10k; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $R/tools/kbench.py $w --iters 12 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rocclr" in r["Name"] or "at::" in r["Name"]: continue
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done | tee $O/kernel_times.txt
