#!/bin/bash
O=$PWD/gpurun_out/c11; mkdir -p $O; R=$PWD
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q --timeout 100 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for w in teddy64 fdr10k; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $R/tools/kbench.py $w --iters 12 > $O/trace_$w.log 2>&1
  grep "kernel avg" $O/trace_$w.log | cut -c1-200
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rocclr" in r["Name"] or "at::" in r["Name"]: continue
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done | tee $O/kernel_times.txt
