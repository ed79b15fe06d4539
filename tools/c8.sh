#!/bin/bash
O=$PWD/gpurun_out/c8; mkdir -p $O; R=$PWD
timeout 120 ./tools/ubench/stream_exp 2>&1 | grep "BLOCKED\|----" | grep -v "aux=2" | head -40 | tee $O/stream.txt
for w in teddy64 fdr10k; do
  timeout 120 python tools/kbench.py $w --iters 12 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-250
done | tee $O/kbench.txt
cd /tmp && export TMPDIR=/tmp
for w in fdr10k; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $R/tools/kbench.py $w --iters 12 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rocclr" in r["Name"] or "at::" in r["Name"]: continue
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done | tee $O/kernel_times.txt
cd $R; timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q --timeout 100 2>&1 | tail -2
