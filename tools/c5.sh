#!/bin/bash
# GPU call 5 of round 2: suite (ordered output, pair filter, reference parity), per-kernel times of both bench workloads
O=$PWD/gpurun_out/c5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for w in teddy64 fdr10k; do
  timeout 300 python tools/kbench.py $w --iters 12 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-250
done | tee $O/kbench.txt
./tools/ubench/stream_exp 2>&1 | head -18 | tee $O/stream_exp.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in teddy64 fdr10k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $R/tools/kbench.py $w --iters 12 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done | tee $O/kernel_times.txt
