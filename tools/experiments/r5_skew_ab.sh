#!/bin/bash
# tools/r5_skew_ab.sh -- on the GPU box: the confirm kernel's mirrored-rank split of the shares (default) against equal halves
# (HSGPU_MODE=no_skew), alternating on one box: bench.py's headline (records checked against the fused pipeline's) and teddy64.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_skew; mkdir -p $OUT
run() { # name mode
  HSGPU_MODE=$2 timeout 300 python $R/bench.py --steps 30 --warmup 5 --no-cpu --no-also --details $OUT/d_$1.json > $OUT/b_$1.json 2> $OUT/b_$1.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/b_$1.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("%-10s step %.4f ms  filter %.4f  confirm stage %.4f  pipeline %.4f  matches %d  parity: %s" % ("$1", d["ms_per_step"], r["kernel_ms_avg"], r["confirm_stage_ms_avg"], r["pipeline_ms_avg"], d["matches_per_step"], d.get("parity",{}).get("whole_corpus","")[:50]))
except Exception as e:
    print("$1 FAILED", e); print(open("$OUT/b_$1.err").read()[-600:])
PY
}
for i in 1 2 3; do run skew_$i ""; run equal_$i no_skew; done
for i in 1 2; do for m in "" no_skew; do echo "teddy64 [$m] $(HSGPU_MODE=$m python $R/tools/kbench.py teddy64 --iters 30 2>&1 | grep -o 'kernel avg [0-9.]* ms.*confirm [0-9.]* ms; matches [0-9]*')"; done; done
