#!/bin/bash
# tools/r5_balance_ab.sh -- on the GPU box: the headline workload with the confirm kernel's balanced partition (default) and with
# equal parts of the shares (HSGPU_MODE=static_parts: rounds 4-5), alternating on one box; step / filter / confirm-stage per run.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_bal; mkdir -p $OUT
run() { # name mode variant extra-args
  local name=$1 mode=$2 var=$3; shift 3
  HSGPU_MODE=$mode HSGPU_LIB_VARIANT=$var timeout 300 python $R/bench.py --steps 30 --warmup 5 --no-cpu --no-also "$@" --details $OUT/d_$name.json > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/b_$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("%-16s step %.4f ms  filter %.4f  confirm stage %.4f  pipeline %.4f  matches %d  parity: %s" % ("$name", d["ms_per_step"], r["kernel_ms_avg"], r["confirm_stage_ms_avg"], r["pipeline_ms_avg"], d["matches_per_step"], d.get("parity",{}).get("whole_corpus","")[:50]))
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/b_$name.err").read()[-600:])
PY
}
for i in 1 2 3; do
  run balanced_$i "" ""
  run static_$i static_parts ""
  for v in $VARIANTS; do run bal${v}_$i "" $v; done
done
