#!/bin/bash
# tools/ab_bloom.sh -- on the GPU box: the headline workload with the key gate (default) and with the opt-in Bloom gate
# (HSGPU_BUILD_FLAGS=8192 = HSGPU_BUILD_FORCE_BLOOM), alternating, same box; prints step / filter / confirm-stage times per run.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_bloom; mkdir -p $OUT
for i in 1 2 3; do
  for f in 0 8192; do
    HSGPU_BUILD_FLAGS=$f python $R/bench.py --steps 30 --warmup 5 --no-cpu --no-also --details $OUT/d_${f}_$i.json > $OUT/b_${f}_$i.json 2> $OUT/b_${f}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$OUT/b_${f}_$i.json") if l.startswith("{")][-1])
r=d["roofline"]
print("flags=%5d run $i: step %.4f ms  filter %.4f  confirm stage %.4f  pipeline %.4f  matches %d  parity: %s" % ($f, d["ms_per_step"], r["kernel_ms_avg"], r["confirm_stage_ms_avg"], r["pipeline_ms_avg"], d["matches_per_step"], d.get("parity",{}).get("whole_corpus","")[:60]))
PY
  done
done
