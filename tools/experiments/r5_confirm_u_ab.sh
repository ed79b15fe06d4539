#!/bin/bash
# tools/r5_confirm_u_ab.sh -- on the GPU box: the headline workload with library variants of the confirm kernel
# (entries per lane and step U / wavefronts per SIMD W: csrc/Makefile VARIANT=_u4 ...), alternating with the product
# library on one box; and the co-scheduling probe (two scans in flight on two streams) with the filter at 16 and at 12
# wavefronts per workgroup (HSGPU_WG_THREADS=768: 3 x 120 registers per SIMD leave room for a confirm wavefront beside it).
# Prints step / filter / confirm-stage times per run; every run checks all records against the fused pipeline's.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_u; mkdir -p $OUT
run() { # name variant wg_threads extra-args
  local name=$1 var=$2 wg=$3; shift 3
  HSGPU_LIB_VARIANT=$var HSGPU_WG_THREADS=$wg timeout 300 python $R/bench.py --steps 30 --warmup 5 --no-cpu --no-also "$@" --details $OUT/d_$name.json > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/b_$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    two=d.get("two_scans_in_flight") or {}
    print("%-14s step %.4f ms  filter %.4f  confirm stage %.4f  pipeline %.4f  matches %d  two-in-flight %s  parity: %s" % ("$name", d["ms_per_step"], r["kernel_ms_avg"], r["confirm_stage_ms_avg"], r["pipeline_ms_avg"], d["matches_per_step"], two.get("ms_per_step"), d.get("parity",{}).get("whole_corpus","")[:50]))
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/b_$name.err").read()[-600:])
PY
}
for i in 1 2; do
  run base_$i "" ""
  for v in ${VARIANTS:-_u4 _u3}; do run ${v}_$i $v ""; done
done



