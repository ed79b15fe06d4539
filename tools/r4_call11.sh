#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
for v in "" _rr _plaindrain _nocopy _noplace; do
  HSGPU_LIB_VARIANT=$v timeout 300 python tools/ab_tail.py fdr10k --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --also class256,rose1000,batch_sweep --class-gib 1 --rose-gib 0.5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3500 $O/bench.json; grep -v "^bench details" $O/bench.err | tail -12
