#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python tools/ab_tail.py fdr10k --modes folded,unfolded,folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
timeout 300 python tools/ab_tail.py teddy64 --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
timeout 300 python tools/ab_tail.py fdr10k --gib 8 --iters 10 --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --also class256,flood,teddy64 --class-gib 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json; grep -v "^bench details" $O/bench.err | tail -12
