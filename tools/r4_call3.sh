#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
HSGPU_LIB_VARIANT= timeout 300 python tools/ab_tail.py fdr10k --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
HSGPU_LIB_VARIANT=_ns timeout 300 python tools/ab_tail.py fdr10k --modes folded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
HSGPU_LIB_VARIANT= timeout 300 python tools/ab_tail.py fdr10k --gib 8 --iters 10 --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
HSGPU_LIB_VARIANT=_ns timeout 300 python tools/ab_tail.py fdr10k --gib 8 --iters 10 --modes folded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
