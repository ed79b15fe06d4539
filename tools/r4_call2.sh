#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
for v in "" _bal; do
  HSGPU_LIB_VARIANT=$v timeout 300 python tools/ab_tail.py fdr10k --wg-stamps --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
  HSGPU_LIB_VARIANT=$v timeout 300 python tools/ab_tail.py teddy64 --wg-stamps --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
done
for v in "" _bal; do
HSGPU_LIB_VARIANT=$v timeout 300 python tools/ab_tail.py fdr10k --gib 8 --iters 10 --modes folded,unfolded 2>&1 | grep -v "^\[\|amdgpu.ids" | tee -a $O/ab.log
done
