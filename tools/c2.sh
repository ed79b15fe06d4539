#!/bin/bash
# GPU call 2 of round 2: suite on the reworked kernels, classic (batched LDS reads) against the pair filter
O=gpurun_out/c2; mkdir -p $O
lscpu > $O/lscpu.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for spec in "fdr10k 0 16" "fdr10k 2048 64" "teddy64 0 64" "lits1000 0 64" "lits1000 1024 64" "lits4000 2048 64" "lits4000 0 64"; do
  set -- $spec
  echo "== $1 flags=$2 cand_div=$3"
  HSGPU_BUILD_FLAGS=$2 HSGPU_CAND_DIV=$3 timeout 300 python tools/kbench.py $1 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/kbench.txt
