#!/bin/bash
# GPU call 4 of round 2: suite (ordered output, reference parity, 10k set), then the filter kernel under
# its tuning variants (prefetch depth, nt loads, workgroup size) on both bench workloads
O=gpurun_out/c4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for v in "" _s4 _s6 _nt _s4nt _s6nt; do
 for wg in 512 1024; do
  for spec in "teddy64 0" "fdr10k 2048" "fdr10k 0"; do
    set -- $spec
    echo "== variant='$v' wg=$wg $1 flags=$2"
    HSGPU_LIB_VARIANT=$v HSGPU_WG_THREADS=$wg HSGPU_BUILD_FLAGS=$2 HSGPU_CAND_DIV=24 timeout 300 python tools/kbench.py $1 --iters 12 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330
  done
 done
done | tee $O/kbench.txt
