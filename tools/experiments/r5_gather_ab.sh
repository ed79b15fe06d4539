#!/bin/bash
# tools/r5_gather_ab.sh -- on the GPU box: kernel trace of the headline step with library variants of the gather's group size
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/gather_ab; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "" $VARIANTS; do
  HSGPU_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$v -- python $R/tools/kbench.py fdr10k --iters 30 > $OUT/k$v.log 2>&1
  echo "variant [$v] $(grep -o 'kernel avg [0-9.]* ms' $OUT/k$v.log) | $(cat $(find $OUT/t$v -name '*kernel_stats.csv' | head -1) | grep -E 'record_sort|confirm' | awk -F, '{printf "%s avg %.1f us; ", substr($1,1,60), $4/1000}')"
done
