#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c21
mkdir -p $OUT
cd $R
( timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
for f in 0 2048; do echo "build_flags=$f $(HSGPU_BUILD_FLAGS=$f timeout 300 python tools/kbench.py fdr10k 2>&1 | tail -1 | cut -c1-250)" >> $OUT/kbench.log; done
cat $OUT/pytest.log $OUT/kbench.log
