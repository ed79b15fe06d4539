#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c23
mkdir -p $OUT
cd $R
for v in "" _gate; do echo "variant '$v' $(HSGPU_LIB_VARIANT=$v timeout 300 python tools/kbench.py fdr10k 2>&1 | tail -1 | cut -c1-250)" >> $OUT/kbench.log; done
( HSGPU_LIB_VARIANT=_gate timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log $OUT/kbench.log
