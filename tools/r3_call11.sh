#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c11
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_exchange.py tests/test_gpu_class_scan.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log
