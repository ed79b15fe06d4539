#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_class_seq.py tests/test_gpu_class_scan.py tests/test_gpu_exchange.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
bash tools/seq_pmc.sh r4n_seq 256 2>&1 | tail -32
