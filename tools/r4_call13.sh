#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_exchange.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
bash tools/seq_pmc.sh r4m_seq 256 2>&1 | tail -45
