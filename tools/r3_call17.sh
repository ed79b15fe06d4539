#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c17
mkdir -p $OUT
cd $R
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_zz_gpu_late_additions.py tests/test_gpu_round3.py tests/test_gpu_hs_api.py -x -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
( python bench.py --steps 20 --warmup 3 --no-cpu --also flood,teddy64 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'value', d['value'], d.get('two_scans_in_flight')); print({k:(v.get('ms_per_step'), v.get('value'), v.get('error')) for k,v in d['also'].items()}); print(d['also']['flood'].get('stages_ms'), d['also']['flood'].get('candidate_overflow_scans'))" ) >> $OUT/kbench.log 2>&1
cat $OUT/pytest.log $OUT/kbench.log
