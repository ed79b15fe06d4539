#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c24
mkdir -p $OUT
cd $R
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_zz_gpu_late_additions.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1
for w in fdr10k teddy64; do echo "$(timeout 300 python tools/kbench.py $w 2>&1 | tail -1 | cut -c1-200)" >> $OUT/kbench.log; done
( python bench.py --steps 20 --warmup 3 --no-cpu --also flood,teddy64 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'value', d['value']); print({k:(v.get('ms_per_step'), v.get('value'), v.get('error')) for k,v in d['also'].items()})" ) >> $OUT/kbench.log 2>&1
bash tools/r3_timeline.sh 2>/dev/null | tail -8 | head -5 >> $OUT/kbench.log
cat $OUT/pytest.log $OUT/kbench.log
