#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c14
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_zz_gpu_late_additions.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
for w in fdr10k teddy64; do echo "$(timeout 300 python tools/kbench.py $w 2>&1 | tail -1 | cut -c1-220)" >> $OUT/kbench.log; done
( python bench.py --steps 20 --warmup 3 --no-cpu --no-also 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'value', d['value'], d.get('two_scans_in_flight'))" ) >> $OUT/kbench.log 2>&1
cat $OUT/pytest.log $OUT/kbench.log
