#!/bin/bash
# round 3, GPU call 2: the key gate inside the filter's spill path -- parity + A/B against the round-2 library
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c2
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_zz_gpu_late_additions.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
for v in _old ""; do
  for w in fdr10k lits1000 teddy64; do
    echo "variant '$v' $(HSGPU_LIB_VARIANT=$v timeout 300 python tools/kbench.py $w 2>&1 | tail -1)" >> $OUT/kbench.log
  done
done
cat $OUT/pytest.log $OUT/kbench.log
