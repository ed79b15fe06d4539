#!/bin/bash
# round 3, GPU call 1: parity of the gate pre-pass + confirm-stage A/B (old library vs pre-pass variants)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c1
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
for v in _old "" _u4 _s2 _s2u4; do
  for w in fdr10k; do
    echo "variant '$v' $(HSGPU_LIB_VARIANT=$v timeout 300 python tools/kbench.py $w 2>&1 | tail -1)" >> $OUT/kbench.log
  done
done
echo "variant '' $(timeout 300 python tools/kbench.py teddy64 2>&1 | tail -1)" >> $OUT/kbench.log
echo "variant '_old' $(HSGPU_LIB_VARIANT=_old timeout 300 python tools/kbench.py teddy64 2>&1 | tail -1)" >> $OUT/kbench.log
cat $OUT/pytest.log $OUT/kbench.log
