#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c7
mkdir -p $OUT
cd $R
( timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --also flood 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest.log; tail -8 $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));
print(json.dumps(d['also'])[:2500])"
