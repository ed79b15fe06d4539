#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c6
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_hs_api.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py --steps 10 --warmup 2 --also rose1000 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest.log; tail -5 $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));
for k in ('value','ms_per_step','end_to_end_resident','host_buffers'): print(k, json.dumps(d[k]))
print(json.dumps(d['also'])[:1500])"
