#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c4
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_class_seq.py tests/test_gpu_hs_api.py tests/test_gpu_class_scan.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --also class256 --class-gib 1 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest.log; tail -5 $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print(json.dumps(d['also'],indent=1)[:3000])"
