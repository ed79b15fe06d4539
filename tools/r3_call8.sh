#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c8
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_late_additions.py -x -q -m gpu -k "alternative_pipelines or other_geometries or overflow or fused" 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --also flood 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest.log; tail -4 $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));
f=d['also']['flood']; print({k:f[k] for k in f if k not in ('workload','table','cpu_baseline','parity')})"
