#!/bin/bash
# tools/gpu_check.sh [round-tag] -- on the GPU box, what the driver does at the end of a round and what gets committed under profiles/:
# the whole GPU suite, smoke(), the default bench line (with its wall time), then tools/round_profile.sh <tag>.
#   gpurun --timeout 3000 -- 'bash tools/gpu_check.sh r04'   ->  gpurun_out/<tag>check/{pytest.log,smoke.log,bench.json,bench.time}, gpurun_out/<tag>/
TAG=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}check; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cp gpurun_out/bench_details.json $O/bench_details.json 2>/dev/null
bash tools/round_profile.sh $TAG > $O/profile.log 2>&1
