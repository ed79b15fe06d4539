#!/bin/bash
# tools/gpu_check.sh [round-tag] -- on the GPU box, ONE call = ONE box: what the box is (tools/box_info.sh), the whole GPU suite, smoke(),
# the rocprofv3 passes (tools/round_profile.sh <tag>), then the default bench line with its wall time -- reading traffic / kernel_ms_trace /
# issue bounds from THIS call's passes (HSGPU_PROFILE_DIR), so that every figure of the committed line comes from one box.
#   gpurun --timeout 3000 -- 'bash tools/gpu_check.sh r06'   ->  gpurun_out/<tag>check/{pytest.log,smoke.log,bench.json,bench.time}, gpurun_out/<tag>/
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}check; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/box_info.sh > gpurun_out/box_info.txt 2>&1; cp gpurun_out/box_info.txt $O/
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
bash tools/round_profile.sh $TAG > $O/profile.log 2>&1
( time HSGPU_PROFILE_DIR=$GRAFT_REPO_ROOT/gpurun_out/$TAG timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cp gpurun_out/bench_details.json $O/bench_details.json 2>/dev/null
