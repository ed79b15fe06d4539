#!/bin/bash
# tools/round_first_call.sh <round-tag> -- ONE gpurun call that settles everything the previous
# round left unmeasured, cheapest and most decisive first (each step has its own timeout; logs
# and summaries under gpurun_out/<tag>/):
#   1. the whole GPU suite (the late-additions file included)
#   2. the same suite with the device-side record sort (HSGPU_DEVICE_SORT=1) -> can it be default?
#   3. bench.py (N=1) and the rocprof / PMC summaries for profiles/
#   4. config 4 / config 5 lines (class8, rose1000) with the facade's timing breakdown, host
#      sort vs device sort
# usage: gpurun --timeout 2400 -- 'bash tools/round_first_call.sh r02'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu: $?" | tee $OUT/status.txt
HSGPU_DEVICE_SORT=1 timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_device_sort.log 2>&1; echo "pytest gpu (device sort): $?" | tee -a $OUT/status.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench: $?" | tee -a $OUT/status.txt
timeout 900 bash tools/round_profile.sh $TAG/prof > $OUT/round_profile.log 2>&1; echo "profile: $?" | tee -a $OUT/status.txt
for sort in 0 1; do
  HSGPU_DEVICE_SORT=$sort HSGPU_FACADE_TIMING=1 timeout 300 python tools/bench_extra.py rose1000 --gib 1 --iters 5 \
      > $OUT/rose1000_sort$sort.json 2> $OUT/rose1000_sort$sort.err
done
timeout 300 python tools/bench_extra.py class8 --gib 1 --iters 5 > $OUT/class8.json 2> $OUT/class8.err
tail -3 $OUT/pytest_gpu.log $OUT/pytest_gpu_device_sort.log; cat $OUT/bench.json $OUT/rose1000_sort*.json $OUT/class8.json
