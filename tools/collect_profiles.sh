#!/bin/bash
# tools/collect_profiles.sh [round-tag] -- HERE, after `gpurun -- 'bash tools/gpu_check.sh <tag>'`: the summaries of that one call
# from gpurun_out/ (scratch) into profiles/<tag>_* (tracked).
TAG=${1:-r06}
cd "$(dirname "$0")/.."
G=gpurun_out/$TAG; C=gpurun_out/${TAG}check; P=profiles
cp $C/bench.json $P/${TAG}_bench.json
cp $C/bench_details.json $P/${TAG}_bench_details.json
cp $C/box_info.txt $P/${TAG}_box_info.txt
cp $G/kernel_stats.csv $P/${TAG}_bench_kernel_stats.csv
cp $G/kernel_stats_also.csv $P/${TAG}_bench_kernel_stats_also.csv
cp $G/pmc_summary.json $P/${TAG}_bench_pmc_summary.json
cp $G/filter_sq.json $P/${TAG}_filter_sq.json
cp $G/profile_meta.json $P/${TAG}_bench_profile_meta.json
cp $G/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json
{ grep -E "passed|failed|error" $C/pytest.log | tail -3; tail -1 $C/pytest.log; tail -1 $C/smoke.log; cat $C/bench.time; } > $P/${TAG}_check_summary.txt
cat $P/${TAG}_check_summary.txt
