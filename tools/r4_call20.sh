#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4fl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $GRAFT_REPO_ROOT/tools/flood_prof.py > $O/flood.log 2>&1
find $O/tr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/tr -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv
rm -rf $O/tr
