#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4fl; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dense or flood or overflow or again or round4" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $GRAFT_REPO_ROOT/tools/flood_prof.py > $O/flood.log 2>&1
find $O/tr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/tr
