#!/bin/bash
# end of round 4: the whole GPU suite, smoke, the default bench line, then the round profile
O=$GRAFT_REPO_ROOT/gpurun_out/r4final; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cp gpurun_out/bench_details.json $O/bench_details.json 2>/dev/null
bash tools/round_profile.sh r04 > $O/profile.log 2>&1
