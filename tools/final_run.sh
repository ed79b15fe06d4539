#!/bin/bash
# tools/final_run.sh <tag> -- the round's closing GPU call: `pytest -m gpu`, smoke(), the driver's bench command (full line with the
# CPU leg), then tools/round_profile.sh; everything under gpurun_out/<tag>_final/ and gpurun_out/<tag>/
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
F=gpurun_out/${TAG}_final
mkdir -p $F
(timeout 240 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $F/tests.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $F/smoke.log
( time timeout 300 python bench.py > $F/bench.json 2> $F/bench.err ) 2> $F/time.txt
bash tools/round_profile.sh $TAG > $F/profile.log 2>&1
cat $F/tests.log $F/smoke.log $F/time.txt
tail -c 1500 $F/bench.json
