#!/bin/bash
O=$PWD/gpurun_out/c10; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q --timeout 100 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err; head -c 1500 $O/bench.json
timeout 900 bash tools/round_profile.sh r02 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile.log
