#!/bin/bash
# first GPU call of round 4: parity of the folded tail, then same-box A/Bs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
for v in "" _hl _nt; do
  HSGPU_LIB_VARIANT=$v timeout 300 python tools/ab_tail.py fdr10k --wg-stamps --modes folded,unfolded 2>&1 | grep -v "^\[" | tee -a $O/ab.log
done
HSGPU_LIB_VARIANT= timeout 300 python tools/ab_tail.py teddy64 --modes folded,unfolded 2>&1 | tee -a $O/ab.log
HSGPU_LIB_VARIANT=_hl timeout 300 python tools/ab_tail.py teddy64 --modes folded 2>&1 | tee -a $O/ab.log
timeout 300 python tools/ab_tail.py fdr10k --gib 8 --iters 10 --modes folded,unfolded 2>&1 | tee -a $O/ab.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-also > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
