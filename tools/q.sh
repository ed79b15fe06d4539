#!/bin/bash
# tools/q.sh "<env assignments>" ... -- one bench line (fdr10k headline, no CPU leg, no also) per argument
for e in "$@"; do
  env $e timeout 200 python bench.py --no-cpu --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$e', '| ms/step', d['ms_per_step'], 'filter', r['kernel_ms_avg'], 'best', r['kernel_ms_best'], 'confirm-stage', r['confirm_stage_ms_avg'], 'pipe', r['pipeline_ms_avg'], 'flags', d['table']['flags'], 'matches', d['matches_per_step'])
"
done
