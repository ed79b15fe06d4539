#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 200 python tools/seq_prof.py 1024 2>&1 | grep seq_prof
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --also flood,fdr10k_8g,teddy64 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'P'
import json
d=json.load(open('/root/repo/gpurun_out/bench_details.json'))
for k,v in d['also'].items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','stages_ms','error','parity_whole_corpus')}, v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('kernel_ms_avg'))
h=d['headline']; print('headline', h['value'], h['ms_per_step'], h['roofline']['frac'], h['roofline']['kernel_ms_avg'], h['roofline'].get('pipeline_ms_avg'), h.get('multi_gpu'))
P
wc -c $O/bench.json
