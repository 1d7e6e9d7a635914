#!/bin/bash
# round 5, first GPU call: the gpu tests, the default bench line with its legs, counters of c2 / a0
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gputest.log
tail -3 gpurun_out/r05_gputest.log
timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/r05_bench_first.json 2> gpurun_out/r05_bench_first.err
tail -c 600 gpurun_out/r05_bench_first.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_first.json') if l.startswith('{')][-1])
print('C2', d['value'], d['ms_per_step'])
for k,v in d['config'].items():
    if isinstance(v,dict) and ('value' in v or 'error' in v or 'runs' in v):
        print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','kernel_ms','valu_frac','hbm_frac','error','udh_gcups','fwd_gcups','sweep_gcups','runs','wall_s')})
P
