#!/bin/bash
# round 5, closing GPU call: counters of the kernels that changed since the profile pass, the whole gpu suite, the bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cp profiles/r05_counters.json gpurun_out/r05_counters.json
WORKLOADS="${WORKLOADS:-c3_a0 blk}" WAVE="" bash tools/profile_r05.sh > gpurun_out/r05_final_prof.log 2>&1
tail -4 gpurun_out/r05_final_prof.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gputest.log
tail -4 gpurun_out/r05_gputest.log
cp gpurun_out/r05_counters.json profiles/r05_counters.json
timeout 1800 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
tail -c 400 gpurun_out/r05_bench_final.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_final.json') if l.startswith('{')][-1])
print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu'))
for k,v in d['config'].items():
    if isinstance(v,dict) and ('value' in v or 'error' in v or 'runs' in v or 'identical_exon_tables' in v):
        print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','kernel_ms','valu_frac','hbm_frac','error','udh_gcups','fwd_gcups','sweep_gcups','profile_stale','identical_exon_tables','different','queries','reference_wall_s','library_s','wall_s')})
        if 'runs' in v:
            for r in v['runs']: print('    ', r['mode'], 'ref', r['reference'].get('wall_s'), 'gpu', r['gpu'].get('wall_s'), 'identical', r.get('identical'), 'ratio', r.get('gpu_over_reference_wall'))
P
