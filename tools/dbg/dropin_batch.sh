#!/bin/bash
# the -Q7 / -Q4 drop-in in batch mode (-t16 workers map, one library call aligns) against the reference
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/dropin_batch
for NQ in ${NQS:-2000 20000}; do
  echo "== queries $NQ"
  timeout 900 python tools/dropin_demo.py --queries $NQ --genes 200 --modes ${MODES:-Q7} --gpu-threads 16 2>gpurun_out/dropin_batch/err_$NQ.txt | tee gpurun_out/dropin_batch/out_$NQ.json | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']:
    print(r['mode'], 'ref', r['reference'].get('wall_s'), 'gpu', r['gpu'].get('wall_s', r['gpu']), 'identical', r.get('identical'), 'differing', r.get('records_differing'), 'ratio', r.get('gpu_over_reference_wall'))
    print('   ', r['gpu'].get('shim','')[:900])
    if r.get('first_difference'): print(r['first_difference'])"
  tail -3 gpurun_out/dropin_batch/err_$NQ.txt
done 2>&1 | tee gpurun_out/dropin_batch/out.txt
