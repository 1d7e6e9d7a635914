#!/bin/bash
# the -Q7 drop-in with several library calls side by side (SPALN_GPU_CONTEXTS) and different thread counts
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/dropin_ctx
for cfg in "1 2000 2000" "4 4000 1000" "8 4000 500" "4 2000 500"; do
  set -- $cfg
  echo "== contexts $1 threads $2 batch $3"
  SPALN_GPU_CONTEXTS=$1 SPALN_GPU_BATCH=$3 timeout 600 python tools/dropin_demo.py --queries ${NQ:-20000} --genes 200 --modes Q7 --gpu-threads $2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']:
    print(r['mode'], 'ref', r['reference'].get('wall_s'), 'gpu', r['gpu'].get('wall_s', r['gpu']), 'identical', r.get('identical'), 'ratio', r.get('gpu_over_reference_wall'))
    print('   ', r['gpu'].get('shim','')[:600])"
done 2>&1 | tee gpurun_out/dropin_ctx/out.txt
