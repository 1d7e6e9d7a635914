#!/bin/bash
# A/B of two builds of the library on the -A0 probes: tools/dbg/a0_ab.sh libspdp_hip_A.so libspdp_hip_B.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/a0_iter
for L in "$@"; do
  echo "=== $L"
  export SPDP_LIB=$PWD/spaln_amd/$L
  if [ "${TESTS:-0}" = 1 ]; then timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_scalar_udh.py tests/test_gpu_a0_pipeline.py tests/test_gpu_noll3.py tests/test_gpu_fullsize_ref.py tests/test_gpu_cip.py tests/test_gpu_seeded.py 2>&1 | tail -3; fi
  timeout 300 python tools/dbg/narrow_probe.py 2>&1 | grep "rows 30" 
  timeout 600 python bench.py --engines a0 --queries 1000 --steps 2 --warmup 1 --legs none --seeded-pairs 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('a0 bulk', d['value'], 'udh', c.get('udh_gcups'), c.get('udh_ms'), 'fwd', c.get('fwd_gcups'), c.get('fwd_ms'))"
done 2>&1 | tee gpurun_out/a0_iter/ab.txt
