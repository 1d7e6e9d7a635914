#!/bin/bash
# one iteration of the -A0 site-code work: parity of the -A0 engines, lone-wave latency, bulk throughput
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/a0_iter
OUT=gpurun_out/a0_iter
if [ "${TESTS:-1}" = 1 ]; then
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_scalar_udh.py tests/test_gpu_a0_pipeline.py tests/test_gpu_noll3.py tests/test_gpu_fullsize_ref.py tests/test_gpu_cip.py tests/test_gpu_seeded.py 2>&1 | tail -8 > $OUT/tests.txt
cat $OUT/tests.txt
fi
timeout 300 python tools/dbg/narrow_probe.py 2>&1 | tail -8 | tee $OUT/narrow.txt
timeout 300 python tools/dbg/narrow_probe.py nosites 2>&1 | tail -8 | tee $OUT/narrow_nosites.txt
timeout 600 python bench.py --engines a0 --queries 1000 --steps 2 --warmup 1 --legs none --seeded-pairs 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('a0 bulk', d['value'], 'udh', c.get('udh_gcups'), c.get('udh_ms'), 'fwd', c.get('fwd_gcups'), c.get('fwd_ms'), d['roofline'].get('kernel_ms'))" | tee $OUT/bulk.txt
