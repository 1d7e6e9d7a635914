#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu -k "seeded or shim or dropin or noll3" 2>&1 | tail -3
for m in 0 1 2; do
SPDP_SEED_SCOUT=$m timeout 900 python tools/dropin_demo.py --protein --queries 10000 --genes 200 --modes Q7 --gpu-threads 16 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
for r in d['runs']: print('protein 10000 scout $m', r['mode'], 'ref', r['reference']['wall_s'], 'gpu', r['gpu']['wall_s'], 'identical', r['identical'], r.get('records_differing'), 'ratio', r.get('gpu_over_reference_wall'))"
done
