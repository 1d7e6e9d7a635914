#!/bin/bash
cd /root/repo
for m in 0 1 2; do
SPDP_SEED_SCOUT=$m SPDP_SEED_VERBOSE=1 timeout 900 python tools/dropin_demo.py --protein --queries 3000 --genes 200 --modes Q7 --gpu-threads 16 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
for r in d['runs']:
    print('protein 3000 scout $m', 'ref', r['reference']['wall_s'], 'gpu', r['gpu']['wall_s'], 'identical', r['identical'])
    print(r['gpu'].get('shim','')[-700:])"
done
