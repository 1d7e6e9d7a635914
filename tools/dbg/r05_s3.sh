#!/bin/bash
cd /root/repo
timeout 900 python tools/dropin_demo.py --queries 600 --genes 120 --modes Q7,Q4 --gpu-threads 16 --strand=-S3 --antisense 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
for r in d['runs']:
    print(r['mode'], 'ref', r['reference']['wall_s'], 'gpu', r['gpu']['wall_s'], 'identical', r['identical'], r.get('records_differing'), 'ratio', r.get('gpu_over_reference_wall'))
    print('   ', r['gpu'].get('shim','')[:400].replace(chr(10),' | '))
"
tail -3 /tmp/e.txt
