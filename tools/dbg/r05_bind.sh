#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "seeded or e2e" 2>&1 | tail -2
for cfg in "20000 3" "20000 1"; do set -- $cfg
SPDP_SEED_VERBOSE=1 timeout 1200 python tools/e2e_q7.py --queries $1 --genes 200 --ori $2 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('Q $1 ori $2:', d['identical_exon_tables'], 'call', d['library_s']['map_align_call'], 'align', d['library_s']['align'], d['library_over_reference'])"
grep "host CPU" /tmp/e.txt | tail -1
done
