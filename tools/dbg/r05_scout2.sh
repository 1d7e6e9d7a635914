#!/bin/bash
cd /root/repo
for rep in 1 2; do
for cfg in "5000 1" "20000 1" "20000 3"; do set -- $cfg
for m in 0 1 2; do
SPDP_SEED_SCOUT=$m timeout 1200 python tools/e2e_q7.py --queries $1 --genes 200 --ori $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rep $rep Q $1 ori $2 scout $m:', d['identical_exon_tables'], 'call', d['library_s']['map_align_call'], 'align', d['library_s']['align'])"
done; done; done
