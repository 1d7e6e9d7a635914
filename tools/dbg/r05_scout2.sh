#!/bin/bash
cd /root/repo
true
for rep in 1 2; do
for cfg in "2000 1 2" "5000 1 2" "20000 1 2" "20000 3 2"; do set -- $cfg
SPDP_SEED_SCOUT=$3 SPDP_SEED_VERBOSE=1 timeout 1200 python tools/e2e_q7.py --queries $1 --genes 200 --ori $2 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rep $rep Q $1 ori $2 scout $3:', d['identical_exon_tables'], 'call', d['library_s']['map_align_call'], 'align', d['library_s']['align'])"
done; done
