#!/bin/bash
cd /root/repo
for cfg in "4 1" "4 2" "16 2" "16 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 1200 python tools/e2e_q7.py --queries 20000 --genes 200 --members $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('queues $1 members', d['members'], 'identical', d['identical_exon_tables'], 'call', d['library_s']['map_align_call'], 'ratio', d['library_over_reference'])"
done
