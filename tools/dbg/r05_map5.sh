#!/bin/bash
cd /root/repo
for q in 50000; do
  SPDP_MAP_CHUNK_MB=2048 SPDP_MAP_VERBOSE=1 SPDP_SEED_VERBOSE=1 timeout 1200 python tools/e2e_q7.py --queries $q --genes 200 2>/tmp/e.txt >/tmp/o.json
  grep "^\[map\]\|lane" /tmp/e.txt | tail -6 | cut -c1-330
  python -c "import json;d=json.load(open('/tmp/o.json'));print(d['queries'], d['identical_exon_tables'], d['reference_wall_s'], d['library_s'], d['library_over_reference'])"
done
free -g | head -2
