#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for lanes in "2,1,1" "2,1,1,2" "2,1,1,4" "2,1,2,4" "2,2,2,6" "3,2,2,8"; do
  echo "== lanes $lanes"
  SPDP_SEED_LANES=$lanes SPDP_MAP_VERBOSE=1 timeout 900 python tools/e2e_q7.py --queries 5000 --genes 200 2>&1 >/tmp/o.json | grep "^\[map\]" | tail -1 | cut -c1-330
  python -c "import json;d=json.load(open('/tmp/o.json'));print(d['identical_exon_tables'], d['library_s'], d['library_over_reference'])"
done
