#!/bin/bash
cd /root/repo
run() {
  echo "== $*"
  env "$@" SPDP_MAP_VERBOSE=1 SPDP_SEED_VERBOSE=1 timeout 1200 python tools/e2e_q7.py --queries 20000 --genes 200 2>/tmp/e.txt >/tmp/o.json
  grep "lane" /tmp/e.txt | tail -4 | cut -c1-100
  python -c "import json;d=json.load(open('/tmp/o.json'));print(d['identical_exon_tables'], d['library_s']['align'], d['library_s']['map_align_call'], d['library_over_reference'])"
}
run SPDP_SEED_WALKS=20000
run SPDP_SEED_WALKS=20000 SPDP_SEED_BATCH=1024
run SPDP_SEED_WALKS=20000 SPDP_SEED_BATCH=4096
run SPDP_SEED_WALKS=10000 SPDP_SEED_BATCH=1024
