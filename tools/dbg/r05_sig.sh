#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "signal or upload or plain or e2e" 2>&1 | tail -3
SPDP_MAP_VERBOSE=1 timeout 1200 python tools/e2e_q7.py --queries 20000 --genes 200 2>/tmp/e.txt >/tmp/o.json
grep "^\[map\]" /tmp/e.txt | tail -2 | cut -c1-200
python -c "import json;d=json.load(open('/tmp/o.json'));print(d['queries'], d['identical_exon_tables'], d['reference_wall_s'], d['library_s'], d['library_over_reference'])"
