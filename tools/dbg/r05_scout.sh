#!/bin/bash
cd /root/repo
Q=20000
env SPDP_SEED_TIMELINE=1 SPDP_MAP_VERBOSE=1 SPDP_SEED_VERBOSE=1 timeout 1200 python tools/e2e_q7.py --queries $Q --genes 200 2>/tmp/e.txt >/tmp/o.json
awk '/\[map\] regions/{n++} n>=2' /tmp/e.txt | grep "t = \|\[map\] chunk\|host CPU\|scout" | cut -c1-330 | grep -v "lane [01] starts" | head -40
awk '/\[map\] regions/{n++} n>=2' /tmp/e.txt | grep "lane [01] starts" | sed -n '1p;$p'
