#!/bin/bash
cd /root/repo
SPDP_MAP_VERBOSE=1 timeout 2400 python tools/e2e_q7.py --queries 20000 --genes 400 --spacer 450000 --frag 500 --ori 3 2>/tmp/e.txt > gpurun_out/r05_c4_e2e.json
tail -c 1500 gpurun_out/r05_c4_e2e.json; grep "\[map\] chunk" /tmp/e.txt | tail -2 | cut -c1-330; tail -5 /tmp/e.txt | cut -c1-300
