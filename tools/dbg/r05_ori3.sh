#!/bin/bash
cd /root/repo
timeout 900 python tools/e2e_q7.py --queries 2000 --genes 120 --ori 3 2>/tmp/e.txt >/tmp/o.json; tail -12 /tmp/e.txt | cut -c1-400; cat /tmp/o.json | cut -c1-900
timeout 900 python tools/e2e_q7.py --queries 2000 --genes 120 --ori 1 2>/tmp/e.txt >/tmp/o.json; tail -3 /tmp/e.txt | cut -c1-300; cat /tmp/o.json | cut -c1-500
