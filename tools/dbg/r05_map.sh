#!/bin/bash
# the one-call map + align entry against the reference program: parity test, then 5 000 and 20 000 queries
cd /root/repo
mkdir -p gpurun_out
true
tail -3 gpurun_out/r05_map_test.log
for q in 5000; do
  SPDP_MAP_VERBOSE=1 SPDP_SEED_VERBOSE=1 timeout 900 python tools/e2e_q7.py --queries $q --genes 200 > gpurun_out/r05_map_$q.json 2> gpurun_out/r05_map_$q.err
  tail -c 1800 gpurun_out/r05_map_$q.json; tail -8 gpurun_out/r05_map_$q.err
done
