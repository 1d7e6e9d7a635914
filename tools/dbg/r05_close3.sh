#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest3.log 2>&1; tail -3 gpurun_out/r05_gputest3.log
timeout 3000 python bench.py --steps 10 --warmup 2 > gpurun_out/r05_bench_final3.json 2> gpurun_out/r05_bench_final3.err
tail -c 300 gpurun_out/r05_bench_final3.json
rm -rf /tmp/p2
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o map -- python tools/e2e_q7.py --queries 20000 --genes 200 > gpurun_out/r05_map_20k.json 2>/dev/null
( echo "# rocprofv3 --kernel-trace --stats -- python tools/e2e_q7.py --queries 20000 --genes 200   (two spdp_map_align_s calls: first + warm)"; cat $(find /tmp/p2 -name '*kernel_stats.csv' | head -1); cat gpurun_out/r05_map_20k.json ) > gpurun_out/r05_map_kernel_stats.txt
head -6 gpurun_out/r05_map_kernel_stats.txt | cut -c1-160
