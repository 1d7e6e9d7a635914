#!/bin/bash
# closing run: the default bench line with its legs, kernel stats of the index builder and of the map + align call
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python bench.py --steps 10 --warmup 2 > gpurun_out/r05_bench_final2.json 2> gpurun_out/r05_bench_final2.err
tail -c 600 gpurun_out/r05_bench_final2.json
rm -rf /tmp/p1 /tmp/p2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o idx -- python tools/idx_scale.py --mb 1000 > gpurun_out/r05_idx_1g.json 2>/dev/null
( echo "# rocprofv3 --kernel-trace --stats -- python tools/idx_scale.py --mb 1000"; cat $(find /tmp/p1 -name '*kernel_stats.csv' | head -1); cat gpurun_out/r05_idx_1g.json ) > gpurun_out/r05_idx_kernel_stats.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o map -- python tools/e2e_q7.py --queries 20000 --genes 200 > gpurun_out/r05_map_20k.json 2>/dev/null
( echo "# rocprofv3 --kernel-trace --stats -- python tools/e2e_q7.py --queries 20000 --genes 200   (two spdp_map_align_s calls: first + warm)"; cat $(find /tmp/p2 -name '*kernel_stats.csv' | head -1); cat gpurun_out/r05_map_20k.json ) > gpurun_out/r05_map_kernel_stats.txt
head -12 gpurun_out/r05_idx_kernel_stats.txt | cut -c1-160
head -10 gpurun_out/r05_map_kernel_stats.txt | cut -c1-160
