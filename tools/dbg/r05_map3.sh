#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_map
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_map -o map -- python tools/e2e_q7.py --queries 5000 --genes 200 > gpurun_out/r05_map_prof.json 2> gpurun_out/r05_map_prof.err
f=$(find /tmp/prof_map -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r05_map_kernel_stats.csv
head -25 "$f"
t=$(find /tmp/prof_map -name '*kernel_trace.csv' | head -1)
python - "$t" <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# the longest single launches
rows.sort(key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']),reverse=True)
for r in rows[:25]:
    print((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,'ms',r['Kernel_Name'][:70],'grid',r.get('Grid_Size_X'),r.get('Workgroup_Size_X'))
P
