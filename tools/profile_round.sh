#!/bin/bash
# Profiles one workload of bench.py on the GPU box: kernel trace (+ the bench line) and two PMC passes
# (FETCH_SIZE, WRITE_SIZE) kept separate from the trace.  usage: tools/profile_round.sh c2|c3 <tag>
set -u
WL=${1:-c2}; TAG=${2:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --workload $WL --steps 3 --warmup 1 --legs none --seeded-pairs 0 > $OUT/bench.json 2> $OUT/bench.err
python tools/prof_summary.py $OUT/kt/kt_results.db > $OUT/kernel_stats.txt
grep '^{' $OUT/bench.json >> $OUT/kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- python bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 16 --seeded-pairs 0 --legs none > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- python bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 16 --seeded-pairs 0 --legs none > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pf/pf_counter_collection.csv $OUT/pw/pw_counter_collection.csv > $OUT/hbm_traffic_pmc.txt 2>&1
cat $OUT/kernel_stats.txt | cut -c1-330; cat $OUT/hbm_traffic_pmc.txt
