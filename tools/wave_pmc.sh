#!/bin/bash
# Where the resident wave time goes (SQ counters, one --pmc pass).  usage: tools/wave_pmc.sh c2|c3 <tag>
set -u
WL=${1:-c2}; TAG=${2:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/wave_${TAG}_${WL}
mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/p -o p --output-format csv -- \
    python bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 16 --legs none --seeded-pairs 0 $EXTRA > $OUT/bench.json 2> $OUT/bench.err
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv > $OUT/wave_pmc.txt 2>&1
cat $OUT/wave_pmc.txt
