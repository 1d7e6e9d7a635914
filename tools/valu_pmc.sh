#!/bin/bash
# VALU utilisation of the DP kernels from SQ counters (one --pmc pass, no trace domains).
# usage: tools/valu_pmc.sh c2|c3 <tag>
set -u
WL=${1:-c2}; TAG=${2:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/valu_${TAG}_${WL}
mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $OUT/p -o p --output-format csv -- \
    python bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 16 --legs none --seeded-pairs 0 > $OUT/bench.json 2> $OUT/bench.err
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv > $OUT/valu_pmc.txt 2>&1
grep '^{' $OUT/bench.json | cut -c1-400 >> $OUT/valu_pmc.txt
cat $OUT/valu_pmc.txt
