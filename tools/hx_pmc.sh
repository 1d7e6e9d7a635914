#!/bin/bash
# counters of the protein -A1 / -A0 engines on the C3 bench shape.  usage: tools/hx_pmc.sh [a1|a0] [queries]
set -u
ENGS=${1:-a1}; NQ=${2:-1000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/hx_pmc_$ENGS
mkdir -p $OUT
SPDP_TRACE_RUNS=1 timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/p -o p --output-format csv -- python bench.py --workload c3 --engines $ENGS --queries $NQ --steps 1 --warmup 0 --legs none --cpu-sample 8 > $OUT/run.txt 2>&1
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv > $OUT/pmc.txt 2>&1
grep "spdp run" $OUT/run.txt | tail -4; cat $OUT/pmc.txt
