#!/bin/bash
# LDS-side counters of the DP kernels (one --pmc pass, no trace domains).  usage: tools/lds_pmc.sh c2|c3 <tag>
set -u
WL=${1:-c2}; TAG=${2:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/lds_${TAG}_${WL}
mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p -o p --output-format csv -- \
    python bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 16 > $OUT/bench.json 2> $OUT/bench.err
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv > $OUT/lds_pmc.txt 2>&1
cat $OUT/lds_pmc.txt
