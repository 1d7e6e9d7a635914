ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/a0h_pmc; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_WAVES"; do
timeout 600 rocprofv3 --pmc $set -d $OUT/p -o p --output-format csv -- python bench.py --workload c3 --engines a0 --queries 1000 --steps 1 --warmup 0 --legs none --cpu-sample 8 > $OUT/run.txt 2>&1
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv | grep -A9 spdh_rowwave
rm -rf $OUT/p
done
