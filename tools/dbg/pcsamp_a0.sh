ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pcsamp; rm -rf $OUT; mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 -L 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -60 > $OUT/avail.txt
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit ${UNIT:-cycles} --pc-sampling-method ${METHOD:-stochastic} --pc-sampling-interval ${IVAL:-1048576} -d $OUT/p -o p --output-format csv -- python bench.py --engines a0 --queries 300 --steps 1 --warmup 0 --legs none --seeded-pairs 0 --cpu-sample 8 > $OUT/run.txt 2>&1
tail -5 $OUT/run.txt | cut -c1-300
ls -la $OUT/p | head
f=$(ls $OUT/p/*pc_sampling*csv 2>/dev/null | head -1); echo $f; head -3 $f; wc -l $f
