ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pcsamp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -A3 "pc.sampl" | head -20
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 -d $OUT/p -o p --output-format csv -- python bench.py --workload c3 --engines a0 --queries 1000 --steps 1 --warmup 0 --legs none --cpu-sample 8 > $OUT/run.txt 2>&1
tail -5 $OUT/run.txt | cut -c1-300
ls -la $OUT/p | head; 
f=$(ls $OUT/p/*pc_sampling*csv 2>/dev/null | head -1); echo $f; head -3 $f; wc -l $f
