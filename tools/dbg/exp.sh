cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --legs none --seeded-pairs 0 --cpu-sample 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step_ms',d['ms_per_step'],'udh_ms',d['config'].get('udh_ms'),'fwd_ms',d['config'].get('fwd_ms'))"
mkdir -p gpurun_out/exp
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/exp/pw -o pw --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 8 --seeded-pairs 0 --legs none > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/exp/pw/pw_counter_collection.csv | grep -A1 sweep_fp
