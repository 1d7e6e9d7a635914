for E in "X=1" "SPDP_CROSS_WPB=4" "SPDP_CROSS=0" "SPDP_CROSS=4" "SPDP_CROSS=16"; do
  env $E SPDP_TRACE_RUNS=1 timeout 300 python bench.py --workload c5 --queries 32 --steps 2 --warmup 1 --cpu-sample 0 --legs none --seeded-pairs 0 2> /tmp/c5t.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', d['value'], d['ms_per_step'], d['config'].get('udh_gcups'))"
  grep "flavour 2" /tmp/c5t.txt | tail -2 | cut -c1-60,140-200
done
