for C in 2 3 4 1 2; do
  SPDP_CHUNKS=$C timeout 300 python bench.py --steps 3 --warmup 1 --legs none --seeded-pairs 0 --cpu-sample 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('SPDP_CHUNKS=$C', d['value'], d['ms_per_step'], 'udh_ms', c.get('udh_ms'), 'fwd_ms', c.get('fwd_ms'))"
done
