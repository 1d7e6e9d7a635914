# the traceback sweep on spdp_kernels.hip (SPDP_FP_FWD=0) against the fp32-issue form, same box
for wl in "" "--workload c4 --queries 20000"; do
for i in 1 2; do
for F in 0 1; do
  SPDP_FP_FWD=$F timeout 400 python bench.py $wl --steps 3 --warmup 1 --legs none --seeded-pairs 0 --cpu-sample 16 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('wl [$wl] SPDP_FP_FWD=$F', d['value'], 'ms/step', d['ms_per_step'], 'fwd_ms', c.get('fwd_ms'), 'fwd_gcups', c.get('fwd_gcups'), 'udh_ms', c.get('udh_ms'))"
done; done; done
