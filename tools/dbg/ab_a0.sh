for i in 1 2; do
for L in spaln_amd/libspdp_hip_prev.so spaln_amd/libspdp_hip.so; do
  SPDP_LIB=$PWD/$L timeout 300 python bench.py --engines a0 --queries 1000 --steps 2 --warmup 1 --legs none --seeded-pairs 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$L', d['value'], c.get('udh_gcups'), c.get('fwd_gcups'), d['roofline'].get('kernel_ms'))"
done; done
