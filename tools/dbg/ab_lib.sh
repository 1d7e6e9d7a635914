# bench.py "$@" alternately on spaln_amd/libspdp_hip_prev.so and the current library
for i in 1 2; do for L in libspdp_hip_prev.so libspdp_hip.so; do
  SPDP_LIB=$PWD/spaln_amd/$L timeout 400 python bench.py "$@" --legs none --seeded-pairs 0 --cpu-sample 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$L', d['value'], d['ms_per_step'], 'udh_ms', c.get('udh_ms'), 'fwd_ms', c.get('fwd_ms'))"
done; done
