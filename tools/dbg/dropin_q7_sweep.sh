# -Q7 drop-in at 20 000 queries: worker threads / batch size of the shim
for cfg in "2000 0 0" "512 0 0" "128 0 0" "2000 500 0" "1000 250 0" "512 128 0" "256 64 2000"; do
  set -- $cfg
  env=""
  [ "$2" != 0 ] && export SPALN_GPU_BATCH=$2 || unset SPALN_GPU_BATCH
  [ "$3" != 0 ] && export SPALN_GPU_WAIT_US=$3 || unset SPALN_GPU_WAIT_US
  timeout 300 python tools/dropin_demo.py --queries 20000 --genes 200 --modes Q7 --gpu-threads $1 2>/dev/null | python -c "
import sys,json
r=json.load(sys.stdin)['runs'][0]
print('threads $1 batch $2 wait_us $3: ref', r['reference']['wall_s'], 'gpu', r['gpu'].get('wall_s'), 'identical', r.get('identical'), r['gpu'].get('shim','')[-60:])"
done
