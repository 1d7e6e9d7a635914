for cfg in "2000" "256"; do
  timeout 300 python tools/dropin_demo.py --queries 20000 --genes 200 --modes Q7 --gpu-threads $cfg 2>/dev/null | python -c "
import sys,json
r=json.load(sys.stdin)['runs'][0]
print('threads $cfg: ref', r['reference']['wall_s'], 'gpu', r['gpu'].get('wall_s'), r['gpu'].get('shim','')[-700:])"
done
