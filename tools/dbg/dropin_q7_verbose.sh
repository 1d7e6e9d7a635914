mkdir -p gpurun_out; rm -f gpurun_out/q7_verbose.txt
SPDP_SEED_VERBOSE=1 DROPIN_STDERR=gpurun_out/q7_verbose.txt SPDP_SEED_DUMP=/tmp/seed_dump.txt timeout 300 python tools/dropin_demo.py --queries 4000 --genes 200 --modes Q7 --gpu-threads 2000 > /dev/null 2>&1
grep -c . gpurun_out/q7_verbose.txt; grep "seeded\]" gpurun_out/q7_verbose.txt | tail -12
python - <<'PY'
import numpy as np
d=np.loadtxt('/tmp/seed_dump.txt',dtype=np.int64)
print('requests',len(d),'batches',len(set(d[:,0])))
rows,cols,kind,cut=d[:,2],d[:,3],d[:,1],d[:,6]
for name,v in (('rows',rows),('cols',cols),('band',d[:,5]-d[:,4])):
    print(name,'mean',v.mean(),'pct 50/90/99/max',np.percentile(v,[50,90,99,100]))
print('kinds',{int(k):int((kind==k).sum()) for k in set(kind)}, 'with cut',int((cut>0).sum()))
cells=rows*np.minimum(cols, d[:,5]-d[:,4]+rows)
print('cells total',cells.sum(),'mean',cells.mean(),'p99',np.percentile(cells,99))
PY
