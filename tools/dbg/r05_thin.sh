#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
python tools/dbg/thin_long.py 25 50000 1,8,256,1024
python tools/dbg/thin_long.py 64 50000 1,256
python tools/dbg/thin_long.py 7 30000 1
rm -rf /tmp/pmc; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pmc -o p --output-format csv -- python tools/dbg/thin_long.py 25 50000 1 > /dev/null 2>&1
python - <<'P'
import csv,glob
f=glob.glob('/tmp/pmc/**/*counter_collection.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'rowwave' in r['Kernel_Name']]
import collections
acc=collections.defaultdict(float)
disp=set()
for r in rows:
    acc[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
print(len(disp),'dispatches', rows[0]['Kernel_Name'][:60])
for k,v in acc.items(): print(k, v/len(disp), 'per step', v/len(disp)/50025)
P
