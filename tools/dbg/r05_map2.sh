#!/bin/bash
cd /root/repo
rm -f /tmp/seed_dump.txt
SPDP_SEED_DUMP=/tmp/seed_dump.txt timeout 900 python tools/e2e_q7.py --queries 5000 --genes 200 > /tmp/o.json 2> /tmp/e.txt
python - <<'P'
import numpy as np, collections
d=np.loadtxt('/tmp/seed_dump.txt',dtype=np.int64)
d=d[len(d)//2:]
long_=d[d[:,10]==3]
print("long-lane requests", len(long_))
per=collections.Counter(long_[:,7].tolist())
print("walks with long requests", len(per), "histogram of count per walk", sorted(collections.Counter(per.values()).items()))
worst=[q for q,c in per.most_common(3)]
for q in worst:
    print("walk", q)
    for r in d[d[:,7]==q]:
        print("   batch %d kind %d rows %d cols %d lw %d up %d cut %d a_left %d b_left %d lane %d" % (r[0],r[1],r[2],r[3],r[4],r[5],r[6],r[8],r[9],r[10]))
P
