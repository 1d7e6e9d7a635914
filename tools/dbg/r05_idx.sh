#!/bin/bash
cd /root/repo
timeout 1500 python bench.py --workload blk --queries 200000 --steps 3 --warmup 1 --legs none --seeded-pairs 0 > gpurun_out/r05_blk_idx.json 2> gpurun_out/r05_blk_idx.err
tail -3 gpurun_out/r05_blk_idx.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r05_blk_idx.json') if l.startswith('{')][-1])
print(d['value'], d['unit'])
print(json.dumps(d['config']['index_build'], indent=1))
P
