#!/bin/bash
# counters of the -A0 wavefront kernels on a batch of 128 x 2 kb queries.  usage: ALIGN=1 tools/a0_pmc.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/a0_pmc
mkdir -p $OUT
cat > /tmp/a0_small.py <<'PY'
import sys
sys.path.insert(0, '.')
from spaln_amd import abi, defaults, engine, synth
intpen, t53 = defaults.exact_tables()
eng = engine.Engine(0)
import os
sc = defaults.scoring(scalar_engines=int(os.environ.get("ENG", "1")), intpen=intpen, t53=t53)
ps = abi.ProblemSet()
for w, q, s5, s3, _ in synth.make_batch(128, seed=7, mrna_len=2000):
    ps.add(q, w, s5, s3, **synth.exact_inputs(w))
print(eng.homscore_s(sc, ps)[:4])
if os.environ.get("ALIGN"): print(len(eng.align_s(sc, ps)))
print(sum((p.a_right - p.a_left) * (p.b_right - p.b_left) for p in ps.items) / 64)
eng.close()
PY
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAVES -d $OUT/p -o p --output-format csv -- python /tmp/a0_small.py > $OUT/run.txt 2>&1
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv > $OUT/pmc.txt 2>&1
grep "spdp run" $OUT/run.txt | tail -8; cat $OUT/pmc.txt
