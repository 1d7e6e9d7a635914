#!/bin/bash
# counters of the -A1 / -A0 score-only kernels on a small batch.  usage: tools/a1_pmc.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/a1_pmc
mkdir -p $OUT
cat > /tmp/a1_small.py <<'PY'
import sys
sys.path.insert(0, '.')
from spaln_amd import abi, defaults, engine, synth
intpen, t53 = defaults.exact_tables()
eng = engine.Engine(0)
import os
sc = defaults.scoring(scalar_engines=int(os.environ.get("ENG", "2")), intpen=intpen, t53=t53)
ps = abi.ProblemSet()
for w, q, s5, s3, _ in synth.make_batch(64, seed=7, mrna_len=500, n_exons=3, flank=300, intron_hi=1500):
    ps.add(q, w, s5, s3, **synth.exact_inputs(w))
print(eng.homscore_s(sc, ps)[:4])
if os.environ.get("ALIGN"): print(len(eng.align_s(sc, ps)))
print(sum((p.a_right - p.a_left) * (p.b_right - p.b_left) for p in ps.items) / 64)
eng.close()
PY
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/p -o p --output-format csv -- python /tmp/a1_small.py > $OUT/run.txt 2>&1
python tools/pmc_summary.py $OUT/p/p_counter_collection.csv > $OUT/pmc.txt 2>&1
cat $OUT/run.txt | tail -3; cat $OUT/pmc.txt
