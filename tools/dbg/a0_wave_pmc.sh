#!/bin/bash
# where the resident wave cycles of the -A0 kernels go: one --pmc pass per counter group (128 x 2 kb queries)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/a0_wave; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/a0_small.py <<'PY'
import sys, os
sys.path.insert(0, '.')
from spaln_amd import abi, defaults, engine, synth
intpen, t53 = defaults.exact_tables()
eng = engine.Engine(0)
sc = defaults.scoring(scalar_engines=1, intpen=intpen, t53=t53)
ps = abi.ProblemSet()
for w, q, s5, s3, _ in synth.make_batch(int(os.environ.get("NQ", "512")), seed=7, mrna_len=2000):
    ps.add(q, w, s5, s3, **synth.exact_inputs(w))
print(len(eng.align_s(sc, ps)))
eng.close()
PY
i=0
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G -d $OUT/p$i -o p --output-format csv -- python /tmp/a0_small.py > $OUT/run$i.txt 2>&1
  python tools/pmc_summary.py $OUT/p$i/p_counter_collection.csv 2>&1 | grep -A12 "rowwave" > $OUT/pmc$i.txt
  tail -2 $OUT/run$i.txt | cut -c1-200
  cat $OUT/pmc$i.txt
done
