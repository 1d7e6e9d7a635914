cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, numpy as np
sys.path.insert(0, '.')
from spaln_amd import synth
rng = np.random.default_rng(1)
genes = [synth.make_gene(np.random.default_rng(100 + i)) for i in range(60)]
td = "/tmp/dd"; os.makedirs(td, exist_ok=True)
with open(td + "/gnm.mfa", "w") as f:
    parts = []
    for g in genes: parts += [synth.random_dna(rng, 5000), g.window]
    s = bytes(np.concatenate(parts)).decode()
    f.write(">chr1\n"); f.writelines(s[i:i+60] + "\n" for i in range(0, len(s), 60))
with open(td + "/q.fa", "w") as f:
    for i in range(1000): f.write(f">q{i}\n{bytes(synth.mutate(rng, genes[i % 60].query, 0.02, 0.002)).decode()}\n")
PY
export ALN_TAB=$PWD/oracle/_ref/table ALN_DBS=/tmp/dd
cd /tmp/dd && $GRAFT_REPO_ROOT/oracle/_ref/spaln -W -KD gnm.mfa > /dev/null 2>&1
( time $GRAFT_REPO_ROOT/oracle/_ref/spaln_gpu -Q7 -S1 -O4 -t1000 -dgnm q.fa > /dev/null ) 2>&1 | tail -6
