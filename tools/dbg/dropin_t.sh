cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from spaln_amd import synth
rng = np.random.default_rng(1)
genes = [synth.make_gene(np.random.default_rng(100 + i)) for i in range(100)]
td = "/tmp/dd"; os.makedirs(td, exist_ok=True)
with open(td + "/gnm.mfa", "w") as f:
    parts = []
    for g in genes: parts += [synth.random_dna(rng, 5000), g.window]
    s = bytes(np.concatenate(parts)).decode()
    f.write(">chr1\n"); f.writelines(s[i:i+60] + "\n" for i in range(0, len(s), 60))
with open(td + "/q.fa", "w") as f:
    for i in range(2000): f.write(f">q{i}\n{bytes(synth.mutate(rng, genes[i % 100].query, 0.02, 0.002)).decode()}\n")
PY
export ALN_TAB=$PWD/oracle/_ref/table ALN_DBS=/tmp/dd
cd /tmp/dd && $GRAFT_REPO_ROOT/oracle/_ref/spaln -W -KD gnm.mfa > /dev/null 2>&1
python - <<'PY'
import subprocess, time, os
R=os.environ["GRAFT_REPO_ROOT"]+"/oracle/_ref/"
for exe,t in (("spaln",16),("spaln",1000),("spaln_gpu",1000),("spaln_gpu",2000)):
    t0=time.time(); r=subprocess.run([R+exe,"-Q7","-S1","-O4",f"-t{t}","-dgnm","q.fa"],cwd="/tmp/dd",capture_output=True,text=True,env=dict(os.environ,SPDP_SEED_VERBOSE="1")); dt=time.time()-t0
    print(exe,t,round(dt,2),"s", [l for l in r.stderr.splitlines() if "spaln_gpu]" in l or "[seeded] call" in l or "upload" in l][-4:])
PY
