#!/bin/bash
# where does q7555 (10 000 proteins, -Q7) differ: alone, in a small file, in the whole run?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'P'
import os, sys, subprocess, argparse
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import dropin_demo
a = argparse.Namespace(queries=10000, genes=200, threads=16, protein=True)
td = "/tmp/dq10k"; os.makedirs(td, exist_ok=True)
_, env = dropin_demo.make_dataset(td, a)
qs = open(td + "/q.fa").read().split(">")[1:]
def write(name, idx):
    open(f"{td}/{name}.fa", "w").write("".join(">" + qs[i] for i in idx))
write("one", [7555]); write("few", list(range(7500, 7600))); write("k2", list(range(7000, 9000)))
REF = dropin_demo.REF
for name in ("one", "few", "k2", "q"):
    out = {}
    for exe in ("spaln", "spaln_gpu"):
        r = subprocess.run([f"{REF}/{exe}", "-Q7", "-O4", "-t16", "-dgnm", f"{name}.fa"], cwd=td, env=dict(env, SPALN_GPU_DUMPJOB="q7555:/tmp/dq10k/job_" + name), capture_output=True, text=True)
        recs = dropin_demo.records(r.stdout)
        out[exe] = {b.splitlines()[-1].split()[6] if False else [l for l in b.splitlines() if l.startswith("@")][0].split()[7]: b for b in recs}
    a_, b_ = out["spaln"], out["spaln_gpu"]
    diff = [k for k in a_ if a_[k] != b_.get(k)]
    print(name, "queries", len(a_), "differing", diff[:10])
    for k in diff[:3]:
        print(a_[k]); print(b_.get(k))
P
