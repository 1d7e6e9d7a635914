#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -f /tmp/tr_fix.txt /tmp/tr_shim.txt
SPDP_SEED_TRACE=/tmp/tr_fix.txt python tools/dbg/live_h.py 2>&1 | grep -v "spdp run" | head -3
python - <<'P'
import os, sys, subprocess, argparse
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import dropin_demo
a = argparse.Namespace(queries=10000, genes=200, threads=16, protein=True)
td = "/tmp/dq10k"; os.makedirs(td, exist_ok=True)
_, env = dropin_demo.make_dataset(td, a)
qs = open(td + "/q.fa").read().split(">")[1:]
open(f"{td}/one.fa", "w").write(">" + qs[7555])
r = subprocess.run([f"{dropin_demo.REF}/spaln_gpu", "-Q7", "-O4", "-t16", "-dgnm", "one.fa"], cwd=td, env=dict(env, SPDP_SEED_TRACE="/tmp/tr_shim.txt"), capture_output=True, text=True)
print(r.stdout[-900:]); print(r.stderr[-300:])
P
echo "== fixture run (first walk)"; awk 'NR<=8' /tmp/tr_fix.txt | cut -c1-400
echo "== shim run"; cat /tmp/tr_shim.txt | cut -c1-400
