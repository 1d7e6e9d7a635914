#!/bin/bash
cd /root/repo
python - <<'P'
import sys, os, types, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import dropin_demo
a=types.SimpleNamespace(protein=False, genes=40, queries=6, threads=4, antisense=True)
d='/tmp/s3d'; os.makedirs(d,exist_ok=True)
tot, env = dropin_demo.make_dataset(d, a)
outs={}
for exe in ('spaln','spaln_gpu'):
    r=subprocess.run([os.path.join(dropin_demo.REF,exe),'-Q7','-O4','-t1','-dgnm','q.fa'],cwd=d,env=env,capture_output=True,text=True)
    outs[exe]=r.stdout
    print(exe, r.returncode, r.stderr[-300:])
A=outs['spaln'].splitlines(); B=outs['spaln_gpu'].splitlines()
for l in A:
    if l.startswith('@') or l.startswith('q1\t'): print('REF', l[:140])
for l in B:
    if l.startswith('@') or l.startswith('q1\t'): print('GPU', l[:140])
P
