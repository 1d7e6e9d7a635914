#!/usr/bin/env python3
"""What the intermediate rows cost the linear-space sweep: spdp_sweep_fp<FL_UDH> on 4096 C2 problems with 1 .. 16 rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spaln_amd import abi, defaults, engine, synth
from oracle import oracle
eng = engine.Engine(0)
sc = defaults.scoring()
batch = synth.make_batch(4096, seed=99)
ps = abi.ProblemSet()
for w, q, s5, s3, _ in batch:
    ps.add(q, w, s5, s3)
cells = sum(oracle.cells(p, oracle.stripe(p, sc.sh)) for p in ps.items[:200]) / 200 * len(batch)
for n_im in (1, 2, 4, 8, 16, 8, 1):
    eng.wip_udh(sc, ps, n_im)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); eng.wip_udh(sc, ps, n_im); best = min(best, time.perf_counter() - t)
    print(f"n_im {n_im:2d}: wall {best * 1e3:7.1f} ms -> {cells / best / 1e9:6.0f} GCUPS", flush=True)
