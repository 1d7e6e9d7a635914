#!/usr/bin/env python3
"""UDH / score-only sweep throughput vs number of problems per launch (engine-level API)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spaln_amd import abi, defaults, engine, synth
from oracle import oracle
eng = engine.Engine(0)
sc = defaults.scoring()
batch = synth.make_batch(10000, seed=99)
for n in (2000, 4096, 6000, 8000, 10000):
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in batch[:n]:
        ps.add(q, w, s5, s3)
    cells = sum(oracle.cells(p, oracle.stripe(p, sc.sh)) for p in ps.items[:200]) / 200 * n
    eng.wip_udh(sc, ps, 8)
    t = time.perf_counter(); eng.wip_udh(sc, ps, 8); d1 = time.perf_counter() - t
    eng.wip_scoreonly(sc, ps)
    t = time.perf_counter(); eng.wip_scoreonly(sc, ps); d2 = time.perf_counter() - t
    print(n, "udh wall %.3f s -> %.0f GCUPS | score wall %.3f s -> %.0f GCUPS" % (d1, cells / d1 / 1e9, d2, cells / d2 / 1e9), flush=True)
