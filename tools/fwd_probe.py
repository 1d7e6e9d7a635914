#!/usr/bin/env python3
"""Throughput of spdp_sweep<FL_FORWARD> on whole C2 problems (no slabs): run under
rocprofv3 --kernel-trace --stats and divide the printed cell count by the kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spaln_amd import abi, defaults, engine, synth
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = engine.Engine(0)
sc = defaults.scoring()
ps = abi.ProblemSet()
for w, q, s5, s3, _ in synth.make_batch(n, seed=99):
    ps.add(q, w, s5, s3)
cells = sum(oracle.cells(p, oracle.stripe(p, sc.sh)) for p in ps.items)
for rep in range(3):
    t = time.perf_counter()
    eng.wip_forward(sc, ps)
    print("forward", n, "problems", cells, "cells", round(time.perf_counter() - t, 3), "s wall (incl. D2H)")
for rep in range(2):
    t = time.perf_counter()
    eng.wip_scoreonly(sc, ps)
    print("score", round(time.perf_counter() - t, 3), "s wall")
for rep in range(2):
    t = time.perf_counter()
    eng.wip_udh(sc, ps, 8)
    print("udh8", round(time.perf_counter() - t, 3), "s wall")
