#!/usr/bin/env python3
"""latency of ONE thin, long forwardS_ng request (a few query rows against tens of thousands of columns: the terminal
stretches of the seeded walk on a block-search locus) and of n of them side by side"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spaln_amd import abi, engine, synth
from tests import spdg

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
counts = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,8,256,1024".split(","))]
fq = spdg.load(os.path.join(ROOT, "tests", "golden", "q_c2_seed0.spdg"))
eng = engine.Engine(0)
rng = np.random.default_rng(7)
CODE = np.array([2, 3, 5, 9], dtype=np.uint8)
sigmodel = abi.signal_model_from_fixture(fq)
ip = np.ascontiguousarray(spdg.load(os.path.join(ROOT, "tests", "golden", "blk_k1.spdg"))["find_intpen"], dtype=np.int16)
sc = spdg.scoring(fq, intpen=ip, scalar_engines=1)
for n in counts:
    ps = abi.ProblemSet()
    for i in range(n):
        b = CODE[rng.integers(0, 4, size=cols)]
        a = CODE[rng.integers(0, 4, size=rows)]
        sg = eng.splice_signals(sigmodel, b, 0, cols)
        ps.add(a, b, sg["sig5"], sg["sig3"], 0, rows, 0, cols, (1, 1, 1, 1), cano5=sg["cano5"], cano3=sg["cano3"], dinc=sg["dinc"])
    eng.scalar_forward(sc, ps)
    t0 = time.perf_counter()
    r = eng.scalar_forward(sc, ps)
    dt = time.perf_counter() - t0
    print(f"rows {rows} cols {cols} n {n}: {dt * 1e3:.1f} ms per call, {dt / (rows + cols) * 1e6:.2f} us per anti-diagonal, score {r[0][0]}")
eng.close()
