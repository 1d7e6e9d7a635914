"""Diagnostic: the -A0 wavefront engines (spdp_rowwave.hip) against the oracle, fixture by fixture."""
import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import spdg
from spaln_amd import abi, engine
from oracle import oracle

names = sys.argv[1:] or [os.path.basename(f)[:-5] for f in sorted(glob.glob("tests/golden/s1_*.spdg"))]
eng = engine.Engine(0)
bad = 0
for nm in names:
    fx = spdg.load(f"tests/golden/{nm}.spdg")
    sc = spdg.scoring(fx)
    ps, p = spdg.problem(fx)
    print(nm, "m", p.a_right - p.a_left, "n", p.b_right - p.b_left, flush=True)
    ws = oracle.scalar_scorealone(sc, p)
    print("   scorealone ...", flush=True)
    gs = int(eng.scalar_scorealone(sc, ps)[0])
    wf = oracle.scalar_forward(sc, p)
    print("   forward ...", flush=True)
    (gscr, gskl), = eng.scalar_forward(sc, ps)
    ok_s = ws == gs
    ok_f = wf[0] == gscr and np.array_equal(np.asarray(wf[1]).reshape(-1, 2), gskl)
    print("   score", ws, gs, "OK" if ok_s else "DIFF", "| forward", wf[0], gscr, "OK" if ok_f else "DIFF", flush=True)
    bad += (not ok_s) + (not ok_f)
print("mismatches:", bad)
