"""-A0 aa x genome engines on sub-ranges of a fixture, tiles pipelined or one wave per problem (GPU box:
python tools/a0_h_probe.py [n])"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = engine.Engine(0)
fx = spdg.load([f for f in golden_files("h1_") if f.endswith("h1_400aa.spdg")][0])
q = fx["prm"]
rng = np.random.default_rng(5)
sc = spdg.scoring_h(fx, scalar_engines=1)
dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
ps = abi.ProblemSetH()
for i in range(n):
    m = int(rng.integers(300, q["a_right"] + 1))
    al = int(rng.integers(0, q["a_right"] - m + 1))
    bl = int(rng.integers(1, 300))
    br = int(rng.integers(q["b_right"] - 300, q["b_right"] + 1))
    ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
           fx["phs5"], fx["phs3"], al, al + m, bl, br, (1, 1, 1, 1), exin=(q["b_left"], q["b_right"]), dinc=dinc)
cells = sum((p.a_right - p.a_left) * (p.b_right - p.b_left) for p in ps.items)
for mode in ("0", "1", "0", "1"):
    os.environ["SPDP_A0_PIPE"] = mode
    t0 = time.perf_counter(); r1 = eng.scalar_forward_h(sc, ps); t1 = time.perf_counter()
    r0 = eng.scalar_forward_h(sc, ps, traceback=False); t2 = time.perf_counter()
    r2 = eng.align_h(sc, ps); t3 = time.perf_counter()
    print(f"pipe {mode}: n {n} cells {cells:.3g}  forward {1e3 * (t1 - t0):.1f} ms ({cells / (t1 - t0) / 1e9:.2f} GCUPS)  "
          f"score {1e3 * (t2 - t1):.1f} ms ({cells / (t2 - t1) / 1e9:.2f})  align_h {1e3 * (t3 - t2):.1f} ms", flush=True)
