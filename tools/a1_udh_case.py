"""hirschbergS1 on one sub-range: GPU (spdp_scalar_udh under SPDP_UDH_ENGINE_A1) against the oracle's exact_udh"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPDP_UDH_ENGINE_A1"] = "1"
from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, engine
from oracle import oracle

name, al, ar, bl, br, exg, n_im = sys.argv[1], *[int(x) for x in sys.argv[2:6]], sys.argv[6], int(sys.argv[7])
fx = spdg.load([f for f in golden_files("s1_") if f.endswith(name + ".spdg")][0])
extra = dict(cano5=fx["cano5"], cano3=fx["cano3"], dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
sc = spdg.scoring(fx, scalar_engines=2)
eng = engine.Engine(0)
ps = abi.ProblemSet()
p = ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, tuple(int(c) for c in exg), **extra)
intvl = (ar - al + n_im) // (n_im + 1)
for pipe in ("0", "1"):
    os.environ["SPDP_A1_PIPE"] = pipe
    scores, cpos, ranges, flags = eng.scalar_udh(sc, ps, n_im, intvl)
    print("GPU pipe", pipe, int(scores[0]), ranges[0].tolist())
    for row in cpos[0]:
        print("   ", [int(x) if x < 2147483000 else "E" for x in row])
ws, wc, wr = oracle.exact_udh(sc, p, n_im)
print("oracle", ws, [int(x) for x in wr])
for row in wc:
    print("   ", [int(x) if x < 2147483000 else "E" for x in row])
eng.close()
