"""scratch probe run on the GPU box (not a test)"""
import ctypes as C
import sys
from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, engine

eng = engine.Engine(0)
for f in golden_files("s1_"):
    if not any(k in f for k in sys.argv[1:] or ["1400nt"]):
        continue
    fx = spdg.load(f)
    sc = spdg.scoring(fx)
    ps = abi.ProblemSet()
    spdg.problem(fx, ps)
    n = len(ps)
    arr = (abi.Alignment * n)()
    rc = eng.lib.spdp_scalar_forward(eng.ctx, C.byref(sc), ps.array(), n, arr)
    print(f.split("/")[-1], "rc", rc, "n_skl", arr[0].n_skl, "score", arr[0].score, eng.lib.spdp_last_error(eng.ctx))
