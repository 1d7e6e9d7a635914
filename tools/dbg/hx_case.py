import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import spdg
from tests.conftest import golden_files
from tests.test_gpu_exact_h import _subranges
from spaln_amd import abi, synth, engine
from oracle import oracle
eng = engine.Engine(0)
fx = spdg.load([f for f in golden_files("h1_") if f.endswith("h1_400aa.spdg")][0])
rng = np.random.default_rng(synth.SEED + 96)
sc = spdg.scoring_h(fx, scalar_engines=2)
for n_im in (1, 3):
    m = 120 + 10 * n_im
    ps = _subranges(fx, rng, 16, m, m)
    out = {}
    for mode in ("0", "1"):
        os.environ["SPDP_HX_PIPE"] = mode
        out[mode] = eng.scalar_udh_h(sc, ps, n_im, (m + n_im) // (n_im + 1))
    for i, p in enumerate(ps.items):
        ws, wcpos, wrng = oracle.exact_udh_h(sc, p, n_im)
        a = [int(out[k][0][i]) for k in "01"]
        same = all(out["0"][j][i].tolist() == out["1"][j][i].tolist() for j in (1, 2)) and a[0] == a[1]
        if not same or a[0] != ws:
            print("case", n_im, i, "rng", p.a_left, p.a_right, p.b_left, p.b_right, "exg", p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr)
            print("  oracle", ws, wrng.tolist(), wcpos.reshape(-1, 10)[: n_im + 1].tolist())
            for k in "01":
                print("  pipe", k, int(out[k][0][i]), out[k][2][i].tolist(), out[k][1][i].reshape(-1, 10)[: n_im + 1].tolist())
