import os, sys, time, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.time()
def log(*a):
    print("[%.1fs]" % (time.time() - t0), *a, flush=True)
from tests import spdg
from spaln_amd import engine, abi
log("imports done")
eng = engine.Engine(0)
log("engine", eng.device_name())
name = sys.argv[1] if len(sys.argv) > 1 else "s1_tiny_m8"
what = sys.argv[2] if len(sys.argv) > 2 else "score"
fx = spdg.load(os.path.join(ROOT, "tests", "golden", name + ".spdg"))
sc = spdg.scoring(fx)
ps, p = spdg.problem(fx)
log("problem", name, p.a_right, p.b_right)
if what == "score":
    log("score", eng.wip_scoreonly(sc, ps), fx["wip_qn_score"])
elif what == "fwd":
    r = eng.wip_forward(sc, ps)
    log("fwd", r[0][0], r[0][1].ravel().tolist(), fx["wip_qn_fwd_scr"], fx["wip_qn_fwd_skl"].tolist())
else:
    n_im = int(what[3:])
    r = eng.wip_udh(sc, ps, n_im)
    log("udh", r, fx["wip_qn_udh%d_scr" % n_im], fx["wip_qn_udh%d_cpos" % n_im].reshape(-1, 10), fx["wip_qn_udh%d_rng" % n_im])
