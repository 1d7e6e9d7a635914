"""diagnostic (not a test): print every GPU / oracle difference of the linear-space fuzz cases"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spaln_amd import abi, defaults, synth, engine
from oracle import oracle
import importlib.util
spec = importlib.util.spec_from_file_location('fz', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'test_gpu_fuzz.py'))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
eng = engine.Engine(0)
for seed in (1, 2, 3):
    rng = np.random.default_rng(synth.SEED + 9000 + seed)
    for rnd in range(4):
        sc = fz._rand_scoring_s(rng)
        ps = abi.ProblemSet()
        for _ in range(40):
            fz._rand_problem_s(rng, ps)
        big = abi.ProblemSet()
        big.items = [p for p in ps.items if p.a_right - p.a_left >= 40]
        big._keep = ps._keep
        if not len(big):
            continue
        n_im = int(rng.integers(1, 3))
        us, ucpos, urng = eng.wip_udh(sc, big, n_im)
        for i, p in enumerate(big.items):
            ws, wcpos, wrng = oracle.wip_udh(sc, p, n_im)
            if int(us[i]) != ws or urng[i].tolist() != wrng.tolist() or ucpos[i].tolist() != wcpos.tolist():
                print("S", seed, rnd, i, "n_im", n_im, "gpu", int(us[i]), urng[i].tolist(), ucpos[i][:, :5].tolist(),
                      "| oracle", ws, wrng.tolist(), wcpos[:, :5].tolist(),
                      "| prob", (p.a_left, p.a_right, p.b_left, p.b_right), (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr))
for seed in (1, 2):
    rng = np.random.default_rng(synth.SEED + 9200 + seed)
    for rnd in range(4):
        sc = fz._rand_scoring_h(rng)
        ps = abi.ProblemSetH()
        while len(ps) < 24:
            p = fz._rand_problem_h(rng, ps)
            if p.a_right - p.a_left < 34:
                ps.items.pop()
        n_im = int(rng.integers(1, 3))
        us, ucpos, urng = eng.wip_udh_h(sc, ps, n_im)
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng = oracle.wip_udh_h(sc, p, n_im)
            if int(us[i]) != ws or urng[i].tolist() != wrng.tolist() or ucpos[i].tolist() != wcpos.tolist():
                fs, fskl, fflag = oracle.wip_forward_h(sc, p)
                print("H", seed, rnd, i, "n_im", n_im, "gpu", int(us[i]), urng[i].tolist(), ucpos[i][:, :5].tolist(),
                      "| oracle", ws, wrng.tolist(), wcpos[:, :5].tolist(),
                      "| prob", (p.a_left, p.a_right, p.b_left, p.b_right), (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr),
                      "fwdflag", fflag)
