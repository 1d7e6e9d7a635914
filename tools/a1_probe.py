import sys, time
sys.path.insert(0, '.')
import numpy as np
from spaln_amd import abi, defaults, engine, synth
intpen, t53 = defaults.exact_tables()
eng = engine.Engine(0)
for eng_mode in (2,):
    sc = defaults.scoring(scalar_engines=eng_mode, intpen=intpen, t53=t53)
    for n in (256, 1000):
        ps = abi.ProblemSet()
        for w, q, s5, s3, _ in synth.make_batch(n, seed=7):
            ps.add(q, w, s5, s3, **synth.exact_inputs(w))
        t0 = time.perf_counter(); s = eng.homscore_s(sc, ps); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); s = eng.homscore_s(sc, ps); dt = time.perf_counter() - t0
        cells = sum((p.a_right - p.a_left) * 1.0 * 0 for p in ps.items)
        print("engines", eng_mode, "n", n, "homscore s", round(dt, 3), "per problem ms", round(dt / n * 1e3, 3), flush=True)
eng.close()
