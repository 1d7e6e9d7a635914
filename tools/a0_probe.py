"""-A0 (exact-intron-length engines, spdp_rowwave.hip) throughput on a C2-shaped batch: python tools/a0_probe.py [queries]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spaln_amd import abi, defaults, engine, synth
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
intpen, t53 = defaults.exact_tables()
sc = defaults.scoring(scalar_engines=1, intpen=intpen, t53=t53, minl=25)
ps = abi.ProblemSet()
for w, q, s5, s3, _ in synth.make_batch(n, seed=4711):
    ps.add(q, w, s5, s3, **synth.exact_inputs(w))
cells = sum(oracle.cells(p, oracle.stripe(p, sc.sh)) for p in ps.items)
eng = engine.Engine(0)
for what, fn in (("HomScoreS_ng -A0 (scorealoneS_ng)", lambda: eng.homscore_s(sc, ps)),
                 ("alignS_ng -A0 (hirschbergS_ng + forwardS_ng)", lambda: eng.align_s(sc, ps, allow_partial=True))):
    fn()
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print(f"{what}: {n} queries, {cells:.3e} band cells, {dt * 1e3:.1f} ms -> {cells / dt / 1e9:.1f} GCUPS (band cells / wall)", flush=True)
