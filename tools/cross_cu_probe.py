#!/usr/bin/env python3
"""Cross-CU pass pipelines (one long cDNA spread over many CUs) against one-CU pipelines: same result, wall time."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from spaln_amd import abi, defaults, engine, synth

def case(mrna, n_exons, seed, lo, hi, **kw):
    rng = np.random.default_rng(synth.SEED + seed)
    g = synth.make_gene(rng, n_exons=n_exons, mrna_len=mrna, flank=1000, intron_lo=lo, intron_hi=hi)
    w, q = defaults.encode(g.window), defaults.encode(g.query)
    s5, s3 = synth.splice_signals(g.window)
    ps = abi.ProblemSet(); ps.add(q, w, s5, s3)
    return ps, defaults.scoring(**kw)

cases = [("6kb", *case(6000, 24, 505, 500, 3000, max_vmf_space=4 * 1024 * 1024), 12),
         ("20kb", *case(20000, 20, 77, 800, 6000), 4),
         ("50kb", *case(50000, 25, 55, 1000, 10000), 3)]
for name, ps, sc, reps in cases:
    os.environ["SPDP_CROSS"] = "0"
    eng = engine.Engine(0)
    t = time.perf_counter(); (ws, wskl), = eng.align_s(sc, ps); t0 = time.perf_counter() - t
    t = time.perf_counter(); eng.align_s(sc, ps); t0 = time.perf_counter() - t
    eng.close()
    os.environ["SPDP_CROSS"] = "1"
    eng = engine.Engine(0)
    bad = 0; ts = []
    for r in range(reps):
        t = time.perf_counter(); (s, skl), = eng.align_s(sc, ps); ts.append(time.perf_counter() - t)
        bad += not (s == ws and skl.tolist() == wskl.tolist())
    eng.close()
    print(f"{name}: one-CU {t0*1e3:.0f} ms, cross-CU {min(ts)*1e3:.0f} ms, {reps} runs, mismatches {bad}, score {ws}, corners {len(wskl)}", flush=True)
