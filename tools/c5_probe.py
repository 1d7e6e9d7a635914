#!/usr/bin/env python3
"""C5 at full size: one 50 kb cDNA with 25 exons against its ~190 kb locus (recursive linear-space
branch all the way down).  Prints wall time and how many planted exon boundaries are corners."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spaln_amd import abi, defaults, engine, synth

rng = np.random.default_rng(synth.SEED + 55)
g = synth.make_gene(rng, n_exons=25, mrna_len=50000, flank=1000, intron_lo=1000, intron_hi=10000)
w, q = defaults.encode(g.window), defaults.encode(g.query)
s5, s3 = synth.splice_signals(g.window)
print("cDNA", len(q), "window", len(w), flush=True)
eng = engine.Engine(0)
sc = defaults.scoring()
ps = abi.ProblemSet()
ps.add(q, w, s5, s3)
for rep in range(2):
    t = time.perf_counter()
    (score, skl), = eng.align_s(sc, ps)
    dt = time.perf_counter() - t
    c = skl[1:]
    cols = set(int(x) for x in c[:, 1])
    hits = sum((e0 in cols) + (e1 in cols) for e0, e1 in g.exons)
    cells = float(len(q)) * len(w)
    print(f"run {rep}: {dt:.2f} s, score {score}, corners {len(c)}, exon boundaries found {hits}/50, "
          f"matrix {cells:.3g} cells -> {cells / dt / 1e9:.1f} GCUPS (matrix cells / wall)", flush=True)
