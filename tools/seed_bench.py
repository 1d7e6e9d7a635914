#!/usr/bin/env python3
"""On an MI355X box: the seeded path (-Q7) of a BATCH at BASELINE's headline shape, timed and compared.

oracle/_ref/seed_bench (oracle/ref_build/seed_bench.cc) sets every pair up as the reference's match_2 does, lets the
reference's geneorient() find the HSPs, then runs (1) the reference's own alignS_ng / alignH_ng on all host cores and
(2) ONE spdp_align_s_seeded / spdp_align_h_seeded call on the whole batch, the reference's Wilip behind the HSP callback;
results are compared pair by pair.  This is the measurement of SURVEY section 8 row f2.

    python tools/seed_bench.py [n_pairs] [c2 | c3] [Q level 1..3] [threads]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaln_amd import synth  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "seed_bench")
ENV = dict(os.environ, ALN_TAB=os.path.join(ROOT, "oracle", "_ref", "table"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    shape = sys.argv[2] if len(sys.argv) > 2 else "c2"
    q = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "list.txt"), "w") as lf:
            for i in range(n):
                rng = np.random.default_rng(synth.SEED + 20000 + i)
                if shape == "c3":
                    g = synth.make_protein_gene(rng, n_exons=6, aa_len=400, flank=1000, sub=0.10, intron_hi=5000)
                else:
                    g = synth.make_gene(rng, sub=0.04 + 0.01 * (i % 5), indel=0.005)
                gf, qf = os.path.join(td, f"g{i}.fa"), os.path.join(td, f"q{i}.fa")
                synth.write_fasta(gf, "win", g.window)
                synth.write_fasta(qf, "qry", g.query)
                lf.write(f"{gf} {qf}\n")
        prefix = os.environ.get("SEED_BENCH_PREFIX", "").split()     # e.g. a profiler in front of the binary
        extra = os.environ.get("SEED_BENCH_OPTS", "").split()        # e.g. "-A 0" (engines behind the walk), "-X 1"
        r = subprocess.run(prefix + [BIN, "-Q", str(q), "-t", str(threads)] + extra + [os.path.join(td, "list.txt")], env=ENV,
                           capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    try:
        d = json.loads(line)
    except ValueError:
        print(r.stdout[-2000:], r.stderr[-2000:])
        return 2
    d["shape"] = shape
    d["Q"] = q + 4
    d["reference_pairs_per_s"] = round(d["walked"] / d["reference_s"], 1)
    d["library_pairs_per_s"] = round(d["walked"] / d["library_s"], 1)
    print(json.dumps(d))
    if r.returncode not in (0, 1) or os.environ.get("SPDP_SEED_VERBOSE"):
        print(r.stderr[-2000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
