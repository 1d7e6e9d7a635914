#!/usr/bin/env python3
"""On an MI355X box: the seeded paths LIVE against the compiled reference, many random cases.

oracle/_ref/shim_check -Q n runs the reference's geneorient() + alignS_ng / alignH_ng and, in the same process,
spdp_align_s_seeded / spdp_align_h_seeded with the reference's own Wilip behind the HSP callback.
No fixture in between: each case is one comparison of the product with the reference itself.

    python tools/gpu_shim_fuzz.py 200 [first_seed] [s | h | sp | hp | s3 | sp3 | c2 | c2w]

c2: BASELINE's headline shape (2 kb cDNA against its locus +- 1 kb), alternately -A0 without seeding (the reference's
exact engines are reliable at that size, its int16 ones are not) and -Q7 under -A2 (the DP calls between HSPs stay below the
rows where the int16 engines re-base).  s / h: cDNA / protein queries through the seeded paths; sp / hp: the same cases without seeding (-Q0), the engine
selector cycling through -A0 / -A1 / -A2 / -A3 -- HomScore*_ng and align*_ng of the reference against spdp_homscore_* /
spdp_align_* for every engine family.
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaln_amd import synth  # noqa: E402
from tests.golden.seed_cases import make_case, make_case_h  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "shim_check")
ENV = dict(os.environ, ALN_TAB=os.path.join(ROOT, "oracle", "_ref", "table"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    mode = sys.argv[3] if len(sys.argv) > 3 else "s"
    dagp = mode.endswith("3") and not mode.startswith("c2")    # s3 / sp3: the same cases under -yl3 (double affine gaps), -A0 engines
    if dagp:
        mode = mode[:-1]
    protein, plain = mode.startswith("h"), mode.endswith("p")
    tally = {}
    with tempfile.TemporaryDirectory() as td:
        gf, qf = os.path.join(td, "g.fa"), os.path.join(td, "q.fa")
        for seed in range(first, first + n):
            if mode == "c2w":                                 # the `_wip` engines themselves at 2 kb: divergent enough that the
                import numpy as np                            # reference's int16 scores never reach their re-basing threshold
                base = float(os.environ.get("C2W_SUB", "0.18"))
                g = synth.make_gene(np.random.default_rng(synth.SEED + 9500 + seed), sub=base + 0.02 * (seed % 6), indel=0.01)
                w, q, desc = g.window, g.query, f"C2 shape, {100 * (base + 0.02 * (seed % 6)):.0f} % substitutions"
                opts = ["-A", "2"]
            elif mode == "c2":
                import numpy as np
                g = synth.make_gene(np.random.default_rng(synth.SEED + 9000 + seed), sub=0.04 + 0.01 * (seed % 5), indel=0.005)
                w, q, desc = g.window, g.query, "C2 shape"
                opts = ["-A", "0"] if seed % 2 == 0 else ["-Q", "3"]
            else:
                w, q, opts, desc = (make_case_h if protein else make_case)(seed)
            keep, i = [], 0
            while i < len(opts):                              # the options shim_check knows
                if plain and opts[i] == "-Q":
                    i += 2
                elif opts[i] in ("-Q", "-X", "-V", "-A"):
                    keep += opts[i:i + 2]; i += 2
                elif opts[i] in ("-L", "-C"):
                    keep.append(opts[i]); i += 1
                else:
                    i += 2 if i + 1 < len(opts) and not opts[i + 1].startswith("-") else 1
            if plain:
                keep += ["-A", "0" if dagp else str(seed % 4)]
            if dagp:
                keep += ["-l", "3"] + ([] if plain else ["-A", "0"])
            synth.write_fasta(gf, "win", w)
            synth.write_fasta(qf, "qry", q)
            try:
                r = subprocess.run([BIN, *keep, gf, qf], env=ENV, capture_output=True, text=True, timeout=180 if mode.startswith("c2") else 60)
                rc = r.returncode
            except subprocess.TimeoutExpired:
                rc = -9
            key = {0: "identical", 1: "DIFFERENT", 4: "reverse strand", 5: "reference undefined", -9: "timeout (the reference loops on some inputs)",
                   -11: "crash (the reference runs first; ref_dump alone crashes on the same input)"}.get(rc, f"rc {rc}")
            if rc == 1 and "Unexpected dir" in r.stderr and "IDENTICAL" not in r.stdout and "DIFFERENT" not in r.stdout:
                key = "reference fatal"                       # the reference's own fatal("Unexpected dir"), exit status 1
            tally[key] = tally.get(key, 0) + 1
            if key == "DIFFERENT" or key.startswith("rc"):
                print(f"seed {seed}: {key} | {desc} {' '.join(keep)}\\n{(r.stdout + r.stderr)[-600:]}", flush=True)
    print(tally)
    return 1 if tally.get("DIFFERENT") else 0


if __name__ == "__main__":
    sys.exit(main())
