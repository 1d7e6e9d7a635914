#!/usr/bin/env python3
"""Dev tool (build container only): the protein seeded walk against the compiled reference on many synthetic cases
(the twin of tools/seed_fuzz.py: `ref_dump -Q n` on a protein query records HSPs, Wilip replies, score + SKL of alignH_ng).

    python tools/seed_fuzz_h.py 200 [first_seed]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaln_amd import abi, synth  # noqa: E402
from tests import spdg  # noqa: E402
from oracle import seeded  # noqa: E402
from oracle import host_logic_h as hh  # noqa: E402
from tests.golden.seed_cases import make_case_h  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
ENV = dict(os.environ, ALN_TAB=os.path.join(ROOT, "oracle", "_ref", "table"))


def run_case(seed, td, verbose=False, joins=None):
    w, q, opts, desc = make_case_h(seed)
    gf, qf, of = (os.path.join(td, x) for x in ("g.fa", "q.fa", "o.spdg"))
    synth.write_fasta(gf, "win", w)
    synth.write_fasta(qf, "qry", q)
    try:
        r = subprocess.run([REF_DUMP, *opts, gf, qf, of], env=ENV, capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        return "ref-failed", desc + " the reference does not come back"   # (it loops on some inputs)
    if r.returncode:
        return "ref-failed", desc + " " + r.stderr.strip()[-120:]
    fx = spdg.load(of)
    bad, uns, und = [], 0, 0
    for alg, simd in ((0, 0), (2, 2)):
        sc = spdg.scoring_h(fx)
        _, p = spdg.problem_h(fx)
        sp = abi.seed_params_from_fixture(fx)
        j, n = seeded.hsps_of(fx)
        wl = seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"])
        try:
            scr, flat, rc = seeded.align_h_seeded(sc, sp, p, j, n, int(fx["seed_params"][1]), wl, simd, joins=joins)
        except (hh.ReferenceUndefined, hh.ReferenceFatal, hh.NotRestated):
            und += 1                     # the reference's own DP is undefined here (or a branch the oracle lacks)
            continue
        except Exception as e:  # noqa: BLE001
            bad.append(f"A{alg}: {type(e).__name__} {e}")
            continue
        if rc == 1:
            uns += 1
            continue
        want = fx[f"seed_skl_A{alg}"].tolist()
        if scr != int(fx[f"seed_scr_A{alg}"][0]) or (flat or []) != want:
            bad.append(f"A{alg}: score {scr} vs {int(fx[f'seed_scr_A{alg}'][0])}, skl {'equal' if (flat or []) == want else 'DIFFERENT'}")
            if verbose:
                print("  got ", flat)
                print("  want", want)
    st = ("MISMATCH " + "; ".join(bad)) if bad else ("unsupported" if uns == 2 else "undefined" if und else "ok")
    return st, desc + " " + " ".join(opts)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    joins, tally = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for seed in range(first, first + n):
            st, desc = run_case(seed, td, joins=joins)
            tally[st.split()[0]] = tally.get(st.split()[0], 0) + 1
            if st not in ("ok", "unsupported", "undefined"):
                print(f"seed {seed}: {st} | {desc}")
    print(tally)
    print({k: v for k, v in joins.items()})


if __name__ == "__main__":
    main()
