#!/usr/bin/env python3
"""Dev tool (build container only): the seeded walk against the compiled reference on many synthetic cases.

For every case `oracle/_ref/ref_dump -Q n` runs the reference's own alignS_ng with seeding on and records HSPs, Wilip
replies and the result; the product's host walk (oracle/libwalkcheck.so = spaln_amd/csrc/spdp_seeded_walk.h over the
oracle's DP engines) must reproduce score and SKL.  Prints mismatches and which joins of interpolateS were reached.

    python tools/seed_fuzz.py 200 [first_seed]
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

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
ENV = dict(os.environ, ALN_TAB=os.path.join(ROOT, "oracle", "_ref", "table"))


from tests.golden.seed_cases import make_case  # noqa: E402


def run_case(seed, td, verbose=False, joins=None):
    w, q, opts, desc = make_case(seed)
    gf, qf, of = (os.path.join(td, x) for x in ("g.fa", "q.fa", "o.spdg"))
    synth.write_fasta(gf, "win", w)
    synth.write_fasta(qf, "qry", q)
    r = subprocess.run([REF_DUMP, *opts, gf, qf, of], env=ENV, capture_output=True, text=True)
    if r.returncode:
        return "ref-failed", desc + " " + r.stderr.strip()[-120:]
    fx = spdg.load(of)
    bad = []
    for alg, simd in ((0, 0), (2, 2)):
        sc = spdg.scoring(fx)
        ps = abi.ProblemSet()
        _, p = spdg.problem(fx, ps)
        h5, h3 = np.ascontiguousarray(fx["phs5"]), np.ascontiguousarray(fx["phs3"])
        p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
        sp = abi.seed_params_from_fixture(fx)
        j, n = seeded.hsps_of(fx)
        wl = seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"])
        try:
            scr, flat, rc = seeded.align_s_seeded(sc, sp, p, j, n, int(fx["seed_params"][1]), wl, simd, joins=joins)
        except Exception as e:  # noqa: BLE001
            bad.append(f"A{alg}: {type(e).__name__} {e}")
            continue
        want = fx[f"seed_skl_A{alg}"].tolist()
        if scr != int(fx[f"seed_scr_A{alg}"][0]) or (flat or []) != want:
            bad.append(f"A{alg}: score {scr} vs {int(fx[f'seed_scr_A{alg}'][0])}, skl {'equal' if (flat or []) == want else 'DIFFERENT'} rc={rc}")
            if verbose:
                print("  got ", flat)
                print("  want", want)
    return ("MISMATCH " + "; ".join(bad)) if bad else "ok", desc + " " + " ".join(opts)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    joins, tally = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for seed in range(first, first + n):
            st, desc = run_case(seed, td, joins=joins)
            tally[st.split()[0]] = tally.get(st.split()[0], 0) + 1
            if st != "ok":
                print(f"seed {seed}: {st} | {desc}")
    print(tally)
    print({k: v for k, v in joins.items()})


if __name__ == "__main__":
    main()
