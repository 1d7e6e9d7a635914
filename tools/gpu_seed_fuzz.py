#!/usr/bin/env python3
"""Dev tool: the cDNA seeded path on the GPU against many reference -Q runs.

Step 1 (build container, reference present):   python tools/gpu_seed_fuzz_h.py make 150 200
    runs `oracle/_ref/ref_dump -Q` on make_case seeds [150, 350) and leaves the fixtures the CPU walk agrees with the
    reference on under gpurun_in/q_fuzz/ (not committed; the snapshot carries them to the GPU box)
Step 2 (GPU box):                               python tools/gpu_seed_fuzz_h.py run
    spdp_align_s_seeded on each of them, -A0 and -A2, against the recorded reference results
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from spaln_amd import abi, synth  # noqa: E402
from tests import spdg  # noqa: E402
from oracle import seeded  # noqa: E402
from oracle import host_logic_h as hh  # noqa: E402

DIR = os.path.join(ROOT, "gpurun_in", "q_fuzz")


def inputs(fx, alg):
    sc = spdg.scoring(fx)
    ps = abi.ProblemSet()
    _, p = spdg.problem(fx, ps)
    h5, h3 = np.ascontiguousarray(fx["phs5"]), np.ascontiguousarray(fx["phs3"])
    p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
    p._phs = (h5, h3)
    sp = abi.seed_params_from_fixture(fx)
    j, n = seeded.hsps_of(fx)
    return sc, sp, p, j, n, int(fx["seed_params"][1]), seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"])


def cpu_status(fx, alg, simd):
    sc, sp, p, j, n, lowest, wl = inputs(fx, alg)
    try:
        scr, flat, rc = seeded.align_s_seeded(sc, sp, p, j, n, lowest, wl, simd)
    except Exception as e:  # noqa: BLE001
        return "undefined"
    if rc == 1:
        return "unsupported"
    ok = scr == int(fx[f"seed_scr_A{alg}"][0]) and (flat or []) == fx[f"seed_skl_A{alg}"].tolist()
    return "ok" if ok else "MISMATCH"


def make(first, count):
    import seed_fuzz as F
    from tests.golden.seed_cases import make_case
    os.makedirs(DIR, exist_ok=True)
    kept = 0
    with tempfile.TemporaryDirectory() as td:
        for seed in range(first, first + count):
            w, q, opts, desc = make_case(seed)
            gf, qf, of = (os.path.join(td, x) for x in ("g.fa", "q.fa", "o.spdg"))
            synth.write_fasta(gf, "win", w)
            synth.write_fasta(qf, "qry", q)
            try:
                r = subprocess.run([F.REF_DUMP, *opts, gf, qf, of], env=F.ENV, capture_output=True, text=True, timeout=120)
            except subprocess.TimeoutExpired:
                continue
            if r.returncode:
                continue
            fx = spdg.load(of)
            st = [cpu_status(fx, 0, 0), cpu_status(fx, 2, 2)]
            if "MISMATCH" in st:
                print(f"seed {seed}: CPU walk differs from the reference {st} | {desc}")
                continue
            if st == ["unsupported", "unsupported"]:
                continue
            spdg.save(os.path.join(DIR, f"q_{seed:04d}.spdg"),
                      {k: v for k, v in fx.items() if k != "prm"})
            kept += 1
    print(f"{kept} fixtures under {DIR}")


def run():
    from spaln_amd import engine
    eng = engine.Engine(0)
    tally = {}
    tot = {"lsp": 0, "trcbk": 0, "trcbk_cut": 0}
    for f in sorted(glob.glob(os.path.join(DIR, "*.spdg"))):
        fx = spdg.load(f)
        for alg, simd, sel in ((0, 0, 1), (2, 2, 0)):
            want = cpu_status(fx, alg, simd)
            sc, sp, p, j, n, lowest, wl = inputs(fx, alg)
            sc.scalar_engines = sel
            res = eng.align_s_seeded(sc, sp, p._owner, [j if n else None], [lowest], [wl], allow_partial=True)
            st = eng.seeded_stats()
            for k in tot:
                tot[k] += st[k]
            scr, skl = res[0]
            flat = [int(x) for x in skl.ravel()] if len(skl) else []
            if want == "ok":
                good = scr == int(fx[f"seed_scr_A{alg}"][0]) and flat == fx[f"seed_skl_A{alg}"].tolist()
                key = "ok" if good else "MISMATCH"
                if not good:
                    print(f"{os.path.basename(f)} A{alg}: score {scr} vs {int(fx[f'seed_scr_A{alg}'][0])}")
            else:                                            # not defined / not served: the product must not invent an alignment
                good = scr == abi.NEVSEL and not flat
                key = want if good else "MISMATCH-" + want
                if not good:
                    print(f"{os.path.basename(f)} A{alg}: {want} on the CPU, but the GPU path returned score {scr}")
            tally[key] = tally.get(key, 0) + 1
    print(tally, tot)
    eng.close()
    return 1 if any(k.startswith("MISMATCH") for k in tally) else 0


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(int(sys.argv[2]), int(sys.argv[3]))
    else:
        sys.exit(run())
