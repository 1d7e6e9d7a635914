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


def make_case(seed):
    """(window, query, harness options, description)"""
    rng = np.random.default_rng(synth.SEED + 40000 + seed)
    kind = seed % 10
    n_exons = int(rng.integers(2, 9))
    mrna = int(rng.integers(200, 1400))
    sub = float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.15, 0.2]))
    indel = float(rng.choice([0.0, 0.002, 0.01, 0.03]))
    exon_min = int(rng.choice([5, 12, 30]))
    g = synth.make_gene(rng, n_exons=n_exons, mrna_len=mrna, flank=int(rng.integers(100, 1200)),
                        intron_hi=int(rng.choice([300, 1500, 6000])), sub=sub, indel=indel, exon_min=exon_min)
    w, q = g.window, g.query
    opts = ["-Q", str(int(rng.integers(1, 4)))]
    if rng.random() < 0.4:
        opts += ["-X", "0"]
    if rng.random() < 0.1:
        opts += ["-C"]
    desc = f"ex{n_exons} m{mrna} sub{sub} indel{indel} emin{exon_min}"
    if kind == 1:                                   # poly-A tail and junk head on the query
        q = np.concatenate([synth.random_dna(rng, int(rng.integers(3, 40))), q, np.frombuffer(b"A" * int(rng.integers(5, 40)), dtype=np.uint8)])
        desc += " junk+polyA"
    elif kind == 2:                                 # window cut inside the gene: the query overhangs
        e = g.exons
        lo = e[0][0] + int(rng.integers(5, max(6, e[0][1] - e[0][0] - 5))) if rng.random() < 0.7 else 0
        hi = e[-1][1] - int(rng.integers(5, max(6, e[-1][1] - e[-1][0] - 5))) if rng.random() < 0.7 else len(w)
        w = w[lo:hi]
        desc += " cut"
    elif kind == 3:
        opts.append("-L")
        desc += " local"
    elif kind == 4:                                 # a block of the query replaced by noise (HSP desert)
        a0 = int(rng.integers(0, max(1, len(q) - 60)))
        ln = int(rng.integers(20, min(400, len(q) - a0)))
        q = q.copy()
        q[a0:a0 + ln] = synth.random_dna(rng, ln)
        desc += f" noise{ln}"
    elif kind == 5:                                 # an exon missing from the query / duplicated piece
        k = int(rng.integers(0, n_exons))
        lens = [b - a for a, b in g.exons]
        off = sum(lens[:k])
        q = np.concatenate([g.transcript[:off], g.transcript[off + lens[k]:]])
        q = synth.mutate(rng, q, sub, indel)
        desc += f" skip_exon{k}"
    elif kind == 6:                                 # genomic insertion / deletion inside an exon of the window
        k = int(rng.integers(0, n_exons))
        a, b = g.exons[k]
        at = a + (b - a) // 2
        if rng.random() < 0.5:
            w = np.concatenate([w[:at], synth.random_dna(rng, int(rng.integers(1, 30))), w[at:]])
        else:
            w = np.concatenate([w[:at], w[at + int(rng.integers(1, min(30, b - at))):]])
        desc += " exon_indel"
    elif kind == 7:                                 # small MaxVmfSpace: DP calls take the linear-space branch
        opts += ["-V", str(int(rng.choice([20000, 100000, 400000])))]
        desc += " smallV"
    elif kind == 8:                                 # unrelated pair
        q = synth.random_dna(rng, int(rng.integers(60, 400)))
        desc += " random"
    elif kind == 9:                                 # a tandem copy of the locus: several HSP units
        w = np.concatenate([w, w[len(w) // 3:]])
        desc += " tandem"
    return w, q, opts, desc


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
