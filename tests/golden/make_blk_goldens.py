#!/usr/bin/env python3
"""Fixtures of the block search from the REAL reference: tests/golden/blk_k1.spdg, blk_k3.spdg, blk_par.spdg
(nucleotide queries, `spaln -W -KD`) and blk_p1.spdg / blk_p1.bkp (protein queries, `spaln -W -KP`: protein_genome_and_queries).

Build container only (needs oracle/_ref/spaln and oracle/_ref/spaln_blktap, `make -C oracle/ref_build`).  A small synthetic
genome (planted genes of spaln_amd.synth between random spacers) is formatted by the compiled reference itself
(`spaln -W -KD`: its own .bkn index, once with the default contiguous k-mer, once with five spaced patterns, -XC5), then
the reference's CLI with the recorder of oracle/ref_build/blk_tap.cc maps a mixed query set onto it (-Q7): transcripts,
reverse complements, 500-nt fragments, short fragments, random sequences, chimeras, diverged copies.  The fixture holds
the index arrays and parameters as the reference's SrchBlk object held them, every query as findblock saw it, the state of
the vote at each TestOutput call and the block pairs handed to FindHsp.  Data only.

    python tests/golden/make_blk_goldens.py [name ...]
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spaln_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.dirname(os.path.abspath(__file__))
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


def revcomp(s):
    return COMP[np.asarray(s, dtype=np.uint8)[::-1]]


def genome_and_queries(n_genes, n_chr, seed):
    rng = np.random.default_rng(synth.SEED + seed)
    genes = [synth.make_gene(np.random.default_rng(synth.SEED + seed + 1 + i), intron_hi=3000) for i in range(n_genes)]
    per = n_genes // n_chr
    chroms = []
    for c in range(n_chr):
        parts = []
        for g in genes[c * per:(c + 1) * per]:
            parts += [synth.random_dna(rng, int(rng.integers(500, 3000))), g.window]
        chroms.append(np.concatenate(parts))
    queries = []
    for i, g in enumerate(genes):
        s, m = g.query, i % 7
        if m == 0:
            queries.append(s)
        elif m == 1:
            queries.append(revcomp(s))
        elif m == 2:
            a = int(rng.integers(0, len(s) - 500))
            queries.append(s[a:a + 500])
        elif m == 3:
            a = int(rng.integers(0, len(s) - 130))
            queries.append(s[a:a + int(rng.integers(40, 120))])
        elif m == 4:
            queries.append(synth.random_dna(rng, int(rng.integers(200, 1200))))
        elif m == 5:
            o = genes[(i * 7 + 3) % n_genes].query
            queries.append(np.concatenate([s[:700], revcomp(o)[:600]]))
        else:
            queries.append(synth.mutate(rng, s, 0.12, 0.01))
    return chroms, queries


def paralog_genome_and_queries(n_genes, n_chr, seed):
    """every gene twice (the second copy diverged by 4 - 10 %, sometimes on the other strand, sometimes next to the first):
    several candidate loci per query -- FindHsp's overlap / order / pruning rules, critjscr, MaxOut > 1 (run with -M4)"""
    rng = np.random.default_rng(synth.SEED + seed)
    genes = [synth.make_gene(np.random.default_rng(synth.SEED + seed + 1 + i), intron_hi=2000) for i in range(n_genes)]
    per = n_genes // n_chr
    chroms = []
    for c in range(n_chr):
        parts = []
        for k, g in enumerate(genes[c * per:(c + 1) * per]):
            copy = synth.mutate(rng, g.window, float(rng.choice([0.04, 0.07, 0.1])), 0.002)
            if k % 3 == 1:
                copy = revcomp(copy)
            parts += [synth.random_dna(rng, int(rng.integers(500, 3000))), g.window]
            parts += [synth.random_dna(rng, int(rng.integers(300, 1500) if k % 2 else rng.integers(4000, 9000))), copy]
        chroms.append(np.concatenate(parts))
    queries = []
    for i, g in enumerate(genes):
        s = g.query
        queries.append([s, revcomp(s), s[200:1100], synth.mutate(rng, s, 0.05, 0.005)][i % 4])
    return chroms, queries


def protein_genome_and_queries(n_genes, n_chr, seed):
    """blk_p1: protein queries against the translated index (`spaln -W -KP`, amino-acid words of the six frames; SrchBlk's
    DvsP = 1 branch: the candidate region as tron codes, the retry with a grown region).  Genes on both strands, every third
    one with a diverged second copy (-M4: two loci); queries: the diverged protein, the exact one, short pieces (below
    shortquery), random sequences, chimeras"""
    rng = np.random.default_rng(synth.SEED + seed)
    genes = [synth.make_protein_gene(np.random.default_rng(synth.SEED + seed + 1 + i), n_exons=5, aa_len=int(rng.integers(150, 500)))
             for i in range(n_genes)]
    per = n_genes // n_chr
    chroms = []
    for c in range(n_chr):
        parts = []
        for k, g in enumerate(genes[c * per:(c + 1) * per]):
            parts += [synth.random_dna(rng, int(rng.integers(500, 3000))), g.window if k % 3 else revcomp(g.window)]
            if k % 3 == 2:
                copy = synth.mutate(rng, g.window, 0.05, 0.0)
                parts += [synth.random_dna(rng, int(rng.integers(300, 6000))), copy if k % 2 else revcomp(copy)]
        chroms.append(np.concatenate(parts))
    queries = []
    for i, g in enumerate(genes):
        s, m = g.query, i % 5
        if m == 0:
            queries.append(s)
        elif m == 1:
            queries.append(g.protein)
        elif m == 2:
            a = int(rng.integers(0, len(s) - 90))
            queries.append(s[a:a + int(rng.integers(40, 90))])
        elif m == 3:
            queries.append(synth._AA_LETTERS[rng.integers(0, 20, size=int(rng.integers(100, 400)))])
        else:
            o = genes[(i * 7 + 3) % n_genes].query
            queries.append(np.concatenate([s[:120], o[:100]]))
    return chroms, queries


def main():
    env = dict(os.environ, ALN_TAB=os.path.join(REF, "table"))
    only = sys.argv[1:]
    for name, fmt_opts, n_genes, seed in (("blk_k1", [], 42, 900), ("blk_k3", ["-XC5"], 28, 950), ("blk_par", [], 24, 980),
                                          ("blk_p1", [], 30, 1200)):
        if only and name not in only:
            continue
        par = name in ("blk_par", "blk_p1")
        prot = name == "blk_p1"
        chroms, queries = (protein_genome_and_queries if prot else paralog_genome_and_queries if par else genome_and_queries)(n_genes, 2, seed)
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "gnm.mfa"), "w") as f:
                for c, s in enumerate(chroms):
                    f.write(f">chr{c + 1}\n")
                    t = bytes(s).decode()
                    f.writelines(t[i:i + 60] + "\n" for i in range(0, len(t), 60))
            with open(os.path.join(td, "q.fa"), "w") as f:
                for i, s in enumerate(queries):
                    f.write(f">q{i}\n{bytes(s).decode()}\n")
            e = dict(env, ALN_DBS=td)
            subprocess.run([os.path.join(REF, "spaln"), "-W", "-KP" if prot else "-KD"] + fmt_opts + ["gnm.mfa"], cwd=td, env=e, check=True,
                           capture_output=True)
            log = os.path.join(td, "log.spdg")
            r = subprocess.run([os.path.join(REF, "spaln_blktap"), "-Q7", "-O4", "-t1"] + (["-M4"] if par else []) + ["-dgnm", "q.fa"], cwd=td,
                               env=dict(e, SPDP_BLK_LOG=log), capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(log):
                sys.exit(f"{name}: reference run failed: {r.stderr[-300:]}")
            # ConvTab[0], [1] (the nil / unknown codes, which no sequence position holds) are never written by the reference:
            # whatever the heap held is replaced by "not a residue", so that regenerating gives the same file
            from tests import spdg
            fx = spdg.load(log)
            fx["blk_convtab"][:2] = 255
            spdg.save(os.path.join(OUT, name + ".spdg"), {k: v for k, v in fx.items() if k != "prm"})
            if prot:
                shutil.copyfile(os.path.join(td, "gnm.bkp"), os.path.join(OUT, name + ".bkp"))
            elif not par:
                shutil.copyfile(os.path.join(td, "gnm.bkn"), os.path.join(OUT, name + ".bkn"))     # the reference's own index file: an input of the reader's test
            print(f"{name}: genome {sum(len(c) for c in chroms)} nt, {len(queries)} queries, "
                  f"{os.path.getsize(log) / 1e6:.2f} MB, {r.stdout.count(chr(10) + '@')} aligned")
            if name == "blk_k3":
                grow_fixture(td, e, os.path.join(OUT, name + ".spdg"))


def grow_fixture(td, env, index_fixture):
    """blk_k3_grow.spdg: queries on which a position hash of the reference's queues GROWS (Dhash::resize) -- found with the
    oracle by random search (noisy fragments of the fixture's own queries; kept from the committed file when it exists) --
    each run through the reference in a process of its own (q_log: what a fresh worker does) and all in one process
    (q_log_batch: the grown tables persist from query to query there).  Index arrays: those of blk_k3.spdg."""
    import ctypes as C
    from oracle import blk, oracle
    from tests import spdg
    fx = spdg.load(index_fixture)
    ix, _keep = blk.index_of(fx)
    out = os.path.join(OUT, "blk_k3_grow.spdg")
    if os.path.exists(out):
        queries = [q["codes"] for q in blk.parse_log(dict(spdg.load(out), blk_prm=fx["blk_prm"]))]
    else:
        grows = C.c_int.in_dll(oracle.lib(), "spdp_oracle_blk_grows")
        rng = np.random.default_rng(7)
        pool = [q["codes"] for q in blk.parse_log(fx)]
        queries = []
        while len(queries) < 12:
            a = pool[int(rng.integers(len(pool)))]
            lo = int(rng.integers(0, max(1, len(a) - 40)))
            b = a[lo:lo + int(rng.integers(200, 900))].copy()
            hits = rng.random(b.size) < rng.choice([0.02, 0.2, 0.25])
            b[hits] = rng.choice(np.array([2, 3, 5, 9], dtype=np.uint8), size=int(hits.sum()))
            for stop in range(4):
                g0 = grows.value
                if blk.vote(ix, b, 0, len(b), stop) is None:
                    break
                if grows.value > g0:
                    queries.append(b)
                    break
    dec = {2: "A", 3: "C", 5: "G", 9: "T", 16: "N"}
    tap = [os.path.join(REF, "spaln_blktap"), "-Q7", "-O4", "-t1", "-dgnm"]
    logs = []
    for i, b in enumerate(queries):
        with open(os.path.join(td, "one.fa"), "w") as f:
            f.write(f">g{i}\n" + "".join(dec[int(x)] for x in b) + "\n")
        log = os.path.join(td, f"one_{i}.spdg")
        subprocess.run(tap + ["one.fa"], cwd=td, env=dict(env, SPDP_BLK_LOG=log), capture_output=True, check=True)
        logs.append(np.asarray(spdg.load(log)["q_log"], dtype=np.int32))
    with open(os.path.join(td, "all.fa"), "w") as f:
        for i, b in enumerate(queries):
            f.write(f">g{i}\n" + "".join(dec[int(x)] for x in b) + "\n")
    log = os.path.join(td, "all.spdg")
    subprocess.run(tap + ["all.fa"], cwd=td, env=dict(env, SPDP_BLK_LOG=log), capture_output=True, check=True)
    spdg.save(out, {"q_log": np.concatenate(logs), "q_log_batch": np.asarray(spdg.load(log)["q_log"], dtype=np.int32)})
    print(f"blk_k3_grow: {len(queries)} queries, {os.path.getsize(out) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
