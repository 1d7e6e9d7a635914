#!/usr/bin/env python3
"""Golden block-index files for the index builder (MakeBlk, `spaln -W`).

Build container only (needs oracle/_ref/spaln, `make -C oracle/ref_build`).  Genomes made here, deterministically, are
formatted by the compiled reference itself and its index file -- data: the tables the reference wrote -- is kept gzip'ed:

  idx_k1_t4    the genome of blk_k1 (tests/golden/make_blk_goldens.py), `spaln -W -KD -t4`: the threaded block walk
  idx_k3_t4    the genome of blk_k3, five bit patterns (-XC5), -t4
  idx_edge_t0  chromosomes whose lengths sit on the block boundaries (shorter than a block, blklen + margin exactly and one
  idx_edge_t4  less / more, a multiple of blklen), runs of N and single N (an ambiguous residue under a '0' of a spaced
               pattern), five bit patterns; the serial walk and the threaded one

`edge_genome()` is imported by the tests to make the same residues again."""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref")


def edge_genome():
    """chromosomes as ASCII arrays; for a FASTA file of this size setupbitpat picks blklen 1024 and 6-mers, whose -XC5
    patterns are 10 wide: margin 9"""
    rng = np.random.default_rng(20260930)
    dna = lambda n: np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    lens = [40000, 1024 + 9, 1024 + 8, 1024 + 10, 2 * 1024 + 9, 3 * 1024, 10, 1, 5 * 1024 + 9 + 512, 30000]
    chroms = []
    for k, n in enumerate(lens):
        s = dna(n)
        if k in (0, 8, 9):
            for _ in range(12):                                  # single ambiguous residues
                s[int(rng.integers(0, n))] = ord("N")
            for _ in range(4):                                   # runs, one across a block boundary
                at = int(rng.integers(0, n - 60))
                s[at:at + int(rng.integers(2, 50))] = ord("N")
            s[1020:1040] = ord("N")
            s[2048 + 5] = ord("R")                               # another IUPAC letter: ambiguous as well
        chroms.append(s)
    return chroms


def edge_genome_p():
    """for the translated index (-KP): a FASTA file of this size gives 3-residue words and blklen 1024; the margin is
    3 k - 1 + MinOrf = 38, so a first block ends after 1062 residues"""
    rng = np.random.default_rng(20260931)
    dna = lambda n: np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    lens = [1, 2, 5, 37, 38, 39, 40, 1023, 1024, 1025, 1061, 1062, 1063, 1064, 2048, 2085, 2086, 2087, 2088, 3000, 3110, 3111, 5000]
    chroms = []
    for k, n in enumerate(lens):
        s = dna(n)
        if n >= 1000:
            at = int(rng.integers(100, n - 50))
            s[at:at + int(rng.integers(1, 9))] = ord("N")
            s[int(rng.integers(100, n - 50))] = ord("R")
        if k % 5 == 0 and n > 2000:
            s[1020:1030] = ord("N")
        chroms.append(s)
    return chroms


def genome_p(seed, lens):
    rng = np.random.default_rng(seed)
    chroms = []
    for n in lens:
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
        for _ in range(3):
            at = int(rng.integers(0, n - 40))
            s[at:at + int(rng.integers(1, 30))] = ord("N")
        chroms.append(s)
    return chroms


P_CASES = {                                                     # name -> (genome, options of `spaln -W -KP`, threaded, MinOrf)
    "idxp_edge_t0": (edge_genome_p, ["-t0"], 0, 30),
    "idxp_edge_t3": (edge_genome_p, ["-t3"], 1, 30),
    "idxp_a12_t2": (lambda: genome_p(77, [9000, 14000]), ["-XA12", "-Xk3", "-Xs2", "-Xr21", "-Xb512", "-t2"], 1, 21),
}


def main_p():
    """the translated index of `spaln -W -KP` (needs oracle/_ref/spaln_idxtap, which also prints MakeBlk::prepacomp's terms);
    tests/golden/idxp_acomp.json holds those per alphabet size"""
    import json
    env = dict(os.environ, ALN_TAB=os.path.join(REF, "table"))
    acomp = {}
    for name, (make, opts, threaded, minorf) in P_CASES.items():
        with tempfile.TemporaryDirectory() as td:
            write_fasta(os.path.join(td, "gnm.mfa"), make())
            r = subprocess.run([os.path.join(REF, "spaln_idxtap"), "-W", "-KP"] + opts + ["gnm.mfa"], cwd=td, env=dict(env, ALN_DBS=td),
                               capture_output=True, text=True)
            if r.returncode:
                sys.exit(f"{name}: {r.stderr[-300:]}")
            line = [l for l in r.stderr.splitlines() if l.startswith("[idx_tap] acomp")][0].split()
            acomp[line[2]] = line[4:]
            raw = bytearray(open(os.path.join(td, "gnm.bkp"), "rb").read())
            raw[36 + 48:36 + 88] = bytes(40)                      # ContBlk's five pointers: whatever the writer's heap was
            with gzip.GzipFile(os.path.join(OUT, name + ".bkp.gz"), "wb", mtime=0) as f:
                f.write(bytes(raw))
            print(name, os.path.getsize(os.path.join(td, "gnm.mfa")), "bytes of FASTA ->", len(raw), "bytes of index")
    with open(os.path.join(OUT, "idxp_acomp.json"), "w") as f:
        json.dump(acomp, f, indent=0)


def write_fasta(path, chroms):
    with open(path, "w") as f:
        for c, s in enumerate(chroms):
            f.write(f">chr{c + 1}\n")
            t = bytes(s).decode()
            f.writelines(t[i:i + 60] + "\n" for i in range(0, len(t), 60))


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_blk_goldens as mb
    env = dict(os.environ, ALN_TAB=os.path.join(REF, "table"))
    k1, _ = mb.genome_and_queries(42, 2, 900)
    k3, _ = mb.genome_and_queries(28, 2, 950)
    for name, chroms, opts in (("idx_k1_t4", k1, ["-t4"]), ("idx_k3_t4", k3, ["-XC5", "-t4"]),
                               ("idx_edge_t0", edge_genome(), ["-XC5"]), ("idx_edge_t4", edge_genome(), ["-XC5", "-t4"])):
        with tempfile.TemporaryDirectory() as td:
            write_fasta(os.path.join(td, "gnm.mfa"), chroms)
            r = subprocess.run([os.path.join(REF, "spaln"), "-W", "-KD"] + opts + ["gnm.mfa"], cwd=td, env=dict(env, ALN_DBS=td),
                               capture_output=True, text=True)
            if r.returncode:
                sys.exit(f"{name}: {r.stderr[-300:]}")
            raw = bytearray(open(os.path.join(td, "gnm.bkn"), "rb").read())
            raw[36 + 48:36 + 88] = bytes(40)                      # ContBlk's five pointers: whatever the writer's heap was
            with gzip.GzipFile(os.path.join(OUT, name + ".bkn.gz"), "wb", mtime=0) as f:
                f.write(bytes(raw))
            print(name, os.path.getsize(os.path.join(td, "gnm.mfa")), "bytes of FASTA ->", len(raw), "bytes of index;", r.stderr.strip().splitlines()[-1])


if __name__ == "__main__":
    if sys.argv[1:] == ["p"]:
        main_p()
    else:
        main()
