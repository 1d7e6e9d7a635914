#!/usr/bin/env python3
"""The drop-in under the reference's own calling pattern.  On an MI355X box:

    python tools/dropin_demo.py [--queries 2000] [--genes 200] [--threads 16] [--gpu-threads 64] [--modes Q7,Q4]

A synthetic genome (planted multi-exon genes between random spacers) is formatted by the compiled reference's own
`spaln -W -KD`; cDNA queries (full transcripts of the planted genes, 2 % substitutions, 0.2 % indels) are mapped and
aligned twice:

  * oracle/_ref/spaln      -Q<n> -S1 -O4 -t<threads>      -dgnm q.fa   the reference, CPU
  * oracle/_ref/spaln_gpu  -Q<n> -S1 -O4 -t<gpu-threads>  -dgnm q.fa   the SAME program with alignS_ng switched to
                                                                       libspdp_hip.so (oracle/ref_build/spaln_gpu_shim.cc)

and the two outputs are compared (records sorted: the order worker threads print in is theirs).  One JSON line with the
md5s, wall times and what the shim counted goes to stdout.  TEST INFRASTRUCTURE: uses oracle/_ref (the prebuilt files)."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaln_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def records(text):
    """-O4 output as a sorted list of per-query blocks (the exon rows of a query, closed by its '@' line)"""
    blocks, cur = [], []
    for line in text.splitlines():
        if line.startswith("#"):                       # the column header, printed once by whichever thread is first
            continue
        cur.append(line)
        if line.startswith("@"):
            blocks.append("\n".join(cur)); cur = []
    if cur:
        blocks.append("\n".join(cur))
    return sorted(blocks)


def make_dataset(td, args):
    """the synthetic genome (formatted by the reference's own `spaln -W`) and the queries, in directory td; returns
    (genome length, environment for the two programs)"""
    rng = np.random.default_rng(synth.SEED + 8800)
    if args.protein:
        genes = [synth.make_protein_gene(np.random.default_rng(synth.SEED + 8801 + i), n_exons=6, flank=1000) for i in range(args.genes)]
    else:
        genes = [synth.make_gene(np.random.default_rng(synth.SEED + 8801 + i)) for i in range(args.genes)]
    n_chr = 4
    per = args.genes // n_chr
    tot = 0
    with open(os.path.join(td, "gnm.mfa"), "w") as f:
        for c in range(n_chr):
            parts = []
            for g in genes[c * per:(c + 1) * per]:
                spacer = int(getattr(args, "spacer", 0) or 0)                     # (--spacer: a larger genome around the same genes)
                parts += [synth.random_dna(rng, int(rng.integers(3000, 20000)) if not spacer else int(rng.integers(spacer // 2, spacer))), g.window]
            s = bytes(np.concatenate(parts)).decode()
            tot += len(s)
            f.write(f">chr{c + 1}\n")
            f.writelines(s[i:i + 60] + "\n" for i in range(0, len(s), 60))
    with open(os.path.join(td, "q.fa"), "w") as f:
        for i in range(args.queries):
            g = genes[int(rng.integers(0, per * n_chr))]
            if args.protein:                                  # the planted protein with another 5 % of its residues replaced
                q = g.query.copy()
                hit = rng.random(q.size) < 0.05
                q[hit] = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)[rng.integers(0, 20, size=int(hit.sum()))]
            else:
                q = synth.mutate(rng, g.query, 0.02, 0.002)
                frag = int(getattr(args, "frag", 0) or 0)                         # (--frag: ESTs, fragments of the transcripts)
                if frag and len(q) > frag:
                    at = int(rng.integers(0, len(q) - frag))
                    q = q[at:at + frag]
                if getattr(args, "antisense", False) and i % 2:
                    q = np.frombuffer(bytes(q).translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1], dtype=np.uint8)
            f.write(f">q{i}\n{bytes(q).decode()}\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "HSA_TOOLS", "LD_PRELOAD"))}
    env.update(ALN_TAB=os.path.join(REF, "table"), ALN_DBS=td)
    subprocess.run([os.path.join(REF, "spaln"), "-W", "-KP" if args.protein else "-KD", f"-t{args.threads}", "gnm.mfa"], cwd=td, env=env, check=True,
                   capture_output=True)
    return tot, env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--genes", type=int, default=200)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--gpu-threads", type=int, default=64)
    ap.add_argument("--modes", default="Q7,Q4")
    ap.add_argument("--strand", default="-S1", help="-S1: the queries as given; -S3 (spaln's default): both orientations")
    ap.add_argument("--antisense", action="store_true", help="every other query reverse-complemented (cDNA)")
    ap.add_argument("--spacer", type=int, default=0, help="longest random stretch between two genes (default 20 000)")
    ap.add_argument("--frag", type=int, default=0, help="queries are fragments of this length of the transcripts (ESTs)")
    ap.add_argument("--extra", default="", help="further options for both programs, e.g. -yl3 (double affine gaps)")
    ap.add_argument("--where", action="store_true", help="also time the reference's own aligner calls inside its program (Amdahl's bound for the drop-in)")
    ap.add_argument("--protein", action="store_true", help="protein queries (alignH_ng) against genes with ORFs instead of cDNAs")
    args = ap.parse_args()
    out = {"queries": args.queries, "genes": args.genes, "query_type": "protein" if args.protein else "cDNA", "extra": args.extra, "runs": []}
    with tempfile.TemporaryDirectory(prefix="spdp_dropin_") as td:
        out["genome_nt"], env = make_dataset(td, args)
        for mode in args.modes.split(","):
            run = {"mode": "-" + mode}
            res = {}
            for name, exe, thr in (("reference", "spaln", args.threads), ("gpu", "spaln_gpu", args.gpu_threads)):
                cmd = [os.path.join(REF, exe), "-" + mode] + ([] if args.protein else [args.strand]) + args.extra.split() + ["-O4", f"-t{thr}", "-dgnm", "q.fa"]
                t0 = time.perf_counter()
                r = subprocess.run(cmd, cwd=td, env=env, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                if name == "gpu" and os.environ.get("DROPIN_STDERR"):          # (tuning: what the library says under SPDP_SEED_VERBOSE)
                    open(os.environ["DROPIN_STDERR"], "a").write(r.stderr)
                if r.returncode != 0:
                    run[name] = {"error": r.stderr[-400:], "rc": r.returncode, "stdout_tail": r.stdout[-200:]}
                    continue
                rec = records(r.stdout)
                res[name] = rec
                run[name] = {"wall_s": round(dt, 3), "threads": thr, "aligned": sum(1 for b in rec if "\n@" in "\n" + b),
                             "md5_sorted_records": hashlib.md5("\n".join(rec).encode()).hexdigest(),
                             "md5_raw": hashlib.md5(r.stdout.encode()).hexdigest()}
                m = re.search(r"\[spaln_gpu\] alignS[^\n]*(\n\[spaln_gpu\] [^\n]*)*", r.stderr)
                if m:
                    run[name]["shim"] = m.group(0)
            if args.where:
                # where the reference's time goes: the same program once more with every aligner call timed (no device involved)
                cmd = [os.path.join(REF, "spaln_gpu"), "-" + mode] + ([] if args.protein else [args.strand]) + args.extra.split() + ["-O4", f"-t{args.threads}", "-dgnm", "q.fa"]
                t0 = time.perf_counter()
                r = subprocess.run(cmd, cwd=td, env=dict(env, SPALN_GPU_TIME_REF="1"), capture_output=True, text=True)
                dt = time.perf_counter() - t0
                m = re.search(r"time-ref mode: (\d+) aligner calls, ([0-9.]+) s inside .* program wall ([0-9.]+) s", r.stderr)
                if m:
                    calls, inside, wall = int(m.group(1)), float(m.group(2)), float(m.group(3))
                    run["where_the_reference_spends_its_time"] = {
                        "wall_s": round(dt, 3), "threads": args.threads, "aligner_calls": calls,
                        "aligner_thread_seconds": inside, "worker_thread_seconds": round(wall * args.threads, 3),
                        "aligner_share_of_worker_time": round(inside / (wall * args.threads), 4),
                        "note": "alignS_ng / alignH_ng timed inside the reference's own program (SPALN_GPU_TIME_REF=1: the shim forwards "
                                "every call to the reference's aligner); the rest is block search, sequence IO, Exinon, rescoring, output"}
            if "reference" in res and "gpu" in res:
                a, b = res["reference"], res["gpu"]
                run["identical"] = a == b
                run["records_differing"] = len(set(a) ^ set(b)) // 2 if a != b else 0
                if a != b:
                    d = sorted(set(a) - set(b))[:1] + sorted(set(b) - set(a))[:1]
                    run["first_difference"] = [x[:600] for x in d]
                run["gpu_over_reference_wall"] = round(run["reference"]["wall_s"] / run["gpu"]["wall_s"], 3)
            out["runs"].append(run)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
