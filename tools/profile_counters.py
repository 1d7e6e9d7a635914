#!/usr/bin/env python3
"""Per-kernel counters of one bench.py workload, in the form bench.py reads at run time (profiles/rNN_counters.json).

Run on the GPU box (rocprofv3), once per workload and round:

    python tools/profile_counters.py --tag r05 --name c2 -- --workload c2
    python tools/profile_counters.py --tag r05 --name a0 -- --engines a0 --queries 1000

Three passes of `python bench.py <args> --plain --steps S --warmup 0 --legs none --seeded-pairs 0` (--plain: nothing
but S aligns of the batch run, so totals divide by S): rocprofv3 --kernel-trace --stats, --pmc SQ_INSTS_VALU
SQ_INSTS_SALU, --pmc FETCH_SIZE, --pmc WRITE_SIZE (each counter set alone, as MI355X_MICROARCH.md asks).  Per kernel of
the workload the entry records: launches per step, mean duration, VALU / SALU wave-instructions per launch, HBM bytes
per launch (FETCH_SIZE counts 64-byte units of 128-byte requests: x 2, an upper bound; WRITE_SIZE as it is; both KiB),
the DP cells one step hands the kernel (from the bench line of the same run) and the sha256 of the kernel's source
file -- bench.py and tests/test_profile_counters.py compare that with the tree, so a kernel that changed without a new
profile is reported instead of priced with stale figures.
"""
import argparse
import collections
import csv
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernels whose counters bench.py prices: name prefix of the demangled symbol -> (key, source file, which cells of the bench line)
KERNELS = [
    ("void spdp_sweep_fp<2,", "udh", "spaln_amd/csrc/spdp_sweep_fp.hip", "udh_cells"),
    ("void spdp_sweep_fp<1,", "forward", "spaln_amd/csrc/spdp_sweep_fp.hip", "fwd_cells"),
    ("void spdh_sweep<", "h", "spaln_amd/csrc/spdp_h_kernels.hip", "fwd_cells"),
    ("void spdp_rowwave_udh<", "a0_udh", "spaln_amd/csrc/spdp_rowwave.hip", "udh_cells"),
    ("void spdp_rowwave<1,", "a0_fwd", "spaln_amd/csrc/spdp_rowwave.hip", "fwd_cells"),
    ("void spdp_exact<2", "a1_udh", "spaln_amd/csrc/spdp_exact.hip", "udh_cells"),
    ("void spdp_exact<1", "a1_fwd", "spaln_amd/csrc/spdp_exact.hip", "fwd_cells"),
    ("void spdh_rowwave<1,", "h_a0_fwd", "spaln_amd/csrc/spdp_h_rowwave.hip", "fwd_cells"),
    ("void spdh_rowwave<2,", "h_a0_udh", "spaln_amd/csrc/spdp_h_rowwave.hip", "udh_cells"),
    ("void spdh_exact<", "h_a1", "spaln_amd/csrc/spdp_h_exact.hip", "fwd_cells"),
    ("(anonymous namespace)::spdp_blk_vote_wave", "blk", "spaln_amd/csrc/spdp_blk_vote.hip", None),
]


def source_sha(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_totals(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    if not os.path.exists(path):
        return acc, calls
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    return acc, calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--name", required=True, help="key of this workload in the json (c2, c4, c3, a0, ...)")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("bench_args", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    bargs = [a for a in args.bench_args if a != "--"]
    out_dir = os.path.join(ROOT, "gpurun_out", f"counters_{args.tag}_{args.name}")
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    base = [sys.executable, os.path.join(ROOT, "bench.py")] + bargs + ["--plain", "--steps", str(args.steps), "--warmup", "0",
                                                                      "--legs", "none", "--seeded-pairs", "0", "--cpu-sample", "8"]
    passes = {"kt": ["--kernel-trace", "--stats"], "valu": ["--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
              "fetch": ["--pmc", "FETCH_SIZE"], "write": ["--pmc", "WRITE_SIZE"]}
    line = None
    for name, flags in passes.items():
        d = os.path.join(out_dir, name)
        cmd = ["rocprofv3"] + flags + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + base
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if js:
            line = json.loads(js[-1])
        else:
            sys.stderr.write(f"pass {name}: no bench line (rc {r.returncode})\n{r.stderr[-400:]}\n")
    if line is None:
        raise SystemExit("no pass produced a bench line")
    cfg = line["config"]
    # durations from the kernel trace
    dur = collections.defaultdict(list)
    kt = os.path.join(out_dir, "kt", "p_kernel_trace.csv")
    if os.path.exists(kt):
        with open(kt) as fh:
            for row in csv.DictReader(fh):
                dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    valu, vcalls = pmc_totals(os.path.join(out_dir, "valu", "p_counter_collection.csv"))
    fetch, _ = pmc_totals(os.path.join(out_dir, "fetch", "p_counter_collection.csv"))
    write, _ = pmc_totals(os.path.join(out_dir, "write", "p_counter_collection.csv"))
    entry = {"command": "python bench.py " + " ".join(bargs) + f" --plain --steps {args.steps} --warmup 0 --legs none",
             "steps": args.steps, "bench_value": line["value"], "kernels": {}}
    for prefix, key, src, cells_key in KERNELS:
        names = [k for k in dur if k.startswith(prefix)] or [k for k in valu if k.startswith(prefix)]
        if not names:
            continue
        # several instantiations of one template may run in a step: the one with the most time is the kernel priced
        names.sort(key=lambda k: -sum(dur.get(k, [0.0])))
        k = names[0]
        n_launch = len(dur[k]) if dur.get(k) else len(vcalls.get(k, ()))
        if not n_launch:
            continue
        cells = cfg.get(cells_key) if cells_key else None
        e = {"kernel": k.replace("(anonymous namespace)::", "").split("(")[0], "source": src, "source_sha256": source_sha(src),
             "launches_per_step": n_launch / args.steps,
             "avg_ms": round(sum(dur[k]) / len(dur[k]), 4) if dur.get(k) else None,
             "ms_per_step": round(sum(dur[k]) / args.steps, 4) if dur.get(k) else None,
             "cells_per_step": cells, "queries": cfg.get("queries_per_gpu")}
        if k in valu:
            nl = max(1, len(vcalls[k]))
            e["valu_per_launch"] = valu[k].get("SQ_INSTS_VALU", 0.0) / nl
            e["salu_per_launch"] = valu[k].get("SQ_INSTS_SALU", 0.0) / nl
            if cells:
                e["valu_per_cell"] = valu[k].get("SQ_INSTS_VALU", 0.0) / args.steps / cells
        if k in fetch and k in write:
            nl = max(1, n_launch)
            e["fetch_kib_per_launch"] = fetch[k].get("FETCH_SIZE", 0.0) / nl
            e["write_kib_per_launch"] = write[k].get("WRITE_SIZE", 0.0) / nl
            e["hbm_bytes_per_launch"] = int((2 * fetch[k].get("FETCH_SIZE", 0.0) + write[k].get("WRITE_SIZE", 0.0)) * 1024 / nl)
        entry["kernels"][key] = e
    path = os.path.join(ROOT, "gpurun_out", f"{args.tag}_counters.json")
    allc = json.load(open(path)) if os.path.exists(path) else {}
    allc[args.name] = entry
    with open(path, "w") as f:
        json.dump(allc, f, indent=1, sort_keys=True)
    # the --stats summary of the same command, for profiles/
    st = os.path.join(out_dir, "kt", "p_kernel_stats.csv")
    if os.path.exists(st):
        with open(os.path.join(ROOT, "gpurun_out", f"{args.tag}_{args.name}_kernel_stats.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- {entry['command']}\n")
            f.write(open(st).read())
            f.write(json.dumps(line) + "\n")
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
