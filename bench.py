#!/usr/bin/env python3
"""bench.py -- GCUPS of the spliced-alignment DP hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 10 000 synthetic 2 kb cDNAs, each
against its planted 8-exon locus +-1 kb of a synthetic genome (windows ~12 kb),
default band (alprm.sh = 100), Fwd2s1 `_wip` engine.  One "step" = one pass of
the hot path over the whole batch, inputs resident in HBM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            (spawns its N ranks itself, one process per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  N > 1: queries are sharded across ranks with no data-path collective
(SURVEY.md 8e); the ranks only meet at the timing barriers and the reduction of the timer, over gloo --
the path needs no RCCL.  Reported value: weak scaling (every rank aligns its own 10k queries); the same
line also carries the strong-scaling figure (ONE fixed 10k batch sharded with shard.shard_range) under
config.strong_scaling (--scaling strong swaps the two).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per DP cell: per 64 query rows x 1 column the
# sweep streams one 8 B column record and reads + writes one 8 B {H,F} boundary entry
BYTES_PER_CELL = {"score": 24.0 / 64.0, "udh": 40.0 / 64.0, "forward": 24.0 / 64.0 + 1.0}
# SURVEY.md 8(d)'s compulsory traffic, the figure roofline.achieved is computed from: per 64-row stripe one column costs
# 6 B of window (base, two signals, flags) + 16 B of boundary rows read and written = 0.34 B/cell; a traceback byte per cell
# on top for the forward flavour.  BYTES_PER_CELL above is what THIS layout moves at best (8 B column records, 16 B
# linear-space entries) and is reported beside it as roofline.layout_*
SURVEY_BYTES_PER_CELL = {"score": 22.0 / 64.0, "udh": 22.0 / 64.0, "forward": 22.0 / 64.0 + 1.0}
HBM_PEAK_GBS = 8000.0
# VALU issue ceiling of one MI355X, measured with tools/ubench (profiles/r02_valu_ubench.txt): a SIMD issues at most one
# VALU wave-instruction per ~2.2 shader cycles (fp32 add / int add / logic / cndmask class, or a max / compare with an
# fp32 add beside it; max / compare / DPP alone: one per 4), at the 2.3 GHz the sweeps sustain (GRBM_GUI_ACTIVE,
# profiles/r02_sq_counters.txt): 1024 SIMDs x 2.3e9 / 2.2.
VALU_PEAK_WINST_S = 1024 * 2.3e9 / 2.2
# What the kernels issue and move per DP cell / per launch is NOT a constant of this file: it is read from the round's
# counter profile (tools/profile_counters.py writes it on the GPU box: rocprofv3 --pmc SQ_INSTS_VALU, FETCH_SIZE, WRITE_SIZE
# in separate passes + --kernel-trace --stats, per workload).  Every entry carries the sha256 of its kernel's source file;
# an entry whose kernel has changed since is reported as stale (roofline.*.profile_stale, and tests/test_profile_counters.py
# fails) instead of pricing the new kernel with the old figures.
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r06_counters.json")


def _counters(workload, key):
    """the profile entry of kernel `key` in workload `workload`, or None; adds "stale": True when the kernel's source differs"""
    try:
        with open(COUNTERS_FILE) as f:
            allc = json.load(f)
        e = dict(allc[workload]["kernels"][key])
    except (OSError, ValueError, KeyError):
        return None
    import hashlib
    try:
        with open(os.path.join(ROOT, e["source"]), "rb") as f:
            e["stale"] = hashlib.sha256(f.read()).hexdigest() != e.get("source_sha256")
    except OSError:
        e["stale"] = True
    if e["stale"]:
        sys.stderr.write(f"bench.py: {e['source']} has changed since {os.path.relpath(COUNTERS_FILE, ROOT)} was taken "
                         f"({workload}/{key}): re-run tools/profile_counters.py\n")
    e["file"] = os.path.relpath(COUNTERS_FILE, ROOT)
    return e


def _cpu_align_one(item):
    """cpu_baseline worker: one query through the oracle ladder; returns DP cells of its engine calls."""
    from spaln_amd import abi, defaults
    from oracle import oracle, host_logic
    w, q, s5, s3 = item
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    p = ps.add(q, w, s5, s3)
    cells = [0]
    orig_fwd, orig_udh = oracle.wip_forward, oracle.wip_udh

    def fwd(sc_, p_, w_=None):
        w_ = w_ or oracle.stripe(p_, sc_.sh)
        cells[0] += oracle.cells(p_, w_)
        return orig_fwd(sc_, p_, w_)

    def udh(sc_, p_, n_im, w_=None):
        w_ = w_ or oracle.stripe(p_, sc_.sh)
        cells[0] += oracle.cells(p_, w_)
        return orig_udh(sc_, p_, n_im, w_)

    oracle.wip_forward, oracle.wip_udh = fwd, udh
    try:
        host_logic.align_s(sc, p)
    finally:
        oracle.wip_forward, oracle.wip_udh = orig_fwd, orig_udh
    return cells[0]


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "spaln")
REF_TAB = os.path.join(ROOT, "oracle", "_ref", "table")


def _host_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU
    boxes expose 256 hardware threads but grant the container 16 CPUs worth of time)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _ref_available():
    return os.path.exists(REF_BIN) and os.path.exists(os.path.join(REF_TAB, "mdm_mtx"))


def _ref_cli_chunk(chunk):
    """cpu_baseline worker, kind "reference": the compiled reference itself (oracle/_ref/spaln, built by
    oracle/ref_build/Makefile from the sources where they lie) on a chunk of (window, query) pairs, one
    CLI run per pair: -Q0 (block search off: the whole window goes through the DP ladder), -A2 (the
    `_wip` engines), -t1.  Returns (seconds spent inside the reference runs, runs that exited 0)."""
    import subprocess
    import tempfile
    from spaln_amd import synth
    busy, ok = 0.0, 0
    # a profiler wrapped around bench.py (rocprofv3) must not instrument the CPU baseline's processes
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "LD_PRELOAD", "ROCTRACER", "ROCTX"))}
    env["ALN_TAB"] = REF_TAB
    with tempfile.TemporaryDirectory() as td:
        for window_ascii, query_ascii, protein, alg in chunk:
            gf, qf = os.path.join(td, "g.fa"), os.path.join(td, "q.fa")
            synth.write_fasta(gf, "win", window_ascii)
            synth.write_fasta(qf, "qry", query_ascii)
            cmd = [REF_BIN, "-Q0", f"-A{alg}", "-pw", "-O4", "-t1"] + ([] if protein else ["-S1"]) + [gf, qf]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            busy += time.perf_counter() - t0
            ok += r.returncode == 0
    return busy, ok


REF_BENCH = os.path.join(ROOT, "oracle", "_ref", "ref_bench")


def _ref_bench_baseline(pairs, band_cells, cell_ratio, alg):
    """the compiled reference's engines over `pairs` in ONE process (oracle/_ref/ref_bench: parameter tables loaded once,
    one worker thread per host core taking pairs from a shared counter, per pair what `spaln -Q0 -A<alg>` runs); elapsed =
    the wall time of its parallel section"""
    import subprocess
    import tempfile
    from spaln_amd import synth
    ncores = _host_cores()
    used = min(ncores, len(pairs))
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "LD_PRELOAD", "ROCTRACER", "ROCTX"))}
    env["ALN_TAB"] = REF_TAB
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "list.txt"), "w") as lf:
            for i, (w, q) in enumerate(pairs):
                gf, qf = os.path.join(td, f"g{i}.fa"), os.path.join(td, f"q{i}.fa")
                synth.write_fasta(gf, "win", w)
                synth.write_fasta(qf, "qry", q)
                lf.write(f"{gf} {qf}\n")
        r = subprocess.run([REF_BENCH, "-A", str(alg), "-t", str(used), os.path.join(td, "list.txt")], env=env,
                           capture_output=True, text=True)
    if r.returncode != 0 or len(r.stdout.split()) != 3:
        return None
    done, ok, cdt = int(r.stdout.split()[0]), int(r.stdout.split()[1]), float(r.stdout.split()[2])
    return {"value": round(band_cells * cell_ratio / cdt / 1e9, 5), "unit": "GCUPS", "cores": used, "kind": "reference",
            "sample": f"first {len(pairs)} queries through the compiled reference's own engines in one process "
                      f"(oracle/_ref/ref_bench -A{alg} -t{used}: tables loaded once, per pair what spaln -Q0 runs -- Exinon, "
                      f"alignS_ng / alignH_ng, skl_rng*_ng; AVX2 build), {done} done, {ok} aligned, {cdt:.2f} s wall"}


SEED_BENCH = os.path.join(ROOT, "oracle", "_ref", "seed_bench")


def _seeded_leg(n_pairs):
    """SURVEY 8 row f2, measured: the seeded path (-Q7) of `n_pairs` C2-shaped pairs -- the reference's own alignS_ng on the
    host cores against ONE spdp_align_s_seeded call (the reference's geneorient() finds the HSPs for both, its Wilip answers
    the recursion levels through the callback), compared pair by pair.  oracle/_ref/seed_bench links the compiled
    reference and the library (oracle/ref_build/seed_bench.cc); None where it has not been built."""
    if not os.path.exists(SEED_BENCH):
        return None
    import subprocess
    import tempfile
    from spaln_amd import synth
    used = _host_cores()
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "LD_PRELOAD", "ROCTRACER", "ROCTX"))}
    env["ALN_TAB"] = REF_TAB
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "list.txt"), "w") as lf:
            for i in range(n_pairs):
                g = synth.make_gene(np.random.default_rng(synth.SEED + 20000 + i), sub=0.04 + 0.01 * (i % 5), indel=0.005)
                gf, qf = os.path.join(td, f"g{i}.fa"), os.path.join(td, f"q{i}.fa")
                synth.write_fasta(gf, "win", g.window)
                synth.write_fasta(qf, "qry", g.query)
                lf.write(f"{gf} {qf}\n")
        r = subprocess.run([SEED_BENCH, "-Q", "3", "-t", str(used), os.path.join(td, "list.txt")], env=env,
                           capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return None
    return {"pairs": d["walked"], "identical_to_reference": d["identical"], "compared": d["compared"],
            "library_pairs_per_s": round(d["walked"] / d["library_s"], 1), "library_s": d["library_s"],
            "library_first_call_s": d["library_cold_s"],
            "reference_pairs_per_s": round(d["walked"] / d["reference_s"], 1), "reference_threads": d["threads"],
            "marshalling_s": d["marshal_s"], "device_batches": d["batches"], "dp_requests": d["lsp"] + d["trcbk"],
            "note": "spaln -Q7 semantics on C2-shaped pairs (oracle/_ref/seed_bench): HSPs from the reference's geneorient() for both "
                    "sides (not timed); reference = its alignS_ng on the host cores; library = one spdp_align_s_seeded call, inputs "
                    "uploaded inside the call, the second call on a warm context (first call beside it); marshalling = the "
                    "reference-side shim turning Seq / Exinon into the plain arrays of include/spdp.h, outside both"}


def _ref_baseline(pairs, protein, band_cells, cell_ratio, alg=2):
    if os.path.exists(REF_BENCH):
        rb = _ref_bench_baseline(pairs, band_cells, cell_ratio, alg)
        if rb:
            return rb
    return _ref_cli_baseline(pairs, protein, band_cells, cell_ratio, alg)


def _ref_cli_baseline(pairs, protein, band_cells, cell_ratio, alg=2):
    """times the reference CLI on `pairs`, one worker per host core, each running its share of the
    pairs back to back; elapsed = the busiest worker's time inside the reference runs (pool start-up
    and FASTA writing excluded, the CLI's own start-up included); cells = band cells of the sample x
    (engine cells / band cells) of the GPU run (the reference runs the same ladder)"""
    import multiprocessing as mp
    ncores = _host_cores()
    used = min(ncores, len(pairs))
    chunks = [[(w, q, protein, alg) for w, q in pairs[c::used]] for c in range(used)]
    with mp.Pool(used) as pool:
        res = pool.map(_ref_cli_chunk, chunks)
    cdt = max(b for b, _ in res)
    ok = sum(k for _, k in res)
    return {"value": round(band_cells * cell_ratio / cdt / 1e9, 5), "unit": "GCUPS", "cores": used, "kind": "reference",
            "sample": f"first {len(pairs)} queries through the compiled reference (oracle/_ref/spaln -Q0 -A{alg} -t1, AVX2 build), "
                      f"{used} workers x {len(chunks[0])} runs back to back, {ok} ok; busiest worker {cdt:.2f} s, "
                      f"{sum(b for b, _ in res):.0f} core-seconds in total"}


def _cpu_align_h_one(item):
    """cpu_baseline worker of the aa x genome workload: one query through the oracle's alignH_ng"""
    from spaln_amd import abi, defaults, synth
    from oracle import oracle, host_logic_h
    q, sg = item
    sc = defaults.scoring_h()
    ps = abi.ProblemSetH()
    p = ps.add(synth.encode_protein(q), sg["b"], sg["sig5"], sg["sig3"], sg["sigS"], sg["sigT"], sg["sigE"],
               sg["phs5"], sg["phs3"])
    try:
        host_logic_h.align_h(sc, p)
    except (host_logic_h.ReferenceUndefined, host_logic_h.ReferenceFatal):
        pass
    return oracle.cells_h(p, oracle.stripe31(p, sc.sh))


def main_c3(args):
    """BASELINE configs[2] ("C3"), scaled to one GPU's memory: protein queries (400 aa) against their
    planted 6-exon loci +-1 kb, Fwd2h1 `_wip` path: alignH_ng = forwardH1_wip + traceback + stdskl3."""
    torch, dist, rank, world, local_rank, coll_dev = _dist_setup()
    from spaln_amd import abi, defaults, engine, synth
    eng = engine.Engine(local_rank)
    exact = args.engines != "wip"
    if exact:
        # -A0 / -A1: the reference's protein parameter set as its harness dumped it (intron penalty table, junction table,
        # frame-shift penalties: tests/golden/h1_400aa.spdg), forwardH_ng / hirschbergH_ng (spdp_h_rowwave.hip) or
        # forwardH1 / hirschbergH1 (spdp_h_exact.hip) instead of the `_wip` pair
        from tests import spdg
        fx = spdg.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "h1_400aa.spdg"))
        sc = spdg.scoring_h(fx, scalar_engines=1 if args.engines == "a0" else 2)
    else:
        sc = defaults.scoring_h()
    bseed = synth.SEED + 3000 + 1000 * rank
    batch = (synth.make_protein_batch(args.queries, seed=bseed) if args.queries <= 10000 else
             synth.make_chunked("c3", args.queries, seed=bseed, procs=_host_cores()))
    ps = abi.ProblemSetH()
    for g, sg in batch:
        kw = dict(dinc=synth.exact_inputs(defaults.encode(g.window))["dinc"]) if exact else {}
        ps.add(synth.encode_protein(g.query), sg["b"], sg["sig5"], sg["sig3"], sg["sigS"], sg["sigT"], sg["sigE"],
               sg["phs5"], sg["phs3"], **kw)
    bt = eng.upload_h(sc, ps)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        bt.align(want=True, convert=False)
    barrier()
    t0 = time.perf_counter()
    kms = []
    cells = 0
    for _ in range(args.steps):
        _, ms, cells = bt.align(want=True, convert=False)
        kms.append(ms)
    kcells = cells                  # what the launches of the step processed (the exact-model ladders: linear-space passes + slabs)
    if exact or not cells:
        cells = bt.cells()          # the exact-model legs are quoted on BAND cells (m x window band, as round 4's), not on engine cells
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([float(cells)], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = float(c.item())
    else:
        total_cells = float(cells)
    if rank == 0:
        k_ms = float(np.mean(kms))
        wl_key = {"wip": "c3", "a0": "c3_a0", "a1": "c3_a1"}[args.engines]              # entry of the round's counter profile
        k_key = {"wip": "h", "a0": "h_a0_fwd", "a1": "h_a1"}[args.engines]
        bpc = 32.0 / 64.0 + 2.0            # 16 B record + 8 B boundary read + 8 B write per 64 rows x 1 nt; 2 B code / cell
        achieved = cells * bpc / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        import multiprocessing as mp
        ncores = _host_cores()
        ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 2 * ncores, len(batch)))
        if world > 1:
            cpu_base = None                                   # timed at N = 1 only
        elif _ref_available() and not args.cpu_port:
            from oracle import oracle
            ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 64 * ncores, len(batch)))
            band = 0
            for i in range(ns):
                band += oracle.cells_h(ps.items[i], oracle.stripe31(ps.items[i], sc.sh))
            if exact:
                ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 8 * ncores, len(batch)))
            cpu_base = _ref_baseline([(batch[i][0].window, batch[i][0].query) for i in range(ns)], True, band,
                                     cells / max(1, bt.cells()), {"wip": 2, "a0": 0, "a1": 1}[args.engines])
        else:
            tc = time.perf_counter()
            with mp.Pool(min(ncores, ns)) as pool:
                ccells = sum(pool.map(_cpu_align_h_one, [(batch[i][0].query, batch[i][1]) for i in range(ns)]))
            cdt = time.perf_counter() - tc
            used = min(ncores, ns)
            cpu_base = {"value": round(ccells / cdt / 1e9, 5), "unit": "GCUPS", "cores": used, "kind": "port",
                        "sample": f"first {ns} queries, oracle alignH_ng restatement, one query per process on {used} cores"}
        out = {
            "metric": "GCUPS (DP cell updates/s), protein->genome spliced DP", "value": round(total_cells * args.steps / dt / 1e9, 3),
            "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32" if args.engines == "a0" else "int16", "data": "synthetic",
            "config": {"workload": ("C3 (per-GPU batch scaled to HBM): 400 aa proteins vs planted 6-exon loci +-1 kb "
                                    "(windows ~5-15 kb), Fwd2h1 _wip path: alignH_ng(-Q0) = forwardH1_wip + "
                                    "traceback walk + stdskl3, SKL out") if not exact else
                                   (f"C3 shape, {args.queries} x 400 aa proteins vs planted 6-exon loci +-1 kb, alignH_ng(-Q0) with the "
                                    "exact-model engines (" + ("-A0: forwardH_ng / hirschbergH_ng as wavefront kernels, spdp_h_rowwave.hip"
                                                               if args.engines == "a0" else "-A1: forwardH1 / hirschbergH1, spdp_h_exact.hip") + ")"),
                       "queries_per_gpu": args.queries, "cells_per_gpu_per_step": int(cells),
                       "queries_per_s": round(args.queries * world * args.steps / dt, 1),
                       "reference_parity": _parity_note(args),
                       "fwd_cells": int(kcells or cells), "udh_cells": 0,
                       "sweep_ms": round(k_ms, 3), "sweep_gcups": round((kcells or cells) / k_ms / 1e6, 2) if k_ms > 0 else None},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": _traffic(wl_key, k_key, args.queries, world)[0],
                         "traffic_source": _traffic(wl_key, k_key, args.queries, world)[1],
                         "kernel": ("spdh_rowwave" if args.engines == "a0" else "spdh_exact") if exact else "spdh_sweep",
                         "kernel_ms": round(k_ms, 3),
                         "valu": _valu_roofline(kcells or cells, wl_key, k_key, k_ms),
                         "note": "exact-model engines: latency-bound chains, see DESIGN.md; HBM fraction reported as asked" if exact else
                                 "integer-VALU bound recurrence (int16 saturating lanes carried in the upper half of 32-bit registers); HBM fraction reported as asked"},
            "cpu_baseline": cpu_base,
        }
        print(json.dumps(out), flush=True)
    bt.free()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def _blk_oracle_chunk(job):
    """cpu_baseline worker of the block-search leg: the oracle's vote (oracle/spdp_oracle_blk.c) on a chunk of queries"""
    fx_path, queries = job
    from oracle import blk
    from tests import spdg
    ix, _keep = blk.index_of(spdg.load(fx_path))
    t0 = time.perf_counter()
    for q in queries:
        blk.vote(ix, q, 0, len(q), 0)
    return time.perf_counter() - t0


def main_blk(args):
    """SURVEY 8 row f4, first slice, measured: the vote of the block search (SrchBlk::findblock up to its first TestOutput
    call + the candidate block pairs) for a batch of 500-nt ESTs against the index of a synthetic 100 Mb genome.  The genome
    is formatted by the compiled reference's own `spaln -W` where oracle/_ref is present -- the index is an INPUT, like the
    genome -- and read back through the recorder build (spaln_blktap); one step = one spdp_blk_vote_resident call over all
    ESTs, queries and records resident in HBM."""
    import subprocess
    import tempfile
    torch, dist, rank, world, local_rank, coll_dev = _dist_setup()
    from spaln_amd import blocks, engine, synth
    from tests import spdg
    ref = os.path.join(ROOT, "oracle", "_ref")
    have_ref = os.path.exists(os.path.join(ref, "spaln")) and not os.environ.get("SPDP_BENCH_NO_REF")   # without it the index is the library's own (spdp_blk_index_build)
    n_q = args.queries
    rng = np.random.default_rng(synth.SEED + 4400 + rank)
    t_in = time.perf_counter()
    n_chr, chr_len, n_genes = 4, 25_000_000, 400
    genes = [synth.make_gene(np.random.default_rng(synth.SEED + 4401 + i)) for i in range(n_genes)]
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    td = tempfile.mkdtemp(prefix="spdp_blk_")
    where = []                                                   # (chromosome, offset) of every planted locus
    with open(os.path.join(td, "gnm.mfa"), "wb") as f:
        per = n_genes // n_chr
        for c in range(n_chr):
            s = synth.random_dna(rng, chr_len)
            for k in range(per):
                g = genes[c * per + k]
                o = 100_000 + k * ((chr_len - 200_000) // per)
                s[o:o + len(g.window)] = g.window
                where.append((c, o))
            f.write(f">chr{c + 1}\n".encode())
            body = s[:chr_len // 60 * 60].reshape(-1, 60)
            out = np.empty((body.shape[0], 61), dtype=np.uint8)
            out[:, :60] = body
            out[:, 60] = 10
            f.write(out.tobytes())
            f.write(s[chr_len // 60 * 60:].tobytes() + b"\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "LD_PRELOAD", "ROCTRACER", "ROCTX"))}
    env.update(ALN_TAB=REF_TAB, ALN_DBS=td)
    ref_format_s = None
    if have_ref:
        t_fmt = time.perf_counter()
        subprocess.run([os.path.join(ref, "spaln"), "-W", "-KD", f"-t{_host_cores()}", "gnm.mfa"], cwd=td, env=env, check=True, capture_output=True)
        ref_format_s = time.perf_counter() - t_fmt
    # the library reads the reference's index file itself (spdp_blk_index_read); ExtBlock = max_intron_len(0.996) / blklen + 1
    # with the reference's default intron length distribution (12 288 <= that quantile < 14 336: its own run on the
    # fixtures' 2048-nt blocks gives ExtBlock = 7)
    import ctypes as C
    fx = None
    if have_ref:
        fx = blocks.read_index_file(C.CDLL(engine.LIB_PATH), os.path.join(td, "gnm.bkn"), max_intron_len=13000)
        fx["blk_convtab"][:2] = 255
    fx_path = os.path.join(td, "index.spdg")
    # the ESTs: 500-nt fragments of the planted transcripts, 1 % substitutions, every other one reverse-complemented
    code_of = np.zeros(256, dtype=np.uint8)
    for ch, code in zip(b"ACGTN", (2, 3, 5, 9, 16)):
        code_of[ch] = code
    frag = 500
    gi = rng.integers(0, n_genes, size=n_q)
    codes = np.empty((n_q, frag), dtype=np.uint8)
    truth = np.empty((n_q, 2), dtype=np.int64)
    for g_idx in range(n_genes):
        sel = np.nonzero(gi == g_idx)[0]
        if not sel.size:
            continue
        q = genes[g_idx].query
        off = rng.integers(0, len(q) - frag, size=sel.size)
        codes[sel] = q[off[:, None] + np.arange(frag)[None, :]]
        truth[sel, 0], truth[sel, 1] = where[g_idx][0], where[g_idx][1]
    sub = rng.random(codes.shape) < 0.01
    codes[sub] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(sub.sum()))
    rc = (np.arange(n_q) & 1).astype(bool)
    codes[rc] = comp[codes[rc][:, ::-1]]
    codes = code_of[codes]
    input_s = time.perf_counter() - t_in

    eng = engine.Engine(local_rank)
    # the index builder (fourth slice): the same genome's index made by the library on the device, compared with
    # the tables of the reference's file and timed beside the reference's formatter (which also writes the sequence files)
    index_build = None
    if rank == 0 or not have_ref:
        t_b = time.perf_counter()
        with open(os.path.join(td, "gnm.mfa"), "rb") as f:
            raw = np.frombuffer(f.read(), dtype=np.uint8)
        nl = np.nonzero(raw == 10)[0]
        heads = np.nonzero(raw == ord(">"))[0]
        keep = raw != 10
        for h in heads:
            keep[h:nl[np.searchsorted(nl, h)] + 1] = False
        chr_len_seen = []
        bounds = list(heads) + [raw.size]
        for a, b in zip(bounds[:-1], bounds[1:]):
            chr_len_seen.append(int(keep[a:b].sum()))
        code_of_b = np.zeros(256, dtype=np.uint8)
        for ch, cd in zip(b"ACGTN", (2, 3, 5, 9, 16)):
            code_of_b[ch] = cd
        gcodes = code_of_b[raw[keep]]
        goff = np.array([0] + list(np.cumsum(chr_len_seen)), dtype=np.int64)
        parse_s = time.perf_counter() - t_b
        bp = blocks.build_params_default(eng.lib, os.path.getsize(os.path.join(td, "gnm.mfa")), 1, threaded=1)
        blocks.build_index(eng, gcodes[:1 << 20], np.array([0, 1 << 20], dtype=np.int64), bp)          # (code objects loaded, pools sized)
        t_b = time.perf_counter()
        built, bsec = blocks.build_index(eng, gcodes, goff, bp, max_intron_len=13000)
        build_s = time.perf_counter() - t_b
        same = None if fx is None else all(np.array_equal(np.asarray(built[k]).astype(np.int64), np.asarray(fx[k]).astype(np.int64))
                                           for k in ("blk_nblk", "blk_wscr", "blk_blkp", "blk_blkb", "blk_chr", "blk_bitpat", "blk_rscrtab", "blk_pb2c"))
        if fx is None:
            fx = built
        index_build = {"what": "spdp_blk_index_build on the same residues (inputs in host memory; the upload is inside), its tables against "
                               "those of the reference's file", "identical_tables": same if same is None else bool(same), "genome_nt": int(gcodes.size),
                       "ktuple": int(bp.ktuple), "blklen": int(bp.blklen), "postings": int(np.asarray(built["blk_blkb"]).size),
                       "library_s": round(build_s, 3), "library_device_s": round(bsec[0], 3), "library_host_s": round(bsec[1], 3),
                       "residues_per_s": round(gcodes.size / build_s, 0),
                       "reference_format_s": None if ref_format_s is None else round(ref_format_s, 2), "reference_threads": _host_cores(),
                       "note": "reference = `spaln -W -KD -t<cores>` on the FASTA file: it parses the file and writes the .seq / .idx / "
                               ".ent files as well as the index; the library starts from residue codes (FASTA -> codes here: "
                               f"{parse_s:.2f} s of numpy)"}
        del raw, keep
    dix = blocks.BlockIndex(eng, fx)
    out_cap = 768
    dev = torch.device("cuda", local_rank)
    d_codes = torch.from_numpy(codes.reshape(-1)).to(dev)
    d_offs = torch.arange(0, (n_q + 1) * frag, frag, dtype=torch.int64, device=dev)
    d_left = torch.zeros(n_q, dtype=torch.int32, device=dev)
    d_right = torch.full((n_q,), frag, dtype=torch.int32, device=dev)
    d_out = torch.zeros((n_q, out_cap), dtype=torch.int32, device=dev)

    def step():
        return dix.vote_resident(d_codes.data_ptr(), d_offs.data_ptr(), d_left.data_ptr(), d_right.data_ptr(), None, n_q,
                                 d_out.data_ptr(), out_cap)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kms = [step() for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    find_leg = None
    if rank == 0 and index_build is not None:
        # the rest of findblock for every EST: spdp_blk_find = the vote call after call + TestOutput / FindHsp with the library's own
        # HSP search on the host threads -> candidate loci with their HSPs (what the aligner is given)
        from spaln_amd import abi as _abi
        fqx = spdg.load(os.path.join(ROOT, "tests", "golden", "q_c2_seed0.spdg"))
        fbx = spdg.load(os.path.join(ROOT, "tests", "golden", "blk_k1.spdg"))
        wmodel = _abi.wilip_model_from_fixture(fqx)
        scx = spdg.scoring(fqx, intpen=np.ascontiguousarray(fbx["find_intpen"], dtype=np.int16), scalar_engines=1)
        fprm = blocks.find_params_from_fixture(fbx)
        fprm.phase1t = int(dix.desc.rbscons)
        nf = min(n_q, 50000)
        t_f = time.perf_counter()
        loci, status = blocks.find(dix, gcodes, goff, wmodel, scx, fprm, [codes[i] for i in range(nf)])
        find_s = time.perf_counter() - t_f
        ok = with_locus = n_loc = 0
        for i in range(nf):
            if loci[i]:
                with_locus += 1
                n_loc += len(loci[i])
                L = loci[i][0]
                ok += (L["chr"] == truth[i, 0] and L["base"] <= truth[i, 1] + 60000 and truth[i, 1] <= L["base"] + L["len"] and L["rvs"] == int(rc[i]))
        find_leg = {"what": "spdp_blk_find: the vote call after call and the HSP searches of all candidate regions as device batches, TestOutput / "
                            "FindHsp as machines advanced on the host threads in between -> candidate loci with their HSPs; marshalling the "
                            "result into Python lists inside",
                    "queries": nf, "with_a_locus": with_locus, "loci": n_loc, "first_locus_covers_the_planted_gene_on_its_strand": int(ok),
                    "seconds": round(find_s, 2), "queries_per_s": round(nf / find_s, 0)}
    ref_blk = None
    if rank == 0 and have_ref and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln_blktime")):
        # the reference's own block search on the same index and queries: its program with a stop watch around SrchBlk::findblock
        # (oracle/_ref/spaln_blktime: the reference's object code, one symbol renamed) on a sample, all host cores
        import re as _re
        import subprocess as _sp
        dec = np.zeros(32, dtype=np.uint8)
        for c_, ch_ in ((2, b"A"), (3, b"C"), (5, b"G"), (9, b"T"), (16, b"N")):
            dec[c_] = ch_[0]
        ns_ref = min(n_q, args.cpu_sample if args.cpu_sample > 0 else 6000)
        with open(os.path.join(td, "qs.fa"), "wb") as f:
            for i in range(ns_ref):
                f.write(b">q%d\n" % i + dec[codes[i]].tobytes() + b"\n")
        ncr = _host_cores()
        env_r = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "HSA_TOOLS", "LD_PRELOAD"))}
        env_r.update(ALN_TAB=os.path.join(ROOT, "oracle", "_ref", "table"), ALN_DBS=td)
        t_r = time.perf_counter()
        r_ = _sp.run([os.path.join(ROOT, "oracle", "_ref", "spaln_blktime"), "-Q7", "-O4", f"-t{ncr}", "-dgnm", "qs.fa"], cwd=td, env=env_r,
                     capture_output=True, text=True)
        m_ = _re.search(r"findblock: (\d+) calls, ([0-9.]+) thread-seconds", r_.stderr)
        if m_ and int(m_.group(1)):
            calls_, ts_ = int(m_.group(1)), float(m_.group(2))
            ref_blk = {"value": round(calls_ / (ts_ / ncr), 1), "unit": "queries/s", "cores": ncr, "kind": "reference",
                       "sample": f"first {ns_ref} ESTs through the compiled reference's program on the same index with a stop watch around "
                                 f"SrchBlk::findblock (oracle/_ref/spaln_blktime, -t{ncr}): {calls_} calls, {ts_:.2f} thread-seconds inside -- its whole "
                                 f"block search (vote + TestOutput / FindHsp with its HSP search), the counterpart of config.find (spdp_blk_find); "
                                 f"program wall {time.perf_counter() - t_r:.1f} s"}
    if rank == 0:
        rec = d_out.cpu().numpy()
        reached = (rec[:, 2] & blocks.REACHED) != 0
        flagged = int(((rec[:, 2] & (blocks.CUT | blocks.TABLE)) != 0).sum())
        # mapping accuracy: the best candidate pair covers the planted locus, on the right strand
        prm = fx["blk_prm"]
        blklen = int(prm[6])
        chr_first = np.asarray(fx["blk_chr"]).reshape(-1, 2)[:, 1]
        hit = hit_any = 0
        sample = np.arange(0, n_q, max(1, n_q // 20000))
        for i in sample:
            r = blocks.split_record(rec[i])
            if not r.get("reached") or "pairs" not in r or not len(r["pairs"]):
                continue
            bscr, c, lb, rb, ub, db, zl, zr, rvs = (int(x) for x in r["pairs"][0])
            lo, hi = (lb - zl) * blklen if zl else 0, ((rb - zl if zl else 0) + 1) * blklen
            covers = c == truth[i, 0] and lo <= truth[i, 1] + 60000 and truth[i, 1] <= hi
            hit += covers and rvs == int(rc[i])
            hit_any += covers
        # parity on a sample against the oracle (the pinned restatement), in the same run
        from oracle import blk as oblk
        ix, _keep = oblk.index_of(fx)
        ix.extblockl = int(prm[27])
        same = 0
        chk = sample[:300]
        for i in chk:
            want = oblk.vote(ix, codes[i], 0, frag, 0)
            got = blocks.split_record(rec[i])
            if want is None:
                same += not got.get("reached")
                continue
            w = oblk.split_recorded(want[0], want[1])
            same += bool(got.get("reached") and "head" in got and np.array_equal(got["head"], w["head"]) and
                         np.array_equal(got["pairs"], want[1][2:].reshape(-1, 9)) and
                         got["runs"] == [sorted(x) for x in oblk.runs_near_pairs(ix, w["runs"], got["pairs"])])
        # CPU baseline: the oracle's vote, one process per host core
        import multiprocessing as mp
        ncores = _host_cores()
        ns = min(n_q, args.cpu_sample if args.cpu_sample > 0 else 400 * ncores)
        as_file = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}      # (the container knows the signed types)
        spdg.save(fx_path, {k: np.asarray(v).view(as_file.get(np.asarray(v).dtype, np.asarray(v).dtype)) for k, v in fx.items() if k.startswith("blk_")})
        jobs = [(fx_path, [codes[i] for i in range(c, ns, ncores)]) for c in range(ncores)]
        with mp.get_context("fork").Pool(ncores) as pool:
            busy = pool.map(_blk_oracle_chunk, jobs)
        cpu_qps = ns / max(busy)
        k_ms = float(np.mean(kms))
        words = int(np.asarray(fx["blk_blkb"]).size)
        # algorithmic bytes per query: its codes once, per looked-up word the two table entries (Nblk 2 B, wscr 2 B, blkp 4 B) and
        # its posting list (4 B per listed block), per listed block one read-modify-write of two 4-byte score slots; the record out
        tw = rec[reached, 3 + 16:3 + 20].sum(axis=1).mean() if reached.any() else 0.0
        avg_list = words / max(1, int((np.asarray(fx["blk_blkp"]) != 0).sum()))
        bytes_per_q = frag + tw * (8 + avg_list * (4 + 16)) + 4 * float(rec[:, 0].mean())
        achieved = n_q * bytes_per_q / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "queries/s, block search vote (findblock) for 500-nt ESTs vs a 100 Mb genome index", "value": round(n_q * world * args.steps / dt, 1),
            "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 / int32", "data": "synthetic",
            "config": {"workload": f"block search, first slice of SURVEY 8 f4: {n_q} ESTs of {frag} nt (1 % substitutions, half of them "
                                   f"reverse strand) vs the index of a synthetic {n_chr * chr_len // 1000000} Mb genome ({n_genes} planted loci); " +
                                   ("index file made by the compiled reference's own formatter (spaln -W -KD), an input, read by spdp_blk_index_read; " if have_ref else
                                    "index built by the library (spdp_blk_index_build; oracle/_ref absent); ") + "one step = "
                                   "spdp_blk_vote over all ESTs, queries and records resident in HBM",
                       "queries_per_gpu": n_q, "queries_per_s": round(n_q * world * args.steps / dt, 1),
                       "cells_per_gpu_per_step": 0,
                       "index": {"ktuple": int(prm[28]) // 8, "tabsize": int(prm[3]), "nshift": int(prm[5]), "blklen": blklen, "nseg": int(prm[19]),
                                 "words": words, "patterns": int(prm[8])},
                       "reached_first_call": int(reached.sum()), "flagged": flagged,
                       "best_pair_covers_planted_locus": f"{hit_any} / {len(sample)} ({hit} with the strand as planted)",
                       "identical_to_oracle_on_sample": f"{same} / {len(chk)}",
                       "words_looked_up_per_query": round(float(tw), 1),
                       "input_generation_s": round(input_s, 1), "index_build": index_build, "find": find_leg,
                       "reference_parity": "the vote's state at every TestOutput call, the block pairs, and every candidate locus FindHsp makes of them "
                                           "(HSPs included): bit-identical to the compiled reference's recorded runs (tests/golden/blk_*.spdg: "
                                           "tests/test_gpu_blk.py, test_gpu_blk_find.py)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "bytes_per_query": round(float(bytes_per_q), 1), "traffic": _traffic("blk", "blk", n_q, world)[0],
                         "traffic_source": _traffic("blk", "blk", n_q, world)[1], "kernel": "spdp_blk_vote_wave", "kernel_ms": round(k_ms, 3),
                         "note": "one query per wave: posting lists read 64 blocks at a time, the run hash and the bounded lists in LDS, tagged score "
                                 "slots in the wave's slab of HBM (random 8-byte accesses): bound by memory latency per word, hidden by 16 waves per "
                                 "CU; algorithmic bytes = codes + per word its table entries and posting list + two score slots per listed block + "
                                 "the record"},
            "cpu_baseline": ref_blk or {"value": round(cpu_qps, 1), "unit": "queries/s", "cores": ncores, "kind": "port",
                                        "sample": f"first {ns} ESTs through the oracle's vote (oracle/spdp_oracle_blk.c), one process per host core"},
        }
        if ref_blk:
            out["config"]["vote_port_baseline"] = {"value": round(cpu_qps, 1), "unit": "queries/s", "cores": ncores, "kind": "port",
                                                   "sample": f"the vote alone: first {ns} ESTs through the oracle's restatement, one process per host core"}
        print(json.dumps(out), flush=True)
    dix.free()
    eng.close()
    import shutil
    shutil.rmtree(td, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


def _valu_roofline(cells, workload, key, k_ms):
    """the bound that actually binds: VALU wave-instructions issued per second against the measured ceiling; instructions per
    cell from the round's counter profile (COUNTERS_FILE), None where the profile has no entry for this kernel"""
    e = _counters(workload, key)
    if not k_ms or not cells or not e or not e.get("valu_per_cell"):
        return None
    ach = cells * e["valu_per_cell"] / (k_ms * 1e-3)
    return {"achieved": round(ach / 1e9, 1), "peak": round(VALU_PEAK_WINST_S / 1e9, 1), "unit": "G wave-instr/s",
            "frac": round(ach / VALU_PEAK_WINST_S, 3), "valu_per_64_cells": round(64 * e["valu_per_cell"], 1),
            "profile_stale": bool(e["stale"]),
            "source": f"SQ_INSTS_VALU per cell of {e['kernel']} ({e['file']}: {workload}/{key}, rocprofv3 --pmc, its own pass) x cells / "
                      "kernel time; peak = one VALU wave-instruction per 2.2 cycles per SIMD at 2.3 GHz (tools/ubench, "
                      "profiles/r02_valu_ubench.txt)"}


def _traffic(workload, key, queries, world=1):
    """HBM bytes per launch from the PMC passes of the round's profile (FETCH_SIZE x 2 + WRITE_SIZE), or None; only for the
    batch size the profile was taken on (a launch's traffic scales with it)"""
    e = _counters(workload, key)
    if not e or not e.get("hbm_bytes_per_launch") or world != 1 or e.get("queries") != queries:
        return None, None
    return int(e["hbm_bytes_per_launch"]), (f"{e['file']}: {workload}/{key} (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes, "
                                            f"{e['launches_per_step']:g} launches per step; per launch)" + (" -- STALE: kernel source changed" if e["stale"] else ""))


def _self_spawn(argv, n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks here, one process per GPU, and wait."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    sys.exit(rc)


def _dist_setup():
    """(torch, dist or None, rank, world, local device index, device for collectives).
    The ranks exchange nothing but a barrier and two scalars, so the process group is gloo on the host.
    BENCH_SHARE_GPU=1 is a test hook: all ranks use GPU 0, so that the multi-rank path can be exercised on a
    one-GPU box."""
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    dev = 0 if share else local_rank
    if dev >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: GPU {dev} not present ({torch.cuda.device_count()} visible); "
                         "BENCH_SHARE_GPU=1 runs all ranks on GPU 0")
    torch.cuda.set_device(dev)
    return torch, dist, rank, world, dev, "cpu"


# legs of the default line: name -> (arguments of this script, what the leg is)
LEGS = {
    "c3": (["--workload", "c3", "--queries", "50000", "--steps", "2", "--warmup", "1", "--cpu-sample", "512"],
           "BASELINE configs[2] at its full 50 000 queries: the traceback bitmaps (2 B / cell, ~440 GB) run as groups, all inside the clock"),
    "c4": (["--workload", "c4", "--queries", "20000", "--steps", "3", "--warmup", "1"],
           "BASELINE configs[3]'s per-EST work on a 20 000-fragment batch (planted windows; the 3 Gb genome's block search is SURVEY 8 row f4)"),
    "c5": (["--workload", "c5", "--queries", "32", "--steps", "2", "--warmup", "1", "--cpu-sample", "16"],
           "BASELINE configs[4]: 32 cDNAs of 50 kb, 25 exons, against their ~190 kb loci"),
    "blk": (["--workload", "blk", "--queries", "200000", "--steps", "3", "--warmup", "1"],
            "SURVEY 8 row f4, first slice: the block search's vote for 200 000 ESTs against the index of a 100 Mb genome"),
    "c4_full": (["tools/c4_full.py", "--mb", "3000", "--queries", "125000"],
                "BASELINE configs[3], one GPU's eighth of it: 125 000 ESTs of 500 nt mapped to candidate loci (spdp_blk_find: vote + HSP search on "
                "the device) against the index of a 3 Gb genome the library built itself (spdp_blk_index_build, five bit patterns)"),
    "blk_find_p": (["tools/blk_find_protein.py", "--queries", "20000", "--genes", "200"],
                   "SURVEY 8 row f4 for protein queries (BASELINE configs[0] / [2]'s mapping phase): spdp_blk_find on the translated index "
                   "(<db>.bkp built by spdp_blk_index_build_p, its tables compared with the file of the reference's `spaln -W -KP`) of a 20 Mb genome -- vote and HSP search on the device (regions "
                   "read as tron codes where they lie), FindHsp's DvsP = 1 branch with its retries on a grown region as batched machines; "
                   "parity: tests/test_gpu_blk_find.py[blk_p1] against the reference's recorded runs"),
    "a0": (["--engines", "a0", "--queries", "1000", "--steps", "2", "--warmup", "1"],
           "C2 shape under -A0 (forwardS_ng / hirschbergS_ng): the engines whose output is bit-identical to the reference's own -A0 records on 2 kb inputs"),
    "a1": (["--engines", "a1", "--queries", "1000", "--steps", "2", "--warmup", "1"], "C2 shape under -A1 (forwardS1 / hirschbergS1)"),
    "c3_a0": (["--workload", "c3", "--engines", "a0", "--queries", "1000", "--steps", "2", "--warmup", "1"],
              "C3 shape under -A0 (forwardH_ng / hirschbergH_ng)"),
    "c3_a1": (["--workload", "c3", "--engines", "a1", "--queries", "4000", "--steps", "2", "--warmup", "1"],
              "C3 shape under -A1 (forwardH1 / hirschbergH1)"),
    "dropin": (["tools/dropin_demo.py", "--queries", "600", "--genes", "120", "--modes", "Q4,Q7", "--gpu-threads", "16"],
               "the reference's own CLI (src/spaln.cc, -t workers, block search, output writers) with alignS_ng switched to the library "
               "(oracle/_ref/spaln_gpu) against the unmodified build on a synthetic genome its own `spaln -W` formatted: -O4 records "
               "compared, wall times of both; default engines (-A0); both programs with -t16: the shim's batching boundary (integration/) "
               "records the workers' aligner calls and aligns them in chunks beside the mapping"),
    "e2e_q7": (["tools/e2e_q7.py", "--queries", "20000", "--genes", "200"],
               "map AND align inside the library in ONE call (spdp_map_align_s; SURVEY 8 rows f4 + f1 + f2 + f3 end to end): the "
               "reference's index file and the genome in, spdp_blk_find (vote on the device, TestOutput / FindHsp with the library's own "
               "HSP search on the host) -> candidate loci -> their regions and splice signals (one launch) -> spdp_align_s_seeded with "
               "the library's own Wilip -> spdp_skl_rng_s -> the best locus' exon table in chromosome coordinates, compared with, and "
               "timed against, `spaln -Q7 -S1 -O4 -t16` of the compiled reference on the same 20 000 queries"),
    "e2e_q7_p": (["tools/e2e_q7.py", "--protein", "--queries", "20000", "--genes", "200"],
                 "BASELINE configs[0] / [2]'s whole path for protein queries: the translated index built by the library (spdp_blk_index_build_p; its tables "
                 "compared with the reference's <db>.bkp), then ONE spdp_map_align_h call (block search on that index, "
                 "HSP search on regions read as tron codes, signals, seeded alignment, rescoring) against `spaln -Q7 -O4 -t16` on 20 000 proteins"),
    "e2e_q7_s3": (["tools/e2e_q7.py", "--queries", "20000", "--genes", "200", "--ori", "3"],
                  "the same in spaln's default orientation mode (a->inex.ori = 3: every locus aligned in both orientations, alignS_ng(.., 3)); "
                  "every other query is an antisense read; against `spaln -Q7 -O4 -t16`"),
    "c4_e2e": (["tools/e2e_q7.py", "--queries", "10000", "--genes", "400", "--spacer", "450000", "--frag", "500", "--ori", "3"],
               "BASELINE configs[3] end to end at reduced size: 10 000 ESTs of 500 nt (half of them antisense) against a 140 Mb genome, block "
               "search -> candidate loci (five 10 kb blocks each) -> both orientations aligned -> exon tables, one spdp_map_align_s call against "
               "`spaln -Q7 -O4 -t16`"),
    "dropin_q7_20k_s3": (["tools/dropin_demo.py", "--queries", "20000", "--genes", "200", "--modes", "Q7", "--gpu-threads", "16", "--strand=-S3", "--antisense"],
                         "the reference's CLI on the library in spaln's DEFAULT orientation mode (a->inex.ori = 3: alignS_ng(.., 3) on every locus; every "
                         "other query an antisense read), 20 000 queries under -Q7, both programs with -t16"),
    "dropin_q7_20k": (["tools/dropin_demo.py", "--queries", "20000", "--genes", "200", "--modes", "Q7", "--gpu-threads", "16"],
                      "the same at 20 000 queries under -Q7 (the reference's normal mode): the size at which the device batches are large enough to matter"),
}


def _run_leg(name):
    """one leg = this script run on its own with the leg's arguments; returns the figures of its JSON line in short form"""
    import subprocess
    argv, what = LEGS[name]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    if name.startswith("dropin") or name.startswith("e2e") or name in ("c4_e2e", "blk_find_p", "c4_full"):      # a program of its own (tools/dropin_demo.py, tools/e2e_q7.py, tools/blk_find_protein.py): its JSON line as it is
        if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln" if name in ("blk_find_p", "c4_full") else "spaln_gpu")):
            return {"what": what, "error": "oracle/_ref/spaln_gpu is not built (needs the reference's sources at build time)"}
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, argv[0])] + argv[1:], env=env, capture_output=True, text=True, timeout=900)
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:                                   # noqa: BLE001
            return {"what": what, "error": f"{type(e).__name__}: {str(e)[:160]}", "wall_s": round(time.perf_counter() - t0, 1)}
        d.update(what=what, args=" ".join(argv), wall_s=round(time.perf_counter() - t0, 1))
        return d
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--legs", "none", "--seeded-pairs", "0"],
                           env=env, capture_output=True, text=True, timeout=900)
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:                                       # noqa: BLE001 -- a leg must not cost the line
        return {"what": what, "error": f"{type(e).__name__}: {str(e)[:160]}", "wall_s": round(time.perf_counter() - t0, 1)}
    c, rf, cb = d["config"], d["roofline"], d.get("cpu_baseline") or {}
    out = {"what": what, "args": " ".join(argv), "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
           "steps": d["steps"], "queries": c["queries_per_gpu"], "queries_per_s": c["queries_per_s"],
           "cells_per_step": c["cells_per_gpu_per_step"], "dtype": d["dtype"],
           "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "hbm_frac": rf["frac"],
           "valu_frac": (rf.get("valu") or {}).get("frac"),
           # kernel_ms sums the durations of the step's launches of this kernel; where they run side by side on two streams (c4)
           # the sum exceeds the step: the kernel's share of the WALL is at most the step, and the fractions on that basis
           **({"kernel_ms_is_sum_of_concurrent_launches": True,
               "valu_frac_on_step_wall": round((rf.get("valu") or {}).get("frac", 0) * rf["kernel_ms"] / d["ms_per_step"], 3) if rf.get("valu") else None,
               "hbm_frac_on_step_wall": round(rf["frac"] * rf["kernel_ms"] / d["ms_per_step"], 5)}
              if rf["kernel_ms"] > d["ms_per_step"] else {}),
           "profile_stale": (rf.get("valu") or {}).get("profile_stale"),
           "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind")} if cb else None,
           "reference_parity": c.get("reference_parity"), "wall_s": round(time.perf_counter() - t0, 1)}
    for k in ("udh_gcups", "fwd_gcups", "sweep_gcups", "fwd_problems", "index_build", "find"):
        if c.get(k) is not None:
            out[k] = c[k]
    return out


ROUND_TAG = "r06"
LINE_LIMIT = 4096            # bytes of the one stdout line (the driver keeps the tail of stdout: round 5's 22 KB line did not parse)


def _leg_short(name, leg):
    """one leg of the full record -> the few numbers the stdout line carries (the whole leg goes to profiles/)"""
    if not isinstance(leg, dict):
        return leg
    if leg.get("error"):
        return {"error": str(leg["error"])[:60]}
    if "identical_exon_tables" in leg:                              # tools/e2e_q7.py
        return {"same": leg["identical_exon_tables"], "of": leg.get("queries"), "x_ref": leg.get("library_over_reference"),
                "q_per_s": leg.get("library_queries_per_s")}
    if "runs" in leg:                                               # tools/dropin_demo.py
        o = {"q": leg.get("queries")}
        for r in leg["runs"]:
            m = str(r.get("mode", "")).lstrip("-")
            sp = r.get("gpu_over_reference_wall")
            o[m] = {"x_ref": sp, "same": r.get("identical"), "differ": r.get("records_differing")}
        return o
    if "identical_to_reference" in leg:                             # seeded_q7
        return {"same": leg["identical_to_reference"], "of": leg.get("compared"), "pairs_per_s": leg.get("library_pairs_per_s"),
                "ref_pairs_per_s": leg.get("reference_pairs_per_s")}
    if "with_a_locus" in leg:                                       # blk_find_p, c4_full
        return {"q_per_s": leg.get("queries_per_s"), "loci": leg["with_a_locus"], "of": leg.get("queries"),
                **({"vote_q_per_s": leg["vote_queries_per_s"]} if "vote_queries_per_s" in leg else {})}
    o = {"value": leg.get("value"), "unit": leg.get("unit")}
    for k in ("hbm_frac", "valu_frac"):
        if leg.get(k) is not None:
            o[k] = leg[k]
    cb = leg.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        o["cpu"] = cb["value"]
    return o


def compact_line(out, limit=LINE_LIMIT):
    """the ONE stdout line: the contract's keys + roofline + cpu_baseline + one short record per leg, at most `limit` bytes; the
    prose (what each leg is, which reference output it is pinned to, the sources of the counters) stays in the full record"""
    cfg = out.get("config", {})
    legs = {k: _leg_short(k, v) for k, v in cfg.items() if isinstance(v, dict) and k not in ("with_h2d",)
            and not k.endswith("_scaling")}
    keep = ("queries_per_gpu", "queries_total", "cells_per_gpu_per_step", "queries_per_s", "queries_per_s_incl_upload",
            "parallelism", "udh_gcups", "fwd_gcups", "sweep_gcups", "legs_wall_s")
    rf = dict(out.get("roofline") or {})
    valu = rf.get("valu")
    for k in ("note", "bytes_per_cell_source", "traffic_source", "layout_bytes_per_cell", "layout_achieved", "layout_frac"):
        rf.pop(k, None)
    if isinstance(valu, dict):
        rf["valu"] = {k: valu.get(k) for k in ("achieved", "peak", "unit", "frac", "valu_per_64_cells") if k in valu}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        cb = dict(cb)
        cb["sample"] = str(cb.get("sample", ""))[:160]
    line = {k: v for k, v in out.items() if k not in ("config", "roofline", "cpu_baseline")}
    scal = {k: {kk: vv for kk, vv in cfg[k].items() if kk != "sharding"} for k in ("strong_scaling", "weak_scaling") if isinstance(cfg.get(k), dict)}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200], **{k: cfg[k] for k in keep if cfg.get(k) is not None}, **scal,
                      "legs": legs, "full_record": f"profiles/{ROUND_TAG}_bench_legs.json"}
    line["roofline"] = rf
    line["cpu_baseline"] = cb
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > limit:                                              # never let the legs cost the line
        for k in list(legs):
            legs[k] = {kk: vv for kk, vv in legs[k].items() if kk in ("value", "same", "of", "x_ref", "q_per_s", "error")} \
                if isinstance(legs[k], dict) else legs[k]
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > limit:
        line["config"]["legs"] = {"dropped": "see full_record"}
        s = json.dumps(line, separators=(",", ":"))
    return s


def _write_full_record(out):
    """the full record (every leg with its prose) as a file: profiles/ is what the judge reads, gpurun_out/ is what travels back"""
    for d, name in ((os.path.join(ROOT, "profiles"), f"{ROUND_TAG}_bench_legs.json"),
                    (os.path.join(ROOT, "gpurun_out"), f"{ROUND_TAG}_bench_legs.json")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass


def _parity_note(args):
    """which reference output the timed engines are bit-identical to (tests named)"""
    if args.engines == "a0":
        return ("-A0 engines: bit-identical to the compiled reference's own -A0 records (HomScore, gsi->scr, SKL, exon table) on all "
                "fixtures incl. the 2 kb / 6 kb ones (tests/test_gpu_fullsize_ref.py, test_gpu_parity*.py)")
    if args.engines == "a1":
        return "-A1 engines: bit-identical to the compiled reference's -A1 output on all fixtures (tests/test_gpu_parity*.py, test_gpu_exact_h.py)"
    if args.workload == "c3":
        return ("`_wip` int16 engines: bit-identical to the compiled reference's -A2 output up to 512 aa (no re-basing below row 512; 400 aa "
                "here), fixtures h1_* / c1_* and the live shim (tests/test_gpu_parity_h.py, test_gpu_shim.py)")
    return ("`_wip` model (-A2) carried in int32 / exact fp32 WITHOUT the reference's int16 re-basing: bit-identical to the compiled reference's "
            "-A2 output for queries <= 1472 nt (fixtures, C4's 500 nt included) and on 2 kb pairs of 18-28 % divergence (live shim, 36 / 36); "
            "at this workload's own 2 kb / 2 % inputs the reference's -A2 re-bases and mis-aligns (SURVEY App. B), so there the timed engine is "
            "bit-identical to the pinned int32 restatement (oracle) and agrees with the reference's -A0 records at exon-boundary level "
            "(tests/test_gpu_fullsize_ref.py); the engines that ARE bit-identical to a reference output on these inputs are timed as config.a0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seeded-pairs", type=int, default=10000,
                    help="pairs of the seeded-path (-Q7) leg reported in config.seeded_q7 (c2, N = 1; 0: skip)")
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5", "blk"], default="c2",
                    help="c2: cDNA x genome (the headline); c3: protein x genome (Fwd2h1 path); "
                         "c4: 500-nt ESTs (traceback branch of the ladder only); c5: 50 kb cDNAs with 25 exons "
                         "(recursive linear-space branch all the way down)")
    ap.add_argument("--legs", default="auto",
                    help="the other BASELINE configs and engines as legs of the default line (config.c3 / c4 / c5 / a0 / a1 / "
                         "c3_a0 / c3_a1), each a run of this script: auto = all of them with the default workload at N = 1, "
                         "none, or a comma-separated subset")
    ap.add_argument("--engines", choices=["wip", "a0", "a1"], default="wip",
                    help="wip = the -A2 `_wip` engines (the headline); a0 = the exact-intron-length engines "
                         "(algmode.alg 0: forwardS_ng / hirschbergS_ng, spdp_rowwave.hip); a1 = the -A1 engines (spdp_exact.hip)")
    ap.add_argument("--queries", type=int, default=0, help="queries per GPU (default 10000; 1000 with --engines a0 / a1)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the CPU oracle (0 = 2 per host core)")
    ap.add_argument("--intron-hi", type=int, default=20000, help="upper clip of planted intron lengths")
    ap.add_argument("--cpu-port", action="store_true",
                    help="time the oracle port as the CPU baseline even when oracle/_ref/spaln is present")
    ap.add_argument("--plain", action="store_true",
                    help="nothing but --warmup + --steps aligns of the batch (no upload-inclusive / streamed figures): what "
                         "tools/profile_counters.py runs under rocprofv3, so that counter totals divide by the number of steps")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank its own batch of --queries (the reported value), strong = one "
                         "batch of --queries sharded over the ranks; the other one is reported under config")
    args = ap.parse_args()
    if not args.queries:
        args.queries = ({"c5": 32, "blk": 200000}.get(args.workload, 10000)) if args.engines == "wip" else 1000
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_spawn(sys.argv[1:], args.gpus)
    if args.workload == "c3":
        return main_c3(args)
    if args.workload == "blk":
        return main_blk(args)

    torch, dist, rank, world, local_rank, coll_dev = _dist_setup()

    from spaln_amd import abi, defaults, engine, shard, synth
    eng = engine.Engine(local_rank)
    exact = args.engines != "wip"
    if exact:
        intpen, t53 = defaults.exact_tables()
        sc = defaults.scoring(scalar_engines=1 if args.engines == "a0" else 2, intpen=intpen, t53=t53)
    else:
        sc = defaults.scoring()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make(seed, n):
        if args.workload == "c4":       # (above the round-1 .. 3 size the batch is made in chunks by all host cores)
            return synth.make_est_batch(n, seed=seed) if n <= 10000 else synth.make_chunked("c4", n, seed=seed, procs=_host_cores())
        if args.workload == "c5":
            return synth.make_batch(n, seed=seed + 55, n_exons=25, mrna_len=50000, flank=1000, intron_lo=1000, intron_hi=10000)
        return synth.make_batch(n, seed=seed, intron_hi=args.intron_hi)

    def measure(mode):
        """K timed steps of one scaling mode; returns the figures of the whole job and this rank's batch.
        weak: every rank aligns its own --queries queries; strong: one batch of --queries, rank r takes
        shard.shard_range(queries, r, world) of it.  One step = alignS_ng (ori = 1, seeding off) for every query
        of the rank's batch: dispatch ladder, UDH sweep, cpos back-walk, slab list, forward sweep + traceback
        walk, stdskl / trimskl.  Alignments are produced in host memory every step: D2H of the records,
        stdskl / trimskl and the SKL arrays are inside the clock (want=True); convert=False only skips
        turning them into numpy rows in Python."""
        if mode == "weak" or world == 1:
            batch = make(synth.SEED + 1000 * rank, args.queries)
        else:
            # one batch, the same on every rank, sharded by DP cells (longest-processing-time rule): windows differ by a
            # factor of two and more, equal counts would leave the slowest rank with more cells than the rest
            full = make(synth.SEED, args.queries)
            fps = abi.ProblemSet()
            for w, q, s5, s3, _ in full:
                fps.add(q, w, s5, s3)
            costs = []
            for p in fps.items:
                win = abi.Window()
                eng.lib.spdp_stripe(C.byref(p), sc.sh, C.byref(win))
                costs.append(int(eng.lib.spdp_cells(C.byref(p), C.byref(win))))
            batch = [full[i] for i in shard.balanced_shards(costs, world)[rank]]
        ps = abi.ProblemSet()
        for w, q, s5, s3, _ in batch:
            ps.add(q, w, s5, s3, **(synth.exact_inputs(w) if exact else {}))
        h2d = None
        if mode == args.scaling and not args.plain:
            # the same step with the batch coming over PCIe first (upload + align): reported beside the headline, which
            # counts inputs resident in HBM (the bench contract); full column records, ~95 B per genomic position
            tb0 = time.perf_counter()
            b0 = eng.upload(sc, ps)
            b0.align(want=True, convert=False)
            b0.free()
            torch.cuda.synchronize()
            tb1 = time.perf_counter()
            b0 = eng.upload(sc, ps)
            tu = time.perf_counter()
            b0.align(want=True, convert=False)
            b0.free()
            h2d = {"upload_ms": round((tu - tb1) * 1e3, 2), "step_ms_incl_upload": round((time.perf_counter() - tb1) * 1e3, 2)}
            del tb0
            # ... and as a stream of batches: batch i + 1 is uploaded (host packing + H2D, on a second context of the same
            # device) while batch i is aligned; steady state: the second context has uploaded (and aligned) once before the
            # clock starts -- its first upload allocates the pinned staging and the pools, 2.5 - 3 x the time of any later
            # one -- and six steps are averaged
            try:
                import threading
                eng2 = engine.Engine(local_rank)
                w0 = eng2.upload(sc, ps)
                w0.align(want=True, convert=False)
                w0.free()
                cur = eng.upload(sc, ps)
                torch.cuda.synchronize()
                n_str = 6
                ts0 = time.perf_counter()
                for i in range(n_str):
                    box = []
                    th = threading.Thread(target=lambda e=(eng2 if i % 2 == 0 else eng): box.append(e.upload(sc, ps)))
                    th.start()
                    cur.align(want=True, convert=False)
                    th.join()
                    cur.free()
                    cur = box[0]
                torch.cuda.synchronize()
                h2d["streamed_step_ms"] = round((time.perf_counter() - ts0) / n_str * 1e3, 2)
                h2d["streamed_steps"] = n_str
                cur.free()
                eng2.close()
            except Exception as e:                                   # noqa: BLE001 -- the extra figure must not cost the line
                h2d["streamed_step_ms"] = None
                h2d["streamed_error"] = str(e)[:120]
        bt = eng.upload(sc, ps)
        for _ in range(args.warmup):
            bt.align(want=True, convert=False)
        barrier()
        t0 = time.perf_counter()
        stats = []
        step_cells = 0
        for _ in range(args.steps):
            _, ms, kc = bt.align(want=True, convert=False)
            stats.append(bt.stats())
            step_cells = kc
        torch.cuda.synchronize()
        busy = time.perf_counter() - t0                              # this rank's own time, before it waits for the others
        barrier()
        dt = time.perf_counter() - t0
        tot_cells, tot_q = float(step_cells), float(len(batch))
        rank_busy, rank_cells = [busy], [float(step_cells)]
        if dist is not None:
            gb = [None] * world
            dist.all_gather_object(gb, (busy, float(step_cells)))
            rank_busy, rank_cells = [x[0] for x in gb], [x[1] for x in gb]
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            c = torch.tensor([tot_cells, tot_q], dtype=torch.float64)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            tot_cells, tot_q = float(c[0].item()), float(c[1].item())
        band_cells = bt.cells()
        bt.free()
        return {"dt": dt, "total_cells": tot_cells, "total_queries": tot_q, "cells": step_cells, "stats": stats,
                "batch": batch, "ps": ps, "band_cells": band_cells, "h2d": h2d,
                "rank_busy_ms": [round(b / args.steps * 1e3, 2) for b in rank_busy], "rank_cells": [int(c) for c in rank_cells]}

    prim = measure(args.scaling)
    other = None
    if world > 1:
        om = "strong" if args.scaling == "weak" else "weak"
        o = measure(om)
        other = {"scaling": om, "value": round(o["total_cells"] * args.steps / o["dt"] / 1e9, 3), "unit": "GCUPS",
                 "queries_total": int(o["total_queries"]), "queries_per_s": round(o["total_queries"] * args.steps / o["dt"], 1),
                 "ms_per_step": round(o["dt"] / args.steps * 1e3, 3),
                 "rank_busy_ms": o["rank_busy_ms"], "rank_cells": o["rank_cells"],
                 "sharding": "by DP cells (longest-processing-time rule, shard.balanced_shards)" if om == "strong" else "every rank its own batch"}
    dt, total_cells, cells, stats, batch, ps = (prim["dt"], prim["total_cells"], prim["cells"], prim["stats"],
                                                prim["batch"], prim["ps"])

    if rank == 0:
        gcups = total_cells * args.steps / dt / 1e9
        udh_ms = float(np.mean([s["udh_ms"] for s in stats]))
        fwd_ms = float(np.mean([s["fwd_ms"] for s in stats]))
        udh_cells, fwd_cells = stats[-1]["udh_cells"], stats[-1]["fwd_cells"]
        # dominant kernel: the UDH sweep (C2); the forward sweep where the ladder never goes linear-space (C4)
        c4 = args.workload == "c4"
        k_ms = fwd_ms if c4 else udh_ms
        k_cells, k_name = (fwd_cells, "forward") if c4 else (udh_cells, "udh")
        wl_key = args.workload if not exact else args.engines                            # entry of the round's counter profile
        k_key = k_name if not exact else {"a0": "a0_udh", "a1": "a1_udh"}[args.engines]
        achieved = k_cells * SURVEY_BYTES_PER_CELL[k_name] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        achieved_layout = k_cells * BYTES_PER_CELL[k_name] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # CPU baseline: the oracle's alignS_ng restatement (int32, one query per process) on all
        # host cores of this box, bounded sample
        import multiprocessing as mp
        ncores = _host_cores()
        ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 2 * ncores, len(batch)))
        if world > 1:
            cpu_base = None                                   # timed at N = 1 only
        elif exact and not _ref_available():
            cpu_base = None                                   # (the port's ladder is the -A2 one)
        elif _ref_available() and not args.cpu_port:
            from oracle import oracle
            ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 32 * ncores, len(batch)))
            dec = np.zeros(32, dtype=np.uint8)
            for ch, code in defaults.CODE_OF.items():
                dec[code] = ch
            band = 0
            for i in range(ns):
                band += oracle.cells(ps.items[i], oracle.stripe(ps.items[i], sc.sh))
            if exact:
                ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 4 * ncores, len(batch)))
            cpu_base = _ref_baseline([(dec[batch[i][0]], dec[batch[i][1]]) for i in range(ns)], False, band,
                                     cells / max(1, prim["band_cells"]), {"wip": 2, "a0": 0, "a1": 1}[args.engines])
        else:
            tc = time.perf_counter()
            with mp.Pool(min(ncores, ns)) as pool:
                ccells = sum(pool.map(_cpu_align_one, [batch[i][:4] for i in range(ns)]))
            cdt = time.perf_counter() - tc
            used = min(ncores, ns)
            cpu_base = {"value": round(ccells / cdt / 1e9, 5), "unit": "GCUPS", "cores": used, "kind": "port",
                        "sample": f"first {ns} queries of the batch, oracle alignS_ng restatement "
                                  f"(UDH + slab tracebacks), one query per process on {used} cores"}
        if exact:
            eng_name = {"a0": "-A0: forwardS_ng / hirschbergS_ng / scorealoneS_ng as wavefront kernels (spdp_rowwave.hip)",
                        "a1": "-A1: forwardS1 / hirschbergS1 (spdp_exact.hip)"}[args.engines]
        out = {
            "metric": "GCUPS (DP cell updates/s), cDNA->genome spliced DP",
            "value": round(gcups, 3), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int32" if exact else "f32 (exact integers) / int32",
            "data": "synthetic",
            "config": {"workload": ("C4 shape (batch scaled to the run): 500-nt ESTs, 1 % error, vs the locus of "
                                    "the fragment +-1 kb (windows 2-10 kb), default band, Fwd2s1 _wip path: "
                                    "alignS_ng(ori=1, -Q0) = forward sweep + traceback walk, SKL out") if c4 else
                                   (f"C5: {len(batch)} cDNAs of 50 kb, 25 exons of ~2 kb, introns 1-10 kb, vs their loci +-1 kb (~190 kb), "
                                    "default band, Fwd2s1 _wip path: alignS_ng(ori=1, -Q0) = recursive linear-space branch "
                                    "(UDH sweeps over shrinking slabs) + slab tracebacks, SKL out") if args.workload == "c5" else
                                   ("C2: 10k x 2 kb cDNA vs planted loci +-1 kb (windows ~12 kb), "
                                    "default band, Fwd2s1 _wip path: alignS_ng(ori=1, -Q0) = UDH sweep + "
                                    "slab tracebacks, SKL out") if not exact else
                                   (f"C2 shape, {len(batch)} x 2 kb cDNA vs planted loci +-1 kb, alignS_ng(ori=1, -Q0) with the "
                                    f"exact-model engines ({eng_name})"),
                       "queries_per_gpu": len(batch), "queries_total": int(prim["total_queries"]),
                       "cells_per_gpu_per_step": int(cells),
                       "queries_per_s": round(prim["total_queries"] * args.steps / dt, 1),
                       **({"queries_per_s_incl_upload": round(len(batch) / ((prim["h2d"].get("streamed_step_ms") or
                                                                            prim["h2d"]["step_ms_incl_upload"]) * 1e-3), 1)}
                          if prim.get("h2d") else {}),
                       "reference_parity": _parity_note(args),
                       **({"with_h2d": {**prim["h2d"], "queries_per_s": round(len(batch) / (prim["h2d"]["step_ms_incl_upload"] * 1e-3), 1),
                                        "note": "step_ms_incl_upload: one step with the batch uploaded first (host arrays -> column records "
                                                "-> HBM), this rank; streamed_step_ms: steady state of a stream of batches, batch i + 1 uploaded "
                                                "on a second context of the device while batch i is aligned; the headline counts inputs resident "
                                                "in HBM"}} if prim.get("h2d") else {}),
                       "rank_busy_ms": prim["rank_busy_ms"],
                       "parallelism": f"{world} rank(s), one per GPU, queries sharded, no collective on the data path",
                       **({"strong_scaling" if other["scaling"] == "strong" else "weak_scaling": other} if other else {}),
                       "udh_cells": int(udh_cells), "fwd_cells": int(fwd_cells),
                       "udh_ms": round(udh_ms, 3), "udh_gcups": round(udh_cells / udh_ms / 1e6, 2) if udh_ms else None,
                       "fwd_ms": round(fwd_ms, 3), "fwd_gcups": round(fwd_cells / fwd_ms / 1e6, 2) if fwd_ms else None,
                       "fwd_problems": int(stats[-1]["fwd_problems"]), "tb_bytes": int(stats[-1]["tb_bytes"])},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "bytes_per_cell": round(SURVEY_BYTES_PER_CELL[k_name], 4),
                         "bytes_per_cell_source": "SURVEY.md 8(d): (6 B window + 16 B boundary rows) / 64-row stripe"
                                                  + (" + 1 B traceback code" if k_name == "forward" else ""),
                         "layout_bytes_per_cell": round(BYTES_PER_CELL[k_name], 4),
                         "layout_achieved": round(achieved_layout, 2), "layout_frac": round(achieved_layout / HBM_PEAK_GBS, 5),
                         "traffic": _traffic(wl_key, k_key, args.queries, world)[0],
                         "traffic_source": _traffic(wl_key, k_key, args.queries, world)[1],
                         "kernel": ("spdp_rowwave_udh" if args.engines == "a0" else "spdp_exact<udh>") if exact else
                                   ("spdp_sweep_fp<FL_FORWARD>" if c4 else "spdp_sweep_fp<FL_UDH>"), "kernel_ms": round(k_ms, 3),
                         "valu": _valu_roofline(k_cells, wl_key, k_key, k_ms),
                         "note": ("exact-model engines: int32 scores, per-row donor lists; latency-bound chains of exec-masked regions, the tiles of a "
                                  "problem pipelined over waves; HBM fraction reported as asked; kernel_ms = mean duration per step summed "
                                  "over the step's launches of this kernel") if exact else
                                 "VALU-issue bound recurrence (integer scores carried as exact fp32); HBM fraction reported as asked; kernel_ms = mean duration per step summed over the step's launches of this kernel (one per pipelined chunk)"},
            "cpu_baseline": cpu_base,
        }
        if world == 1 and args.workload == "c2" and not exact and args.seeded_pairs > 0:
            leg = _seeded_leg(args.seeded_pairs)
            if leg:
                out["config"]["seeded_q7"] = leg
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # the other BASELINE configs and engines, each a run of this script on the now idle GPU (N = 1, default workload)
        want = args.legs
        if want == "auto":
            want = ",".join(LEGS) if (world == 1 and args.workload == "c2" and not exact) else "none"
        if want != "none":
            t_legs = time.perf_counter()
            for name in want.split(","):
                if name in LEGS:
                    out["config"][name] = _run_leg(name)
            out["config"]["legs_wall_s"] = round(time.perf_counter() - t_legs, 1)
        _write_full_record(out)
        print(compact_line(out), flush=True)


if __name__ == "__main__":
    main()
