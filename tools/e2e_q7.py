#!/usr/bin/env python3
"""Map AND align inside the library, against the reference's own program (SURVEY 8 rows f4 + f2 + f3 end to end; round 5).

    python tools/e2e_q7.py [--queries 2000] [--genes 200]          (an MI355X box; oracle/_ref for the comparison)

The data set of tools/dropin_demo.py (a synthetic genome with planted multi-exon genes, formatted by the compiled
reference's own `spaln -W -KD`; cDNA queries = mutated transcripts).  Two runs:

  * reference:  oracle/_ref/spaln -Q7 -S1 -O4 -t<threads> -dgnm q.fa
  * library:    the reference's index file read by spdp_blk_index_read, the genome's residue codes, and then nothing of the
                reference: ONE spdp_map_align_s call = spdp_blk_find (vote on the device, TestOutput / FindHsp with the
                library's own HSP search on the host) -> candidate loci -> their regions and splice signals (one launch)
                -> spdp_align_s_seeded on every locus (the recursion levels searched by the library's own Wilip) ->
                spdp_skl_rng_s (exon table) -> the best locus of a query in chromosome coordinates.

Compared: per query the exon table (query range, chromosome range of every exon) of the best locus.  The parameter sets
(scoring, seeded walk, signal model, HSP-search model, block-search constants) are the reference's defaults as its own dumps
hold them (tests/golden/q_c2_seed0.spdg, blk_k1.spdg).  One JSON line."""
import argparse
import ctypes as C
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dropin_demo  # noqa: E402
from spaln_amd import abi, blocks, engine, synth  # noqa: E402
from tests import spdg  # noqa: E402

CODE_OF = np.zeros(256, dtype=np.uint8)
for _ch, _code in zip(b"ACGTNacgtn", (2, 3, 5, 9, 16, 2, 3, 5, 9, 16)):
    CODE_OF[_ch] = _code
COMP = np.arange(256, dtype=np.uint8)
for _a, _b in ((2, 9), (9, 2), (3, 5), (5, 3)):
    COMP[_a] = _b


def read_fasta(path):
    names, seqs, cur = [], [], []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                names.append(line[1:].split()[0].decode())
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return names, [CODE_OF[np.frombuffer(s, dtype=np.uint8)] for s in seqs]


def cli_parameters(td, env, extra):
    """the parameters the reference's PROGRAM holds for this data set -- the intron-length limits it derives from its length model
    (IntronPrm.minl / maxl / llmt: src/codepot.cc:134-135, 176-212), the IntPen table over every length, the HSP-search model and the
    block search's constants -- read from a one-query run of the recorder (oracle/_ref/spaln_blktap, oracle/ref_build/blk_tap.cc).  A
    binding reads them from the reference's objects; a harness that set its own defaults would hold other values (a fixture of
    ref_dump: minl 25, llmt 20, no maxl -- the program on a 5 Mb genome: 15, 15, 5 530)."""
    lines = open(os.path.join(td, "q.fa")).read().split(">")
    with open(os.path.join(td, "cli_one.fa"), "w") as f:
        f.write(">" + lines[1])
    log = os.path.join(td, "cli_one.spdg")
    r = subprocess.run([os.path.join(dropin_demo.REF, "spaln_blktap"), "-Q7", "-O4", "-t1"] + extra + ["-dgnm", "cli_one.fa"], cwd=td,
                       env=dict(env, SPDP_BLK_LOG=log), capture_output=True, text=True)
    if r.returncode or not os.path.exists(log):
        raise SystemExit("spaln_blktap failed: " + r.stderr[-300:])
    return spdg.load(log)


def reference_exons(text):
    """-O4 output -> {query: [(ref_l, ref_r, tgt_l, tgt_r)]} of the records as printed (the first locus of a query)"""
    out, cur = {}, []
    for line in text.splitlines():
        if line.startswith("#"):
            continue
        if line.startswith("@"):
            name = line.split()[7] if len(line.split()) > 7 else None
            m = re.search(r"\) (\S+) \[", line)
            name = m.group(1) if m else name
            out.setdefault(name, cur)
            cur = []
            continue
        f = line.split("\t")
        if len(f) >= 10:
            cur.append((int(f[6]), int(f[7]), int(f[8]), int(f[9])))
    return out


def main_protein(args):
    """BASELINE configs[0] / [2]'s whole path: protein queries, `spaln -Q7 -O4` against ONE spdp_map_align_h call"""
    t_all = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="spdp_e2e_p_") as td:
        genome_nt, env = dropin_demo.make_dataset(td, args)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(dropin_demo.REF, "spaln"), "-Q7", "-O4", f"-t{args.threads}", "-dgnm", "q.fa"], cwd=td, env=env,
                           capture_output=True, text=True)
        ref_s = time.perf_counter() - t0
        if r.returncode:
            raise SystemExit("reference run failed: " + r.stderr[-300:])
        want = reference_exons(r.stdout)
        eng = engine.Engine(0)
        lib = eng.lib
        cli = cli_parameters(td, env, [])
        model = abi.wilip_model_from_fixture(cli)
        # the alignment parameters of a protein run of the reference with the program's cross-species setting (a ref_dump fixture), the
        # intron-length limits and the IntPen table as the program holds them for THIS genome
        qh = "live_h_q7555.spdg" if model.crs else "qh_0013.spdg"      # (live_h_*: a pair recorded inside the program itself, oracle/ref_build/dumpq.cc)
        fq = spdg.load(os.path.join(ROOT, "tests", "golden", qh))
        assert int(fq["seed_params"][13]) == model.crs, (qh, model.crs)
        fsig = fq if "pm5_f32" in fq else spdg.load(os.path.join(ROOT, "tests", "golden", "h1_basic.spdg"))
        t0 = time.perf_counter()
        chr_names, chroms = read_fasta(os.path.join(td, "gnm.mfa"))
        gen = np.concatenate(chroms).astype(np.uint8)
        off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
        # the translated index: built by the library from the same residues (spdp_blk_index_build_p), compared with the reference's
        # file, and the one searched below
        theirs = blocks.read_index_file(lib, os.path.join(td, "gnm.bkp"), ext_block=int(cli["blk_prm"][blocks._PRM["extblock"]]))
        bprm = blocks.build_params_default_p(lib, os.path.getsize(os.path.join(td, "gnm.mfa")), threaded=1)        # (make_dataset formats with -t)
        blocks.build_index_p(eng, gen[:1 << 16], np.array([0, min(len(gen), 1 << 16)], dtype=np.int64), bprm)     # (first launch of the kernels)
        tb = time.perf_counter()
        _, bsec = blocks.build_index_p(eng, gen, off, bprm, write_to=os.path.join(td, "ours.bkp"))
        build_s = time.perf_counter() - tb
        fx = blocks.read_index_file(lib, os.path.join(td, "ours.bkp"), ext_block=int(cli["blk_prm"][blocks._PRM["extblock"]]))
        index_same = all(np.array_equal(np.asarray(fx[k]), np.asarray(theirs[k])) for k in ("blk_nblk", "blk_wscr", "blk_blkp", "blk_blkb", "blk_chr", "blk_prm"))
        fx["blk_convtab"][:2] = 255
        dix = blocks.BlockIndex(eng, fx)
        q_names, q_raw = [], []
        for blk_ in open(os.path.join(td, "q.fa")).read().split(">")[1:]:
            nm, seq = blk_.split("\n", 1)
            q_names.append(nm.split()[0]); q_raw.append(seq.replace("\n", ""))
        queries = [synth.encode_protein(np.frombuffer(s_.encode(), dtype=np.uint8)) for s_ in q_raw]
        ip = np.ascontiguousarray(cli["find_intpen"], dtype=np.int16)
        llmt, minl, _rlmt, maxl = (int(x) for x in cli["cli_intron_prm"][:4])
        sc = spdg.scoring_h(fq, intpen=ip, llmt=llmt, minl=minl)
        sc.scalar_engines = 1
        sp = abi.seed_params_from_fixture(fq)
        sp.qck, sp.minl, sp.ip_maxl = 3, minl, maxl
        sigmodel = abi.signal_model_h_from_fixture(fsig)
        prm = blocks.find_params_from_fixture(cli)
        prm.phase1t = int(dix.desc.rbscons)
        rp = [int(x) for x in fq["rparams"]]
        hp = dict(zip(spdg.HPARAM_NAMES, (int(x) for x in fq["hparams"])))
        rescore = abi.RescoreParamsH(minl, rp[4], hp["lcl"], rp[1])
        load_s = time.perf_counter() - t0
        sp.wilip = C.addressof(model)
        runs = []
        for _ in range(2):
            t0 = time.perf_counter()
            genes, phases, rc = blocks.map_align_h(dix, gen, off, sc, sp, sigmodel, prm, rescore, queries)
            runs.append((time.perf_counter() - t0, phases))
        lib_s, phases = runs[-1]
        got = {q_names[i]: g["exons"] for i, g in enumerate(genes) if g is not None}
        n_same = sum(1 for k, v in want.items() if got.get(k) == v)
        diff = [k for k, v in want.items() if got.get(k) != v]
        for k in diff[:args.show]:
            sys.stderr.write(f"{k}\n  reference {want[k]}\n  library   {got.get(k)}\n")
        if args.dump_diff:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_diff)), exist_ok=True)
            gi = {q_names[i]: g for i, g in enumerate(genes)}
            json.dump([{"name": k, "reference": want[k], "library": gi.get(k)} for k in diff], open(args.dump_diff, "w"), indent=1)
        print(json.dumps({"what": "protein queries: block search on the translated index -> HSPs -> seeded alignment -> exon table inside the library "
                                  "(spdp_map_align_h) against `spaln -Q7 -O4`",
                          "queries": args.queries, "genome_nt": genome_nt, "reference_aligned": len(want), "library_aligned": len(got),
                          "identical_exon_tables": n_same, "different": len(diff), "reference_wall_s": round(ref_s, 2), "reference_threads": args.threads,
                          "index": {"built_by": "spdp_blk_index_build_p", "tables_identical_to_the_reference_file": bool(index_same),
                                    "build_and_write_s": round(build_s, 3), "device_s": round(bsec[0], 3), "host_s": round(bsec[1], 3)},
                          "library_s": {"index_and_genome_load": round(load_s, 3), "map_align_call": round(lib_s, 3), "first_call": round(runs[0][0], 3), "first_call_phases": [round(x, 3) for x in runs[0][1]],
                                        "find": round(phases[0], 3), "regions_and_signals": round(phases[1], 3), "align": round(phases[2], 3),
                                        "rescore": round(phases[3], 3)},
                          "library_queries_per_s": round(len(got) / (load_s + lib_s), 1), "library_over_reference": round(ref_s / (load_s + lib_s), 2),
                          "return_code": rc, "wall_s": round(time.perf_counter() - t_all, 1)}))
        dix.free()
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--genes", type=int, default=200)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--show", type=int, default=3)
    ap.add_argument("--spacer", type=int, default=0, help="longest random stretch between two genes (default 20 000): the genome's size")
    ap.add_argument("--frag", type=int, default=0, help="queries are fragments of this length of the transcripts (ESTs, BASELINE configs[3])")
    ap.add_argument("--members", type=int, default=1,
                    help="> 1: spdp_group_map_align_s over that many members ON THIS ONE DEVICE (each member its own context, index and "
                         "chain of device batches: the members' request latencies overlap)")
    ap.add_argument("--ori", type=int, default=1, choices=[1, 3],
                    help="1: the queries as given against `spaln -S1`; 3: every other query reverse-complemented, both orientations "
                         "tried, against spaln's default (-S3)")
    ap.add_argument("--protein", action="store_true", help="protein queries against the translated index (spaln -W -KP): spdp_map_align_h")
    ap.add_argument("--dump-diff", default="", help="write the queries whose exon tables differ (name, both tables, the library's gene record) to this JSON file")
    args = ap.parse_args()
    if args.protein:
        return main_protein(args)
    t_all = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="spdp_e2e_") as td:
        genome_nt, env = dropin_demo.make_dataset(td, args)
        if args.ori == 3:                                        # antisense reads among the queries
            lines = open(os.path.join(td, "q.fa")).read().split("\n")
            comp = str.maketrans("ACGTacgt", "TGCAtgca")
            for i in range(2, len(lines) - 1, 4):
                lines[i + 1] = lines[i + 1].translate(comp)[::-1]
            open(os.path.join(td, "q.fa"), "w").write("\n".join(lines))
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(dropin_demo.REF, "spaln"), "-Q7"] + (["-S1"] if args.ori == 1 else []) + ["-O4", f"-t{args.threads}", "-dgnm", "q.fa"],
                           cwd=td, env=env, capture_output=True, text=True)
        ref_s = time.perf_counter() - t0
        if r.returncode:
            raise SystemExit("reference run failed: " + r.stderr[-300:])
        want = reference_exons(r.stdout)

        # ---- the library's run
        eng = engine.Engine(0)
        lib = eng.lib
        fq = spdg.load(os.path.join(ROOT, "tests", "golden", "q_c2_seed0.spdg"))
        cli = cli_parameters(td, env, ["-S1"] if args.ori == 1 else [])
        t0 = time.perf_counter()
        # ExtBlock = max_intron_len() / blklen + 1 follows from the program's IntronPrm.maxl (src/blksrc.cc:2071-2082, 2222): as recorded
        fx = blocks.read_index_file(lib, os.path.join(td, "gnm.bkn"), ext_block=int(cli["blk_prm"][blocks._PRM["extblock"]]))
        fx["blk_convtab"][:2] = 255
        dix = blocks.BlockIndex(eng, fx)
        chr_names, chroms = read_fasta(os.path.join(td, "gnm.mfa"))
        gen = np.concatenate(chroms).astype(np.uint8)
        off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
        q_names, queries = read_fasta(os.path.join(td, "q.fa"))
        sigmodel = abi.signal_model_from_fixture(fq)
        model = abi.wilip_model_from_fixture(cli)
        ip = np.ascontiguousarray(cli["find_intpen"], dtype=np.int16)
        llmt, minl, _rlmt, maxl = (int(x) for x in cli["cli_intron_prm"][:4])
        sc = spdg.scoring(fq, intpen=ip, scalar_engines=1, llmt=llmt, minl=minl)
        sp = abi.seed_params_from_fixture(fq)
        sp.minl, sp.ip_maxl = minl, maxl
        prm = blocks.find_params_from_fixture(cli)
        prm.phase1t = int(dix.desc.rbscons)              # Phase1T = (int) (RbsBias * avr), RbsBias = RbsBase = 3 (src/blksrc.cc:64-66)
        load_s = time.perf_counter() - t0
        sp.wilip = C.addressof(model)
        fs = fq["rng_fstat_A0"] if "rng_fstat_A0" in fq else [0, 0, 0, 0, 0, 0, 3, 1]
        rescore = (fq["prm"]["codonk1"], minl, int(fs[6]), int(fs[7]))
        # one call: spdp_blk_find -> regions and their signals (one launch) -> spdp_align_s_seeded -> spdp_skl_rng_s -> the
        # locus that stays.  Twice: the first call of a context also loads the kernels' code objects and sizes its pools
        runs = []
        if args.members > 1:
            grp = engine.Group([0] * args.members)
            glib = grp.lib
            glib.spdp_group_context.restype = C.c_void_p
            glib.spdp_group_context.argtypes = [C.c_void_p, C.c_int]

            class Member:                                    # what blocks.BlockIndex needs of an engine
                def __init__(self, ctx):
                    self.lib, self.ctx = glib, ctx

                def _check(self, rc, what):
                    assert rc == 0, what
            midx = [blocks.BlockIndex(Member(glib.spdp_group_context(grp.h, r)), fx) for r in range(args.members)]
            handles = (C.c_void_p * args.members)(*[i.h for i in midx])
            nq = len(queries)
            qoffs = np.zeros(nq + 1, dtype=np.int64)
            qoffs[1:] = np.cumsum([len(q) for q in queries])
            qcodes = np.ascontiguousarray(np.concatenate(queries))
            gg = blocks.Genome()
            goff = np.ascontiguousarray(off, dtype=np.int64)
            gg.codes, gg.chr_off, gg.n_chr = gen.ctypes.data, goff.ctypes.data, len(goff) - 1
            rp = abi.RescoreParams(*(int(x) for x in rescore))
            glib.spdp_group_map_align_s.restype = C.c_int
            glib.spdp_group_map_align_s.argtypes = [C.c_void_p] * 11 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
            libc = C.CDLL(None)
            libc.free.argtypes = [C.c_void_p]
            for _ in range(2):
                gs = (blocks.MapGene * nq)()
                ex = C.POINTER(blocks.MapExon)()
                t0 = time.perf_counter()
                rc = glib.spdp_group_map_align_s(grp.h, handles, C.byref(midx[0].desc), C.byref(gg), C.byref(sc), C.byref(sp), C.addressof(sigmodel),
                                                 C.byref(prm), C.byref(rp), qcodes.ctypes.data, qoffs.ctypes.data, nq, args.ori, gs, C.byref(ex))
                runs.append((time.perf_counter() - t0, [0, 0, 0, 0]))
                assert rc >= 0, glib.spdp_group_last_error(grp.h)
                genes = [None if gs[i].chr < 0 else dict(chr=gs[i].chr, rvs=gs[i].rvs, q_rev=gs[i].q_rev, score=gs[i].score, val=gs[i].val, n_loci=gs[i].n_loci,
                                                         exons=[(ex[gs[i].exon_off + j].q_left, ex[gs[i].exon_off + j].q_right, ex[gs[i].exon_off + j].g_left,
                                                                 ex[gs[i].exon_off + j].g_right) for j in range(gs[i].n_exons)]) for i in range(nq)]
                libc.free(ex)
            phases = runs[-1][1]
            for i in midx:
                i.free()
            grp.close()
        for _ in range(0 if args.members > 1 else 2):
            t0 = time.perf_counter()
            genes, phases, rc = blocks.map_align(dix, gen, off, sc, sp, sigmodel, prm, rescore, queries, ori=args.ori)
            runs.append((time.perf_counter() - t0, phases))
        lib_s, phases = runs[-1]
        got = {q_names[i]: g["exons"] for i, g in enumerate(genes) if g is not None}
        n_loci = sum(g["n_loci"] for g in genes if g is not None)
        n_same = sum(1 for k, v in want.items() if got.get(k) == v)
        diff = [k for k, v in want.items() if got.get(k) != v]
        for k in diff[:args.show]:
            sys.stderr.write(f"{k}\n  reference {want[k]}\n  library   {got.get(k)}\n")
        if args.dump_diff:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_diff)), exist_ok=True)
            gi = {q_names[i]: g for i, g in enumerate(genes)}
            json.dump([{"name": k, "reference": want[k], "library": gi.get(k)} for k in diff], open(args.dump_diff, "w"), indent=1)
        out = {"what": "block search -> HSPs -> seeded alignment -> exon table inside the library against `spaln -Q7 %s-O4`" % ("-S1 " if args.ori == 1 else ""),
               "queries": args.queries, "fragment_nt": args.frag or None, "ori": args.ori, "members": args.members, "query_reversed": sum(1 for g in genes if g is not None and g["q_rev"]), "genome_nt": genome_nt, "reference_aligned": len(want), "library_aligned": len(got),
               "identical_exon_tables": n_same, "different": len(diff),
               "reference_wall_s": round(ref_s, 2), "reference_threads": args.threads,
               "reference_queries_per_s": round(len(want) / ref_s, 1),
               "library_s": {"index_and_genome_load": round(load_s, 3), "map_align_call": round(lib_s, 3),
                             "first_call": round(runs[0][0], 3), "find": round(phases[0], 3),
                             "regions_and_signals": round(phases[1], 3), "align": round(phases[2], 3), "rescore": round(phases[3], 3)},
               "library_queries_per_s": round(len(got) / (load_s + lib_s), 1),
               "library_over_reference": round(ref_s / (load_s + lib_s), 2),
               "loci_aligned": n_loci, "return_code": rc, "wall_s": round(time.perf_counter() - t_all, 1),
               "note": "reference wall = its whole process (index + genome read, 16 threads); library = index + genome load "
                       "+ ONE spdp_map_align_s call on a warm context (the first call of the context beside it)"}
        print(json.dumps(out))
        dix.free()
        eng.close()


if __name__ == "__main__":
    main()
