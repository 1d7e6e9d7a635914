#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU's share: --queries ESTs of 500 nt (a node's eighth of 1 M: 125 000) against the block index of
a --mb million residue genome (3 000: human scale), everything the library's own -- spdp_blk_index_build makes the index on the
device (five bit patterns at this size, as `spaln -W -KD` picks them), spdp_blk_find maps every EST to its candidate loci (vote and
HSP search on the device, the genome resident).  The genome is random sequence with --genes planted loci; an EST counts as mapped
when its first locus covers the gene it was cut from, on its strand.  One JSON line.  (Identity with the reference's block search
is the tests' business -- tests/test_gpu_blk*.py on its recorded runs; its index files up to 100 Mb: bench.py's blk leg.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaln_amd import abi, blocks, engine, synth  # noqa: E402
from tests import spdg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=3000)
    ap.add_argument("--chr", type=int, default=24)
    ap.add_argument("--genes", type=int, default=400)
    ap.add_argument("--queries", type=int, default=125000)
    ap.add_argument("--patterns", type=int, default=0, help="bit patterns of the index (0: by genome size, as the reference's formatter)")
    args = ap.parse_args()
    n = args.mb * 1_000_000
    rng = np.random.default_rng(synth.SEED + 3300)
    code = np.array([2, 3, 5, 9], dtype=np.uint8)
    t0 = time.perf_counter()
    gen = np.empty(n, dtype=np.uint8)
    for a in range(0, n, 1 << 28):
        b = min(n, a + (1 << 28))
        gen[a:b] = code[rng.integers(0, 4, size=b - a, dtype=np.uint8)]
    cuts = np.sort(rng.integers(0, n, size=args.chr - 1))
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    code_of = np.zeros(256, dtype=np.uint8)
    for ch, cd in zip(b"ACGTN", (2, 3, 5, 9, 16)):
        code_of[ch] = cd
    genes = [synth.make_gene(np.random.default_rng(synth.SEED + 4401 + i)) for i in range(args.genes)]
    where = []
    for g in genes:                                          # planted inside a chromosome, apart from each other
        while True:
            c = int(rng.integers(0, args.chr))
            lo, hi = int(off[c]) + 50000, int(off[c + 1]) - 50000 - len(g.window)
            if hi <= lo:
                continue
            at = int(rng.integers(lo, hi))
            if all(abs(at - w[2]) > 200000 for w in where):
                break
        gen[at:at + len(g.window)] = code_of[g.window]
        where.append((c, at - int(off[c]), at))
    frag = 500
    gi = rng.integers(0, args.genes, size=args.queries)
    queries, rc = [], rng.random(args.queries) < 0.5
    comp = np.zeros(32, dtype=np.uint8)
    for a_, b_ in ((2, 9), (9, 2), (3, 5), (5, 3), (16, 16)):
        comp[a_] = b_
    for i, g_idx in enumerate(gi):
        q = code_of[genes[g_idx].query]
        o = int(rng.integers(0, len(q) - frag))
        e = q[o:o + frag].copy()
        sub = rng.random(frag) < 0.01
        e[sub] = code[rng.integers(0, 4, size=int(sub.sum()))]
        queries.append(comp[e[::-1]] if rc[i] else e)
    gen_s = time.perf_counter() - t0
    eng = engine.Engine(0)
    npat = args.patterns or (5 if args.mb >= 2000 else 3 if args.mb >= 500 else 1)
    prm = blocks.build_params_default(eng.lib, int(n * 61 / 60) + 8 * args.chr, npat, threaded=1)
    blocks.build_index(eng, gen[:1 << 20], np.array([0, 1 << 20], dtype=np.int64), prm)
    t0 = time.perf_counter()
    built, bsec = blocks.build_index(eng, gen, off, prm, max_intron_len=13000)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    dix = blocks.BlockIndex(eng, built)
    upload_s = time.perf_counter() - t0
    fqx = spdg.load(os.path.join(ROOT, "tests", "golden", "q_c2_seed0.spdg"))
    fbx = spdg.load(os.path.join(ROOT, "tests", "golden", "blk_k1.spdg"))
    wmodel = abi.wilip_model_from_fixture(fbx)
    sc = spdg.scoring(fqx, intpen=np.ascontiguousarray(fbx["find_intpen"], dtype=np.int16), scalar_engines=1)
    fprm = blocks.find_params_from_fixture(fbx)
    fprm.phase1t = int(dix.desc.rbscons)
    blocks.find(dix, gen, off, wmodel, sc, fprm, queries[:512])           # (the genome goes to the device, code objects load)
    t0 = time.perf_counter()
    _, ms = dix.vote(queries, out_cap=768)
    vote_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    loci, status = blocks.find(dix, gen, off, wmodel, sc, fprm, queries)
    find_s = time.perf_counter() - t0
    ok = with_locus = 0
    for i, g_idx in enumerate(gi):
        if loci[i]:
            with_locus += 1
            L = loci[i][0]
            c, o, _ = where[g_idx]
            ok += L["chr"] == c and L["base"] <= o + 60000 and o <= L["base"] + L["len"] and L["rvs"] == int(rc[i])
    print(json.dumps({"what": "BASELINE configs[3], one GPU's share: ESTs mapped to candidate loci (spdp_blk_find) on an index the library "
                              "built itself (spdp_blk_index_build)",
                      "genome_nt": n, "chromosomes": args.chr, "queries": args.queries,
                      "index": {"ktuple": int(prm.ktuple), "nshift": int(prm.nshift), "blklen": int(prm.blklen), "patterns": int(prm.nbitpat),
                                "blocks": int(built["nseg"]) - 1, "postings": int(np.asarray(built["blk_blkb"]).size), "maxblk": int(built["maxblk"])},
                      "index_build_s": round(build_s, 2), "index_build_device_s": round(bsec[0], 2), "index_to_device_s": round(upload_s, 2),
                      "vote_kernel_ms": round(float(ms), 1), "vote_call_s": round(vote_s, 2), "vote_queries_per_s": round(args.queries / (float(ms) * 1e-3), 0),
                      "find_s": round(find_s, 2), "queries_per_s": round(args.queries / find_s, 0),
                      "with_a_locus": with_locus, "first_locus_covers_the_planted_gene_on_its_strand": int(ok),
                      "input_generation_s": round(gen_s, 1)}))
    dix.free()
    eng.close()


if __name__ == "__main__":
    main()
