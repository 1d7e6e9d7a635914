#!/usr/bin/env python3
"""The index builder at genome scale: spdp_blk_index_build on a random genome of --mb million residues in
--chr chromosomes, parameters as `spaln -W -KD` would pick them for a FASTA file of that size.  One JSON line.  (Identity with
the reference's tables is checked up to 100 Mb by bench.py's blk leg and the tests; here: the rate, and the oracle on the
first --check-mb million residues built with the same parameters when asked.)  --protein: the translated index of `spaln -W -KP`
(spdp_blk_index_build_p)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaln_amd import blocks, engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=1000)
    ap.add_argument("--chr", type=int, default=24)
    ap.add_argument("--patterns", type=int, default=1)
    ap.add_argument("--threaded", type=int, default=1)
    ap.add_argument("--check-mb", type=int, default=0)
    ap.add_argument("--protein", action="store_true")
    args = ap.parse_args()
    n = args.mb * 1_000_000
    rng = np.random.default_rng(77)
    code = np.array([2, 3, 5, 9], dtype=np.uint8)
    t0 = time.perf_counter()
    gen = np.empty(n, dtype=np.uint8)
    for a in range(0, n, 1 << 28):
        b = min(n, a + (1 << 28))
        gen[a:b] = code[rng.integers(0, 4, size=b - a, dtype=np.uint8)]
    for at in rng.integers(0, n - 100000, size=200):                 # assembly gaps
        gen[at:at + int(rng.integers(100, 50000))] = 16
    cuts = np.sort(rng.integers(0, n, size=args.chr - 1))
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    gen_s = time.perf_counter() - t0
    eng = engine.Engine(0)
    if args.protein:
        return main_protein(args, eng, gen, off, n, gen_s)
    prm = blocks.build_params_default(eng.lib, int(n * 61 / 60) + 8 * args.chr, args.patterns, threaded=args.threaded)
    blocks.build_index(eng, gen[:1 << 20], np.array([0, 1 << 20], dtype=np.int64), prm)
    t0 = time.perf_counter()
    built, sec = blocks.build_index(eng, gen, off, prm)
    call_s = time.perf_counter() - t0
    out = {"what": "spdp_blk_index_build on a random genome", "residues": n, "chromosomes": args.chr, "ktuple": int(prm.ktuple),
           "nshift": int(prm.nshift), "blklen": int(prm.blklen), "patterns": int(prm.nbitpat), "threaded_walk": int(prm.threaded),
           "blocks": int(built["nseg"]) - 1, "postings": int(built["blk_blkb"].size), "maxblk": int(built["maxblk"]),
           "call_s": round(call_s, 3), "device_s": round(sec[0], 3), "host_s": round(sec[1], 3),
           "python_copy_out_s": round(call_s - sec[2], 3), "residues_per_s": round(n / sec[2], 0), "input_generation_s": round(gen_s, 1)}
    if args.check_mb:
        from oracle import blk
        m = min(n, args.check_mb * 1_000_000)
        o2 = np.concatenate([off[off < m], [m]]).astype(np.int64)
        got, _ = blocks.build_index(eng, gen[:m], o2, prm)
        want = blk.index_build(gen[:m], o2, blk.BuildParams(prm.ktuple, prm.nshift, prm.blklen, prm.maxgene, prm.nbitpat, prm.afact,
                                                           prm.bitpat, prm.bitpat2, prm.threaded))
        out["oracle_check"] = {"residues": m, "identical": bool(all(np.array_equal(np.asarray(got[a]).astype(np.int64), np.asarray(want[b]).astype(np.int64))
                                                                   for a, b in (("blk_nblk", "nblk"), ("blk_wscr", "wscr"), ("blk_blkp", "blkp"),
                                                                                ("blk_blkb", "blkb"), ("blk_chr", "chr"))))}
    print(json.dumps(out))
    eng.close()


def main_protein(args, eng, gen, off, n, gen_s):
    prm = blocks.build_params_default_p(eng.lib, int(n * 61 / 60) + 8 * args.chr, threaded=args.threaded)
    blocks.build_index_p(eng, gen[:1 << 20], np.array([0, 1 << 20], dtype=np.int64), prm)
    t0 = time.perf_counter()
    built, sec = blocks.build_index_p(eng, gen, off, prm)
    call_s = time.perf_counter() - t0
    out = {"what": "spdp_blk_index_build_p (translated index) on a random genome", "residues": n, "chromosomes": args.chr, "ktuple": int(prm.b.ktuple),
           "nshift": int(prm.b.nshift), "blklen": int(prm.b.blklen), "threaded_walk": int(prm.b.threaded),
           "blocks": int(built["nseg"]) - 1, "postings": int(built["blk_blkb"].size),
           "call_s": round(call_s, 3), "device_s": round(sec[0], 3), "host_s": round(sec[1], 3),
           "python_copy_out_s": round(call_s - sec[2], 3), "residues_per_s": round(n / sec[2], 0), "input_generation_s": round(gen_s, 1)}
    if args.check_mb:
        from oracle import blk
        from spaln_amd import defaults
        m = min(n, args.check_mb * 1_000_000)
        o2 = np.concatenate([off[off < m], [m]]).astype(np.int64)
        got, _ = blocks.build_index_p(eng, gen[:m], o2, prm)
        t0 = time.perf_counter()
        want = blk.index_build_tron(gen[:m], o2, blk.build_params_p(prm.b.ktuple, prm.b.nshift, prm.b.blklen, prm.b.maxgene, prm.b.afact, prm.b.threaded,
                                                                    bytes(prm.convtab)[:27], defaults.BLOCK_ACOMP_20))
        out["oracle_check"] = {"residues": m, "oracle_s": round(time.perf_counter() - t0, 2),
                               "identical": bool(all(np.array_equal(np.asarray(got[a]).astype(np.int64), np.asarray(want[b]).astype(np.int64))
                                                     for a, b in (("blk_nblk", "nblk"), ("blk_wscr", "wscr"), ("blk_blkp", "blkp"),
                                                                  ("blk_blkb", "blkb"), ("blk_chr", "chr"))))}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
