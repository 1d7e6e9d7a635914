"""Protein queries through the block search (SrchBlk::findblock's DvsP = 1 branch): a synthetic genome with
planted protein genes on both strands, formatted by the compiled reference (`spaln -W -KP`: the amino-acid words of the
translated genome, <db>.bkp -- an INPUT, like the genome), the file read by the library (spdp_blk_index_read), then
spdp_blk_find for a batch of diverged proteins: vote on the device, TestOutput / FindHsp (region -> tron codes, HSP search,
retry on a grown region) on the host threads.  Reports queries/s and how many first loci cover the planted gene on its strand.
    python tools/blk_find_protein.py [--queries 20000] [--genes 200] [--chr-mb 10]
Needs oracle/_ref/spaln (built where /root/reference exists; it travels with the tree)."""
import argparse, ctypes as C, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from spaln_amd import abi, blocks, defaults, engine, synth
from tests import spdg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=20000)
    ap.add_argument("--genes", type=int, default=200)
    ap.add_argument("--chr-mb", type=float, default=10.0)
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref")
    rng = np.random.default_rng(synth.SEED + 7700)
    n_chr, chr_len = 2, int(a.chr_mb * 1e6)
    genes = [synth.make_protein_gene(np.random.default_rng(synth.SEED + 7701 + i), n_exons=5, aa_len=int(rng.integers(200, 500)), sub=0.0)
             for i in range(a.genes)]
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y
    td = tempfile.mkdtemp(prefix="spdp_blkp_")
    where, chroms = [], []
    per = a.genes // n_chr
    for c in range(n_chr):
        s = synth.random_dna(rng, chr_len)
        for k in range(per):
            g = genes[c * per + k]
            o = 50_000 + k * ((chr_len - 100_000) // per)
            rv = k & 1
            s[o:o + len(g.window)] = comp[g.window[::-1]] if rv else g.window
            where.append((c, o, len(g.window), rv))
        chroms.append(s)
    with open(os.path.join(td, "gnm.mfa"), "wb") as f:
        for c, s in enumerate(chroms):
            f.write(f">chr{c + 1}\n".encode())
            body = s[:chr_len // 60 * 60].reshape(-1, 60)
            out = np.empty((body.shape[0], 61), dtype=np.uint8); out[:, :60] = body; out[:, 60] = 10
            f.write(out.tobytes()); f.write(s[chr_len // 60 * 60:].tobytes() + b"\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "HSA_TOOLS", "LD_PRELOAD"))}
    env.update(ALN_TAB=os.path.join(ref, "table"), ALN_DBS=td)
    t0 = time.perf_counter()
    subprocess.run([os.path.join(ref, "spaln"), "-W", "-KP", "-t16", "gnm.mfa"], cwd=td, env=env, check=True, capture_output=True)
    fmt_s = time.perf_counter() - t0
    code_of = np.zeros(256, dtype=np.uint8)
    for ch, cd in zip(b"ACGTN", (2, 3, 5, 9, 16)):
        code_of[ch] = cd
    gcodes = np.concatenate([code_of[s] for s in chroms]); goff = np.array([0, chr_len, 2 * chr_len], dtype=np.int64)
    eng = engine.Engine(0)
    fbx = spdg.load(os.path.join(ROOT, "tests", "golden", "blk_p1.spdg"))           # Wilip tables, gap and intron penalties of a protein run
    theirs = blocks.read_index_file(eng.lib, os.path.join(td, "gnm.bkp"), max_intron_len=13000, max_out=1)
    # the index searched below is the library's own (spdp_blk_index_build_p), compared with the reference's file
    bprm = blocks.build_params_default_p(eng.lib, os.path.getsize(os.path.join(td, "gnm.mfa")), threaded=1)
    blocks.build_index_p(eng, gcodes[:1 << 16], np.array([0, 1 << 16], dtype=np.int64), bprm)
    t0 = time.perf_counter()
    _, bsec = blocks.build_index_p(eng, gcodes, goff, bprm, write_to=os.path.join(td, "ours.bkp"))
    build_s = time.perf_counter() - t0
    fx = blocks.read_index_file(eng.lib, os.path.join(td, "ours.bkp"), max_intron_len=13000, max_out=1)
    index_same = all(np.array_equal(np.asarray(fx[k]), np.asarray(theirs[k])) for k in ("blk_nblk", "blk_wscr", "blk_blkp", "blk_blkb", "blk_chr", "blk_prm"))
    fx["blk_convtab"][:2] = 255
    dix = blocks.BlockIndex(eng, fx)
    model = abi.wilip_model_from_fixture(fbx)
    v = [int(x) for x in fbx["find_prm"]]
    sc = defaults.scoring(intpen=np.ascontiguousarray(fbx["find_intpen"], dtype=np.int16))
    sc.gop, sc.gep, sc.lgop, sc.lgep, sc.codonk1 = v[13], v[14], v[15], v[16], v[17]
    prm = blocks.find_params_from_fixture(fbx)
    prm.phase1t = int(dix.desc.rbscons); prm.max_out = 1
    gi = rng.integers(0, a.genes, size=a.queries)
    queries = []
    for g_idx in gi:
        p = genes[g_idx].protein.copy()
        hit = rng.random(p.size) < 0.1
        p[hit] = synth._AA_LETTERS[rng.integers(0, 20, size=int(hit.sum()))]
        queries.append(synth.encode_protein(p))
    blocks.find(dix, gcodes, goff, model, sc, prm, queries[:256])                    # (code objects loaded)
    t0 = time.perf_counter()
    blocks.find(dix, gcodes, goff, model, sc, prm, queries)                          # (the waves' slabs sized for the batch: kept by the index object)
    first_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    loci, status = blocks.find(dix, gcodes, goff, model, sc, prm, queries)
    dt = time.perf_counter() - t0
    ok = with_locus = 0
    for i, g_idx in enumerate(gi):
        if loci[i]:
            with_locus += 1
            L = loci[i][0]; c, o, wl, rv = where[g_idx]
            ok += L["chr"] == c and L["rvs"] == rv and L["base"] < o + wl and o < L["base"] + L["len"]
    print(json.dumps({"what": "spdp_blk_find, protein queries (10 % substitutions) against the translated index the library built (compared with the file of the reference's own formatter)",
                      "genome_nt": int(gcodes.size), "genes": a.genes, "queries": a.queries, "with_a_locus": with_locus,
                      "first_locus_covers_the_planted_gene_on_its_strand": int(ok), "seconds": round(dt, 3), "first_call_s": round(first_s, 3),
                      "queries_per_s": round(a.queries / dt, 0), "reference_format_s": round(fmt_s, 2),
                      "index_build": {"by": "spdp_blk_index_build_p", "build_and_write_s": round(build_s, 3), "device_s": round(bsec[0], 3), "host_s": round(bsec[1], 3),
                                      "tables_identical_to_the_reference_file": bool(index_same)},
                      "index": {k: int(fx[k]) for k in ("nalpha", "tabsize", "nshift", "blklen", "nseg")}}))


if __name__ == "__main__":
    main()
