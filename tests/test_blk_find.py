"""From the vote to candidate loci (spaln_amd/csrc/spdp_loci.h: TestOutput's second half and FindHsp as a machine advanced with
search answers) against the reference, on the CPU: the oracle's vote, the product's HSP search in its host form
(spdp_hsp_host.h, level -1) and the product's locus logic, compiled by the host compiler into the tests' checker, run every query of
the blk_* fixtures call after call as findblock does -- and leave, at every TestOutput call, what the compiled reference left there (oracle/ref_build/blk_tap.cc,
snap_find): critjscr, the candidate block pairs as FindHsp moved their ends, and the candidate loci (chromosome, strand,
region, range, score, HSPs).  blk_par: every gene twice in the genome, -M4 -- two loci per query, their overlap / order /
pruning rules.  blk_p1: protein queries against the translated index (-KP) -- the region as tron codes, the retry with a grown
region (SrchBlk's DvsP = 1 branch), short queries whose cut-offs Wilip scales."""
import ctypes as C

import numpy as np
import pytest

from spaln_amd import abi
from tests import spdg
from tests.conftest import golden_files
from oracle import blk, oracle
from tests.golden import make_blk_goldens as mb

CASES = [("blk_k1", 42, 900, False), ("blk_k3", 28, 950, False), ("blk_par", 24, 980, True)]      # nucleotide queries (also the index builder's and the map + align tests' cases)
PROTEIN_CASES = [("blk_p1", 30, 1200, True)]                                                      # protein queries, the translated index (-KP)
CODE_OF = np.zeros(256, dtype=np.uint8)
for _ch, _code in zip(b"ACGTN", (2, 3, 5, 9, 16)):
    CODE_OF[_ch] = _code


def parse_find(L):
    """find_log -> [(query, call, critjscr, pairs (n, 10), [(header[9], hsps)])]"""
    L = [int(x) for x in L]
    i, recs = 0, []
    while i < len(L):
        assert L[i] == -4, (i, L[i])
        q, call, crit, npair = L[i + 1], L[i + 2], L[i + 3], L[i + 4]
        i += 5
        pairs = np.asarray(L[i:i + 10 * npair]).reshape(npair, 10)
        i += 10 * npair
        n = L[i]
        i += 1
        loci = []
        for _ in range(n):
            hd = L[i:i + 9]
            i += 9
            nj = hd[7] + 1 if hd[7] else 0
            jx = np.asarray(L[i:i + 5 * nj]).reshape(-1, 5).tolist()
            i += 5 * nj
            if jx:
                jx[-1][3] = 0                                    # (nid of the closing record: unset in the reference)
            loci.append((hd[:8], jx))
        recs.append((q, call, crit, pairs, loci))
    return recs


def genome_of(name, n_genes, seed, par):
    chroms, _ = (mb.protein_genome_and_queries if name == "blk_p1" else mb.paralog_genome_and_queries if par else mb.genome_and_queries)(n_genes, 2, seed)
    gen = np.concatenate([CODE_OF[c] for c in chroms]).astype(np.uint8)
    off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
    return gen, off


def find_calls(lib, fx, ix, keep, gen, off, model, prm, ip, q):
    """the block search of one query, call after call as findblock makes them: the oracle's vote up to each TestOutput call, the
    product's host logic (oracle/blk_check.cpp: spdp_loci.h + the host form of the HSP search) on its state; -> the log in the
    recorder's layout"""
    f = lib.loci_check_call
    f.restype = C.c_int
    codes = np.ascontiguousarray(q["codes"], dtype=np.uint8)
    chr_tab = np.ascontiguousarray(fx["blk_chr"], dtype=np.int32) if "blk_chr" in fx else None
    crit, verdict = C.c_int32(0), C.c_int32(0)
    out = []
    for call in range(64):
        v = blk.vote(ix, codes, q["left"], q["right"], call)
        if v is None:
            break
        forced = blk.last_vote_was_forced()
        w = blk.split_recorded(v[0], v[1])
        pairs = np.ascontiguousarray(v[1][2:].reshape(-1, 9), dtype=np.int32)
        runs = blk.runs_near_pairs(ix, w["runs"], pairs)
        flat = np.ascontiguousarray([x for d in range(4) for (b, s) in runs[d] for x in (b | d << 28, s)], dtype=np.int32)
        mmct = np.ascontiguousarray(w["head"][4:8], dtype=np.int32)
        log = np.zeros(1 << 16, dtype=np.int32)
        n = f(C.c_void_p(gen.ctypes.data), C.c_void_p(off.ctypes.data), C.c_int(len(off) - 1), C.c_void_p(ix.chr), C.c_void_p(ix.rscrtab),
              C.c_float(ix.rbscoef), C.c_float(ix.rbscons), C.c_int(ix.gdb), C.c_void_p(codes.ctypes.data), C.c_int(len(codes)),
              C.c_int(q["left"]), C.c_int(q["right"]), C.c_void_p(prm.ctypes.data), C.c_void_p(ip.ctypes.data), C.c_int(len(ip)),
              C.c_void_p(C.addressof(model)), C.c_void_p(pairs.ctypes.data), C.c_int(len(pairs)), C.c_void_p(mmct.ctypes.data),
              C.c_void_p(flat.ctypes.data), C.c_int(len(flat) // 2), C.c_int(int(forced)), C.c_int(call), C.byref(crit), C.byref(verdict),
              C.c_void_p(log.ctypes.data), C.c_int(len(log)))
        assert 0 <= n <= len(log)
        out.extend(log[:n].tolist())
        if verdict.value != 0:
            break
    return out


@pytest.mark.parametrize("name,n_genes,seed,par", CASES + PROTEIN_CASES, ids=[c[0] for c in CASES + PROTEIN_CASES])
def test_every_testoutput_call_equals_the_reference(name, n_genes, seed, par):
    fx = spdg.load([f for f in golden_files("blk_") if f.endswith(name + ".spdg")][0])
    gen, off = genome_of(name, n_genes, seed, par)
    ix, _keep = blk.index_of(fx)
    model = abi.wilip_model_from_fixture(fx)
    prm = np.ascontiguousarray(fx["find_prm"], dtype=np.int32)
    ip = np.ascontiguousarray(fx["find_intpen"], dtype=np.int16)
    by_q = {}
    for r in parse_find(fx["find_log"]):
        by_q.setdefault(r[0], []).append(r)
    lib = C.CDLL(oracle._BLK_SO)
    n_loci = two = 0
    for qi, q in enumerate(blk.parse_log(fx)):
        got, want = parse_find(find_calls(lib, fx, ix, _keep, gen, off, model, prm, ip, q)), by_q.get(qi, [])
        assert len(got) == len(want), (qi, len(got), len(want))
        for g, w in zip(got, want):
            np_ = g[3].shape[0]
            assert g[1] == w[1] and g[2] == w[2], (qi, g[1], g[2], w[2])
            assert g[3].tolist() == w[3][:np_].tolist(), (qi, g[1])
            assert g[4] == w[4], (qi, g[1], g[4][:1], w[4][:1])
            n_loci += len(g[4])
            two += len(g[4]) >= 2
    assert n_loci >= (30 if par else 15)
    if par:
        assert two >= (4 if name == "blk_p1" else 10)


def test_region_to_tron_codes_equals_the_references_nuc2tron():
    """blk_find::nuc2tron (what FindHsp's protein branch makes of a candidate region) against Seq::nuc2tron of the compiled
    reference: the windows of the protein fixtures (tests/golden/make_goldens.py regenerates the nucleotides; h1_*.spdg hold the
    tron codes the reference's harness made of them) -- ambiguous residues next to junctions, a random sequence, cut windows,
    the first and the last position (the codon reaches into the sequence's pads there)."""
    import os
    from tests.golden import make_goldens as mg
    lib = C.CDLL(oracle._BLK_SO)
    lib.blk_check_nuc2tron.argtypes = [C.c_void_p, C.c_int]
    n = amb = 0
    for name, (window, _q, _opts) in mg.protein_cases().items():
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".spdg")
        if not os.path.exists(path):
            continue
        want = np.asarray(spdg.load(path)["b_codes"], dtype=np.uint8)
        codes = np.ascontiguousarray(CODE_OF[np.asarray(window, dtype=np.uint8)])
        if len(want) != len(codes) + 1:
            continue                                             # (the fixture holds the sequence and the pad behind it)
        want = want[:-1]
        amb += int((codes == 16).sum())
        lib.blk_check_nuc2tron(codes.ctypes.data, len(codes))
        assert np.array_equal(codes, want), (name, np.nonzero(codes != want)[0][:8], codes[:4], want[:4])
        n += 1
    assert n >= 20 and amb >= 4
