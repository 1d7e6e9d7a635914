"""The index builder's restatement (oracle/spdp_oracle_blkidx.c: MakeBlk::idxblk / m_idxblk, Block::c2w, blkscrtab,
findChrBbound) against the reference's own index files: the tables `spaln -W -KD` wrote for the genomes of the block-search
fixtures (tests/golden/blk_*.spdg: the serial walk, one and five bit patterns) and the files of
tests/golden/make_idx_goldens.py (the threaded walk; chromosome lengths on the block boundaries, ambiguous residues)."""
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import blk
from tests import spdg
from tests.conftest import GOLDEN_DIR, golden_files
from tests.test_blk_find import CASES, CODE_OF, genome_of
from tests.golden import make_idx_goldens as mk


def read_bkn(path):
    """the reference's index file (a struct image of its 64-bit build; src/blksrc.cc:598-622) -> dict"""
    raw = (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")).read()
    wcp = struct.unpack_from("<8IhH", raw, 0)             # Nalpha Ktuple Bitpat2 TabSize BitPat Nshift blklen MaxGene Nbitpat afact
    o = 36
    conv_ts, _, word_no, word_sz, chr_no, glen, avr, maxblk, bytblk, ver = struct.unpack_from("<II4Q4H", raw, o)
    o += 88
    b2c = np.frombuffer(raw, np.float64, 3, o); o += 24
    chr_ = np.frombuffer(raw, np.uint32, 2 * (chr_no + 1), o); o += 8 * (chr_no + 1)
    tab = wcp[3]
    nblk = np.frombuffer(raw, np.uint16, tab, o); o += 2 * tab
    blkp = np.frombuffer(raw, np.uint32, tab, o); o += 4 * tab
    if bytblk == 2:
        blkb = np.frombuffer(raw, np.uint16, word_sz, o).astype(np.uint32); o += 2 * word_sz
    else:
        blkb = np.frombuffer(raw, np.uint32, word_no, o); o += 4 * word_no
    wscr = np.frombuffer(raw, np.int16, tab, o); o += 2 * tab
    conv = np.frombuffer(raw, np.uint8, conv_ts, o); o += conv_ts
    assert o == len(raw)
    return dict(wcp=wcp, conv_ts=conv_ts, word_no=word_no, word_sz=word_sz, n_chr=chr_no, glen=glen, avrscr=avr, maxblk=maxblk,
                bytblk=bytblk, ver=ver, b2c=b2c, chr=chr_, nblk=nblk, blkp=blkp, blkb=blkb, wscr=wscr, conv=conv, raw=raw)


def params_of_file(f, threaded):
    w = f["wcp"]
    return blk.BuildParams(w[1], w[5], w[6], w[7], w[8], w[9], w[4], w[2], threaded)


def same_tables(got, want):
    for k in ("nblk", "wscr", "blkp", "blkb", "chr"):
        assert np.array_equal(np.asarray(got[k]).astype(np.int64), np.asarray(want[k]).astype(np.int64)), k
    assert got["b2c"].tolist() == np.asarray(want["b2c"]).tolist()


def genome_of_golden(name):
    if name.startswith("idx_edge"):
        chroms = mk.edge_genome()
    else:
        import tests.golden.make_blk_goldens as mb
        chroms, _ = mb.genome_and_queries(*{"idx_k1_t4": (42, 2, 900), "idx_k3_t4": (28, 2, 950)}[name])
    conv = CODE_OF.copy()
    conv[ord("R")] = 1                                     # any residue code but A / C / G / T is an ambiguous residue
    gen = np.concatenate([conv[c] for c in chroms]).astype(np.uint8)
    off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
    return gen, off


@pytest.mark.parametrize("name,n_genes,seed,par", CASES, ids=[c[0] for c in CASES])
def test_serial_walk_equals_the_reference_tables(name, n_genes, seed, par):
    fx = spdg.load([f for f in golden_files("blk_") if f.endswith(name + ".spdg")][0])
    gen, off = genome_of(name, n_genes, seed, par)
    got = blk.index_build(gen, off, blk.build_params_of(fx, 0))
    want = {k: fx["blk_" + k] for k in ("nblk", "wscr", "blkp", "blkb", "chr")}
    want["b2c"] = np.frombuffer(np.asarray(fx["blk_pb2c"], dtype=np.uint8).tobytes(), dtype=np.float64)
    same_tables(got, want)
    v = fx["blk_prm"]
    assert (got["word_no"], got["avrscr"], got["maxblk"]) == (int(v[blk.PRM["wordno"]]), int(v[blk.PRM["avrscr"]]), int(v[blk.PRM["maxblk"]]))
    other = blk.index_build(gen, off, blk.build_params_of(fx, 1))        # the threaded walk's blocks hold more words
    assert other["word_no"] > got["word_no"]


IDX = ["idx_k1_t4", "idx_k3_t4", "idx_edge_t0", "idx_edge_t4"]


@pytest.mark.parametrize("name", IDX)
def test_index_files_of_the_reference(name):
    f = read_bkn(os.path.join(GOLDEN_DIR, name + ".bkn.gz"))
    gen, off = genome_of_golden(name)
    got = blk.index_build(gen, off, params_of_file(f, 1 if name.endswith("_t4") else 0))
    same_tables(got, f)
    assert (got["word_no"], got["glen"], got["avrscr"], got["maxblk"], got["bytblk"]) == (f["word_no"], f["glen"], f["avrscr"], f["maxblk"], f["bytblk"])
    assert f["ver"] == 26 and f["n_chr"] == len(off) - 1


def test_default_parameters_are_the_reference_choice():
    """spdp_blk_build_params_default (host code of the product, no device needed) against the BlkWcPrm the reference's
    setupbitpat wrote into its files; the FASTA sizes are those tests/golden/make_idx_goldens.py printed"""
    import ctypes as C
    from spaln_amd import blocks, engine
    lib = C.CDLL(engine.LIB_PATH)
    for name, size, nbit in (("idx_k1_t4", 524081, 1), ("idx_k3_t4", 359387, 5), ("idx_edge_t0", 85346, 5)):
        w = read_bkn(os.path.join(GOLDEN_DIR, name + ".bkn.gz"))["wcp"]
        p = blocks.build_params_default(lib, size, nbit)
        assert (4, p.ktuple, p.bitpat2, 1 << (2 * p.ktuple), p.bitpat, p.nshift, p.blklen, p.maxgene, p.nbitpat, p.afact) == tuple(w), name
    with pytest.raises(RuntimeError):
        blocks.build_params_default(lib, 0)


# ---- the translated index (`spaln -W -KP`, <db>.bkp): oracle/spdp_oracle_blkidx.c orc_blk_index_build_tron ------------------------
IDXP = [("idxp_edge_t0", 0, 30), ("idxp_edge_t3", 1, 30), ("idxp_a12_t2", 1, 21)]     # (file, the threaded walk, MinOrf)


def acomp_of(nalpha):
    import json
    return [float.fromhex(x) for x in json.load(open(os.path.join(GOLDEN_DIR, "idxp_acomp.json")))[str(nalpha)]]


def genome_of_golden_p(name):
    chroms = mk.P_CASES[name][0]()
    conv = CODE_OF.copy()
    conv[ord("R")] = 1
    gen = np.concatenate([conv[c] for c in chroms]).astype(np.uint8)
    off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
    return gen, off


def params_of_file_p(f, threaded, minorf):
    w = f["wcp"]
    return blk.build_params_p(w[1], w[5], w[6], w[7], w[9], threaded, f["conv"], acomp_of(w[0]), nalpha=w[0], minorf=minorf)


@pytest.mark.parametrize("name,threaded,minorf", IDXP, ids=[c[0] for c in IDXP])
def test_translated_index_files_of_the_reference(name, threaded, minorf):
    """edge genome: chromosomes of 1 .. 5000 residues around the block boundaries (a last block whose ring spills into one block
    more), ambiguous runs, with and without -t; a twelve-class alphabet with 3-residue words every 2 codons and MinOrf 21"""
    f = read_bkn(os.path.join(GOLDEN_DIR, name + ".bkp.gz"))
    gen, off = genome_of_golden_p(name)
    got = blk.index_build_tron(gen, off, params_of_file_p(f, threaded, minorf))
    same_tables(got, f)
    assert (got["word_no"], got["glen"], got["avrscr"], got["bytblk"]) == (f["word_no"], f["glen"], f["avrscr"], f["bytblk"])
    assert f["maxblk"] == 65535                      # ContBlk::MaxBlk is never set on this path of the reference
    assert f["ver"] == 26 and f["n_chr"] == len(off) - 1
    other = blk.index_build_tron(gen, off, params_of_file_p(f, 1 - threaded, minorf))       # the two walks differ
    assert other["word_no"] != got["word_no"]


def test_translated_index_of_the_protein_fixture():
    """tests/golden/blk_p1.bkp: the index the protein block-search fixture was searched in (`spaln -W -KP`, no -t)"""
    f = read_bkn(os.path.join(GOLDEN_DIR, "blk_p1.bkp"))
    gen, off = genome_of("blk_p1", 30, 1200, True)
    from spaln_amd import defaults
    assert defaults.BLOCK_ACOMP_20 == acomp_of(20)
    got = blk.index_build_tron(gen, off, params_of_file_p(f, 0, 30))
    same_tables(got, f)
    assert (got["word_no"], got["avrscr"]) == (f["word_no"], f["avrscr"])


def test_default_parameters_of_a_translated_index():
    """spdp_blk_build_params_default_p against the BlkWcPrm and ConvTab of the reference's files (FASTA sizes as
    tests/golden/make_idx_goldens.py printed them; blk_p1's genome file: 129 632 bytes)"""
    import ctypes as C
    from spaln_amd import blocks, engine
    lib = C.CDLL(engine.LIB_PATH)
    for path, size in ((os.path.join(GOLDEN_DIR, "idxp_edge_t0.bkp.gz"), 32797), (os.path.join(GOLDEN_DIR, "blk_p1.bkp"), None)):
        f = read_bkn(path)
        w = f["wcp"]
        if size is None:
            gen, off = genome_of("blk_p1", 30, 1200, True)
            size = sum(len(f">chr{c + 1}\n") + int(off[c + 1] - off[c]) + (int(off[c + 1] - off[c]) + 59) // 60 for c in range(len(off) - 1))
        p = blocks.build_params_default_p(lib, size)
        assert (p.nalpha, p.b.ktuple, p.b.bitpat2, 20 ** p.b.ktuple, p.b.bitpat, p.b.nshift, p.b.blklen, p.b.maxgene, p.b.nbitpat, p.b.afact) == tuple(w), path
        conv = bytes(p.convtab)[:p.convts]
        assert p.convts == f["conv_ts"] and list(conv[3:25]) + [conv[26]] == f["conv"][3:25].tolist() + [int(f["conv"][26])]
