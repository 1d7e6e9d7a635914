"""The index builder on the device (spdp_blk_index_build; SURVEY 8 row f4, `spaln -W -KD`): the tables of the reference's own
index files -- the genomes of the block-search fixtures (serial block walk, one and five bit patterns), the files of
tests/golden/make_idx_goldens.py (threaded walk; chromosome lengths on the block boundaries, ambiguous residues) --, the
file it writes byte for byte against the reference's (but the five heap pointers and the three ConvTab entries the
reference leaves unset), larger random genomes against the oracle, and the compiled reference searching with OUR index."""
import gzip
import os
import subprocess
import tempfile

import numpy as np
import pytest

from spaln_amd import blocks
from tests import spdg
from tests.conftest import GOLDEN_DIR, golden_files
from tests.test_blk_find import CASES, genome_of
from tests.test_oracle_blkidx import IDX, genome_of_golden, read_bkn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def params(ktuple, nshift, blklen, maxgene, nbitpat, afact, bitpat, bitpat2, threaded):
    return blocks.BlkBuildParams(ktuple, nshift, blklen, maxgene, nbitpat, afact, bitpat, bitpat2, threaded)


@pytest.mark.parametrize("name,n_genes,seed,par", CASES, ids=[c[0] for c in CASES])
def test_tables_of_the_fixture_indexes(eng, name, n_genes, seed, par):
    from oracle import blk
    fx = spdg.load([f for f in golden_files("blk_") if f.endswith(name + ".spdg")][0])
    gen, off = genome_of(name, n_genes, seed, par)
    q = blk.build_params_of(fx, 0)
    got, _ = blocks.build_index(eng, gen, off, params(q.ktuple, q.nshift, q.blklen, q.maxgene, q.nbitpat, q.afact, q.bitpat, q.bitpat2, 0),
                                max_intron_len=int(fx["blk_prm"][blk.PRM["extblock"]] - 1) * q.blklen,
                                max_out=int(fx["blk_prm"][blk.PRM["ncand"]]) - 10)
    for k in ("blk_nblk", "blk_wscr", "blk_blkp", "blk_blkb", "blk_chr", "blk_bitpat", "blk_rscrtab", "blk_pb2c"):
        assert np.array_equal(np.asarray(got[k]).astype(np.int64), np.asarray(fx[k]).astype(np.int64)), k
    want = np.asarray(fx["blk_prm"], dtype=np.int32)
    for key, pos in blocks._PRM.items():                  # the search parameters derived from the built index = those the reference derived
        if key in ("extblock",):
            continue
        assert int(got["blk_prm"][pos]) == int(want[pos]), key
    assert got["blk_prm"][36] == want[36] and got["blk_prm"][37] == want[37]


@pytest.mark.parametrize("name", IDX)
def test_written_file_equals_the_reference_file(eng, name):
    f = read_bkn(os.path.join(GOLDEN_DIR, name + ".bkn.gz"))
    gen, off = genome_of_golden(name)
    w = f["wcp"]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ours.bkn")
        got, sec = blocks.build_index(eng, gen, off, params(w[1], w[5], w[6], w[7], w[8], w[9], w[4], w[2], 1 if name.endswith("_t4") else 0),
                                      write_to=path)
        ours = bytearray(open(path, "rb").read())
    ref = bytearray(f["raw"])
    assert len(ours) == len(ref)
    conv_at = len(ref) - 17
    for i in (0, 1, 16):                                  # ConvTab entries the reference never sets (heap contents)
        ours[conv_at + i] = ref[conv_at + i] = 0
    assert ours[36 + 48:36 + 88] == bytes(40)             # the pointers: zeros in ours; the golden's were zeroed when it was made
    assert bytes(ours) == bytes(ref)
    assert sec[2] > 0


@pytest.mark.parametrize("n_chr,total,k,nbit,threaded,blklen", [(3, 3_000_000, 9, 1, 0, 2048), (40, 2_500_000, 8, 5, 1, 1024),
                                                                (1, 5_000_000, 10, 3, 1, 4096), (200, 1_000_000, 7, 5, 0, 1024)])
def test_larger_genomes_against_the_oracle(eng, n_chr, total, k, nbit, threaded, blklen):
    from oracle import blk
    rng = np.random.default_rng(4100 + n_chr + k)
    code = np.array([2, 3, 5, 9], dtype=np.uint8)
    cuts = np.sort(rng.integers(0, total, size=n_chr - 1)) if n_chr > 1 else np.zeros(0, dtype=np.int64)
    off = np.concatenate([[0], cuts, [total]]).astype(np.int64)
    gen = code[rng.integers(0, 4, size=total)]
    gen[rng.integers(0, total, size=300)] = 16            # scattered N
    for at in rng.integers(0, total - 5000, size=12):     # and runs of them
        gen[at:at + int(rng.integers(2, 4000))] = 16
    p0 = blocks.build_params_default(eng.lib, total, nbit)
    assert p0.nbitpat == nbit
    # the default's patterns for this k (DefBitPat), other sizes as the case says
    pk = blocks.build_params_default(eng.lib, int(np.exp((k + 0.5) / 0.59)), nbit)
    assert pk.ktuple == k
    prm = params(k, k, blklen, 65536, nbit, 10, pk.bitpat, pk.bitpat2, threaded)
    got, sec = blocks.build_index(eng, gen, off, prm)
    want = blk.index_build(gen, off, blk.BuildParams(k, k, blklen, 65536, nbit, 10, pk.bitpat, pk.bitpat2, threaded))
    for a, b in (("blk_nblk", "nblk"), ("blk_wscr", "wscr"), ("blk_blkp", "blkp"), ("blk_blkb", "blkb"), ("blk_chr", "chr")):
        assert np.array_equal(np.asarray(got[a]).astype(np.int64), np.asarray(want[b]).astype(np.int64)), a
    assert want["word_no"] > 10000 and got["maxblk"] == want["maxblk"]


def test_the_reference_searches_with_our_index(eng):
    """`spaln -Q7` of the compiled reference on a genome it formatted itself, then with its .bkn replaced by the file the
    library built from the same residues: the same records"""
    ref = os.path.join(ROOT, "oracle", "_ref", "spaln")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/spaln is not built")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import types
    import dropin_demo
    import e2e_q7
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(protein=False, genes=40, queries=200, threads=1)
        _, env = dropin_demo.make_dataset(td, args)
        run = lambda: subprocess.run([ref, "-Q7", "-S1", "-O4", "-t1", "-dgnm", "q.fa"], cwd=td, env=env, capture_output=True, text=True)
        a = run()
        assert a.returncode == 0 and a.stdout.count("\n@") > 150
        theirs = read_bkn(os.path.join(td, "gnm.bkn"))
        _, chroms = e2e_q7.read_fasta(os.path.join(td, "gnm.mfa"))
        gen = np.concatenate(chroms).astype(np.uint8)
        off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
        prm = blocks.build_params_default(eng.lib, os.path.getsize(os.path.join(td, "gnm.mfa")), 1, threaded=1)    # (make_dataset formats with -t)
        w = theirs["wcp"]
        assert (prm.ktuple, prm.nshift, prm.blklen, prm.maxgene, prm.nbitpat, prm.afact, prm.bitpat, prm.bitpat2) == (w[1], w[5], w[6], w[7], w[8], w[9], w[4], w[2])
        os.remove(os.path.join(td, "gnm.bkn"))
        blocks.build_index(eng, gen, off, prm, write_to=os.path.join(td, "gnm.bkn"))
        b = run()
        assert b.returncode == 0, b.stderr[-300:]
        assert b.stdout == a.stdout
