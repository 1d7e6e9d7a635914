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


# ---- the translated index (`spaln -W -KP`, <db>.bkp): spdp_blk_index_build_p --------------------------------------------------------
from tests.test_oracle_blkidx import IDXP, acomp_of, genome_of_golden_p  # noqa: E402


def params_p(eng, f, threaded, minorf):
    w = f["wcp"]
    p = blocks.build_params_default_p(eng.lib, 1 << 20, threaded, acomp=acomp_of(w[0]))
    p.b.ktuple, p.b.nshift, p.b.blklen, p.b.maxgene, p.b.afact, p.b.bitpat = w[1], w[5], w[6], w[7], w[9], w[4]
    p.nalpha, p.minorf = w[0], minorf
    for i, v in enumerate(f["conv"]):
        p.convtab[i] = int(v)
    for i in (0, 1, 2):                                   # entries the reference never sets (heap contents)
        p.convtab[i] = w[0]
    return p


@pytest.mark.parametrize("name,threaded,minorf", IDXP + [("blk_p1", 0, 30)], ids=[c[0] for c in IDXP] + ["blk_p1"])
def test_translated_index_written_file_equals_the_reference_file(eng, name, threaded, minorf):
    """the reference's own <db>.bkp files (edge genome with and without -t, a twelve-class alphabet, the index of the protein
    block-search fixture): tables, header, and the file byte for byte but the five heap pointers, four bytes of struct padding and
    the ConvTab entries the reference leaves unset"""
    path = os.path.join(GOLDEN_DIR, name + (".bkp" if name == "blk_p1" else ".bkp.gz"))
    f = read_bkn(path)
    gen, off = genome_of("blk_p1", 30, 1200, True) if name == "blk_p1" else genome_of_golden_p(name)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ours.bkp")
        got, sec = blocks.build_index_p(eng, gen, off, params_p(eng, f, threaded, minorf), write_to=out)
        ours = bytearray(open(out, "rb").read())
    for a, b in (("blk_nblk", "nblk"), ("blk_wscr", "wscr"), ("blk_blkp", "blkp"), ("blk_blkb", "blkb"), ("blk_chr", "chr")):
        assert np.array_equal(np.asarray(got[a]).astype(np.int64), np.asarray(f[b]).astype(np.int64)), a
    ref = bytearray(f["raw"])
    assert len(ours) == len(ref)
    conv_at = len(ref) - int(f["conv_ts"])
    for i in (0, 1, 2):
        ours[conv_at + i] = ref[conv_at + i] = 0
    ref[36 + 48:36 + 88] = bytes(40)                      # the pointers (blk_p1.bkp still holds its writer's)
    ref[40:44] = bytes(4)                                 # the padding behind ContBlk::ConvTS: the writer's heap as well
    assert ours[36 + 48:36 + 88] == bytes(40) and ours[40:44] == bytes(4)
    assert bytes(ours) == bytes(ref)
    if name == "blk_p1":                                  # the search parameters derived from the built index = those the reference derived
        from oracle import blk
        fx = spdg.load([x for x in golden_files("blk_") if x.endswith("blk_p1.spdg")][0])
        got2, _ = blocks.build_index_p(eng, gen, off, params_p(eng, f, threaded, minorf),
                                       max_intron_len=int(fx["blk_prm"][blk.PRM["extblock"]] - 1) * int(f["wcp"][6]),
                                       max_out=int(fx["blk_prm"][blk.PRM["ncand"]]) - 10)
        want = np.asarray(fx["blk_prm"], dtype=np.int32)
        for key, pos in blocks._PRM.items():
            if key not in ("extblock",):
                assert int(got2["blk_prm"][pos]) == int(want[pos]), key
        for k in ("blk_bitpat", "blk_rscrtab", "blk_pb2c"):
            assert np.array_equal(np.asarray(got2[k]).astype(np.int64), np.asarray(fx[k]).astype(np.int64)), k


@pytest.mark.parametrize("lens,k,nshift,blklen,minorf", [
    ([70000, 50000, 90000], 4, 4, 1024, 30), ([30000, 12345, 7, 1100], 3, 3, 512, 30), ([40000], 4, 1, 256, 30), ([60000, 100], 5, 2, 1024, 30),
    ([50000], 4, 3, 1024, 45), ([20000, 20000], 4, 4, 300, 9), ([2_000_000, 1_000_000, 1_500_000], 5, 5, 2048, 30), ([300_000], 6, 6, 4096, 60)])
def test_translated_index_against_the_oracle(eng, lens, k, nshift, blklen, minorf):
    """random genomes with ambiguous runs, both block walks, word lengths 3 .. 6, word steps 1 .. 6, MinOrf 9 .. 60"""
    from oracle import blk
    from spaln_amd import defaults
    rng = np.random.default_rng(1000 + k + nshift)
    parts = []
    for n in lens:
        s = np.array([2, 3, 5, 9], dtype=np.uint8)[rng.integers(0, 4, size=n)]
        for _ in range(3 if n > 300 else 0):
            a = int(rng.integers(0, n - 20))
            s[a:a + int(rng.integers(1, 40))] = 15
        parts.append(s)
    gen = np.concatenate(parts)
    off = np.array([0] + list(np.cumsum(lens)), dtype=np.int64)
    for threaded in (0, 1):
        p = blocks.build_params_default_p(eng.lib, 1 << 20, threaded)
        p.b.ktuple, p.b.nshift, p.b.blklen, p.b.bitpat, p.minorf = k, nshift, blklen, (1 << k) - 1, minorf
        got, _ = blocks.build_index_p(eng, gen, off, p)
        want = blk.index_build_tron(gen, off, blk.build_params_p(k, nshift, blklen, p.b.maxgene, 10, threaded, bytes(p.convtab)[:27],
                                                                defaults.BLOCK_ACOMP_20, minorf=minorf))
        for a, b in (("blk_nblk", "nblk"), ("blk_wscr", "wscr"), ("blk_blkp", "blkp"), ("blk_blkb", "blkb"), ("blk_chr", "chr")):
            assert np.array_equal(np.asarray(got[a]).astype(np.int64), np.asarray(want[b]).astype(np.int64)), (a, threaded)
        assert want["word_no"] > 5000


def test_translated_index_refusals(eng):
    gen = np.array([2, 3, 5, 9], dtype=np.uint8)[np.random.default_rng(1).integers(0, 4, size=5000)]
    off = np.array([0, 5000], dtype=np.int64)
    for change in (dict(ktuple=8), dict(nbitpat=3), dict(blklen=40), dict(minorf=16, ktuple=3, nshift=3), dict(nalpha=21)):
        p = blocks.build_params_default_p(eng.lib, 1 << 20)
        for key, v in change.items():
            setattr(p if key in ("minorf", "nalpha") else p.b, key, v)
        if "ktuple" in change:
            p.b.bitpat = (1 << p.b.ktuple) - 1
        with pytest.raises(RuntimeError):
            blocks.build_index_p(eng, gen, off, p)


def test_the_reference_searches_proteins_with_our_index(eng):
    """`spaln -Q7` of the compiled reference with protein queries on a genome it formatted itself (-KP), then with its .bkp replaced
    by the file the library built from the same residues: the same records"""
    ref = os.path.join(ROOT, "oracle", "_ref", "spaln")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/spaln is not built")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import types
    import dropin_demo
    import e2e_q7
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(protein=True, genes=40, queries=200, threads=1)
        _, env = dropin_demo.make_dataset(td, args)
        run = lambda: subprocess.run([ref, "-Q7", "-O4", "-t1", "-dgnm", "q.fa"], cwd=td, env=env, capture_output=True, text=True)
        a = run()
        assert a.returncode == 0 and a.stdout.count("\n@") > 100, a.stderr[-300:]
        theirs = read_bkn(os.path.join(td, "gnm.bkp"))
        _, chroms = e2e_q7.read_fasta(os.path.join(td, "gnm.mfa"))
        gen = np.concatenate(chroms).astype(np.uint8)
        off = np.array([0] + list(np.cumsum([len(c) for c in chroms])), dtype=np.int64)
        prm = blocks.build_params_default_p(eng.lib, os.path.getsize(os.path.join(td, "gnm.mfa")), threaded=1)    # (make_dataset formats with -t)
        w = theirs["wcp"]
        assert (prm.nalpha, prm.b.ktuple, prm.b.nshift, prm.b.blklen, prm.b.maxgene, prm.b.afact, prm.b.bitpat) == (w[0], w[1], w[5], w[6], w[7], w[9], w[4])
        os.remove(os.path.join(td, "gnm.bkp"))
        got, _ = blocks.build_index_p(eng, gen, off, prm, write_to=os.path.join(td, "gnm.bkp"))
        assert np.array_equal(np.asarray(got["blk_blkb"]).astype(np.int64), theirs["blkb"].astype(np.int64))
        b = run()
        assert b.returncode == 0, b.stderr[-300:]
        assert b.stdout == a.stdout
