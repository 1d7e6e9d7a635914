"""spdp_blk_find on the device + host (round 5): the vote of every query on the GPU, call after call, TestOutput's second
half and FindHsp with the library's own HSP search on the host threads in between -- the candidate loci the compiled
reference's findblock left for the same queries (tests/golden/blk_*.spdg, find_log: chromosome, strand, region, range,
score, HSPs), incl. the paralog genome under -M4 (two loci per query)."""
import numpy as np
import pytest

from spaln_amd import abi, blocks, defaults
from tests import spdg
from tests.conftest import golden_files
from oracle import blk
from tests.test_blk_find import CASES, PROTEIN_CASES, genome_of, parse_find

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name,n_genes,seed,par", CASES + PROTEIN_CASES, ids=[c[0] for c in CASES + PROTEIN_CASES])
def test_loci_equal_the_reference(eng, name, n_genes, seed, par):
    fx = spdg.load([f for f in golden_files("blk_") if f.endswith(name + ".spdg")][0])
    gen, off = genome_of(name, n_genes, seed, par)
    dix = blocks.BlockIndex(eng, fx)
    model = abi.wilip_model_from_fixture(fx)
    prm = blocks.find_params_from_fixture(fx)
    v = [int(x) for x in fx["find_prm"]]
    sc = defaults.scoring(intpen=np.ascontiguousarray(fx["find_intpen"], dtype=np.int16))
    sc.gop, sc.gep, sc.lgop, sc.lgep, sc.codonk1 = v[13], v[14], v[15], v[16], v[17]
    qs = blk.parse_log(fx)
    got, status = blocks.find(dix, gen, off, model, sc, prm, [q["codes"] for q in qs], [(q["left"], q["right"]) for q in qs])
    last = {}
    for r in parse_find(fx["find_log"]):
        last[r[0]] = r                                       # what the query's last TestOutput call left
    n_loci = 0
    for qi in range(len(qs)):
        want = last[qi][4] if qi in last else []
        g = [([d["chr"], 3 if d["rvs"] else 0, d["base"], d["len"], d["left"], d["right"], d["jscr"], len(d["hsps"]) - 1],
              [[int(x) for x in row[:3]] + [0 if k == len(d["hsps"]) - 1 else int(row[3])] + [int(row[4])] for k, row in enumerate(d["hsps"])])
             for d in got[qi]]
        assert g == want, (qi, g[:1], want[:1])
        assert (status[qi] > 0) == bool(want), (qi, int(status[qi]))
        n_loci += len(want)
    assert n_loci >= 15
    dix.free()


@pytest.mark.parametrize("name,n_genes,seed,par", CASES + PROTEIN_CASES, ids=[c[0] for c in CASES + PROTEIN_CASES])
def test_random_queries_device_path_equals_host_path(eng, name, n_genes, seed, par):
    """fragments, mutated copies, chimeras and noise made from the fixtures' queries: spdp_blk_find (vote and HSP search on the device,
    the genome resident) against the same logic served by the oracle's vote and the host form of the HSP search (oracle/blk_check.cpp:
    tests/test_blk_find.py pins that path to the reference's recorded runs) -- the loci of every query, HSP for HSP"""
    import ctypes as C
    from oracle import oracle
    from tests.test_blk_find import find_calls
    fx = spdg.load([f for f in golden_files("blk_") if f.endswith(name + ".spdg")][0])
    gen, off = genome_of(name, n_genes, seed, par)
    dix = blocks.BlockIndex(eng, fx)
    ix, keep = blk.index_of(fx)
    model = abi.wilip_model_from_fixture(fx)
    prm = blocks.find_params_from_fixture(fx)
    v = [int(x) for x in fx["find_prm"]]
    sc = defaults.scoring(intpen=np.ascontiguousarray(fx["find_intpen"], dtype=np.int16))
    sc.gop, sc.gep, sc.lgop, sc.lgep, sc.codonk1 = v[13], v[14], v[15], v[16], v[17]
    protein = name == "blk_p1"
    letters = np.array(list(range(3, 23)) + [2] if protein else [2, 3, 5, 9, 16], dtype=np.uint8)
    rng = np.random.default_rng(4242)
    pool = [q["codes"] for q in blk.parse_log(fx)]
    queries = []
    for t in range(160):
        a = pool[int(rng.integers(len(pool)))]
        lo = int(rng.integers(0, max(1, len(a) - 40)))
        b = a[lo:lo + int(rng.integers(40 if protein else 120, 500 if protein else 1500))].copy()
        if t % 7 == 3:                                        # a chimera of two queries
            c = pool[int(rng.integers(len(pool)))]
            b = np.concatenate([b[:len(b) // 2], c[:len(c) // 2]])
        hits = rng.random(b.size) < rng.choice([0.0, 0.03, 0.1, 1.0], p=[0.4, 0.3, 0.25, 0.05])
        b[hits] = rng.choice(letters, size=int(hits.sum()))
        queries.append(np.ascontiguousarray(b))
    got, status = blocks.find(dix, gen, off, model, sc, prm, queries)
    lib = C.CDLL(oracle._BLK_SO)
    prm_i = np.ascontiguousarray(fx["find_prm"], dtype=np.int32)
    ip = np.ascontiguousarray(fx["find_intpen"], dtype=np.int16)
    n_loci = 0
    for qi, b in enumerate(queries):
        recs = parse_find(find_calls(lib, fx, ix, keep, gen, off, model, prm_i, ip, dict(codes=b, left=0, right=len(b))))
        want = recs[-1][4] if recs else []
        g = [([d["chr"], 3 if d["rvs"] else 0, d["base"], d["len"], d["left"], d["right"], d["jscr"], len(d["hsps"]) - 1],
              [[int(x) for x in row[:3]] + [0 if k == len(d["hsps"]) - 1 else int(row[3])] + [int(row[4])] for k, row in enumerate(d["hsps"])])
             for d in got[qi]]
        assert g == want, (qi, len(b), g[:1], want[:1])
        n_loci += len(want)
    assert n_loci >= 60
    dix.free()
