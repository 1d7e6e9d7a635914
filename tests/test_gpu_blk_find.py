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
