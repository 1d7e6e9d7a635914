"""Map and align inside the library, end to end (tools/e2e_q7.py; round 5): the reference's index file and the genome in,
spdp_blk_find -> candidate loci -> spdp_align_s_seeded with the library's own HSP search -> spdp_skl_rng_s, and the exon
tables in chromosome coordinates against `spaln -Q7 -S1 -O4` of the compiled reference (oracle/_ref/spaln: test
infrastructure, prebuilt) on the same synthetic genome and queries."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from spaln_amd import abi, blocks
from tests import spdg
from tests.conftest import golden_files
from oracle import blk
from tests.test_blk_find import CASES, genome_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ori", [1, 3])
def test_exon_tables_equal_the_reference_program(ori):
    """ori = 1: the queries as given, `spaln -Q7 -S1`; ori = 3: every other query reverse-complemented, both orientations
    tried (alignS_ng(.., 3) on every locus), spaln's default"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln")):
        pytest.skip("oracle/_ref/spaln is not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_q7.py"), "--queries", "300", "--genes", "60", "--ori", str(ori)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["reference_aligned"] == 300 and d["library_aligned"] == 300
    assert d["identical_exon_tables"] == 300, (d, r.stderr[-600:])
    assert d["query_reversed"] == (150 if ori == 3 else 0)


def test_protein_exon_tables_equal_the_reference_program():
    """protein queries against the translated index (`spaln -W -KP`): ONE spdp_map_align_h call -- block search, regions as tron codes
    and their signals, seeded alignment with the library's own HSP searches, the walks' junction phases, rescoring -- against
    `spaln -Q7 -O4` (BASELINE configs[0] / [2]'s whole path)"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln")):
        pytest.skip("oracle/_ref/spaln is not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_q7.py"), "--protein", "--queries", "300", "--genes", "60"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["reference_aligned"] == 300 and d["library_aligned"] == 300
    assert d["identical_exon_tables"] == 300, (d, r.stderr[-600:])
    assert d["index"]["built_by"] == "spdp_blk_index_build_p" and d["index"]["tables_identical_to_the_reference_file"] is True     # (the index searched is the library's own)


def test_protein_rescoring_window_changes_nothing():
    """spdp_skl_rng_h with the corners' span of every region on the device (the default) and with whole regions: the same exon tables"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln")):
        pytest.skip("oracle/_ref/spaln is not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_q7.py"), "--protein", "--queries", "200", "--genes", "40"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SPDP_RESCORE_WINDOW="0"))
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["identical_exon_tables"] == 200 and d["library_aligned"] == 200, d


@pytest.mark.parametrize("scout", ["0", "1", "2"])
def test_scout_pass_changes_nothing(scout, monkeypatch):
    """the walks of a seeded call as one run (0), with a scout run that hands the slow class of requests over without waiting (1),
    or every request (2: the default at this size): the same exon tables, those of the reference"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln")):
        pytest.skip("oracle/_ref/spaln is not built")
    env = dict(os.environ, SPDP_SEED_SCOUT=scout, SPDP_SEED_VERBOSE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_q7.py"), "--queries", "400", "--genes", "60", "--ori", "3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["identical_exon_tables"] == 400 and d["library_aligned"] == 400, d
    assert ("[seeded] scout:" in r.stderr) == (scout != "0")


COMP = np.arange(256, dtype=np.uint8)
for _a, _b in ((2, 9), (9, 2), (3, 5), (5, 3)):
    COMP[_a] = _b


@pytest.mark.parametrize("name,n_genes,seed,par", [c for c in CASES if c[0] in ("blk_par", "blk_k1")],
                         ids=[c[0] for c in CASES if c[0] in ("blk_par", "blk_k1")])
def test_one_call_equals_its_steps(name, n_genes, seed, par):
    """spdp_map_align_s against the entries it is made of, called one by one from here: spdp_blk_find, per locus the region
    and spdp_splice_signals, spdp_align_s_seeded, spdp_skl_rng_s, the locus with the highest fstat.val (the paralog genome
    under -M4 gives two loci per query to choose from)"""
    from spaln_amd import engine
    eng = engine.Engine(0)
    fx = spdg.load([f for f in golden_files("blk_") if f.endswith(name + ".spdg")][0])
    fq = spdg.load(os.path.join(ROOT, "tests", "golden", "q_c2_seed0.spdg"))
    gen, off = genome_of(name, n_genes, seed, par)
    dix = blocks.BlockIndex(eng, fx)
    # the intron-length limits are those of the PROGRAM's run the block fixture records (its HSP-search model holds them: minl, maxl,
    # llmt as IntronPenalty derived them for that genome), not the defaults of the alignment fixture's harness
    model = abi.wilip_model_from_fixture(fx)
    sigmodel = abi.signal_model_from_fixture(fq)
    prm = blocks.find_params_from_fixture(fx)
    sc = spdg.scoring(fq, intpen=np.ascontiguousarray(fx["find_intpen"], dtype=np.int16), scalar_engines=1, llmt=model.llmt, minl=model.minl)
    sp = abi.seed_params_from_fixture(fq)
    sp.minl, sp.ip_maxl = model.minl, model.maxl
    fs = fq["rng_fstat_A0"] if "rng_fstat_A0" in fq else [0, 0, 0, 0, 0, 0, 3, 1]
    rescore = (fq["prm"]["codonk1"], model.minl, int(fs[6]), int(fs[7]))
    queries = [q["codes"][q["left"]:q["right"]] for q in blk.parse_log(fx)]
    # ---- the steps
    loci, _ = blocks.find(dix, gen, off, model, sc, prm, queries)
    ps, owner, hs = abi.ProblemSet(), [], []
    for qi, ls in enumerate(loci):
        for L in ls:
            reg = gen[off[L["chr"]] + L["base"]:off[L["chr"]] + L["base"] + L["len"]]
            if L["rvs"]:
                reg = COMP[reg[::-1]]
            sg = eng.splice_signals(sigmodel, reg, L["left"], L["right"])
            ps.add(queries[qi], reg, sg["sig5"], sg["sig3"], 0, len(queries[qi]), L["left"], L["right"], (1, 1, 1, 1),
                   cano5=sg["cano5"], cano3=sg["cano3"], dinc=sg["dinc"])
            owner.append((qi, L))
            hs.append(L["hsps"])
    assert len(owner) > len(queries) or name != "blk_par"          # (more loci than queries: there is something to choose)
    res = eng.align_s_seeded(sc, sp, ps, hs, [0] * len(owner), model)
    rs = eng.skl_rng_s(sc, ps, [skl.ravel() for _, skl in res], codonk1=rescore[0], minl=rescore[1], jneibr=rescore[2], lsg=rescore[3])
    want = {}
    for i, ((scr, skl), (h, fst, recs)) in enumerate(zip(res, rs)):
        if not len(skl):
            continue
        qi, L = owner[i]
        if qi in want and want[qi]["val"] >= fst[4]:
            continue
        site = (lambda n, L=L: L["base"] + (L["len"] - n if L["rvs"] else n + 1))
        want[qi] = dict(chr=L["chr"], rvs=L["rvs"], score=h, val=fst[4],
                        exons=[(int(r[2]) + 1, int(r[3]), site(int(r[0])), site(int(r[1]) - 1)) for r in recs if int(r[0]) <= (1 << 30)])
    # ---- the one call
    sp.wilip = C.addressof(model)
    genes, seconds, rc = blocks.map_align(dix, gen, off, sc, sp, sigmodel, prm, rescore, queries)
    assert rc == 0 and len(genes) == len(queries)
    n_aligned = 0
    for qi, g in enumerate(genes):
        if qi not in want:
            assert g is None, qi
            continue
        n_aligned += 1
        assert g is not None, qi
        assert {k: g[k] for k in ("chr", "rvs", "score", "val", "exons")} == want[qi], (qi, g, want[qi])
        assert g["n_loci"] == sum(1 for i, (q, _) in enumerate(owner) if q == qi and len(res[i][1]))
    assert n_aligned >= 10
    dix.free()
    eng.close()
