"""The seeded path on the GPU (spdp_align_s_seeded): alignS_ng with algmode.qck = 1 .. 3 through the C ABI
against the reference's own seeded runs (the q_* fixtures of `ref_dump -Q`: HSPs of geneorient(), the Wilip replies its
walk received, score and SKL under -A0 and -A2).  The host walk batches every lspS_ng / trcbkalignS_ng call of every
query in flight into common device launches; the fixtures run one by one and all together."""
import numpy as np
import pytest

from spaln_amd import abi, engine
from tests import spdg
from tests.conftest import golden_files, golden_ids
from tests.test_oracle_seeded import seeded_inputs, Q_FILES, O3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def _flat(res):
    scr, skl = res
    return scr, ([int(x) for x in skl.ravel()] if len(skl) else [])


@pytest.mark.parametrize("alg,eng_sel", [(0, 1), (2, 0)])
@pytest.mark.parametrize("path", Q_FILES, ids=[f.split("/")[-1][:-5] for f in Q_FILES])
def test_seeded_alignment_equals_reference(eng, path, alg, eng_sel):
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, alg)
    sc.scalar_engines = eng_sel
    ps = p._owner
    res = eng.align_s_seeded(sc, sp, ps, [hsps if n else None], [lowest], [wl])
    scr, flat = _flat(res[0])
    assert scr == int(fx[f"seed_scr_A{alg}"][0])
    assert flat == fx[f"seed_skl_A{alg}"].tolist()


@pytest.mark.parametrize("alg,eng_sel", [(0, 1), (2, 0)])
def test_whole_fixture_set_as_one_batch(eng, alg, eng_sel):
    """queries with one parameter set in ONE call: their DP requests share device batches, results stay per query"""
    groups = {}
    for f in Q_FILES:
        fx = spdg.load(f)
        seedp = [int(x) for x in fx["seed_params"]]
        key = (seedp[0], tuple(seedp[3:]), tuple(int(x) for x in fx["params"][:19]),     # ([1], [2]: per query: wllvl, #HSPs)
               int(fx["params"][27]), int(fx["params"][28]))
        groups.setdefault(key, []).append(fx)
    assert max(len(v) for v in groups.values()) >= 4
    n_dp = 0
    for fxs in groups.values():
        ps = abi.ProblemSet()
        sc = sp = None
        hs, lv, wls, keep = [], [], [], []
        for fx in fxs:
            sc = spdg.scoring(fx)
            sc.scalar_engines = eng_sel
            spdg.problem(fx, ps)
            p = ps.items[-1]
            h5, h3 = np.ascontiguousarray(fx["phs5"]), np.ascontiguousarray(fx["phs3"])
            keep += [h5, h3]
            p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
            sp = abi.seed_params_from_fixture(fx)
            from oracle import seeded
            j, n = seeded.hsps_of(fx)
            hs.append(j if n else None)
            lv.append(int(fx["seed_params"][1]))
            wls.append(seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"]))
        # the intron-length table must cover the longest window of the group
        longest = max(fxs, key=lambda f: len(f["intpen"]))
        sc = spdg.scoring(longest)
        sc.scalar_engines = eng_sel
        res = eng.align_s_seeded(sc, sp, ps, hs, lv, wls)
        st = eng.seeded_stats()
        n_dp += st["lsp"] + st["trcbk"]
        assert st["walks"] == len(fxs) and (st["batches"] <= st["lsp"] + st["trcbk"] or st["batches"] == 0)
        for fx, r in zip(fxs, res):
            scr, flat = _flat(r)
            assert scr == int(fx[f"seed_scr_A{alg}"][0])
            assert flat == fx[f"seed_skl_A{alg}"].tolist()
    assert n_dp > 50                     # the device was reached for the gaps between HSPs


def test_missing_hsp_source_is_reported(eng):
    """a walk that needs a recursion level without an HSP source comes back without an alignment, with return value 1"""
    fx = spdg.load([f for f in golden_files("q_") if f.endswith("q_0745.spdg")][0])
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 2)
    with pytest.raises(KeyError):
        eng.align_s_seeded(sc, sp, p._owner, [hsps], [lowest], [{}], allow_partial=True)


@pytest.mark.parametrize("alg,eng_sel", [(0, 1), (2, 0)])
def test_seeded_ori3_equals_reference(eng, alg, eng_sel):
    """alignS_ng(seqs, pwd, gsi, 3) with seeding on, all ori = 3 fixtures of one parameter set in one call: 2 n walks share
    the device batches; orientation, score, SKL and the A_RevCom bit as the reference returns them"""
    from oracle import seeded
    groups = {}
    for f in O3:
        fx = spdg.load(f)
        groups.setdefault(int(fx["seed_params"][0]), []).append(fx)
    for fxs in groups.values():
        psf, psr = abi.ProblemSet(), abi.ProblemSet()
        hs, lv, wf, wr, keep = [], [], [], [], []
        for fx in fxs:
            spdg.problem(fx, psf)
            spdg.problem_rev(fx, psr)
            for p, pre in ((psf.items[-1], ""), (psr.items[-1], "r_")):
                h5, h3 = np.ascontiguousarray(fx[pre + "phs5"]), np.ascontiguousarray(fx[pre + "phs3"])
                keep += [h5, h3]
                p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
            j, n = seeded.hsps_of(fx)
            hs.append(j if n else None)
            lv.append(int(fx["seed_params"][1]))
            wf.append(seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"], 0))
            wr.append(seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"], 1))
        sc = spdg.scoring(max(fxs, key=lambda f: len(f["intpen"])))
        sc.scalar_engines = eng_sel
        sp = abi.seed_params_from_fixture(fxs[0])
        res, orient = eng.align_s_seeded_ori3(sc, sp, psf, psr, hs, lv, wf + wr)
        for fx, r, o in zip(fxs, res, orient):
            scr, flat = _flat(r)
            assert o == int(fx[f"seed_rev_A{alg}"][0])
            assert scr == int(fx[f"seed_scr_A{alg}"][0])
            assert flat == fx[f"seed_skl_A{alg}"].tolist()


@pytest.mark.parametrize("path", golden_files("q_a1_"), ids=[f.split("/")[-1][:-5] for f in golden_files("q_a1_")])
def test_seeded_alignment_under_a1_equals_reference(eng, path):
    """the -A1 engines (spdp_exact.hip) behind the walk: `ref_dump -Q -A 0,1,2` runs"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 1)
    sc.scalar_engines = 2
    res = eng.align_s_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
    scr, flat = _flat(res[0])
    assert scr == int(fx["seed_scr_A1"][0])
    assert flat == fx["seed_skl_A1"].tolist()
