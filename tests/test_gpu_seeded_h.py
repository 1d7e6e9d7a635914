"""The protein seeded path on the GPU (spdp_align_h_seeded): alignH_ng with algmode.qck = 1 .. 3 through the C ABI against the
reference's own seeded runs (the qh_* fixtures of `ref_dump -Q`: HSPs of geneorient(), the Wilip replies its walk
received, score and SKL under -A0 and -A2).  The host walk (spdp_seeded_walk_h.h) parks every lspH_ng / trcbkalignH_ng
call -- with and without introns, with and without a cut range -- and the protein ladder serves them on the device."""
import pytest

from spaln_amd import abi, engine
from tests import spdg
from oracle import seeded
from tests.conftest import golden_files
from tests.test_oracle_seeded_h import seeded_inputs_h, QH, UNDEFINED

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
@pytest.mark.parametrize("path", QH, ids=[f.split("/")[-1][:-5] for f in QH])
def test_seeded_alignment_equals_reference(eng, path, alg, eng_sel):
    fx = spdg.load(path)
    name = path.split("/")[-1][:-5]
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, alg)
    sc.scalar_engines = eng_sel
    ps = p._owner
    if (name, alg) in UNDEFINED:                          # the reference's own traceback is undefined: reported, not aligned
        res = eng.align_h_seeded(sc, sp, ps, [hsps if n else None], [lowest], [wl], allow_partial=True)
        assert res[0][0] == abi.NEVSEL and len(res[0][1]) == 0
        return
    res = eng.align_h_seeded(sc, sp, ps, [hsps if n else None], [lowest], [wl])
    scr, flat = _flat(res[0])
    assert scr == int(fx[f"seed_scr_A{alg}"][0])
    assert flat == fx[f"seed_skl_A{alg}"].tolist()
    # the phases the reference's walk wrote into its Exinon (skl_rngH_ng reads them next): handed out by spdp_seeded_phase_marks
    want = {int(n_): [int(a), int(b)] for n_, a, b in fx[f"seed_marks_A{alg}"].reshape(-1, 3)}
    assert seeded.marks_changed(fx, eng.seeded_phase_marks(0)) == want


def test_requests_of_all_kinds_reach_the_device(eng):
    """lspH_ng calls, tracebacks and tracebacks with a cut range are all served by device batches"""
    tot = {"lsp": 0, "trcbk": 0, "trcbk_cut": 0, "batches": 0}
    for path in QH:
        fx = spdg.load(path)
        sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
        sc.scalar_engines = 1
        eng.align_h_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
        st = eng.seeded_stats()
        for k in tot:
            tot[k] += st[k]
    assert tot["lsp"] > 10 and tot["trcbk"] > 5 and tot["trcbk_cut"] > 3 and tot["batches"] > 0


@pytest.mark.parametrize("alg,eng_sel", [(0, 1), (2, 0)])
def test_fixtures_with_one_parameter_set_as_one_batch(eng, alg, eng_sel):
    """queries of one parameter set in ONE call: their DP requests share device batches, results stay per query"""
    groups = {}
    for f in QH:
        fx = spdg.load(f)
        if (f.split("/")[-1][:-5], alg) in UNDEFINED:
            continue
        seedp = [int(x) for x in fx["seed_params"]]
        key = (seedp[0], tuple(seedp[3:]), tuple(int(x) for x in fx["params"][:9]), tuple(int(x) for x in fx["hparams"]),
               int(fx["params"][27]), int(fx["params"][28]))
        groups.setdefault(key, []).append(fx)
    assert max(len(v) for v in groups.values()) >= 3
    for fxs in groups.values():
        if len(fxs) < 2:
            continue
        ps = abi.ProblemSetH()
        hs, lv, wls = [], [], []
        for fx in fxs:
            spdg.problem_h(fx, ps)
            j, n = seeded.hsps_of(fx)
            hs.append(j if n else None)
            lv.append(int(fx["seed_params"][1]))
            wls.append(seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"]))
        longest = max(fxs, key=lambda f: len(f["intpen"]))       # the intron-length table must cover the longest window
        sc = spdg.scoring_h(longest)
        sc.scalar_engines = eng_sel
        sp = abi.seed_params_from_fixture(longest)
        res = eng.align_h_seeded(sc, sp, ps, hs, lv, wls)
        st = eng.seeded_stats()
        assert st["walks"] == len(fxs)
        for fx, r in zip(fxs, res):
            scr, flat = _flat(r)
            assert scr == int(fx[f"seed_scr_A{alg}"][0])
            assert flat == fx[f"seed_skl_A{alg}"].tolist()


@pytest.mark.parametrize("path", golden_files("qh_a1_"), ids=[f.split("/")[-1][:-5] for f in golden_files("qh_a1_")])
def test_seeded_alignment_under_a1_equals_reference(eng, path):
    """the protein -A1 engines (spdp_h_exact.hip) behind the walk: `ref_dump -Q -A 0,1,2` runs"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 1)
    sc.scalar_engines = 2
    res = eng.align_h_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
    scr, flat = _flat(res[0])
    assert scr == int(fx["seed_scr_A1"][0])
    assert flat == fx["seed_skl_A1"].tolist()


LIVE_H = golden_files("live_h_")


@pytest.mark.parametrize("path", LIVE_H, ids=[f.split("/")[-1][:-5] for f in LIVE_H])
def test_live_pairs_equal_reference(eng, path):
    """pairs out of whole-program runs where the drop-in once differed from the reference (tools/dumpq_case.py;
    live_h_q7555: one of 10 000 proteins under -Q7, an intron 2 nt to the left)"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
    sc.scalar_engines = 1
    res = eng.align_h_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
    scr, flat = _flat(res[0])
    assert scr == int(fx["seed_scr_A0"][0])
    assert flat == fx["seed_skl_A0"].tolist()
    # what the reference's walk left in its Exinon (the phases of the junctions it chose itself; skl_rngH_ng reads them):
    # the library hands the same marks out (spdp_seeded_phase_marks) instead of writing into the caller's arrays
    want = {int(n_): [int(a), int(b)] for n_, a, b in fx["seed_marks_A0"].reshape(-1, 3)}
    assert want, "the case was kept for its marks"
    assert seeded.marks_changed(fx, eng.seeded_phase_marks(0)) == want
