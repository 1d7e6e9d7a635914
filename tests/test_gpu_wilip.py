"""The seeded path with the library's own HSP search (SpdpSeedParams.wilip, spdp_wilip.h; round 5): no SpdpHspSource, no
recorded reply -- alignS_ng / alignH_ng with seeding on through the C ABI, every Wilip request of the recursion levels
answered inside the library, against the reference's own runs (score, corner list, protein phase marks)."""
import pytest

from spaln_amd import abi
from tests import spdg
from tests.conftest import golden_files
from oracle import seeded
from tests.test_oracle_seeded import seeded_inputs
from tests.test_oracle_seeded_h import seeded_inputs_h, UNDEFINED

pytestmark = pytest.mark.gpu

FILES = [f for f in golden_files("q_") + golden_files("ql3_") + golden_files("qh_") + golden_files("qhl3_")
         if "/q_o3_" not in f]


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("path", FILES, ids=_name)
def test_seeded_alignment_with_own_hsp_search(eng, path):
    fx = spdg.load(path)
    model = abi.wilip_model_from_fixture(fx)
    prot = "is_protein" in fx and int(fx["is_protein"][0]) == 1
    sc, sp, p, hsps, n, lowest, wl = (seeded_inputs_h if prot else seeded_inputs)(fx, 0)
    sc.scalar_engines = 1
    call = eng.align_h_seeded if prot else eng.align_s_seeded
    (scr, skl), = call(sc, sp, p._owner, [hsps if n else None], [lowest], model)
    assert scr == int(fx["seed_scr_A0"][0])
    assert ([int(x) for x in skl.ravel()] if len(skl) else []) == fx["seed_skl_A0"].tolist()
    st = eng.seeded_stats()
    assert st["wilip"] == sum(1 for _ in wl)                 # as many searches as the reference's walk made
    if prot:
        want = {int(n_): [int(a), int(b)] for n_, a, b in fx["seed_marks_A0"].reshape(-1, 3)}
        assert seeded.marks_changed(fx, eng.seeded_phase_marks(0)) == want


def test_batch_of_queries_with_own_hsp_search(eng):
    """queries of one parameter set in one call: the searches run on the walks' worker threads, side by side"""
    fxs = [spdg.load(f) for f in golden_files("q_") if "/q_o3_" not in f and "/q_a1_" not in f and "/q_cip" not in f]
    groups = {}
    for fx in fxs:
        key = (tuple(int(x) for x in fx["seed_params"][[0] + list(range(3, 20))]), tuple(int(x) for x in fx["params"][:9]),
               int(fx["params"][27]), int(fx["params"][18]), tuple(fx["wl_levels"].tolist()))
        groups.setdefault(key, []).append(fx)
    big = max(groups.values(), key=len)
    assert len(big) >= 5
    ps = abi.ProblemSet()
    hs, lv = [], []
    for fx in big:
        sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 0)
        spdg.problem(fx, ps)
        hs.append(hsps if n else None)
        lv.append(lowest)
    for p_, fx in zip(ps.items, big):                        # the walk reads the phase marks of its problem
        import numpy as np
        h5, h3 = np.ascontiguousarray(fx["phs5"]), np.ascontiguousarray(fx["phs3"])
        p_.phs5, p_.phs3 = h5.ctypes.data, h3.ctypes.data
        p_._phs = (h5, h3)
    longest = max(big, key=lambda f: len(f["intpen"]))
    sc = spdg.scoring(longest)
    sc.scalar_engines = 1
    res = eng.align_s_seeded(sc, abi.seed_params_from_fixture(longest), ps, hs, lv, abi.wilip_model_from_fixture(longest))
    for fx, (scr, skl) in zip(big, res):
        assert scr == int(fx["seed_scr_A0"][0])
        assert ([int(x) for x in skl.ravel()] if len(skl) else []) == fx["seed_skl_A0"].tolist()
