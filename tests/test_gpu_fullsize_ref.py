"""The GPU path at BASELINE's headline size against the reference's own output (fixtures c2_seed*, c5_6kb =
ref_dump -A 0; see tests/test_oracle_fullsize.py for why only -A0 records exist at that size): the `_wip` ladder the
product runs by default finds the reference's -A0 gene (every exon boundary of it: exact counts per fixture; rescored total where
the corner lists coincide) and equals the oracle's int32 `_wip` ladder bit for bit."""
import pytest

from tests import spdg
from tests.conftest import golden_files
from tests.test_oracle_fullsize import exon_bounds, WIP_VS_A0

pytestmark = pytest.mark.gpu
BIG = golden_files("c2_") + golden_files("c5_")


def test_wip_ladder_vs_reference_A0_records():
    from spaln_amd import abi, engine
    from oracle import host_logic
    eng = engine.Engine(0)
    n_same = 0
    for path in BIG:
        fx = spdg.load(path)
        sc = spdg.scoring(fx)
        ps, p = spdg.problem(fx)
        (scr, skl), = eng.align_s(sc, ps)
        skl = skl.ravel().tolist()
        wscr, wskl = host_logic.align_s(sc, p, simd=2)
        assert scr == wscr and skl == (wskl or []), path            # bit-exact against the int32 restatement
        ref = fx["aln_skl_A0"].tolist()
        minl = fx["prm"]["minl"]
        ex_g, ex_r = exon_bounds(skl, minl), exon_bounds(ref, minl)
        ends_g = {x for e in ex_g for x in e}
        ends_r = [x for e in ex_r for x in e]
        name = path.split("/")[-1][:-5]
        assert (sum(x in ends_g for x in ends_r), len(ends_r), skl == ref) == WIP_VS_A0[name], path     # exact counts
        fs = fx["rng_fstat_A0"]
        (h, fst, recs), = eng.skl_rng_s(sc, ps, [skl], codonk1=fx["prm"]["codonk1"], minl=minl, jneibr=int(fs[6]), lsg=int(fs[7]))
        if skl == ref:                                              # same traceback -> the CLI's score and exon table
            assert h == int(fx["rng_scr_A0"][0]), path
            assert recs.tolist() == fx["rng_eij_A0"].reshape(-1, 21).tolist(), path
            n_same += 1
    eng.close()
    assert n_same == 3


def test_A0_ladder_equals_reference_at_full_size():
    """-A0 (scalar_engines = 1: scorealoneS_ng, hirschbergS_ng rounds, forwardS_ng slabs -- spdp_rowwave.hip) against the
    reference's own -A0 records at 2 kb and 6 kb: HomScoreS_ng, gsi->scr, the SKL, skl_rngS_ng's score and exon table"""
    import time
    from spaln_amd import engine
    eng = engine.Engine(0)
    for path in BIG:
        fx = spdg.load(path)
        sc = spdg.scoring(fx, scalar_engines=1)
        ps, p = spdg.problem(fx)
        assert int(eng.homscore_s(sc, ps)[0]) == int(fx["hom_scr_A0"][0]), path
        t0 = time.perf_counter()
        (scr, skl), = eng.align_s(sc, ps)
        dt = time.perf_counter() - t0
        skl = skl.ravel().tolist()
        assert scr == int(fx["aln_scr_A0"][0]), path
        assert skl == fx["aln_skl_A0"].tolist(), path
        fs = fx["rng_fstat_A0"]
        (h, fst, recs), = eng.skl_rng_s(sc, ps, [skl], codonk1=fx["prm"]["codonk1"], minl=fx["prm"]["minl"], jneibr=int(fs[6]), lsg=int(fs[7]))
        assert h == int(fx["rng_scr_A0"][0]) and recs.tolist() == fx["rng_eij_A0"].reshape(-1, 21).tolist(), path
        assert dt < 60, (path, dt)                    # one wave per problem: a 2 kb query well under a second
    eng.close()
