"""The HSP search restated (spaln_amd/csrc/spdp_wilip.h: Wilip / Wlp of src/wln.cc; SURVEY 8 row f4, second slice) against the
reference, on the CPU: the product's header compiled into the tests' checker answers

  * every Wilip request the reference's own seeded walks made in the `ref_dump -Q` fixtures (recursion levels 1 and 2, cDNA
    and protein queries, plain and double affine gaps) -- with the end flags the product's walk holds at that call, which the
    recorded replies thereby pin as well (Wlp's end bonus reads a->inex.exgl / exgr);
  * the top-level search geneorient() made on the whole pair (the fixtures' seed_jxt: the best unit at the recorded level).

The model (word parameters per level, reduced alphabets, the HSP-search matrix, a few scalars) is the wl_* part of a fixture."""
import numpy as np
import pytest

from spaln_amd import abi
from tests import spdg
from tests.conftest import golden_files
from oracle import seeded
from tests.test_oracle_seeded import seeded_inputs
from tests.test_oracle_seeded_h import seeded_inputs_h

FILES = [f for f in golden_files("q_") + golden_files("ql3_") + golden_files("qh_") + golden_files("qhl3_")
         if "/q_o3_" not in f]


def _name(f):
    return f.split("/")[-1][:-5]


def _inputs(fx):
    prot = "is_protein" in fx and int(fx["is_protein"][0]) == 1
    return prot, (seeded_inputs_h(fx, 0) if prot else seeded_inputs(fx, 0))


@pytest.mark.parametrize("path", FILES, ids=_name)
def test_requests_of_the_walk_equal_the_reference(path):
    fx = spdg.load(path)
    model = abi.wilip_model_from_fixture(fx)
    prot, (sc, sp, p, hsps, n, lowest, wl) = _inputs(fx)
    tr = []
    (seeded.align_h_seeded if prot else seeded.align_s_seeded)(sc, sp, p, hsps, n, lowest, wl, 0, trace=tr)
    for kind, a, data in tr:
        if kind != 2:
            continue
        got = seeded.wilip(model, p, sc, a[14], a[:4], exg=(a[4], a[5]))
        assert seeded.same_units(got, list(data)), (a[14], a[:8], got[:30], list(data)[:30])


@pytest.mark.parametrize("path", FILES, ids=_name)
def test_top_level_search_equals_geneorient(path):
    fx = spdg.load(path)
    model = abi.wilip_model_from_fixture(fx)
    prot, (sc, sp, p, hsps, n, lowest, wl) = _inputs(fx)
    if n == 0:
        pytest.skip("geneorient found nothing on this strand")
    got = seeded.wilip(model, p, sc, lowest, (p.a_left, p.a_right, p.b_left, p.b_right), exg=(p.a_exgl, p.a_exgr))
    assert got[0] >= 1 and got[1] == n
    g = [got[7 + 5 * k:12 + 5 * k] for k in range(n + 1)]
    w = [list(x) for x in np.asarray(hsps).reshape(-1, 5).tolist()]
    g[-1][3] = w[-1][3] = 0                                 # (nid of the closing record: unset in the reference)
    assert g == w


def test_requests_are_many_and_of_both_levels():
    seen = {}
    for path in FILES:
        fx = spdg.load(path)
        prot, (sc, sp, p, hsps, n, lowest, wl) = _inputs(fx)
        for (level, *_), flat in wl.items():
            seen[(prot, level, flat[0] > 0)] = seen.get((prot, level, flat[0] > 0), 0) + 1
    assert sum(seen.values()) >= 150
    for prot in (False, True):
        for level in (1, 2):
            assert seen.get((prot, level, True), 0) >= 3, (prot, level, seen)


def test_walk_on_its_own_hsp_search_equals_reference():
    """the seeded walk with NO recorded reply: every Wilip request answered by the restated search (the checker binds the
    walk's HSP callback to it) -- score and corner list of the reference"""
    n_ok = 0
    for path in FILES:
        fx = spdg.load(path)
        model = abi.wilip_model_from_fixture(fx)
        prot, (sc, sp, p, hsps, n, lowest, wl) = _inputs(fx)
        own = {}                                            # (the checker's driver looks replies up by (level, span))
        tr = []
        (seeded.align_h_seeded if prot else seeded.align_s_seeded)(sc, sp, p, hsps, n, lowest, wl, 0, trace=tr)
        for kind, a, data in tr:
            if kind == 2:
                own[(a[14], a[0], a[1], a[2], a[3])] = seeded.wilip(model, p, sc, a[14], a[:4], exg=(a[4], a[5]))
        scr, flat, rc = (seeded.align_h_seeded if prot else seeded.align_s_seeded)(sc, sp, p, hsps, n, lowest, own, 0)
        assert rc == 0 and scr == int(fx["seed_scr_A0"][0]) and (flat or []) == fx["seed_skl_A0"].tolist(), _name(path)
        n_ok += 1
    assert n_ok >= 80


def test_c_abi_entry_answers_like_the_checker():
    """spdp_wilip of the product library itself (host code: callable without a GPU) on the requests of a few fixtures"""
    import ctypes as C
    from spaln_amd import engine
    lib = C.CDLL(engine.LIB_PATH)
    lib.spdp_wilip.restype = C.c_int
    lib.spdp_wilip.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    n = 0
    for path in [f for f in FILES if _name(f) in ("q_0687", "q_0745", "ql3_0036", "qh_0085", "qh_0116", "qhl3_0098")]:
        fx = spdg.load(path)
        model = abi.wilip_model_from_fixture(fx)
        prot, (sc, sp, p, hsps, n_h, lowest, wl) = _inputs(fx)
        tr = []
        (seeded.align_h_seeded if prot else seeded.align_s_seeded)(sc, sp, p, hsps, n_h, lowest, wl, 0, trace=tr)
        for kind, a, data in tr:
            if kind != 2:
                continue
            span = (C.c_int32 * 4)(*a[:4])
            exg = (C.c_int32 * 2)(a[4], a[5])
            flat = C.POINTER(C.c_int32)()
            args = (None, None, C.addressof(p), C.addressof(sc)) if prot else (C.addressof(p), C.addressof(sc), None, None)
            k = lib.spdp_wilip(C.addressof(model), *args, a[14], span, exg, C.byref(flat))
            assert k >= 1
            got = [int(flat[i]) for i in range(k)]
            libc.free(flat)
            assert seeded.same_units(got, list(data))
            n += 1
    assert n >= 10
