"""the CPU restatement of the splice-signal precompute (oracle/signals.py: Exinon::intron53_c /
intron53_n over PatMat::calcPatMat) against the arrays the reference itself produced for every S fixture.

Two cells per window are excluded: sig5 / cano5 of position right - 1 (and cano5 of `right`) and sig3 / cano3 of
position left -- the reference's loops never assign the class cells behind them, so what it dumps there is whatever
the allocation held (it differs between runs of the same input; the DP never reads a donor there / an acceptor
there)."""
import numpy as np
import pytest

from oracle import signals
from tests import spdg
from tests.conftest import golden_files, golden_ids

FILES = [p for pre in ("s1_", "c2_", "o3_") for p in golden_files(pre)]
IDS = [i for pre in ("s1_", "c2_", "o3_") for i in golden_ids(pre)]


def _stale(n, left, right):
    ok5 = np.ones(n + 1, dtype=bool); ok3 = np.ones(n + 1, dtype=bool)
    ok5[right - 1:right + 1] = False
    ok3[left] = False
    return ok5, ok3


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_signals_equal_reference(path):
    fx = spdg.load(path)
    md = signals.model_of(fx)
    b = fx["b_codes"]
    n = b.size
    left, right = fx["prm"]["b_left"], fx["prm"]["b_right"]
    ok5, ok3 = _stale(n, left, right)
    d5, d3, c5, c3 = signals.classes(b, left, right, md["any"], md["both_ori"])
    assert np.array_equal(d5, fx["dinc5"]) and np.array_equal(d3, fx["dinc3"])
    assert np.array_equal((c5 > 0)[ok5], (fx["cano5"] > 0)[ok5])
    assert np.array_equal((c3 > 0)[ok3], (fx["cano3"] > 0)[ok3])
    # the pure-Python scan costs ~50 us per position: whole window when short, both ends + a slice otherwise
    spans = [(left, right)] if right - left <= 3000 else [(left, left + 700), ((left + right) // 2, (left + right) // 2 + 700),
                                                            (right - 700, right)]
    for lo, hi in spans:
        s5, s3 = signals.splice_signals(md, b, left, right, lo, hi)
        sel = np.zeros(n + 1, dtype=bool); sel[lo:hi] = True
        assert np.array_equal(s5[sel & ok5], fx["sig5"][sel & ok5]), (path, lo, hi)
        assert np.array_equal(s3[sel & ok3], fx["sig3"][sel & ok3]), (path, lo, hi)
    outside = np.ones(n + 1, dtype=bool); outside[left:right] = False
    assert not fx["sig5"][outside].any() and not fx["sig3"][outside].any()


def test_model_is_the_same_in_every_fixture():
    """one species table behind all fixtures: the model is data of the parameter set, not of the window"""
    ref = None
    for path in FILES:
        fx = spdg.load(path)
        key = (fx["pm5_hdr"].tolist(), fx["pm3_hdr"].tolist(), fx["pm5_f32"].tobytes(), fx["pm3_f32"].tobytes(),
               fx["sig53tab01"].tolist(), int(fx["sigmodel"][0]))
        ref = ref or key
        assert key == ref, path


def test_scan_edge_rules():
    fx = spdg.load(golden_files("s1_basic")[0])
    md = signals.model_of(fx)
    pm = md["pm5"]
    x = signals.RED_STRICT[fx["b_codes"][:200]].copy()
    floor = np.float32(np.float32(pm.cols) * pm.min_elem) + pm.tonic
    assert signals.scan(pm, x, x.size - 1) == floor              # the window runs off the end: "bad"
    x[100] = 4                                                   # an N anywhere under the window: "bad"
    assert signals.scan(pm, x, 100) == floor and signals.scan(pm, x, 100 - pm.cols) != floor
    assert signals.scan(pm, x, 0) != floor                       # a window starting before base 0 just skips columns
