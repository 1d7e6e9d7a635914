"""Parity at BASELINE's full problem size (C2: 2 kb cDNA vs ~12 kb windows, scores beyond the
int16 range, multi-intermediate UDH + slab tracebacks): the GPU alignS_ng path over a large
batch, checked (a) on every query through size-independent properties of a spliced alignment
and (b) bit for bit against the oracle ladder on a random sample (one query per host process)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_align(item):
    from spaln_amd import abi, defaults
    from oracle import host_logic
    w, q, s5, s3 = item
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    p = ps.add(q, w, s5, s3)
    return host_logic.align_s(sc, p)


def test_c2_fullsize_batch():
    from spaln_amd import abi, defaults, engine, synth
    n_q = 1500
    batch = synth.make_batch(n_q, seed=synth.SEED + 77)
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in batch:
        ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    res = eng.align_s(sc, ps)
    scores = eng.homscore_s(sc, ps)
    eng.close()
    assert max(s for s, _ in res) > 32767            # beyond what the int16 reference engines can hold
    n_aligned = 0
    for (score, skl), (w, q, _, _, exons), hs in zip(res, batch, scores):
        assert skl.shape[0] >= 3
        flags, cnt = int(skl[0][0]), int(skl[0][1])
        assert flags == 1 and cnt == skl.shape[0] - 1
        c = skl[1:]
        # corners are monotone, inside the sequences, and every step is a diagonal run or a gap
        assert (np.diff(c[:, 0]) >= 0).all() and (np.diff(c[:, 1]) >= 0).all()
        assert c[0, 0] >= 0 and c[-1, 0] <= len(q) and c[0, 1] >= 0 and c[-1, 1] <= len(w)
        dm, dn = np.diff(c[:, 0]), np.diff(c[:, 1])
        assert ((dm == dn) | (dm == 0) | (dn == 0)).all()
        # the planted gene is found: most exon boundaries are corner coordinates
        cols = set(int(x) for x in c[:, 1])
        hits = sum((e0 in cols) + (e1 in cols) for e0, e1 in exons)
        n_aligned += hits >= len(exons)              # at least half of the 2 * n_exons boundaries
        assert score >= int(hs) - 2000               # UDH score and score-only sweep are the same model
    assert n_aligned > 0.9 * n_q
    # exact comparison on a sample
    rng = np.random.default_rng(5)
    pick = sorted(rng.choice(n_q, size=min(96, n_q), replace=False).tolist())
    with mp.Pool(min(os.cpu_count() or 1, len(pick))) as pool:
        want = pool.map(_oracle_align, [batch[i][:4] for i in pick])
    for i, (ws, wskl) in zip(pick, want):
        s, skl = res[i]
        assert s == ws, i
        assert skl.ravel().tolist() == (wskl or []), i
