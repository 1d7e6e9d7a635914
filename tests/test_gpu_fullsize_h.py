"""Parity at BASELINE's C3 problem size (400 aa proteins vs 5-15 kb windows): the GPU alignH_ng path
over a large batch, checked (a) on every query through size-independent properties of a spliced
protein alignment and (b) bit for bit against the oracle ladder on a random sample."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_align_h(item):
    from spaln_amd import abi, defaults, synth
    from oracle import host_logic_h as hh
    q, sg = item
    sc = defaults.scoring_h()
    ps = abi.ProblemSetH()
    p = ps.add(synth.encode_protein(q), sg["b"], sg["sig5"], sg["sig3"], sg["sigS"], sg["sigT"], sg["sigE"],
               sg["phs5"], sg["phs3"])
    try:
        return hh.align_h(sc, p) + (0,)
    except hh.ReferenceUndefined:
        return (None, None, -2)
    except hh.ReferenceFatal:
        return (None, None, -1)


def test_c3_fullsize_batch():
    from spaln_amd import abi, defaults, engine, synth
    n_q = 768
    batch = synth.make_protein_batch(n_q, seed=synth.SEED + 4077)
    sc = defaults.scoring_h()
    ps = abi.ProblemSetH()
    for g, sg in batch:
        ps.add(synth.encode_protein(g.query), sg["b"], sg["sig5"], sg["sig3"], sg["sigS"], sg["sigT"], sg["sigE"],
               sg["phs5"], sg["phs3"])
    eng = engine.Engine(0)
    res = eng.align_h(sc, ps)
    eng.close()
    n_found = n_def = 0
    for (score, skl, flag), (g, _) in zip(res, batch):
        if flag != 0:
            assert flag in (-1, -2)
            continue
        n_def += 1
        assert skl.shape[0] >= 3
        flags, cnt = int(skl[0][0]), int(skl[0][1])
        assert flags == 1 and cnt == skl.shape[0] - 1
        c = skl[1:]
        # corners are monotone and inside the sequences; a step is a codon-diagonal run (3 nt per
        # residue), a gap on either side, or a frame shift of one or two nucleotides
        dm, dn = np.diff(c[:, 0]), np.diff(c[:, 1])
        assert (dm >= 0).all() and (dn >= 0).all()
        assert c[0, 0] >= 0 and c[-1, 0] <= len(g.query) and c[0, 1] >= 0 and c[-1, 1] <= len(g.window)
        assert ((dn == 3 * dm) | (dm == 0) | (dn == 0) | (np.abs(dn - 3 * dm) <= 2)).all()
        # the planted gene is found: most coding columns lie on codon-diagonal runs
        on_diag = np.zeros(len(g.window) + 1, dtype=bool)
        for (m0, n0), (m1, n1) in zip(c[:-1], c[1:]):
            if m1 > m0 and n1 > n0:
                on_diag[n0:n1] = True
        coding = np.zeros(len(g.window) + 1, dtype=bool)
        for e0, e1 in g.exons:
            coding[e0:e1] = True
        n_found += (on_diag & coding).sum() >= 0.8 * coding.sum()
    assert n_def > 0.9 * n_q and n_found > 0.9 * n_def
    # exact comparison on a sample
    rng = np.random.default_rng(6)
    pick = sorted(rng.choice(n_q, size=min(128, n_q), replace=False).tolist())
    with mp.Pool(min(os.cpu_count() or 1, len(pick))) as pool:
        want = pool.map(_oracle_align_h, [(batch[i][0].query, batch[i][1]) for i in pick])
    for i, (ws, wskl, wflag) in zip(pick, want):
        s, skl, flag = res[i]
        assert flag == wflag, i
        if wflag == 0:
            assert s == ws, i
            assert skl.ravel().tolist() == (wskl or []), i
