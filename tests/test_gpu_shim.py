"""The reference-side binding end to end: oracle/_ref/shim_check is that shim compiled
against the reference's own headers and linked with libspdp_hip.so.  It sets a pair up the way the
reference does, runs HomScoreS_ng / alignS_ng (alignH_ng) of the compiled reference AND the *_gpu
replacements in the same process, and exits 0 only when scores and SKL corner lists are identical."""
import os
import subprocess

import numpy as np
import pytest

from spaln_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "shim_check")
TAB = os.path.join(ROOT, "oracle", "_ref", "table")


def _run(tmp_path, window_ascii, query_ascii, opts=()):
    gf, qf = str(tmp_path / "g.fa"), str(tmp_path / "q.fa")
    synth.write_fasta(gf, "win", window_ascii)
    synth.write_fasta(qf, "qry", query_ascii)
    r = subprocess.run([BIN, *opts, gf, qf], env=dict(os.environ, ALN_TAB=TAB), capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/shim_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("seed,kw", [(1, dict(n_exons=5, mrna_len=700, flank=300, intron_hi=1500)),
                                     (2, dict(n_exons=8, mrna_len=1400, flank=600, intron_hi=2500)),
                                     (3, dict(n_exons=3, mrna_len=400, flank=200, intron_hi=600, sub=0.1, indel=0.01))])
def test_cdna_through_the_reference_side_shim(tmp_path, seed, kw):
    rng = np.random.default_rng(synth.SEED + 7000 + seed)
    g = synth.make_gene(rng, **kw)
    rc, out = _run(tmp_path, g.window, g.query)
    assert rc == 0 and "IDENTICAL" in out, out[-1500:]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/shim_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("seed,kw", [(1, dict(n_exons=3, aa_len=150, flank=200, intron_hi=600)),
                                     (2, dict(n_exons=5, aa_len=400, flank=500, intron_hi=1500)),
                                     (3, dict(n_exons=4, aa_len=220, flank=300, intron_hi=900, sub=0.3))])
def test_protein_through_the_reference_side_shim(tmp_path, seed, kw):
    rng = np.random.default_rng(synth.SEED + 7100 + seed)
    g = synth.make_protein_gene(rng, **kw)
    rc, out = _run(tmp_path, g.window, g.query)
    assert rc == 0 and "IDENTICAL" in out, out[-1500:]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/shim_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("qck", [1, 2, 3])
@pytest.mark.parametrize("seed,kw", [(11, dict(n_exons=5, mrna_len=900, flank=400, intron_hi=1500, sub=0.03, indel=0.005)),
                                     (12, dict(n_exons=8, mrna_len=2000, flank=800, intron_hi=3000, sub=0.08, indel=0.01)),
                                     (13, dict(n_exons=4, mrna_len=600, flank=300, intron_hi=800, sub=0.15, indel=0.02))])
def test_seeded_path_through_the_reference_side_shim(tmp_path, seed, kw, qck):
    """-Q5 .. -Q7 live: the reference's own geneorient() supplies the HSPs and its own Wilip answers the recursion levels
    through SpdpHspSource; alignS_ng of the reference against spdp_align_s_seeded in one process"""
    rng = np.random.default_rng(synth.SEED + 7200 + seed)
    g = synth.make_gene(rng, **kw)
    rc, out = _run(tmp_path, g.window, g.query, ("-Q", str(qck)))
    if rc == 4:
        pytest.skip("geneorient() preferred the reverse strand")
    assert rc == 0 and "IDENTICAL" in out, out[-1500:]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/shim_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("qck", [1, 2, 3])
@pytest.mark.parametrize("seed,kw", [(21, dict(n_exons=4, aa_len=300, flank=400, intron_hi=900, sub=0.1)),
                                     (22, dict(n_exons=6, aa_len=420, flank=500, intron_hi=1500, sub=0.25)),
                                     (23, dict(n_exons=3, aa_len=180, flank=300, intron_hi=600, sub=0.05))])
def test_protein_seeded_path_through_the_reference_side_shim(tmp_path, seed, kw, qck):
    """the protein walk live: geneorient() and Wilip of the compiled reference behind spdp_align_h_seeded"""
    rng = np.random.default_rng(synth.SEED + 7300 + seed)
    g = synth.make_protein_gene(rng, **kw)
    rc, out = _run(tmp_path, g.window, g.query, ("-Q", str(qck)))
    if rc == 4:
        pytest.skip("geneorient() preferred the reverse strand")
    if rc == 5:
        pytest.skip("a DP call of this case is undefined in the reference")
    assert rc == 0 and "IDENTICAL" in out, out[-1500:]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/shim_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("seed,opts", [(0, ("-A", "0")), (1, ("-Q", "3")), (2, ("-A", "0")), (3, ("-Q", "3"))])
def test_headline_shape_live(tmp_path, seed, opts):
    """BASELINE's C2 shape (2 kb cDNA, locus +- 1 kb) live against the compiled reference: its exact engines (-A0: HomScoreS_ng
    and alignS_ng, reliable at this size) and the seeded path under -A2 (-Q7: the DP calls between HSPs stay below the rows
    where the reference's int16 engines re-base)"""
    g = synth.make_gene(np.random.default_rng(synth.SEED + 9000 + seed), sub=0.04 + 0.01 * (seed % 5), indel=0.005)
    rc, out = _run(tmp_path, g.window, g.query, opts)
    if rc == 4:
        pytest.skip("geneorient() preferred the reverse strand")
    assert rc == 0 and "IDENTICAL" in out, out[-1500:]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/shim_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_wip_engines_at_headline_size_live(tmp_path, seed):
    """the engines bench.py times (`_wip`: linear-space sweep + slab tracebacks) at BASELINE's 2 kb size against the
    reference's OWN -A2 run: queries divergent enough (18 - 24 % substitutions) that its int16 scores never reach the
    re-basing threshold above which its output stops being a function of the input (at 6 - 8 % the same
    comparison fails on the reference's side)"""
    g = synth.make_gene(np.random.default_rng(synth.SEED + 9500 + seed), sub=0.18 + 0.02 * (seed % 6), indel=0.01)
    rc, out = _run(tmp_path, g.window, g.query, ("-A", "2"))
    assert rc == 0 and "IDENTICAL" in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n", [("c2", 600), ("c3", 600)])
def test_seeded_batch_live(shape, n):
    """a BATCH through the seeded path: the reference's alignS_ng / alignH_ng on host threads against one
    spdp_align_*_seeded call (thousands of walks in flight on fibers, requests in latency classes), pair by pair"""
    import json
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "seed_bench.py")
    if not os.path.exists(os.path.join(os.path.dirname(tool), "..", "oracle", "_ref", "seed_bench")):
        pytest.skip("oracle/_ref/seed_bench not built")
    r = subprocess.run([sys.executable, tool, str(n), shape, "3", "8"], capture_output=True, text=True, timeout=600)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["walked"] >= n - n // 20                     # (pairs whose other strand wins geneorient() are left out)
    assert d["compared"] == d["walked"] and d["identical"] == d["compared"], d
