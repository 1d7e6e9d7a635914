"""spdp_exon_form / spdp_exon_form_text (Gsinfo::ExonForm, src/sqpr.cc:820-996) against the reference's own
-O4 text, captured per fixture by the harness (Gsinfo::printgene(seqs, EXN_FORM) on the reference's alignment): fed the
reference's EISCR records, the library must print the same bytes -- cDNA and protein queries, -A0 and -A2 alignments,
frame shifts included.  Host-only code: runs without a GPU.  The binary ExonRecord / GeneRecord hold the same numbers;
their layout is checked against the reference's struct sizes."""
import ctypes as C

import numpy as np
import pytest

from spaln_amd import abi, engine
from tests import spdg
from tests.conftest import golden_files

FILES = [f for pre in ("s1_", "c2_", "o3_", "h1_", "c1_") for f in golden_files(pre)]


def _text(fx, alg, protein, lib, header):
    prm = [int(x) for x in fx[f"rng_exnprm_A{alg}"]]
    scale = np.array(prm[0], dtype=np.int32).view(np.float32)
    aln_scale = np.array(prm[1], dtype=np.int32).view(np.float32)
    gmap, qmap = (prm[2], prm[3] - prm[2]), (prm[6], prm[7] - prm[6])
    return engine.exon_form(lib, fx[f"rng_eij_A{alg}"], scr=prm[11], gene_codes=fx["b_codes"], protein=protein,
                            q_left=prm[12], q_right=prm[13], q_len=prm[8], q_many=prm[10], q_sens=prm[9],
                            gmap=gmap, qmap=qmap, scale=float(scale), aln_scale=float(aln_scale), header=header)


def test_record_layouts():
    assert C.sizeof(abi.ExonRecord) == 72 and C.sizeof(abi.GeneRecord) == 72        # sizeof(ExonRecord), sizeof(GeneRecord)


@pytest.mark.parametrize("path", FILES, ids=[f.split("/")[-1][:-5] for f in FILES])
def test_exon_form_text_equals_reference(path):
    fx = spdg.load(path)
    lib = engine.load_library()
    protein = path.split("/")[-1].startswith(("h1_", "c1_"))
    n = 0
    for alg in (0, 2):
        if f"rng_exn_A{alg}" not in fx:
            continue
        want = bytes(fx[f"rng_exn_A{alg}"])
        ex, g, got = _text(fx, alg, protein, lib, header=want.startswith(b"#"))
        assert got == want, (alg, got.decode(), want.decode())
        assert g.nexn == len(ex) and (len(ex) == 0 or ex[0].Ilen == 0)
        n += 1
    if not n:
        pytest.skip("no alignment in this fixture")
