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


def _text(fx, alg, protein, lib, header, gene_id=0):
    prm = [int(x) for x in fx[f"rng_exnprm_A{alg}"]]
    scale = np.array(prm[0], dtype=np.int32).view(np.float32)
    aln_scale = np.array(prm[1], dtype=np.int32).view(np.float32)
    gmap, qmap = (prm[2], prm[3] - prm[2]), (prm[6], prm[7] - prm[6])
    return engine.exon_form(lib, fx[f"rng_eij_A{alg}"], scr=prm[11], gene_codes=fx["b_codes"], protein=protein,
                            q_left=prm[12], q_right=prm[13], q_len=prm[8], q_many=prm[10], q_sens=prm[9],
                            gmap=gmap, qmap=qmap, scale=float(scale), aln_scale=float(aln_scale), header=header, gene_id=gene_id)


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


def test_o12_files(tmp_path):
    """spdp_o12_*: the three record files -O12 writes (layout of src/sqpr.cc:853-985): record counts, the running exon
    index in GeneRecord::Nrecord, the query index, NUL-separated names"""
    lib = engine.load_library()
    lib.spdp_o12_open.restype = C.c_void_p
    lib.spdp_o12_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.spdp_o12_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
    lib.spdp_o12_close.argtypes = [C.c_void_p]
    prefix = str(tmp_path / "out")
    h = lib.spdp_o12_open(prefix.encode(), b"genome_db")
    assert h
    counts = []
    for k, name in enumerate(("s1_basic", "s1_indels", "s1_1400nt")):
        fx = spdg.load([f for f in FILES if f.endswith(name + ".spdg")][0])
        ex, g, _ = _text(fx, 2, False, lib, header=False)
        arr = (abi.ExonRecord * len(ex))(*ex)
        assert lib.spdp_o12_write(h, arr, len(ex), C.byref(g), f"q{k}".encode()) == 0
        counts.append(len(ex))
    assert lib.spdp_o12_close(h) == 0
    grd = open(prefix + ".grd", "rb").read(); erd = open(prefix + ".erd", "rb").read(); qrd = open(prefix + ".qrd", "rb").read()
    assert len(grd) == 3 * 72 and len(erd) == sum(counts) * 72
    assert qrd == b"genome_db\0q0\0q1\0q2\0"
    genes = (abi.GeneRecord * 3).from_buffer_copy(grd)
    assert [g.Nrecord for g in genes] == [0, counts[0], counts[0] + counts[1]]
    assert [g.Rid for g in genes] == [1, 2, 3] and [g.nexn for g in genes] == counts


O12 = golden_files("o12_")


@pytest.mark.parametrize("path", O12, ids=[f.split("/")[-1][:-5] for f in O12])
def test_o12_files_equal_the_references_bytes(path, tmp_path):
    """the reference's own -O12 output (ref_dump -B: Gsinfo::ExonForm with BIN_FORM writes <prefix>.grd / .erd / .qrd for the
    -A0 and then the -A2 alignment of the case, src/sqpr.cc:853-985) against spdp_exon_form + spdp_o12_* fed the same EISCR
    records: all three files byte for byte -- struct layouts, float arithmetic, running Nrecord / Rid, padding bytes"""
    fx = spdg.load(path)
    lib = engine.load_library()
    lib.spdp_o12_open.restype = C.c_void_p
    lib.spdp_o12_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.spdp_o12_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
    lib.spdp_o12_close.argtypes = [C.c_void_p]
    protein = "_h1_" in path
    prefix = str(tmp_path / "out")
    h = lib.spdp_o12_open(prefix.encode(), b"fixture_db")
    assert h
    n = 0
    for alg in (0, 2):                                   # the order the harness wrote them in
        if f"rng_eij_A{alg}" not in fx or not len(fx[f"rng_eij_A{alg}"]):
            continue
        ex, g, _ = _text(fx, alg, protein, lib, header=False, gene_id=int(fx["o12_meta"][0]))
        arr = (abi.ExonRecord * max(1, len(ex)))(*ex)
        assert lib.spdp_o12_write(h, arr, len(ex), C.byref(g), b"qry") == 0
        n += 1
    assert lib.spdp_o12_close(h) == 0 and n >= 1
    for ext in ("grd", "erd", "qrd"):
        got = open(f"{prefix}.{ext}", "rb").read()
        want = bytes(fx[f"o12_{ext}"])
        assert got == want, (ext, len(got), len(want))
