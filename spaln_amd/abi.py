"""ctypes mirror of include/spdp.h (the C ABI of libspdp_hip.so).

Pure declarations: structure layouts and helpers that fill them from numpy
arrays.  Used by the host-side binding (spaln_amd/engine.py) and, for the
struct layouts only, by the oracle's Python wrapper under oracle/.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

MAX_QUANT = 8
NEVSEL = -(2 ** 31) // 16 * 7            # INT_MIN / 16 * 7 (C truncation: INT_MIN is divisible by 16)
END_OF_ULK = 2 ** 31 - 1 - 2
REF_NELEM = 16


class Scoring(C.Structure):
    _fields_ = [
        ("mtx_dim", C.c_int32),
        ("mtx", C.c_int32 * (32 * 32)),
        ("gop", C.c_int32), ("gep", C.c_int32),
        ("lgop", C.c_int32), ("lgep", C.c_int32),
        ("noll", C.c_int32),
        ("spj", C.c_int32),
        ("llmt", C.c_int32),
        ("ipen", C.c_int32),
        ("nquant", C.c_int32),
        ("qm_len", C.c_int32 * MAX_QUANT),
        ("qm_pen", C.c_int32 * MAX_QUANT),
        ("local", C.c_int32),
        ("sh", C.c_int32),
        ("max_vmf_space", C.c_int32),
        ("ubh", C.c_int32),
        ("ref_nelem", C.c_int32),
        ("intpen", C.c_void_p),
        ("intpen_len", C.c_int32),
        ("t53", C.c_int16 * 256),
        ("scalar_engines", C.c_int32),
        ("minl", C.c_int32),
        ("recursive", C.c_int32),
        ("sigmodel", C.c_void_p),
        ("codonk1", C.c_int32),
    ]


class SignalModel(C.Structure):          # SpdpSignalModel
    _fields_ = [
        ("rows", C.c_int32),
        ("cols5", C.c_int32), ("off5", C.c_int32),
        ("cols3", C.c_int32), ("off3", C.c_int32),
        ("fs", C.c_float),
        ("tonic5", C.c_float), ("min5", C.c_float),
        ("tonic3", C.c_float), ("min3", C.c_float),
        ("mtx5", C.c_void_p), ("mtx3", C.c_void_p),
        ("tab5", C.c_int16 * 16), ("tab3", C.c_int16 * 16),
        ("any", C.c_int32), ("both_ori", C.c_int32),
    ]


def make_signal_model(*, rows, cols5, off5, mtx5, tonic5, min5, cols3, off3, mtx3, tonic3, min3, fs, tab5, tab3,
                      any=0, both_ori=0) -> SignalModel:
    m = SignalModel()
    m.rows, m.cols5, m.off5, m.cols3, m.off3 = int(rows), int(cols5), int(off5), int(cols3), int(off3)
    m.fs, m.tonic5, m.min5, m.tonic3, m.min3 = float(fs), float(tonic5), float(min5), float(tonic3), float(min3)
    m5 = np.ascontiguousarray(mtx5, dtype=np.float32).ravel()
    m3 = np.ascontiguousarray(mtx3, dtype=np.float32).ravel()
    assert m5.size == rows * cols5 and m3.size == rows * cols3
    m._keep = (m5, m3)
    m.mtx5, m.mtx3 = m5.ctypes.data, m3.ctypes.data
    for i in range(16):
        m.tab5[i] = int(tab5[i]); m.tab3[i] = int(tab3[i])
    m.any, m.both_ori = int(any), int(both_ori)
    return m


def signal_model_from_fixture(fx: dict) -> SignalModel:
    """the model a reference dump carries (tests/golden/*.spdg: pm5_* / pm3_* / sig53tab01 / sigmodel)"""
    h5, h3 = fx["pm5_hdr"], fx["pm3_hdr"]
    f5 = np.asarray(fx["pm5_f32"], dtype=np.int32).view(np.float32)
    f3 = np.asarray(fx["pm3_f32"], dtype=np.int32).view(np.float32)
    sm = np.asarray(fx["sigmodel"], dtype=np.int32)
    return make_signal_model(rows=h5[0], cols5=h5[1], off5=h5[2], mtx5=f5[2:], tonic5=f5[0], min5=f5[1],
                             cols3=h3[1], off3=h3[2], mtx3=f3[2:], tonic3=f3[0], min3=f3[1],
                             fs=sm[:1].view(np.float32)[0], tab5=fx["sig53tab01"][:16], tab3=fx["sig53tab01"][16:],
                             any=sm[1], both_ori=sm[3] if sm.size > 3 else 0)


class Problem(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a_len", C.c_int32),
        ("b", C.c_void_p), ("b_len", C.c_int32),
        ("sig5", C.c_void_p),
        ("sig3", C.c_void_p),
        ("a_left", C.c_int32), ("a_right", C.c_int32),
        ("b_left", C.c_int32), ("b_right", C.c_int32),
        ("a_exgl", C.c_uint8), ("a_exgr", C.c_uint8),
        ("b_exgl", C.c_uint8), ("b_exgr", C.c_uint8),
        ("cano5", C.c_void_p), ("cano3", C.c_void_p), ("dinc", C.c_void_p),
        ("cip", C.c_void_p),
        ("phs5", C.c_void_p), ("phs3", C.c_void_p),
        ("exin_left", C.c_int32), ("exin_right", C.c_int32),
    ]


class Juxt(C.Structure):                 # SpdpJuxt / JUXT
    _fields_ = [(k, C.c_int32) for k in ("jx", "jy", "jlen", "nid", "jscr")]


class SeedParams(C.Structure):           # SpdpSeedParams
    _fields_ = [("qck", C.c_int32), ("wl_width", C.c_int32 * 4), ("elmt", C.c_int32), ("minl", C.c_int32),
                ("vthr", C.c_int32), ("desert", C.c_int32), ("maxsp", C.c_float), ("crs", C.c_int32),
                ("smn4", C.c_float), ("w2", C.c_float), ("gc_sig5", C.c_int32), ("lcl", C.c_int32),
                ("codonk1", C.c_int32), ("any", C.c_int32), ("both_ori", C.c_int32), ("ip_maxl", C.c_int32),
                ("ip_mode", C.c_int32), ("wilip", C.c_void_p)]


HSP_UNITS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                           C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32))
HSP_RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_int32))


class HspSource(C.Structure):            # SpdpHspSource
    _fields_ = [("user", C.c_void_p), ("units", HSP_UNITS_FN), ("release", HSP_RELEASE_FN)]


def seed_params_from_fixture(fx: dict) -> SeedParams:
    """SpdpSeedParams of a `ref_dump -Q` fixture (tests/golden/q*_*.spdg: seed_params + params)"""
    v = np.asarray(fx["seed_params"], dtype=np.int32)
    sp = SeedParams()
    sp.qck = int(v[0])
    for k in range(4):
        sp.wl_width[k] = int(v[3 + k])
    sp.elmt, sp.minl, sp.vthr, sp.desert = int(v[7]), int(v[8]), int(v[10]), int(v[11])
    sp.maxsp = float(v[12:13].view(np.float32)[0])
    sp.crs = int(v[13])
    sp.smn4 = float(v[14:15].view(np.float32)[0])
    sp.w2 = float(v[15:16].view(np.float32)[0])
    sp.gc_sig5, sp.lcl = int(v[16]), int(v[17])
    sp.codonk1 = int(fx["params"][7])
    if "sigmodel" in fx:
        sm = np.asarray(fx["sigmodel"], dtype=np.int32)
        sp.any, sp.both_ori = int(sm[1]), int(sm[3]) if sm.size > 3 else 0
    else:                                                # protein fixtures: sigmodel_i32 = any, DvsP, ...
        sp.any, sp.both_ori = int(fx["sigmodel_i32"][0]), 0
    if v.size > 21:
        sp.ip_maxl, sp.ip_mode = int(v[20]), int(v[21])
    return sp


class Window(C.Structure):
    _fields_ = [("lw", C.c_int32), ("up", C.c_int32), ("width", C.c_int32)]


class Skl(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32)]


class Alignment(C.Structure):
    _fields_ = [("score", C.c_int32), ("n_skl", C.c_int32), ("skl", C.POINTER(Skl)), ("flags", C.c_int32),
                ("reserved", C.c_int32)]


ALN_LEFT_EDGE = 1                       # SPDP_ALN_LEFT_EDGE


class Exon(C.Structure):                 # SpdpExon / EISCR
    _fields_ = [(k, C.c_int32) for k in
                ("left", "right", "rleft", "rright", "mch", "mmc", "gap", "unp", "mch5", "mmc5", "gap5", "unp5",
                 "mch3", "mmc3", "gap3", "unp3", "phs", "escr", "iscr", "sig3", "sig5")]


class RescoreParams(C.Structure):
    _fields_ = [("codonk1", C.c_int32), ("minl", C.c_int32), ("jneibr", C.c_int32), ("lsg", C.c_int32)]


class Rescored(C.Structure):
    _fields_ = [("score", C.c_int32), ("mch", C.c_int32), ("mmc", C.c_int32), ("gap", C.c_int32),
                ("unp", C.c_int32), ("val", C.c_int32), ("n_exons", C.c_int32), ("exons", C.POINTER(Exon))]


class Edit(C.Structure):                 # SpdpEdit
    _fields_ = [("op", C.c_int32), ("alen", C.c_int32), ("blen", C.c_int32)]


class Edits(C.Structure):                # SpdpEdits
    _fields_ = [("n", C.c_int32), ("rec", C.POINTER(Edit)), ("sam_flag", C.c_int32), ("sam_pos", C.c_int32),
                ("sam_mapq", C.c_int32), ("sam_left", C.c_int32), ("sam_right", C.c_int32)]


FMT_CIGAR, FMT_VULGAR, FMT_SAM = 1, 2, 3


class SiteMap(C.Structure):              # SpdpSiteMap
    _fields_ = [("site0", C.c_int32), ("step", C.c_int32)]


class ExonRecord(C.Structure):           # SpdpExonRecord = ExonRecord, src/seq.h:1212
    _fields_ = [(k, C.c_int32) for k in ("Elen", "Nmmc", "Nunp", "Rleft", "Rright", "Gleft", "Gright", "Ilen", "Bmmc",
                                          "Bunp", "miss", "phase")] + \
               [(k, C.c_float) for k in ("Pmatch", "Escore", "Iscore", "Sig3", "Sig5")] + [("Iends", C.c_char * 4)]


class GeneRecord(C.Structure):           # SpdpGeneRecord = GeneRecord, src/seq.h:1235
    _fields_ = [("Cid", C.c_int32), ("Gstart", C.c_int32), ("Gend", C.c_int32), ("Nrecord", C.c_uint32),
                ("nexn", C.c_uint32)] + \
               [(k, C.c_int32) for k in ("Rid", "Rlen", "Rstart", "Rend", "mmc", "unp", "bmmc", "bunp", "ng")] + \
               [(k, C.c_float) for k in ("Gscore", "Pmatch", "Pcover")] + [("Csense", C.c_int16), ("Rsense", C.c_int16)]


class ExonFormIn(C.Structure):           # SpdpExonFormIn
    _fields_ = [("eij", C.c_void_p), ("n_eij", C.c_int32), ("scr", C.c_int32), ("gene_codes", C.c_void_p),
                ("gene_is_tron", C.c_int32), ("qry_is_protein", C.c_int32),
                ("q_left", C.c_int32), ("q_right", C.c_int32), ("q_len", C.c_int32), ("q_many", C.c_int32),
                ("q_sens", C.c_int32), ("gmap", SiteMap), ("qmap", SiteMap), ("scale", C.c_float),
                ("aln_scale", C.c_float), ("hsp_len", C.c_int32), ("gene_id", C.c_int32), ("qry_id", C.c_int32),
                ("first_exon_record", C.c_int32)]


def make_scoring(*, mtx, mtx_dim, gop, gep, lgop=0, lgep=0, noll=2, spj=1, llmt=20,
                 ipen=0, qm_len=(0,), qm_pen=(0,), nquant=None, local=0, sh=100,
                 max_vmf_space=32 * 1024 * 1024, ubh=0, ref_nelem=REF_NELEM,
                 intpen=None, t53=None, scalar_engines=0, minl=0, recursive=0, sigmodel=None, codonk1=0) -> Scoring:
    sc = Scoring()
    sc.mtx_dim = int(mtx_dim)
    flat = np.asarray(mtx, dtype=np.int32).ravel()
    assert flat.size == mtx_dim * mtx_dim and mtx_dim <= 32
    for i, v in enumerate(flat):
        sc.mtx[i] = int(v)
    sc.gop, sc.gep, sc.lgop, sc.lgep = int(gop), int(gep), int(lgop), int(lgep)
    sc.noll, sc.spj, sc.llmt, sc.ipen = int(noll), int(spj), int(llmt), int(ipen)
    nq = len(qm_len) if nquant is None else int(nquant)
    assert 1 <= nq <= MAX_QUANT
    sc.nquant = nq
    for j in range(min(len(qm_len), MAX_QUANT)):
        sc.qm_len[j] = int(qm_len[j])
        sc.qm_pen[j] = int(qm_pen[j])
    sc.local, sc.sh = int(local), int(sh)
    sc.max_vmf_space, sc.ubh, sc.ref_nelem = int(max_vmf_space), int(ubh), int(ref_nelem)
    sc.scalar_engines = int(scalar_engines)
    sc.minl = int(minl)
    sc.recursive = int(recursive)
    if int(noll) == 3 and int(codonk1) < 1:
        raise ValueError("noll = 3 needs codonk1 >= 1 (alprm2.k1 of the reference): the first column's GapPenalty(1) depends on it")
    sc.codonk1 = int(codonk1)
    if sigmodel is not None:
        sc._keep_sigmodel = sigmodel
        sc.sigmodel = C.addressof(sigmodel)
    if intpen is not None:
        ip = np.ascontiguousarray(intpen, dtype=np.int16)
        sc._keep_intpen = ip                      # keep the buffer alive with the struct
        sc.intpen, sc.intpen_len = ip.ctypes.data, ip.size
    if t53 is not None:
        for i, v in enumerate(np.asarray(t53).ravel()[:256]):
            sc.t53[i] = int(v)
    return sc


class ProblemSet:
    """Owns the numpy buffers a ctypes Problem array points into."""

    def __init__(self):
        self._keep = []
        self.items = []

    def add(self, a, b, sig5, sig3, a_left=0, a_right=None, b_left=0, b_right=None,
            exg=(1, 1, 1, 1), cano5=None, cano3=None, dinc=None, cip=None, phs5=None, phs3=None):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        p = Problem()
        p.a, p.a_len = a.ctypes.data, a.size
        p.b, p.b_len = b.ctypes.data, b.size
        self._keep += [a, b]
        if sig5 is not None:                          # None: Scoring.sigmodel computes them on the device
            sig5 = np.ascontiguousarray(sig5, dtype=np.int16)
            sig3 = np.ascontiguousarray(sig3, dtype=np.int16)
            assert sig5.size >= b.size + 1 and sig3.size >= b.size + 1
            self._keep += [sig5, sig3]
            p.sig5, p.sig3 = sig5.ctypes.data, sig3.ctypes.data
        p.a_left, p.a_right = int(a_left), int(a.size if a_right is None else a_right)
        p.b_left, p.b_right = int(b_left), int(b.size if b_right is None else b_right)
        p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr = (int(x) for x in exg)
        if cano5 is not None:
            c5 = np.ascontiguousarray(cano5, dtype=np.uint8)
            c3 = np.ascontiguousarray(cano3, dtype=np.uint8)
            dc = np.ascontiguousarray(dinc, dtype=np.uint8)
            assert min(c5.size, c3.size, dc.size) >= b.size + 1
            self._keep += [c5, c3, dc]
            p.cano5, p.cano3, p.dinc = c5.ctypes.data, c3.ctypes.data, dc.ctypes.data
        if cip is not None:
            cp = np.ascontiguousarray(cip, dtype=np.int32)
            assert cp.size >= a.size + 1
            self._keep.append(cp)
            p.cip = cp.ctypes.data
        if phs5 is not None:
            h5 = np.ascontiguousarray(phs5, dtype=np.int8)
            h3 = np.ascontiguousarray(phs3, dtype=np.int8)
            assert min(h5.size, h3.size) >= b.size + 1
            self._keep += [h5, h3]
            p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
        self.items.append(p)
        return p

    def __len__(self):
        return len(self.items)

    def array(self):
        arr = (Problem * len(self.items))(*self.items)
        self._keep.append(arr)
        return arr


# ---- protein x genome (Fwd2h1 `_wip`) ------------------------------------------------------
class ScoringH(C.Structure):
    _fields_ = [
        ("mtx_rows", C.c_int32), ("mtx_cols", C.c_int32),
        ("mtx", C.c_int32 * (32 * 32)),
        ("gop", C.c_int32), ("gep", C.c_int32),
        ("lgep", C.c_int32), ("codonk1", C.c_int32),
        ("gapw1", C.c_int32), ("gapw2", C.c_int32), ("gapw3", C.c_int32),
        ("spj", C.c_int32),
        ("llmt", C.c_int32), ("ipen", C.c_int32),
        ("nquant", C.c_int32),
        ("qm_len", C.c_int32 * MAX_QUANT),
        ("qm_pen", C.c_int32 * MAX_QUANT),
        ("local", C.c_int32),
        ("term_codon", C.c_int32),
        ("sh", C.c_int32),
        ("max_vmf_space", C.c_int32),
        ("ubh", C.c_int32),
        ("ref_nelem", C.c_int32),
        ("lgop", C.c_int32),
        ("gape1", C.c_int32), ("gape2", C.c_int32), ("extragop", C.c_int32),
        ("diffu", C.c_int32), ("k1", C.c_int32),
        ("intpen", C.c_void_p),
        ("intpen_len", C.c_int32),
        ("t53", C.c_int16 * 256),
        ("minl", C.c_int32),
        ("scalar_engines", C.c_int32),
        ("recursive", C.c_int32),
        ("noll", C.c_int32),
        ("sigmodel", C.c_void_p),
    ]


class PatMatC(C.Structure):              # SpdpPatMat
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("offset", C.c_int32), ("order", C.c_int32),
                ("tonic", C.c_float), ("min_elem", C.c_float), ("mtx", C.c_void_p)]


class SignalModelH(C.Structure):         # SpdpSignalModelH
    _fields_ = [("pm5", PatMatC), ("pm3", PatMatC), ("pmI", PatMatC), ("pmT", PatMatC),
                ("pot_ndata", C.c_int32), ("pot", C.c_void_p),
                ("fE", C.c_float), ("fT", C.c_float), ("fO", C.c_float), ("fS", C.c_float), ("fs", C.c_float),
                ("tonic5", C.c_float), ("tonic3", C.c_float),
                ("tab5", C.c_int16 * 16), ("tab3", C.c_int16 * 16),
                ("any", C.c_int32), ("dvsp", C.c_int32), ("trm", C.c_int32), ("trm2", C.c_int32),
                ("pmB", PatMatC), ("fB", C.c_float), ("tonicB", C.c_float), ("maxb3d", C.c_int32)]


def signal_model_h_from_fixture(fx: dict) -> SignalModelH:
    """the protein-side model a reference dump carries (pm*_hdr / pm*_f32, potC_*, sigmodel_f32 / _i32, sig53tab01)"""
    m = SignalModelH()
    keep = []
    for tag, dst in (("pm5", m.pm5), ("pm3", m.pm3), ("pmI", m.pmI), ("pmT", m.pmT), ("pmB", m.pmB)):
        hd = [int(v) for v in fx[tag + "_hdr"]]
        if not hd[0]:
            continue
        f = np.asarray(fx[tag + "_f32"], dtype=np.int32).view(np.float32)
        mtx = np.ascontiguousarray(f[2:], dtype=np.float32)
        keep.append(mtx)
        dst.rows, dst.cols, dst.offset, dst.order = hd[0], hd[1], hd[2], hd[3]
        dst.tonic, dst.min_elem, dst.mtx = float(f[0]), float(f[1]), mtx.ctypes.data
    pot = np.ascontiguousarray(np.asarray(fx["potC_f32"], dtype=np.int32).view(np.float32))
    keep.append(pot)
    m.pot_ndata, m.pot = int(fx["potC_hdr"][0]), (pot.ctypes.data if pot.size else None)
    f = np.asarray(fx["sigmodel_f32"], dtype=np.int32).view(np.float32)
    i = [int(v) for v in fx["sigmodel_i32"]]
    m.fE, m.fT, m.fO, m.fS, m.fs, m.tonic3, m.tonic5 = (float(f[k]) for k in (0, 2, 4, 5, 6, 7, 8))
    for k in range(16):
        m.tab5[k] = int(fx["sig53tab01"][k]); m.tab3[k] = int(fx["sig53tab01"][16 + k])
    m.any, m.dvsp, m.trm, m.trm2 = i[0], int(i[1] != 3), i[5], i[6]
    m.fB, m.tonicB, m.maxb3d = float(f[3]), float(f[9]), i[2]
    m._keep = keep
    return m


class RescoreParamsH(C.Structure):
    _fields_ = [("minl", C.c_int32), ("jneibr", C.c_int32), ("lcl", C.c_int32), ("sup_tcodon", C.c_int32)]


class ProblemH(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a_len", C.c_int32),
        ("b", C.c_void_p), ("b_len", C.c_int32),
        ("sig5", C.c_void_p), ("sig3", C.c_void_p),
        ("sigS", C.c_void_p), ("sigT", C.c_void_p), ("sigE", C.c_void_p),
        ("phs5", C.c_void_p), ("phs3", C.c_void_p),
        ("exin_left", C.c_int32), ("exin_right", C.c_int32),
        ("a_left", C.c_int32), ("a_right", C.c_int32),
        ("b_left", C.c_int32), ("b_right", C.c_int32),
        ("a_exgl", C.c_uint8), ("a_exgr", C.c_uint8),
        ("b_exgl", C.c_uint8), ("b_exgr", C.c_uint8),
        ("a_pad", C.c_uint8), ("reserved_", C.c_uint8 * 3),
        ("dinc", C.c_void_p),
        ("cip", C.c_void_p),
    ]


def make_scoring_h(*, mtx, mtx_rows, mtx_cols, gop, gep, lgep, codonk1, gapw1, gapw2, gapw3,
                   spj=1, llmt=20, ipen=0, qm_len=(0,), qm_pen=(0,), nquant=None, local=0,
                   term_codon=1, sh=100, max_vmf_space=32 * 1024 * 1024, ubh=0,
                   ref_nelem=REF_NELEM, lgop=0, gape1=0, gape2=0, extragop=0, diffu=0, k1=0,
                   intpen=None, t53=None, minl=0, scalar_engines=0, recursive=0, sigmodel=None, noll=2) -> ScoringH:
    sc = ScoringH()
    sc.mtx_rows, sc.mtx_cols = int(mtx_rows), int(mtx_cols)
    flat = np.asarray(mtx, dtype=np.int32).ravel()
    assert flat.size == mtx_rows * mtx_cols <= 32 * 32
    for i, v in enumerate(flat):
        sc.mtx[i] = int(v)
    sc.gop, sc.gep, sc.lgep, sc.codonk1 = int(gop), int(gep), int(lgep), int(codonk1)
    sc.gapw1, sc.gapw2, sc.gapw3 = int(gapw1), int(gapw2), int(gapw3)
    sc.spj, sc.llmt, sc.ipen = int(spj), int(llmt), int(ipen)
    nq = len(qm_len) if nquant is None else int(nquant)
    assert 1 <= nq <= MAX_QUANT
    sc.nquant = nq
    for j in range(min(len(qm_len), MAX_QUANT)):
        sc.qm_len[j] = int(qm_len[j])
        sc.qm_pen[j] = int(qm_pen[j])
    sc.local, sc.term_codon, sc.sh = int(local), int(term_codon), int(sh)
    sc.max_vmf_space, sc.ubh, sc.ref_nelem = int(max_vmf_space), int(ubh), int(ref_nelem)
    sc.lgop, sc.gape1, sc.gape2, sc.extragop = int(lgop), int(gape1), int(gape2), int(extragop)
    sc.diffu, sc.k1 = int(diffu), int(k1)
    sc.minl = int(minl)
    sc.scalar_engines = int(scalar_engines)
    sc.recursive = int(recursive)
    sc.noll = int(noll)
    if sigmodel is not None:
        sc._keep_sigmodel = sigmodel
        sc.sigmodel = C.addressof(sigmodel)
    if intpen is not None:
        ip = np.ascontiguousarray(intpen, dtype=np.int16)
        sc._keep_intpen = ip
        sc.intpen, sc.intpen_len = ip.ctypes.data, ip.size
    if t53 is not None:
        for i, v in enumerate(np.asarray(t53).ravel()[:256]):
            sc.t53[i] = int(v)
    return sc


class ProblemSetH:
    """Owns the numpy buffers a ctypes ProblemH array points into."""

    def __init__(self):
        self._keep = []
        self.items = []

    def add(self, a, b, sig5, sig3, sigS, sigT, sigE, phs5, phs3, a_left=0, a_right=None,
            b_left=0, b_right=None, exg=(1, 1, 1, 1), exin=None, dinc=None, cip=None, a_pad=0):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)          # b_len + 1 entries
        b_len = b.size - 1
        p = ProblemH()
        p.a_pad = int(a_pad)
        p.a, p.a_len = a.ctypes.data, a.size
        p.b, p.b_len = b.ctypes.data, b_len
        self._keep += [a, b]
        if sig5 is not None:                          # None (all seven): ScoringH.sigmodel computes them on the device
            sg = [np.ascontiguousarray(x, dtype=np.int16) for x in (sig5, sig3, sigS, sigT, sigE)]
            ph = [np.ascontiguousarray(x, dtype=np.int8) for x in (phs5, phs3)]
            assert min(x.size for x in sg + ph) >= b_len + 3
            self._keep += sg + ph
            p.sig5, p.sig3, p.sigS, p.sigT, p.sigE = (x.ctypes.data for x in sg)
            p.phs5, p.phs3 = (x.ctypes.data for x in ph)
        p.a_left, p.a_right = int(a_left), int(a.size if a_right is None else a_right)
        p.b_left, p.b_right = int(b_left), int(b_len if b_right is None else b_right)
        p.exin_left, p.exin_right = (p.b_left, p.b_right) if exin is None else (int(exin[0]), int(exin[1]))
        p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr = (int(x) for x in exg)
        if dinc is not None:
            dc = np.ascontiguousarray(dinc, dtype=np.uint8)
            assert dc.size >= b_len + 1
            self._keep.append(dc)
            p.dinc = dc.ctypes.data
        if cip is not None:
            cp = np.ascontiguousarray(cip, dtype=np.int32)
            assert cp.size >= 3 * a.size + 2
            self._keep.append(cp)
            p.cip = cp.ctypes.data
        self.items.append(p)
        return p

    def __len__(self):
        return len(self.items)

    def array(self):
        arr = (ProblemH * len(self.items))(*self.items)
        self._keep.append(arr)
        return arr


class WilipLevel(C.Structure):           # SpdpWilipLevel
    _fields_ = [("elem", C.c_int32), ("tpl", C.c_int32), ("mask", C.c_int32), ("width", C.c_int32), ("gain", C.c_int32),
                ("gain1", C.c_int32), ("thr", C.c_int32), ("xdrp", C.c_int32), ("cutoff", C.c_int32), ("vthr", C.c_int32),
                ("bitpat_len", C.c_int32), ("bitpat", C.c_uint8 * 32), ("convtab", C.c_uint8 * 32)]


class WilipModel(C.Structure):           # SpdpWilipModel
    _fields_ = [("level", WilipLevel * 3), ("mtx_rows", C.c_int32), ("mtx_cols", C.c_int32), ("mtx", C.c_int32 * (32 * 32)),
                ("dvsp", C.c_int32), ("end_bonus", C.c_int32), ("crs", C.c_int32), ("lsg", C.c_int32), ("mlt", C.c_int32),
                ("hard_minl", C.c_int32), ("hard_maxl", C.c_int32), ("minl", C.c_int32), ("maxl", C.c_int32), ("llmt", C.c_int32),
                ("avrsig", C.c_int32), ("shortquery", C.c_int32), ("min_hit", C.c_int32),
                ("met", C.c_int32), ("ser", C.c_int32), ("ser2", C.c_int32)]


def wilip_model_from_fixture(fx) -> WilipModel:
    """the wl_* fields a `ref_dump -Q` fixture carries (oracle/ref_build/ref_dump_common.h: dump_wilip_model)"""
    m = WilipModel()
    lv = np.asarray(fx["wl_levels"]).reshape(3, 12)
    bits = np.asarray(fx["wl_bitpat"])
    ct = np.asarray(fx["wl_convtab"]).reshape(3, -1)
    for i in range(3):
        L = m.level[i]
        (L.elem, L.tpl, L.mask, L.width, L.gain, L.gain1, L.thr, L.xdrp, L.cutoff, L.vthr) = (int(x) for x in lv[i, :10])
        bl, off = int(lv[i, 10]), int(lv[i, 11])
        assert bl <= 32
        L.bitpat_len = bl
        for k in range(bl):
            L.bitpat[k] = int(bits[off + k])
        for c in range(32):
            L.convtab[c] = min(255, int(ct[i, c])) if c < ct.shape[1] else 127
    mx = np.asarray(fx["wl_mtx"])
    m.mtx_rows, m.mtx_cols = int(mx[0]), int(mx[1])
    for k, v in enumerate(mx[2:]):
        m.mtx[k] = int(v)
    g = [int(x) for x in fx["wl_glob"]]
    (m.dvsp, _vab, m.end_bonus, _reppen, _dirrep, m.crs, m.lsg, m.mlt, m.hard_minl, m.hard_maxl, m.minl, m.maxl, m.shortquery,
     _afact, m.met, m.ser, m.ser2, _zzz, m.avrsig, m.llmt, m.min_hit) = g
    if "blk_prm" in fx:
        # `shortquery` is a file-static of src/wln.h:34: every translation unit has a copy of its own.  SrchBlk::initialize
        # (src/blksrc.cc:2219) sets the copy of blksrc.cc -- which the block-search recorder, being that unit, wrote here --
        # to 8 * Ktuple, and findblock reads that one; Wlp::Wlp (src/wln.cc:222) reads the copy of wln.cc, which nobody sets
        m.shortquery = 50
    return m
