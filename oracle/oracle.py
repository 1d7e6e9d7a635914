"""ctypes wrapper of liboracle.so (oracle/spdp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (spaln_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

from spaln_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("spdp_oracle.c", "spdp_oracle_scalar.c", "spdp_oracle_h.c",
                                                "spdp_oracle_h_scalar.c", "spdp_oracle_blk.c", "spdp_oracle_blkidx.c")]
    hdr = os.path.join(_HERE, "..", "include", "spdp.h")
    newest = max(os.path.getmtime(f) for f in srcs + [hdr])
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        tmp = f"{_SO}.{os.getpid()}.tmp"          # several test processes may get here at once: build aside, swap in
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", tmp] + srcs + ["-lm"])
        os.replace(tmp, _SO)
    build_walk_check(force)
    build_blk_check(force)
    return _SO


_BLK_SO = os.path.join(_HERE, "libblkcheck.so")


def build_blk_check(force: bool = False) -> str:
    """the block search's CPU checker: the product's host-side logic (spdp_loci.h, the HSP search in its host form), host-compiled"""
    d = os.path.join(_HERE, "..", "spaln_amd", "csrc")
    srcs = [os.path.join(_HERE, "blk_check.cpp"), os.path.join(d, "spdp_loci.h"), os.path.join(d, "spdp_hsp_host.h"),
            os.path.join(d, "spdp_hsp_chain.h"), os.path.join(d, "spdp_region.h"), os.path.join(d, "spdp_gencode.h"),
            os.path.join(_HERE, "..", "include", "spdp.h")]
    newest = max(os.path.getmtime(f) for f in srcs)
    if force or not os.path.exists(_BLK_SO) or os.path.getmtime(_BLK_SO) < newest:
        tmp = f"{_BLK_SO}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", tmp, srcs[0]])
        os.replace(tmp, _BLK_SO)
    return _BLK_SO


_WALK_SO = os.path.join(_HERE, "libwalkcheck.so")


def build_walk_check(force: bool = False) -> str:
    """the seeded walk's CPU checker: the product's host walk (a header) compiled with callbacks in place of the device"""
    srcs = [os.path.join(_HERE, "walk_check.cpp"), os.path.join(_HERE, "..", "spaln_amd", "csrc", "spdp_walk.h"),
            os.path.join(_HERE, "..", "spaln_amd", "csrc", "spdp_seeded_rv.h"), os.path.join(_HERE, "..", "spaln_amd", "csrc", "spdp_hostcpus.h"),
            os.path.join(_HERE, "..", "spaln_amd", "csrc", "spdp_gencode.h"), os.path.join(_HERE, "..", "include", "spdp.h"),
            os.path.join(_HERE, "..", "spaln_amd", "csrc", "spdp_hsp_host.h"), os.path.join(_HERE, "..", "spaln_amd", "csrc", "spdp_hsp_chain.h")]
    newest = max(os.path.getmtime(f) for f in srcs)
    if force or not os.path.exists(_WALK_SO) or os.path.getmtime(_WALK_SO) < newest:
        tmp = f"{_WALK_SO}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", tmp, srcs[0]])
        os.replace(tmp, _WALK_SO)
    return _WALK_SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_cells.restype = C.c_int64
        _lib.orc_cells_h.restype = C.c_int64
    return _lib


def stripe(p: abi.Problem, sh: int) -> abi.Window:
    w = abi.Window()
    lib().orc_stripe(C.byref(p), C.c_int(sh), C.byref(w))
    return w


def cells(p: abi.Problem, w: abi.Window) -> int:
    return int(lib().orc_cells(C.byref(p), C.byref(w)))


def wip_scoreonly(sc: abi.Scoring, p: abi.Problem, w: abi.Window | None = None) -> int:
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    rc = lib().orc_wip_scoreonly(C.byref(sc), C.byref(p), C.byref(w), C.byref(s))
    if rc:
        raise RuntimeError(f"orc_wip_scoreonly rc={rc}")
    return s.value


def wip_forward(sc, p, w=None):
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_wip_forward(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), C.byref(skl), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_wip_forward rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    C.CDLL(None).free(skl)
    return s.value, out


def wip_udh(sc, p, n_im: int, w=None):
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    cpos = np.zeros((n_im + 1, 10), dtype=np.int32)
    cpos[:, 0] = abi.END_OF_ULK
    cpos[:, 2] = abi.END_OF_ULK
    rng = np.zeros(4, dtype=np.int32)
    rc = lib().orc_wip_udh(C.byref(sc), C.byref(p), C.byref(w), C.c_int(n_im), C.byref(s),
                           cpos.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"orc_wip_udh rc={rc}")
    return s.value, cpos, rng


def scalar_scorealone(sc, p, w=None) -> int:
    """Aln2s1::scorealoneS_ng (the -A0 HomScoreS_ng engine)."""
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    rc = lib().orc_scalar_scorealone(C.byref(sc), C.byref(p), C.byref(w), C.byref(s))
    if rc:
        raise RuntimeError(f"orc_scalar_scorealone rc={rc}")
    return s.value


def scalar_forward(sc, p, w=None):
    """Aln2s1::forwardS_ng + the record hand-over of trcbkalignS_ng."""
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_scalar_forward(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), C.byref(skl), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_scalar_forward rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    if n.value:
        C.CDLL(None).free(skl)
    return s.value, out


def scalar_forward_cut(sc, p, w, cut_l: int, cut_r: int):
    """trcbkalignS_ng(wdw, spj, mc): forwardS_ng jumping over the genomic range (cut_l, cut_r] (shortcutS_ng)."""
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_scalar_forward_cut(C.byref(sc), C.byref(p), C.byref(w), C.c_int(cut_l), C.c_int(cut_r),
                                      C.byref(s), C.byref(skl), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_scalar_forward_cut rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    if n.value:
        C.CDLL(None).free(skl)
    return s.value, out


def scalar_udh(sc, p, n_im: int, imd_intvl: int, w=None):
    """Aln2s1::hirschbergS_ng: (score, cpos rows, written-back ranges, flag); flag -3: the reference
    reads / writes outside its arrays on this input (undefined)."""
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    cpos = np.full((n_im + 1, 10), abi.END_OF_ULK, dtype=np.int32)
    rng = np.zeros(4, dtype=np.int32)
    rc = lib().orc_scalar_udh(C.byref(sc), C.byref(p), C.byref(w), C.c_int(n_im), C.c_int(imd_intvl), C.byref(s),
                              cpos.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    if rc not in (0, -3):
        raise RuntimeError(f"orc_scalar_udh rc={rc}")
    return s.value, cpos, rng, rc


def exact_scoreonly(sc, p, w=None) -> int:
    """SimdAln2s1::scoreonlyS1 (the -A1 HomScoreS_ng engine: vector H / E / F with exact intron lists)."""
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    rc = lib().orc_exact_scoreonly(C.byref(sc), C.byref(p), C.byref(w), C.byref(s))
    if rc:
        raise RuntimeError(f"orc_exact_scoreonly rc={rc}")
    return s.value


def exact_forward(sc, p, w=None):
    """SimdAln2s1::forwardS1 + Vmf traceback + the record hand-over of trcbkalignS_ng (-A1):
    (score, records end -> start, flag); flag -3: mode 3 pointer overflow in the reference (undefined)."""
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_exact_forward(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), C.byref(skl), C.byref(n))
    if rc not in (0, -3):
        raise RuntimeError(f"orc_exact_forward rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    if n.value:
        C.CDLL(None).free(skl)
    return s.value, out, rc


def exact_udh(sc, p, n_im: int, w=None):
    """SimdAln2s1::hirschbergS1 (-A1): (score, cpos rows, written-back ranges)"""
    w = w or stripe(p, sc.sh)
    s = C.c_int32()
    cpos = np.zeros((n_im + 1, 10), dtype=np.int32)
    cpos[:, 0] = abi.END_OF_ULK
    cpos[:, 2] = abi.END_OF_ULK
    rng = np.zeros(4, dtype=np.int32)
    rc = lib().orc_exact_udh(C.byref(sc), C.byref(p), C.byref(w), C.c_int(n_im), C.byref(s),
                             cpos.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"orc_exact_udh rc={rc}")
    return s.value, cpos, rng


# ---- protein x genome ------------------------------------------------------------------
def stripe31(p: abi.ProblemH, sh: int) -> abi.Window:
    w = abi.Window()
    lib().orc_stripe31(C.byref(p), C.c_int(sh), C.byref(w))
    return w


def cells_h(p: abi.ProblemH, w: abi.Window) -> int:
    return int(lib().orc_cells_h(C.byref(p), C.byref(w)))


def wip_forward_h(sc: abi.ScoringH, p: abi.ProblemH, w=None):
    """SimdAln2h1::forwardH1_wip(mfd): (score, records end -> start, flag); flag 0 ok, -2 the
    reference's fatal "Unexpected dir", -3 its traceback starts outside the bitmap (undefined)."""
    w = w or stripe31(p, sc.sh)
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_wip_forward_h(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), C.byref(skl), C.byref(n))
    if rc not in (0, -2, -3):
        raise RuntimeError(f"orc_wip_forward_h rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    C.CDLL(None).free(skl)
    return s.value, out, rc


def wip_udh_h(sc, p, n_im: int, w=None):
    """SimdAln2h1::hirschbergH1_wip: (score, cpos rows, written-back ranges)"""
    w = w or stripe31(p, sc.sh)
    s = C.c_int32()
    cpos = np.full((n_im + 1, 10), abi.END_OF_ULK, dtype=np.int32)
    rng = np.zeros(4, dtype=np.int32)
    rc = lib().orc_wip_udh_h(C.byref(sc), C.byref(p), C.byref(w), C.c_int(n_im), C.byref(s),
                             cpos.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"orc_wip_udh_h rc={rc}")
    return s.value, cpos, rng


def exact_forward_h(sc, p, w=None):
    """SimdAln2h1::forwardH1 + Vmf traceback + the record hand-over of trcbkalignH_ng (-A1):
    (score, records end -> start, flag); flag -3: mode 3 pointer overflow in the reference (undefined)."""
    w = w or stripe31(p, sc.sh)
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_exact_forward_h(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), C.byref(skl), C.byref(n))
    if rc not in (0, -3):
        raise RuntimeError(f"orc_exact_forward_h rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    if n.value:
        C.CDLL(None).free(skl)
    return s.value, out, rc


def exact_udh_h(sc, p, n_im: int, w=None):
    """SimdAln2h1::hirschbergH1 (-A1): (score, cpos rows, written-back ranges)"""
    w = w or stripe31(p, sc.sh)
    s = C.c_int32()
    cpos = np.full((n_im + 1, 10), abi.END_OF_ULK, dtype=np.int32)
    rng = np.zeros(4, dtype=np.int32)
    rc = lib().orc_exact_udh_h(C.byref(sc), C.byref(p), C.byref(w), C.c_int(n_im), C.byref(s),
                               cpos.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"orc_exact_udh_h rc={rc}")
    return s.value, cpos, rng


def scalar_forward_h(sc: abi.ScoringH, p: abi.ProblemH, w=None, traceback=True):
    """Aln2h1::forwardH_ng (+ the record hand-over of trcbkalignH_ng when traceback): the -A0 engine,
    also the -A2/-A3 fallback below 8 query rows.  Returns (score, records end -> start)."""
    w = w or stripe31(p, sc.sh)
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    if traceback:
        rc = lib().orc_scalar_forward_h(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), C.byref(skl), C.byref(n))
    else:
        rc = lib().orc_scalar_forward_h(C.byref(sc), C.byref(p), C.byref(w), C.byref(s), None, None)
    if rc:
        raise RuntimeError(f"orc_scalar_forward_h rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    if n.value:
        C.CDLL(None).free(skl)
    return s.value, out


def scalar_forward_h_cut(sc: abi.ScoringH, p: abi.ProblemH, w, cut):
    """forwardH_ng with a cut range (shortcutH_ng): the sweep jumps over genomic columns (cut[0], cut[1]]"""
    s = C.c_int32()
    n = C.c_int32()
    skl = C.POINTER(abi.Skl)()
    rc = lib().orc_scalar_forward_h_cut(C.byref(sc), C.byref(p), C.byref(w), C.c_int(cut[0]), C.c_int(cut[1]),
                                        C.byref(s), C.byref(skl), C.byref(n))
    if rc:
        raise RuntimeError(f"orc_scalar_forward_h_cut rc={rc}")
    out = np.array([(skl[i].m, skl[i].n) for i in range(n.value)], dtype=np.int32).reshape(-1, 2)
    if n.value:
        C.CDLL(None).free(skl)
    return s.value, out


def scalar_udh_h(sc, p, n_im: int, imd_intvl: int, w=None):
    """Aln2h1::hirschbergH_ng: (score, cpos rows, written-back ranges, flag); flag -3: the reference
    indexes outside its arrays on this input (undefined)."""
    w = w or stripe31(p, sc.sh)
    s = C.c_int32()
    cpos = np.full((n_im + 1, 10), abi.END_OF_ULK, dtype=np.int32)
    rng = np.zeros(4, dtype=np.int32)
    rc = lib().orc_scalar_udh_h(C.byref(sc), C.byref(p), C.byref(w), C.c_int(n_im), C.c_int(imd_intvl), C.byref(s),
                                cpos.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    if rc not in (0, -3):
        raise RuntimeError(f"orc_scalar_udh_h rc={rc}")
    return s.value, cpos, rng, rc
