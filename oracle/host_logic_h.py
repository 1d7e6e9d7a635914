"""Oracle-side restatement of the reference's dispatch around the aa x genome `_wip` engine.
TEST INFRASTRUCTURE ONLY (see oracle/spdp_oracle.c header).

  HomScoreH_ng (-A2/-A3)        src/fwd2h1.cc:3288-3308
  alignH_ng / globalH_ng (-Q0)  src/fwd2h1.cc:3310-3316, 3267-3286
  lspH_ng                       src/fwd2h1.cc:2140-2231   (traceback branch; UDH not restated)
  trcbkalignH_ng                src/fwd2h1.cc:1997-2041   (m >= 8: forwardH1_wip)
  stdskl3                       src/gaps.cc:178-227       (UNITE_INDEL_FS = 0)
"""
from __future__ import annotations

import numpy as np

from spaln_amd import abi
from . import oracle

NELEM = 16
COEF_B = 2.0                     # sizeof(short), fwd2h1.cc:60


class NotRestated(Exception):
    """a branch of the reference this restatement does not cover (scalar engine, UDH, diagonal)"""


class ReferenceFatal(Exception):
    """the reference stops with fatal("Unexpected dir") on this input"""


class ReferenceUndefined(Exception):
    """the reference starts its traceback outside the bitmap (out-of-bounds read) on this input"""


def homscore_h(sc: abi.ScoringH, p: abi.ProblemH) -> int:
    if p.a_right - p.a_left < 8:
        raise NotRestated("forwardH_ng (m < 8)")
    s, _, _ = oracle.wip_forward_h(sc, p, oracle.stripe31(p, sc.sh))
    return s


def trcbk_h(sc, p, w, rec):
    if w.width < 0:
        return abi.NEVSEL
    if p.a_right - p.a_left < 8:
        raise NotRestated("forwardH_ng (m < 8)")
    s, skl, bad = oracle.wip_forward_h(sc, p, w)
    if bad == -2:
        raise ReferenceFatal("Unexpected dir")
    if bad == -3:
        raise ReferenceUndefined("start cell outside the traceback bitmap")
    rec.extend((int(m), int(n)) for m, n in skl)
    return s


def lsp_h(sc, p, w, rec):
    m = p.a_right - p.a_left
    n = p.b_right - p.b_left
    if not m and not n:
        return 0
    if not m or not n:
        raise NotRestated("empty range")
    if w.up == w.lw:
        raise NotRestated("diagonalH_ng")
    if abs(n - m) < NELEM or m == 1 or n <= 3:
        return trcbk_h(sc, p, w, rec)
    cvol = float(np.float32(m) * np.float32(n + 3 * m))
    if COEF_B * cvol < sc.max_vmf_space:
        return trcbk_h(sc, p, w, rec)
    raise NotRestated("hirschbergH1_wip")


def std_skl3(rec):
    if len(rec) < 2:
        return list(rec)
    org = sorted(rec)
    out, pr, prv = [], -2, org[0]
    for o in org[1:]:
        dm, dn = (o[0] - prv[0]) * 3, o[1] - prv[1]
        if not dm and not dn:
            continue
        if dn < 0:
            continue
        dd = min(dm, dn)
        df = dn - dm
        dr = (1 if df > 0 else -1) if df else 0
        if dd and df:
            if pr:
                out.append(prv)
            bn = prv[1] + dd
            if df < 0 and _cmod(df, 3):
                dd += 2
            bm = prv[0] + _cdiv(dd, 3)
            out.append((bm, bn))
            if df > 0 and _cmod(df, 3):
                out.append((bm, bn + _cmod(df, 3)))
        elif dr != pr or not dm:
            out.append(prv)
        pr, prv = dr, o
    out.append(prv)
    return out


def _cdiv(a, b):                 # C integer division (truncation toward zero)
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _cmod(a, b):
    return a - _cdiv(a, b) * b


def align_h(sc, p):
    """alignH_ng with seeding off.  Returns (score, [flags, n, m1, n1, ...] or None)."""
    rec = []
    scr = lsp_h(sc, p, oracle.stripe31(p, sc.sh), rec)
    if len(rec) < 2:
        return scr, None
    s = std_skl3(rec)
    flat = [1, len(s)]
    for m, n in s:
        flat += [m, n]
    return scr, flat
