"""Oracle-side restatement of the reference's dispatch around the `_wip` engines.
TEST INFRASTRUCTURE ONLY (see oracle/spdp_oracle.c header).

Plain Python over the C oracle engines, one query at a time, written from the
reference semantics independently of spaln_amd/csrc/spdp_host.cpp:
  alignS_ng (ori = 1, -Q0)      src/fwd2s1.cc:2746-2760
  globalS_ng                    src/fwd2s1.cc:2674-2694
  lspS_ng                       src/fwd2s1.cc:1801-1897
  trcbkalignS_ng                src/fwd2s1.cc:1667-1710  (m >= 8: forwardS1_wip)
  mimd_postwork / rcsv_postwork src/fwd2s1.cc:1714-1799
  diagonalS_ng                  src/fwd2s1.cc:1629-1665
  stdskl / trimskl              src/gaps.cc:140-180, 254-273
"""
from __future__ import annotations

import copy
import ctypes as C
import math

import numpy as np

from spaln_amd import abi
from . import oracle

END = abi.END_OF_ULK
NELEM = 16


class ReferenceUndefined(Exception):
    """the reference reads or writes outside its arrays on this input"""


class NeedsScalarEngine(Exception):
    """trcbkalignS_ng with m < 8 uses the scalar exact-ILD engine (not restated yet)."""


def _f32(x):
    return float(np.float32(x))


def _sub(p: abi.Problem, al, ar, bl, br, flags) -> abi.Problem:
    q = abi.Problem()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(abi.Problem))
    q.a_left, q.a_right, q.b_left, q.b_right = al, ar, bl, br
    q.a_exgl, q.a_exgr, q.b_exgl, q.b_exgr = flags
    return q


def _codes(ptr, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,))


def diagonal(sc, p, rec):
    local = bool(sc.local)
    LL = local and p.a_exgl and p.b_exgl
    LR = local and p.a_exgr and p.b_exgr
    dlt = 0 if local else (p.b_right - p.b_left) - (p.a_right - p.a_left)
    a = _codes(p.a, p.a_len)
    b = _codes(p.b, p.b_len)
    al, ar, bl = p.a_left, p.a_right, p.b_left
    if dlt < 0:
        a, b = b, a
        al, ar, bl = p.b_left, p.b_right, p.a_left
    mL, mR, scr, maxh = al, ar, 0, abi.NEVSEL
    dim = sc.mtx_dim
    for m in range(al + 1, ar + 1):
        x, y = int(a[m - 1]), int(b[bl + m - 1 - al])
        scr += sc.mtx[y * dim + x] if dlt < 0 else sc.mtx[x * dim + y]
        if LL and scr < 0:
            scr, mL = 0, m
        if LR and scr > maxh:
            maxh, mR = scr, m
    r = bl - al
    if dlt < 0:
        r -= dlt
    rec.append((mL, mL + r))
    rec.append((mR, mR + r))
    return maxh if LR else scr


def trcbk(sc, p, w, rec, simd=2):
    if w.width < 0:
        return abi.NEVSEL
    if simd == 0 or p.a_right - p.a_left < 8:          # scalar forwardS_ng (src/fwd2s1.cc:1677)
        if not sc.intpen or not p.cano5:
            raise NeedsScalarEngine()
        s, skl = oracle.scalar_forward(sc, p, w)
        rec.extend((int(m), int(n)) for m, n in skl)
        return s
    if simd == 1:                                         # forwardS1 (mode 3 / 5) + Vmf::traceback
        if not sc.intpen or not p.cano5:
            raise NeedsScalarEngine()
        s, skl, flag = oracle.exact_forward(sc, p, w)
        if flag:
            raise ReferenceUndefined("forwardS1: mode 3 pointer beyond int16")
        rec.extend((int(m), int(n)) for m, n in skl)
        return s
    s, skl = oracle.wip_forward(sc, p, w)
    rec.extend((int(m), int(n)) for m, n in skl)
    return s


def lsp(sc, p, w, rec, simd=2):
    m, n = p.a_right - p.a_left, p.b_right - p.b_left
    if not m and not n:
        return 0
    if not m or not n:
        rec.append((p.a_left, p.b_left))
        rec.append((p.a_right, p.b_right))
        if m:
            return sc.gep if (p.a_exgl or p.a_exgr) else sc.gop + m * sc.gep
        return sc.gep if (p.b_exgl or p.b_exgr) else n * sc.gep
    if w.up == w.lw:
        return diagonal(sc, p, rec)
    if abs(n - m) < 8 or m == 1 or n == 1:
        return trcbk(sc, p, w, rec, simd)
    coef_B, coef_C = 2.0, float((sc.noll + 1) * 4)
    if simd < 2:                                          # hexagonal
        k = _f32(w.lw - p.b_left + p.a_right)
        q = _f32(p.b_right - p.a_left - w.up)
        cvol = _f32(_f32(_f32(m) * _f32(n)) - _f32(_f32(_f32(k * k) + _f32(q * q)) / 2))
    else:                                                 # rhombic
        cvol = _f32(_f32(m) * _f32(n + m))
    if _f32(coef_B * cvol) < sc.max_vmf_space:
        return trcbk(sc, p, w, rec, simd)
    recursive = bool(sc.recursive)                      # algmode.alg & 4 (-A4 .. -A7)
    n_imd = 1
    imd_intvl = (m + 1) // 2
    if not recursive:
        z = 2.0 * m * coef_B / coef_C
        imd1 = int(math.pow(z, 1.0 / 3) + 0.5) - 1
        spc = _f32(_f32(_f32(coef_C * n) * imd1) + _f32(_f32(_f32(coef_B * cvol) / (imd1 + 1)) / (imd1 + 1)))
        if spc > sc.max_vmf_space:
            recursive = True
        else:
            imd3 = m // NELEM
            n_imd = sc.ubh if sc.ubh else min(imd1, imd3)
            intvl = imd_intvl = (m + n_imd) // (n_imd + 1)
            if intvl * n_imd == m:
                n_imd -= 1
            if n_imd == 0:
                return trcbk(sc, p, w, rec, simd)
    if simd == 0:
        scr, cpos, rng, flag = oracle.scalar_udh(sc, p, n_imd, imd_intvl, w)
        if flag:
            raise ReferenceUndefined("hirschbergS_ng outside its arrays")
    elif simd == 1:
        if not sc.intpen or not p.cano5:
            raise NeedsScalarEngine("hirschbergS1 without its inputs")
        scr, cpos, rng = oracle.exact_udh(sc, p, n_imd, w)
    else:
        scr, cpos, rng = oracle.wip_udh(sc, p, n_imd, w)
    if scr > abi.NEVSEL:
        cur = _sub(p, int(rng[0]), int(rng[1]), int(rng[2]), int(rng[3]),
                   (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr))
        if cpos[0][0] == END:
            rec.append((cur.a_left, cur.b_left))
            rec.append((cur.a_right, cur.b_right))
        elif recursive:
            rcsv(sc, cur, cpos, rec, simd)
        else:
            mimd(sc, cur, cpos, n_imd, rec, simd)
    return scr


def _slab_window(sc, cur, row, simd):
    """mimd_postwork / rcsv_postwork: stripe() under SIMD, the recorded diagonal bounds under -A0"""
    if simd:
        return oracle.stripe(cur, sc.sh)
    w = abi.Window()
    w.lw, w.up = int(row[8]), int(row[9])
    w.width = w.up - w.lw + 3
    return w


def mimd(sc, cur, cpos, n_imd, rec, simd=2):
    aleft, bleft = cur.a_left, cur.b_left
    cur = _sub(cur, cur.a_left, cur.a_right, cur.b_left, cur.b_right, (0, 0, 0, 0))
    i = n_imd - 1
    while i >= 0 and cpos[i][0] == END:
        i -= 1
    while i >= 0 and cpos[i][0] != END:
        cur.a_left = int(cpos[i][0])
        cur.b_exgl = 1 if cpos[i][1] else 0
        cur.b_left = int(cpos[i][2])
        if cur.b_left < 0 or cur.b_left > cur.b_right:
            break
        c = 3
        while c < 10 and cpos[i][c] < END:
            rec.append((cur.a_left, int(cpos[i][c])))
            c += 1
        trcbk(sc, cur, _slab_window(sc, cur, cpos[i + 1], simd), rec, simd)
        cur.a_right = cur.a_left
        cur.b_right = int(cpos[i][c - 1])
        i -= 1
    if (i < 0 and cpos[0][0] != END) or cpos[0][2] != END:
        cur.a_left, cur.b_left = aleft, bleft
        trcbk(sc, cur, _slab_window(sc, cur, cpos[0], simd), rec, simd)


def rcsv(sc, cur, cpos, rec, simd=2):
    base = _sub(cur, cur.a_left, cur.a_right, cur.b_left, cur.b_right, (0, 0, 0, 0))
    row = cpos[0]
    if row[0] < END:
        c = 2
        while c < 10 and row[c] < END:
            rec.append((int(row[0]), int(row[c])))
            c += 1
        first = _sub(base, base.a_left, int(row[0]), base.b_left, int(row[c - 1]), (0, 0, 0, 0))
        lsp(sc, first, _slab_window(sc, first, cpos[0], simd), rec, simd)
        second = _sub(base, int(row[0]), base.a_right, int(row[2]), base.b_right,
                      (0, 0, 1 if row[1] else 0, 0))
        lsp(sc, second, _slab_window(sc, second, cpos[1], simd), rec, simd)
    elif sc.local:
        trcbk(sc, base, oracle.stripe(base, sc.sh), rec, simd)


def std_skl(rec):
    if len(rec) < 2:
        return list(rec)
    org = sorted(rec)
    out, pr, prv = [], 2, org[0]
    for o in org[1:]:
        dm, dn = o[0] - prv[0], o[1] - prv[1]
        if not dm and not dn:
            continue
        if dm < 0 or dn < 0:
            continue
        dd = min(dm, dn)
        df = dn - dm
        if df:
            df = 1 if df > 0 else -1
        if dd and df:
            if pr:
                out.append(prv)
            out.append((prv[0] + dd, prv[1] + dd))
        elif df != pr or not dm:
            out.append(prv)
        pr, prv = df, o
    out.append(prv)
    return out


def trim_skl(s, p):
    s = list(s)
    if len(s) >= 2:
        i, j = s[1][0] - s[0][0], s[1][1] - s[0][1]
        if (p.a_exgl and not i) or (p.b_exgl and not j):
            s.pop(0)
    if len(s) >= 2:
        i, j = s[-1][0] - s[-2][0], s[-1][1] - s[-2][1]
        if (p.a_exgr and not i) or (p.b_exgr and not j):
            s.pop()
    return s


def homscore_s(sc, p, simd=2):
    """HomScoreS_ng (src/fwd2s1.cc:2696-2716): scalar below 4 rows or under -A0, else scoreonlyS1_wip"""
    w = oracle.stripe(p, sc.sh)
    if simd == 0 or p.a_right - p.a_left < 4:
        if not sc.intpen or not p.cano5:
            raise NeedsScalarEngine()
        return oracle.scalar_scorealone(sc, p, w)
    if simd == 1:                                         # -A1: scoreonlyS1 (mode 3 / 5 do not change the score)
        if not sc.intpen or not p.cano5:
            raise NeedsScalarEngine()
        return oracle.exact_scoreonly(sc, p, w)
    return oracle.wip_scoreonly(sc, p, w)


def align_s_ori3(sc, p_fwd, p_rev, simd=2):
    """alignS_ng(ori = 3) with seeding off: infer_orientation (src/fwd2s1.cc:2718-2730) + one alignment"""
    ori = 1 if homscore_s(sc, p_rev, simd) > homscore_s(sc, p_fwd, simd) else 0
    scr, flat = align_s(sc, p_rev if ori else p_fwd, simd)
    if ori and flat:
        flat[0] |= 0x10                                   # A_RevCom: a->inex.sens after comrev (src/fwd2s1.cc:2692)
    return (scr, flat), ori


def align_s(sc, p, simd=2):
    """alignS_ng(ori=1) with seeding off.  Returns (score, skl) with skl = [flags, n, m1, n1, ...] or None.
    simd = algmode.alg & 3: 0 runs the scalar engines (forwardS_ng, hirschbergS_ng) throughout."""
    rec = []
    scr = lsp(sc, p, oracle.stripe(p, sc.sh), rec, simd)
    if len(rec) < 2:
        return scr, None
    s = trim_skl(std_skl(rec), p)
    flat = [1, len(s)]
    for m, n in s:
        flat += [m, n]
    return scr, flat


# ---- skl_rngS_ng (src/fwd2s1.cc:446-693): rescoring of a finished alignment -----------------------
# The value the CLI prints ("S:") and the per-exon records come from here, not from the engines.
# Restated without the output-format side channels (Cigar / Vulgar / SAM) and without the query's
# intron-position profile (`PfqItr`: empty unless the query carries one).
EIJ_FIELDS = ["left", "right", "rleft", "rright", "mch", "mmc", "gap", "unp", "mch5", "mmc5", "gap5", "unp5",
              "mch3", "mmc3", "gap3", "unp3", "phs", "escr", "iscr", "sig3", "sig5"]
ENDRNG = 2 ** 31 - 1


def skl_rng_s(sc, p, skl, *, codonk1, minl, jneibr, lsg=1):
    """skl = [flags, n, m1, n1, ...] as align_s returns it.  Returns (h, fstat[5], [21-int records])."""
    a = _codes(p.a, p.a_len)
    b = _codes(p.b, p.b_len)
    sig5 = np.ctypeslib.as_array(C.cast(p.sig5, C.POINTER(C.c_int16)), shape=(p.b_len + 1,))
    sig3 = np.ctypeslib.as_array(C.cast(p.sig3, C.POINTER(C.c_int16)), shape=(p.b_len + 1,))
    dinc = _codes(p.dinc, p.b_len + 1)
    intpen = np.ctypeslib.as_array(C.cast(sc.intpen, C.POINTER(C.c_int16)), shape=(sc.intpen_len,))
    mtx = np.array(sc.mtx[:sc.mtx_dim * sc.mtx_dim]).reshape(sc.mtx_dim, sc.mtx_dim)

    def gap_penalty(i):                                  # PwdB::GapPenalty, src/aln.h:275
        if i == 0:
            return 0
        return sc.lgop + i * sc.lgep if i > codonk1 else sc.gop + i * sc.gep

    def spjscr(n5, n3):                                  # SpJunc::spjscr = IntPen(len) + sig53(IE53)
        return int(intpen[n3 - n5]) + int(sig3[n3]) + int(sc.t53[16 * (dinc[n5] >> 4) + (dinc[n3] & 15)])

    corners = [(skl[2 + 2 * i], skl[3 + 2 * i]) for i in range(skl[1])]
    num = len(corners)
    w = 0
    h = ha = hb = 0
    s5 = s3 = 0
    insert = deletn = intlen = preint = 0
    fst = dict(mch=0, mmc=0, gap=0, unp=0, val=0)
    pst = dict(fst)
    psp = 0
    rbuf = dict.fromkeys(EIJ_FIELDS, 0)
    recs = []
    que = [dict(fst) for _ in range(jneibr)]            # Eijnc(true): ring of the last jneibr statistics
    qpos = [0]

    def shift(near):                                     # Eijnc::shift, src/gsinfo.cc:1255
        if near:
            for k in ("mch", "mmc", "unp", "gap"):
                rbuf[k + "5"] = fst[k] - que[qpos[0]][k]
        que[qpos[0]] = dict(fst)
        qpos[0] = (qpos[0] + 1) % jneibr

    def store(prv, near):                                # Eijnc::store, :1237
        for k in ("mch", "mmc", "gap", "unp"):
            rbuf[k] = fst[k] - prv[k]
        if near:
            for k in ("mch", "mmc", "gap", "unp"):
                rbuf[k + "5"] = rbuf[k]
        for k in ("mch", "mmc", "unp", "gap"):
            rbuf[k + "3"] = fst[k] - que[qpos[0]][k]

    def push():
        recs.append([int(rbuf[k]) for k in EIJ_FIELDS])

    if num >= 2 and corners[1][1] == corners[0][1] and p.b_exgl:
        w += 1
        num -= 1
    m, n = corners[w]
    ai, bi = m, n                                        # as = a->at(m), bs = b->at(n)
    rbuf["left"], rbuf["rleft"], rbuf["iscr"], rbuf["sig3"] = n, m, abi.NEVSEL, 0
    left = num
    while left > 1:
        left -= 1
        w += 1
        wm, wn = corners[w]
        mi = wm - m
        if mi and insert:
            j = p.a_exgl and m == p.a_left
            x = 0 if j else gap_penalty(insert)
            xi = abi.NEVSEL
            if intlen:
                insert -= intlen
                xi = rbuf["iscr"] + gap_penalty(insert)
            if xi >= x:                                  # intron
                hb = ha
                if rbuf["right"] - rbuf["left"] > 0:
                    push()
                rbuf["left"] = rbuf["right"] + intlen
                rbuf["rleft"] = m
                rbuf["sig3"] = s3
                rbuf["iscr"] = abi.NEVSEL
                h += xi
                insert -= preint
            else:
                h += x
            if insert:
                insert = intlen = preint = 0
        ni = wn - n
        if ni and deletn:
            if not (p.b_exgl and n == p.b_left):
                h += gap_penalty(deletn)
                fst["gap"] += 1
            ai += deletn
            deletn = 0
        i = mi - ni
        d = ni if i >= 0 else mi
        if d:
            m += d
            x = 0
            for _ in range(d):
                shift(psp == jneibr)
                psp += 1
                x += int(mtx[a[ai], b[bi]])
                if a[ai] == b[bi]:
                    fst["mch"] += 1
                else:
                    fst["mmc"] += 1
                ai += 1
                bi += 1
                n += 1
            h += x
            fst["val"] += x
        if i > 0:
            deletn += i
            for _ in range(i):
                shift(psp == jneibr)
                psp += 1
                fst["unp"] += 1
        elif i < 0:
            i = -i
            n3 = n + i
            if lsg and i > minl:
                s5 = int(sig5[n])
                s3 = int(sig3[n3])
                xi = s5 + spjscr(n, n3)
                if p.cip:                               # use_spb(): PfqItr::match_score(m) = Cip_score::cip_score(m) (:615)
                    xi += int(C.cast(p.cip, C.POINTER(C.c_int32))[m]) if 0 <= m <= p.a_len else 0
            else:
                xi = abi.NEVSEL
            if xi > gap_penalty(i) and xi > rbuf["iscr"]:
                preint = insert                         # intron
                intlen = i
                rbuf["right"], rbuf["rright"], rbuf["iscr"] = n, m, xi
                rbuf["escr"] = h + s5 - hb
                rbuf["sig5"] = s5
                ha = h + xi - s3
                store(pst, psp < jneibr)
                pst = dict(fst)
                psp = 0
            elif not p.a_exgl or m != p.a_left:
                if not insert:
                    fst["gap"] += 1
                for _ in range(i):
                    shift(psp == jneibr)
                    psp += 1
                    fst["unp"] += 1
                    n += 1
            bi += i
            insert += i
        m, n = wm, wn
    if insert and not (p.a_exgr and m == p.a_right):
        h += gap_penalty(insert)
        fst["gap"] += 1
        fst["unp"] += insert
    if deletn and not (p.b_exgr and n == p.b_right):
        h += gap_penalty(deletn)
        fst["gap"] += 1
        fst["unp"] += deletn
    rbuf["escr"] = h - hb
    rbuf["iscr"] = 0
    rbuf["sig5"] = 0
    rbuf["right"], rbuf["rright"] = n, m
    store(pst, n - rbuf["left"] <= jneibr)
    push()
    rbuf["left"] = rbuf["right"] = ENDRNG
    push()
    fst["val"] += sc.gop * fst["gap"] + sc.gep * fst["unp"]
    return h, [fst[k] for k in ("mch", "mmc", "gap", "unp", "val")], recs
