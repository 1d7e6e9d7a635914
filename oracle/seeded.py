"""The seeded path (alignS_ng with algmode.qck > 0) on the CPU, for the tests.  TEST INFRASTRUCTURE ONLY.

The walk itself (seededS_ng / interpolateS and the closed-form joins, src/fwd2s1.cc:1899-2672) is the product's host
code, spaln_amd/csrc/spdp_seeded_walk.h, compiled into oracle/libwalkcheck.so with callbacks where the product has the
device.  Here those callbacks are bound to the oracle: lspS_ng and trcbkalignS_ng to host_logic.lsp / trcbk over the C
engines, a trcbkalignS_ng with a cut range to the scalar forward sweep with that range, Wilip to a reply table (what a
`ref_dump -Q` fixture recorded from the reference's own walk).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from spaln_amd import abi
from . import oracle, host_logic

_lib = None
_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32)),
                  C.POINTER(C.c_int32))


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(oracle.build_walk_check())
    return _lib


def parse_wilip_log(log, strand: int = 0) -> dict:
    """seed_wilip_A* of a fixture -> {(level, a_left, a_right, b_left, b_right): flat unit record} for the walk on the pair
    as given (strand 0) or on the reverse-complemented pair of an ori = 3 run (strand 1: the harness adds 16 to the level)"""
    log = [int(x) for x in log]
    out, at = {}, 0
    while at < len(log):
        key = (log[at] & 15,) + tuple(log[at + 1:at + 5])
        mine = (log[at] >> 4) == strand
        n_units = log[at + 5]
        at += 6
        flat = [n_units]
        for _ in range(n_units):
            num = log[at]
            flat += log[at:at + 6 + 5 * (num + 1)]
            at += 6 + 5 * (num + 1)
        if mine:
            out.setdefault(key, flat)
    return out


def turned_hsps(hsps, n_hsps: int, a_len: int, b_len: int):
    """Seq::revjxt (src/seq.cc:745-755) on a list whose free slot holds {a->len, b->len}, as the forward walk leaves it"""
    j = np.array(hsps, dtype=np.int32).reshape(-1, 5).copy()
    if n_hsps:
        j[:n_hsps, 0] = a_len - j[:n_hsps, 0] - j[:n_hsps, 2]
        j[:n_hsps, 1] = b_len - j[:n_hsps, 1] - j[:n_hsps, 2]
        j[:n_hsps] = j[:n_hsps][::-1]
    return j


def align_s_seeded_ori3(sc, sp, p_fwd, p_rev, hsps, n_hsps, lowest_level, wilip_fwd, wilip_rev, simd=2):
    """alignS_ng(seqs, pwd, gsi, 3) with seeding on (src/fwd2s1.cc:2762-2777): (score, flat SKL, reverse taken)"""
    sf, ff, _ = align_s_seeded(sc, sp, p_fwd, hsps, n_hsps, lowest_level, wilip_fwd, simd)
    sr, fr, _ = align_s_seeded(sc, sp, p_rev, turned_hsps(hsps, n_hsps, p_fwd.a_len, p_fwd.b_len), n_hsps, lowest_level,
                               wilip_rev, simd)
    if sf >= sr:
        return sf, ff, 0
    if fr:
        fr = [fr[0] | 0x10] + fr[1:]
    return sr, fr, 1


def hsps_of(fx: dict):
    j = np.asarray(fx["seed_jxt"], dtype=np.int32).reshape(-1, 5)
    n = int(fx["seed_params"][2])
    assert j.shape[0] == (n + 1 if n else 0) or (n == 0 and j.shape[0] <= 1)
    return j, n


JOINS = ["abut", "diagonal", "head_cont", "head_short", "head_nogenome", "head_extend", "head_exon", "tail_short",
         "tail_nogenome", "tail_extend", "tail_exon", "junction", "micro_exon", "shortcut", "backforth", "small_dp",
         "recurse", "dp", "giveup_localc", "giveup_head", "giveup_tail", "giveup_inner", "pick_unit"]


def align_s_seeded(sc, sp, p, hsps, n_hsps: int, lowest_level: int, wilip: dict, simd: int = 2, trace=None, joins=None):
    """alignS_ng(ori = 1) with seeding on: (gsi->scr, flat SKL [flags, n, m0, n0, ...] or None, status)"""
    keep = []

    def cb(_user, kind, args, out, n_out):
        a = [args[i] for i in range(15)]
        try:
            if kind == 2:
                key = (a[14], a[0], a[1], a[2], a[3])
                if key not in wilip:
                    raise KeyError(f"no recorded Wilip reply for {key}")
                data = wilip[key]
            else:
                q = host_logic._sub(p, a[0], a[1], a[2], a[3], tuple(a[4:8]))
                w = abi.Window()
                w.lw, w.up, w.width = a[8], a[9], a[10]
                rec = []
                if kind == 0:
                    scr = host_logic.lsp(sc, q, w, rec, simd)
                elif a[11]:
                    scr, skl = oracle.scalar_forward_cut(sc, q, w, a[12], a[13])
                    rec = [(int(m), int(n)) for m, n in skl]
                else:
                    scr = host_logic.trcbk(sc, q, w, rec, simd)
                data = [int(scr)] + [int(x) for mn in rec for x in mn]
            if trace is not None:
                trace.append((kind, a, list(data)))
        except Exception as e:                                   # noqa: BLE001 -- reported through the return code
            keep.append(e)
            return 1
        arr = np.asarray(data, dtype=np.int32)
        keep.append(arr)
        out[0] = arr.ctypes.data_as(C.POINTER(C.c_int32))
        n_out[0] = arr.size
        return 0

    fn = _FN(cb)
    jx = np.ascontiguousarray(hsps, dtype=np.int32)
    cap = 1 << 16
    rec = (abi.Skl * cap)()
    score, n_rec = C.c_int32(), C.c_int()
    jn = (C.c_int32 * lib().walk_check_n_joins())()
    rc = lib().walk_check_run(C.byref(sc), C.byref(sp), C.byref(p), jx.ctypes.data_as(C.c_void_p), C.c_int(n_hsps),
                              C.c_int(lowest_level), fn, None, C.byref(score), rec, C.c_int(cap), C.byref(n_rec), jn)
    if joins is not None:
        for k, v in enumerate(jn):
            joins[JOINS[k]] = joins.get(JOINS[k], 0) + int(v)
    errs = [e for e in keep if isinstance(e, Exception)]
    if errs:
        raise errs[0]
    if rc < 0:
        raise RuntimeError(f"walk_check_run rc={rc}")
    recs = [(rec[i].m, rec[i].n) for i in range(1, n_rec.value)]      # [0] is globalS_ng's dummy record
    if len(recs) < 2:
        return score.value, None, rc
    fin = host_logic.trim_skl(host_logic.std_skl(recs), p)
    return score.value, [1, len(fin)] + [x for mn in fin for x in mn], rc


def marks_changed(fx, marks: dict) -> dict:
    """{position: [phs5, phs3]} of the positions whose marks differ from the fixture's inputs -- the form of seed_marks_A*"""
    p5, p3 = fx["phs5"], fx["phs3"]
    out = {}
    for n, v in marks.items():
        a = int(p5[n]) if v[0] is None else v[0]
        b = int(p3[n]) if v[1] is None else v[1]
        if a != int(p5[n]) or b != int(p3[n]):
            out[n] = [a, b]
    return out


JOINS_H = ["diagonal", "head_nogenome", "head_cds", "head_exon", "tail_nogenome", "tail_cds", "tail_exon", "junction",
           "micro_exon", "shortcut", "backforth", "small_dp", "recurse", "dp", "giveup_head", "giveup_tail", "giveup_inner",
           "pick_unit", "exact_head", "exact_tail"]


def align_h_seeded(sc, sp, p, hsps, n_hsps: int, lowest_level: int, wilip: dict, simd: int = 2, trace=None, joins=None, marks=None):
    """alignH_ng with seeding on (the product's protein walk, spdp_seeded_walk_h.h, over the oracle's ladder):
    (gsi->scr, flat SKL or None, status); status 1 = the walk met a join it does not serve"""
    from . import host_logic_h as hh
    keep = []

    def cb(_user, kind, args, out, n_out):
        a = [args[i] for i in range(15)]
        try:
            if kind == 2:
                key = (a[14], a[0], a[1], a[2], a[3])
                if key not in wilip:
                    raise KeyError(f"no recorded Wilip reply for {key}")
                data = wilip[key]
            else:
                q = hh._sub(p, a[0], a[1], a[2], a[3], tuple(a[4:8]))
                w = abi.Window()
                w.lw, w.up, w.width = a[8], a[9], a[10]
                rec = []
                cut = (a[12], a[13]) if a[11] else None
                scr = hh.lsp_h(sc, q, w, rec, simd) if kind == 0 else hh.trcbk_h(sc, q, w, rec, simd, cut, spj=kind == 1)
                data = [int(scr)] + [int(x) for mn in rec for x in mn]
            if trace is not None:
                trace.append((kind, a, list(data)))
        except Exception as e:                                   # noqa: BLE001
            keep.append(e)
            return 1
        arr = np.asarray(data, dtype=np.int32)
        keep.append(arr)
        out[0] = arr.ctypes.data_as(C.POINTER(C.c_int32))
        n_out[0] = arr.size
        return 0

    fn = _FN(cb)
    jx = np.ascontiguousarray(hsps, dtype=np.int32)
    cap = 1 << 16
    rec = (abi.Skl * cap)()
    score, n_rec = C.c_int32(), C.c_int()
    jn = (C.c_int32 * lib().walk_check_n_joins_h())()
    rc = lib().walk_check_run_h(C.byref(sc), C.byref(sp), C.byref(p), jx.ctypes.data_as(C.c_void_p), C.c_int(n_hsps),
                                C.c_int(lowest_level), fn, None, C.byref(score), rec, C.c_int(cap), C.byref(n_rec), jn)
    if joins is not None:
        for k, v in enumerate(jn):
            joins[JOINS_H[k]] = joins.get(JOINS_H[k], 0) + int(v)
    errs = [e for e in keep if isinstance(e, Exception)]
    if errs:
        raise errs[0]
    if rc < 0:
        raise RuntimeError(f"walk_check_run_h rc={rc}")
    if marks is not None:                                        # the phases the walk wrote at junctions of its own choice
        mk = C.POINTER(C.c_int32)()
        for k in range(lib().walk_check_marks_h(C.byref(mk))):
            marks.setdefault(int(mk[3 * k]), [None, None])[0 if mk[3 * k + 1] == 5 else 1] = int(mk[3 * k + 2])
    if rc == 1:
        return score.value, None, 1
    recs = [(rec[i].m, rec[i].n) for i in range(1, n_rec.value)]
    if len(recs) < 2:
        return score.value, None, rc
    fin = hh.std_skl3(recs)
    return score.value, [1, len(fin)] + [x for mn in fin for x in mn], rc


def marks_changed(fx, marks: dict) -> dict:
    """{position: [phs5, phs3]} of the positions whose marks differ from the fixture's inputs -- the form of seed_marks_A*"""
    p5, p3 = fx["phs5"], fx["phs3"]
    out = {}
    for n, v in marks.items():
        a = int(p5[n]) if v[0] is None else v[0]
        b = int(p3[n]) if v[1] is None else v[1]
        if a != int(p5[n]) or b != int(p3[n]):
            out[n] = [a, b]
    return out


def wilip(model, p, sc, level: int, span, exg=(0, 0)):
    """the product's HSP search (spaln_amd/csrc/spdp_wilip.h, compiled into the checker) on one request, as
    Wilip::Wilip(seqs, pwd, level) answers it: the flat unit record SpdpHspSource::units would hand over.
    p: abi.Problem (nucleotide query) or abi.ProblemH (protein query); span = (a_left, a_right, b_left, b_right)"""
    from spaln_amd import abi as _abi
    protein = isinstance(p, _abi.ProblemH)
    out = (C.c_int32 * 65536)()
    al, ar, bl, br = (int(x) for x in span[:4])
    f = lib().walk_check_wilip
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                  C.c_void_p, C.c_int]
    sig = (p.sigS, p.sigE, p.sigT) if protein else (None, None, None)
    n = f(C.addressof(model), p.a, p.a_len, al, ar, int(exg[0]), int(exg[1]), p.b, p.b_len, bl, br, 3 if protein else 1, *sig,
          sc.intpen, sc.intpen_len, sc.gop, sc.gep, sc.lgop, sc.lgep, sc.codonk1, level, out, len(out))
    if n < 0:
        raise RuntimeError("walk_check_wilip: reply too long")
    return [int(out[i]) for i in range(n)]


def same_units(got, want) -> bool:
    """two flat unit records, but for the `nid` of each unit's closing record (the reference leaves that int as its fresh
    array held it)"""
    if len(got) != len(want) or got[:1] != want[:1]:
        return False
    at = 1
    for _ in range(got[0]):
        num = want[at]
        blk = 6 + 5 * (num + 1)
        g, w = list(got[at:at + blk]), list(want[at:at + blk])
        g[6 + 5 * num + 3] = w[6 + 5 * num + 3] = 0
        if g != w:
            return False
        at += blk
    return True
