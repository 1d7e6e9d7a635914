"""Host-side binding of libspdp_hip.so (include/spdp.h) -- the product path.

Mirrors the reference's Aln2 surface for the cDNA x genome path
(src/aln.h:348-357): `homscore_s` ~ HomScoreS_ng, `align_s` ~ alignS_ng, plus
the three SimdAln2s1 `_wip` engine methods.  Everything computes on the GPU;
if the HIP library is missing or no device is present this module raises --
there is deliberately no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPDP_LIB") or os.path.join(_HERE, "libspdp_hip.so")     # SPDP_LIB: A/B builds (tools/ab_variants.sh)

EXPORTS = [
    "spdp_create", "spdp_destroy", "spdp_last_error", "spdp_device_name", "spdp_stripe",
    "spdp_cells", "spdp_wip_scoreonly", "spdp_wip_forward", "spdp_wip_udh", "spdp_homscore_s",
    "spdp_align_s", "spdp_align_s_ori3", "spdp_free_alignments", "spdp_skl_rng_s", "spdp_skl_rng_h", "spdp_free_rescored", "spdp_scalar_forward", "spdp_scalar_scorealone", "spdp_scalar_udh", "spdp_batch_upload", "spdp_batch_free",
    "spdp_batch_cells", "spdp_batch_homscore", "spdp_batch_align", "spdp_batch_stats",
    "spdp_stripe31", "spdp_cells_h", "spdp_wip_forward_h", "spdp_wip_udh_h", "spdp_homscore_h", "spdp_align_h",
    "spdp_scalar_forward_h", "spdp_scalar_udh_h",
    "spdp_batch_upload_h", "spdp_batch_free_h", "spdp_batch_cells_h", "spdp_batch_align_h",
    "spdp_submit_align_s", "spdp_submit_homscore_s", "spdp_submit_align_h", "spdp_submit_homscore_h",
    "spdp_poll", "spdp_wait",
    "spdp_group_create", "spdp_group_destroy", "spdp_group_size", "spdp_group_last_error",
    "spdp_group_homscore_s", "spdp_group_align_s", "spdp_group_homscore_h", "spdp_group_align_h",
    "spdp_group_align_s_seeded", "spdp_group_align_h_seeded", "spdp_group_skl_rng_s", "spdp_group_skl_rng_h",
    "spdp_group_context", "spdp_group_blk_vote", "spdp_group_map_align_s",
    "spdp_align_s_seeded", "spdp_align_s_seeded_ori3", "spdp_seeded_stats",
    "spdp_blk_index_create", "spdp_blk_index_destroy", "spdp_blk_vote", "spdp_blk_vote_resident",
    "spdp_blk_search_opts_default", "spdp_blk_index_read", "spdp_blk_index_host_desc", "spdp_blk_index_host_free",
]


def load_library() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `make -C spaln_amd/csrc` (or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    lib.spdp_create.restype = C.c_void_p
    lib.spdp_create.argtypes = [C.c_int]
    lib.spdp_destroy.argtypes = [C.c_void_p]
    lib.spdp_last_error.restype = C.c_char_p
    lib.spdp_last_error.argtypes = [C.c_void_p]
    lib.spdp_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.spdp_cells.restype = C.c_int64
    lib.spdp_splice_signals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_splice_signals_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8
    lib.spdp_batch_upload.restype = C.c_void_p
    lib.spdp_batch_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.spdp_batch_free.argtypes = [C.c_void_p]
    lib.spdp_batch_cells.restype = C.c_int64
    lib.spdp_batch_cells.argtypes = [C.c_void_p]
    lib.spdp_batch_homscore.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_batch_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_batch_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    for f in ("spdp_wip_scoreonly", "spdp_homscore_s", "spdp_scalar_scorealone"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_wip_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_align_s.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_align_s_ori3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.spdp_scalar_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_wip_udh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_scalar_udh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_free_alignments.argtypes = [C.c_void_p, C.c_int]
    lib.spdp_skl_rng_s.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.spdp_skl_rng_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.spdp_free_rescored.argtypes = [C.c_void_p, C.c_int]
    lib.spdp_skl_edits_s.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_free_edits.argtypes = [C.c_void_p, C.c_int]
    lib.spdp_skl_edits_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_cells_h.restype = C.c_int64
    for f in ("spdp_wip_forward_h", "spdp_homscore_h", "spdp_align_h"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_wip_udh_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_scalar_udh_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_scalar_forward_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.spdp_batch_upload_h.restype = C.c_void_p
    lib.spdp_batch_upload_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.spdp_batch_free_h.argtypes = [C.c_void_p]
    lib.spdp_batch_cells_h.restype = C.c_int64
    lib.spdp_batch_cells_h.argtypes = [C.c_void_p]
    lib.spdp_batch_align_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for f in ("spdp_submit_align_s", "spdp_submit_homscore_s", "spdp_submit_align_h", "spdp_submit_homscore_h"):
        getattr(lib, f).restype = C.c_void_p
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_poll.argtypes = [C.c_void_p]
    lib.spdp_wait.argtypes = [C.c_void_p]
    lib.spdp_lsp_s.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_collector_create.restype = C.c_void_p
    lib.spdp_collector_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.spdp_collector_destroy.argtypes = [C.c_void_p]
    lib.spdp_collector_align_s.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_collector_last_error.restype = C.c_char_p
    lib.spdp_collector_last_error.argtypes = [C.c_void_p]
    lib.spdp_collector_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.spdp_group_create.restype = C.c_void_p
    lib.spdp_group_create.argtypes = [C.c_void_p, C.c_int]
    lib.spdp_group_destroy.argtypes = [C.c_void_p]
    lib.spdp_group_size.argtypes = [C.c_void_p]
    lib.spdp_group_last_error.restype = C.c_char_p
    lib.spdp_group_last_error.argtypes = [C.c_void_p]
    lib.spdp_group_last_shards.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.spdp_stripe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_cells.argtypes = [C.c_void_p, C.c_void_p]
    for f in ("spdp_group_homscore_s", "spdp_group_align_s", "spdp_group_homscore_h", "spdp_group_align_h"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    return lib


def exon_form(lib, eij, *, scr, gene_codes, protein, q_left, q_right, q_len, gmap, qmap, scale, aln_scale,
              q_many=1, q_sens=0, hsp_len=0, qname="qry", gname="win", header=False, gene_id=0):
    """Gsinfo::ExonForm from EISCR records (k x 21 ints, as skl_rng_s / _h return them): (exon records, gene record,
    the -O4 text).  Host only: works without a GPU."""
    eij = np.ascontiguousarray(eij, dtype=np.int32).reshape(-1, 21)
    codes = np.ascontiguousarray(gene_codes, dtype=np.uint8)
    a = abi.ExonFormIn()
    a.eij, a.n_eij, a.scr = eij.ctypes.data, eij.shape[0], int(scr)
    a.gene_codes, a.gene_is_tron, a.qry_is_protein = codes.ctypes.data, int(bool(protein)), int(bool(protein))
    a.q_left, a.q_right, a.q_len, a.q_many, a.q_sens = int(q_left), int(q_right), int(q_len), int(q_many), int(q_sens)
    a.gmap = abi.SiteMap(int(gmap[0]), int(gmap[1])); a.qmap = abi.SiteMap(int(qmap[0]), int(qmap[1]))
    a.scale, a.aln_scale, a.hsp_len = float(scale), float(aln_scale), int(hsp_len)
    a.gene_id = int(gene_id)
    lib.spdp_exon_form.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.spdp_exon_form_text.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    ex = (abi.ExonRecord * max(1, eij.shape[0]))()
    g = abi.GeneRecord()
    n = lib.spdp_exon_form(C.byref(a), ex, eij.shape[0], C.byref(g))
    if n < 0:
        raise RuntimeError("spdp_exon_form failed")
    buf = C.create_string_buffer(256 * (eij.shape[0] + 4))
    k = lib.spdp_exon_form_text(C.byref(a), qname.encode(), gname.encode(), int(header), buf, len(buf))
    if k < 0:
        raise RuntimeError("spdp_exon_form_text failed")
    return [ex[i] for i in range(n)], g, buf.raw[:k]


class Collector:
    """SpdpCollector: single-problem calls from many host threads, run as device batches.
    Owns the engine's context while it lives; align_s() may be called from any number of threads."""

    def __init__(self, eng: "Engine", sc, max_batch: int = 256, max_wait_us: int = 200, raw: bool = False):
        self.eng, self.lib, self._sc = eng, eng.lib, sc
        self.lib.spdp_collector_create.restype = C.c_void_p
        self.lib.spdp_collector_create_h.restype = C.c_void_p
        self.lib.spdp_collector_create.argtypes = self.lib.spdp_collector_create_h.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        self.lib.spdp_collector_align_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.protein = isinstance(sc, abi.ScoringH)
        make = self.lib.spdp_collector_create_h if self.protein else self.lib.spdp_collector_create
        self.h = make(eng.ctx, C.byref(sc), int(max_batch), int(max_wait_us), 1 if raw else 0)
        if not self.h:
            raise RuntimeError("spdp_collector_create failed")
        self.raw = raw

    def align_h(self, p: abi.ProblemH):
        out = abi.Alignment()
        rc = self.lib.spdp_collector_align_h(self.h, C.byref(p), C.byref(out))
        if rc < 0:
            raise RuntimeError("spdp_collector_align_h: " + self.lib.spdp_collector_last_error(self.h).decode())
        skl = np.array([(out.skl[i].m, out.skl[i].n) for i in range(max(out.n_skl, 0))], dtype=np.int32).reshape(-1, 2)
        res = int(out.score), skl, (out.n_skl if out.n_skl < 0 else 0)
        self.lib.spdp_free_alignments(C.byref(out), 1)
        return res

    def align_s(self, p: abi.Problem):
        out = abi.Alignment()
        rc = self.lib.spdp_collector_align_s(self.h, C.byref(p), C.byref(out))
        if rc < 0:
            raise RuntimeError("spdp_collector_align_s: " + self.lib.spdp_collector_last_error(self.h).decode())
        skl = np.array([(out.skl[i].m, out.skl[i].n) for i in range(out.n_skl)], dtype=np.int32).reshape(-1, 2)
        self.lib.spdp_free_alignments(C.byref(out), 1)
        return int(out.score), skl

    def stats(self) -> dict:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.spdp_collector_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(requests=a.value, batches=b.value, largest=c.value)

    def close(self):
        if self.h:
            self.lib.spdp_collector_destroy(self.h)
            self.h = None


def _seeded_call(lib, handle, check, name, sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial):
    """one seeded batch call on a context or a group handle (the HSP source replays recorded Wilip replies)"""
    n = len(ps)
    keep = []
    jx = (C.c_void_p * n)()
    nh = (C.c_int32 * n)()
    lv = (C.c_int32 * n)(*[int(x) for x in lowest_levels])
    for i, h in enumerate(hsps):
        if h is None or len(h) < 2:
            continue
        a = np.ascontiguousarray(h, dtype=np.int32)
        keep.append(a)
        jx[i] = a.ctypes.data
        nh[i] = a.shape[0] - 1
    missing = []

    def units(_user, query, level, span, flat, n_flat):
        key = (level, span[0], span[1], span[2], span[3])
        tab = wilip_tables[query] if wilip_tables else None
        if not tab or key not in tab:
            missing.append((query,) + key)
            return 1
        a = np.asarray(tab[key], dtype=np.int32)
        keep.append(a)                                   # (appends are atomic; the arrays live until the call returns)
        flat[0] = a.ctypes.data_as(C.POINTER(C.c_int32))
        n_flat[0] = a.size
        return 0

    src = abi.HspSource()
    src.user = None
    src.units = abi.HSP_UNITS_FN(units)
    src.release = abi.HSP_RELEASE_FN(lambda _u, _q, _f: None)
    arr = (abi.Alignment * n)()
    fn = getattr(lib, name)
    fn.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 5
    # wilip_tables = an abi.WilipModel: no HSP source at all -- the library's own Wilip (spdp_wilip.h) answers the levels
    own = isinstance(wilip_tables, abi.WilipModel)
    if own:
        sp.wilip = C.addressof(wilip_tables)
    rc = fn(handle, C.byref(sc), C.byref(sp), ps.array(), n, jx, nh, lv, None if own else C.byref(src), arr)
    if own:
        sp.wilip = None
    if missing:
        raise KeyError(f"no Wilip reply for {missing[:3]}")
    if not (allow_partial and rc == 1):
        check(rc, name)
    res = []
    for i in range(n):
        k = arr[i].n_skl
        skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)], dtype=np.int32).reshape(-1, 2)
        res.append((int(arr[i].score), skl))
    lib.spdp_free_alignments(arr, n)
    return res



class Group:
    """Several GPUs behind one handle (spdp_group_*): the batched calls shard the query list over the members."""

    def __init__(self, devices):
        self.lib = load_library()
        arr = (C.c_int * len(devices))(*devices)
        self.h = self.lib.spdp_group_create(arr, len(devices))
        if not self.h:
            raise RuntimeError("spdp_group_create failed: no usable HIP device (there is no CPU path)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.spdp_group_destroy(self.h)
            self.h = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.spdp_group_last_error(self.h).decode()}")

    def homscore_s(self, sc, ps) -> np.ndarray:
        out = np.zeros(len(ps), dtype=np.int32)
        self._check(self.lib.spdp_group_homscore_s(self.h, C.byref(sc), ps.array(), len(ps), out.ctypes.data),
                    "spdp_group_homscore_s")
        return out

    def align_s(self, sc, ps):
        n = len(ps)
        arr = (abi.Alignment * n)()
        self._check(self.lib.spdp_group_align_s(self.h, C.byref(sc), ps.array(), n, arr), "spdp_group_align_s")
        res = []
        for i in range(n):
            k = arr[i].n_skl
            skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)], dtype=np.int32).reshape(-1, 2)
            res.append((int(arr[i].score), skl))
        self.lib.spdp_free_alignments(arr, n)
        return res

    def align_s_seeded(self, sc, sp, ps, hsps, lowest_levels, wilip_tables=None, allow_partial=False):
        return _seeded_call(self.lib, self.h, self._check, "spdp_group_align_s_seeded", sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial)

    def align_h_seeded(self, sc, sp, ps, hsps, lowest_levels, wilip_tables=None, allow_partial=False):
        return _seeded_call(self.lib, self.h, self._check, "spdp_group_align_h_seeded", sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial)

    def shards(self, n) -> np.ndarray:
        member = np.zeros(n, dtype=np.int32)
        self.lib.spdp_group_last_shards(self.h, member.ctypes.data_as(C.c_void_p), n)
        return member


class Engine:
    """One SpdpContext on one GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = self.lib.spdp_create(device)
        if not self.ctx:
            raise RuntimeError("spdp_create failed: no usable HIP device (there is no CPU path)")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.spdp_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.spdp_last_error(self.ctx).decode()}")

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self.lib.spdp_device_name(self.ctx, buf, 256)
        return buf.value.decode()

    # ---- SimdAln2s1 `_wip` engines ------------------------------------------------
    def splice_signals(self, model: abi.SignalModel, b_codes, left: int = 0, right=None) -> dict:
        """Exinon::intron53_c / intron53_n of one window on the device: sig5, sig3, cano5, cano3, dinc by position"""
        b = np.ascontiguousarray(b_codes, dtype=np.uint8)
        n1 = b.size + 1
        out = dict(sig5=np.zeros(n1, np.int16), sig3=np.zeros(n1, np.int16), cano5=np.zeros(n1, np.uint8),
                   cano3=np.zeros(n1, np.uint8), dinc=np.zeros(n1, np.uint8))
        self._check(self.lib.spdp_splice_signals(self.ctx, C.addressof(model), b.ctypes.data, b.size, int(left),
                                                 int(b.size if right is None else right),
                                                 *(out[k].ctypes.data for k in ("sig5", "sig3", "cano5", "cano3", "dinc"))),
                    "spdp_splice_signals")
        return out

    def splice_signals_h(self, model: abi.SignalModelH, b_codes, left: int = 0, right=None) -> dict:
        """Exinon::intron53_c / intron53_p of one tron window on the device (b_codes: b_len + 1 codes)"""
        b = np.ascontiguousarray(b_codes, dtype=np.uint8)
        b_len = b.size - 1
        n = b_len + 3
        out = {k: np.zeros(n, np.int16) for k in ("sig5", "sig3", "sigS", "sigT", "sigE")}
        out.update(phs5=np.zeros(n, np.int8), phs3=np.zeros(n, np.int8), dinc=np.zeros(n, np.uint8))
        self._check(self.lib.spdp_splice_signals_h(self.ctx, C.addressof(model), b.ctypes.data, b_len, int(left),
                                                   int(b_len if right is None else right),
                                                   *(out[k].ctypes.data for k in ("sig5", "sig3", "sigS", "sigT", "sigE",
                                                                                  "phs5", "phs3", "dinc"))),
                    "spdp_splice_signals_h")
        return out

    def wip_scoreonly(self, sc: abi.Scoring, ps: abi.ProblemSet) -> np.ndarray:
        out = np.zeros(len(ps), dtype=np.int32)
        self._check(self.lib.spdp_wip_scoreonly(self.ctx, C.byref(sc), ps.array(), len(ps),
                                                out.ctypes.data), "spdp_wip_scoreonly")
        return out

    def homscore_s(self, sc, ps, allow_partial=False) -> np.ndarray:
        out = np.zeros(len(ps), dtype=np.int32)
        rc = self.lib.spdp_homscore_s(self.ctx, C.byref(sc), ps.array(), len(ps), out.ctypes.data)
        if not (allow_partial and rc == 1):
            self._check(rc, "spdp_homscore_s")
        return out

    def align_s_ori3(self, sc, ps_fwd, ps_rev):
        """alignS_ng(ori = 3), -Q0: ([(score, skl)], orientation 0 / 1 per query)"""
        n = len(ps_fwd)
        assert len(ps_rev) == n
        arr = (abi.Alignment * n)()
        orient = np.zeros(n, dtype=np.int32)
        self._check(self.lib.spdp_align_s_ori3(self.ctx, C.byref(sc), ps_fwd.array(), ps_rev.array(), n, arr,
                                               orient.ctypes.data), "spdp_align_s_ori3")
        res = []
        for i in range(n):
            k = arr[i].n_skl
            skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)], dtype=np.int32).reshape(-1, 2)
            res.append((int(arr[i].score), skl))
        self.lib.spdp_free_alignments(arr, n)
        return res, orient

    def _alignments(self, fn, sc, ps, what, allow_partial=False, with_flags=False):
        n = len(ps)
        arr = (abi.Alignment * n)()
        rc = fn(self.ctx, C.byref(sc), ps.array(), n, arr)
        if not (allow_partial and rc == 1):
            self._check(rc, what)
        res = []
        for i in range(n):
            k = arr[i].n_skl
            skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)],
                           dtype=np.int32).reshape(-1, 2)
            res.append((int(arr[i].score), skl, int(arr[i].flags)) if with_flags else (int(arr[i].score), skl))
        self.lib.spdp_free_alignments(arr, n)
        return res

    def align_s_seeded(self, sc, sp: abi.SeedParams, ps, hsps, lowest_levels, wilip_tables=None, allow_partial=False):
        """alignS_ng (ori = 1) with seeding on (-Q5 .. -Q7): hsps[i] = (n_i + 1, 5) int32 array (b->jxt with its free slot) or
        None, lowest_levels[i] = b->wllvl; wilip_tables[i] = {(level, a_left, a_right, b_left, b_right): flat unit record}
        serves the Wilip calls of the recursion levels (a replay of what a reference run recorded, in the tests; the
        reference's own wln.cc in an integration).  Returns [(score, skl)] like align_s."""
        return self._align_seeded("spdp_align_s_seeded", sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial)

    def align_h_seeded(self, sc, sp: abi.SeedParams, ps, hsps, lowest_levels, wilip_tables=None, allow_partial=False):
        """alignH_ng with seeding on (the protein walk, spdp_align_h_seeded): arguments and result as align_s_seeded; with
        allow_partial a query whose walk is not served comes back as (NEVSEL, empty)"""
        return self._align_seeded("spdp_align_h_seeded", sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial)

    def _align_seeded(self, name, sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial):
        return _seeded_call(self.lib, self.ctx, self._check, name, sc, sp, ps, hsps, lowest_levels, wilip_tables, allow_partial)

    def align_s_seeded_ori3(self, sc, sp, ps_fwd, ps_rev, hsps, lowest_levels, wilip_tables):
        """alignS_ng(ori = 3) with seeding on: wilip_tables has 2 n entries (the reverse walk of query i is query n + i).
        Returns ([(score, skl)], orient)."""
        n = len(ps_fwd)
        keep = []
        jx = (C.c_void_p * n)()
        nh = (C.c_int32 * n)()
        lv = (C.c_int32 * n)(*[int(x) for x in lowest_levels])
        for i, h in enumerate(hsps):
            if h is None or len(h) < 2:
                continue
            a = np.ascontiguousarray(h, dtype=np.int32)
            keep.append(a)
            jx[i] = a.ctypes.data
            nh[i] = a.shape[0] - 1
        missing = []

        def units(_user, query, level, span, flat, n_flat):
            key = (level, span[0], span[1], span[2], span[3])
            tab = wilip_tables[query]
            if not tab or key not in tab:
                missing.append((query,) + key)
                return 1
            a = np.asarray(tab[key], dtype=np.int32)
            keep.append(a)
            flat[0] = a.ctypes.data_as(C.POINTER(C.c_int32))
            n_flat[0] = a.size
            return 0

        src = abi.HspSource()
        src.user = None
        src.units = abi.HSP_UNITS_FN(units)
        src.release = abi.HSP_RELEASE_FN(lambda _u, _q, _f: None)
        arr = (abi.Alignment * n)()
        orient = (C.c_int32 * n)()
        self.lib.spdp_align_s_seeded_ori3.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6
        rc = self.lib.spdp_align_s_seeded_ori3(self.ctx, C.byref(sc), C.byref(sp), ps_fwd.array(), ps_rev.array(), n, jx, nh, lv,
                                               C.byref(src), arr, orient)
        if missing:
            raise KeyError(f"no Wilip reply for {missing[:3]}")
        self._check(rc, "spdp_align_s_seeded_ori3")
        res = []
        for i in range(n):
            k = arr[i].n_skl
            skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)], dtype=np.int32).reshape(-1, 2)
            res.append((int(arr[i].score), skl))
        self.lib.spdp_free_alignments(arr, n)
        return res, [int(x) for x in orient]

    def seeded_phase_marks(self, q: int) -> dict:
        """spdp_seeded_phase_marks: {position: [phs5 or None, phs3 or None]} the walk of query q (last spdp_align_h_seeded call) wrote"""
        class Mark(C.Structure):
            _fields_ = [("n", C.c_int32), ("side", C.c_int8), ("value", C.c_int8), ("reserved", C.c_int16)]
        mk = C.POINTER(Mark)()
        self.lib.spdp_seeded_phase_marks.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.lib.spdp_seeded_phase_marks.restype = C.c_int
        n = self.lib.spdp_seeded_phase_marks(self.ctx, q, C.byref(mk))
        out = {}
        for k in range(n):
            out.setdefault(int(mk[k].n), [None, None])[0 if mk[k].side == 5 else 1] = int(mk[k].value)
        return out

    def seeded_stats(self) -> dict:
        v = (C.c_int64 * 6)()
        self.lib.spdp_seeded_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.lib.spdp_seeded_stats(self.ctx, v, 6)
        return dict(zip(("batches", "lsp", "trcbk", "trcbk_cut", "wilip", "walks"), (int(x) for x in v)))

    def lsp_s(self, sc, ps):
        """lspS_ng level: (score, raw Mfile records) per problem"""
        n = len(ps)
        arr = (abi.Alignment * n)()
        self._check(self.lib.spdp_lsp_s(self.ctx, C.byref(sc), ps.array(), n, arr), "spdp_lsp_s")
        res = [(int(a.score), np.array([(a.skl[i].m, a.skl[i].n) for i in range(a.n_skl)], dtype=np.int32).reshape(-1, 2))
               for a in arr]
        self.lib.spdp_free_alignments(arr, n)
        return res

    def wip_forward(self, sc, ps):
        return self._alignments(self.lib.spdp_wip_forward, sc, ps, "spdp_wip_forward")

    def scalar_forward(self, sc, ps):
        """Aln2s1::forwardS_ng via trcbkalignS_ng (scalar exact engine, -A0)."""
        return self._alignments(self.lib.spdp_scalar_forward, sc, ps, "spdp_scalar_forward")

    def scalar_scorealone(self, sc, ps) -> np.ndarray:
        out = np.zeros(len(ps), dtype=np.int32)
        self._check(self.lib.spdp_scalar_scorealone(self.ctx, C.byref(sc), ps.array(), len(ps),
                                                    out.ctypes.data), "spdp_scalar_scorealone")
        return out

    def scalar_udh(self, sc, ps, n_im: int, imd_intvl: int):
        """Aln2s1::hirschbergS_ng: (scores, cpos rows, written-back ranges, flags)"""
        n = len(ps)
        scores = np.zeros(n, dtype=np.int32)
        cpos = np.zeros((n, n_im + 1, 10), dtype=np.int32)
        ranges = np.zeros((n, 4), dtype=np.int32)
        flags = np.zeros(n, dtype=np.int32)
        self._check(self.lib.spdp_scalar_udh(self.ctx, C.byref(sc), ps.array(), n, n_im, imd_intvl,
                                             scores.ctypes.data, cpos.ctypes.data, ranges.ctypes.data,
                                             flags.ctypes.data), "spdp_scalar_udh")
        return scores, cpos, ranges, flags

    def submit_align_s(self, sc, ps):
        """spdp_submit_align_s: returns a callable that waits and yields [(score, skl)]"""
        n = len(ps)
        arr = (abi.Alignment * n)()
        parr = ps.array()
        t = self.lib.spdp_submit_align_s(self.ctx, C.byref(sc), parr, n, arr)
        if not t:
            raise RuntimeError("spdp_submit_align_s failed")

        def wait():
            rc = self.lib.spdp_wait(t)
            wait.done = True                     # the ticket is gone
            self._check(rc, "spdp_submit_align_s")
            res = []
            for i in range(n):
                k = arr[i].n_skl
                skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)], dtype=np.int32).reshape(-1, 2)
                res.append((int(arr[i].score), skl))
            self.lib.spdp_free_alignments(arr, n)
            return res
        wait.poll = lambda: True if wait.done else bool(self.lib.spdp_poll(t))
        wait.done = False
        wait.keep = (parr, ps, sc)               # inputs stay alive until waited
        return wait

    def align_s(self, sc, ps, allow_partial=False, with_flags=False):
        """alignS_ng (ori = 1, -Q0).  allow_partial: accept return value 1 (some problem needed the
        scalar engine for a < 8-row slab and the exact-model inputs were not supplied; those come
        back without an alignment) instead of raising.  with_flags: (score, skl, SpdpAlignment.flags)."""
        return self._alignments(self.lib.spdp_align_s, sc, ps, "spdp_align_s", allow_partial, with_flags)

    def skl_rng_s(self, sc, ps, alignments, *, codonk1, minl, jneibr, lsg=1):
        """skl_rngS_ng over finished alignments (rows of (m, n) with the header row first, as align_s
        returns them).  Returns [(score, [mch, mmc, gap, unp, val], exon records (k x 21))]."""
        n = len(ps)
        keep = []
        arr = (abi.Alignment * n)()
        for i, skl in enumerate(alignments):
            skl = np.ascontiguousarray(skl, dtype=np.int32).reshape(-1, 2)
            keep.append(skl)
            arr[i].score, arr[i].n_skl = 0, skl.shape[0]
            arr[i].skl = C.cast(skl.ctypes.data, C.POINTER(abi.Skl))
        rp = abi.RescoreParams(int(codonk1), int(minl), int(jneibr), int(lsg))
        out = (abi.Rescored * n)()
        self._check(self.lib.spdp_skl_rng_s(self.ctx, C.byref(sc), C.byref(rp), ps.array(), n, arr, out), "spdp_skl_rng_s")
        res = []
        for i in range(n):
            k = out[i].n_exons
            ex = np.ctypeslib.as_array(C.cast(out[i].exons, C.POINTER(C.c_int32)), shape=(k, 21)).copy() if k else \
                np.zeros((0, 21), dtype=np.int32)
            res.append((int(out[i].score), [out[i].mch, out[i].mmc, out[i].gap, out[i].unp, out[i].val], ex))
        self.lib.spdp_free_rescored(out, n)
        return res

    def skl_edits_s(self, sc, ps, alignments, fmt, *, codonk1, minl, jneibr, lsg=1):
        """the edit records skl_rngS_ng collects for the Cigar / Vulgar / SAM writers (fmt = abi.FMT_*): per query
        (records (k x 3: op, alen, blen), [sam_flag, sam_pos, sam_mapq, sam_left, sam_right])"""
        n = len(ps)
        keep = []
        arr = (abi.Alignment * n)()
        for i, skl in enumerate(alignments):
            skl = np.ascontiguousarray(skl, dtype=np.int32).reshape(-1, 2)
            keep.append(skl)
            arr[i].score, arr[i].n_skl = 0, skl.shape[0]
            arr[i].skl = C.cast(skl.ctypes.data, C.POINTER(abi.Skl))
        rp = abi.RescoreParams(int(codonk1), int(minl), int(jneibr), int(lsg))
        out = (abi.Edits * n)()
        self._check(self.lib.spdp_skl_edits_s(self.ctx, C.byref(sc), C.byref(rp), ps.array(), n, arr, int(fmt), out),
                    "spdp_skl_edits_s")
        res = []
        for i in range(n):
            k = out[i].n
            rec = np.ctypeslib.as_array(C.cast(out[i].rec, C.POINTER(C.c_int32)), shape=(k, 3)).copy() if k else \
                np.zeros((0, 3), dtype=np.int32)
            res.append((rec, [out[i].sam_flag, out[i].sam_pos, out[i].sam_mapq, out[i].sam_left, out[i].sam_right]))
        self.lib.spdp_free_edits(out, n)
        return res

    def skl_edits_h(self, sc, ps, alignments, fmt, *, minl, jneibr, lcl=15, sup_tcodon=0):
        """the edit records skl_rngH_ng collects for the Cigar / Vulgar writers: per query records (k x 3: op, alen, blen)"""
        n = len(ps)
        keep = []
        arr = (abi.Alignment * n)()
        for i, skl in enumerate(alignments):
            skl = np.ascontiguousarray(skl, dtype=np.int32).reshape(-1, 2)
            keep.append(skl)
            arr[i].score, arr[i].n_skl = 0, skl.shape[0]
            arr[i].skl = C.cast(skl.ctypes.data, C.POINTER(abi.Skl))
        rp = abi.RescoreParamsH(int(minl), int(jneibr), int(lcl), int(sup_tcodon))
        out = (abi.Edits * n)()
        self._check(self.lib.spdp_skl_edits_h(self.ctx, C.byref(sc), C.byref(rp), ps.array(), n, arr, int(fmt), out),
                    "spdp_skl_edits_h")
        res = []
        for i in range(n):
            k = out[i].n
            res.append(np.ctypeslib.as_array(C.cast(out[i].rec, C.POINTER(C.c_int32)), shape=(k, 3)).copy() if k else
                       np.zeros((0, 3), dtype=np.int32))
        self.lib.spdp_free_edits(out, n)
        return res

    def skl_rng_h(self, sc, ps, alignments, *, minl, jneibr, lcl=15, sup_tcodon=0):
        """skl_rngH_ng over finished protein alignments; returns as skl_rng_s does"""
        n = len(ps)
        keep = []
        arr = (abi.Alignment * n)()
        for i, skl in enumerate(alignments):
            skl = np.ascontiguousarray(skl, dtype=np.int32).reshape(-1, 2)
            keep.append(skl)
            arr[i].score, arr[i].n_skl = 0, skl.shape[0]
            arr[i].skl = C.cast(skl.ctypes.data, C.POINTER(abi.Skl))
        rp = abi.RescoreParamsH(int(minl), int(jneibr), int(lcl), int(sup_tcodon))
        out = (abi.Rescored * n)()
        self._check(self.lib.spdp_skl_rng_h(self.ctx, C.byref(sc), C.byref(rp), ps.array(), n, arr, out), "spdp_skl_rng_h")
        res = []
        for i in range(n):
            k = out[i].n_exons
            ex = np.ctypeslib.as_array(C.cast(out[i].exons, C.POINTER(C.c_int32)), shape=(k, 21)).copy() if k else \
                np.zeros((0, 21), dtype=np.int32)
            res.append((int(out[i].score), [out[i].mch, out[i].mmc, out[i].gap, out[i].unp, out[i].val], ex))
        self.lib.spdp_free_rescored(out, n)
        return res

    def wip_udh(self, sc, ps, n_im: int):
        n = len(ps)
        scores = np.zeros(n, dtype=np.int32)
        cpos = np.zeros((n, n_im + 1, 10), dtype=np.int32)
        ranges = np.zeros((n, 4), dtype=np.int32)
        self._check(self.lib.spdp_wip_udh(self.ctx, C.byref(sc), ps.array(), n, n_im,
                                          scores.ctypes.data, cpos.ctypes.data, ranges.ctypes.data),
                    "spdp_wip_udh")
        return scores, cpos, ranges

    # ---- aa x genome (SimdAln2h1 `_wip`) ------------------------------------------------
    def _alignments_h(self, fn, sc, ps, what):
        """[(score, records, flag)]: flag 0 ok, -1 the reference's fatal "Unexpected dir", -2 its
        traceback starts outside its bitmap, 1 the problem needs an engine that is not built."""
        n = len(ps)
        arr = (abi.Alignment * n)()
        rc = fn(self.ctx, C.byref(sc), ps.array(), n, arr)
        if rc not in (0, 1):
            self._check(rc, what)
        res = []
        for i in range(n):
            k = arr[i].n_skl
            skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(max(k, 0))],
                           dtype=np.int32).reshape(-1, 2)
            flag = k if k < 0 else (1 if (rc == 1 and arr[i].score == abi.NEVSEL and k == 0) else 0)
            res.append((int(arr[i].score), skl, flag))
        self.lib.spdp_free_alignments(arr, n)
        return res

    def wip_forward_h(self, sc: abi.ScoringH, ps: abi.ProblemSetH):
        """SimdAln2h1::forwardH1_wip(mfd) with the stripe31() band: raw records end -> start."""
        return self._alignments_h(self.lib.spdp_wip_forward_h, sc, ps, "spdp_wip_forward_h")

    def scalar_forward_h(self, sc: abi.ScoringH, ps: abi.ProblemSetH, traceback: bool = True):
        """Aln2h1::forwardH_ng on the stripe31() band: [(score, records end -> start)]"""
        n = len(ps)
        arr = (abi.Alignment * n)()
        self._check(self.lib.spdp_scalar_forward_h(self.ctx, C.byref(sc), ps.array(), n, 1 if traceback else 0, arr),
                    "spdp_scalar_forward_h")
        res = []
        for i in range(n):
            k = arr[i].n_skl
            skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(max(k, 0))],
                           dtype=np.int32).reshape(-1, 2)
            res.append((int(arr[i].score), skl))
        self.lib.spdp_free_alignments(arr, n)
        return res

    def scalar_udh_h(self, sc, ps, n_im: int, imd_intvl: int):
        """Aln2h1::hirschbergH_ng: (scores, cpos rows, written-back ranges, flags)"""
        n = len(ps)
        scores = np.zeros(n, dtype=np.int32)
        cpos = np.zeros((n, n_im + 1, 10), dtype=np.int32)
        ranges = np.zeros((n, 4), dtype=np.int32)
        flags = np.zeros(n, dtype=np.int32)
        self._check(self.lib.spdp_scalar_udh_h(self.ctx, C.byref(sc), ps.array(), n, n_im, imd_intvl,
                                               scores.ctypes.data, cpos.ctypes.data, ranges.ctypes.data,
                                               flags.ctypes.data), "spdp_scalar_udh_h")
        return scores, cpos, ranges, flags

    def align_h(self, sc, ps):
        """alignH_ng (-Q0): [flags, n, corners...] as rows of (m, n) after the header row."""
        return self._alignments_h(self.lib.spdp_align_h, sc, ps, "spdp_align_h")

    def lsp_h(self, sc, ps):
        """lspH_ng level: (score, raw Mfile records, flag) per problem"""
        self.lib.spdp_lsp_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        return self._alignments_h(self.lib.spdp_lsp_h, sc, ps, "spdp_lsp_h")

    def wip_udh_h(self, sc, ps, n_im: int):
        """SimdAln2h1::hirschbergH1_wip: (scores, cpos rows, written-back ranges)"""
        n = len(ps)
        scores = np.zeros(n, dtype=np.int32)
        cpos = np.zeros((n, n_im + 1, 10), dtype=np.int32)
        ranges = np.zeros((n, 4), dtype=np.int32)
        self._check(self.lib.spdp_wip_udh_h(self.ctx, C.byref(sc), ps.array(), n, n_im,
                                            scores.ctypes.data, cpos.ctypes.data, ranges.ctypes.data),
                    "spdp_wip_udh_h")
        return scores, cpos, ranges

    def homscore_h(self, sc, ps) -> np.ndarray:
        out = np.zeros(len(ps), dtype=np.int32)
        rc = self.lib.spdp_homscore_h(self.ctx, C.byref(sc), ps.array(), len(ps), out.ctypes.data)
        if rc not in (0, 1):
            self._check(rc, "spdp_homscore_h")
        return out

    def upload_h(self, sc, ps):
        h = self.lib.spdp_batch_upload_h(self.ctx, C.byref(sc), ps.array(), len(ps))
        if not h:
            raise RuntimeError("spdp_batch_upload_h: " + self.lib.spdp_last_error(self.ctx).decode())
        return BatchH(self, h, len(ps))

    # ---- resident batches ------------------------------------------------------------
    def upload(self, sc, ps):
        h = self.lib.spdp_batch_upload(self.ctx, C.byref(sc), ps.array(), len(ps))
        if not h:
            raise RuntimeError("spdp_batch_upload: " + self.lib.spdp_last_error(self.ctx).decode())
        return Batch(self, h, len(ps))


class Batch:
    def __init__(self, eng: Engine, handle, n: int):
        self.eng, self.h, self.n = eng, handle, n

    def cells(self) -> int:
        return int(self.eng.lib.spdp_batch_cells(self.h))

    def homscore(self, want_scores: bool = True):
        ms = C.c_float()
        out = np.zeros(self.n, dtype=np.int32) if want_scores else None
        rc = self.eng.lib.spdp_batch_homscore(self.h, out.ctypes.data if want_scores else None, C.byref(ms))
        self.eng._check(rc, "spdp_batch_homscore")
        return out, ms.value

    def align(self, want: bool = True, convert: bool = True):
        """alignS_ng (ori = 1, -Q0) over the resident batch.  Returns (alignments, kernel_ms, kernel_cells).
        want: the library produces the final SKL arrays in host memory (stdskl / trimskl / allocation);
        convert=False leaves them as the C arrays they are (freed again) instead of building numpy rows --
        what bench.py times: the whole product path, no Python per-record work."""
        ms = C.c_float()
        cells = C.c_int64()
        arr = (abi.Alignment * self.n)() if want else None
        rc = self.eng.lib.spdp_batch_align(self.h, arr, C.byref(ms), C.byref(cells))
        self.eng._check(rc, "spdp_batch_align")
        res = None
        if want and not convert:
            res = sum(arr[i].n_skl for i in range(0, self.n, max(1, self.n // 64)))      # touch a few
            self.eng.lib.spdp_free_alignments(arr, self.n)
        elif want:
            res = []
            for i in range(self.n):
                k = arr[i].n_skl
                skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(k)],
                               dtype=np.int32).reshape(-1, 2)
                res.append((int(arr[i].score), skl))
            self.eng.lib.spdp_free_alignments(arr, self.n)
        return res, ms.value, cells.value

    def stats(self) -> dict:
        v = (C.c_double * 8)()
        self.eng.lib.spdp_batch_stats(self.h, v, 8)
        keys = ["udh_ms", "udh_cells", "udh_problems", "fwd_ms", "fwd_cells", "fwd_problems",
                "udh_rounds", "tb_bytes"]
        return dict(zip(keys, (float(x) for x in v)))

    def free(self):
        if self.h:
            self.eng.lib.spdp_batch_free(self.h)
            self.h = None


class BatchH:
    """resident aa x genome batch (one live batch of this kind per Engine)"""

    def __init__(self, eng: Engine, handle, n: int):
        self.eng, self.h, self.n = eng, handle, n

    def cells(self) -> int:
        return int(self.eng.lib.spdp_batch_cells_h(self.h))

    def align(self, want: bool = True, convert: bool = True):
        """alignH_ng over the resident batch.  Returns (alignments, sweep_kernel_ms, cells); want / convert as in
        Batch.align."""
        ms = C.c_float()
        cells = C.c_int64()
        arr = (abi.Alignment * self.n)() if want else None
        rc = self.eng.lib.spdp_batch_align_h(self.h, arr, C.byref(ms), C.byref(cells))
        if rc not in (0, 1):
            self.eng._check(rc, "spdp_batch_align_h")
        res = None
        if want and not convert:
            res = sum(max(arr[i].n_skl, 0) for i in range(0, self.n, max(1, self.n // 64)))
            self.eng.lib.spdp_free_alignments(arr, self.n)
        elif want:
            res = []
            for i in range(self.n):
                k = arr[i].n_skl
                skl = np.array([(arr[i].skl[j].m, arr[i].skl[j].n) for j in range(max(k, 0))],
                               dtype=np.int32).reshape(-1, 2)
                res.append((int(arr[i].score), skl, min(k, 0)))
            self.eng.lib.spdp_free_alignments(arr, self.n)
        return res, ms.value, cells.value

    def free(self):
        if self.h:
            self.eng.lib.spdp_batch_free_h(self.h)
            self.h = None
