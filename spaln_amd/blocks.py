"""Binding of the block search's vote (include/spdp.h, "block search") -- product path, device only.

`desc_from_arrays` fills SpdpBlkIndexDesc from the arrays a reference-side dump of an index holds (the layout the tests'
fixtures and the bench use: blk_prm, blk_nblk, ...); an integration fills the struct from its SrchBlk object instead
(INTEGRATION.md)."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np


class BlkIndexDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nalpha", "tabsize", "nshift", "nbitpat", "convts", "n_chr", "maxblk", "kk", "drna", "maxmmc", "nseg",
        "minsigpr", "ncand", "nascr", "maxblock", "extblock", "extblockl", "shortquery", "hh_size", "hh_step", "hb_size", "hb_step",
        "ha_size", "ha_step", "gdb", "blklen")] + [
        ("rbscoef", C.c_float), ("rbscons", C.c_float),
        ("bclw", C.c_double), ("bcup", C.c_double), ("bcce", C.c_double), ("cfact", C.c_double),
        ("convtab", C.c_void_p), ("nblk", C.c_void_p), ("wscr", C.c_void_p), ("blkp", C.c_void_p),
        ("blkb", C.c_void_p), ("n_words", C.c_int64), ("rscrtab", C.c_void_p), ("chr", C.c_void_p),
        ("bitpat", C.c_void_p), ("n_bitpat", C.c_int32)]


# positions in the parameter record of a reference-side index dump (oracle/ref_build/blk_tap.cc, dump_index)
_PRM = dict(nalpha=0, tabsize=3, nshift=5, nbitpat=8, convts=10, n_chr=12, maxblk=14, kk=15, drna=16, maxmmc=17, nseg=19,
            minsigpr=22, ncand=23, nascr=24, maxblock=25, extblock=26, extblockl=27, shortquery=28, hh_size=29, hh_step=30, hb_size=31,
            hb_step=32, ha_size=33, ha_step=34, gdb=38, blklen=6)
REACHED, CUT, TABLE = 1, 2, 4


def desc_from_arrays(fx: dict):
    """(BlkIndexDesc, keep-alive list)"""
    prm = np.asarray(fx["blk_prm"], dtype=np.int32)
    d = BlkIndexDesc()
    for name, pos in _PRM.items():
        setattr(d, name, int(prm[pos]))
    d.rbscoef = struct.unpack("<f", struct.pack("<i", int(prm[36])))[0]
    d.rbscons = struct.unpack("<f", struct.pack("<i", int(prm[37])))[0]
    d.bclw, d.bcup, d.bcce = (float(x) for x in np.frombuffer(np.asarray(fx["blk_pb2c"], dtype=np.uint8).tobytes(), dtype=np.float64))
    d.cfact = float(np.frombuffer(np.asarray(fx["blk_cfact"], dtype=np.uint8).tobytes(), dtype=np.float64)[0])
    keep = []
    for field, key, dt in (("convtab", "blk_convtab", np.uint8), ("nblk", "blk_nblk", np.uint16), ("wscr", "blk_wscr", np.int16),
                           ("blkp", "blk_blkp", np.int32), ("blkb", "blk_blkb", np.uint32), ("rscrtab", "blk_rscrtab", np.int32),
                           ("chr", "blk_chr", np.int32), ("bitpat", "blk_bitpat", np.int32)):
        a = np.asarray(fx[key])
        a = np.ascontiguousarray(a.view(dt) if a.dtype.itemsize == np.dtype(dt).itemsize else a.astype(dt))
        keep.append(a)
        setattr(d, field, a.ctypes.data)
    d.n_words = int(keep[4].size)
    d.n_bitpat = int(keep[7].size)
    return d, keep


class BlockIndex:
    """an index resident on the engine's device"""

    def __init__(self, eng, fx: dict):
        self.eng, self.lib = eng, eng.lib
        self.lib.spdp_blk_index_create.restype = C.c_void_p
        self.lib.spdp_blk_index_create.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.spdp_blk_index_destroy.argtypes = [C.c_void_p]
        self.lib.spdp_blk_vote.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        self.lib.spdp_blk_vote_resident.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        self.desc, self._keep = desc_from_arrays(fx)
        self.h = self.lib.spdp_blk_index_create(eng.ctx, C.byref(self.desc))
        if not self.h:
            raise RuntimeError("spdp_blk_index_create: " + self.lib.spdp_last_error(eng.ctx).decode())

    def free(self):
        if getattr(self, "h", None):
            self.lib.spdp_blk_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def vote(self, queries, ranges=None, stop_at=None, out_cap: int = 4096):
        """queries: list of uint8 code arrays; ranges: [(left, right)] (default whole query); stop_at: per query or None.
        Returns (records as an (n, out_cap) int32 array, kernel ms)."""
        n = len(queries)
        offs = np.zeros(n + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(q) for q in queries])
        codes = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in queries]) if n else np.zeros(0, np.uint8))
        left = np.array([0 if ranges is None else ranges[i][0] for i in range(n)], dtype=np.int32)
        right = np.array([len(queries[i]) if ranges is None else ranges[i][1] for i in range(n)], dtype=np.int32)
        st = None if stop_at is None else np.ascontiguousarray(stop_at, dtype=np.int32)
        out = np.zeros((n, out_cap), dtype=np.int32)
        ms = C.c_float()
        rc = self.lib.spdp_blk_vote(self.eng.ctx, self.h, codes.ctypes.data, offs.ctypes.data, left.ctypes.data, right.ctypes.data,
                                    None if st is None else st.ctypes.data, n, out.ctypes.data, out_cap, C.byref(ms))
        self.eng._check(rc, "spdp_blk_vote")
        return out, ms.value

    def vote_resident(self, d_codes, d_offs, d_left, d_right, d_stop_at, n, d_out, out_cap):
        """device pointers in (ints), records written to d_out; returns kernel ms"""
        ms = C.c_float()
        rc = self.lib.spdp_blk_vote_resident(self.eng.ctx, self.h, d_codes, d_offs, d_left, d_right, d_stop_at, n, d_out, out_cap,
                                             C.byref(ms))
        self.eng._check(rc, "spdp_blk_vote_resident")
        return ms.value


def split_record(rec: np.ndarray):
    """one query's record -> dict (see include/spdp.h, spdp_blk_vote)"""
    n, calls, flags = int(rec[0]), int(rec[1]), int(rec[2])
    if not flags & REACHED or flags & CUT:
        return dict(reached=bool(flags & REACHED), calls=calls, flags=flags)
    r = rec[:n]
    j = 3
    head = r[j:j + 20].copy(); j += 20
    qb = []
    for _ in range(4):
        k = int(r[j]); qb.append(r[j + 1:j + 1 + 2 * k].reshape(k, 2).copy()); j += 1 + 2 * k
    npairs = int(r[j]); pairs = r[j + 1:j + 1 + 9 * npairs].reshape(npairs, 9).copy(); j += 1 + 9 * npairs
    nruns = int(r[j]); runs = r[j + 1:j + 1 + 2 * nruns].reshape(nruns, 2).copy(); j += 1 + 2 * nruns
    assert j == n, (j, n)
    by_d = [[] for _ in range(4)]
    for code, scr in runs:
        by_d[int(code) >> 28].append((int(code) & 0xfffffff, int(scr)))
    return dict(reached=True, calls=calls, flags=flags, head=head, qb=qb, pairs=pairs, runs=[sorted(x) for x in by_d])


class SearchOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_out", "max_mmc", "min_sigpr", "nascr", "ext_block", "max_intron_len", "local",
                                          "genomic_db")] + [("rbs_fact", C.c_float), ("rbs_base", C.c_float), ("cfact", C.c_double)]


def read_index_file(lib, path: str, **opts):
    """the reference's <db>.bkn read by the library (host only): dict with the same keys a reference-side dump has
    (blk_prm positions filled where the library knows them), for comparison and for BlockIndex"""
    lib.spdp_blk_index_read.restype = C.c_void_p
    lib.spdp_blk_index_read.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.spdp_blk_index_host_desc.restype = C.POINTER(BlkIndexDesc)
    lib.spdp_blk_index_host_desc.argtypes = [C.c_void_p]
    lib.spdp_blk_index_host_free.argtypes = [C.c_void_p]
    o = SearchOpts()
    lib.spdp_blk_search_opts_default(C.byref(o))
    for k, v in opts.items():
        setattr(o, k, v)
    err = C.create_string_buffer(256)
    h = lib.spdp_blk_index_read(path.encode(), C.byref(o), err, 256)
    if not h:
        raise RuntimeError(err.value.decode())
    out = _host_index_to_dict(lib, h)
    lib.spdp_blk_index_host_free(h)
    return out


def _host_index_to_dict(lib, h) -> dict:
    """a SpdpBlkIndexHost -> the arrays in the layout of a reference-side dump (BlockIndex, the tests' oracle)"""
    lib.spdp_blk_index_host_desc.restype = C.POINTER(BlkIndexDesc)
    lib.spdp_blk_index_host_desc.argtypes = [C.c_void_p]
    d = lib.spdp_blk_index_host_desc(h).contents

    def arr(ptr, n, dt):
        if not n:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()
    out = {name: int(getattr(d, name)) for name, _ in BlkIndexDesc._fields_[:26]}
    out.update(rbscoef=float(d.rbscoef), rbscons=float(d.rbscons), bclw=d.bclw, bcup=d.bcup, bcce=d.bcce, cfact=d.cfact,
               blk_convtab=arr(d.convtab, d.convts, np.uint8), blk_nblk=arr(d.nblk, d.tabsize, np.uint16),
               blk_wscr=arr(d.wscr, d.tabsize, np.int16), blk_blkp=arr(d.blkp, d.tabsize, np.int32),
               blk_blkb=arr(d.blkb, d.n_words, np.uint32), blk_rscrtab=arr(d.rscrtab, 128, np.int32),
               blk_chr=arr(d.chr, 2 * (d.n_chr + 1), np.int32), blk_bitpat=arr(d.bitpat, d.n_bitpat, np.int32))
    prm = np.zeros(42, dtype=np.int32)
    for name, pos in _PRM.items():
        prm[pos] = out[name]
    prm[36] = struct.unpack("<i", struct.pack("<f", out["rbscoef"]))[0]
    prm[37] = struct.unpack("<i", struct.pack("<f", out["rbscons"]))[0]
    out["blk_prm"] = prm
    out["blk_pb2c"] = np.frombuffer(np.array([out["bclw"], out["bcup"], out["bcce"]], dtype=np.float64).tobytes(), dtype=np.uint8).copy()
    out["blk_cfact"] = np.frombuffer(np.array([out["cfact"]], dtype=np.float64).tobytes(), dtype=np.uint8).copy()
    return out


class BlkBuildParams(C.Structure):       # SpdpBlkBuildParams
    _fields_ = [("ktuple", C.c_int32), ("nshift", C.c_int32), ("blklen", C.c_int32), ("maxgene", C.c_int32), ("nbitpat", C.c_int32),
                ("afact", C.c_int32), ("bitpat", C.c_uint32), ("bitpat2", C.c_uint32), ("threaded", C.c_int32)]


def build_params_default(lib, fasta_bytes: int, nbitpat: int = 1, threaded: int = 0) -> BlkBuildParams:
    """what `spaln -W -KD [-XC<n>]` picks for a FASTA file of that size (spdp_blk_build_params_default)"""
    p = BlkBuildParams()
    lib.spdp_blk_build_params_default.argtypes = [C.c_int64, C.c_int32, C.c_void_p]
    if lib.spdp_blk_build_params_default(int(fasta_bytes), int(nbitpat), C.byref(p)):
        raise RuntimeError("spdp_blk_build_params_default: out of range")
    p.threaded = threaded
    return p


def build_index(eng, genome_codes, chr_off, prm: BlkBuildParams, write_to: str = None, **opts):
    """spdp_blk_index_build (+ spdp_blk_index_write): the block index of a genome made on the device.  Returns (the arrays in
    the layout read_index_file gives, seconds [device, host, call])."""
    lib = eng.lib
    g = Genome()
    gc = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    go = np.ascontiguousarray(chr_off, dtype=np.int64)
    g.codes, g.chr_off, g.n_chr = gc.ctypes.data, go.ctypes.data, len(go) - 1
    o = SearchOpts()
    lib.spdp_blk_search_opts_default(C.byref(o))
    for k, v in opts.items():
        setattr(o, k, v)
    sec = (C.c_double * 3)()
    lib.spdp_blk_index_build.restype = C.c_void_p
    lib.spdp_blk_index_build.argtypes = [C.c_void_p] * 5
    lib.spdp_blk_index_host_free.argtypes = [C.c_void_p]
    h = lib.spdp_blk_index_build(eng.ctx, C.byref(g), C.byref(prm), C.byref(o), sec)
    if not h:
        raise RuntimeError(lib.spdp_last_error(eng.ctx).decode())
    try:
        if write_to is not None:
            lib.spdp_blk_index_write.argtypes = [C.c_void_p, C.c_char_p]
            if lib.spdp_blk_index_write(h, write_to.encode()):
                raise RuntimeError("spdp_blk_index_write: cannot write " + write_to)
        out = _host_index_to_dict(lib, h)
    finally:
        lib.spdp_blk_index_host_free(h)
    return out, list(sec)


class BlkBuildParamsP(C.Structure):      # SpdpBlkBuildParamsP
    _fields_ = [("b", BlkBuildParams), ("nalpha", C.c_int32), ("minorf", C.c_int32), ("aaafact", C.c_double), ("acomp", C.c_double * 20),
                ("convts", C.c_int32), ("convtab", C.c_uint8 * 32)]


def build_params_default_p(lib, fasta_bytes: int, threaded: int = 0, acomp=None) -> BlkBuildParamsP:
    """what `spaln -W -KP` picks for a FASTA file of that size (spdp_blk_build_params_default_p); acomp: MakeBlk::prepacomp's terms,
    by default those of the reference's default tables (defaults.BLOCK_ACOMP_20)"""
    from . import defaults
    p = BlkBuildParamsP()
    for i, v in enumerate(defaults.BLOCK_ACOMP_20 if acomp is None else acomp):
        p.acomp[i] = float(v)
    lib.spdp_blk_build_params_default_p.argtypes = [C.c_int64, C.c_void_p]
    if lib.spdp_blk_build_params_default_p(int(fasta_bytes), C.byref(p)):
        raise RuntimeError("spdp_blk_build_params_default_p: out of range")
    p.b.threaded = threaded
    return p


def build_index_p(eng, genome_codes, chr_off, prm: BlkBuildParamsP, write_to: str = None, **opts):
    """spdp_blk_index_build_p (+ spdp_blk_index_write): the translated block index (`spaln -W -KP`, <db>.bkp) made on the device.
    Returns (the arrays in the layout read_index_file gives, seconds [device, host, call])."""
    lib = eng.lib
    g = Genome()
    gc = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    go = np.ascontiguousarray(chr_off, dtype=np.int64)
    g.codes, g.chr_off, g.n_chr = gc.ctypes.data, go.ctypes.data, len(go) - 1
    o = SearchOpts()
    lib.spdp_blk_search_opts_default(C.byref(o))
    for k, v in opts.items():
        setattr(o, k, v)
    sec = (C.c_double * 3)()
    lib.spdp_blk_index_build_p.restype = C.c_void_p
    lib.spdp_blk_index_build_p.argtypes = [C.c_void_p] * 5
    lib.spdp_blk_index_host_free.argtypes = [C.c_void_p]
    h = lib.spdp_blk_index_build_p(eng.ctx, C.byref(g), C.byref(prm), C.byref(o), sec)
    if not h:
        raise RuntimeError(lib.spdp_last_error(eng.ctx).decode())
    try:
        if write_to is not None:
            lib.spdp_blk_index_write.argtypes = [C.c_void_p, C.c_char_p]
            if lib.spdp_blk_index_write(h, write_to.encode()):
                raise RuntimeError("spdp_blk_index_write: cannot write " + write_to)
        out = _host_index_to_dict(lib, h)
    finally:
        lib.spdp_blk_index_host_free(h)
    return out, list(sec)


class BlkFindParams(C.Structure):        # SpdpBlkFindParams
    _fields_ = [("vthr", C.c_int32), ("drop_rate", C.c_float), ("max_out", C.c_int32), ("max_out2", C.c_int32),
                ("min_agap", C.c_int32), ("phase1t", C.c_int32), ("a_exgl", C.c_int32), ("a_exgr", C.c_int32)]


class Genome(C.Structure):               # SpdpGenome
    _fields_ = [("codes", C.c_void_p), ("chr_off", C.c_void_p), ("n_chr", C.c_int32)]


class Locus(C.Structure):                # SpdpLocus
    _fields_ = [(k, C.c_int32) for k in ("query", "chr", "rvs", "base", "len", "left", "right", "jscr", "n_hsp")] + [("hsp_off", C.c_int64)]


def find_params_from_fixture(fx) -> BlkFindParams:
    """find_prm of a blk_* fixture (oracle/ref_build/blk_tap.cc, dump_find_prm)"""
    v = np.asarray(fx["find_prm"], dtype=np.int32)
    p = BlkFindParams()
    p.vthr = int(v[0])
    p.drop_rate = float(v[1:2].view(np.float32)[0])
    p.max_out, p.max_out2, p.min_agap, p.phase1t = int(v[4]), int(v[5]), int(v[7]), int(v[11])
    p.a_exgl, p.a_exgr = int(v[20]), int(v[21])
    return p


def find(index: "BlockIndex", genome_codes, chr_off, model, sc, prm: BlkFindParams, queries, ranges=None):
    """spdp_blk_find: the block search of every query up to its candidate loci.  Returns (per query a list of dicts
    {chr, rvs, base, len, left, right, jscr, hsps (n + 1, 5)}, status array)."""
    lib, eng = index.lib, index.eng
    n = len(queries)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    codes = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in queries]))
    left = np.array([0 if ranges is None else ranges[i][0] for i in range(n)], dtype=np.int32)
    right = np.array([len(queries[i]) if ranges is None else ranges[i][1] for i in range(n)], dtype=np.int32)
    g = Genome()
    gc = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    go = np.ascontiguousarray(chr_off, dtype=np.int64)
    g.codes, g.chr_off, g.n_chr = gc.ctypes.data, go.ctypes.data, len(go) - 1
    loci = C.POINTER(Locus)()
    hsps = C.POINTER(C.c_int32)()
    nl = C.c_int32()
    status = np.zeros(n, dtype=np.int32)
    lib.spdp_blk_find.restype = C.c_int
    lib.spdp_blk_find.argtypes = [C.c_void_p] * 11 + [C.c_int32] + [C.c_void_p] * 4
    rc = lib.spdp_blk_find(eng.ctx, index.h, C.byref(index.desc), C.byref(g), C.addressof(model), C.byref(sc), C.byref(prm),
                           codes.ctypes.data, offs.ctypes.data, left.ctypes.data, right.ctypes.data, n,
                           C.byref(loci), C.byref(nl), C.byref(hsps), status.ctypes.data)
    eng._check(rc, "spdp_blk_find")
    out = [[] for _ in range(n)]
    for k in range(nl.value):
        L = loci[k]
        h = np.array([[hsps[5 * (L.hsp_off + j) + c] for c in range(5)] for j in range(L.n_hsp + 1)], dtype=np.int32)
        out[L.query].append(dict(chr=L.chr, rvs=L.rvs, base=L.base, len=L.len, left=L.left, right=L.right, jscr=L.jscr, hsps=h))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(loci)
    libc.free(hsps)
    return out, status


class MapExon(C.Structure):
    _fields_ = [("q_left", C.c_int32), ("q_right", C.c_int32), ("g_left", C.c_int32), ("g_right", C.c_int32)]


class MapGene(C.Structure):
    _fields_ = [("chr", C.c_int32), ("rvs", C.c_int32), ("q_rev", C.c_int32), ("score", C.c_int32), ("val", C.c_int32), ("n_loci", C.c_int32),
                ("n_exons", C.c_int32), ("exon_off", C.c_int64)]


def map_align(index: "BlockIndex", genome_codes, chr_off, sc, sp, sigmodel, prm: BlkFindParams, rescore, queries, ori: int = 1):
    """spdp_map_align_s: block search -> loci -> signals -> seeded alignment -> rescoring, one call for all queries.
    rescore = (codonk1, minl, jneibr, lsg).  Returns (per query None or dict(chr, rvs, score, val, n_loci,
    exons = [(q_left, q_right, g_left, g_right)]), seconds [find, regions + signals, align, rescore], return code)."""
    from . import abi
    lib, eng = index.lib, index.eng
    n = len(queries)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    codes = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in queries]))
    g = Genome()
    gc = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    go = np.ascontiguousarray(chr_off, dtype=np.int64)
    g.codes, g.chr_off, g.n_chr = gc.ctypes.data, go.ctypes.data, len(go) - 1
    rp = abi.RescoreParams(*(int(x) for x in rescore))
    genes = (MapGene * n)()
    exons = C.POINTER(MapExon)()
    sec = (C.c_double * 4)()
    lib.spdp_map_align_s.restype = C.c_int
    lib.spdp_map_align_s.argtypes = [C.c_void_p] * 11 + [C.c_int32, C.c_int32] + [C.c_void_p] * 3
    rc = lib.spdp_map_align_s(eng.ctx, index.h, C.byref(index.desc), C.byref(g), C.byref(sc), C.byref(sp), C.addressof(sigmodel),
                              C.byref(prm), C.byref(rp), codes.ctypes.data, offs.ctypes.data, n, int(ori), genes, C.byref(exons), sec)
    if rc < 0:
        eng._check(rc, "spdp_map_align_s")
    out = []
    for i in range(n):
        G = genes[i]
        if G.chr < 0:
            out.append(None)
            continue
        ex = [(exons[G.exon_off + j].q_left, exons[G.exon_off + j].q_right, exons[G.exon_off + j].g_left, exons[G.exon_off + j].g_right)
              for j in range(G.n_exons)]
        out.append(dict(chr=G.chr, rvs=G.rvs, q_rev=G.q_rev, score=G.score, val=G.val, n_loci=G.n_loci, exons=ex))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(exons)
    return out, list(sec), rc


def map_align_h(index: "BlockIndex", genome_codes, chr_off, sc, sp, sigmodel, prm: BlkFindParams, rescore, queries):
    """spdp_map_align_h: the same for protein queries against the translated index.  sc: abi.ScoringH; rescore = abi.RescoreParamsH;
    Returns as map_align."""
    lib, eng = index.lib, index.eng
    n = len(queries)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    codes = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in queries]))
    g = Genome()
    gc = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    go = np.ascontiguousarray(chr_off, dtype=np.int64)
    g.codes, g.chr_off, g.n_chr = gc.ctypes.data, go.ctypes.data, len(go) - 1
    genes = (MapGene * n)()
    exons = C.POINTER(MapExon)()
    sec = (C.c_double * 4)()
    lib.spdp_map_align_h.restype = C.c_int
    lib.spdp_map_align_h.argtypes = [C.c_void_p] * 11 + [C.c_int32] + [C.c_void_p] * 3
    rc = lib.spdp_map_align_h(eng.ctx, index.h, C.byref(index.desc), C.byref(g), C.byref(sc), C.byref(sp), C.addressof(sigmodel),
                              C.byref(prm), C.byref(rescore), codes.ctypes.data, offs.ctypes.data, n, genes, C.byref(exons), sec)
    if rc < 0:
        eng._check(rc, "spdp_map_align_h")
    out = []
    for i in range(n):
        G = genes[i]
        if G.chr < 0:
            out.append(None)
            continue
        ex = [(exons[G.exon_off + j].q_left, exons[G.exon_off + j].q_right, exons[G.exon_off + j].g_left, exons[G.exon_off + j].g_right)
              for j in range(G.n_exons)]
        out.append(dict(chr=G.chr, rvs=G.rvs, q_rev=G.q_rev, score=G.score, val=G.val, n_loci=G.n_loci, exons=ex))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(exons)
    return out, list(sec), rc
