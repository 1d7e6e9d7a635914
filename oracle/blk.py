"""ctypes wrapper of the block-search restatement (oracle/spdp_oracle_blk.c).  TEST INFRASTRUCTURE ONLY.

The index and parameter record come from a fixture written by the reference itself (oracle/ref_build/blk_tap.cc):
`index_of(fx)` turns it into the BlkIndex the oracle reads; `parse_log(fx)` splits the recorder's query log."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import oracle as _o


class BlkIndex(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nalpha", "ktuple", "tabsize", "nshift", "blklen", "nbitpat", "convts", "n_chr", "avrscr", "maxblk",
        "kk", "drna", "maxmmc", "nseg", "minsigpr", "ncand", "nascr", "maxblock", "extblock", "shortquery",
        "hh_size1", "hh_size2", "hb_size1", "hb_size2", "ha_size1", "ha_size2", "phase1t", "gdb", "has_chrid", "extblockl")] + [
        ("rbscoef", C.c_float), ("rbscons", C.c_float),
        ("bclw", C.c_double), ("bcup", C.c_double), ("bcce", C.c_double), ("cfact", C.c_double),
        ("convtab", C.c_void_p), ("nblk", C.c_void_p), ("wscr", C.c_void_p), ("blkp", C.c_void_p), ("blkb", C.c_void_p),
        ("rscrtab", C.c_void_p), ("chr", C.c_void_p), ("bitpat", C.c_void_p)]


# positions in the recorder's blk_prm record (blk_tap.cc, dump_index)
PRM = dict(nalpha=0, ktuple=1, bitpat2=2, tabsize=3, bitpat=4, nshift=5, blklen=6, maxgene=7, nbitpat=8, afact=9, convts=10,
           wordno=11, n_chr=12, avrscr=13, maxblk=14, kk=15, drna=16, maxmmc=17, ptpl=18, nseg=19, bbt=20, min_agap=21,
           minsigpr=22, ncand=23, nascr=24, maxblock=25, extblock=26, extblockl=27, shortquery=28, hh_size1=29, hh_size2=30,
           hb_size1=31, hb_size2=32, ha_size1=33, ha_size2=34, phase1t=35, rbscoef=36, rbscons=37, gdb=38, maxout=39,
           maxout2=40, has_chrid=41)


def index_of(fx: dict):
    """(BlkIndex, keep-alive list) from the arrays of a blk_*.spdg fixture"""
    prm = np.asarray(fx["blk_prm"], dtype=np.int32)
    ix = BlkIndex()
    for name, _ in BlkIndex._fields_[:30]:
        if name in PRM:
            setattr(ix, name, int(prm[PRM[name]]))
    ix.rbscoef = struct.unpack("<f", struct.pack("<i", int(prm[PRM["rbscoef"]])))[0]
    ix.rbscons = struct.unpack("<f", struct.pack("<i", int(prm[PRM["rbscons"]])))[0]
    ix.bclw, ix.bcup, ix.bcce = np.frombuffer(np.asarray(fx["blk_pb2c"], dtype=np.uint8).tobytes(), dtype=np.float64)
    ix.cfact = float(np.frombuffer(np.asarray(fx["blk_cfact"], dtype=np.uint8).tobytes(), dtype=np.float64)[0])
    keep = []
    for field, key, dt in (("convtab", "blk_convtab", np.uint8), ("nblk", "blk_nblk", np.uint16), ("wscr", "blk_wscr", np.int16),
                           ("blkp", "blk_blkp", np.int32), ("blkb", "blk_blkb", np.uint32), ("rscrtab", "blk_rscrtab", np.int32),
                           ("chr", "blk_chr", np.int32), ("bitpat", "blk_bitpat", np.int32)):
        a = np.ascontiguousarray(np.asarray(fx[key]).view(dt) if np.asarray(fx[key]).dtype.itemsize == np.dtype(dt).itemsize
                                 else np.asarray(fx[key]).astype(dt))
        keep.append(a)
        setattr(ix, field, a.ctypes.data)
    return ix, keep


def parse_log(fx: dict):
    """the recorder's query log -> [dict(left, right, codes, calls=[(vote record, pairs record or None)])]"""
    L = np.asarray(fx["q_log"], dtype=np.int32)
    ncand1 = int(fx["blk_prm"][PRM["ncand"]]) + 1
    out, i = [], 0
    while i < L.size:
        assert L[i] == -1, (i, L[i])
        left, right, n = int(L[i + 1]), int(L[i + 2]), int(L[i + 3])
        q = dict(left=left, right=right, codes=L[i + 4:i + 4 + n].astype(np.uint8), calls=[])
        i += 4 + n
        while i < L.size and L[i] == -2:
            j = i + 1 + 20
            for _ in range(4):
                for _ in range(4):              # prqueue_b, prqueue_a, bscr, ascr: count + pairs
                    j += 1 + 2 * int(L[j])
            vote = L[i:j].copy()
            pairs = None
            if j < L.size and L[j] == -3:
                assert L[j + 1] == ncand1
                pairs = L[j:j + 2 + 9 * ncand1].copy()
                j += 2 + 9 * ncand1
            q["calls"].append((vote, pairs))
            i = j
        out.append(q)
    return out


def vote(ix: BlkIndex, codes: np.ndarray, left: int, right: int, stop_at: int = 0):
    """state at the stop_at-th TestOutput call + the block pairs built there: (vote record, pairs record) as int32 arrays in
    the recorder's layout (pairs: -3, n, 9 ints per pair), or None if findblock ends before that call"""
    lib = _o.lib()
    q = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(64 + 16 * (4 * int(ix.nseg) + 64), dtype=np.int32)
    lib.spdp_oracle_blk_vote.restype = C.c_int
    n = lib.spdp_oracle_blk_vote(C.byref(ix), q.ctypes.data_as(C.c_void_p), C.c_int(q.size), C.c_int(left), C.c_int(right),
                                 C.c_int(stop_at), out.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise RuntimeError("hash table overflow in the block vote (the reference resizes there)")
    if n == 0:
        return None
    rec = out[:n]
    k = int(np.nonzero(rec == -3)[0][-1])
    # (a -3 can also occur as data only if a score is -3: scan from the structure instead)
    j = 1 + 20
    for _ in range(4):
        for _ in range(4):
            j += 1 + 2 * int(rec[j])
    assert rec[j] == -3, (j, k)
    return rec[:j].copy(), rec[j:].copy()


def split_product_record(rec: np.ndarray):
    """one query's record of the product (spdp_blk_core.h, blk_emit_and_reset) -> dict, or None when the asked call was not
    reached"""
    n, calls, flags = int(rec[0]), int(rec[1]), int(rec[2])
    if not flags & 1:
        return dict(reached=False, calls=calls, flags=flags)
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


def split_recorded(vote_rec: np.ndarray, pairs_rec):
    """the recorder's -2 / -3 records -> the same dict shape (word-hit counts and the all-hits queue are left out: the product
    does not report them)"""
    r = vote_rec
    head = r[1:21].copy()
    j = 21
    qb, runs = [], []
    for _ in range(4):
        k = int(r[j]); qb.append(r[j + 1:j + 1 + 2 * k].reshape(k, 2).copy()); j += 1 + 2 * k
        k = int(r[j]); j += 1 + 2 * k                                   # prqueue_a
        k = int(r[j]); runs.append([(int(a), int(b)) for a, b in r[j + 1:j + 1 + 2 * k].reshape(k, 2)]); j += 1 + 2 * k
        k = int(r[j]); j += 1 + 2 * k                                   # ascr
    pairs = None if pairs_rec is None else pairs_rec[2:].reshape(-1, 9).copy()
    return dict(head=head, qb=qb, runs=runs, pairs=pairs)


def last_vote_was_forced() -> bool:
    """the snapshot vote() returned last is findblock's closing call, TestOutput(1)"""
    return bool(C.c_int.in_dll(_o.lib(), "spdp_oracle_blk_last_forced").value)


def vote_carry(ix: BlkIndex, codes, left, right, stop_at, carry: np.ndarray):
    """as vote(), with the memory the reference's worker thread keeps from query to query (see spdp_oracle_blk_vote_carry);
    `carry` is updated in place.  Returns the raw record (vote + pairs) or None."""
    lib = _o.lib()
    q = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(64 + 16 * (4 * int(ix.nseg) + 64), dtype=np.int32)
    n = lib.spdp_oracle_blk_vote_carry(C.byref(ix), q.ctypes.data_as(C.c_void_p), C.c_int(q.size), C.c_int(left), C.c_int(right),
                                       C.c_int(stop_at), out.ctypes.data_as(C.c_void_p), carry.ctypes.data_as(C.c_void_p))
    return None if n <= 0 else out[:n].copy()


def new_carry(ix: BlkIndex) -> np.ndarray:
    return np.zeros(8 * (int(ix.nascr) + 1) + 8, dtype=np.int32)


def grows() -> int:
    """how often a hash table of the oracle has grown so far (Dhash::resize)"""
    return int(C.c_int.in_dll(_o.lib(), "spdp_oracle_blk_grows").value)


def runs_near_pairs(ix: BlkIndex, runs, pairs):
    """the part of the recorded run scores the product reports: blocks within ExtBlockL of a reported pair, inside its
    chromosome, on its strand (runs: per direction [(block, score)], pairs: rows of nine ints)"""
    e = 4 * max(int(ix.extblockl), int(ix.extblock))     # (as spdp_blk_vote.hip, write_state: FindHsp's reach over its retries)
    out = []
    for d in range(4):
        keep = []
        for blk_, scr in runs[d]:
            for bscr, c, lb, rb, ub, db, zl, zr, rvs in (tuple(int(x) for x in p) for p in pairs):
                if rvs == d >> 1 and max(lb - e, zl, 0) <= blk_ <= min(rb + e, zr):
                    keep.append((blk_, scr))
                    break
        out.append(keep)
    return out


class BuildParams(C.Structure):
    """OrcBlkBuildParams (oracle/spdp_oracle_blkidx.c) = SpdpBlkBuildParams (include/spdp.h)"""
    _fields_ = [("ktuple", C.c_int32), ("nshift", C.c_int32), ("blklen", C.c_int32), ("maxgene", C.c_int32), ("nbitpat", C.c_int32),
                ("afact", C.c_int32), ("bitpat", C.c_uint32), ("bitpat2", C.c_uint32), ("threaded", C.c_int32)]


def build_params_of(fx: dict, threaded: int = 0) -> BuildParams:
    """the BlkWcPrm the reference's own index of a blk_* fixture was built with"""
    v = np.asarray(fx["blk_prm"], dtype=np.int64)
    p = BuildParams()
    p.ktuple, p.nshift, p.blklen, p.maxgene = int(v[PRM["ktuple"]]), int(v[PRM["nshift"]]), int(v[PRM["blklen"]]), int(v[PRM["maxgene"]])
    p.nbitpat, p.afact = int(v[PRM["nbitpat"]]), int(v[PRM["afact"]])
    p.bitpat, p.bitpat2 = int(v[PRM["bitpat"]]) & 0xffffffff, int(v[PRM["bitpat2"]]) & 0xffffffff
    p.threaded = threaded
    return p


def index_build(codes, chr_off, prm: BuildParams) -> dict:
    """MakeBlk::idxblk + blkscrtab on the CPU: dict(nblk, blkp, wscr, blkb, chr, b2c, word_no, glen, avrscr, maxblk, bytblk,
    n_blocks, minscr)"""
    lib = _o.lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    off = np.ascontiguousarray(chr_off, dtype=np.int64)
    n_chr = len(off) - 1
    tab = 1 << (2 * prm.ktuple)
    nblk = np.zeros(tab, np.uint16); blkp = np.zeros(tab, np.int32); wscr = np.zeros(tab, np.int16)
    chr_ = np.zeros(2 * (n_chr + 1), np.int32); b2c = np.zeros(3, np.float64); head = np.zeros(8, np.int64)
    blkb = C.POINTER(C.c_uint32)()
    lib.orc_blk_index_build.restype = C.c_int
    lib.orc_blk_index_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 7
    rc = lib.orc_blk_index_build(codes.ctypes.data, off.ctypes.data, n_chr, C.byref(prm), nblk.ctypes.data, blkp.ctypes.data,
                                 wscr.ctypes.data, C.byref(blkb), chr_.ctypes.data, b2c.ctypes.data, head.ctypes.data)
    if rc:
        raise RuntimeError(f"orc_blk_index_build: {rc}")
    n = int(head[0])
    words = np.ctypeslib.as_array(blkb, shape=(max(n, 1),))[:n].copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(blkb)
    return dict(nblk=nblk, blkp=blkp, wscr=wscr, blkb=words, chr=chr_, b2c=b2c, word_no=n, glen=int(head[1]), avrscr=int(head[2]),
                maxblk=int(head[3]), bytblk=int(head[4]), n_blocks=int(head[5]), minscr=int(head[6]))


class BuildParamsP(C.Structure):
    """OrcBlkBuildParamsP (oracle/spdp_oracle_blkidx.c) = SpdpBlkBuildParamsP (include/spdp.h): the translated index (`spaln -W -KP`)"""
    _fields_ = [("b", BuildParams), ("nalpha", C.c_int32), ("minorf", C.c_int32), ("aaafact", C.c_double), ("acomp", C.c_double * 24),
                ("codon_class", C.c_uint8 * 64)]


_STANDARD_CODE = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"     # codon 16 b1 + 4 b2 + b3, A C G T = 0 1 2 3
_TRON_ORDER = "ARNDCQEGHILKMFPSTWYV"                                                      # tron codes 3 .. 22


def codon_classes(convtab, nalpha: int) -> np.ndarray:
    """codon -> class of the reduced amino-acid alphabet as ReducWord::ReducWord builds g2r (src/bitpat.cc:88-106) from the index's
    ConvTab (tron code -> class); stop codons and Sec: 255 / nalpha, neither a class"""
    out = np.full(64, 255, dtype=np.uint8)
    for g, aa in enumerate(_STANDARD_CODE):
        if aa == "*":
            continue                                                                     # (TGA: class "U" = nalpha, TAA / TAG: none)
        code = 3 + _TRON_ORDER.index(aa)
        if aa == "S" and g >> 4 == 0:
            code = 23                                                                    # AGY serines: 'J' of the tron alphabet
        out[g] = int(convtab[code])
    return out


def build_params_p(ktuple, nshift, blklen, maxgene, afact, threaded, convtab, acomp, nalpha=20, minorf=30, aaafact=1.0) -> BuildParamsP:
    p = BuildParamsP()
    p.b.ktuple, p.b.nshift, p.b.blklen, p.b.maxgene, p.b.nbitpat, p.b.afact = ktuple, nshift, blklen, maxgene, 1, afact
    p.b.bitpat, p.b.bitpat2, p.b.threaded = (1 << ktuple) - 1, 0, threaded
    p.nalpha, p.minorf, p.aaafact = nalpha, minorf, aaafact
    for i, v in enumerate(acomp):
        p.acomp[i] = float(v)
    cc = codon_classes(convtab, nalpha)
    for i in range(64):
        p.codon_class[i] = int(cc[i])
    return p


def index_build_tron(codes, chr_off, prm: BuildParamsP) -> dict:
    """MakeBlk::idxblk / m_idxblk + blkscrtab(segn) for a translated index on the CPU; the dict of index_build"""
    lib = _o.lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    off = np.ascontiguousarray(chr_off, dtype=np.int64)
    n_chr = len(off) - 1
    tab = int(prm.nalpha) ** int(prm.b.ktuple)
    nblk = np.zeros(tab, np.uint16); blkp = np.zeros(tab, np.int32); wscr = np.zeros(tab, np.int16)
    chr_ = np.zeros(2 * (n_chr + 1), np.int32); b2c = np.zeros(3, np.float64); head = np.zeros(8, np.int64)
    blkb = C.POINTER(C.c_uint32)()
    lib.orc_blk_index_build_tron.restype = C.c_int
    lib.orc_blk_index_build_tron.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 7
    rc = lib.orc_blk_index_build_tron(codes.ctypes.data, off.ctypes.data, n_chr, C.byref(prm), nblk.ctypes.data, blkp.ctypes.data,
                                      wscr.ctypes.data, C.byref(blkb), chr_.ctypes.data, b2c.ctypes.data, head.ctypes.data)
    if rc:
        raise RuntimeError(f"orc_blk_index_build_tron: {rc}")
    n = int(head[0])
    words = np.ctypeslib.as_array(blkb, shape=(max(n, 1),))[:n].copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(blkb)
    return dict(nblk=nblk, blkp=blkp, wscr=wscr, blkb=words, chr=chr_, b2c=b2c, word_no=n, glen=int(head[1]), avrscr=int(head[2]),
                maxblk=int(head[3]), bytblk=int(head[4]), n_blocks=int(head[5]), minscr=int(head[6]))
