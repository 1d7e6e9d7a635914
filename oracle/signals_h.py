"""CPU restatement of the protein-side signal precompute (TEST INFRASTRUCTURE ONLY).

What it follows (ogotoh/spaln v3.0.7):
  Exinon::intron53_p                src/codepot.cc:524-611   the SGPT6 arrays: sig5 / sig3 (+ phases), sigS, sigT, sigE
  Exinon::intron53_c                src/codepot.cc:435-476   dinucleotide classes / canonical-site levels on a tron sequence
  PatMat::calcPatMat                src/utilseq.cc:905-1000  position weight matrices of Markov order <= 1 and 2
  ExinPot::calcScr_3                src/utilseq.cc:1423-1460 5th-order Markov coding potential, three phases
The model (matrices, potential table, factors) is data read back from a reference run: every protein fixture carries it
(pm*_hdr / pm*_f32, potC_*, sigmodel_f32 / _i32, written by oracle/ref_build/ref_dump_h.cc).
A tron sequence is read through tnredctab (src/seq.cc:41): tron code -> the middle base of its codon, 4 = none."""
from __future__ import annotations

import numpy as np

from oracle import signals

TNRED = np.full(256, 4, dtype=np.int64)
TNRED[:26] = [4, 4, 4, 1, 2, 0, 0, 2, 0, 0, 2, 0, 3, 3, 0, 3, 3, 1, 1, 1, 2, 0, 3, 2, 2, 0]
F32 = np.float32


class PatMat:
    def __init__(self, hdr, f32):
        self.rows, self.cols, self.offset, self.order, self.nalpha = (int(x) for x in hdr)
        self.present = self.rows > 0
        if self.present:
            f = np.asarray(f32, dtype=np.int32).view(np.float32)
            self.tonic, self.min_elem = F32(f[0]), F32(f[1])
            self.mtx = f[2:].astype(np.float32)


def scan1(pm: PatMat, x: np.ndarray, pos: int) -> np.float32:
    """calcPatMat, Markov order <= 1 (utilseq.cc:918-945): a bad base stops the sum (the remaining columns add 0)"""
    ln = x.size
    n = pos - pm.offset
    c0 = 0
    q = 1 if n + pm.cols >= ln else 0
    if n < 0:
        c0, n = -n, 0
    last = min(n + (pm.cols - c0), ln - pm.order)
    fit = F32(0)
    col = c0
    first = True
    for s in range(n, last):
        row = pm.mtx[col * pm.rows:(col + 1) * pm.rows]
        k = int(x[s])
        if k > 3:
            q += 1
        if pm.order and not q:
            if first:
                fit = F32(fit + row[k])
            j = int(x[s + 1])
            if j > 3:
                q += 1
            k = 4 * k + j + 4
        if not q:
            fit = F32(fit + row[k])
        first = False
        col += 1
    return F32(fit + pm.tonic)


def scan(pm: PatMat, x: np.ndarray, pos: int) -> np.float32:
    if pm.order == 2:
        return signals.scan(pm, x, pos)
    return scan1(pm, x, pos)


def coding_potential(tab: np.ndarray, ndata: int, x: np.ndarray, start: int, stop: int) -> np.ndarray:
    """ExinPot::calcScr_3 over sequence positions [start, stop): out[j] belongs to position start + j (the value the
    loop computes five bases further on); the hash restarts at `start` and after every base that is not A C G T"""
    n = stop - start
    out = np.zeros(max(n, 0), dtype=np.float32)
    buf = [0, 0, 0]
    w, need, p = 0, 6, 1
    for t in range(n):
        c = int(x[start + t])
        if c < 4:
            w = (4 * w + c) % ndata
            buf[p] = 3 * w
            if need:
                need -= 1
        else:
            w, need = 0, 6
        val = F32(0)
        if not need:
            val = F32(val + tab[buf[(p + 1) % 3] + 2])
            val = F32(val + tab[buf[(p + 2) % 3]])
            val = F32(val + tab[buf[p] + 1])
        if t >= 5:
            out[t - 5] = val
        p = (p + 1) % 3
    return out


def model_of(fx: dict) -> dict:
    f = np.asarray(fx["sigmodel_f32"], dtype=np.int32).view(np.float32)
    i = [int(v) for v in fx["sigmodel_i32"]]
    return dict(pm5=PatMat(fx["pm5_hdr"], fx["pm5_f32"]), pm3=PatMat(fx["pm3_hdr"], fx["pm3_f32"]),
                pmI=PatMat(fx["pmI_hdr"], fx["pmI_f32"]), pmT=PatMat(fx["pmT_hdr"], fx["pmT_f32"]),
                pmB=PatMat(fx["pmB_hdr"], fx["pmB_f32"]),
                pot=np.asarray(fx["potC_f32"], dtype=np.int32).view(np.float32), pot_ndata=int(fx["potC_hdr"][0]),
                fE=F32(f[0]), fI=F32(f[1]), fT=F32(f[2]), fB=F32(f[3]), fO=F32(f[4]), fS=F32(f[5]), fs=F32(f[6]),
                tonic3=F32(f[7]), tonic5=F32(f[8]), tonicB=F32(f[9]), maxb3d=i[2],
                any=i[0], dvsp=i[1] != 3, trm=(i[5], i[6]), tab=np.asarray(fx["sig53tab01"], dtype=np.int64))


def classes(b: np.ndarray, left: int, right: int, any_: int = 0):
    """intron53_c on a tron sequence: as oracle.signals.classes with the tron reduction table"""
    c = TNRED[b].copy()
    c[c > 3] = 1
    n = b.size
    prev = np.concatenate([[1], c[:-1]])
    if left < n:
        prev[left] = 1
    nc = ((prev << 2) + c) & 0xf
    k5t, k3t = signals.cano_tables(any_, 0)
    d5 = np.zeros(n + 3, dtype=np.uint8); d3 = np.zeros(n + 3, dtype=np.uint8)
    c5 = np.zeros(n + 3, dtype=np.uint8); c3 = np.zeros(n + 3, dtype=np.uint8)
    for i in range(left, right):
        if i - 1 >= 0:
            d5[i - 1] = nc[i]; c5[i - 1] = k5t[nc[i]]
        d3[i + 1] = nc[i]; c3[i + 1] = k3t[nc[i]]
    return d5, d3, c5, c3


def splice_signals_h(md: dict, b: np.ndarray, b_len: int, left: int, right: int) -> dict:
    """the SGPT6 arrays of positions left .. right - 1 (arrays of b_len + 3 entries, -2 / 0 elsewhere).  `b` holds
    b_len + 1 tron codes (with the terminator the engines read)."""
    b = np.asarray(b, dtype=np.uint8)
    seq = b[:b_len]
    x = TNRED[seq]
    N = b_len + 3
    d5, d3, c5, c3 = classes(seq, left, right, md["any"])
    out = {k: np.zeros(N, dtype=np.int16) for k in ("sig5", "sig3", "sigS", "sigT", "sigE")}
    out["phs5"] = np.full(N, -2, dtype=np.int8); out["phs3"] = np.full(N, -2, dtype=np.int8)
    pot = coding_potential(md["pot"], md["pot_ndata"], x, max(left - 1, 0), min(right + 1, b_len)) if md["pot"].size else None
    pot0 = max(left - 1, 0)
    th5 = int(F32(md["fS"] * md["tonic5"])); th3 = int(F32(md["fS"] * md["tonic3"]))
    trm = md["trm"]
    # the branch-point carry (-yB; src/codepot.cc:536, 567-568, 586-597)
    use_b = md["pmB"].present
    th_b = float(np.int16(int(md["tonicB"]))) if use_b else 0.0
    sig_b, pos_b = 0, -1
    for pos in range(left, right):
        if md["dvsp"] and md["pmI"].present:
            out["sigS"][pos] = int(F32(md["fT"] * scan(md["pmI"], x, pos)))
        if md["dvsp"] and md["pmT"].present:
            out["sigT"][pos] = int(F32(md["fT"] * scan(md["pmT"], x, pos)))
        if pot is not None:
            e = F32(md["fE"] * pot[pos - pot0])
            if int(seq[pos]) in trm:
                e = F32(e + md["fO"])
            elif pos + 3 < right and pos + 3 < b_len and int(seq[pos + 3]) in trm:
                e = F32(0)
            out["sigE"][pos] = int(e)
        s5 = int(F32(md["fs"] * scan(md["pm5"], x, pos))) + int(md["tab"][d5[pos]])
        s3 = int(F32(md["fs"] * scan(md["pm3"], x, pos))) + int(md["tab"][16 + d3[pos]])
        if use_b:
            s3 = int(np.int16(s3 + sig_b))
            sb = scan(md["pmB"], x, pos)
            if sb > th_b:
                sig_b, pos_b = int(np.int16(int(F32(md["fB"] * sb)))), pos
            if pos_b >= 0 and pos - pos_b > md["maxb3d"]:
                sig_b, pos_b = 0, -1
        out["sig5"][pos] = s5
        out["sig3"][pos] = s3
        for sig, cano, ph, th in ((s5, c5, out["phs5"], th5), (s3, c3, out["phs3"], th3)):
            if ph[pos] == -2 and ((md["any"] == 2 and sig > th) or cano[pos]):
                ph[pos] = 0
                if cano[pos] > 1:
                    ph[pos + 1] = 1
                    if pos >= 1:
                        ph[pos - 1] = 2 if ph[pos - 1] == 1 else -1
    return out
