"""CPU restatement of the splice-signal precompute (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

What it follows (ogotoh/spaln v3.0.7):
  Exinon::intron53_n                src/codepot.cc:479-520   sig5[n] = (short)(fs * P5(n)) + tab0[dinc5[n]], sig3 likewise
  PatMat::calcPatMat (order 2)      src/utilseq.cc:946-996   P(n): second-order Markov position weight matrix scan
  Exinon::intron53_c                src/codepot.cc:435-476   dinucleotide classes (see spaln_amd.synth.exact_inputs)
The model (two PatMat tables, the 2 x 16 per-class terms, the scale fs) is data read back from a reference run: every
fixture carries it (pm5_* / pm3_* / sig53tab01 / sigmodel, written by oracle/ref_build/ref_dump.cc).
float32 arithmetic in the reference's order of additions: numpy float32 scalars, no fused operations."""
from __future__ import annotations

import numpy as np

RED = np.full(256, 1, dtype=np.int64)            # ncredctab: A C G T -> 0..3, everything else counts as 'C' in the classes
for _code, _c in ((2, 0), (3, 1), (5, 2), (9, 3)):
    RED[_code] = _c
RED_STRICT = np.full(256, 4, dtype=np.int64)     # for the scan: anything but A C G T is a "bad char"
for _code, _c in ((2, 0), (3, 1), (5, 2), (9, 3)):
    RED_STRICT[_code] = _c


class PatMat:
    def __init__(self, hdr, f32):
        self.rows, self.cols, self.offset, self.order, self.nalpha = (int(x) for x in hdr)
        f = np.asarray(f32, dtype=np.int32).view(np.float32)
        self.tonic, self.min_elem = np.float32(f[0]), np.float32(f[1])
        self.mtx = f[2:].astype(np.float32)
        assert self.mtx.size == self.rows * self.cols and self.order == 2 and self.nalpha == 4 and self.rows == 84


def model_of(fx: dict) -> dict:
    return dict(pm5=PatMat(fx["pm5_hdr"], fx["pm5_f32"]), pm3=PatMat(fx["pm3_hdr"], fx["pm3_f32"]),
                tab=np.asarray(fx["sig53tab01"], dtype=np.int64), any=int(fx["sigmodel"][1]), both_ori=int(fx["sigmodel"][3]),
                fs=np.asarray(fx["sigmodel"][:1], dtype=np.int32).view(np.float32)[0])


def scan(pm: PatMat, x: np.ndarray, pos: int) -> np.float32:
    """calcPatMat's value for sequence position `pos` (window start pos - offset); x = strict reduced codes"""
    ln = x.size
    n = pos - pm.offset
    c0 = 0
    if n < 0:
        c0, n = -n, 0                               # the first -n columns of the matrix have no base under them
    q = 1 if (pos - pm.offset) + pm.cols >= ln else 0
    last = min(n + (pm.cols - c0), ln - 2)          # tt = at(n + cols) clipped to at(len - order)
    fit = np.float32(0)
    first = True
    col = c0
    for s in range(n, last):
        row = pm.mtx[col * pm.rows:(col + 1) * pm.rows]
        i0 = int(x[s]); i1 = int(x[s + 1]); i2 = int(x[s + 2])
        k = i0
        if i0 > 3:
            q += 1
        if first and q == 0:
            fit = np.float32(fit + row[k])
        if i1 > 3:
            q += 1
        elif q == 0:
            k = 4 * k + i1
            if first:
                fit = np.float32(fit + row[k + 4])
        if i2 > 3:
            q += 1
        elif q == 0:
            k = 4 * k + i2
            fit = np.float32(fit + row[k + 20])
        first = False
        col += 1
    if q:
        fit = np.float32(np.float32(pm.cols) * pm.min_elem)
    return np.float32(fit + pm.tonic)


# cano levels by dinucleotide class (A C G T = 0..3, class = 4 * first + second) for algmode.any = 0..3:
# codepot.cc:438-439 jlevelac = {0, 2, 3, 1}, jlevelgt = {0, 0, 3, 1}
def cano_tables(any_: int, both_ori: int):
    ac, gt = (0, 2, 3, 1)[any_], (0, 0, 3, 1)[any_]
    base = 1 if any_ == 3 else 0
    k5 = [base] * 16; k3 = [base] * 16
    k3[0] = ac
    k3[1] = 2
    k3[2] = 3
    k5[3] = 2; k3[3] = ac
    k3[6] = gt
    k5[7] = gt
    k5[8] = gt
    k5[9] = 3
    k5[10] = gt; k3[10] = gt
    k5[11] = 3
    k3[14] = gt
    k5[15] = gt
    if both_ori:
        k5[1] = 1; k3[7] = 1; k3[11] = 1
    return np.array(k5, dtype=np.uint8), np.array(k3, dtype=np.uint8)


def classes(b_codes: np.ndarray, left: int, right: int, any_: int = 0, both_ori: int = 0):
    """dinc5 / dinc3 / cano5 / cano3 per position 0 .. len(b) as intron53_c leaves them for the range [left, right):
    the class chain starts at `left` from 'C'; cells it does not write stay 0"""
    b = np.asarray(b_codes, dtype=np.uint8)
    n = b.size
    c = RED[b]
    prev = np.concatenate([[1], c[:-1]])
    prev[left] = 1
    nc = ((prev << 2) + c) & 0xf
    k5t, k3t = cano_tables(any_, both_ori)
    d5 = np.zeros(n + 3, dtype=np.uint8); d3 = np.zeros(n + 3, dtype=np.uint8)
    c5 = np.zeros(n + 3, dtype=np.uint8); c3 = np.zeros(n + 3, dtype=np.uint8)
    for i in range(left, right):                    # base i: wk5 = position i - 1, wk3 = position i + 1
        if i - 1 >= 0:
            d5[i - 1] = nc[i]; c5[i - 1] = k5t[nc[i]]
        d3[i + 1] = nc[i]; c3[i + 1] = k3t[nc[i]]
    return d5[:n + 1], d3[:n + 1], c5[:n + 1], c3[:n + 1]


def splice_signals(model: dict, b_codes: np.ndarray, left: int, right: int, lo: int | None = None, hi: int | None = None):
    """sig5, sig3 (int16, len(b) + 1) for positions left .. right - 1, zero elsewhere (vset ZeroSGPT2), with the
    classes of the same range; lo / hi restrict the positions evaluated (tests on long windows)"""
    b = np.asarray(b_codes, dtype=np.uint8)
    x = RED_STRICT[b]
    n = b.size
    d5, d3, _, _ = classes(b, left, right, int(model.get("any", 0)), int(model.get("both_ori", 0)))
    s5 = np.zeros(n + 1, dtype=np.int16); s3 = np.zeros(n + 1, dtype=np.int16)
    fs = np.float32(model["fs"])
    for pos in range(left if lo is None else max(lo, left), right if hi is None else min(hi, right)):
        v5 = np.float32(fs * scan(model["pm5"], x, pos))
        v3 = np.float32(fs * scan(model["pm3"], x, pos))
        s5[pos] = int(v5) + int(model["tab"][d5[pos]])          # (STYPE) truncates towards zero
        s3[pos] = int(v3) + int(model["tab"][16 + d3[pos]])
    return s5, s3
