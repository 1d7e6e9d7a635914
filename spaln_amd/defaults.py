"""Default scoring bundle of the cDNA x genome path, as the reference CLI sets it up
(setdefparam, src/spaln.cc:1471-1494; PwdB, src/aln2.cc:99-137; IntronPenalty,
src/codepot.cc:126-233 with the generic table/ parameter set).  These are DATA
read back from a reference run (the same values every tests/golden/*.spdg
fixture carries in its `params`, `mtx`, `qm_*` records), not code.
"""
import numpy as np

from . import abi

# residue codes of the reference's nucleotide alphabet (src/cmn.h:114): A=2 C=3 G=5 T=9, N=16
CODE_OF = {ord("A"): 2, ord("C"): 3, ord("G"): 5, ord("T"): 9, ord("N"): 16}
NSIMD = 17

# Simmtx::Nmtx for setNpam(4, -6), alprm.scale = 10: match +20, mismatch -60, gap column -20
NMTX = np.array([
    [   0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0],
    [   0,    0,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20,  -20],
    [   0,  -20,   20,  -60,    0,  -60,    0,  -60,  -30,  -60,    0,  -60,  -30,  -60,  -30,  -60,  -30],
    [   0,  -20,  -60,   20,    0,  -60,  -60,    0,  -30,  -60,  -60,    0,  -30,  -60,  -60,  -30,  -30],
    [   0,  -20,    0,    0,    0,  -60,  -30,  -30,  -30,  -60,  -30,  -30,  -30,  -60,  -60,  -60,  -30],
    [   0,  -20,  -60,  -60,  -60,   20,    0,    0,  -30,  -60,  -60,  -60,  -60,    0,  -30,  -30,  -30],
    [   0,  -20,    0,  -60,  -30,    0,    0,  -30,  -30,  -60,  -30,  -60,  -60,  -30,  -30,  -60,  -30],
    [   0,  -20,  -60,    0,  -30,    0,  -30,    0,  -30,  -60,  -60,  -30,  -60,  -30,  -60,  -30,  -30],
    [   0,  -20,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -60,  -60,  -60,  -30,  -60,  -30,  -30,  -30],
    [   0,  -20,  -60,  -60,  -60,  -60,  -60,  -60,  -60,   20,    0,    0,  -30,    0,  -30,  -30,  -30],
    [   0,  -20,    0,  -60,  -30,  -60,  -30,  -60,  -60,    0,    0,  -30,  -30,  -30,  -30,  -60,  -30],
    [   0,  -20,  -60,    0,  -30,  -60,  -60,  -30,  -60,    0,  -30,    0,  -30,  -30,  -60,  -30,  -30],
    [   0,  -20,  -30,  -30,  -30,  -60,  -60,  -60,  -30,  -30,  -30,  -30,  -30,  -60,  -30,  -30,  -30],
    [   0,  -20,  -60,  -60,  -60,    0,  -30,  -30,  -60,    0,  -30,  -30,  -60,    0,  -30,  -30,  -30],
    [   0,  -20,  -30,  -60,  -60,  -30,  -30,  -60,  -30,  -30,  -30,  -60,  -30,  -30,  -30,  -30,  -30],
    [   0,  -20,  -60,  -30,  -60,  -30,  -60,  -30,  -30,  -30,  -60,  -30,  -30,  -30,  -30,  -30,  -30],
    [   0,  -20,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30,  -30]
], dtype=np.int32)

GOP, GEP = -60, -20              # PwdB::BasicGOP / BasicGEP  (v = 6, u = 2, Vab = 10)
LGOP, LGEP = -158, -6
LLMT = 20                       # IntronPrm.llmt
IPEN = -245                     # IntronPenalty::Penalty() = GapWI
QM_LEN = [73, 136, 317, 959, 1523]    # IntronPenalty::qm[].len (5 equi-probable quantiles)
QM_PEN = [-230, -229, -251, -273, -323]
SH = 100                        # alprm.sh band shoulder
MAX_VMF_SPACE = 32 * 1024 * 1024


def exact_tables():
    """IntPen(length) of the default intron-length distribution, materialised up to 29 450 nt (longer introns
    price like the last entry), and the junction table behind Exinon::sig53(.., IE53): both read back from a
    reference run (the c5_6kb fixture).  What SpdpScoring.intpen / t53 carry for the -A0 / -A1 engines."""
    import os
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "exact_tables.npz"))
    return t["intpen"], t["t53"]


def scoring(nquant=None, **over) -> abi.Scoring:
    kw = dict(mtx=NMTX, mtx_dim=NSIMD, gop=GOP, gep=GEP, lgop=LGOP, lgep=LGEP, noll=2, spj=1,
              llmt=LLMT, ipen=IPEN, qm_len=QM_LEN, qm_pen=QM_PEN, nquant=nquant, local=0, sh=SH,
              max_vmf_space=MAX_VMF_SPACE, ubh=0)
    kw.update(over)
    return abi.make_scoring(**kw)


def encode(seq_ascii: np.ndarray) -> np.ndarray:
    """ASCII nucleotides -> reference residue codes."""
    lut = np.full(256, 16, dtype=np.uint8)
    for k, v in CODE_OF.items():
        lut[k] = v
        lut[k + 32] = v
    return lut[np.asarray(seq_ascii, dtype=np.uint8)]


# ---- aa x genome path (protein query vs genomic DNA): PwdB set-up for DvsP = 1 -------------------
# values read back from a reference run (tests/golden/h1_*.spdg carry the same numbers); the
# 23 x 26 amino-acid x tron matrix (Simmtx of mdm_mtx, PAM level 150) is data/aa_tron_mtx.npy
H_GOP, H_GEP, H_LGEP = -90, -20, -6
H_CODONK1 = 1610612733                      # effectively "never": GapExtPen3 = BasicGEP
H_GAPW1, H_GAPW2, H_GAPW3 = -410, -430, -110
H_IPEN = -401
H_QM_PEN = [-367, -370, -413, -456, -557]


def scoring_h(nquant=None, **over) -> abi.ScoringH:
    import os
    mtx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "aa_tron_mtx.npy"))
    kw = dict(mtx=mtx.astype(np.int32), mtx_rows=mtx.shape[0], mtx_cols=mtx.shape[1], gop=H_GOP, gep=H_GEP,
              lgep=H_LGEP, codonk1=H_CODONK1, gapw1=H_GAPW1, gapw2=H_GAPW2, gapw3=H_GAPW3, spj=1,
              llmt=LLMT, ipen=H_IPEN, qm_len=QM_LEN, qm_pen=H_QM_PEN, nquant=nquant, local=0,
              term_codon=1, sh=SH, max_vmf_space=MAX_VMF_SPACE, ubh=0)
    kw.update(over)
    return abi.make_scoring_h(**kw)


# MakeBlk::prepacomp's per-class terms of the translated block index's word scores (src/blksrc.cc:844-877) for the reference's defaults
# (twenty classes, -Xp20, -Xq1, its mdm tables) -- recorded from the compiled reference (`spaln_idxtap -W -KP`, oracle/ref_build/idx_tap.cc:
# the "[idx_tap] acomp" line, exact hex floats); spdp_blk_index_build_p's acomp
BLOCK_ACOMP_20 = [float.fromhex(x) for x in (
    "-0x1.905a039bd2496p+2 -0x1.20cd81c8d9c8p+1 -0x1.ea713dcbdeac6p+2 -0x1.d7828e58105d8p+0 0x1.018bdf0832daep+3 "
    "0x1.9dac9b7222838p+0 -0x1.2fef0207cbap+1 0x1.44fd3d196c1cfp+3 -0x1.4611967298f91p+2 -0x1.7045848bd36fcp+2 "
    "0x1.157df2a1998dep+2 0x1.b29309c2da184p-3 -0x1.9e5b95c02c5ccp+1 0x1.aeb2e0dfc9daep+2 0x1.8ffa6b1bb2d76p+1 "
    "-0x1.785b7311540c6p+3 -0x1.f7373e897ef4ap+2 0x1.7b75b1ad70547p+3 0x1.3672571d50d5bp+3 -0x1.ae4b8287ac8dp+0").split()]
