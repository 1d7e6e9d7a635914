// spdp_h_dev.h -- device-side layout of the aa x genome (Fwd2h1 `_wip`) path.
//
// HBM layout of one batch (offsets in elements of the named array):
//   a_codes  uint8   concatenated amino-acid codes, row m of problem p is a_codes[a_off + m - 1]
//   cols     int4    one record per genomic position n of a parent window, n in [0, col_len):
//                      x = (uint16) cp | tron << 16 | flags << 24
//                            cp    = good(n-2) ? sigE[n-2] : 0           (fwd2h1_wip_simd.h:124)
//                            tron  = b[n-2], the codon ending at n       (:188-190)
//                            flags = acc0 (2 bits: 0 none, 1 ACCM, 2 ACCZ, 3 ACCP) | acc1 << 2
//                                    | don0 << 3 (2 bits, DONM/DONZ/DONP) | don1 << 5
//                      y = (uint16) s3_0 | s3_1 << 16      acceptor signals of the 2 candidates (:214-222)
//                      z = (uint16) s5_0 | s5_1 << 16      donor signals + ipen           (:262-270)
//                      w = dinc5 << 4 | dinc3 of the position (scalar engine / rescoring; 0 if not given)
//                    The kernel applies the window itself (nothing splices at n >= b_right, no
//                    substitution score outside [b_left + 3, b_right + 2]).
//   aux      short4  {sigS, sigT, sigE, sig5} raw per position: boundary set-up / end selection
//   bnd      int2    {H, F} by diagonal r = n - 3m: entry r - lw + 3 (hv / fv of fwd2h1_simd.h:335-338)
//   bnd (linear-space engine) int4 {H, F, Hlink, Flink}
//   tb       uint16  traceback codes in the reference's own Anti_rhomb_coord<SHORT> layout, step 3:
//                    cell (m, n) at ((3 (m - a_left) + n - b_left) * m_width + m - a_left)
#ifndef SPDP_H_DEV_H_
#define SPDP_H_DEV_H_

#include <stdint.h>

#define SPDH_NELEM     16
#define SPDH_LAG       8           // blocks of 16 steps between consecutive stripes of a pass
#define SPDH_BND_PAD   32          // boundary entries after width + 6 * nelem (prefetch over-read)
#define SPDH_COL_PAD   64          // zero column records after b_len + 3
#define SPDH_NEV       (-32768 + 1024)
#define SPDH_MIN_SSV   (-1000)
#define SPDH_ZCODE     31          // mtx row / column forced to zero

struct DevScoringH {
    int32_t gop, gep, lgep, codonk1;
    int32_t g1, g2, g3;
    int32_t spj, llmt, nquant, local, term_codon;
    int32_t qm_len[8];
    int32_t qm_pen[8];
    int32_t mtx[32 * 32];          // stride 32: mtx[aa * 32 + tron]; row 31 and column 31 are zero
};

struct DevProblemH {
    int32_t a_left, a_right, b_left, b_right;
    int32_t lw, up, width, buf_size;
    int32_t flags;                 // bit0 a_exgl&1, bit1 a_exgr&1 ... see spdp_h_api.cpp: raw 2-bit values
    int32_t a_exgl, a_exgr, b_exgl, b_exgr;
    int32_t m_width, n_width;
    int32_t col_len;
    int32_t n_im;                  // linear-space engine: number of intermediate rows
    int32_t a_len, b_len;          // parent sequence lengths (scalar engine: positions beyond read as padding)
    int32_t imd_intvl;             // scalar linear-space engine: rows between intermediates (Aln2h1::imd_intvl)
    int32_t cip_off;               // -A0 / -A1 engines: first entry of the query's cip row in HScalarArgs::cip, -1 = none
    int32_t a_pad;                 // SpdpProblemH::a_pad
    int32_t nospj;                 // scalar engine: no introns (trcbkalignH_ng(wdw, false))
    int32_t cut_l, cut_len;        // scalar forward engine: the sweep jumps over columns (cut_l, cut_l + cut_len]; `width` is
    int32_t pad_;                  //   the window's width minus cut_len (forwardH_ng's cutrng, src/fwd2h1.cc:308-312, 589-603)
    int64_t a_off;
    int64_t col_off;               // into cols / aux
    int64_t bnd_off;               // into bnd (entries)
    int64_t tb_off;                // into tb (uint16 elements)
    int64_t tb_size;               // m_width * n_width + 32, the reference's allocation
    int64_t imd_off;               // linear-space engine: into imd (ints), 5 * width per intermediate row:
                                   //   hlnk[2][width], vlnk[2][width], event[width], index r - lw + 1
    int64_t cells;
};

struct DevResultH {
    int32_t score;                 // what forwardH1_wip returns (nevsel unless a local end was tracked)
    int32_t mr, nr;                // start cell of the traceback
    int32_t maxt, maxr;            // fhlastH1's choice
    int32_t pad[3];                // pad[0]: linear-space engine: link of the end cell (maxh.ulk)
};

#endif
