// spdp_dev.h -- device-side data layout shared by the HIP kernels and the host API.
//
// HBM layout of one batch (all offsets in elements of the named array):
//   a_codes  uint8   concatenated query residues, problem p at [a_off, a_off + a_len)
//   cols     int2    per-genome-position records {sigpack, base} of a whole parent sequence:
//                    position n is cols[col_off + n], n in [0, b_len + COL_PAD);
//                    sigpack = (uint16)(sig5[n] + ipen) | sig3[n] << 16, base = b[n-1].
//                    The kernel applies the window itself: records beyond b_right read as
//                    zero (the reference feeds 0 there, fwd2s1_wip_simd.h:159,194) and the
//                    residue at n <= b_left is code 0 (mtx row/column 0 is all zero,
//                    simmtx.cc:160-164) -- so sub-problems (UDH slabs) reuse the arrays
//   bnd      int2/4  stripe-boundary rows by diagonal: entry (r - lw + 1) holds
//                    {H, F} (score / forward) or {H, F, Hlink, Flink} (UDH) of the
//                    reference's hv/fv(/hc/fc) arrays (fwd2s1_simd.h:129-132), in place
#ifndef SPDP_DEV_H_
#define SPDP_DEV_H_

#include <stdint.h>

#define SPDP_NELEM     16          // rows per reference stripe (AVX2 int16 lanes)
#define SPDP_COL_PAD   64          // zero column records appended after b_right
#define SPDP_BND_PAD   48          // boundary entries appended after width + 2 * nelem
#define SPDP_NEV16     (-32768 + 1024)
#define SPDP_FLOOR16   (-32768)
#define SPDP_GROUP_LAG 4           // blocks of 16 steps between consecutive stripes of a pass

struct DevScoring {
    int32_t mtx_dim;
    int32_t gop, gep;
    int32_t spj, llmt, nquant, local;
    int32_t qm_len[8];
    int32_t qm_pen[8];
    int32_t mtx[32 * 32];          // stride 32, row / column 0 forced to zero
};

struct DevProblem {
    int32_t a_left, a_right, b_left, b_right;
    int32_t lw, up, width, buf_size;
    int32_t flags;                 // bit0 a_exgl, bit1 a_exgr, bit2 b_exgl, bit3 b_exgr
    int32_t n_im;                  // UDH: number of intermediate rows
    int32_t imd_intvl;             // scalar UDH: rows between intermediates (Aln2s1::imd_intvl)
    int32_t cip_off;               // exact engines: first entry of the query's cip row in ScalarArgs::cip, -1 = none
    int32_t cut_l, cut_len;        // forwardS_ng with a cut range: the sweep jumps from column cut_l over cut_len columns
    int64_t a_off;                 // into a_codes; residue of row m is a_codes[a_off + m - 1]
    int64_t col_off;               // into cols
    int64_t bnd_off;               // into bnd (entries)
    int64_t tb_off;                // forward: into the traceback-code buffer (bytes)
    int64_t imd_off;               // UDH: into the intermediate-link buffer (ints)
    int64_t cells;                 // DP cells of this call (reference loop bounds)
};

struct DevResult {
    int32_t score;                 // maxh.val
    int32_t mr, nr;                // end cell
    int32_t ml, ulk;               // UDH: left end row, link of the end cell
    int32_t maxr;                  // diagonal chosen by fhlastS1
    int32_t pad[2];
};

#endif
