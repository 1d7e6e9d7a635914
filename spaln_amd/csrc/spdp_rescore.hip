// spdp_rescore.hip -- skl_rngS_ng (src/fwd2s1.cc:446-693) on the device: the walk over a finished
// corner list that re-derives the total score ("S:" of the CLI), the alignment statistics (FSTAT) and
// the per-exon records (EISCR: boundaries, exon / intron scores, splice signals, match / mismatch /
// gap counts in the exon and within `jneibr` positions of its junctions).  One thread per query: the
// work is O(alignment length) and the inputs (residues, per-position signals) are already resident.
// The output-format side channels of the reference (Cigar / Vulgar / SAM strings) are not produced,
// and queries carrying an intron-position profile (PfqItr) are not supported.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

#define NEVSEL_I (INT32_MIN / 16 * 7)
#define MAX_JNEIBR 32

struct Fst { int mch, mmc, gap, unp; };

__global__ void spdp_rescore_s(RescoreArgs A)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= A.n_probs) return;
    const DevProblem P = A.probs[qi];
    const uint8_t* __restrict__ a = A.a_codes + P.a_off;          // a[m]: residue of row m + 1
    const int2* __restrict__ cols = A.cols + P.col_off;           // cols[n + 1].y: residue of column n + 1
    const uint8_t* __restrict__ aux = A.aux + 2 * P.col_off;      // aux[2 n + 1]: dinc5 << 4 | dinc3
    const DevScoring* __restrict__ sc = A.sc;
    const int2* __restrict__ skl = A.skl + A.skl_off[qi];
    int num = A.skl_cnt[qi];
    int* hdr = A.out_hdr + 8 * qi;
    int* rec_out = A.out_rec + (int64_t) A.rec_off[qi] * 21;
    int n_rec = 0;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const int jn = A.jneibr;

    auto gap_penalty = [&](int i) { return i == 0 ? 0 : (i > A.codonk1 ? A.lgop + i * A.lgep : A.gop + i * A.gep); };
    auto sig5_at = [&](int n) { return (int) (short) (cols[n].x & 0xffff) - A.ipen; };
    auto sig3_at = [&](int n) { return (int) (short) (cols[n].x >> 16); };
    auto spjscr = [&](int n5, int n3) {
        const int d = 16 * (aux[2 * n5 + 1] >> 4) + (aux[2 * n3 + 1] & 15);
        return (int) A.intpen[min(n3 - n5, A.intpen_len - 1)] + sig3_at(n3) + (int) A.t53[d];
    };

    int rb[21];                                                   // EISCR, src/gsinfo.h:262-284
    for (int i = 0; i < 21; ++i) rb[i] = 0;
    enum { LEFT = 0, RIGHT, RLEFT, RRIGHT, MCH, MMC, GAP, UNP, MCH5, MMC5, GAP5, UNP5, MCH3, MMC3, GAP3, UNP3, PHS,
           ESCR, ISCR, SIG3, SIG5 };
    Fst fst = {0, 0, 0, 0}, pst = {0, 0, 0, 0};
    int fval = 0;
    Fst que[MAX_JNEIBR];
    for (int i = 0; i < jn; ++i) que[i] = fst;
    int qp = 0;
    auto shift = [&](bool near) {                                 // Eijnc::shift, src/gsinfo.cc:1255
        if (near) {
            rb[MCH5] = fst.mch - que[qp].mch; rb[MMC5] = fst.mmc - que[qp].mmc;
            rb[UNP5] = fst.unp - que[qp].unp; rb[GAP5] = fst.gap - que[qp].gap;
        }
        que[qp] = fst;
        if (++qp == jn) qp = 0;
    };
    auto store = [&](const Fst& prv, bool near) {                 // Eijnc::store, :1237
        rb[MCH] = fst.mch - prv.mch; rb[MMC] = fst.mmc - prv.mmc; rb[GAP] = fst.gap - prv.gap; rb[UNP] = fst.unp - prv.unp;
        if (near) { rb[MCH5] = rb[MCH]; rb[MMC5] = rb[MMC]; rb[GAP5] = rb[GAP]; rb[UNP5] = rb[UNP]; }
        rb[MCH3] = fst.mch - que[qp].mch; rb[MMC3] = fst.mmc - que[qp].mmc;
        rb[UNP3] = fst.unp - que[qp].unp; rb[GAP3] = fst.gap - que[qp].gap;
    };
    auto push = [&]() { for (int i = 0; i < 21; ++i) rec_out[21 * n_rec + i] = rb[i]; ++n_rec; };

    int w = 0;
    if (num >= 2 && skl[1].y == skl[0].y && b_exgl) { ++w; --num; }
    int m = skl[w].x, n = skl[w].y;
    int ai = m, bi = n;
    int h = 0, ha = 0, hb = 0, s5 = 0, s3 = 0;
    int insert = 0, deletn = 0, intlen = 0, preint = 0, psp = 0;
    rb[LEFT] = n; rb[RLEFT] = m; rb[ISCR] = NEVSEL_I; rb[SIG3] = 0;
    while (--num > 0) {
        ++w;
        const int wm = skl[w].x, wn = skl[w].y;
        const int mi = wm - m;
        if (mi && insert) {
            const bool j = a_exgl && m == P.a_left;
            const int x = j ? 0 : gap_penalty(insert);
            int xi = NEVSEL_I;
            if (intlen) { insert -= intlen; xi = rb[ISCR] + gap_penalty(insert); }
            if (xi >= x) {                                        // intron
                hb = ha;
                if (rb[RIGHT] - rb[LEFT] > 0) push();
                rb[LEFT] = rb[RIGHT] + intlen;
                rb[RLEFT] = m;
                rb[SIG3] = s3;
                rb[ISCR] = NEVSEL_I;
                h += xi;
                insert -= preint;
            } else h += x;
            if (insert) insert = intlen = preint = 0;
        }
        const int ni = wn - n;
        if (ni && deletn) {
            if (!(b_exgl && n == P.b_left)) { h += gap_penalty(deletn); fst.gap += 1; }
            ai += deletn;
            deletn = 0;
        }
        int i = mi - ni;
        int d = (i >= 0) ? ni : mi;
        if (d) {
            m += d;
            int x = 0;
            for ( ; d; --d, ++ai, ++bi, ++n) {
                shift(psp++ == jn);
                const int ac = a[ai], bc = cols[bi + 1].y;
                x += sc->mtx[ac * 32 + bc];
                if (ac == bc) ++fst.mch; else ++fst.mmc;
            }
            h += x;
            fval += x;
        }
        if (i > 0) {
            deletn += i;
            for (int j = 0; j < i; ++j) { shift(psp++ == jn); ++fst.unp; }
        } else if (i < 0) {
            i = -i;
            const int n3 = n + i;
            int xi = NEVSEL_I;
            if (A.lsg && i > A.minl) {
                s5 = sig5_at(n);
                s3 = sig3_at(n3);
                xi = s5 + spjscr(n, n3);
            }
            if (xi > gap_penalty(i) && xi > rb[ISCR]) {           // intron
                preint = insert;
                intlen = i;
                rb[RIGHT] = n; rb[RRIGHT] = m; rb[ISCR] = xi;
                rb[ESCR] = h + s5 - hb;
                rb[SIG5] = s5;
                ha = h + xi - s3;
                store(pst, psp < jn);
                pst = fst;
                psp = 0;
            } else if (!a_exgl || m != P.a_left) {
                if (!insert) fst.gap += 1;
                for (int j = 0; j < i; ++j, ++n) { shift(psp++ == jn); ++fst.unp; }
            }
            bi += i;
            insert += i;
        }
        m = wm; n = wn;
    }
    if (insert && !(a_exgr && m == P.a_right)) { h += gap_penalty(insert); fst.gap += 1; fst.unp += insert; }
    if (deletn && !(b_exgr && n == P.b_right)) { h += gap_penalty(deletn); fst.gap += 1; fst.unp += deletn; }
    rb[ESCR] = h - hb; rb[ISCR] = 0; rb[SIG5] = 0; rb[RIGHT] = n; rb[RRIGHT] = m;
    store(pst, n - rb[LEFT] <= jn);
    push();
    rb[LEFT] = rb[RIGHT] = INT32_MAX;                            // endrng, src/cmn.h:140
    push();
    fval += A.gop * fst.gap + A.gep * fst.unp;
    hdr[0] = h; hdr[1] = fst.mch; hdr[2] = fst.mmc; hdr[3] = fst.gap; hdr[4] = fst.unp; hdr[5] = fval;
    hdr[6] = n_rec; hdr[7] = 0;
}

extern "C" hipError_t spdp_launch_rescore(const RescoreArgs* a, hipStream_t stream)
{
    RescoreArgs A = *a;
    hipLaunchKernelGGL(spdp_rescore_s, dim3((A.n_probs + 63) / 64), dim3(64), 0, stream, A);
    return hipGetLastError();
}
