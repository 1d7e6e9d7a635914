// spdp_rescore.hip -- skl_rngS_ng (src/fwd2s1.cc:446-693) on the device: the walk over a finished
// corner list that re-derives the total score ("S:" of the CLI), the alignment statistics (FSTAT) and
// the per-exon records (EISCR: boundaries, exon / intron scores, splice signals, match / mismatch /
// gap counts in the exon and within `jneibr` positions of its junctions).  One thread per query: the
// work is O(alignment length) and the inputs (residues, per-position signals) are already resident.
// With RescoreArgs::ops_format set, the walk also writes the edit records the reference collects for its Cigar /
// Vulgar / SAM writers (Cigar::push, Vulgar::push, Samfmt, src/fwd2s1.cc:469-475, 509-689) -- one format per launch, as
// algmode.nsa selects one in the reference.  Queries carrying an intron-position profile (PfqItr) are not supported.

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
    // edit records: format 1 Cigar {ope, len}, 2 Vulgar {ope, alen, blen}, 3 SAM {ope, len}
    const int fmt = A.ops_format;
    int3* ops = fmt ? A.ops + A.ops_off[qi] : nullptr;
    const int ops_cap = fmt ? (int) (A.ops_off[qi + 1] - A.ops_off[qi]) : 0;
    int n_ops = 0;
    auto op = [&](int f, int ope, int x, int y) { if (fmt == f) { if (n_ops < ops_cap) ops[n_ops] = make_int3(ope, x, y); ++n_ops; } };

    int w = 0;
    if (num >= 2 && skl[1].y == skl[0].y && b_exgl) { ++w; --num; }
    int m = skl[w].x, n = skl[w].y;
    if (m) op(3, 'H', m, 0);                                      // SAM: local alignment, clipped query start
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
                if (preint) { op(1, 'D', preint, 0); op(3, 'D', preint, 0); op(2, 'G', 0, preint); }
                op(1, 'N', intlen, 0); op(3, 'N', intlen, 0);
                op(2, '5', 0, 2); op(2, 'I', 0, intlen - 4); op(2, '3', 0, 2);
                hb = ha;
                if (rb[RIGHT] - rb[LEFT] > 0) push();
                rb[LEFT] = rb[RIGHT] + intlen;
                rb[RLEFT] = m;
                rb[SIG3] = s3;
                rb[ISCR] = NEVSEL_I;
                h += xi;
                insert -= preint;
            } else h += x;
            if (insert) {
                op(1, 'D', insert, 0); op(3, 'D', insert, 0); op(2, 'G', 0, insert);
                insert = intlen = preint = 0;
            }
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
            op(1, 'M', d, 0); op(2, 'M', d, d);
            int x = 0, run = 0;                                   // SAM: the running '=' / 'X' stretch is pushed at every base
            for ( ; d; --d, ++ai, ++bi, ++n) {
                shift(psp++ == jn);
                const int ac = a[ai], bc = cols[bi + 1].y;
                x += sc->mtx[ac * 32 + bc];
                if (ac == bc) { if (run < 0) { op(3, 'X', -run, 0); run = 0; } ++fst.mch; ++run; }
                else { if (run > 0) { op(3, '=', run, 0); run = 0; } ++fst.mmc; --run; }
                if (run > 0) op(3, '=', run, 0); else if (run < 0) op(3, 'X', -run, 0);
            }
            h += x;
            fval += x;
        }
        if (i > 0) {
            deletn += i;
            op(1, 'I', i, 0); op(3, 'I', i, 0); op(2, 'G', i, 0);
            for (int j = 0; j < i; ++j) { shift(psp++ == jn); ++fst.unp; }
        } else if (i < 0) {
            i = -i;
            const int n3 = n + i;
            int xi = NEVSEL_I;
            if (A.lsg && i > A.minl) {
                s5 = sig5_at(n);
                s3 = sig3_at(n3);
                xi = s5 + spjscr(n, n3);
                // use_spb(): an annotated intron position of the query (PfqItr::match_score(m) = Cip_score::cip_score(m))
                if (A.cip && P.cip_off >= 0 && m >= 0 && m <= A.a_len_all[qi]) xi += A.cip[P.cip_off + m];
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
    if (insert && !(a_exgr && m == P.a_right)) {
        h += gap_penalty(insert); fst.gap += 1; fst.unp += insert;
        op(1, 'D', insert, 0); op(3, 'D', insert, 0); op(2, 'G', 0, insert);
    }
    if (deletn && !(b_exgr && n == P.b_right)) {
        h += gap_penalty(deletn); fst.gap += 1; fst.unp += deletn;
        op(1, 'I', deletn, 0); op(3, 'I', deletn, 0); op(2, 'G', deletn, 0);
    }
    if (fmt == 3 && m < A.a_len[qi]) op(3, 'H', A.a_len[qi] - m, 0);     // clipped query end
    if (fmt) A.ops_cnt[qi] = n_ops;
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

// ======================================================================================
// skl_rngH_ng (src/fwd2h1.cc:635-940): the protein x genome walk.  Codon-diagonal runs add sim2 and the
// coding potential; a deletion of >= minl nucleotides is tried as an intron in the phase(s) the splice
// flags allow, with the codon an intron splits re-scored from the two exon halves (spjseq); gaps that
// are not multiples of three become frame-shift records.  One thread per query.
// ======================================================================================

__global__ void spdh_rescore(HRescoreArgs A)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= A.n_probs) return;
    const HRescoreProb P = A.probs[qi];
    const uint8_t* __restrict__ a = A.a_codes + P.a_off;
    const uint8_t* __restrict__ bq = A.b_codes + P.b_off;
    const short* __restrict__ sg = A.sig + 5 * P.col_off;
    const int8_t* __restrict__ ph = A.phs + 2 * P.col_off;
    const uint8_t* __restrict__ dc = A.dinc + P.col_off;
    const int N = P.b_len + 3;
    int2* skl = const_cast<int2*>(A.skl) + A.skl_off[qi];
    int num = A.skl_cnt[qi];
    int* hdr = A.out_hdr + 8 * qi;
    int* rec_out = A.out_rec + (int64_t) A.rec_off[qi] * 21;
    int n_rec = 0;
    const int jn = A.jneibr;
    const int AMBc = 2, SERc = 18, SER2c = 23, TRM2c = 24, TRMc = 25;

    auto aat = [&](int i) -> int { return (i >= 0 && i < P.a_len) ? a[i] : AMBc; };
    // the launch holds positions [w_lo, w_hi) of the region (what the corners span, and a margin); a read outside them that lies inside
    // the region is reported (hdr[7]) and refused by the host: nothing is ever made up
    int missed = 0;
    auto held = [&](int i) -> bool { const bool in = i >= P.w_lo && i < P.w_hi; if (!in) missed = 1; return in; };
    auto bat = [&](int i) -> int { return (i >= 0 && i <= P.b_len && held(i)) ? bq[i - P.w_lo] : AMBc; };
    auto sgat = [&](int i, int f) -> int { return (i >= 0 && i < N && held(i)) ? sg[5 * (i - P.w_lo) + f] : 0; };     // 0 sig5 1 sig3 2 sigS 3 sigT 4 sigE
    auto phat = [&](int i, int f) -> int { return (i >= 0 && i < N && held(i)) ? ph[2 * (i - P.w_lo) + f] : -2; };
    auto dcat = [&](int i) -> int { return (i >= 0 && i <= P.b_len && held(i)) ? dc[i - P.w_lo] : 0; };
    auto cdiv = [](int x, int y) { return x / y; };                                            // C division
    auto gap_penalty3 = [&](int i) {
        if (i == 0) return 0;
        const int d = i / 3;
        const int x = (i % 3 == 1) ? A.gape1 : ((i % 3 == 2) ? A.gape2 : 0);
        return x + ((i > A.codonk1) ? cdiv(A.lgop * A.gop, A.gop) + d * A.lgep : A.gop + d * A.gep);
    };
    auto unp_penalty3 = [&](int i) {
        const int d = i / 3;
        const int unp = d * A.gep;
        const int egop = (i % 3 == 1) ? A.gape1 : ((i % 3 == 2) ? A.gape2 : 0);
        return (i <= A.codonk1) ? unp + egop : unp - A.diffu * (d - A.k1) + egop;
    };
    auto sig53_ie53 = [&](int n5, int n3) {
        const int d5 = dcat(n5) >> 4, d3 = dcat(n3) & 15;
        return sgat(n3, 1) + (int) A.t53[16 * d5 + d3];
    };
    auto spjscr = [&](int n5, int n3) { return (int) A.intpen[min(max(n3 - n5, 0), A.intpen_len - 1)] + sig53_ie53(n5, n3); };
    auto avst_equal = [&](int ar, int br) { return ar == br || (br == SER2c && ar == SERc); };
    auto is_term = [&](int x) { return x == TRMc || x == TRM2c; };
    // SpJunc::spjseq: the two codons a split codon can spell, as tron codes packed lo | hi << 8
    auto spjseq = [&](int n5, int n3) -> int {
        if (n5 < P.b_left || n3 >= P.b_right) return AMBc | (AMBc << 8);
        const int c0 = A.mid[bat(n5 - 2) & 31], c1 = A.mid[bat(n5 - 1) & 31], c2 = A.mid[bat(n3) & 31], c3 = A.mid[bat(n3 + 1) & 31];
        if ((c1 | c2) >= 4) return AMBc | (AMBc << 8);         // a codon is defined when its own three bases are
        return (c0 < 4 ? A.tron_of[16 * c0 + 4 * c1 + c2] : AMBc) | ((c3 < 4 ? A.tron_of[16 * c1 + 4 * c2 + c3] : AMBc) << 8);
    };

    int rb[21];
    for (int i = 0; i < 21; ++i) rb[i] = 0;
    enum { LEFT = 0, RIGHT, RLEFT, RRIGHT, MCH, MMC, GAP, UNP, MCH5, MMC5, GAP5, UNP5, MCH3, MMC3, GAP3, UNP3, PHS,
           ESCR, ISCR, SIG3, SIG5 };
    Fst fst = {0, 0, 0, 0}, pst = {0, 0, 0, 0};
    int fval = 0;
    Fst que[MAX_JNEIBR];
    for (int i = 0; i < jn; ++i) que[i] = fst;
    int qp = 0;
    auto shift = [&](bool near) {
        if (near) {
            rb[MCH5] = fst.mch - que[qp].mch; rb[MMC5] = fst.mmc - que[qp].mmc;
            rb[UNP5] = fst.unp - que[qp].unp; rb[GAP5] = fst.gap - que[qp].gap;
        }
        que[qp] = fst;
        if (++qp == jn) qp = 0;
    };
    auto store = [&](const Fst& prv, bool near) {
        rb[MCH] = fst.mch - prv.mch; rb[MMC] = fst.mmc - prv.mmc; rb[GAP] = fst.gap - prv.gap; rb[UNP] = fst.unp - prv.unp;
        if (near) { rb[MCH5] = rb[MCH]; rb[MMC5] = rb[MMC]; rb[GAP5] = rb[GAP]; rb[UNP5] = rb[UNP]; }
        rb[MCH3] = fst.mch - que[qp].mch; rb[MMC3] = fst.mmc - que[qp].mmc;
        rb[UNP3] = fst.unp - que[qp].unp; rb[GAP3] = fst.gap - que[qp].gap;
    };
    auto push = [&]() { for (int i = 0; i < 21; ++i) rec_out[21 * n_rec + i] = rb[i]; ++n_rec; };
    // edit records (Cigar::push / Vulgar::push, src/fwd2h1.cc:695-924): format 1 Cigar {ope, len}, 2 Vulgar {ope, alen, blen}
    const int fmt = A.ops_format;
    int3* ops = fmt ? A.ops + A.ops_off[qi] : nullptr;
    const int ops_cap = fmt ? (int) (A.ops_off[qi + 1] - A.ops_off[qi]) : 0;
    int n_ops = 0;
    auto op = [&](int f, int ope, int x, int y) { if (fmt == f) { if (n_ops < ops_cap) ops[n_ops] = make_int3(ope, x, y); ++n_ops; } };

    if (A.sup_tcodon) {
        const int cs0 = bat(skl[num - 1].y - 2);
        if (is_term(cs0)) skl[num - 1].y -= 3;          // private copy of the corner list
    }
    int w = 0;
    if (num >= 2 && skl[1].y == skl[0].y && P.b_exgl) { ++w; --num; }
    int m = skl[w].x, n = skl[w].y;
    int ai = m, bi = n, bbn = n;
    int cs = -1;                                        // < 0: none
    int h = 0, hi = NEVSEL_I, ha = 0, hb = 0, hvl = 0;
    bool ivl = false;
    int ngop = 0, s5 = 0, s3 = 0;
    int insert = 0, deletn = 0, intlen = 0, preint = 0, phs = 0, psp = 0;
    if ((A.lcl & 17) && sgat(bbn + 1, 2) > h) h = sgat(bbn + 1, 2);
    if ((A.lcl & 20) && sgat(bbn, 1) > h) h = sgat(bbn, 1);
    rb[LEFT] = n; rb[RLEFT] = m; rb[ISCR] = NEVSEL_I; rb[SIG3] = h;
    if (m) op(1, 'H', m, 0);                                       // local alignment
    for (;;) {
        --num;
        if (!(num > 0 || hi > NEVSEL_I)) break;
        if (num > 0) ++w;
        const int wm = skl[w].x, wn = skl[w].y;
        const bool term = num == 1;
        const int mi = (wm - m) * 3;
        if (insert && (mi || (h > NEVSEL_I && hi > NEVSEL_I) || term)) {
            const bool termgap = (P.a_exgl && m == P.a_left) || (P.a_exgr && m == P.a_right);
            h += termgap ? unp_penalty3(insert) : gap_penalty3(insert);
            if (hi > NEVSEL_I && insert > intlen) hi += gap_penalty3(insert - intlen);
            if (hi > NEVSEL_I && hi >= h) {              // intron
                if (preint) { op(1, 'D', preint, 0); op(2, 'G', 0, preint); }
                op(1, 'N', intlen, 0);
                // (`phs` is what the last frame shift left in it, not the intron's phase: as in the reference)
                if (phs == -1) op(2, 'S', 0, 1); else if (phs == 1) op(2, 'S', 0, 2);
                op(2, '5', 0, 2); op(2, 'I', 0, intlen - 4); op(2, '3', 0, 2);
                if (phs == -1) op(2, 'S', 1, 2); else if (phs == 1) op(2, 'S', 1, 1);
                hb = ha;
                if (rb[RIGHT] - rb[LEFT] > 1) push();
                rb[LEFT] = rb[RIGHT] + intlen;
                rb[RLEFT] = m;
                rb[SIG3] = s3;
                h = hi;
                insert -= preint + intlen;
            }
            hi = NEVSEL_I;
            if (insert) {                               // post-intron gap
                if (term && is_term(bat(bi - 1))) insert -= 3;
                op(1, 'D', insert, 0);
                phs = insert % 3;
                insert -= phs;
                if (!((P.a_exgl && m == P.a_left) || (P.a_exgr && m == P.a_right))) fst.gap += ngop;
                for (int j = 0; j < insert; j += 3, psp += 3) { shift(psp / 3 == jn); fst.unp += 3; }
                if (phs) {                              // insertion frame shift
                    rb[RIGHT] = n - phs; rb[RRIGHT] = m; rb[ISCR] = NEVSEL_I;
                    push();
                    op(2, 'F', 0, phs);
                    rb[LEFT] = n; rb[RLEFT] = m;
                    h += (phs == 1) ? A.gape1 : A.gape2;
                    fval += (phs == 1) ? A.gape1 : A.gape2;
                }
                if (insert) op(2, 'G', 0, insert);
                ngop = insert = intlen = preint = 0;
            }
        }
        const int ni = wn - n;
        if (ni && deletn) {
            if (!(P.b_exgl && n == P.b_left)) { h += gap_penalty3(deletn); fst.gap += 1; }
            ai += deletn / 3;
            if ((phs = deletn % 3)) {                   // deletion frame shift
                rb[RIGHT] = n + phs; rb[RRIGHT] = m; rb[ISCR] = NEVSEL_I;
                push();
                op(2, 'F', phs, 0);
                rb[LEFT] = n; rb[RLEFT] = m;
                h += A.extragop;
                fval += A.extragop;
                ++ai;
                deletn -= phs;
                phs = 3 - phs;
                bi += phs;
                bbn += phs;
            }
            if (deletn > 2) op(2, 'G', deletn / 3, 0);
            deletn = 0;
        }
        int i = mi - ni;
        int d = (i >= 0) ? ni : mi;
        if (d) {
            op(1, 'M', d, 0); op(2, 'M', d / 3, d);
            n += d;
            m += d / 3;
            for ( ; d > 2; d -= 3, ++ai, bi += 3, bbn += 3, psp += 3) {
                shift(psp / 3 == jn);
                const int gs = (cs >= 0) ? (cs >> 8) : bat(bi + 1);
                hvl = A.mtx[aat(ai) * 32 + (gs & 31)];
                fval += hvl;
                hvl += (cs >= 0) ? 0 : sgat(bbn + 1, 4);
                h += hvl;
                ivl = avst_equal(aat(ai), gs);
                if (ivl) ++fst.mch; else ++fst.mmc;
                cs = -1;
            }
        }
        if (i > 0) {
            cs = -1;
            deletn += i;
            op(1, 'I', i, 0);
            for (int j = 0; j < i; j += 3, psp += 3) { shift(psp / 3 == jn); fst.unp += 3; }
        } else if (i < 0) {
            i = -i;
            const int b3n = bbn + i;
            if (hi <= NEVSEL_I && i >= A.minl && wn < P.b_right) {        // intron?
                int cm = -1, sig5m = 0;
                int ph5 = (phat(bbn, 0) == 2) ? phat(b3n, 1) : phat(bbn, 0);
                int ph3 = (phat(b3n, 1) == 2) ? phat(bbn, 0) : phat(b3n, 1);
                int xm = NEVSEL_I, xi = NEVSEL_I;
                int nb, n3;
                if (ph3 == 2 && ph5 == 2) {              // GTGT....AGAG
                    nb = n + 1; n3 = nb + i;
                    sig5m = sgat(nb, 0);
                    xm = sig5m + spjscr(nb, n3);
                    cm = spjseq(nb, n3);
                    ph3 = ph5 = 1;
                }
                nb = n - ph3; n3 = nb + i;
                if (ph5 == ph3 && ph5 > -2) {            // isJunct
                    s5 = sgat(nb, 0);
                    s3 = sig53_ie53(nb, n3);
                    xi = s5 + spjscr(nb, n3);
                    cs = spjseq(nb, n3);
                    preint = insert;
                    if (ph3 == 0) cs = -1;
                    if (insert == 0 && ph3 == 1) {
                        const int c0 = cs & 0xff;
                        const int hdlt = A.mtx[aat(ai - 1) * 32 + (c0 & 31)] - hvl;
                        xi += hdlt;
                        fval += hdlt;
                        const bool match = avst_equal(aat(ai - 1), c0);
                        if (match && !ivl) { ++fst.mch; --fst.mmc; }
                        else if (!match && ivl) { --fst.mch; ++fst.mmc; }
                    }
                }
                if (xm > xi) {
                    xi = xm;
                    ph3 = -1;
                    nb = n - ph3; n3 = nb + i;
                    s5 = sig5m;
                    s3 = sig53_ie53(nb, n3);
                    cs = cm;
                }
                if (xi > NEVSEL_I) {
                    if (ph3 != -1) cs = -1;
                    hi = h + xi;
                    intlen = i;
                    rb[RIGHT] = nb; rb[RRIGHT] = m; rb[PHS] = ph3;
                    rb[ISCR] = xi; rb[SIG5] = s5;
                    rb[ESCR] = h + gap_penalty3(insert);
                    ha = rb[ESCR] + xi - s3;
                    rb[ESCR] += s5 - hb;
                    store(pst, psp < jn);
                    pst = fst;
                    psp = 0;
                }
            } else if (!(term && is_term(bat(bi + 1)))) {
                int y = 0;                              // SumCodePot(bb, i, 0, pwd), :619-633
                for (int k = i, pos = bbn + 1; k > 0; k -= 3, pos += 3) y += sgat(pos, 4);
                h += y;
                if (hi <= NEVSEL_I) ++ngop;
            }
            bbn = b3n;
            bi += i;
            insert += i;
        }
        m = wm; n = wn;
    }
    s5 = 0;
    if (n > 1) {
        if ((A.lcl & 18) && sgat(bbn - 2, 3) > 0) s5 = sgat(bbn - 2, 3);
        if ((A.lcl & 24) && sgat(bbn, 0) > 0 && sgat(bbn, 0) > sgat(bbn - 2, 3)) s5 = sgat(bbn, 0);
        h += s5;
    }
    rb[ESCR] = h - hb; rb[ISCR] = 0; rb[SIG5] = s5; rb[RIGHT] = n; rb[RRIGHT] = m;
    store(pst, n - rb[LEFT] <= jn);
    push();
    rb[LEFT] = rb[RIGHT] = INT32_MAX;
    push();
    const int unp3 = fst.unp / 3;
    fval += A.gop * fst.gap + A.gep * unp3;
    hdr[0] = h; hdr[1] = fst.mch; hdr[2] = fst.mmc; hdr[3] = fst.gap; hdr[4] = unp3; hdr[5] = fval;
    hdr[6] = n_rec; hdr[7] = missed;
    if (fmt == 2) {
        // Vulgar::postproc (src/gsinfo.cc:1206-1226) over the records before the trailing dummy: match lengths next to
        // split codons and frame shifts
        const int cnt = min(n_ops, ops_cap);
        for (int k = 0; k < cnt; ++k) {
            const int o = ops[k].x;
            if (o == 'M' || o == 'D') {
                const bool split = (k + 1 < cnt) ? (ops[k + 1].x == 'S' && ops[k + 1].z == 2) : false;      // (the dummy 'E' follows the last one)
                if (split) { --ops[k].y; ops[k].z -= 3; }
            } else if (o == 'F' && k > 0) {
                if (ops[k].y == 1) { ops[k - 1].z -= 2; ops[k].z += 2; }
                else if (ops[k].y == 2) { ops[k - 1].z -= 1; ops[k].z += 1; ops[k].y = 1; }
            }
        }
    }
    if (fmt) A.ops_cnt[qi] = n_ops;
}

extern "C" hipError_t spdh_launch_rescore(const void* args, hipStream_t stream)
{
    HRescoreArgs A = *reinterpret_cast<const HRescoreArgs*>(args);
    hipLaunchKernelGGL(spdh_rescore, dim3((A.n_probs + 63) / 64), dim3(64), 0, stream, A);
    return hipGetLastError();
}
