/* spdp_oracle_h.c -- CPU restatement of the reference's aa x genome `_wip` engine.
 *
 * TEST INFRASTRUCTURE ONLY (see spdp_oracle.c): never part of the product path.
 *
 * What it restates (ogotoh/spaln v3.0.7, paths relative to /root/reference):
 *   orc_stripe31        stripe31()                          src/aln2.cc:178-198
 *   h_init              SimdAln2h1::fhinitH1 (mode 1)       src/fwd2h1_simd.h:546-689
 *   h_sweep             SimdAln2h1::forwardH1_wip           src/fwd2h1_wip_simd.h:50-334
 *   h_last              SimdAln2h1::fhlastH1 (mode 1)       src/fwd2h1_simd.h:691-785
 *   hrh_back / trace    Anti_rhomb_coord<SHORT>, step = 3   src/rhomb_coord.h:65-235
 *
 * How: as for the cDNA engines, the 16-lane stripe sweep is replayed one lane at a
 * time.  Lane k of a stripe holds cell (m = ml+1+k, n-3k) at sweep step n; the six
 * rotating phase buffers hv_a[q] / fv_a[q] (q = (n + 3m) mod 6), the three E buffers and
 * the lane-shifted signal pipes are kept as the small arrays they are in the reference,
 * because what lanes outside the DP matrix compute is part of the observable result
 * here: the end cell picked by fhlastH1 can lie beyond b_right, on codes written by such
 * lanes.  Scores are int16 with saturating adds (both sides), plain int16 wrap where
 * the reference stores an int expression into a short.
 *
 * Not restated: the re-basing at `checkpoint` rows (fwd2h1_wip_simd.h:318-329).  With
 * AvTrc() = 57 it first fires at row 512, so queries up to 512 aa are exact; the
 * goldens stay below that.
 *
 * Reference behaviour worth knowing (all reproduced): fhlastH1 never sets maxh.val, so
 * forwardH1_wip returns nevsel (+ accscr) unless a local right end was tracked; the row-0
 * initialiser passes HOR1 where its comment says VERT; a winning b-side end gap moves the
 * end cell beyond b_right.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "../include/spdp.h"

#define NELEM    16
#define NP1      17
#define NEV      (SHRT_MIN + 1024)      /* nevsel, fwd2h1_simd.h:208 */
#define MIN_SSV  (-1000)                /* fwd2h1_wip_simd.h:48 */

enum {  /* TraceBackCode, rhomb_coord.h:36-61 */
    C_DIAG = 1, C_HORI = 2, C_HORL = 3, C_HOR1 = 4, C_HOR2 = 5, C_VERT = 8, C_VERL = 9,
    C_VER1 = 10, C_VER2 = 11, C_ACCM = 13, C_ACCZ = 14, C_ACCP = 15,
    C_NHOR = 16, C_NVER = 32, C_NHOL = 64, C_DONM = 64, C_NVEL = 128, C_DONZ = 128, C_DONP = 256
};
static const int donor_code[4] = {C_DONM, C_DONZ, C_DONP, 0};
static const int accpr_code[4] = {C_ACCM, C_ACCZ, C_ACCP, 0};

static inline int sat16(int x) { return x < SHRT_MIN ? SHRT_MIN : (x > SHRT_MAX ? SHRT_MAX : x); }
static inline int sadd(int x, int y) { return sat16(x + y); }          /* adds_epi16 */
static inline int w16(int x) { return (int16_t) x; }                    /* int -> short store */
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int mod6(int x) { x %= 6; return x < 0 ? x + 6 : x; }

void orc_stripe31(const SpdpProblemH* p, int sh, SpdpWindow* w)
{
    if (sh < 0) {
        int shorter = imin(p->a_right - p->a_left, p->b_right - p->b_left);
        sh = -sh * shorter / 100;
    }
    sh *= 3;
    w->up = p->b_right - 3 * p->a_right;
    w->lw = p->b_left - 3 * p->a_left;
    if (w->up < w->lw) { int t = w->up; w->up = w->lw; w->lw = t; }
    w->up += sh;
    w->lw -= sh;
    int q;
    if ((q = p->b_right - 3 * p->a_left) < w->up) w->up = q;
    if ((q = p->b_left - 3 * p->a_right) > w->lw) w->lw = q;
    w->width = w->up - w->lw + 7;
}

/* (aa, nt) cells of the band, counted as the scalar loops bound n (fwd2h1.cc:322-330) */
int64_t orc_cells_h(const SpdpProblemH* p, const SpdpWindow* w)
{
    int64_t c = 0;
    for (int m = p->a_left + 1; m <= p->a_right; ++m) {
        int n0 = imax(3 * m + w->lw - 1, p->b_left);
        int n9 = imin(3 * m + w->up, p->b_right);
        if (n9 > n0) c += n9 - n0;
    }
    return c;
}

/* ---- traceback bitmap: Anti_rhomb_coord<SHORT>(a_right, b_right, a_left, b_left, 3) ---- */
typedef struct {
    int m_base, n_base, step, m_width, n_width;
    uint16_t* buf;
    size_t size;
    int cur_m, cur_n;
    int64_t cur_p;
} HRhomb;

static int hrh_open(HRhomb* t, int mmax, int nmax, int mbase, int nbase)
{
    t->m_base = mbase; t->n_base = nbase; t->step = 3;
    t->m_width = mmax - mbase + 1;
    t->n_width = nmax - nbase + 1 + 3 * t->m_width;
    t->size = (size_t) t->m_width * t->n_width + 32;
    /* the reference's "+ 32 play" is what its last vector stores run into; keep more here so
     * that lanes past the play area stay inside this allocation */
    t->buf = (uint16_t*) calloc(t->size + 64 + 8 * (size_t) t->m_width, sizeof(uint16_t));
    return t->buf ? 0 : -1;
}
static int64_t hrh_point(HRhomb* t, int m, int n)
{
    t->cur_m = m - t->m_base; t->cur_n = n - t->n_base;
    t->cur_p = (int64_t) (3 * t->cur_m + t->cur_n) * t->m_width + t->cur_m;
    return t->cur_p;
}
static unsigned hrh_left(HRhomb* t, int* m, int* n, int s)
{
    *m = t->cur_m;
    *n = t->cur_n -= s;
    if (*n < 0) { t->cur_n = *n = 0; return 0; }
    t->cur_p -= (int64_t) s * t->m_width;
    return t->buf[t->cur_p];
}
static unsigned hrh_upper(HRhomb* t, int* m, int* n, int s)
{
    *m = --t->cur_m;
    *n = t->cur_n -= s;
    if (*m < 0) { t->cur_m = *m = 0; t->cur_n = *n += s; return 0; }
    else if (*n < 0) {
        if (s > 0) t->cur_m = *m -= *n / s;
        t->cur_n = *n = 0;
        return 0;
    }
    t->cur_p -= (int64_t) (3 + s) * t->m_width + 1;
    return t->buf[t->cur_p];
}
#define BAD_DIR 0xffffffffu
static unsigned hrh_back(HRhomb* t, unsigned code, int* m, int* n)
{
    unsigned dir;
    switch (code & 15) {
    case 0: break;
    case C_DIAG:
        do { if (!(code = hrh_upper(t, m, n, 3))) return 0; } while ((code & 15) == C_DIAG);
        break;
    case C_HORI:
        while (!(code & C_NHOR)) { if (!(code = hrh_left(t, m, n, 3))) return 0; }
        dir = code & 15;
        if (dir != C_HOR1 && dir != C_HOR2) code = hrh_left(t, m, n, 3);
        break;
    case C_HORL:
        while (!(code & C_NHOL)) { if (!(code = hrh_left(t, m, n, 3))) return 0; }
        dir = code & 15;
        if (dir != C_HOR1 && dir != C_HOR2) code = hrh_left(t, m, n, 3);
        break;
    case C_VERT:
        while (!(code & C_NVER)) { if (!(code = hrh_upper(t, m, n, 0))) return 0; }
        dir = code & 15;
        if (dir != C_VER1 && dir != C_VER2) code = hrh_upper(t, m, n, 0);
        break;
    case C_VERL:
        while (!(code & C_NVEL)) { if (!(code = hrh_upper(t, m, n, 0))) return 0; }
        dir = code & 15;
        if (dir != C_VER1 && dir != C_VER2) code = hrh_upper(t, m, n, 0);
        break;
    case C_ACCZ:
        do { if (!(code = hrh_left(t, m, n, 1))) return 0; } while (!(code & C_DONZ));
        break;
    case C_ACCM:
        do { if (!(code = hrh_left(t, m, n, 1))) return 0; } while (!(code & C_DONM));
        break;
    case C_ACCP:
        do { if (!(code = hrh_left(t, m, n, 1))) return 0; } while (!(code & C_DONP));
        code = hrh_upper(t, m, n, 3);
        ++*m; *n += 3;
        break;
    case C_HOR1: code = hrh_left(t, m, n, 1); break;
    case C_HOR2: code = hrh_left(t, m, n, 2); break;
    case C_VER1: code = hrh_upper(t, m, n, 1); break;
    case C_VER2: code = hrh_upper(t, m, n, 2); break;
    default: return BAD_DIR;                        /* fatal("Unexpected dir") in the reference */
    }
    return code;
}

/* ---- engine ------------------------------------------------------------------ */
typedef struct {
    const SpdpScoringH* sc;
    const SpdpProblemH* p;
    SpdpWindow w;
    int buf_size;
    int16_t *vbuf, *hv, *fv;
    HRhomb trb;
    int LocalL, LocalR;
    int max_val, max_mr, max_nr;
} HEng;

static inline int good(const SpdpProblemH* p, int n) { return p->exin_left - 1 <= n && n < p->exin_right; }
static inline int gap_ext3(const SpdpScoringH* sc, int i) { return i > sc->codonk1 ? sc->lgep : sc->gep; }
static inline int qpen(const SpdpScoringH* sc, int hil)
{
    int pv = sc->qm_pen[0];
    for (int j = 1; j < sc->nquant; ++j)
        if (hil > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
    return pv;
}

/* fhinitH1 with a traceback bitmap (mode 1, no Vmf) */
static void h_init(HEng* e)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int lw = e->w.lw, up = e->w.up;
    int16_t *hv = e->hv, *fv = e->fv;
    for (int i = 0; i < 2 * e->buf_size; ++i) e->vbuf[i] = NEV;
    const int rl = p->b_left - 3 * p->a_left;
    int64_t row0 = hrh_point(&e->trb, p->a_left, p->b_left);
    const int mw = e->trb.m_width;

    if (p->b_exgl == 1) { for (int r = lw; r < rl; ++r) hv[r] = 0; }
    else if (p->b_exgl == 2) fv[rl] = 0;

    int rr = p->b_right - 3 * p->a_left;
    if (up < rr) rr = up;
    int r = rl;
    if (!p->a_exgl) {                                   /* global */
        if (p->b_exgl) fv[r] = 0;
        hv[r++] = 0;
        hv[r++] = w16(sc->gapw1);
        hv[r++] = w16(sc->gapw2);
        hv[r++] = w16(sc->gapw3);
        if (sc->gep) {
            int x = (NEV - sc->gapw3) / sc->gep + r;
            if (x < rr) rr = x;
            for ( ; r < rr; ++r) hv[r] = w16(hv[r - 3] + sc->gep);
        } else if (rr > r)
            for (int v = hv[r - 1]; r < rr; ++r) hv[r] = v;
        return;
    }
    /* semi-global: the best of "start here" (sigS) and "extend the leading gap" per frame */
    int n = p->b_left;
    int lend[3] = {r, r + 1, r + 2};
    int bb = n + 1;                                     /* position bb points at */
    for (int f = 0; f < 3; ++f, ++r, ++n, ++bb) {
        hv[r] = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
        row0 += mw;
    }
    for (int f = 0; r < rr; ++r, ++n, ++bb, f = (f + 1) % 3) {
        int h = hv[r - 3];
        const int gl = r - lend[f];
        if (!(p->a_exgl & 1) && gl == 3) h = w16(h + sc->gop);
        if (!(p->a_exgl & 2)) h = w16(h + gap_ext3(sc, gl));
        h = w16(h + p->sigE[bb - 3]);
        hv[r] = h;
        if (h < NEV) break;
        int x = w16(hv[r - 1] + sc->gapw1);
        if (x > h) { hv[r] = h = x; e->trb.buf[row0] = C_HOR1; }
        x = w16(hv[r - 2] + sc->gapw2);
        if (x > h) { hv[r] = h = x; e->trb.buf[row0] = C_HOR2; }
        x = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
        if (x > h) { hv[r] = x; lend[f] = r; }
        else e->trb.buf[row0] = C_HORI;
        row0 += mw;
    }
}

/* fhlastH1 (mode 1): picks the end cell, edits the last row's codes */
static void h_last(HEng* e)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int lw = e->w.lw, up = e->w.up;
    int16_t* hv = e->hv;
    int glen[3] = {0, 0, 0};
    int tcdn[3] = {0, 0, 0};
    const int m3 = 3 * p->a_right;
    int rw = lw;
    int rf = p->b_left - m3;
    if (rf > rw) rw = rf; else rf = rw;
    const int rr = p->b_right - m3;
    int maxr = rr;
    int mx = rr;                                        /* diagonal of the best so far */
    int bb = rw + m3;
    int64_t rowM = hrh_point(&e->trb, p->a_right, rw + m3);
    const int mw = e->trb.m_width;

    if (p->a_exgr) {
        int f = 0;
        for (int h = rw; h <= rr; ++h, ++rf, ++bb, f = (f + 1) % 3) {
            glen[f] += 3;
            int cand[3] = {hv[h], NEV, NEV};
            if (rf - rw >= 3 && !tcdn[f]) {
                cand[1] = hv[h - 3] + p->sigE[bb - 2];
                if (!(p->a_exgr & 2)) cand[1] += gap_ext3(sc, glen[f]);
                if (!(p->a_exgr & 1) && glen[f] == 3) cand[1] += sc->gop;
                if (sc->term_codon) cand[2] = hv[h - 3] + p->sigT[bb - 2];
            }
            if (rf - rw >= 3) tcdn[f] = tcdn[f] || p->sigT[bb - 2] > 0;
            const int s5 = (sc->local && p->sig5[bb] > 0) ? p->sig5[bb] : 0;
            cand[0] += s5;
            cand[1] += s5;
            int k = 0;
            if (cand[1] > cand[k]) k = 1;
            if (cand[2] > cand[k]) k = 2;
            if (k == 0) { glen[f] = 0; tcdn[f] = 0; }
            else if (k == 1) { hv[h] = w16(cand[1] - s5); e->trb.buf[rowM] = C_HORI; }
            else { hv[h] = w16(cand[2]); e->trb.buf[rowM] = C_HORI; }
            if (hv[h] > hv[mx]) { mx = h; maxr = rf - (k == 2 ? 3 : 0); }
            if (glen[f] == 3) e->trb.buf[rowM] |= C_NHOR;
            rowM += mw;
        }
    } else {
        const int y = w16(hv[rr - 3] + p->sigT[bb + (rr - rw)]);
        if (y > hv[rr]) { hv[rr] = y; maxr = rr - 3; }
    }
    if (p->b_exgr) {
        rw = imin(up - 1, p->b_right - 3 * p->a_left);
        int g[3] = {NEV, NEV, NEV};
        int f = 0;
        for (int h = rw - 3; h > rr; --h, f = (f + 1) % 3) {
            int x = hv[h + 3];
            if (!(p->b_exgr & 1)) x = w16(x + sc->gop);
            if (x > g[f]) g[f] = x;
            if (!(p->b_exgr & 2)) g[f] = w16(g[f] + sc->gep);
            if (hv[h] > g[f]) g[f] = NEV;
            else if (g[f] > hv[mx]) { mx = h; hv[h] = g[f]; }
        }
    }
    const int maxt = mx;
    const int q = maxr - rr;
    if (q > 0) e->max_mr = (p->b_right - maxr) / 3;
    else       e->max_nr = maxt + m3;
}

/* forwardH1_wip main loop */
static void h_sweep(HEng* e)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int lw = e->w.lw, up = e->w.up;
    int16_t *hv = e->hv, *fv = e->fv;
    const int a_left = p->a_left, a_right = p->a_right, b_left = p->b_left, b_right = p->b_right;
    const int ge = sc->gep, g1 = sc->gapw1, g2 = sc->gapw2, g3 = sc->gapw3;
    const int spj = sc->spj;
    const int ipen = spj ? sc->ipen : NEV;
    const int llmt = sc->llmt;
    const int mw = a_right - a_left;
    const int mb = a_right - NELEM;
    const int mt = a_left + mw / NELEM * NELEM;
    const int nlast = a_right - mt;                     /* rows of the last, partial stripe */
    uint16_t* tb = e->trb.buf;

    for (int ml = a_left; ml < a_right; ml += NELEM) {
        const int j9 = imin(NELEM, a_right - ml);
        const int j8 = j9 - 1;
        int n = imax(b_left, lw + 3 * ml);
        const int n9 = imin(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;
        const int mp1 = ml + 1;
        int q = mod6(n + 3 * mp1);
        int r = n - 3 * mp1;
        int H[6][NP1], F[6][NP1], E[3][NELEM];
        int CP[3][NP1], S5[6][NP1], S3[6][NP1], P5[6][NP1], P3[6][NP1], SM[NP1];
        int hiv[3][NELEM], hil[3][NELEM];
        for (int i = 0; i < 6; ++i) for (int k = 0; k < NP1; ++k) {
            H[i][k] = F[i][k] = NEV; S5[i][k] = S3[i][k] = P5[i][k] = P3[i][k] = 0;
        }
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < NELEM; ++k) { E[i][k] = NEV; hiv[i][k] = NEV; hil[i][k] = 0; }
            for (int k = 0; k < NP1; ++k) CP[i][k] = 0;
        }
        for (int k = 0; k < NP1; ++k) SM[k] = 0;

        for ( ; n <= n9; ++n, ++r, q = mod6(q + 1)) {
            const int f3 = q % 3;                       /* frame of this step */
            const int nb = imax(0, n - b_right + 1);
            const int kb = (nb - 1) / 3;
            const int ke = imin(j9, (n - b_left) / 3);
            const int64_t tp = hrh_point(&e->trb, mp1, n);
            int cv[NELEM], ev[NELEM], fvv[NELEM], hx[NELEM], dv[NELEM];
            int eb[NELEM], hb[NELEM], pb[NELEM], ab[NELEM];

            /* coding potential pipe */
            CP[f3][0] = good(p, n - 2) ? p->sigE[n - 2] : 0;
            for (int k = 0; k < NELEM; ++k) cv[k] = CP[f3][k];
            for (int k = 0; k < NELEM; ++k) CP[f3][k + 1] = cv[k];

            /* horizontal: 1-nt / 2-nt frame shift, new codon insertion, extension */
            const int q1 = mod6(q - 1), q2 = mod6(q - 2), q3 = mod6(q - 3), q4 = mod6(q - 4), q5 = mod6(q - 5);
            for (int k = 0; k < NELEM; ++k) {
                int h = sadd(H[q1][k + 1], g1);
                int x = sadd(H[q2][k + 1], g2);
                int m = h > x;
                h = m ? h : x;
                int b = m ? C_HOR1 : C_HOR2;
                x = sadd(sadd(H[q3][k + 1], g3), cv[k]);
                m = h > x;
                h = m ? h : x;
                b = m ? b : C_HORI;
                int ee = sadd(sadd(E[f3][k], ge), cv[k]);
                m = ee > h;
                ee = m ? ee : h;
                hb[k] = m ? 0 : C_NHOR;
                eb[k] = m ? C_HORI : b;
                ev[k] = E[f3][k] = ee;
            }
            /* vertical: extension, codon deletion, 2-nt / 1-nt frame shift */
            F[q3][0] = fv[r + 3];
            H[q3][0] = hv[r + 3];
            H[q4][0] = hv[r + 2];
            H[q5][0] = hv[r + 1];
            for (int k = 0; k < NELEM; ++k) {
                int f = sadd(F[q3][k], ge);
                int h = sadd(H[q3][k], g3);
                int x = sadd(H[q4][k], g2);
                int m = h > x;
                h = m ? h : x;
                int b = m ? C_VERT : C_VER1;
                x = sadd(H[q5][k], g1);
                m = h > x;
                h = m ? h : x;
                b = m ? b : C_VER2;
                m = f > h;
                f = m ? f : h;
                hb[k] |= m ? 0 : C_NVER;
                pb[k] = m ? C_VERT : b;
                fvv[k] = f;
            }
            for (int k = 0; k < NELEM; ++k) F[q][k + 1] = fvv[k];
            /* diagonal */
            if (nb) for (int k = 0; k < NELEM; ++k) SM[k] = 0;
            for (int k = kb; k < ke; ++k)
                SM[k] = w16(sc->mtx[p->a[ml + k] * sc->mtx_cols + p->b[n - 3 * k - 2]]);
            H[q][0] = hv[r];
            for (int k = 0; k < NELEM; ++k) {
                dv[k] = H[q][k];
                int h = sadd(sadd(SM[k], dv[k]), cv[k]);
                int m = fvv[k] > h;
                h = m ? fvv[k] : h;
                pb[k] = m ? pb[k] : C_DIAG;
                m = ev[k] > h;
                h = m ? ev[k] : h;
                pb[k] = m ? eb[k] : pb[k];
                hx[k] = h;
                ab[k] = 0;
            }
            /* intron 3' boundary */
            if (spj) {
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int ph3 = nb ? -2 : p->phs3[n];
                    const int leg = !nb && ph3 > -2 && (!k2 || ph3 == 2);
                    const int phase = leg ? (ph3 == 2 ? (k2 ? 1 : -1) : (k2 ? 2 : ph3)) : 2;
                    const int pk = 2 * f3 + k2;
                    S3[pk][0] = phase < 2 ? p->sig3[n - phase] : MIN_SSV;
                    P3[pk][0] = accpr_code[phase + 1];
                    int ss[NELEM], ph[NELEM];
                    for (int k = 0; k < NELEM; ++k) { ss[k] = S3[pk][k]; ph[k] = P3[pk][k]; }
                    for (int k = 0; k < NELEM; ++k) { S3[pk][k + 1] = ss[k]; P3[pk][k + 1] = ph[k]; }
                    for (int f = k2 ? 2 : 0; f < 3; ++f)
                        for (int k = 0; k < NELEM; ++k) {
                            int x = sadd(hiv[f][k], ss[k]);
                            x = sadd(x, qpen(sc, hil[f][k]));
                            x = (ph[k] == accpr_code[f]) ? x : NEV;
                            x = (hil[f][k] > llmt) ? x : NEV;
                            const int m = x > hx[k];
                            hx[k] = m ? x : hx[k];
                            pb[k] = m ? accpr_code[f] : pb[k];
                            ab[k] |= m ? ph[k] : 0;
                        }
                }
            }
            /* local left end (accscr stays 0 here) */
            if (e->LocalL)
                for (int k = 0; k < NELEM; ++k)
                    if (0 > hx[k]) { hx[k] = 0; hb[k] = 0; }
            for (int k = 0; k < NELEM; ++k) H[q][k + 1] = hx[k];
            if (e->LocalR) {
                int best = 1;
                for (int k = 2; k <= j9; ++k) if (H[q][k] > H[q][best]) best = k;
                if (H[q][best] > e->max_val) {
                    e->max_val = H[q][best];
                    e->max_mr = ml + best;
                    e->max_nr = n - 3 * best + 3;
                }
            }
            /* intron 5' boundary */
            if (spj) {
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int ph5 = nb ? -2 : p->phs5[n];
                    const int leg = !nb && ph5 > -2 && (!k2 || ph5 == 2);
                    const int phase = leg ? (ph5 == 2 ? (k2 ? 1 : -1) : (k2 ? 2 : ph5)) : 2;
                    const int pk = 2 * f3 + k2;
                    S5[pk][0] = phase < 2 ? w16(p->sig5[n - phase] + ipen) : MIN_SSV;
                    P5[pk][0] = donor_code[phase + 1];
                    int ss[NELEM], ph[NELEM];
                    for (int k = 0; k < NELEM; ++k) { ss[k] = S5[pk][k]; ph[k] = P5[pk][k]; }
                    for (int k = 0; k < NELEM; ++k) { S5[pk][k + 1] = ss[k]; P5[pk][k + 1] = ph[k]; }
                    for (int f = k2 ? 2 : 0; f < 3; ++f)
                        for (int k = 0; k < NELEM; ++k) {
                            int x = (f == 2) ? sadd(dv[k], ss[k]) : sadd(hx[k], ss[k]);
                            x = (ab[k] == 0) ? x : NEV;                     /* no empty exon */
                            x = (ph[k] == donor_code[f]) ? x : NEV;
                            const int m = x > hiv[f][k];
                            hiv[f][k] = m ? x : hiv[f][k];
                            hb[k] |= m ? donor_code[f] : 0;
                            hil[f][k] = m ? 0 : hil[f][k];
                        }
                }
                for (int f = 0; f < 3; ++f)
                    for (int k = 0; k < NELEM; ++k) hil[f][k] = sadd(hil[f][k], 1);
            }
            /* hand the bottom row to the next stripe */
            const int r0 = r - 6 * j8;
            if (j9 == ke && lw <= r0 && r0 <= up) {
                hv[r0] = (int16_t) H[q][j9];
                fv[r0] = (int16_t) F[q][j9];
            }
            /* traceback codes: one vector store, OR-ed where the last stripe overlaps the row end */
            for (int k = 0; k < NELEM; ++k) {
                int code = hb[k] | pb[k];
                if (ml == mt && k >= nlast) code = 0;
                if (ml > mb) code |= tb[tp + k];
                tb[tp + k] = (uint16_t) code;
            }
        }
    }
}

/* forwardH1_wip: score + raw Mfile records (end -> start); caller frees *skl.
 * returns 0, -2 when the reference would stop with "Unexpected dir", -3 when it would start its
 * traceback outside the bitmap. */
int orc_wip_forward_h(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w,
                      int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    HEng e;
    memset(&e, 0, sizeof e);
    e.sc = sc; e.p = p; e.w = *w;
    e.buf_size = w->width + 6 * NELEM;
    e.vbuf = (int16_t*) malloc(sizeof(int16_t) * 2 * e.buf_size);
    if (!e.vbuf) return -1;
    e.hv = e.vbuf - w->lw + 3;
    e.fv = e.hv + e.buf_size;
    e.LocalL = sc->local && p->a_exgl && p->b_exgl;
    e.LocalR = sc->local && p->a_exgr && p->b_exgr;
    e.max_val = NEV; e.max_mr = p->a_right; e.max_nr = p->b_right;
    if (hrh_open(&e.trb, p->a_right, p->b_right, p->a_left, p->b_left)) { free(e.vbuf); return -1; }
    if (!p->a_exgl)                                     /* initialize_m0(4) */
        for (int n = 1; n < e.trb.n_width; ++n) e.trb.buf[(size_t) n * e.trb.m_width] = 4;
    h_init(&e);
    h_sweep(&e);
    if (!e.LocalR || e.max_mr == p->a_right) h_last(&e);
    *score = e.max_val;

    int cap = 64, cnt = 0, rc = 0;
    SpdpSkl* out = (SpdpSkl*) malloc(cap * sizeof(SpdpSkl));
    HRhomb* t = &e.trb;
    int m = e.max_mr, n = e.max_nr;
    /* a winning b-side end gap puts the start cell beyond b_right, possibly beyond the bitmap:
     * the reference then reads past its allocation -- undefined, reported as rc = -3 with the
     * single start record */
    const int64_t sp = hrh_point(t, m, n);
    const int inside = sp >= 0 && (size_t) sp < t->size;
    unsigned code = inside ? t->buf[sp] : 0;
    if (!inside) rc = -3;
    m -= t->m_base; n -= t->n_base;
    for (;;) {
        if (cnt == cap) { cap *= 2; out = (SpdpSkl*) realloc(out, cap * sizeof(SpdpSkl)); }
        out[cnt].m = m + t->m_base; out[cnt].n = n + t->n_base; ++cnt;
        if (!code) break;
        code = hrh_back(t, code, &m, &n);
        if (code == BAD_DIR) { rc = -2; break; }
    }
    *skl = out; *n_skl = cnt;
    free(t->buf);
    free(e.vbuf);
    return rc;
}
