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
    HRhomb trb;                         /* forward only (buf == NULL otherwise) */
    int udh;                            /* hirschbergH1_wip: link arrays below are live */
    int16_t *bbuf, *hb, *fb;            /* left-end row ("ml") by diagonal */
    int *cbuf, *hc, *fc;                /* link by diagonal (the reference packs it into one or two shorts) */
    int LocalL, LocalR;
    int max_val, max_mr, max_nr, max_ml, max_ulk;
    int rlst[3];
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
    const int codes = e->trb.buf != NULL;
    int64_t row0 = codes ? hrh_point(&e->trb, p->a_left, p->b_left) : 0;
    const int mw = codes ? e->trb.m_width : 0;
    int* hc = e->hc;
    if (e->udh) {                                       /* :594-603 */
        for (int i = 0; i < 2 * e->buf_size; ++i) e->bbuf[i] = (int16_t) p->a_left;
        const int re = p->a_exgl ? rl : up;
        for (int r = lw; r < re; ++r) hc[r] = r;
        for (int i = 0, r = rl; r >= lw; --r) e->hb[r] = (int16_t) (p->a_left + (i++ / 3));
    }

    if (p->b_exgl == 1) { for (int r = lw; r < rl; ++r) hv[r] = 0; }
    else if (p->b_exgl == 2) { fv[rl] = 0; if (e->udh) e->fc[rl] = rl; }

    int rr = p->b_right - 3 * p->a_left;
    if (up < rr) rr = up;
    int r = rl;
    if (!p->a_exgl) {                                   /* global */
        if (p->b_exgl) { fv[r] = 0; if (e->udh) e->fc[r] = hc[r]; }
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
        if (e->udh) hc[r] = r;
        row0 += mw;
    }
    for (int f = 0; r < rr; ++r, ++n, ++bb, f = (f + 1) % 3) {
        int h = hv[r - 3];
        if (e->udh) hc[r] = hc[r - 3];
        const int gl = r - lend[f];
        if (!(p->a_exgl & 1) && gl == 3) h = w16(h + sc->gop);
        if (!(p->a_exgl & 2)) h = w16(h + gap_ext3(sc, gl));
        h = w16(h + p->sigE[bb - 3]);
        hv[r] = h;
        if (h < NEV) break;
        int x = w16(hv[r - 1] + sc->gapw1);
        if (x > h) { hv[r] = h = x; if (e->udh) hc[r] = hc[r - 1]; if (codes) e->trb.buf[row0] = C_HOR1; }
        x = w16(hv[r - 2] + sc->gapw2);
        if (x > h) { hv[r] = h = x; if (e->udh) hc[r] = hc[r - 2]; if (codes) e->trb.buf[row0] = C_HOR2; }
        x = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
        if (x > h) { hv[r] = x; lend[f] = r; if (e->udh) hc[r] = r; }
        else if (codes) e->trb.buf[row0] = C_HORI;
        row0 += mw;
    }
}

/* fhlastH1 (mode 1): picks the end cell, edits the last row's codes */
static int h_last(HEng* e)
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
    const int codes = e->trb.buf != NULL;
    int64_t rowM = codes ? hrh_point(&e->trb, p->a_right, rw + m3) : 0;
    const int mw = codes ? e->trb.m_width : 0;

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
            else if (k == 1) { hv[h] = w16(cand[1] - s5); if (codes) e->trb.buf[rowM] = C_HORI; }
            else { hv[h] = w16(cand[2]); if (codes) e->trb.buf[rowM] = C_HORI; }
            if (hv[h] > hv[mx]) { mx = h; maxr = rf - (k == 2 ? 3 : 0); }
            if (codes && glen[f] == 3) e->trb.buf[rowM] |= C_NHOR;
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
    if (e->udh) { e->hb[maxt] = e->hb[maxr]; e->max_ulk = e->hc[maxr]; }
    const int q = maxr - rr;
    if (q > 0) e->max_mr = (p->b_right - maxr) / 3;
    else       e->max_nr = maxt + m3;
    return maxt;
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
                SM[k] = w16(sc->mtx[(ml + k < p->a_len ? p->a[ml + k] : 0 /* the byte behind the query: 0 in the reference process */) * sc->mtx_cols + p->b[n - 3 * k - 2]]);
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
                    int ss[NELEM], ph[NELEM], any = 0;
                    for (int k = 0; k < NELEM; ++k) { ss[k] = S3[pk][k]; ph[k] = P3[pk][k]; any |= ph[k]; }
                    for (int k = 0; k < NELEM; ++k) { S3[pk][k + 1] = ss[k]; P3[pk][k + 1] = ph[k]; }
                    if (!any) continue;                                 /* AllZero(ph_v), :224 */
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

/* ======================================================================================== */
/* hirschbergH1_wip (src/fwd2h1_wip_simd.h:338-773): the same lanes carrying, instead of      */
/* traceback codes, the diagonal at which the path crossed the previous intermediate row.     */
/* ======================================================================================== */
typedef struct {            /* UdhIntermediate (udh_intermediate.h:29-66), NOL = 2 */
    int mi;
    int *hlnk[2], *vlnk[2];
    int* buf;
} HImd;

static int h_sweep_udh(HEng* e, int n_im, HImd* imds)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int lw = e->w.lw, up = e->w.up, width = e->w.width;
    int16_t *hv = e->hv, *fv = e->fv, *hb = e->hb, *fb = e->fb;
    int *hc = e->hc, *fc = e->fc;
    const int a_left = p->a_left, a_right = p->a_right, b_left = p->b_left, b_right = p->b_right;
    const int ge = sc->gep, g1 = sc->gapw1, g2 = sc->gapw2, g3 = sc->gapw3;
    const int spj = sc->spj;
    const int ipen = spj ? sc->ipen : NEV;
    const int llmt = sc->llmt;
    const int LocalL = e->LocalL, LocalR = e->LocalR;

    int imd_i = 0;
    HImd* imd = &imds[0];
    int mm = a_left + (imd->mi - a_left - 1) / NELEM * NELEM;
    int k9 = imd->mi - mm;
    int k8 = k9 - 1;

    /* lane buffers that the reference does NOT re-initialise per stripe keep their content */
    static const int ZI[6][NP1] = {{0}};
    int HC[6][NP1], FC[6][NP1], EC[3][NELEM], PV[3][NELEM], SM[NP1];
    memcpy(HC, ZI, sizeof HC); memcpy(FC, ZI, sizeof FC);
    memset(EC, 0, sizeof EC); memset(PV, 0, sizeof PV); memset(SM, 0, sizeof SM);

    for (int ml = a_left; ml < a_right; ml += NELEM) {
        const int j9 = imin(NELEM, a_right - ml);
        const int j8 = j9 - 1;
        int n = imax(b_left, lw + 3 * ml);
        const int n9 = imin(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;
        const int mp1 = ml + 1;
        int q = mod6(n + 3 * mp1);
        int r = n - 3 * mp1;
        int donor_r[3] = {r, r, r};
        int H[6][NP1], F[6][NP1], E[3][NELEM];
        int HB[6][NP1], FB[6][NP1], EB[3][NELEM];
        int CP[3][NP1], S5[6][NP1], S3[6][NP1], P5[6][NP1], P3[6][NP1];
        int hiv[3][NELEM], hil[3][NELEM], hic[3][NELEM], hib[3][NELEM];
        for (int i = 0; i < 6; ++i) for (int k = 0; k < NP1; ++k) {
            H[i][k] = F[i][k] = NEV; HB[i][k] = FB[i][k] = 0;
            S5[i][k] = S3[i][k] = P5[i][k] = P3[i][k] = 0;
        }
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < NELEM; ++k) { E[i][k] = NEV; EB[i][k] = 0; hiv[i][k] = NEV; hil[i][k] = hic[i][k] = hib[i][k] = 0; }
            for (int k = 0; k < NP1; ++k) CP[i][k] = 0;
        }
        for (int k = 0; k < NP1; ++k) SM[k] = 0;                 /* vec_clear(sm_a, 28 * Np1) */
        const int is_imd_ = ml == mm;

        for ( ; n < n9; ++n, ++r, q = mod6(q + 1)) {
            const int f3 = q % 3;
            const int rj = r - 6 * k8;
            const int nb = imax(0, n - b_right + 1);
            const int kb = (nb - 1) / 3;
            const int ke = imin(j9, (n - b_left) / 3);
            const int is_imd = is_imd_ && rj >= lw && rj <= up;
            int cv[NELEM], ev[NELEM], ec[NELEM], eb[NELEM], fvv[NELEM], fcc[NELEM], fbb[NELEM];
            int hx[NELEM], hcx[NELEM], hbx[NELEM], dv[NELEM], pb[NELEM], ab[NELEM];

            CP[f3][0] = (n - 2 >= 0 && good(p, n - 2)) ? p->sigE[n - 2] : 0;
            for (int k = 0; k < NELEM; ++k) cv[k] = CP[f3][k];
            for (int k = 0; k < NELEM; ++k) CP[f3][k + 1] = cv[k];

            const int q1 = mod6(q - 1), q2 = mod6(q - 2), q3 = mod6(q - 3), q4 = mod6(q - 4), q5 = mod6(q - 5);
            /* insertion */
            for (int k = 0; k < NELEM; ++k) {
                int h = sadd(H[q1][k + 1], g1), c = HC[q1][k + 1], b = HB[q1][k + 1];
                int x = sadd(H[q2][k + 1], g2);
                int m = h > x;
                h = m ? h : x; c = m ? c : HC[q2][k + 1]; b = m ? b : HB[q2][k + 1];
                x = sadd(sadd(H[q3][k + 1], g3), cv[k]);
                m = h > x;
                h = m ? h : x; c = m ? c : HC[q3][k + 1]; b = m ? b : HB[q3][k + 1];
                x = sadd(sadd(E[f3][k], ge), cv[k]);
                m = x > h;
                ev[k] = E[f3][k] = m ? x : h;
                ec[k] = EC[f3][k] = m ? EC[f3][k] : c;
                eb[k] = m ? EB[f3][k] : b;
                if (LocalL) EB[f3][k] = eb[k];
            }
            /* deletion */
            F[q3][0] = fv[r + 3]; FC[q3][0] = fc[r + 3]; if (LocalL) FB[q3][0] = fb[r + 3];
            H[q3][0] = hv[r + 3]; HC[q3][0] = hc[r + 3]; if (LocalL) HB[q3][0] = hb[r + 3];
            H[q4][0] = hv[r + 2]; HC[q4][0] = hc[r + 2]; if (LocalL) HB[q4][0] = hb[r + 2];
            H[q5][0] = hv[r + 1]; HC[q5][0] = hc[r + 1]; if (LocalL) HB[q5][0] = hb[r + 1];
            for (int k = 0; k < NELEM; ++k) {
                int h = sadd(F[q3][k], ge), c = FC[q3][k], b = FB[q3][k];
                int x = sadd(H[q3][k], g3);
                int m = h > x;
                h = m ? h : x; c = m ? c : HC[q3][k]; b = m ? b : HB[q3][k];
                x = sadd(H[q4][k], g2);
                m = h > x;
                h = m ? h : x; c = m ? c : HC[q4][k]; b = m ? b : HB[q4][k];
                x = sadd(H[q5][k], g1);
                m = h > x;
                fvv[k] = m ? h : x; fcc[k] = m ? c : HC[q5][k]; fbb[k] = m ? b : HB[q5][k];
            }
            for (int k = 0; k < NELEM; ++k) {
                F[q][k + 1] = fvv[k]; FC[q][k + 1] = fcc[k];
                if (LocalL) FB[q][k + 1] = fbb[k];
            }
            /* diagonal */
            if (nb) for (int k = 0; k < NELEM; ++k) SM[k] = 0;
            for (int k = kb; k < ke; ++k)
                SM[k] = w16(sc->mtx[(ml + k < p->a_len ? p->a[ml + k] : 0 /* the byte behind the query: 0 in the reference process */) * sc->mtx_cols + p->b[n - 3 * k - 2]]);
            H[q][0] = hv[r]; HC[q][0] = hc[r]; if (LocalL) HB[q][0] = hb[r];
            for (int k = 0; k < NELEM; ++k) {
                dv[k] = H[q][k];
                int h = sadd(sadd(SM[k], dv[k]), cv[k]);
                int c = HC[q][k], b = HB[q][k];
                int m = fvv[k] > h;
                h = m ? fvv[k] : h; c = m ? fcc[k] : c; b = m ? fbb[k] : b;
                pb[k] = m ? 2 : 0;
                m = ev[k] > h;
                h = m ? ev[k] : h; c = m ? ec[k] : c; b = m ? eb[k] : b;
                pb[k] = m ? 1 : pb[k];
                hx[k] = h; hcx[k] = c; hbx[k] = b; ab[k] = 0;
            }
            if (is_imd) for (int k = 0; k < NELEM; ++k) PV[f3][k] = pb[k];
            /* intron 3' boundary */
            if (spj) {
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int ph3 = nb ? -2 : p->phs3[n];
                    const int leg = !nb && ph3 > -2 && (!k2 || ph3 == 2);
                    const int phase = leg ? (ph3 == 2 ? (k2 ? 1 : -1) : (k2 ? 2 : ph3)) : 2;
                    const int pk = 2 * f3 + k2;
                    S3[pk][0] = phase < 2 ? p->sig3[n - phase] : MIN_SSV;
                    P3[pk][0] = accpr_code[phase + 1];
                    int ss[NELEM], ph[NELEM], any = 0;
                    for (int k = 0; k < NELEM; ++k) { ss[k] = S3[pk][k]; ph[k] = P3[pk][k]; any |= ph[k]; }
                    for (int k = 0; k < NELEM; ++k) { S3[pk][k + 1] = ss[k]; P3[pk][k + 1] = ph[k]; }
                    if (!any) continue;
                    for (int f = 0; f < 3; ++f) {
                        int acc[NELEM];
                        for (int k = 0; k < NELEM; ++k) {
                            int x = sadd(hiv[f][k], ss[k]);
                            x = sadd(x, qpen(sc, hil[f][k]));
                            x = (ph[k] == accpr_code[f]) ? x : NEV;
                            x = (hil[f][k] > llmt) ? x : NEV;
                            const int m = x > hx[k];
                            hx[k] = m ? x : hx[k];
                            hcx[k] = m ? hic[f][k] : hcx[k];
                            if (LocalL) hbx[k] = m ? hib[f][k] : hbx[k];
                            acc[k] = m ? 1 : 0;
                            ab[k] |= acc[k];
                        }
                        if (is_imd) {
                            for (int k = 0; k < NELEM; ++k) SM[k] = acc[k];      /* Store(sm_a, qv_v) */
                            if (SM[k8]) {
                                imd->hlnk[0][rj] = donor_r[f];
                                imd->hlnk[1][rj] = donor_r[f] + width;
                                e->rlst[f3] = rj;
                            }
                        }
                    }
                }
            }
            /* ends */
            if (LocalL) for (int k = 0; k < NELEM; ++k) if (0 > hx[k]) hx[k] = 0;
            for (int k = 0; k < NELEM; ++k) {
                H[q][k + 1] = hx[k]; HC[q][k + 1] = hcx[k];
                if (LocalL) HB[q][k + 1] = hbx[k];
            }
            if (LocalL)
                for (int k = kb; k < ke; ++k)
                    if (H[q][k + 1] == 0) { HB[q][k + 1] = w16(ml + k); HC[q][k + 1] = r - 6 * k; }
            if (LocalR) {
                int best = 1;
                for (int k = 2; k <= j9; ++k) if (H[q][k] > H[q][best]) best = k;
                if (H[q][best] > e->max_val) {
                    e->max_val = H[q][best];
                    e->max_ml = HB[q][best];
                    e->max_ulk = HC[q][best];
                    e->max_mr = ml + best + 1;
                    e->max_nr = n - 3 * best;
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
                    int ss[NELEM], ph[NELEM], any = 0;
                    for (int k = 0; k < NELEM; ++k) { ss[k] = S5[pk][k]; ph[k] = P5[pk][k]; any |= ph[k]; }
                    for (int k = 0; k < NELEM; ++k) { S5[pk][k + 1] = ss[k]; P5[pk][k + 1] = ph[k]; }
                    if (!any) continue;
                    for (int f = k2 ? 2 : 0; f < 3; ++f) {
                        int don[NELEM];
                        for (int k = 0; k < NELEM; ++k) {
                            /* hx / hcx / hbx here are the values stored above (local resets included for
                             * hx; the link registers hc_v / hb_v are the pre-reset ones) */
                            int x = (f == 2) ? sadd(dv[k], ss[k]) : sadd(hx[k], ss[k]);
                            x = (ab[k] == 0) ? x : NEV;
                            x = (ph[k] == donor_code[f]) ? x : NEV;
                            const int m = x > hiv[f][k];
                            hiv[f][k] = m ? x : hiv[f][k];
                            hil[f][k] = sadd(m ? 0 : hil[f][k], 1);
                            hic[f][k] = m ? hcx[k] : hic[f][k];
                            if (LocalL) hib[f][k] = m ? hbx[k] : hib[f][k];
                            don[k] = m ? 1 : 0;
                        }
                        if (is_imd) {
                            for (int k = 0; k < NELEM; ++k) SM[k] = don[k];      /* Store(sm_a, pv_v) */
                            if (SM[k8]) donor_r[f] = rj;
                        }
                    }
                }
            }
            /* intermediate row */
            if (is_imd) {
                for (int k = 0; k < NELEM; ++k) SM[k] = ab[k];
                if (PV[f3][k8] == 0) e->rlst[f3] = rj;
                if (!SM[k8] && PV[f3][k8] == 1) imd->hlnk[0][rj] = e->rlst[f3];
                imd->vlnk[0][rj] = HC[q][k9];
                HC[q][k9] = rj;
                imd->vlnk[1][rj] = FC[q][k9];
                FC[q][k9] = rj + width;
            }
            /* hand the bottom row to the next stripe */
            const int r0 = r - 6 * j8;
            if (j9 == ke && lw <= r0 && r0 <= up) {
                hv[r0] = (int16_t) H[q][j9]; hc[r0] = HC[q][j9];
                fv[r0] = (int16_t) F[q][j9]; fc[r0] = FC[q][j9];
                if (LocalL) { hb[r0] = (int16_t) HB[q][j9]; fb[r0] = (int16_t) FB[q][j9]; }
            }
        }
        if (is_imd_ && ++imd_i < n_im) {
            imd = &imds[imd_i];
            mm = a_left + (imd->mi - a_left - 1) / NELEM * NELEM;
            k9 = imd->mi - mm;
            k8 = k9 - 1;
        }
    }
    return 0;
}

/* cpos = (n_im + 1) rows of 10 ints, pre-set by the caller to end_of_ulk as lspH_ng does;
 * ranges = a_left, a_right, b_left, b_right after the engine's write-back */
int orc_wip_udh_h(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w, int n_im,
                  int32_t* score, int32_t* cpos, int32_t* ranges)
{
    HEng e;
    memset(&e, 0, sizeof e);
    e.sc = sc; e.p = p; e.w = *w; e.udh = 1;
    e.buf_size = w->width + 6 * NELEM;
    e.vbuf = (int16_t*) malloc(sizeof(int16_t) * 4 * e.buf_size);
    e.cbuf = (int*) calloc(2 * (size_t) e.buf_size, sizeof(int));
    if (!e.vbuf || !e.cbuf) return -1;
    e.hv = e.vbuf - w->lw + 3;
    e.fv = e.hv + e.buf_size;
    e.bbuf = e.vbuf + 2 * e.buf_size;
    e.hb = e.bbuf - w->lw + 3;
    e.fb = e.hb + e.buf_size;
    e.hc = e.cbuf - w->lw + 3;
    e.fc = e.hc + e.buf_size;
    e.LocalL = sc->local && p->a_exgl && p->b_exgl;
    e.LocalR = sc->local && p->a_exgr && p->b_exgr;
    e.max_val = NEV; e.max_ulk = SPDP_END_OF_ULK;
    e.max_ml = p->a_left; e.max_mr = p->a_right; e.max_nr = p->b_right;
    e.rlst[0] = e.rlst[1] = e.rlst[2] = INT_MAX;
    h_init(&e);

    const int step = (p->a_right - p->a_left + n_im) / (n_im + 1);
    HImd* imds = (HImd*) calloc(n_im, sizeof(HImd));
    for (int i = 0; i < n_im; ++i) {
        imds[i].mi = p->a_left + (i + 1) * step;
        imds[i].buf = (int*) malloc(sizeof(int) * 4 * w->width);
        for (int j = 0; j < 4 * w->width; ++j) imds[i].buf[j] = SPDP_END_OF_ULK;
        imds[i].hlnk[0] = imds[i].buf - w->lw + 1;
        imds[i].hlnk[1] = imds[i].hlnk[0] + w->width;
        imds[i].vlnk[0] = imds[i].hlnk[0] + 2 * w->width;
        imds[i].vlnk[1] = imds[i].vlnk[0] + w->width;
    }
    h_sweep_udh(&e, n_im, imds);

    int a_left = p->a_left, a_right = p->a_right, b_left = p->b_left, b_right = p->b_right;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    if (e.LocalR && e.max_mr < a_right) {
        a_right = e.max_mr; b_right = e.max_nr;
    } else {
        const int rt = h_last(&e);
        e.max_ml = e.LocalL ? e.hb[rt] : a_left;
        a_right = e.max_mr; b_right = e.max_nr;
    }
    int val = e.max_val;
    int i = n_im;
    while (--i >= 0 && imds[i].mi > a_right) ;
    if (i < 0 && imds[0].mi > a_right) CPOS(0, 2) = b_right;
    int r = e.max_ulk;
    HImd* imd;
    for ( ; i >= 0 && (imd = imds + i)->mi > e.max_ml; --i) {
        int c = 0, d = 0;
        for ( ; r > w->up; r -= w->width) ++d;
        if (imd->vlnk[d][r] < SPDP_END_OF_ULK) {
            CPOS(i, c++) = imd->mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            const int mm3 = 3 * imd->mi;
            for (int rp = imd->hlnk[d][r];
                 w->lw <= rp && rp < w->up && r != rp;
                 rp = imd->hlnk[d][r = rp])
                CPOS(i, c++) = r + mm3;
            CPOS(i, c++) = r + mm3;
            CPOS(i, c) = SPDP_END_OF_ULK;
            r = imd->vlnk[d][r];
            if (r == SPDP_END_OF_ULK) break;
        } else
            CPOS(i, 0) = SPDP_END_OF_ULK;
    }
    for ( ; r > w->up; r -= w->width) ;
    if (e.LocalL) {
        a_left = e.max_ml;
        b_left = r + 3 * a_left;
    } else {
        const int rl = b_left - 3 * a_left;
        if (p->b_exgl && rl > r) {
            a_left = (b_left - r) / 3;
            for (int j = 0; j < n_im && imds[j].mi < a_left; ++j) CPOS(j, 0) = SPDP_END_OF_ULK;
        }
        if (p->a_exgl && rl < r) b_left = 3 * a_left + r;
    }
    ++i;
    if ((i >= 0 && i < n_im && imds[i].mi < a_left) || CPOS(i, 2) < b_left) val = SPDP_NEVSEL;
#undef CPOS
    *score = val;
    ranges[0] = a_left; ranges[1] = a_right; ranges[2] = b_left; ranges[3] = b_right;
    for (int j = 0; j < n_im; ++j) free(imds[j].buf);
    free(imds);
    free(e.vbuf); free(e.cbuf);
    return 0;
}
