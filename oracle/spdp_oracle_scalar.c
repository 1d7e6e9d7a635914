/* spdp_oracle_scalar.c -- CPU restatement of the reference's SCALAR exact-ILD engines.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as spdp_oracle.c).
 *
 * Restates (ogotoh/spaln v3.0.7):
 *   orc_scalar_scorealone  Aln2s1::scorealoneS_ng + sinitS_ng/slastS_ng  src/fwd2s1.cc:1163-1336, 1112-1161
 *   orc_scalar_forward     Aln2s1::forwardS_ng + initS_ng/lastS_ng       src/fwd2s1.cc:217-444, 142-215
 *                          + Vmf::traceback                              src/vmf.cc:125-140
 *                          + the record fix-up of trcbkalignS_ng          src/fwd2s1.cc:1690-1707
 * These are the -A0 engines (int32, row by row, exact intron-length penalty
 * with the top-NCAND donor list per row) -- also what the -A2/-A3 dispatch
 * falls back to for sub-problems with fewer than 8 query rows
 * (trcbkalignS_ng, src/fwd2s1.cc:1677).  Affine gaps (Noll = 2), no cip, no
 * cut range.  The junction score is
 *   spjscr(jnc, n) = IntPen(n - jnc) + sig3[n] + T53[16*dinc5[jnc] + dinc3[n]]
 * (SpJunc::spjscr src/codepot.cc:74-77, Exinon::sig53 IE53 src/codepot.cc:411-415).
 */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include "../include/spdp.h"

#define NCAND 4
#define NOD 5                                   /* 2 * Noll - 1 states at most: DIAG, HORI, VERT, HORL, VERL (src/fwd2s1.cc:223); three when Noll = 2 */
#define E2_PSP 2
/* PwdB::GapPenalty / GapExtPen (src/aln.h:275-282): beyond codonk1 a gap is priced with the long pair (Noll = 3 only:
 * codonk1 is LARGEN otherwise, src/aln2.cc:114) */
static inline int gap_penalty(const SpdpScoring* sc, int i) { if (!i) return 0; return (sc->noll == 3 && i > sc->codonk1) ? sc->lgop + i * sc->lgep : sc->gop + i * sc->gep; }
static inline int gap_ext_pen(const SpdpScoring* sc, int i) { return (sc->noll == 3 && i > sc->codonk1) ? sc->lgep : sc->gep; }
static const unsigned char psp_bit[5] = {4, 1, 8, 2, 16};    /* src/aln.h:56 */
#define E1_PSP 1

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

static inline int intpen(const SpdpScoring* sc, int len)
{
    if (len < 0) return SHRT_MIN;
    if (len >= sc->intpen_len) len = sc->intpen_len - 1;
    return sc->intpen[len];
}
static inline int spjscr(const SpdpScoring* sc, const SpdpProblem* p, int jnc, int n)
{
    return intpen(sc, n - jnc) + p->sig3[n] + sc->t53[16 * (p->dinc[jnc] >> 4) + (p->dinc[n] & 15)];
}

typedef struct { int val, dir, jnc, ptr; } Cand;

/* ---- score only ------------------------------------------------------------------------ */
int orc_scalar_scorealone(const SpdpScoring* sc, const SpdpProblem* p, const SpdpWindow* w, int32_t* score)
{
    if ((sc->noll != 2 && sc->noll != 3) || !sc->intpen || !p->cano5) return -1;
    const int dagp = sc->noll == 3, Nod = dagp ? 5 : 3;
    const int GOP[3] = {0, sc->gop, sc->lgop};          /* PwdB::GOP, src/aln2.cc:111 */
    const int NEV = SPDP_NEVSEL;
    const int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int LocalL = sc->local && p->a_exgl && p->b_exgl;
    const int LocalR = sc->local && p->a_exgr && p->b_exgr;
    const int dim = sc->mtx_dim;
    const size_t bufsiz = (size_t) sc->noll * w->width;
    int* wbuf = (int*) malloc(bufsiz * sizeof(int));
    for (size_t i = 0; i < bufsiz; ++i) wbuf[i] = NEV;
    int* hh0 = wbuf - w->lw + 1;
    int* hh1 = hh0 + w->width;
    int* hh2 = hh1 + w->width;                          /* F2, Noll = 3 */
    int blackv = NEV;
    /* sinitS_ng */
    {
        int r = bl - al, rr = br - al;
        int* h = hh0 + r;
        *h = 0;
        if (p->a_exgl) { if (w->up < rr) rr = w->up; for (int i = 1; i <= rr - r; ++i) h[i] = 0; }
        rr = bl - ar;
        if (w->lw > rr) rr = w->lw;
        if (p->b_exgl) { for (int i = rr; i < r; ++i) hh0[i] = 0; }
        else {
            int* f = hh1 + r;
            for (int i = 1; --r >= rr; ++i) {
                --h; --f;
                *h = h[1];
                if (i == 1) { *h += gap_penalty(sc, 1); *f = *h; }
                else { *f = f[1]; *h += gap_ext_pen(sc, i); *f += sc->gep; }
            }
        }
    }
    int maxh = NEV;
    int m = al;
    if (!p->a_exgl) --m;
    int n1 = m + w->lw, n2 = m + w->up + 1;
    for ( ; ++m <= ar; ++n1, ++n2) {
        int n = imax(n1, bl);
        const int n9 = imin(n2, br);
        unsigned psp = 0;
        int r = n - m;
        int *h = hh0 + r, *f = hh1 + r, *f2 = dagp ? hh2 + r : &blackv;
        int e1 = NEV, e2 = NEV;
        int* hf[NOD] = {0, &e1, 0, &e2, 0};
        Cand rcd[NCAND + 1];
        int idx[NCAND + 1];
        for (int l = 0; l <= NCAND; ++l) { rcd[l].val = NEV; rcd[l].dir = rcd[l].jnc = rcd[l].ptr = 0; idx[l] = l; }
        int ncand = -1;
        const int32_t* qprof = (m >= 1) ? sc->mtx + (size_t) p->a[m - 1] * dim : sc->mtx;
        for ( ; ++n <= n9; ) {
            int x;
            ++h; ++f; if (dagp) ++f2;
            hf[0] = h; hf[2] = f; hf[4] = f2;
            int* from = h;
            int* mx = h;
            if (m != al) {
                *h += qprof[p->b[n - 1]];
                x = *++from + sc->gop;
                *f = imax(x, f[1]) + sc->gep;
                if (*f > *mx) mx = f;
                if (dagp) {                                 /* :1227-1231 */
                    x = *from + sc->lgop;
                    *f2 = imax(x, f2[1]) + sc->lgep;
                    if (*f2 > *mx) mx = f2;
                }
            }
            x = h[-1] + sc->gop;
            const unsigned prev_psp = psp;
            if (x > e1) { e1 = x; psp = psp ? E1_PSP : 0; }
            else psp &= E1_PSP;
            e1 += sc->gep;
            if (e1 > *mx) mx = &e1;
            if (dagp) {                                     /* :1245-1254 */
                x = h[-1] + sc->lgop;
                if (x > e2) { e2 = x; if (prev_psp) psp |= E2_PSP; }
                else psp |= (prev_psp & E2_PSP);
                e2 += sc->lgep;
                if (e2 > *mx) mx = &e2;
            }
            if (p->cano3[n]) {
                const Cand* maxphl[NOD] = {0, 0, 0, 0, 0};
                for (int l = 0; l <= ncand; ++l) {
                    const Cand* prd = rcd + idx[l];
                    if (n - prd->jnc < sc->llmt) continue;
                    from = hf[prd->dir];
                    x = prd->val + (p->cip ? p->cip[m] : 0) + spjscr(sc, p, prd->jnc, n);     /* sigB = cip_score(m) */
                    if (x > *from) { *from = x; maxphl[prd->dir] = prd; }
                }
                for (int k = 0; k < Nod; ++k) {
                    if (!maxphl[k]) continue;
                    psp |= psp_bit[k];
                    from = hf[k];
                    if (*from > *mx) mx = from;
                }
            }
            int y = *h;
            if (h != mx) *h = *mx;
            else if (LocalR && y > maxh) maxh = y;
            if (LocalL && *h < 0) *h = 0;
            int hd = 0;
            for ( ; mx != hf[hd]; ++hd) ;
            if (p->cano5[n]) {
                const int sigJ = p->sig5[n];
                for (int k = hd == 0 ? 0 : 1; k < Nod; ++k) {
                    from = hf[k];
                    if (psp & psp_bit[k]) continue;
                    if (k != hd) {
                        y = *mx;
                        if (hd == 0 || (k - hd) % 2) y += GOP[k / 2];
                        if (*from <= y) continue;
                    }
                    x = *from + sigJ;
                    int l = ncand < NCAND ? ++ncand : NCAND;
                    while (--l >= 0) {
                        if (x >= rcd[idx[l]].val) { int t = idx[l]; idx[l] = idx[l + 1]; idx[l + 1] = t; }
                        else break;
                    }
                    if (++l < NCAND) {
                        Cand* prd = rcd + idx[l];
                        prd->val = x; prd->jnc = n; prd->dir = k;
                    } else --ncand;
                }
            }
        }
    }
    if (!LocalR) {  /* slastS_ng */
        int* h9 = hh0 + br - ar;
        int mx = *h9;
        if (p->b_exgr) {
            int rw = imin(w->up, br - al);
            for (int* h = hh0 + rw; h > h9; --h) if (*h > mx) mx = *h;
        }
        if (p->a_exgr) {
            int rw = imax(w->lw, bl - ar);
            for (int* h = hh0 + rw; h < h9; ++h) if (*h > mx) mx = *h;
        }
        maxh = mx;
    }
    free(wbuf);
    *score = maxh;
    return 0;
}

/* ---- forward with Vmf traceback ------------------------------------------------------------ */
typedef struct { int m, n, p; } Sklp;
typedef struct { Sklp* rec; int n, cap; } Vmf;
static int vmf_add(Vmf* v, int m, int n, int p)
{
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->rec = (Sklp*) realloc(v->rec, v->cap * sizeof(Sklp)); }
    v->rec[v->n].m = m; v->rec[v->n].n = n; v->rec[v->n].p = p;
    return v->n++;
}
typedef struct { int val, ptr; } Rvp;

enum { NEWD = 8 };               /* the file-local Newd of src/fwd2s1.cc:48 (a direction byte holds the state that won, 0 .. 4, or Newd) */

/* returns the records trcbkalignS_ng hands to the Mfile (end -> start); caller frees *skl */
/* cut_l < cut_r: forwardS_ng's cut range (`cutrng`, src/fwd2s1.cc:217, 423-430; initS_ng :157-161, lastS_ng :191-204):
 * after column cut_l of a row the sweep charges the horizontal gap for the cut_r - cut_l columns it jumps over and goes on
 * behind them; the diagonal arrays are narrower by that much and simply keep counting (shortcutS_ng, :1899-1930) */
static int scalar_forward_impl(const SpdpScoring* sc, const SpdpProblem* p, const SpdpWindow* w, int cut_l, int cut_r,
                       int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    *skl = 0; *n_skl = 0;
    const int has_cut = cut_r > cut_l || (cut_l | cut_r) != 0;
    const int cutlen = has_cut ? cut_r - cut_l : 0;
    if ((sc->noll != 2 && sc->noll != 3) || !sc->intpen || !p->cano5) return -1;
    const int dagp = sc->noll == 3, Nod = dagp ? 5 : 3;
    const int GOP[3] = {0, sc->gop, sc->lgop};
    if (w->width < 0) { *score = SPDP_NEVSEL; return 0; }
    const int NEV = SPDP_NEVSEL;
    const int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int Local = sc->local;
    const int LocalL = Local && p->a_exgl && p->b_exgl;
    const int LocalR = Local && p->a_exgr && p->b_exgr;
    const int spj = sc->spj;
    const int dim = sc->mtx_dim;
    const int width = w->width - cutlen;
    if (width < 3) { *score = NEV; return -1; }
    const size_t bufsiz = (size_t) sc->noll * width;
    Rvp* jbuf = (Rvp*) malloc(bufsiz * sizeof(Rvp));
    for (size_t i = 0; i < bufsiz; ++i) { jbuf[i].val = NEV; jbuf[i].ptr = 0; }
    Rvp* hh0 = jbuf - w->lw + 1;
    Rvp* hh1 = hh0 + width;
    Rvp* hh2 = hh1 + width;                             /* F2, Noll = 3 */
    Rvp blackvp = {NEV, 0};
    unsigned char* dbuf = (unsigned char*) calloc(width, 1);
    unsigned char* hdir = dbuf - w->lw + 1;
    Vmf vmf = {0, 0, 0};
    vmf_add(&vmf, 0, 0, 0);                     /* skip 0-th record */
    /* initS_ng */
    {
        int n = bl, r = bl - al, rr = br - al;
        Rvp* h = hh0 + r;
        unsigned char* hd = hdir + r;
        h->val = 0; *hd = 0;
        h->ptr = vmf_add(&vmf, al, n, 0);
        if (p->a_exgl) {
            if (w->up < rr) rr = w->up;
            while (++r <= rr) {
                if (has_cut && n == cut_l) { n += cutlen; r += cutlen; continue; }
                ++h; h->val = 0; *++hd = 1; h->ptr = 0;
            }
        }
        r = bl - al; rr = bl - ar;
        h = hh0 + r; hd = hdir + r;
        if (w->lw > rr) rr = w->lw;
        for (int i = 1; --r >= rr; ++i) {
            --h; *--hd = 2;
            if (p->b_exgl) { h->val = 0; h->ptr = 0; }
            else {
                *h = h[1];
                if (i == 1) h->val += gap_penalty(sc, 1);
                else h->val += gap_ext_pen(sc, i);
            }
        }
    }
    int maxh_val = NEV, maxh_m = al, maxh_n = bl, maxh_p = 0;
    int m = al;
    if (!p->a_exgl) --m;
    int n1 = m + w->lw, n2 = m + w->up + 1;
    for ( ; ++m <= ar; ++n1, ++n2) {
        const int internal = spj && (!p->a_exgr || m < ar);
        int n = imax(n1, bl);
        const int n9 = imin(n2, br);
        int r = n - m;
        Rvp *h = hh0 + r, *f = hh1 + r, *f2 = dagp ? hh2 + r : &blackvp;
        unsigned char* dir = hdir + r;
        unsigned psp = 0;
        Rvp e1 = {NEV, 0}, e2 = {NEV, 0};
        Rvp* hf[NOD] = {0, &e1, 0, &e2, 0};
        Cand rcd[NCAND + 1];
        int idx[NCAND + 1];
        for (int l = 0; l <= NCAND; ++l) { rcd[l].val = NEV; rcd[l].ptr = rcd[l].dir = rcd[l].jnc = 0; idx[l] = l; }
        int ncand = -1;
        const int32_t* qprof = (m >= 1) ? sc->mtx + (size_t) p->a[m - 1] * dim : sc->mtx;
        for ( ; ++n <= n9; ) {
            int x;
            ++dir; ++h; ++f; if (dagp) ++f2;
            hf[0] = h; hf[2] = f; hf[4] = f2;
            Rvp* from = h;
            Rvp* mx = h;
            const int diag = h->val;
            if (m != al) {
                h->val += qprof[p->b[n - 1]];
                *dir = (*dir % NEWD) ? NEWD : 0;
                x = (++from)->val + sc->gop;
                if (x >= f[1].val) { f->val = x; f->ptr = from->ptr; }
                else *f = f[1];
                f->val += sc->gep;
                if (f->val > mx->val) mx = f;
                if (dagp) {                                 /* :297-305 */
                    x = from->val + sc->lgop;
                    if (x >= f2[1].val) { f2->val = x; f2->ptr = from->ptr; }
                    else *f2 = f2[1];
                    f2->val += sc->lgep;
                    if (f2->val > mx->val) mx = f2;
                }
            }
            x = h[-1].val + sc->gop;
            const unsigned prev_psp = psp;
            if (x >= e1.val) { e1.val = x; e1.ptr = h[-1].ptr; psp = psp ? E1_PSP : 0; }
            else psp &= E1_PSP;
            e1.val += sc->gep;
            if (e1.val >= mx->val) mx = &e1;
            if (dagp) {                                     /* :320-330 */
                x = h[-1].val + sc->lgop;
                if (x >= e2.val) { e2.val = x; e2.ptr = h[-1].ptr; if (prev_psp) psp |= E2_PSP; }
                else psp |= (prev_psp & E2_PSP);
                e2.val += sc->lgep;
                if (e2.val >= mx->val) mx = &e2;
            }
            if (internal && p->cano3[n]) {
                const Cand* maxphl[NOD] = {0, 0, 0, 0, 0};
                for (int l = 0; l <= ncand; ++l) {
                    const Cand* prd = rcd + idx[l];
                    if (n - prd->jnc < sc->llmt) continue;
                    x = prd->val + (p->cip ? p->cip[m] : 0) + spjscr(sc, p, prd->jnc, n);     /* sigB = cip_score(m) */
                    from = hf[prd->dir];
                    if (x >= from->val) { from->val = x; maxphl[prd->dir] = prd; }
                }
                for (int k = 0; k < Nod; ++k) {
                    const Cand* prd = maxphl[k];
                    if (!prd) continue;
                    from = hf[k];
                    psp |= psp_bit[k];
                    from->ptr = vmf_add(&vmf, m, n, vmf_add(&vmf, m, prd->jnc, prd->ptr));
                    if (from->val >= mx->val) mx = from;
                }
            }
            int hd = 0;
            if (h != mx) {
                *h = *mx;
                while (mx != hf[++hd]) ;
                *dir = (unsigned char) hd;
            } else if (Local && h->val > diag) {
                if (LocalL && diag == 0) h->ptr = vmf_add(&vmf, m - 1, n - 1, 0);
                else if (LocalR && h->val > maxh_val) { maxh_val = h->val; maxh_p = h->ptr; maxh_m = m; maxh_n = n; }
            }
            if (LocalL && h->val <= 0) { h->val = 0; *dir = 1; }
            else if (*dir == NEWD && !(psp & psp_bit[0]))
                h->ptr = vmf_add(&vmf, m - 1, n - 1, h->ptr);
            if (internal && p->cano5[n]) {
                const int sigJ = p->sig5[n];
                for (int k = hd == 0 ? 0 : 1; k < Nod; ++k) {
                    from = hf[k];
                    if (psp & psp_bit[k]) continue;
                    if (k != hd) {
                        int z = mx->val;
                        if (hd == 0 || (k - hd) % 2) z += GOP[k / 2];
                        if (from->val <= z) continue;
                    }
                    x = from->val + sigJ;
                    int l = ncand < NCAND ? ++ncand : NCAND;
                    while (--l >= 0) {
                        if (x > rcd[idx[l]].val) { int t = idx[l]; idx[l] = idx[l + 1]; idx[l + 1] = t; }
                        else break;
                    }
                    if (++l < NCAND) {
                        Cand* prd = rcd + idx[l];
                        prd->val = x; prd->jnc = n; prd->dir = k; prd->ptr = from->ptr;
                    } else --ncand;
                }
            }
            if (has_cut && n == cut_l) {                /* shortcut: the gap runs on over the cut */
                e1.val += sc->gep * cutlen;
                if (dagp) e2.val += sc->lgep * cutlen;
                *h = dagp ? e2 : e1;
                f->val = NEV; f->ptr = 0;
                n += cutlen;
            }
        }
    }
    int ptr = 0, scr;
    if (LocalR) { ptr = vmf_add(&vmf, maxh_m, maxh_n, maxh_p); scr = maxh_val; }
    else {      /* lastS_ng */
        int rw = w->lw;
        int rf = (has_cut ? cut_l : bl) - ar;
        if (rf > rw) rw = rf;
        Rvp* h = hh0 + rw - cutlen;
        Rvp* h9 = hh0 + br - ar - cutlen;
        Rvp* mx = h9;
        if (p->a_exgr) for ( ; h <= h9; ++h) if (h->val > mx->val) mx = h;
        if (p->b_exgr) {
            rw = imin(w->up, br - al) - cutlen;
            for (Rvp* hq = hh0 + rw; hq > h9; --hq) if (hq->val > mx->val) mx = hq;
        }
        const int i = (int) (mx - h9);
        int m9 = ar, n9 = br;
        if (i > 0) m9 -= i;
        if (i < 0) n9 += i;
        mx->ptr = vmf_add(&vmf, m9, n9, mx->ptr);
        scr = mx->val; ptr = mx->ptr;
    }
    /* trcbkalignS_ng: Vmf::traceback(ptr) -> records, plus the boundary fix-up */
    if (ptr) {
        int cap = 64, cnt = 0;
        SpdpSkl* out = (SpdpSkl*) malloc(cap * sizeof(SpdpSkl));
        Sklp sv = vmf.rec[ptr];
        for (;;) {
            if (cnt + 2 > cap) { cap *= 2; out = (SpdpSkl*) realloc(out, cap * sizeof(SpdpSkl)); }
            out[cnt].m = sv.m; out[cnt].n = sv.n; ++cnt;
            if (!sv.p) break;
            sv = vmf.rec[sv.p];
        }
        const int r = out[cnt - 1].n - out[cnt - 1].m;
        const int rd = Local ? 0 : (r - bl + al);
        if (rd > 0) { out[cnt].m = al; out[cnt].n = bl + rd; ++cnt; }
        else if (rd < 0) { out[cnt].m = al - rd; out[cnt].n = bl; ++cnt; }
        *skl = out; *n_skl = cnt;
    }
    free(jbuf); free(dbuf); free(vmf.rec);
    *score = scr;
    return 0;
}

int orc_scalar_forward(const SpdpScoring* sc, const SpdpProblem* p, const SpdpWindow* w,
                       int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    return scalar_forward_impl(sc, p, w, 0, 0, score, skl, n_skl);
}

/* trcbkalignS_ng(wdw, spj, mc) with a cut range: always the scalar engine (src/fwd2s1.cc:1674-1678) */
int orc_scalar_forward_cut(const SpdpScoring* sc, const SpdpProblem* p, const SpdpWindow* w, int cut_l, int cut_r,
                           int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    return scalar_forward_impl(sc, p, w, cut_l, cut_r, score, skl, n_skl);
}

/* ---- unidirectional Hirschberg, scalar ------------------------------------------------------
 * orc_scalar_udh   Aln2s1::hirschbergS_ng + hinitS_ng / hlastS_ng        src/fwd2s1.cc:762-1104, 701-760
 *                  with UdhIntermediate (lub = true)                      src/udh_intermediate.h:29-66
 * The -A0 linear-space engine: forwardS_ng's recurrence with every state carrying the diagonal
 * range it has visited since the last intermediate row (upr / lwr), the row it started on (ml) and a
 * link (ulk) to where its path crossed the previous intermediate row.  cpos[i] comes back as
 * lspS_ng reads it: [0] = mi, [1] = entered in a gap, [2..] = n coordinates of the horizontal run on
 * row mi, terminated by end_of_ulk, [8] / [9] = diagonal bounds of the slab below.  Entries the
 * reference leaves uninitialised (it allocates cpos with new[]) are end_of_ulk here.
 * rc -3: the reference would dereference udhimds[n_im] (a null pointer) at :1093. */
typedef struct { int val, upr, lwr, ml, ulk; } Rvwml;
typedef struct { int val, dir, upr, lwr, ml, ulk, jnc; } Rvdwmlj;
typedef struct { int mi; int* buf; int *hlnk[3], *vlnk[3], *lwrb[3], *uprb[3]; } UImd;    /* a plane per gap state: Noll of them */

int orc_scalar_udh(const SpdpScoring* sc, const SpdpProblem* p, const SpdpWindow* w, int n_im, int imd_intvl,
                   int32_t* score, int32_t* cpos, int32_t* ranges)
{
    if ((sc->noll != 2 && sc->noll != 3) || !sc->intpen || !p->cano5 || n_im < 1) return -1;
    const int NEV = SPDP_NEVSEL, EOU = SPDP_END_OF_ULK;
    int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int Local = sc->local;
    const int LocalL = Local && p->a_exgl && p->b_exgl;
    const int LocalR = Local && p->a_exgr && p->b_exgr;
    const int dim = sc->mtx_dim;
    const int lw = w->lw, up = w->up, width = w->width;
    const int dagp = sc->noll == 3, Nol = sc->noll, Nod = 2 * Nol - 1;
    const int GOP[3] = {0, sc->gop, sc->lgop};
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    for (int i = 0; i <= n_im; ++i) for (int c = 0; c < 10; ++c) CPOS(i, c) = EOU;
    const size_t bufsiz = (size_t) Nol * width;
    Rvwml* wbuf = (Rvwml*) malloc((bufsiz + 4) * sizeof(Rvwml));
    int r = bl - ar;
    const Rvwml black = {NEV, r, r, 0, EOU};
    for (size_t i = 0; i < bufsiz + 4; ++i) wbuf[i] = black;
    Rvwml* hh0 = wbuf - lw + 1;
    Rvwml* hh1 = hh0 + width;
    Rvwml* hh2 = hh1 + width;                           /* F2, Noll = 3 */
    /* hinitS_ng */
    {
        int rr = br - al;
        const int r0 = bl - al;
        r = r0;
        Rvwml* h = hh0 + r;
        h->val = 0; h->lwr = h->upr = h->ulk = r; h->ml = al;
        if (p->a_exgl) {
            if (up < rr) rr = up;
            while (++r <= rr) { ++h; h->val = 0; h->lwr = h->upr = h->ulk = r; h->ml = al; }
        }
        r = r0;
        rr = bl - ar;
        if (lw > rr) rr = lw;
        h = hh0 + r - 1;
        for (int i = 1; --r >= rr; ++i, --h) {
            if (p->b_exgl) { h->val = 0; h->lwr = h->upr = h->ulk = r; h->ml = h[1].ml + 1; }
            else {
                *h = h[1];
                ++h->ml;
                h->val += (i == 1) ? gap_penalty(sc, 1) : gap_ext_pen(sc, i);
                h->lwr = r;
                h->ulk = r0;
            }
        }
    }
    /* Udh_Imds(n_im, a->left, imd_intvl, wdw, Noll, lub = true) */
    UImd* imds = (UImd*) calloc(n_im, sizeof(UImd));
    {
        int mi = al;
        const size_t us = (size_t) Nol * width;
        for (int i = 0; i < n_im; ++i) {
            UImd* d = imds + i;
            d->mi = (mi += imd_intvl);
            d->buf = (int*) malloc(4 * us * sizeof(int));
            for (size_t k = 0; k < 2 * us; ++k) d->buf[k] = EOU;
            for (size_t k = 0; k < us; ++k) { d->buf[2 * us + k] = INT_MAX; d->buf[3 * us + k] = INT_MIN; }
            d->hlnk[0] = d->buf - lw + 1;  d->vlnk[0] = d->hlnk[0] + us;
            d->lwrb[0] = d->vlnk[0] + us;  d->uprb[0] = d->lwrb[0] + us;
            for (int k = 1; k < Nol; ++k) {
                d->hlnk[k] = d->hlnk[k - 1] + width; d->vlnk[k] = d->vlnk[k - 1] + width;
                d->lwrb[k] = d->lwrb[k - 1] + width; d->uprb[k] = d->uprb[k - 1] + width;
            }
        }
    }
    UImd* imd = imds;
    int mm = imd->mi;
    int rlst = INT_MAX;
    int maxh_val = NEV, maxh_upr = 0, maxh_lwr = 0, maxh_ml = al, maxh_ulk = 0, maxh_mr = ar, maxh_nr = br;
    int m = al;
    if (!p->a_exgl) --m;
    int n1 = m + lw, n2 = m + up + 1;
    for (int i = 0; ++m <= ar; ++n1, ++n2) {
        int n = imax(n1, bl);
        const int n9 = imin(n2, br);
        const int is_imd = m == mm;
        unsigned psp = 0;
        r = n - m;
        Rvwml *h = hh0 + r, *f = hh1 + r, *f2 = dagp ? hh2 + r : wbuf + bufsiz - 1;
        Rvwml e1 = black, e2 = black;
        Rvwml* hf[NOD] = {0, &e1, 0, &e2, 0};
        Rvdwmlj rcd[NCAND + 1];
        int idx[NCAND + 1];
        for (int l = 0; l <= NCAND; ++l) {
            rcd[l].val = NEV; rcd[l].dir = 0; rcd[l].upr = INT_MIN; rcd[l].lwr = INT_MAX;
            rcd[l].ml = 0; rcd[l].ulk = EOU; rcd[l].jnc = 0;
            idx[l] = l;
        }
        int ncand = -1;
        const int32_t* qprof = (m >= 1) ? sc->mtx + (size_t) p->a[m - 1] * dim : sc->mtx;
        for ( ; ++n <= n9; ) {
            int x;
            ++r; ++h; ++f; if (dagp) ++f2;
            hf[0] = h; hf[2] = f; hf[4] = f2;
            Rvwml* from = h;
            Rvwml* mx = h;
            if (m != al) {
                h->val += qprof[p->b[n - 1]];
                x = (++from)->val + sc->gop;
                if (x >= f[1].val) { *f = *from; f->val = x; }
                else *f = f[1];
                f->val += sc->gep;
                if (f->val >= mx->val) mx = f;
                if (dagp) {                                 /* Vertical2, :847-856 */
                    x = from->val + sc->lgop;
                    if (x >= f2[1].val) { *f2 = *from; f2->val = x; }
                    else *f2 = f2[1];
                    f2->val += sc->lgep;
                    if (f2->val >= mx->val) mx = f2;
                }
            }
            x = h[-1].val + sc->gop;
            const unsigned prev_psp = psp;
            if (x >= e1.val) { e1 = h[-1]; e1.val = x; psp = psp ? E1_PSP : 0; }
            else psp &= 3;                                  /* e_psp = e1_psp + e2_psp */
            e1.val += sc->gep;
            if (e1.val >= mx->val) mx = &e1;
            if (dagp) {                                     /* Horizontal2, :870-880 */
                x = h[-1].val + sc->lgop;
                if (x >= e2.val) { e2 = h[-1]; e2.val = x; if (prev_psp) psp |= E2_PSP; }
                else psp |= (prev_psp & E2_PSP);
                e2.val += sc->lgep;
                if (e2.val >= mx->val) mx = &e2;
            }
            int spj3 = 0;
            if (p->cano3[n]) {
                const Rvdwmlj* maxphl[NOD] = {0, 0, 0, 0, 0};
                for (int l = 0; l <= ncand; ++l) {
                    const Rvdwmlj* prd = rcd + idx[l];
                    if (n - prd->jnc < sc->llmt) continue;
                    from = hf[prd->dir];
                    x = prd->val + (p->cip ? p->cip[m] : 0) + spjscr(sc, p, prd->jnc, n);     /* sigB = cip_score(m) */
                    if (x > from->val) { from->val = x; maxphl[prd->dir] = prd; }
                }
                int maxk = Nod;
                for (int k = 0; k < Nod; ++k) {
                    const Rvdwmlj* prd = maxphl[k];
                    if (!prd) continue;
                    psp |= psp_bit[k];
                    if (!k) spj3 = 1;
                    from = hf[k];
                    from->upr = imax(prd->upr, r);
                    from->lwr = imin(prd->lwr, r);
                    from->ml = prd->ml;
                    from->ulk = prd->ulk;
                    if (from->val > mx->val) { maxk = k; mx = from; }
                }
                if (is_imd && maxk < Nod) {
                    const Rvdwmlj* phl = maxphl[maxk];
                    imd->hlnk[0][r] = phl->ulk;
                    mx->ulk = rlst = r;
                    if (maxk == 0) {
                        for (int c = 1, d = 1; c < Nol; ++c, d += 2) {
                            if ((phl = maxphl[d]) && hf[d]->val > mx->val + GOP[c]) {
                                hf[d]->ulk = r + c * width;
                                imd->hlnk[c][r] = phl->ulk;
                            }
                            if (maxphl[d + 1] && hf[d + 1]->val > mx->val + GOP[c]) hf[d + 1]->ulk = r + c * width;
                        }
                    }
                }
            }
            int hd = 0;
            if (h == mx) {
                if (LocalR && h->val > maxh_val) {
                    maxh_val = h->val; maxh_upr = h->upr; maxh_lwr = h->lwr; maxh_ml = h->ml; maxh_ulk = h->ulk;
                    maxh_mr = m; maxh_nr = n;
                }
            } else {
                while (mx != hf[++hd]) ;
                *h = *mx;
                if (h->upr < r) h->upr = r;
                if (h->lwr > r) h->lwr = r;
            }
            if (LocalL && h->val <= 0) { h->val = 0; h->ml = m; h->ulk = h->upr = h->lwr = r; }
            if (p->cano5[n]) {
                const int sigJ = p->sig5[n];
                for (int k = (mx == h) ? 0 : 1; k < Nod; ++k) {
                    from = hf[k];
                    if (psp & psp_bit[k]) continue;
                    if (k != hd) {
                        int y = mx->val;
                        if (hd == 0 || (k - hd) % 2) y += GOP[k / 2];
                        if (from->val <= y) continue;
                    }
                    x = from->val + sigJ;
                    int l = ncand < NCAND ? ++ncand : NCAND;
                    while (--l >= 0) {
                        if (x > rcd[idx[l]].val) { int t = idx[l]; idx[l] = idx[l + 1]; idx[l + 1] = t; }
                        else break;
                    }
                    if (++l < NCAND) {
                        Rvdwmlj* prd = rcd + idx[l];
                        prd->val = x; prd->jnc = n; prd->dir = k;
                        prd->upr = from->upr; prd->lwr = from->lwr; prd->ml = from->ml;
                        if (is_imd) {
                            if (k == 1) imd->hlnk[0][r] = rlst;
                            prd->ulk = r;
                        } else prd->ulk = from->ulk;
                    } else --ncand;
                }
            }
            if (is_imd) {
                if (hd == 0) rlst = r;
                else if (!spj3 && hd % 2) imd->hlnk[0][r] = rlst;
                for (int k = 0; k < Nol; ++k) {
                    Rvwml* g = hf[2 * k];
                    imd->vlnk[k][r] = g->ulk;
                    imd->lwrb[k][r] = imin(r, g->lwr);
                    imd->uprb[k][r] = imax(r, g->upr);
                    g->lwr = g->upr = r;
                    g->ulk = r + k * width;
                }
            }
        }
        if (is_imd && ++i < n_im) { imd = imds + i; mm = imd->mi; }
    }

    int rc = 0;
    const int rr = br - ar;
    if (LocalR) {
        int i = n_im;
        while (--i >= 0 && imds[i].mi > ar) ;
        ar = maxh_mr; br = maxh_nr;
        if (i < 0) i = 0;
        CPOS(i, 8) = maxh_lwr;
        CPOS(i, 9) = maxh_upr;
    } else {    /* hlastS_ng */
        Rvwml* h9 = hh0 + br - ar;
        Rvwml* mx = h9;
        if (p->b_exgr) { const int rw = imin(up, br - al); for (Rvwml* h = hh0 + rw; h > h9; --h) if (h->val > mx->val) mx = h; }
        if (p->a_exgr) { const int rw = imax(lw, bl - ar); for (Rvwml* h = hh0 + rw; h < h9; ++h) if (h->val > mx->val) mx = h; }
        maxh_val = mx->val; maxh_lwr = mx->lwr; maxh_upr = mx->upr; maxh_ulk = mx->ulk; maxh_ml = mx->ml;
        r = (int) (mx - hh0);
        if (p->b_exgr && rr < r) ar = br - r;
        if (p->a_exgr && rr > r) br = ar + r;
    }
    int i = n_im;
    while (--i >= 0 && imds[i].mi > ar) ;
    if (i < 0 && imds[0].mi > ar) CPOS(0, 2) = br;
    r = br - ar;
    CPOS(i + 1, 8) = imin(maxh_lwr, r);
    CPOS(i + 1, 9) = imax(maxh_upr, r);
    r = maxh_ulk;
    int d = 0;
    for ( ; i >= 0 && (imd = imds + i)->mi > maxh_ml; --i) {
        int c = 0;
        for (d = 0; r > up; r -= width) ++d;
        if (d > Nol - 1 || r < lw - 1) { rc = -3; break; }         /* outside the link arrays */
        if (imd->vlnk[d][r] < EOU) {
            CPOS(i, c++) = imd->mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            for (int rp = imd->hlnk[d][r]; lw <= rp && rp < up && r != rp; rp = imd->hlnk[0][r = rp]) {
                if (c >= 6) { rc = -3; break; }              /* the terminator would land on [8] */
                CPOS(i, c++) = r + imd->mi;
            }
            if (rc) break;
            CPOS(i, c++) = r + imd->mi;
            CPOS(i, c) = EOU;
            CPOS(i, 8) = imd->lwrb[d][r];
            CPOS(i, 9) = imd->uprb[d][r];
            r = imd->vlnk[d][r];
            if (r == EOU) break;
        } else
            CPOS(i, 0) = EOU;
    }
    if (!rc) {
        for ( ; r > up; r -= width) ;
        if (LocalL) { al = maxh_ml; bl = r + maxh_ml; }
        else {
            const int rl = bl - al;
            if (p->b_exgl && rl > r) {
                al = bl - r;
                for (int j = 0; j < n_im && imds[j].mi < al; ++j) CPOS(j, 0) = EOU;
            }
            if (p->a_exgl && rl < r) bl = al + r;
        }
        ++i;
        if (i >= n_im) rc = -3;
        else if (imds[i].mi < al || CPOS(i, 2) < bl) maxh_val = NEV;
        else if (CPOS(i, 8) == EOU || CPOS(i, 9) == EOU) rc = -3;   /* bounds the reference never set */
        else {
            const int rl = bl - al;
            CPOS(i, 8) = imin(rl, CPOS(i, 8));
            CPOS(i, 9) = imax(rl, CPOS(i, 9));
        }
    }
#undef CPOS
    *score = maxh_val;
    ranges[0] = al; ranges[1] = ar; ranges[2] = bl; ranges[3] = br;
    for (int j = 0; j < n_im; ++j) free(imds[j].buf);
    free(imds); free(wbuf);
    return rc;
}
