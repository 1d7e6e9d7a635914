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
#include <string.h>
#include <limits.h>
#include "../include/spdp.h"

#define NCAND 4
#define NOD 3                                   /* 2 * Noll - 1 for Noll = 2: DIAG, HORI, VERT */
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
    if (sc->noll != 2 || !sc->intpen || !p->cano5) return -1;
    const int NEV = SPDP_NEVSEL;
    const int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int LocalL = sc->local && p->a_exgl && p->b_exgl;
    const int LocalR = sc->local && p->a_exgr && p->b_exgr;
    const int dim = sc->mtx_dim;
    const size_t bufsiz = (size_t) 2 * w->width;
    int* wbuf = (int*) malloc(bufsiz * sizeof(int));
    for (size_t i = 0; i < bufsiz; ++i) wbuf[i] = NEV;
    int* hh0 = wbuf - w->lw + 1;
    int* hh1 = hh0 + w->width;
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
                if (i == 1) { *h += sc->gop + sc->gep; *f = *h; }
                else { *f = f[1]; *h += sc->gep; *f += sc->gep; }
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
        int *h = hh0 + r, *f = hh1 + r;
        int e1 = NEV;
        int* hf[NOD] = {0, &e1, 0};
        Cand rcd[NCAND + 1];
        int idx[NCAND + 1];
        for (int l = 0; l <= NCAND; ++l) { rcd[l].val = NEV; rcd[l].dir = rcd[l].jnc = rcd[l].ptr = 0; idx[l] = l; }
        int ncand = -1;
        const int32_t* qprof = (m >= 1) ? sc->mtx + (size_t) p->a[m - 1] * dim : sc->mtx;
        for ( ; ++n <= n9; ) {
            int x;
            ++h; ++f;
            hf[0] = h; hf[2] = f;
            int* from = h;
            int* mx = h;
            if (m != al) {
                *h += qprof[p->b[n - 1]];
                x = *++from + sc->gop;
                *f = imax(x, f[1]) + sc->gep;
                if (*f > *mx) mx = f;
            }
            x = h[-1] + sc->gop;
            if (x > e1) { e1 = x; psp = psp ? E1_PSP : 0; }
            else psp &= E1_PSP;
            e1 += sc->gep;
            if (e1 > *mx) mx = &e1;
            if (p->cano3[n]) {
                const Cand* maxphl[NOD] = {0, 0, 0};
                for (int l = 0; l <= ncand; ++l) {
                    const Cand* prd = rcd + idx[l];
                    if (n - prd->jnc < sc->llmt) continue;
                    from = hf[prd->dir];
                    x = prd->val + spjscr(sc, p, prd->jnc, n);
                    if (x > *from) { *from = x; maxphl[prd->dir] = prd; }
                }
                for (int k = 0; k < NOD; ++k) {
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
                for (int k = hd == 0 ? 0 : 1; k < NOD; ++k) {
                    from = hf[k];
                    if (psp & psp_bit[k]) continue;
                    if (k != hd) {
                        y = *mx;
                        if (hd == 0 || (k - hd) % 2) y += (k / 2 == 1) ? sc->gop : 0;   /* GOP[k/2] */
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

enum { DIAG = 2, NEWD = 3 };    /* TraceBackDir, src/aln.h:30-35: DEAD, RSRV, DIAG, NEWD, ... */

/* returns the records trcbkalignS_ng hands to the Mfile (end -> start); caller frees *skl */
int orc_scalar_forward(const SpdpScoring* sc, const SpdpProblem* p, const SpdpWindow* w,
                       int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    *skl = 0; *n_skl = 0;
    if (sc->noll != 2 || !sc->intpen || !p->cano5) return -1;
    if (w->width < 0) { *score = SPDP_NEVSEL; return 0; }
    const int NEV = SPDP_NEVSEL;
    const int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int Local = sc->local;
    const int LocalL = Local && p->a_exgl && p->b_exgl;
    const int LocalR = Local && p->a_exgr && p->b_exgr;
    const int spj = sc->spj;
    const int dim = sc->mtx_dim;
    const int width = w->width;
    const size_t bufsiz = (size_t) 2 * width;
    Rvp* jbuf = (Rvp*) malloc(bufsiz * sizeof(Rvp));
    for (size_t i = 0; i < bufsiz; ++i) { jbuf[i].val = NEV; jbuf[i].ptr = 0; }
    Rvp* hh0 = jbuf - w->lw + 1;
    Rvp* hh1 = hh0 + width;
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
            while (++r <= rr) { ++h; h->val = 0; *++hd = 1; h->ptr = 0; }
        }
        r = bl - al; rr = bl - ar;
        h = hh0 + r; hd = hdir + r;
        if (w->lw > rr) rr = w->lw;
        for (int i = 1; --r >= rr; ++i) {
            --h; *--hd = 2;
            if (p->b_exgl) { h->val = 0; h->ptr = 0; }
            else {
                *h = h[1];
                if (i == 1) h->val += sc->gop + sc->gep;
                else h->val += sc->gep;             /* GapExtPen(i) */
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
        Rvp *h = hh0 + r, *f = hh1 + r;
        unsigned char* dir = hdir + r;
        unsigned psp = 0;
        Rvp e1 = {NEV, 0};
        Rvp* hf[NOD] = {0, &e1, 0};
        Cand rcd[NCAND + 1];
        int idx[NCAND + 1];
        for (int l = 0; l <= NCAND; ++l) { rcd[l].val = NEV; rcd[l].ptr = rcd[l].dir = rcd[l].jnc = 0; idx[l] = l; }
        int ncand = -1;
        const int32_t* qprof = (m >= 1) ? sc->mtx + (size_t) p->a[m - 1] * dim : sc->mtx;
        for ( ; ++n <= n9; ) {
            int x;
            ++dir; ++h; ++f;
            hf[0] = h; hf[2] = f;
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
            }
            x = h[-1].val + sc->gop;
            if (x >= e1.val) { e1.val = x; e1.ptr = h[-1].ptr; psp = psp ? E1_PSP : 0; }
            else psp &= E1_PSP;
            e1.val += sc->gep;
            if (e1.val >= mx->val) mx = &e1;
            if (internal && p->cano3[n]) {
                const Cand* maxphl[NOD] = {0, 0, 0};
                for (int l = 0; l <= ncand; ++l) {
                    const Cand* prd = rcd + idx[l];
                    if (n - prd->jnc < sc->llmt) continue;
                    x = prd->val + spjscr(sc, p, prd->jnc, n);
                    from = hf[prd->dir];
                    if (x >= from->val) { from->val = x; maxphl[prd->dir] = prd; }
                }
                for (int k = 0; k < NOD; ++k) {
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
                for (int k = hd == 0 ? 0 : 1; k < NOD; ++k) {
                    from = hf[k];
                    if (psp & psp_bit[k]) continue;
                    if (k != hd) {
                        int z = mx->val;
                        if (hd == 0 || (k - hd) % 2) z += (k / 2 == 1) ? sc->gop : 0;    /* GOP[k/2] */
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
        }
    }
    int ptr = 0, scr;
    if (LocalR) { ptr = vmf_add(&vmf, maxh_m, maxh_n, maxh_p); scr = maxh_val; }
    else {      /* lastS_ng */
        int rw = w->lw;
        int rf = bl - ar;
        if (rf > rw) rw = rf;
        Rvp* h = hh0 + rw;
        Rvp* h9 = hh0 + br - ar;
        Rvp* mx = h9;
        if (p->a_exgr) for ( ; h <= h9; ++h) if (h->val > mx->val) mx = h;
        if (p->b_exgr) {
            rw = imin(w->up, br - al);
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
