/* spdp_oracle_h_scalar.c -- CPU restatement of the reference's SCALAR protein x genome engine.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as spdp_oracle.c): tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it as the checker; the product never links or calls it.
 *
 * Restates (ogotoh/spaln v3.0.7):
 *   orc_scalar_forward_h   Aln2h1::forwardH_ng                        src/fwd2h1.cc:294-617
 *                          + initH_ng / lastH_ng                      src/fwd2h1.cc:143-208 / 210-292
 *                          + Vmf::traceback                           src/vmf.cc:125-140
 *                          + the record fix-up of trcbkalignH_ng      src/fwd2h1.cc:2019-2036
 *   orc_scalar_udh_h       Aln2h1::hirschbergH_ng                     src/fwd2h1.cc:1085-1520
 *   orc_exact_forward_h    SimdAln2h1::forwardH1   (-A1)              src/fwd2h1_simd.h:820-1096   } second half
 *   orc_exact_udh_h        SimdAln2h1::hirschbergH1 (-A1)             src/fwd2h1_simd.h:1100-1470  } of this file
 * Pinned: alignH_ng over these engines equals the compiled reference's -A0 / -A1 records on every protein
 * fixture (tests/test_oracle_h_golden.py, tests/test_oracle_h_scalar.py).
 * This is the -A0 engine (int32, row by row, exact intron-length penalty with the top-NCAND donor
 * list per row and codon phase) -- also what the -A2/-A3 dispatch falls back to for sub-problems
 * with fewer than 8 query rows (trcbkalignH_ng src/fwd2h1.cc:2005, HomScoreH_ng :3297).
 * Affine gaps (Noll = 2) and double affine gaps (Noll = 3, -yl3: SpdpScoringH.noll; round 5).  Junction terms:
 *   spjscr(jnc, n) = IntPen(n - jnc) + sig3[n] + T53[16 * dinc5[jnc] + dinc3[n]]
 *     (SpJunc::spjscr src/codepot.cc:74-77, Exinon::sig53 IE53 src/codepot.cc:411-415)
 *   spjseq(jnc, n): the two codons the four bases around an intron spell (src/codepot.cc:79-107),
 *     rebuilt here from the standard genetic code; a word with an ambiguous base reads as (AMB, AMB).
 * Positions outside the sequences read the Seq padding (amb_code, Seq::fillpad src/seq.cc:491-496).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "../include/spdp.h"

#define NCAND 4
#define NOD 3                                   /* 2 * Noll - 1 for Noll = 2: DIAG, HORI, VERT */
#define NQUE 3
#define AMB 2

/* TraceBackDir, src/aln.h:30-35 */
enum { DEAD, RSRV, DIAG, NEWD, VERT, SLA1, SLA2, VERL, HORI, HOR1, HOR2, HORL, NEWV, NEWH, SPIN = 16 };
static const int dir2nod[16] = {-1, -1, 0, 0, 2, 2, 2, 4, 1, 1, 1, 3, 2, 1, -1, -1};    /* src/aln.h:50 */
static const int nod2dir[5] = {DIAG, HORI, VERT, HORL, VERL};
static const char is_diag[16] = {0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      /* src/aln.h:67-69 */
static const char is_vert[16] = {0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0};
static const char is_hori[16] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 0};

typedef struct { int val, ptr, dir; } Rvpd;
typedef struct { int val, ptr, dir, jnc; } Rvpdj;
typedef struct { int m, n, p; } Sklp;
typedef struct { Sklp* rec; int n, cap, on; } Vmf;

static int vmf_add(Vmf* v, int m, int n, int p)
{
    if (!v->on) return 0;
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->rec = (Sklp*) realloc(v->rec, v->cap * sizeof(Sklp)); }
    v->rec[v->n].m = m; v->rec[v->n].n = n; v->rec[v->n].p = p;
    return v->n++;
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* genetic code -> tron codes (aa codes 3.. in the order ARNDCQEGHILKMFPSTWYV; AGY serine 23, TGA 24,
 * TAA / TAG 25) and the middle base (A C G T = 0..3) each tron code pins */
static uint8_t g_tron_of[64];                   /* index 16 * b1 + 4 * b2 + b3, bases A C G T */
static uint8_t g_mid[32];
static int g_tabs = 0;
static void code_tables(void)
{
    if (g_tabs) return;
    static const char* aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";   /* T C A G order */
    static const char* order = "ARNDCQEGHILKMFPSTWYV";
    static const int tcag[4] = {3, 1, 0, 2};    /* T C A G as A C G T indices */
    memset(g_mid, 4, sizeof g_mid);
    for (int c = 0; c < 64; ++c) {
        const int b1 = tcag[c >> 4], b2 = tcag[(c >> 2) & 3], b3 = tcag[c & 3];
        const char aa = aas[c];
        int code;
        if (aa == '*') code = (b1 == 3 && b2 == 2 && b3 == 0) ? 24 : 25;
        else if (aa == 'S' && b1 == 0) code = 23;
        else code = 3 + (int) (strchr(order, aa) - order);
        g_tron_of[16 * b1 + 4 * b2 + b3] = (uint8_t) code;
        g_mid[code] = (uint8_t) b2;
    }
    g_tabs = 1;
}

typedef struct {
    const SpdpScoringH* sc;
    const SpdpProblemH* p;
    int minl;
} Ctx;

static inline int a_code(const Ctx* c, int i) { return i < 0 ? AMB : (i >= c->p->a_len ? c->p->a_pad : c->p->a[i]); }
static inline int b_code(const Ctx* c, int i) { return (i < 0 || i > c->p->b_len) ? AMB : c->p->b[i]; }
static inline int mtx_at(const Ctx* c, int aa, int tron) { return c->sc->mtx[aa * c->sc->mtx_cols + tron]; }
static inline int gap_ext3(const SpdpScoringH* sc, int i) { return i > sc->codonk1 ? sc->lgep : sc->gep; }
static inline int intpen(const SpdpScoringH* sc, int len)
{
    if (len < 0) return SHRT_MIN;
    if (len >= sc->intpen_len) len = sc->intpen_len - 1;
    return sc->intpen[len];
}
static inline int spjscr(const Ctx* c, int jnc, int n)
{
    const SpdpProblemH* p = c->p;
    return intpen(c->sc, n - jnc) + p->sig3[n] + c->sc->t53[16 * (p->dinc[jnc] >> 4) + (p->dinc[n] & 15)];
}
static void spjseq(const Ctx* c, int n5, int n3, int cs[2])
{
    const SpdpProblemH* p = c->p;
    cs[0] = cs[1] = AMB;
    if (n5 < p->b_left || n3 >= p->b_right) return;
    const int idx[4] = {n5 - 2, n5 - 1, n3, n3 + 1};
    int w[4];
    for (int i = 0; i < 4; ++i) {
        const int t = b_code(c, idx[i]);
        w[i] = (t >= 32) ? 4 : g_mid[t];
    }
    /* an ambiguous second or third base: neither codon; an ambiguous first (last) one leaves the second (first)
     * codon standing (spj_amb_tron_tab / spj_tron_amb_tab, src/codepot.cc:84-106) */
    if (w[1] > 3 || w[2] > 3) return;
    if (w[0] <= 3) cs[0] = g_tron_of[16 * w[0] + 4 * w[1] + w[2]];
    if (w[3] <= 3) cs[1] = g_tron_of[16 * w[1] + 4 * w[2] + w[3]];
}
/* test hook: SpJunc::spjseq(n5, n3) as the engines of this file see it */
void orc_spjseq_h(const SpdpProblemH* p, int n5, int n3, int32_t cs[2])
{
    Ctx cx; memset(&cx, 0, sizeof cx);
    cx.p = p;
    code_tables();
    int c2[2];
    spjseq(&cx, n5, n3, c2);
    cs[0] = c2[0]; cs[1] = c2[1];
}

/* rc 0 ok; -1 unsupported parameters.  skl / n_skl may be NULL (score only: no Vmf, as HomScoreH_ng
 * runs it).  With a traceback the records come back end -> start as trcbkalignH_ng writes them. */
/* cut_r > cut_l: forwardH_ng's cut range (`cutrng`, src/fwd2h1.cc:294-312, 589-603; lastH_ng :210-281): after column cut_l
 * the three insertion states, charged for the codons of the cut, replace the last three entries of the diagonal arrays
 * and the sweep goes on behind the cut (shortcutH_ng, :2232-2260) */
static int scalar_forward_h_impl(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w, int cut_l, int cut_r,
                         int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    if (skl) { *skl = 0; *n_skl = 0; }
    const int has_cut = cut_r > cut_l;
    const int cutlen = has_cut ? cut_r - cut_l : 0;
    if (!sc->intpen || !p->dinc) return -1;
    if (w->width < 0) { *score = SPDP_NEVSEL; return 0; }
    code_tables();
    const int minl = sc->minl ? sc->minl : sc->llmt;
    Ctx cx = {sc, p, minl};
    const int NEV = SPDP_NEVSEL;
    const Rvpd black = {NEV, 0, 0};
    const Rvpdj blackj = {NEV, 0, 0, 0};
    const int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int Local = sc->local;
    const int LocalL = Local && p->a_exgl && p->b_exgl;
    const int LocalR = Local && p->a_exgr && p->b_exgr;
    const int spj = sc->spj;
    const int lw = w->lw, up = w->up, width = w->width - cutlen;
    if (width < 7) { *score = SPDP_NEVSEL; return -1; }
    /* double affine gaps (PwdB::Noll = 3, -yl3; src/fwd2h1.cc:297, 343, 365, 413-424, 437-449, 577, 591-598): a second
     * vertical (F2, by diagonal like F) and a second insertion state (E2, a queue like E1) priced with GapW3L / LongGEP,
     * five states a candidate can leave from */
    const int dagp = sc->noll == 3;
    const int nod = dagp ? 5 : 3;
    const int gapw3l = sc->lgop + sc->lgep;            /* PwdB::GapW3L, src/aln2.cc:124 */
    const int GOP[3] = {0, sc->gop, sc->lgop};          /* PwdB::GOP, src/aln2.cc:111 */
    const size_t bufsiz = (size_t) (dagp ? 3 : 2) * width;
    Rvpd* buf = (Rvpd*) malloc((bufsiz + 8) * sizeof(Rvpd));
    for (size_t i = 0; i < bufsiz + 8; ++i) buf[i] = black;
    Rvpd* hh0 = buf - lw + 3;
    Rvpd* hh1 = hh0 + width;
    Rvpd* hh2 = hh1 + width;                            /* F2 (Noll = 3 only) */
    Vmf vmf = {0, 0, 0, skl != 0};
    vmf_add(&vmf, 0, 0, 0);                     /* skip 0-th record */

    /* ---- initH_ng ---- */
    {
        int n = bl;
        int r = bl - 3 * al;
        int rr = br - 3 * al;
        const int dir = p->a_exgl ? DEAD : DIAG;
        int jnc[3] = {n, 0, 0};
        int bb = n + 1;
        Rvpd* h = hh0 + r;
        h->val = (p->a_exgl && p->sigS[bb] > 0) ? p->sigS[bb] : 0;
        h->dir = dir;
        h->ptr = vmf_add(&vmf, al, n, 0);
        if (p->a_exgl) {
            if (up < rr) rr = up;
            for (int i = 1; ++r <= rr; ++i) {
                ++h; ++bb; ++n;
                if (i < 3) {
                    h->val = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
                    h->dir = dir;
                    h->ptr = vmf_add(&vmf, al, n, 0);
                    jnc[i] = n;
                } else {
                    *h = h[-3];
                    const int k = n - jnc[i % 3];
                    if (k == 3 && !(p->a_exgl & 1)) h->val += sc->gop;
                    if (!(p->a_exgl & 2)) h->val += gap_ext3(sc, k);
                    h->val += p->sigE[bb - 3];
                    h->dir = HORI;
                    int x = h[-1].val + sc->gapw1;
                    if (x > h->val) { *h = h[-1]; h->val = x; h->dir = HOR1; }
                    x = h[-2].val + sc->gapw2;
                    if (x > h->val) { *h = h[-2]; h->val = x; h->dir = HOR2; }
                }
                const int x = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
                if (h->val < x) {
                    h->val = x;
                    h->dir = DEAD;
                    h->ptr = vmf_add(&vmf, al, n, 0);
                    jnc[i % 3] = n;
                }
            }
        }
        r = bl - 3 * al;
        rr = bl - 3 * ar;
        h = hh0 + r - 1;
        if (lw > rr) rr = lw;
        for (int i = 1; --r >= rr; ++i, --h) {
            if (p->b_exgl == 1) { h->val = 0; h->dir = DEAD; h->ptr = 0; }
            else if (i <= 3) {
                *h = h[i];
                if (!(p->b_exgl & 2)) h->val += sc->gep;
                if (!(p->b_exgl & 1)) h->val += sc->gop;
                if (i < 3) h->val += sc->extragop;
                h->dir = VERT;
            } else {
                *h = h[3];
                if (!(p->b_exgl & 2)) h->val += gap_ext3(sc, i);
            }
        }
    }

    int maxh_val = NEV, maxh_m = al, maxh_n = bl, maxh_p = 0;
    int m = al;
    if (!p->a_exgl) --m;
    int n1 = 3 * m + lw - 1;
    int n2 = 3 * m + up;
    for ( ; ++m <= ar; ) {
        n1 += 3; n2 += 3;
        const int n0 = imax(n1, bl);
        const int n9 = imin(n2, br);
        int n = n0;
        int r = n - 3 * m;
        Rvpd e1[NQUE] = {black, black, black};
        Rvpd e2[NQUE] = {black, black, black};
        if (!p->b_exgl && m == al) { e1[2] = e2[2] = hh0[r]; e1[2].val = sc->gapw3; e2[2].val = gapw3l; }
        Rvpd* h = hh0 + r;
        Rvpd* f = hh1 + r;
        Rvpd* f2 = dagp ? hh2 + r : 0;
        Rvpd* hf[5] = {h, 0, f, 0, f2};
        const int aa0 = a_code(&cx, m - 1), aa1 = a_code(&cx, m);
        Rvpdj hl[3][NCAND + 1];
        int nx[3][NCAND + 1];
        for (int ph = 0; ph < 3; ++ph)
            for (int l = 0; l <= NCAND; ++l) { hl[ph][l] = blackj; nx[ph][l] = l; }
        int ncand[3] = {-1, -1, -1};
        for (int q = 0; n <= n9; ++n, ++h, ++f, f2 += dagp) {
            int x, y;
            hf[0] = h; hf[2] = f; hf[4] = f2;
            const int sigE = (n > bl && n >= 2) ? p->sigE[n - 2] : 0;       /* position -1 is not in the arrays */
            Rvpd* const eq1 = hf[1] = e1 + q;
            Rvpd* const eq2 = hf[3] = dagp ? e2 + q : 0;
            const Rvpd hq = *h;                 /* previous state */
            Rvpd* from = h;
            Rvpd* mx = h;
            if (m != al) {
                /* diagonal match */
                if (n < bl + 3) *h = black;
                else {
                    h->val += mtx_at(&cx, aa0, b_code(&cx, n - 2)) + sigE;
                    h->dir = is_diag[from->dir & 15] ? DIAG : NEWD;
                }
                /* vertical gap extension, 1 / 2 nt deletions, codon deletion */
                y = f[3].val + sc->gep;
                ++from;
                x = from->val + (is_vert[from->dir & 15] ? sc->gape1 : sc->gapw1);
                if (x > y) { f->val = x; f->dir = SLA2; f->ptr = from->ptr; }
                else f->val = y;
                ++from;
                x = from->val + (is_vert[from->dir & 15] ? sc->gape2 : sc->gapw2);
                if (x > f->val) { f->val = x; f->dir = SLA1; f->ptr = from->ptr; }
                x = (++from)->val + sc->gapw3;
                if (x >= f->val) { f->val = x; f->dir = VERT; f->ptr = from->ptr; }
                else if (y >= f->val) { f->val = y; f->dir = VERT; f->ptr = f[3].ptr; }
                if (f->val > mx->val) mx = f;
                if (dagp) {                     /* long deletion */
                    x = from->val + gapw3l;
                    y = f2[3].val + sc->lgep;
                    if (x >= y) { f2->val = x; f2->dir = VERL; f2->ptr = from->ptr; }
                    else { *f2 = f2[3]; f2->val = y; }
                    if (f2->val > mx->val) mx = f2;
                }
            }
            /* insertions of a codon, 2 nt, 1 nt */
            if (n > n0 + 2) {
                from = h - 3;
                x = from->val + sc->gapw3;
                y = eq1->val += sc->gep;
                if (x > y) { *eq1 = *from; eq1->val = x; }
                eq1->val += sigE;
                eq1->dir = (eq1->dir & SPIN) + HORI;
                if (dagp) {                     /* long insertion */
                    x = from->val + gapw3l;
                    y = eq2->val += sc->lgep;
                    if (x > y) { *eq2 = *from; eq2->val = x; }
                    eq2->val += sigE;
                    eq2->dir = (eq2->dir & SPIN) + HORL;
                    if (eq2->val > mx->val) mx = eq2;
                }
            }
            if (n > n0 + 1) {
                from = h - 2;
                x = from->val + sc->gapw2;
                if (x > eq1->val) { *eq1 = *from; eq1->val = x; eq1->dir = (eq1->dir & SPIN) + HOR2; }
            }
            from = h - 1;
            x = from->val + sc->gapw1;
            if (x > eq1->val) { *eq1 = *from; eq1->val = x; eq1->dir = (eq1->dir & SPIN) + HOR1; }
            if (eq1->val > mx->val) mx = eq1;
            if (++q == NQUE) q = 0;

            /* intron 3' boundary */
            if (spj && p->phs3[n] > -2) {
                int phs = (p->phs3[n] == 2) ? -1 : p->phs3[n];
                for (;;) {
                    const int nb = n - phs;
                    const int* pnx = nx[phs + 1];
                    const Rvpdj* maxphl[5] = {0, 0, 0, 0, 0};
                    for (int l = 0; l <= ncand[phs + 1]; ++l) {
                        const Rvpdj* phl = hl[phs + 1] + pnx[l];
                        if (phs == 1 && phl->dir == 2) continue;
                        if (nb - phl->jnc < minl) continue;
                        x = phl->val + (p->cip ? p->cip[3 * m - phs] : 0) + spjscr(&cx, phl->jnc, nb);   /* sigB[phs] = cip_score(3 m - phs) */
                        if (phl->dir == 0 && phs) {
                            int cs[2];
                            spjseq(&cx, phl->jnc, nb, cs);
                            if (phs == 1) x += mtx_at(&cx, aa0, cs[0]);
                            else x += mtx_at(&cx, aa1, cs[1]) - mtx_at(&cx, aa1, b_code(&cx, n + 1)) - p->sigE[n + 1];
                        }
                        from = hf[phl->dir];
                        if (x > from->val) { from->val = x; maxphl[phl->dir] = phl; }
                    }
                    for (int d = 0; d < nod; ++d) {
                        const Rvpdj* phl = maxphl[d];
                        if (!phl) continue;
                        from = hf[d];
                        if (vmf.on) {
                            const int inner = vmf_add(&vmf, m, phl->jnc + phs, phl->ptr);
                            from->ptr = vmf_add(&vmf, m, n, inner);
                        }
                        from->dir = nod2dir[phl->dir] | SPIN;
                        if (from->val > mx->val) mx = from;
                    }
                    if (p->phs3[n] - phs == 3) { phs = 1; continue; }   /* AGAG */
                    break;
                }
            }

            /* optimal path */
            y = h->val;
            if (h != mx) *h = *mx;
            else if (Local && y > hq.val) {
                if (LocalL && hq.dir == 0 && !(h->dir & SPIN)) h->ptr = vmf_add(&vmf, m - 1, n - 3, 0);
                else if (LocalR && y > maxh_val) { maxh_val = y; maxh_p = h->ptr; maxh_m = m; maxh_n = n; }
            }
            if (LocalL && h->val <= 0) h->val = h->dir = 0;
            else if (vmf.on && h->dir == NEWD) h->ptr = vmf_add(&vmf, m - 1, n - 3, h->ptr);

            /* intron 5' boundary */
            if (spj && p->phs5[n] > -2) {
                int phs = (p->phs5[n] == 2) ? -1 : p->phs5[n];
                for (;;) {
                    const int nb = n - phs;
                    const int sigJ = p->sig5[nb];
                    const int hd = dir2nod[mx->dir & 15];
                    for (int k = (hd == 0 || phs == 1) ? 0 : 1; k < nod; ++k) {
                        const int crossspj = phs == 1 && k == 0;
                        const Rvpd* src = crossspj ? &hq : hf[k];
                        if (!src->dir || (src->dir & SPIN)) continue;        /* no orphan exon */
                        if (!crossspj && k != hd && hd >= 0) {
                            y = mx->val;
                            if (hd == 0 || (k - hd) % 2) y += GOP[k / 2];
                            if (src->val <= y) continue;                     /* prune */
                        }
                        x = src->val + sigJ;
                        Rvpdj* phl = hl[phs + 1];
                        int* pnx = nx[phs + 1];
                        int* nc = &ncand[phs + 1];
                        int l = *nc < NCAND ? ++*nc : NCAND;
                        while (--l >= 0) {
                            if (x >= phl[pnx[l]].val) { const int t = pnx[l]; pnx[l] = pnx[l + 1]; pnx[l + 1] = t; }
                            else break;
                        }
                        if (++l < NCAND) {
                            phl += pnx[l];
                            phl->val = x; phl->jnc = nb; phl->dir = k; phl->ptr = src->ptr;
                        } else --*nc;
                    }
                    if (p->phs5[n] - phs == 3) { phs = 1; continue; }   /* GTGT */
                    break;
                }
            }
            if (has_cut && n == cut_l) {                /* shortcut: the insertion states run on over the cut */
                h -= 3; f -= 3;
                if (dagp) f2 -= 3;
                for (int pp = 0; pp < 3; ++pp) {
                    if (++q == 3) q = 0;
                    e1[q].val += sc->gep * cutlen / 3;
                    if (dagp) e2[q].val += sc->lgep * cutlen / 3;
                    *++h = dagp ? e2[q] : e1[q];
                    *++f = black;
                    if (dagp) *++f2 = black;
                }
                n += cutlen;
            }
        }
    }

    int ptr = 0, scr;
    if (!LocalR || maxh_m == ar) {              /* ---- lastH_ng ---- */
        static const int next_p[3] = {1, 2, 0};
        int glen[3] = {0, 0, 0};
        int rw = lw;
        const int m3 = 3 * ar;
        int rf = (has_cut ? cut_l : bl) - m3;
        if (rf > rw) rw = rf; else rf = rw;
        Rvpd* h = hh0 + rw - cutlen;
        Rvpd* h9 = hh0 + br - m3 - cutlen;
        Rvpd* mx = h9;
        int bb = rw + m3;
        int done = 0;
        if (p->a_exgr) {
            for (int ph = 0; h <= h9; ++h, ++bb, ++rf, ph = next_p[ph]) {
                glen[ph] += 3;
                int cand[3] = {h->val, NEV, NEV};
                if (rf - rw >= 3 && h[-3].dir != DEAD) {
                    cand[1] = h[-3].val + p->sigE[bb - 2];
                    if (!(p->a_exgr & 2)) cand[1] += gap_ext3(sc, glen[ph]);
                    if (!(p->a_exgr & 1) && glen[ph] == 3) cand[1] += sc->gop;
                    if (p->sigT[bb - 2] > 0 && !(h->dir & SPIN)) cand[2] = h[-3].val + p->sigT[bb - 2];
                }
                const int sig5 = (Local && p->sig5[bb] > 0) ? p->sig5[bb] : 0;
                cand[0] += sig5;
                cand[1] += sig5;
                int k = 0;
                if (cand[1] > cand[k]) k = 1;
                if (cand[2] > cand[k]) k = 2;
                if (k == 0) { if (!is_hori[h->dir & 15]) glen[ph] = 0; }
                else if (k == 1) { *h = h[-3]; h->dir = HORI; h->val = cand[k] - sig5; }
                else {
                    *h = h[-3];
                    h->dir = DEAD;
                    h->val = cand[k];
                    if (h->val > mx->val && vmf.on) h->ptr = vmf_add(&vmf, ar, rf + m3 - 3, h->ptr);
                }
                if (h->val > mx->val) mx = h;
            }
        } else {
            bb += (int) (h9 - h);
            const int y = h9[-3].val + p->sigT[bb - 2];
            if (y > h9->val) { *h9 = h9[-3]; h9->val = y; h9->dir = HORI; }
        }
        if (p->b_exgr == 1) {
            rw = imin(up, br - 3 * al) - cutlen;
            int g[3] = {NEV, NEV, NEV};
            h = hh0 + rw - 3;
            for (int ph = 0; h >= h9; --h) {
                int x = h[3].val;
                if (!(p->b_exgr & 1)) x += sc->gop;
                if (x > g[ph]) g[ph] = x;
                if (!(p->b_exgr & 2)) g[ph] += sc->gep;
                if (h->val > g[ph]) g[ph] = NEV;
                else if (g[ph] > mx->val) { mx = h; mx->val = g[ph]; }
                if (++ph == 3) ph = 0;
            }
        } else if (p->b_exgr == 2) {
            mx = hh1 + br - m3 - cutlen;
            mx->ptr = vmf_add(&vmf, ar, br, mx->ptr);
            done = 1;
        }
        if (!done) {
            int pp = (int) (mx - h9);
            rf = ar;
            rw = br;
            if (pp > 0) { rf -= (pp + 2) / 3; if (pp %= 3) rw -= (3 - pp); }
            else if (pp < 0) rw += pp;
            mx->ptr = vmf_add(&vmf, rf, rw, mx->ptr);
        }
        scr = mx->val;
        ptr = mx->ptr;
    } else {
        scr = maxh_val;
        ptr = vmf_add(&vmf, maxh_m, maxh_n, maxh_p);
    }

    /* trcbkalignH_ng: Vmf::traceback(ptr) -> records, plus the boundary fix-up */
    if (skl && ptr) {
        int cap = 64, cnt = 0;
        SpdpSkl* out = (SpdpSkl*) malloc(cap * sizeof(SpdpSkl));
        Sklp sv = vmf.rec[ptr];
        for (;;) {
            if (cnt + 2 > cap) { cap *= 2; out = (SpdpSkl*) realloc(out, cap * sizeof(SpdpSkl)); }
            out[cnt].m = sv.m; out[cnt].n = sv.n; ++cnt;
            if (!sv.p) break;
            sv = vmf.rec[sv.p];
        }
        const int r = out[cnt - 1].n - 3 * out[cnt - 1].m;
        const int rd = Local ? 0 : (r - bl + 3 * al);
        if (rd > 0) { out[cnt].m = al; out[cnt].n = bl + rd; ++cnt; }
        else if (rd < 0) { out[cnt].m = al - rd / 3; out[cnt].n = bl; ++cnt; }
        *skl = out; *n_skl = cnt;
    }
    free(buf); free(vmf.rec);
    *score = scr;
    return 0;
}

int orc_scalar_forward_h(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w,
                         int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    return scalar_forward_h_impl(sc, p, w, 0, 0, score, skl, n_skl);
}
int orc_scalar_forward_h_cut(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w, int cut_l, int cut_r,
                             int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    return scalar_forward_h_impl(sc, p, w, cut_l, cut_r, score, skl, n_skl);
}

/* ---- unidirectional Hirschberg, scalar -------------------------------------------------------
 * orc_scalar_udh_h   Aln2h1::hirschbergH_ng + hinitH_ng / hlastH_ng     src/fwd2h1.cc:1085-1520, 941-1083
 *                    with UdhIntermediate (lub = true)                   src/udh_intermediate.h:29-66
 * forwardH_ng's recurrence with every state carrying the diagonal range visited since the last
 * intermediate row (upr / lwr), its start row (ml) and a link (ulk) to where it crossed the previous
 * intermediate.  Not the same boundary rules as forwardH_ng: one `jnc` for all frames in the leading
 * gap, termination codons gated by algmode.lcl & 2, `(dir & DIAG)` as the diagonal-continuation test,
 * `>=` where the forward engine has `>` -- all as in the reference.  cpos rows as lspH_ng reads them;
 * entries the reference leaves uninitialised are end_of_ulk.  rc -3: the reference indexes outside
 * its arrays on this input. */
typedef struct { int val, dir, upr, lwr, ml, ulk; } Rvdwml;
typedef struct { int val, dir, upr, lwr, ml, ulk, jnc; } Rvdwmlj;
typedef struct { int mi; int* buf; int *hlnk[3], *vlnk[3], *lwrb[3], *uprb[3]; } HUImd;

int orc_scalar_udh_h(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w, int n_im, int imd_intvl,
                     int32_t* score, int32_t* cpos, int32_t* ranges)
{
    if (!sc->intpen || !p->dinc || n_im < 1) return -1;
    code_tables();
    const int minl = sc->minl ? sc->minl : sc->llmt;
    Ctx cx = {sc, p, minl};
    const int NEV = SPDP_NEVSEL, EOU = SPDP_END_OF_ULK;
    int al = p->a_left, ar = p->a_right, bl = p->b_left, br = p->b_right;
    const int Local = sc->local;
    const int LocalL = Local && p->a_exgl && p->b_exgl;
    const int LocalR = Local && p->a_exgr && p->b_exgr;
    const int spj = sc->spj;
    const int lw = w->lw, up = w->up, width = w->width;
    const int dagp = sc->noll == 3;                     /* double affine gaps (src/fwd2h1.cc:1088, 1140, 1162, 1211-1247, 1316-1330, 1412-1440) */
    const int noll = dagp ? 3 : 2, nod = dagp ? 5 : 3;
    const int gapw3l = sc->lgop + sc->lgep;
    const int GOP[3] = {0, sc->gop, sc->lgop};
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    for (int i = 0; i <= n_im; ++i) for (int c = 0; c < 10; ++c) CPOS(i, c) = EOU;
    const size_t bufsiz = (size_t) noll * width;
    Rvdwml* wbuf = (Rvdwml*) malloc((bufsiz + 8) * sizeof(Rvdwml));
    int r = bl - 3 * ar;
    const Rvdwml black = {NEV, 0, r, r, 0, EOU};
    const Rvdwmlj blackj = {NEV, 0, INT_MIN, INT_MAX, 0, EOU, 0};
    for (size_t i = 0; i < bufsiz + 8; ++i) wbuf[i] = black;
    Rvdwml* hh0 = wbuf - lw + 3;
    Rvdwml* hh1 = hh0 + width;
    Rvdwml* hh2 = hh1 + width;                          /* F2 (Noll = 3) */

    /* ---- hinitH_ng ---- */
    {
        int n = bl;
        r = bl - 3 * al;
        const int r0 = r;
        int rr = br - 3 * al;
        const int dir = p->a_exgl ? DEAD : DIAG;
        int bb = n + 1;
        Rvdwml* h = hh0 + r;
        h->val = (p->a_exgl && p->sigS[bb] > 0) ? p->sigS[bb] : 0;
        h->dir = dir;
        h->lwr = h->upr = h->ulk = r0;
        h->ml = al;
        if (p->a_exgl) {
            if (up < rr) rr = up;
            int jnc = n;
            for (int i = 1; ++r <= rr; ++i) {
                ++h; ++bb; ++n;
                if (i < 3) {
                    h->val = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
                    h->dir = dir;
                    h->lwr = h->ulk = r;
                    h->ml = al;
                } else {
                    *h = h[-3];
                    const int d = n - jnc;
                    if (!(p->a_exgl & 1) && d == 3) h->val += sc->gop;
                    if (!(p->a_exgl & 2)) h->val += gap_ext3(sc, d);
                    h->val += p->sigE[bb - 3];
                    h->dir = HORI;
                    int x = h[-1].val + sc->gapw1;
                    if (x > h->val) { *h = h[-1]; h->val = x; h->dir = HOR1; }
                    x = h[-2].val + sc->gapw2;
                    if (x > h->val) { *h = h[-2]; h->val = x; h->dir = HOR2; }
                }
                const int x = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
                if (h->val < x) { h->val = x; h->dir = DEAD; jnc = n; h->lwr = h->ulk = r; }
                h->upr = r;
            }
        }
        r = r0;
        rr = bl - 3 * ar;
        if (lw > rr) rr = lw;
        h = hh0 + r - 1;
        for (int i = 1; --r >= rr; ++i, --h) {
            if (p->b_exgl == 1) {
                h->val = 0; h->dir = DEAD;
                h->upr = h->lwr = h->ulk = r;
                h->ml = al + i / 3;
            } else if (i <= 3) {
                *h = h[i];
                if (!(p->b_exgl & 2)) h->val += sc->gep;
                if (!(p->b_exgl & 1)) h->val += sc->gop;
                if (i < 3) h->val += sc->extragop;
                h->dir = VERT;
                h->ml += i / 3;
                h->lwr = h->ulk = r;
            } else {
                *h = h[3];
                if (!(p->b_exgl & 2)) h->val += gap_ext3(sc, i);
                h->lwr = h->ulk = r;
                ++h->ml;
            }
        }
    }
    HUImd* imds = (HUImd*) calloc(n_im, sizeof(HUImd));
    {
        int mi = al;
        const size_t us = (size_t) noll * width;
        for (int i = 0; i < n_im; ++i) {
            HUImd* d = imds + i;
            d->mi = (mi += imd_intvl);
            d->buf = (int*) malloc(4 * us * sizeof(int));
            for (size_t k = 0; k < 2 * us; ++k) d->buf[k] = EOU;
            for (size_t k = 0; k < us; ++k) { d->buf[2 * us + k] = INT_MAX; d->buf[3 * us + k] = INT_MIN; }
            d->hlnk[0] = d->buf - lw + 1;  d->vlnk[0] = d->hlnk[0] + us;
            d->lwrb[0] = d->vlnk[0] + us;  d->uprb[0] = d->lwrb[0] + us;
            for (int k = 1; k < noll; ++k) {
                d->hlnk[k] = d->hlnk[k - 1] + width; d->vlnk[k] = d->vlnk[k - 1] + width;
                d->lwrb[k] = d->lwrb[k - 1] + width; d->uprb[k] = d->uprb[k - 1] + width;
            }
        }
    }
    HUImd* imd = imds;
    int mm = imd->mi;
    int rlst[3] = {INT_MAX, INT_MAX, INT_MAX};
    int maxh_val = NEV, maxh_upr = 0, maxh_lwr = 0, maxh_ml = al, maxh_ulk = 0, maxh_mr = ar, maxh_nr = br;
    int m = al;
    if (!p->a_exgl) --m;
    int n1 = 3 * m + lw - 1;
    int n2 = 3 * m + up;
    for (int i = 0; ++m <= ar; ) {
        n1 += 3; n2 += 3;
        const int n0 = imax(n1, bl);
        const int n9 = imin(n2, br);
        const int is_imd = m == mm;
        int n = n0;
        r = n - 3 * m;
        Rvdwml e1[NQUE] = {black, black, black};
        Rvdwml e2[NQUE] = {black, black, black};
        if (!p->b_exgl && m == al) { e1[2] = e2[2] = hh0[r]; e1[2].val += sc->gapw3; e2[2].val += gapw3l; }
        Rvdwml* h = hh0 + r;
        Rvdwml* f = hh1 + r;
        Rvdwml* f2 = dagp ? hh2 + r : 0;
        Rvdwml* hf[5] = {h, 0, f, 0, f2};
        const int aa0 = a_code(&cx, m - 1), aa1 = a_code(&cx, m);
        Rvdwmlj hl[3][NCAND + 1];
        int nx[3][NCAND + 1];
        for (int ph = 0; ph < 3; ++ph)
            for (int l = 0; l <= NCAND; ++l) { hl[ph][l] = blackj; nx[ph][l] = l; }
        int ncand[3] = {-1, -1, -1};
        int q = 0;
        for ( ; n <= n9; ++n, ++r, ++h, ++f, f2 += dagp) {
            int x, y;
            hf[0] = h; hf[2] = f; hf[4] = f2;
            const int sigE = (n > bl && n >= 2) ? p->sigE[n - 2] : 0;
            Rvdwml* const eq1 = hf[1] = e1 + q;
            Rvdwml* const eq2 = hf[3] = dagp ? e2 + q : 0;
            const Rvdwml hq = *h;
            Rvdwml* from = h;
            Rvdwml* mx = h;
            if (m != al) {
                if (n < bl + 3) *h = black;
                else {
                    h->val += mtx_at(&cx, aa0, b_code(&cx, n - 2)) + sigE;
                    h->dir = (from->dir & DIAG) ? DIAG : NEWD;
                }
                y = f[3].val + sc->gep;
                ++from;
                x = from->val + (is_vert[from->dir & 15] ? sc->gape1 : sc->gapw1);
                if (x > y) { *f = *from; f->val = x; f->dir = SLA2; }
                else f->val = y;
                ++from;
                x = from->val + (is_vert[from->dir & 15] ? sc->gape2 : sc->gapw2);
                if (x > f->val) { *f = *from; f->val = x; f->dir = SLA1; }
                x = (++from)->val + sc->gapw3;
                if (x >= f->val) { *f = *from; f->val = x; f->dir = VERT; }
                else if (y >= f->val) { *f = f[3]; f->val = y; f->dir = VERT; }
                if (f->val >= mx->val) mx = f;
                if (dagp) {                     /* long deletion */
                    x = from->val + gapw3l;
                    y = f2[3].val + sc->lgep;
                    if (x >= y) { *f2 = *from; f2->val = x; f2->dir = VERL; }
                    else { *f2 = f2[3]; f2->val = y; }
                    if (f2->val >= mx->val) mx = f2;
                }
            }
            if (n > n0 + 2) {
                from = h - 3;
                x = from->val + sc->gapw3;
                y = eq1->val += sc->gep;
                if (x > y) { *eq1 = *from; eq1->val = x; }
                eq1->val += sigE;
                eq1->dir = (eq1->dir & SPIN) + HORI;
                if (dagp) {                     /* long insertion */
                    x = from->val + gapw3l;
                    y = eq2->val += sc->lgep;
                    if (x > y) { *eq2 = *from; eq2->val = x; }
                    eq2->val += sigE;
                    eq2->dir = (eq2->dir & SPIN) + HORL;
                    if (eq2->val > mx->val) mx = eq2;
                }
            }
            if (n > n0 + 1) {
                from = h - 2;
                x = from->val + sc->gapw2;
                if (x > eq1->val) { *eq1 = *from; eq1->val = x; eq1->dir = HOR2; }
            }
            from = h - 1;
            x = from->val + sc->gapw1;
            if (x > eq1->val) { *eq1 = *from; eq1->val = x; eq1->dir = HOR1; }
            if (eq1->val > mx->val) mx = eq1;
            if (++q == NQUE) q = 0;

            int spj3 = 0;
            if (spj && p->phs3[n] > -2) {
                int phs = (p->phs3[n] == 2) ? -1 : p->phs3[n];
                for (;;) {
                    const int nb = n - phs;
                    const int* pnx = nx[phs + 1];
                    const Rvdwmlj* maxphl[5] = {0, 0, 0, 0, 0};
                    for (int l = 0; l <= ncand[phs + 1]; ++l) {
                        const Rvdwmlj* phl = hl[phs + 1] + pnx[l];
                        if (phs == 1 && phl->dir == 2) continue;
                        if (nb - phl->jnc < minl) continue;
                        x = phl->val + (p->cip ? p->cip[3 * m - phs] : 0) + spjscr(&cx, phl->jnc, nb);   /* sigB[phs] = cip_score(3 m - phs) */
                        if (phl->dir == 0 && phs) {
                            int cs[2];
                            spjseq(&cx, phl->jnc, nb, cs);
                            if (phs == 1) x += mtx_at(&cx, aa0, cs[0]);
                            else x += mtx_at(&cx, aa1, cs[1]) - mtx_at(&cx, aa1, b_code(&cx, n + 1)) - p->sigE[n + 1];
                        }
                        from = hf[phl->dir];
                        if (x > from->val) { from->val = x; maxphl[phl->dir] = phl; }
                    }
                    int maxk = nod;
                    for (int k = 0; k < nod; ++k) {
                        const Rvdwmlj* phl = maxphl[k];
                        if (!phl) continue;
                        from = hf[k];
                        from->dir = nod2dir[phl->dir] | SPIN;
                        from->upr = imax(phl->upr, r);
                        from->lwr = imin(phl->lwr, r);
                        from->ml = phl->ml;
                        from->ulk = phl->ulk;
                        if (from->val >= mx->val) { maxk = k; mx = from; }
                    }
                    if (is_imd && maxk < nod) {
                        const Rvdwmlj* phl = maxphl[maxk];
                        imd->hlnk[0][r] = phl->ulk;
                        mx->ulk = rlst[q] = r;
                        spj3 = 1;
                        if (maxk == 0) {
                            for (int c = 1, d = 1; c < noll; ++c, d += 2) {
                                if ((phl = maxphl[d]) && hf[d]->val > mx->val + GOP[c]) {
                                    hf[d]->ulk = r + c * width;
                                    imd->hlnk[c][r] = phl->ulk;
                                }
                                if (maxphl[d + 1] && hf[d + 1]->val > mx->val + GOP[c]) hf[d + 1]->ulk = r + c * width;
                            }
                        }
                    }
                    if (p->phs3[n] - phs == 3) { phs = 1; continue; }
                    break;
                }
            }

            y = h->val;
            if (h == mx) {
                if (LocalR && y > maxh_val) {
                    maxh_val = h->val; maxh_upr = h->upr; maxh_lwr = h->lwr; maxh_ml = h->ml; maxh_ulk = h->ulk;
                    maxh_mr = m; maxh_nr = n;
                }
            } else {
                if (mx->upr < r) mx->upr = r;
                if (mx->lwr > r) mx->lwr = r;
                *h = *mx;
            }
            if (LocalL && h->val <= 0) {
                h->val = h->dir = 0;
                h->ml = m;
                h->ulk = h->upr = h->lwr = r;
            }

            const int hd = dir2nod[mx->dir & 15];
            if (spj && p->phs5[n] > -2) {
                int phs = (p->phs5[n] == 2) ? -1 : p->phs5[n];
                for (;;) {
                    const int nb = n - phs;
                    const int sigJ = p->sig5[nb];
                    for (int k = (hd == 0 || phs == 1) ? 0 : 1; k < nod; ++k) {
                        const int crossspj = phs == 1 && k == 0;
                        const Rvdwml* src = crossspj ? &hq : hf[k];
                        if (!src->dir || (src->dir & SPIN)) continue;
                        if (k != hd && !crossspj && hd >= 0) {
                            y = mx->val;
                            if (hd == 0 || (k - hd) % 2) y += GOP[k / 2];
                            if (src->val <= y) continue;
                        }
                        x = src->val + sigJ;
                        Rvdwmlj* phl = hl[phs + 1];
                        int* pnx = nx[phs + 1];
                        int* nc = &ncand[phs + 1];
                        int l = *nc < NCAND ? ++*nc : NCAND;
                        while (--l >= 0) {
                            if (x >= phl[pnx[l]].val) { const int t = pnx[l]; pnx[l] = pnx[l + 1]; pnx[l + 1] = t; }
                            else break;
                        }
                        if (++l < NCAND) {
                            phl += pnx[l];
                            phl->val = x; phl->jnc = nb; phl->dir = k;
                            phl->upr = src->upr; phl->lwr = src->lwr; phl->ml = src->ml;
                            if (is_imd) {
                                if (k == 1) imd->hlnk[0][r] = rlst[q];
                                phl->ulk = r;
                            } else phl->ulk = src->ulk;
                        } else --*nc;
                    }
                    if (p->phs5[n] - phs == 3) { phs = 1; continue; }
                    break;
                }
            }

            if (is_imd) {
                if (hd == 0) rlst[q] = r;
                else if (!spj3 && hd % 2) imd->hlnk[0][r] = rlst[q];
                for (int k = 0; k < noll; ++k) {
                    Rvdwml* g = hf[2 * k];
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
    const int rr = br - 3 * ar;
    if (LocalR) {
        int i = n_im;
        while (--i >= 0 && imds[i].mi > ar) ;
        ar = maxh_mr; br = maxh_nr;
        if (i < 0) i = 0;
        CPOS(i, 8) = maxh_lwr;
        CPOS(i, 9) = maxh_upr;
    } else {        /* ---- hlastH_ng ---- */
        static const int next_p[3] = {1, 2, 0};
        int glen[3] = {0, 0, 0};
        const int m3 = 3 * ar;
        int rw = lw;
        int rf = bl - m3;
        if (rf > rw) rw = rf; else rf = rw;
        Rvdwml* h = hh0 + rw;
        Rvdwml* h9 = hh0 + br - m3;
        Rvdwml* mx = h9;
        int bb = rw + m3;
        if (p->a_exgr) {
            for (int ph = 0; h <= h9; ++h, ++bb, ++rf, ph = next_p[ph]) {
                glen[ph] += 3;
                int cand[3] = {h->val, NEV, NEV};
                if (rf - rw >= 3 && h[-3].dir != DEAD) {
                    cand[1] = h[-3].val + p->sigE[bb - 2];
                    if (!(p->a_exgr & 2)) cand[1] += gap_ext3(sc, glen[ph]);
                    if (glen[ph] == 3 && !(p->a_exgr & 1)) cand[1] += sc->gop;
                    if (sc->term_codon && !(h->dir & SPIN)) cand[2] = h[-3].val + p->sigT[bb - 2];
                }
                const int sig5 = (Local && p->sig5[bb] > 0) ? p->sig5[bb] : 0;
                cand[0] += sig5;
                cand[1] += sig5;
                int k = 0;
                if (cand[1] > cand[k]) k = 1;
                if (cand[2] > cand[k]) k = 2;
                if (k == 0) { if (!is_hori[h->dir & 15]) glen[ph] = 0; }
                else if (k == 1) { *h = h[-3]; h->dir = HORI; h->val = cand[k] - sig5; }
                else { *h = h[-3]; h->dir = DEAD; h->val = cand[k]; h->upr = imax(rf, h->upr); }
                if (h->val > mx->val) mx = h;
            }
        } else {
            bb += (int) (h9 - h);
            const int y = h9[-3].val + p->sigT[bb - 2];
            if (y > h9->val) { *h9 = h9[-3]; h9->val = y; h9->dir = HORI; h9->upr = imax(br - m3, h9->upr); }
        }
        if (p->b_exgr == 1) {
            rw = imin(up, br - 3 * al);
            for (h = hh0 + rw; h > h9; --h, --rw) {
                const int x = h->val + ((rw % 3) ? sc->extragop : 0);
                if (x > mx->val) { mx = h; mx->val = x; }
            }
        } else if (p->b_exgr == 2)
            mx = hh1 + br - m3;
        maxh_val = mx->val; maxh_lwr = mx->lwr; maxh_upr = mx->upr; maxh_ulk = mx->ulk; maxh_ml = mx->ml;
        r = (int) (mx - hh0);
        if (p->b_exgr && rr < r) ar = (br - r) / 3;
        if (p->a_exgr && rr > r) br = 3 * ar + r;
    }
    int i = n_im;
    while (--i >= 0 && imds[i].mi > ar) ;
    if (i < 0 && imds[0].mi > ar) CPOS(0, 2) = br;
    r = br - 3 * ar;
    CPOS(i + 1, 8) = imin(maxh_lwr, r);
    CPOS(i + 1, 9) = imax(maxh_upr, r);
    r = maxh_ulk;
    for ( ; i >= 0 && (imd = imds + i)->mi > maxh_ml; --i) {
        int c = 0, d = 0;
        for ( ; r > up; r -= width) ++d;
        if (d > noll - 1 || r < lw - 1) { rc = -3; break; }
        if (imd->vlnk[d][r] < EOU) {
            CPOS(i, c++) = imd->mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            const int mm3 = 3 * imd->mi;
            for (int rp = imd->hlnk[d][r]; lw <= rp && rp < up && r != rp; rp = imd->hlnk[0][r = rp]) {
                if (c >= 6) { rc = -3; break; }
                CPOS(i, c++) = r + mm3;
            }
            if (rc) break;
            CPOS(i, c++) = r + mm3;
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
        if (LocalL) { al = maxh_ml; bl = r + 3 * maxh_ml; }
        else {
            const int rl = bl - 3 * al;
            if (p->b_exgl && rl > r) {
                al = (bl - r) / 3;
                for (int j = 0; j < n_im && imds[j].mi < al; ++j) CPOS(j, 0) = EOU;
            }
            if (p->a_exgl && rl < r) bl = 3 * al + r;
        }
        ++i;
        if ((i < n_im && imds[i].mi < al) || CPOS(i, 2) < bl) maxh_val = NEV;
        else if (CPOS(i, 8) == EOU || CPOS(i, 9) == EOU) rc = -3;
        else {
            r = bl - 3 * al;
            CPOS(i, 8) = imin(r, CPOS(i, 8));
            CPOS(i, 9) = imax(r, CPOS(i, 9));
        }
    }
#undef CPOS
    *score = maxh_val;
    ranges[0] = al; ranges[1] = ar; ranges[2] = bl; ranges[3] = br;
    for (int j = 0; j < n_im; ++j) free(imds[j].buf);
    free(imds); free(wbuf);
    return rc;
}

/* ======================================================================================== */
/* The -A1 ("full-precision intron-length distribution") SIMD engines of the protein path.   */
/*   orc_exact_forward_h  SimdAln2h1::forwardH1 (modes 3 / 5)     src/fwd2h1_simd.h:820-1096  */
/*                        + fhinitH1 / fhlastH1 with a Vmf        src/fwd2h1_simd.h:546-791   */
/*                        + Sjsites::get / put, from_spj / to_spj src/fwd2h1_simd.h:388-543, 793-815 */
/*                        + the record fix-up of trcbkalignH_ng   src/fwd2h1.cc:2019-2036     */
/*   orc_exact_udh_h      SimdAln2h1::hirschbergH1 (modes 2 / 4)  src/fwd2h1_simd.h:1100-1470 */
/* 16 int16 lanes, six codon-phase planes (H, F by n mod 6; E and the per-frame side lanes by */
/* n mod 3) as in the `_wip` engines; the intron model is the scalar engines': each lane keeps */
/* the top-4 donor candidates (value, junction, state, phase) of its row, an acceptor column   */
/* re-scores them with the exact IntPen(len) + pair signal and the codon the junction spells.  */
/* The reference keeps its lane planes from one stripe to the next without clearing the link   */
/* planes; they are kept alive across stripes here as well.  No cip, no checkpoint re-basing   */
/* (the score planes of a 16-bit run are not re-based below 0.9 * SHRT_MAX / AvTrc rows).      */
/* ======================================================================================== */
#define XN    16
#define XN1   17
#define XNEV  (SHRT_MIN + 1024)
static inline int xsat(int x) { return x < SHRT_MIN ? SHRT_MIN : (x > SHRT_MAX ? SHRT_MAX : x); }
static inline int xadd(int x, int y) { return xsat(x + y); }
static inline int xw16(int x) { return (int16_t) x; }
static inline int xmod6(int x) { x %= 6; return x < 0 ? x + 6 : x; }
static inline int xgood(const SpdpProblemH* p, int n) { return p->exin_left - 1 <= n && n < p->exin_right; }

typedef struct { int val, ulk, jnc, ml, dir, phs; } XhCand;
typedef struct { XhCand rcd[5]; int idx[5]; int ncand; } XhSites;
typedef struct { int q[XN]; int ne, qs, qe; } XhQueue;              /* Queue2<int>(nelem), src/clib.h:397-441 */
static void xq_clear(XhQueue* q) { memset(q, 0, sizeof *q); }
static void xq_push(XhQueue* q, int v)
{
    q->q[q->qe] = v;
    if (++q->qe == XN) q->qe = 0;
    if (q->ne < XN) ++q->ne;
    else if (++q->qs == XN) q->qs = 0;
}
static void xq_pull(XhQueue* q) { if (++q->qs == XN) q->qs = 0; if (q->ne > 0) --q->ne; }

typedef struct { int mi; int *hlnk[2], *vlnk[2]; int* buf; } XhImd;

typedef struct {
    Ctx cx;
    const SpdpScoringH* sc;
    const SpdpProblemH* p;
    int lw, up, width, buf_size;
    int a_left, a_right, b_left, b_right;
    int *hv, *fv, *hb, *fb, *hc, *fc;           /* boundary rows by diagonal */
    int *mem;
    int HV[6][XN1], FV[6][XN1], EV[3][XN], QV[3][XN], PS[3][XN], PV[3][XN], CP[3][XN1], SM[XN1];
    int HB[6][XN1], FB[6][XN1], EB[3][XN], QB[3][XN];
    int HC[6][XN1], FC[6][XN1], EC[3][XN], QC[3][XN];
    XhSites sites[XN];
    XhQueue dq[3], aq[3];
    Vmf* vmf;                                   /* forward (modes 3 / 5) */
    XhImd* imd; int mm3; int rlst[3];           /* linear space (modes 2 / 4) */
    int mode, LocalL, LocalR;
    int max_val, max_ulk, max_ml, max_mr, max_nr;
} XhEng;

static int* xh_slot(XhEng* e, int plane, int qq, int j, int d)
{   /* hfesv / hfesb / hfesc [qq][j][d], d = 0 H, 1 E, 2 F, 3 the diagonal predecessor (fwd2h1_simd.h:301-325) */
    const int f3 = qq % 3;
    if (plane == 0) return d == 0 ? &e->HV[qq][j + 1] : d == 1 ? &e->EV[f3][j] : d == 2 ? &e->FV[qq][j + 1] : &e->QV[f3][j];
    if (plane == 1) return d == 0 ? &e->HB[qq][j + 1] : d == 1 ? &e->EB[f3][j] : d == 2 ? &e->FB[qq][j + 1] : &e->QB[f3][j];
    return d == 0 ? &e->HC[qq][j + 1] : d == 1 ? &e->EC[f3][j] : d == 2 ? &e->FC[qq][j + 1] : &e->QC[f3][j];
}

static void xs_reset_h(XhSites* s)
{
    s->ncand = -1;
    for (int i = 0; i <= 4; ++i) {
        s->rcd[i].val = XNEV; s->rcd[i].ulk = s->rcd[i].jnc = s->rcd[i].ml = s->rcd[i].dir = 0; s->rcd[i].phs = -2;
        s->idx[i] = i;
    }
}

/* Sjsites::get, src/fwd2h1_simd.h:388-494 */
static void xh_get(XhEng* e, XhSites* s, int j, int m, int n, int q)
{
    static const int psp_bit[3] = {4, 1, 8};
    const SpdpProblemH* p = e->p;
    const int acc = n - 1;
    const int r = acc - 3 * (m + 1);
    const XhCand* maxprd[3] = {0, 0, 0};
    const XhCand* brd = 0;
    const int is_imd = e->imd && (m + 1) == e->imd->mi;
    for (int l = 0; l <= s->ncand; ++l) {
        const XhCand* prd = s->rcd + s->idx[l];
        const int rr = r + prd->phs;
        if (rr < e->lw || rr >= e->up) continue;
        const int d = prd->dir, don = prd->jnc;
        if (d == 2 && prd->phs == 1) continue;
        if (acc - don < e->cx.minl) continue;
        int x = prd->val + (p->cip ? p->cip[3 * (m + 1) - prd->phs] : 0) + spjscr(&e->cx, don, acc);
        if (d == 0 && prd->phs) {
            int cs[2];
            spjseq(&e->cx, don, acc, cs);
            if (prd->phs == 1) x += mtx_at(&e->cx, a_code(&e->cx, m), cs[0]);
            else x += mtx_at(&e->cx, a_code(&e->cx, m + 1), cs[1]) - mtx_at(&e->cx, a_code(&e->cx, m + 1), b_code(&e->cx, acc))
                      - p->sigE[acc];
        }
        const int qq = xmod6(q + prd->phs);
        int* v[3] = {xh_slot(e, 0, qq, j, 0), xh_slot(e, 0, qq, j, 1), xh_slot(e, 0, qq, j, 2)};
        if (x <= *v[d]) continue;
        if (!maxprd[d] || x > maxprd[d]->val) {
            maxprd[d] = prd;
            if (!brd || x > brd->val) brd = prd;
        }
        *v[d] = xw16(x);
        e->PS[qq % 3][j] |= psp_bit[d];
        int* b[3] = {xh_slot(e, 1, qq, j, 0), xh_slot(e, 1, qq, j, 1), xh_slot(e, 1, qq, j, 2)};
        int* c[3] = {xh_slot(e, 2, qq, j, 0), xh_slot(e, 2, qq, j, 1), xh_slot(e, 2, qq, j, 2)};
        *b[d] = prd->ml;
        if (e->vmf) {
            const int inner = vmf_add(e->vmf, m + 1, don + prd->phs, prd->ulk);
            *c[d] = vmf_add(e->vmf, m + 1, acc + prd->phs, inner);
        } else
            *c[d] = prd->ulk;
        if (d && *v[d] > *v[0]) { *v[0] = *v[d]; *b[0] = *b[d]; *c[0] = *c[d]; }
        if (j + 1 == XN) {
            e->hv[rr] = *v[0];
            if (is_imd) e->imd->hlnk[0][rr] = prd->ulk;
            else { e->hb[rr] = *b[0]; e->hc[rr] = *c[0]; }
            if (d == 2) {
                e->fv[rr] = *v[d];
                if (is_imd) e->imd->hlnk[1][rr] = prd->ulk;
                else { e->fb[rr] = *b[d]; e->fc[rr] = *c[d]; }
            }
        }
    }
    if (is_imd && brd) {
        const int maxd = brd->dir;
        const XhCand* prd = maxprd[maxd];
        const int qq = xmod6(q + prd->phs);
        const int lstr = e->rlst[qq % 3] = acc + prd->phs - e->mm3;
        e->imd->hlnk[0][lstr] = prd->ulk;
        int* v[3] = {xh_slot(e, 0, qq, j, 0), xh_slot(e, 0, qq, j, 1), xh_slot(e, 0, qq, j, 2)};
        int* c[3] = {xh_slot(e, 2, qq, j, 0), xh_slot(e, 2, qq, j, 1), xh_slot(e, 2, qq, j, 2)};
        *c[maxd] = lstr;
        e->PV[qq % 3][j] = maxd;
        if (maxd) { *c[0] = *c[maxd]; return; }
        if ((prd = maxprd[1]) && *v[1] > *v[0] + e->sc->gop) {
            e->imd->hlnk[1][lstr] = prd->ulk;
            *c[1] = lstr + e->width;
        }
        if (maxprd[2] && *v[2] > *v[0] + e->sc->gop) *c[2] = lstr + e->width;
    }
}

/* Sjsites::put, src/fwd2h1_simd.h:496-543 */
static void xh_put(XhEng* e, XhSites* s, int j, int m, int n, int q)
{
    static const int psp_bit[3] = {4, 1, 8};
    const SpdpProblemH* p = e->p;
    const int don = n - 1;
    const int sigJ = p->sig5[don];
    const int is_imd = e->imd && (m + 1) == e->imd->mi;
    for (int phs = 1; phs > -2; --n, --phs, q = xmod6(q - 1)) {
        const int rr = don - 3 * (m + 1) + phs;
        if (rr < e->lw || rr >= e->up) continue;
        const int f3 = q % 3;
        const int h = e->PV[f3][j];
        const int thrscr = *xh_slot(e, 0, q, j, 0) + e->sc->gop;
        for (int k = (h && phs < 1) ? 1 : 0; k < 3; ++k) {
            if (e->PS[f3][j] & psp_bit[k]) continue;
            const int cross = (phs == 1 && k == 0) ? 3 : k;
            const int from = *xh_slot(e, 0, q, j, cross);
            if (k && from <= thrscr) continue;
            const int x = from + sigJ;
            if (x <= XNEV) continue;
            int l = s->ncand < 4 ? ++s->ncand : 4;
            while (--l >= 0) {
                if (x >= s->rcd[s->idx[l]].val) { const int t = s->idx[l]; s->idx[l] = s->idx[l + 1]; s->idx[l + 1] = t; }
                else break;
            }
            if (++l < 4) {
                XhCand* prd = s->rcd + s->idx[l];
                prd->val = xw16(x);
                prd->ml = *xh_slot(e, 1, q, j, k);
                const int r = n - e->mm3;
                const int rl = *xh_slot(e, 2, q, j, cross);
                if (is_imd) {
                    if (k == 1) e->imd->hlnk[0][r] = e->rlst[f3];
                    prd->ulk = r;
                } else
                    prd->ulk = rl;
                prd->jnc = don; prd->dir = k; prd->phs = phs;
            } else --s->ncand;
        }
    }
}

static void xh_from_spj(XhEng* e, XhQueue* l, int m, int n, int q)
{
    for (int ns = 0; ns < l->ne; ++ns) {
        const int nj = l->q[(l->qs + ns) % XN];
        const int j = (n - nj) / 3, mj = m + j;
        if (mj + 1 < e->a_right) xh_get(e, e->sites + j, j, mj, nj, q - 1);
    }
}
static void xh_to_spj(XhEng* e, XhQueue* l, int m, int n, int q)
{
    for (int ns = 0; ns < l->ne; ++ns) {
        const int nj = l->q[(l->qs + ns) % XN];
        const int j = (n - nj) / 3, mj = m + j;
        if (mj + 1 < e->a_right) xh_put(e, e->sites + j, j, mj, nj, q);
    }
}

static int xh_open(XhEng* e, const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w, int mode, Vmf* vmf)
{
    memset(e, 0, sizeof *e);
    code_tables();
    e->sc = sc; e->p = p;
    e->cx.sc = sc; e->cx.p = p; e->cx.minl = sc->minl ? sc->minl : sc->llmt;
    e->lw = w->lw; e->up = w->up; e->width = w->width;
    e->buf_size = w->width + 6 * XN;
    e->a_left = p->a_left; e->a_right = p->a_right; e->b_left = p->b_left; e->b_right = p->b_right;
    e->mem = (int*) calloc((size_t) 6 * e->buf_size, sizeof(int));
    if (!e->mem) return -1;
    e->hv = e->mem - w->lw + 3;       e->fv = e->hv + e->buf_size;
    e->hb = e->fv + e->buf_size;      e->fb = e->hb + e->buf_size;
    e->hc = e->fb + e->buf_size;      e->fc = e->hc + e->buf_size;
    e->mode = mode; e->vmf = vmf;
    e->LocalL = sc->local && p->a_exgl && p->b_exgl;
    e->LocalR = sc->local && p->a_exgr && p->b_exgr;
    e->max_val = XNEV; e->max_ulk = SPDP_END_OF_ULK;
    e->max_ml = p->a_left; e->max_mr = p->a_right; e->max_nr = p->b_right;
    e->rlst[0] = e->rlst[1] = e->rlst[2] = INT_MAX;
    return 0;
}

/* fhinitH1 without a traceback bitmap, src/fwd2h1_simd.h:546-689 */
static void xh_init(XhEng* e)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int lw = e->lw, up = e->up;
    int *hv = e->hv, *fv = e->fv, *hb = e->hb, *hc = e->hc, *fc = e->fc;
    const int B = e->buf_size;
    for (int i = 0; i < 2 * B; ++i) (hv + lw - 3)[i] = XNEV;
    const int rl = e->b_left - 3 * e->a_left;
    Vmf* vmf = e->vmf;
    if (vmf) {
        for (int i = 0; i < 2 * B; ++i) (hb + lw - 3)[i] = 0;
        int ptr = vmf_add(vmf, 0, 0, 0);
        if (!(p->a_exgl && p->b_exgl)) ptr = vmf_add(vmf, e->a_left, e->b_left, ptr);
        for (int r = rl; r < up; ++r) hc[r] = p->a_exgl ? 0 : ptr;
        for (int r = lw; r < rl; ++r) hc[r] = p->b_exgl ? 0 : ptr;
        if (p->b_exgl == 2) fc[rl] = ptr;
    } else {
        for (int i = 0; i < 2 * B; ++i) (hb + lw - 3)[i] = e->a_left;
        const int re = p->a_exgl ? rl : up;
        for (int r = lw; r < re; ++r) hc[r] = r;
        for (int i = 0, r = rl; r >= lw; --r) hb[r] = e->a_left + (i++ / 3);
    }
    if (p->b_exgl == 1) { for (int r = lw; r < rl; ++r) hv[r] = 0; }
    else if (p->b_exgl == 2) { fv[rl] = 0; fc[rl] = rl; }

    int rr = e->b_right - 3 * e->a_left;
    if (up < rr) rr = up;
    int r = rl;
    if (!p->a_exgl) {
        if (p->b_exgl) { fv[r] = 0; fc[r] = hc[r]; }
        hv[r++] = 0;
        hv[r++] = xw16(sc->gapw1);
        hv[r++] = xw16(sc->gapw2);
        hv[r++] = xw16(sc->gapw3);
        if (sc->gep) {
            int x = (XNEV - sc->gapw3) / sc->gep + r;
            if (x < rr) rr = x;
            for ( ; r < rr; ++r) hv[r] = xw16(hv[r - 3] + sc->gep);
        } else if (rr > r)
            for (int v = hv[r - 1]; r < rr; ++r) hv[r] = v;
        return;
    }
    int n = e->b_left;
    int lend[3] = {r, r + 1, r + 2};
    int bb = n + 1;
    for (int f = 0; f < 3; ++f, ++r, ++n, ++bb) {
        hv[r] = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
        if (vmf) { hc[r] = vmf_add(vmf, e->a_left, n, 0); hb[r] = 1; }
        else hc[r] = r;
    }
    for (int f = 0; r < rr; ++r, ++n, ++bb, f = (f + 1) % 3) {
        int h = hv[r - 3];
        hc[r] = hc[r - 3];
        const int gl = r - lend[f];
        if (!(p->a_exgl & 1) && gl == 3) h = xw16(h + sc->gop);
        if (!(p->a_exgl & 2)) h = xw16(h + gap_ext3(sc, gl));
        h = xw16(h + p->sigE[bb - 3]);
        hv[r] = h;
        if (h < XNEV) break;
        int x = xw16(hv[r - 1] + sc->gapw1);
        if (x > h) { hv[r] = h = x; hc[r] = hc[r - 1]; }
        x = xw16(hv[r - 2] + sc->gapw2);
        if (x > h) { hv[r] = h = x; hc[r] = hc[r - 2]; }
        x = p->sigS[bb] > 0 ? p->sigS[bb] : 0;
        if (x > h) {
            hv[r] = x; lend[f] = r;
            if (vmf) { hc[r] = vmf_add(vmf, e->a_left, n, 0); hb[r] = 1; }
            else hc[r] = r;
        }
    }
}

/* fhlastH1, src/fwd2h1_simd.h:691-791; returns the diagonal of the end cell */
static int xh_last(XhEng* e)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int lw = e->lw, up = e->up;
    int* hv = e->hv;
    int glen[3] = {0, 0, 0}, tcdn[3] = {0, 0, 0};
    const int m3 = 3 * e->a_right;
    int rw = lw;
    int rf = e->b_left - m3;
    if (rf > rw) rw = rf; else rf = rw;
    const int rr = e->b_right - m3;
    int maxr = rr, mx = rr;
    int bb = rw + m3;
    if (p->a_exgr) {
        int f = 0;
        for (int h = rw; h <= rr; ++h, ++rf, ++bb, f = (f + 1) % 3) {
            glen[f] += 3;
            int cand[3] = {hv[h], XNEV, XNEV};
            if (rf - rw >= 3 && !tcdn[f]) {
                cand[1] = hv[h - 3] + p->sigE[bb - 2];
                if (!(p->a_exgr & 2)) cand[1] += gap_ext3(sc, glen[f]);
                if (!(p->a_exgr & 1) && glen[f] == 3) cand[1] += sc->gop;
                if (sc->term_codon) cand[2] = hv[h - 3] + p->sigT[bb - 2];
            }
            if (rf - rw >= 3) tcdn[f] = tcdn[f] || p->sigT[bb - 2] > 0;
            const int s5 = (sc->local && p->sig5[bb] > 0) ? p->sig5[bb] : 0;
            cand[0] += s5; cand[1] += s5;
            int k = 0;
            if (cand[1] > cand[k]) k = 1;
            if (cand[2] > cand[k]) k = 2;
            if (k == 0) { glen[f] = 0; tcdn[f] = 0; }
            else if (k == 1) hv[h] = xw16(cand[1] - s5);
            else hv[h] = xw16(cand[2]);
            if (hv[h] > hv[mx]) { mx = h; maxr = rf - (k == 2 ? 3 : 0); }
        }
    } else {
        const int y = xw16(hv[rr - 3] + p->sigT[bb + (rr - rw)]);
        if (y > hv[rr]) { hv[rr] = y; maxr = rr - 3; }
    }
    if (p->b_exgr) {
        rw = imin(up - 1, e->b_right - 3 * e->a_left);
        int g[3] = {XNEV, XNEV, XNEV};
        int f = 0;
        for (int h = rw - 3; h > rr; --h, f = (f + 1) % 3) {
            int x = hv[h + 3];
            if (!(p->b_exgr & 1)) x = xw16(x + sc->gop);
            if (x > g[f]) g[f] = x;
            if (!(p->b_exgr & 2)) g[f] = xw16(g[f] + sc->gep);
            if (hv[h] > g[f]) g[f] = XNEV;
            else if (g[f] > hv[mx]) { mx = h; hv[h] = g[f]; }
        }
    }
    const int maxt = mx;
    if (e->mode == 2 || e->mode == 4) e->hb[maxt] = e->hb[maxr];
    e->max_ulk = e->hc[maxr];
    int q = maxr - rr;
    if (e->vmf) {
        int m9 = e->a_right, n9 = e->b_right;
        if (q > 0) { m9 -= (q + 2) / 3; if (q %= 3) n9 -= 3 - q; }
        else if (q < 0) n9 += q;
        e->max_ulk = vmf_add(e->vmf, m9, n9, e->max_ulk);
        if (maxr != maxt) e->max_ulk = vmf_add(e->vmf, e->a_right, maxt + m3, e->max_ulk);
    } else {
        if (q > 0) e->max_mr = (e->b_right - maxr) / 3;
        else       e->max_nr = maxt + m3;
    }
    return maxt;
}

/* one anti-diagonal step of forwardH1 / hirschbergH1 up to the splice phases (src/fwd2h1_simd.h:864-1047,
 * 1157-1358): `udh` carries the left-end row in the B planes instead of the diagonal flag */
static void xh_step(XhEng* e, int ml, int j9, int n, int r, int q, int udh)
{
    const SpdpScoringH* sc = e->sc;
    const SpdpProblemH* p = e->p;
    const int ge = sc->gep, g1 = sc->gapw1, g2 = sc->gapw2, g3 = sc->gapw3;
    const int f3 = q % 3;
    const int nb = imax(0, n - e->b_right + 1);
    const int kb = (nb - 1) / 3;
    const int ke = imin(j9, (n - e->b_left) / 3);
    const int LL = e->LocalL;
    if (sc->spj && !nb) {
        if (p->phs5[n] > 0) xq_push(&e->dq[f3], n);
        if (p->phs3[n] > 0) xq_push(&e->aq[f3], n);
        e->CP[f3][0] = xgood(p, n - 2) ? p->sigE[n - 2] : 0;
    }
    int cv[XN];
    for (int k = 0; k < XN; ++k) cv[k] = e->CP[f3][k];
    for (int k = 0; k < XN; ++k) e->CP[f3][k + 1] = cv[k];
    const int q1 = xmod6(q - 1), q2 = xmod6(q - 2), q3 = xmod6(q - 3), q4 = xmod6(q - 4), q5 = xmod6(q - 5);
    int ev[XN], eb[XN], ec[XN], fvv[XN], fbv[XN], fcv[XN];
    for (int k = 0; k < XN; ++k) {                              /* insertion */
        int h = xadd(e->HV[q1][k + 1], g1), hb = e->HB[q1][k + 1], hc = e->HC[q1][k + 1];
        int x = xadd(e->HV[q2][k + 1], g2);
        int m = h > x;
        h = m ? h : x; hb = m ? hb : e->HB[q2][k + 1]; hc = m ? hc : e->HC[q2][k + 1];
        x = xadd(xadd(e->HV[q3][k + 1], g3), cv[k]);
        m = h > x;
        h = m ? h : x; hb = m ? hb : e->HB[q3][k + 1]; hc = m ? hc : e->HC[q3][k + 1];
        x = xadd(xadd(e->EV[f3][k], ge), cv[k]);
        m = x > h;
        ev[k] = m ? x : h; eb[k] = m ? e->EB[f3][k] : hb; ec[k] = m ? e->EC[f3][k] : hc;
    }
    for (int k = 0; k < XN; ++k) { e->EV[f3][k] = ev[k]; e->EC[f3][k] = ec[k]; if (udh && LL) e->EB[f3][k] = eb[k]; }
    e->FV[q3][0] = e->fv[r + 3]; e->FC[q3][0] = e->fc[r + 3];     /* deletion */
    e->HV[q3][0] = e->hv[r + 3]; e->HC[q3][0] = e->hc[r + 3];
    e->HV[q4][0] = e->hv[r + 2]; e->HC[q4][0] = e->hc[r + 2];
    e->HV[q5][0] = e->hv[r + 1]; e->HC[q5][0] = e->hc[r + 1];
    if (udh && LL) {
        e->FB[q3][0] = e->fb[r + 3]; e->HB[q3][0] = e->hb[r + 3];
        e->HB[q4][0] = e->hb[r + 2]; e->HB[q5][0] = e->hb[r + 1];
    }
    for (int k = 0; k < XN; ++k) {
        int f = xadd(e->FV[q3][k], ge), fb = e->FB[q3][k], fc = e->FC[q3][k];
        int x = xadd(e->HV[q3][k], g3);
        int m = f > x;
        f = m ? f : x; fb = m ? fb : e->HB[q3][k]; fc = m ? fc : e->HC[q3][k];
        x = xadd(e->HV[q4][k], g2);
        m = f > x;
        f = m ? f : x; fb = m ? fb : e->HB[q4][k]; fc = m ? fc : e->HC[q4][k];
        x = xadd(e->HV[q5][k], g1);
        m = f > x;
        f = m ? f : x; fb = m ? fb : e->HB[q5][k]; fc = m ? fc : e->HC[q5][k];
        fvv[k] = f; fbv[k] = fb; fcv[k] = fc;
    }
    for (int k = 0; k < XN; ++k) { e->FV[q][k + 1] = fvv[k]; e->FC[q][k + 1] = fcv[k]; if (udh && LL) e->FB[q][k + 1] = fbv[k]; }
    if (nb) for (int k = 0; k < XN; ++k) e->SM[k] = 0;           /* diagonal */
    for (int k = kb; k < ke; ++k)
        e->SM[k] = xw16(sc->mtx[(ml + k < p->a_len ? p->a[ml + k] : 0 /* the byte behind the query: 0 in the reference process */) * sc->mtx_cols + p->b[n - 3 * k - 2]]);
    e->HV[q][0] = e->hv[r]; e->HC[q][0] = e->hc[r];
    if (!udh || LL) e->HB[q][0] = e->hb[r];
    int hx[XN], hbx[XN], hcx[XN], qb[XN];
    for (int k = 0; k < XN; ++k) {
        const int qv = e->HV[q][k], qc = e->HC[q][k], qbb = e->HB[q][k];
        int h = xadd(xadd(e->SM[k], qv), cv[k]);
        e->QV[f3][k] = qv; e->QC[f3][k] = qc;
        if (udh && LL) e->QB[f3][k] = qbb;
        int m = fvv[k] > h;
        h = m ? fvv[k] : h;
        int hc = m ? fcv[k] : qc, hb = m ? fbv[k] : qbb, code = m ? 2 : 0;
        m = ev[k] > h;
        h = m ? ev[k] : h; hc = m ? ec[k] : hc; hb = m ? eb[k] : hb; code = m ? 1 : code;
        e->PV[f3][k] = code;
        e->PS[f3][k] &= code;
        if (!sc->local) { if (!(h > XNEV)) h = XNEV; }
        else if (LL) {
            if (0 > h) { h = 0; if (!udh) { code = 1; hc = 0; } }
        }
        if (!udh) {
            const int diag = code == 0;
            qb[k] = diag & ~qbb & 1;
            hb = diag;
            e->QB[f3][k] = qb[k];
        }
        hx[k] = h; hbx[k] = hb; hcx[k] = hc;
    }
    for (int k = 0; k < XN; ++k) {
        e->HV[q][k + 1] = hx[k]; e->HC[q][k + 1] = hcx[k];
        if (!udh || LL) e->HB[q][k + 1] = hbx[k];
    }
    if (udh && LL)
        for (int k = kb; k < ke; ++k)
            if (e->HV[q][k + 1] == 0) { e->HB[q][k + 1] = ml + k; e->HC[q][k + 1] = r - 6 * k; }
    if (e->LocalR) {
        int bk = 1;
        for (int k = 2; k <= j9; ++k) if (e->HV[q][k] > e->HV[q][bk]) bk = k;
        if (e->HV[q][bk] > e->max_val) {
            e->max_val = e->HV[q][bk]; e->max_ulk = e->HC[q][bk];
            if (udh) { e->max_ml = e->HB[q][bk]; e->max_mr = ml + bk + 1; e->max_nr = n - 3 * bk; }   /* sic, :1355-1356 */
            else     { e->max_mr = ml + bk; e->max_nr = n - 3 * bk + 3; }
        }
    }
    if (!udh)
        for (int k = kb; k < ke; ++k)
            if (qb[k]) e->HC[q][k + 1] = vmf_add(e->vmf, ml + k, n - 3 * (k + 1), e->HC[q][k + 1]);
}

static void xh_stripe_reset(XhEng* e, int j9)
{
    for (int i = 0; i < 6; ++i) for (int k = 0; k < XN1; ++k) { e->HV[i][k] = e->FV[i][k] = XNEV; e->HB[i][k] = e->FB[i][k] = 0; }
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < XN; ++k) { e->EV[i][k] = XNEV; e->EB[i][k] = 0; e->PS[i][k] = e->PV[i][k] = 0; }
        for (int k = 0; k < XN1; ++k) e->CP[i][k] = 0;
    }
    for (int k = 0; k < XN1; ++k) e->SM[k] = 0;
    if (e->sc->spj) {
        for (int j = 0; j < j9; ++j) xs_reset_h(e->sites + j);
        for (int i = 0; i < 3; ++i) { xq_clear(&e->dq[i]); xq_clear(&e->aq[i]); }
    }
}

static void xh_splice(XhEng* e, int ml, int n, int n0, int q)
{
    const int f3 = q % 3;
    if (!e->sc->spj) return;
    if (e->aq[f3].ne) {
        if (e->aq[f3].q[e->aq[f3].qs] < n0) xq_pull(&e->aq[f3]);
        if (e->aq[f3].ne) xh_from_spj(e, &e->aq[f3], ml, n, q);
    }
    if (e->dq[f3].ne) {
        if (e->dq[f3].q[e->dq[f3].qs] < n0) xq_pull(&e->dq[f3]);
        if (e->dq[f3].ne) xh_to_spj(e, &e->dq[f3], ml, n, q);
    }
}

/* rc 0 ok, -1 unsupported parameters, -3 the reference keeps the record pointer in one int16 lane (mode 3)
 * and this run needs more than 32767 records (undefined there).  Records come back end -> start. */
int orc_exact_forward_h(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w,
                        int32_t* score, SpdpSkl** skl, int32_t* n_skl)
{
    *skl = 0; *n_skl = 0;
    if (!sc->intpen || !p->dinc) return -1;
    if (w->width < 0) { *score = SPDP_NEVSEL; return 0; }
    Vmf vmf = {0, 0, 0, 1};
    XhEng* e = (XhEng*) malloc(sizeof(XhEng));
    if (xh_open(e, sc, p, w, 3, &vmf)) { free(e); return -1; }
    xh_init(e);
    for (int ml = e->a_left; ml < e->a_right; ml += XN) {
        const int j9 = imin(XN, e->a_right - ml);
        const int j8 = j9 - 1;
        int n = imax(e->b_left, e->lw + 3 * ml);
        const int n9 = imin(e->b_right, e->up + 3 * (ml + j9) + 1) + 3 * j9;
        int n0 = n - 3 * j8;
        int q = (n + 3 * (ml + 1)) % 6;
        int r = n - 3 * (ml + 1);
        xh_stripe_reset(e, j9);
        for ( ; n < n9; ++n, ++n0, ++r, q = xmod6(q + 1)) {
            const int ke = imin(j9, (n - e->b_left) / 3);
            xh_step(e, ml, j9, n, r, q, 0);
            xh_splice(e, ml, n, n0, q);
            const int r0 = r - 6 * j8;
            if (j9 == ke && e->lw <= r0 && r0 <= e->up) {
                e->hv[r0] = e->HV[q][j9]; e->hb[r0] = e->HB[q][j9]; e->hc[r0] = e->HC[q][j9];
                e->fv[r0] = e->FV[q][j9]; e->fc[r0] = e->FC[q][j9];
            }
        }
    }
    int ptr, val = e->max_val;
    if (!e->LocalR || e->max_mr == e->a_right) { xh_last(e); ptr = e->max_ulk; }
    else ptr = vmf_add(&vmf, e->max_mr, e->max_nr, e->max_ulk);
    int rc = 0;
    {
        const int m = e->a_right - e->a_left;
        float cvol = (float) (w->lw - e->b_left + 3 * e->a_right);
        cvol = (float) m * (float) (e->b_right - e->b_left) - cvol * cvol / 3;
        if (cvol < 65535.f && vmf.n > 32767) rc = -3;
    }
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
        const int rdiag = out[cnt - 1].n - 3 * out[cnt - 1].m;
        const int rd = sc->local ? 0 : (rdiag - e->b_left + 3 * e->a_left);
        if (rd > 0) { out[cnt].m = e->a_left; out[cnt].n = e->b_left + rd; ++cnt; }
        else if (rd < 0) { out[cnt].m = e->a_left - rd / 3; out[cnt].n = e->b_left; ++cnt; }
        *skl = out; *n_skl = cnt;
    }
    *score = val;
    free(vmf.rec); free(e->mem); free(e);
    return rc;
}

/* hirschbergH1: cpos rows of 10, ranges = {a_left, a_right, b_left, b_right} as the reference writes
 * them back into the Seq objects */
int orc_exact_udh_h(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow* w, int n_im,
                    int32_t* score, int32_t* cpos, int32_t* ranges)
{
    if (!sc->intpen || !p->dinc || n_im < 1) return -1;
    XhEng* e = (XhEng*) malloc(sizeof(XhEng));
    if (xh_open(e, sc, p, w, 2, 0)) { free(e); return -1; }
    xh_init(e);
    const int step = (e->a_right - e->a_left + n_im) / (n_im + 1);
    XhImd* imds = (XhImd*) calloc(n_im, sizeof(XhImd));
    for (int i = 0; i < n_im; ++i) {
        imds[i].mi = e->a_left + (i + 1) * step;
        imds[i].buf = (int*) malloc(sizeof(int) * 4 * w->width);
        for (int j = 0; j < 4 * w->width; ++j) imds[i].buf[j] = SPDP_END_OF_ULK;
        imds[i].hlnk[0] = imds[i].buf - w->lw + 1;
        imds[i].hlnk[1] = imds[i].hlnk[0] + w->width;
        imds[i].vlnk[0] = imds[i].hlnk[0] + 2 * w->width;
        imds[i].vlnk[1] = imds[i].vlnk[0] + w->width;
    }
    e->imd = imds;
    int mm = e->a_left + (e->imd->mi - e->a_left - 1) / XN * XN;
    e->mm3 = 3 * e->imd->mi;
    int k9 = e->imd->mi - mm, k8 = k9 - 1;
    for (int ml = e->a_left, i = 0; ml < e->a_right; ml += XN) {
        const int j9 = imin(XN, e->a_right - ml);
        const int j8 = j9 - 1;
        int n = imax(e->b_left, e->lw + 3 * ml);
        const int n9 = imin(e->b_right, e->up + 3 * (ml + j9) + 1) + 3 * j9;
        int n0 = n - 3 * j8;
        int q = xmod6(n + 3 * (ml + 1));
        int r = n - 3 * (ml + 1);
        xh_stripe_reset(e, j9);
        const int is_imd_ = ml == mm;
        for ( ; n < n9; ++n, ++n0, ++r, q = xmod6(q + 1)) {
            const int rj = r - 6 * k8;
            const int f3 = q % 3;
            const int ke = imin(j9, (n - e->b_left) / 3);
            const int is_imd = is_imd_ && rj >= e->lw && rj <= e->up;
            xh_step(e, ml, j9, n, r, q, 1);
            xh_splice(e, ml, n, n0, q);
            if (is_imd) {
                if (e->PV[f3][k8] == 0) e->rlst[f3] = rj;
                if (e->PV[f3][k8] == 1) e->imd->hlnk[0][rj] = e->rlst[f3];
                e->imd->vlnk[0][rj] = e->HC[q][k9];
                e->HC[q][k9] = rj;
                e->imd->vlnk[1][rj] = e->FC[q][k9];
                e->FC[q][k9] = rj + e->width;
            }
            const int r0 = r - 6 * j8;
            if (j9 == ke && e->lw <= r0 && r0 < e->up) {
                e->hv[r0] = e->HV[q][j9]; e->hc[r0] = e->HC[q][j9];
                e->fv[r0] = e->FV[q][j9]; e->fc[r0] = e->FC[q][j9];
                if (e->LocalL) { e->hb[r0] = e->HB[q][j9]; e->fb[r0] = e->FB[q][j9]; }
            }
        }
        if (is_imd_ && ++i < n_im) {
            e->imd = imds + i;
            e->mm3 = 3 * e->imd->mi;
            mm = e->a_left + (e->imd->mi - e->a_left - 1) / XN * XN;
            k9 = e->imd->mi - mm; k8 = k9 - 1;
        }
    }

    int a_left = e->a_left, a_right = e->a_right, b_left = e->b_left, b_right = e->b_right;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    if (e->LocalR && e->max_mr < a_right) {
        a_right = e->max_mr; b_right = e->max_nr;
    } else {
        const int rt = xh_last(e);
        e->max_ml = e->LocalL ? e->hb[rt] : a_left;
        a_right = e->max_mr; b_right = e->max_nr;
    }
    int val = e->max_val;
    int i = n_im;
    while (--i >= 0 && imds[i].mi > a_right) ;
    if (i < 0 && imds[0].mi > a_right) CPOS(0, 2) = b_right;
    int r = e->max_ulk;
    XhImd* imd;
    for ( ; i >= 0 && (imd = imds + i)->mi > e->max_ml; --i) {
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
    if (e->LocalL) {
        a_left = e->max_ml;
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
    free(e->mem); free(e);
    return 0;
}
