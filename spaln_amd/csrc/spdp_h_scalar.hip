// spdp_h_scalar.hip -- the reference's scalar protein x genome engine on the GPU.
//
//   spdh_scalar<FORWARD>   Aln2h1::forwardH_ng + initH_ng / lastH_ng        src/fwd2h1.cc:294-617, 143-292
//                          + Vmf::traceback + the record fix-up of trcbkalignH_ng
//                                                                             src/vmf.cc:125, src/fwd2h1.cc:2019-2036
//
// The -A0 engine (int32, row by row, top-NCAND donor list per row and codon phase, exact
// intron-length penalty, codons split by an intron re-scored through spjseq).  The -A2 / -A3
// dispatch needs it for sub-problems with fewer than 8 query rows (trcbkalignH_ng
// src/fwd2h1.cc:2005, HomScoreH_ng :3297): a handful of tiny problems per batch, so the mapping
// is one thread per problem with its two rows of {val, ptr, dir} and its Vmf records in global
// memory.  Correct at any size (tests run it on whole fixtures), not tuned.
// FORWARD = false is the score-only use (no Vmf records, as HomScoreH_ng runs it).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

#define HS_NCAND 4
#define HS_NOD 3
#define HS_NQUE 3
#define HS_AMB 2
#define HS_NEV (INT32_MIN / 16 * 7)

// TraceBackDir, src/aln.h:30-35
enum { D_DEAD, D_RSRV, D_DIAG, D_NEWD, D_VERT, D_SLA1, D_SLA2, D_VERL, D_HORI, D_HOR1, D_HOR2, D_HORL, D_NEWV, D_NEWH,
       D_SPIN = 16 };
// dir2nod / nod2dir / _is_diag / _is_vert / _is_hori (src/aln.h:50-69) as bit masks over dir & 15
__device__ __forceinline__ int hs_dir2nod(int d)
{
    d &= 15;
    if (d == D_DIAG || d == D_NEWD) return 0;
    if (d == D_VERT || d == D_SLA1 || d == D_SLA2 || d == D_NEWV) return 2;
    if (d == D_HORI || d == D_HOR1 || d == D_HOR2 || d == D_NEWH) return 1;
    if (d == D_VERL) return 4;
    if (d == D_HORL) return 3;
    return -1;
}
__device__ __forceinline__ bool hs_isdiag(int d) { d &= 15; return d == D_DIAG || d == D_NEWD; }
__device__ __forceinline__ bool hs_isvert(int d) { d &= 15; return (d >= D_VERT && d <= D_VERL) || d == D_NEWV; }
__device__ __forceinline__ bool hs_ishori(int d) { d &= 15; return (d >= D_HORI && d <= D_HORL) || d == D_NEWH; }

struct HsRvpd { int val, ptr, dir; };
struct HsRvpdj { int val, ptr, dir, jnc; };

template <bool FORWARD>
__global__ void spdh_scalar(HScalarArgs A)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= A.n_probs) return;
    const DevProblemH P = A.probs[pi];
    const DevScoringH* sc = A.sc;
    const int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, lgep = sc->lgep, codonk1 = sc->codonk1;
    const int gw1 = sc->g1, gw2 = sc->g2, gw3 = sc->g3, ge1 = A.gape1, ge2 = A.gape2;
    const bool spj = sc->spj;
    const int minl = A.minl;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int4* cols = A.cols + P.col_off;                  // .x: tron of n - 2 | flags, .y: sig3 candidates, .w: dinc
    const short4* aux = A.aux + P.col_off;                  // {sigS, sigT, sigE, sig5} raw
    HsRvpd* const buf = reinterpret_cast<HsRvpd*>(A.work + P.bnd_off);
    HsRvpd* const hh0 = buf - lw + 3;
    HsRvpd* const hh1 = hh0 + width;
    int3* vrec = A.vmf + P.tb_off;
    const int vcap = (int) P.imd_off;
    int vn = 0;
    bool vover = false;
    auto vadd = [&](int m, int n, int p) -> int {
        if (!FORWARD) return 0;
        if (vn < vcap) vrec[vn] = make_int3(m, n, p); else vover = true;
        return vn++;
    };
    auto gext3 = [&](int i) { return i > codonk1 ? lgep : gep; };
    auto acode = [&](int i) -> int { return (i < 0 || i >= P.a_len) ? HS_AMB : acod[i]; };
    auto bcode = [&](int i) -> int { return (i < 0 || i > P.b_len) ? HS_AMB : ((cols[i + 2].x >> 16) & 0xff); };
    auto mtx = [&](int aa, int tron) -> int { return sc->mtx[aa * 32 + tron]; };
    auto ipen = [&](int len) -> int {
        if (len < 0) return -32768;
        if (len >= A.intpen_len) len = A.intpen_len - 1;
        return A.intpen[len];
    };
    // spjscr(jnc, nb) with the acceptor's own sig3 passed in (column record of the cell it is read at)
    auto spjscr = [&](int jnc, int nb, int s3) -> int {
        return ipen(nb - jnc) + s3 + A.t53[16 * ((cols[jnc].w >> 4) & 15) + (cols[nb].w & 15)];
    };
    auto spjseq = [&](int n5, int n3, int& c0, int& c1) {
        c0 = c1 = HS_AMB;
        if (n5 < bl || n3 >= br) return;
        const int t0 = bcode(n5 - 2), t1 = bcode(n5 - 1), t2 = bcode(n3), t3 = bcode(n3 + 1);
        if (t0 >= 32 || t1 >= 32 || t2 >= 32 || t3 >= 32) return;
        const int w0 = A.mid[t0], w1 = A.mid[t1], w2 = A.mid[t2], w3 = A.mid[t3];
        if (w0 > 3 || w1 > 3 || w2 > 3 || w3 > 3) return;
        c0 = A.tron_of[16 * w0 + 4 * w1 + w2];
        c1 = A.tron_of[16 * w1 + 4 * w2 + w3];
    };
    const HsRvpd black = {HS_NEV, 0, 0};
    for (int i = 0; i < 2 * width + 8; ++i) buf[i] = black;
    vadd(0, 0, 0);                                          // skip 0-th record

    // ---- initH_ng
    {
        int n = bl;
        int r = bl - 3 * al;
        int rr = br - 3 * al;
        const int dir = a_exgl ? D_DEAD : D_DIAG;
        int jnc[3] = {n, 0, 0};
        int bb = n + 1;
        HsRvpd* h = hh0 + r;
        h->val = (a_exgl && aux[bb].x > 0) ? aux[bb].x : 0;
        h->dir = dir;
        h->ptr = vadd(al, n, 0);
        if (a_exgl) {
            if (up < rr) rr = up;
            for (int i = 1; ++r <= rr; ++i) {
                ++h; ++bb; ++n;
                const int sS = aux[bb].x > 0 ? aux[bb].x : 0;
                if (i < 3) {
                    h->val = sS; h->dir = dir; h->ptr = vadd(al, n, 0);
                    jnc[i] = n;
                } else {
                    *h = h[-3];
                    const int k = n - jnc[i % 3];
                    if (k == 3 && !(a_exgl & 1)) h->val += gop;
                    if (!(a_exgl & 2)) h->val += gext3(k);
                    h->val += aux[bb - 3].z;
                    h->dir = D_HORI;
                    int x = h[-1].val + gw1;
                    if (x > h->val) { *h = h[-1]; h->val = x; h->dir = D_HOR1; }
                    x = h[-2].val + gw2;
                    if (x > h->val) { *h = h[-2]; h->val = x; h->dir = D_HOR2; }
                }
                if (h->val < sS) {
                    h->val = sS; h->dir = D_DEAD; h->ptr = vadd(al, n, 0);
                    jnc[i % 3] = n;
                }
            }
        }
        r = bl - 3 * al;
        rr = bl - 3 * ar;
        h = hh0 + r - 1;
        if (lw > rr) rr = lw;
        for (int i = 1; --r >= rr; ++i, --h) {
            if (b_exgl == 1) { h->val = 0; h->dir = D_DEAD; h->ptr = 0; }
            else if (i <= 3) {
                *h = h[i];
                if (!(b_exgl & 2)) h->val += gep;
                if (!(b_exgl & 1)) h->val += gop;
                if (i < 3) h->val += A.extragop;
                h->dir = D_VERT;
            } else {
                *h = h[3];
                if (!(b_exgl & 2)) h->val += gext3(i);
            }
        }
    }

    int maxh_val = HS_NEV, maxh_m = al, maxh_n = bl, maxh_p = 0;
    int m = al;
    if (!a_exgl) --m;
    int n1 = 3 * m + lw - 1;
    int n2 = 3 * m + up;
    for ( ; ++m <= ar; ) {
        n1 += 3; n2 += 3;
        const int n0 = max(n1, bl);
        const int n9 = min(n2, br);
        int n = n0;
        const int r0 = n - 3 * m;
        HsRvpd e1[HS_NQUE] = {black, black, black};
        if (!b_exgl && m == al) { e1[2] = hh0[r0]; e1[2].val = gw3; }
        HsRvpd* h = hh0 + r0;
        HsRvpd* f = hh1 + r0;
        const int aa0 = acode(m - 1), aa1 = acode(m);
        HsRvpdj hl[3][HS_NCAND + 1];
        int nx[3][HS_NCAND + 1];
        for (int ph = 0; ph < 3; ++ph)
            for (int l = 0; l <= HS_NCAND; ++l) { hl[ph][l].val = HS_NEV; hl[ph][l].ptr = hl[ph][l].dir = hl[ph][l].jnc = 0; nx[ph][l] = l; }
        int ncand[3] = {-1, -1, -1};
        for (int q = 0; n <= n9; ++n, ++h, ++f) {
            int x, y;
            const int4 col = cols[n];
            const int sigE = (n > bl && n >= 2) ? aux[n - 2].z : 0;     // position -1 is not in the arrays
            HsRvpd* const eq1 = e1 + q;
            HsRvpd* hf[HS_NOD] = {h, eq1, f};
            const HsRvpd hq = *h;                            // previous state
            HsRvpd* mx = h;
            if (m != al) {
                // diagonal match
                if (n < bl + 3) *h = black;
                else {
                    h->val += mtx(aa0, bcode(n - 2)) + sigE;
                    h->dir = hs_isdiag(hq.dir) ? D_DIAG : D_NEWD;
                }
                // vertical gap extension, deletions of 1 / 2 nt, of a codon
                y = f[3].val + gep;
                x = h[1].val + (hs_isvert(h[1].dir) ? ge1 : gw1);
                if (x > y) { f->val = x; f->dir = D_SLA2; f->ptr = h[1].ptr; }
                else f->val = y;
                x = h[2].val + (hs_isvert(h[2].dir) ? ge2 : gw2);
                if (x > f->val) { f->val = x; f->dir = D_SLA1; f->ptr = h[2].ptr; }
                x = h[3].val + gw3;
                if (x >= f->val) { f->val = x; f->dir = D_VERT; f->ptr = h[3].ptr; }
                else if (y >= f->val) { f->val = y; f->dir = D_VERT; f->ptr = f[3].ptr; }
                if (f->val > mx->val) mx = f;
            }
            // insertions of a codon, of 2 nt, of 1 nt
            if (n > n0 + 2) {
                x = h[-3].val + gw3;
                y = eq1->val += gep;
                if (x > y) { *eq1 = h[-3]; eq1->val = x; }
                eq1->val += sigE;
                eq1->dir = (eq1->dir & D_SPIN) + D_HORI;
            }
            if (n > n0 + 1) {
                x = h[-2].val + gw2;
                if (x > eq1->val) { *eq1 = h[-2]; eq1->val = x; eq1->dir = (eq1->dir & D_SPIN) + D_HOR2; }
            }
            x = h[-1].val + gw1;
            if (x > eq1->val) { *eq1 = h[-1]; eq1->val = x; eq1->dir = (eq1->dir & D_SPIN) + D_HOR1; }
            if (eq1->val > mx->val) mx = eq1;
            if (++q == HS_NQUE) q = 0;

            const unsigned fl = (unsigned) col.x >> 24;
            // intron 3' boundary: flags bits 0-1 = phase + 2 (0: none), bit 2: phase +1 as well
            if (spj && (fl & 3)) {
                int phs = (int) (fl & 3) - 2;
                int s3 = (int) (short) (col.y & 0xffff);
                for (;;) {
                    const int nb = n - phs;
                    const int* pnx = nx[phs + 1];
                    const HsRvpdj* maxphl[HS_NOD] = {nullptr, nullptr, nullptr};
                    for (int l = 0; l <= ncand[phs + 1]; ++l) {
                        const HsRvpdj* phl = hl[phs + 1] + pnx[l];
                        if (phs == 1 && phl->dir == 2) continue;
                        if (nb - phl->jnc < minl) continue;
                        x = phl->val + spjscr(phl->jnc, nb, s3);
                        if (phl->dir == 0 && phs) {
                            int c0, c1;
                            spjseq(phl->jnc, nb, c0, c1);
                            if (phs == 1) x += mtx(aa0, c0);
                            else x += mtx(aa1, c1) - mtx(aa1, bcode(n + 1)) - aux[n + 1].z;
                        }
                        HsRvpd* from = hf[phl->dir];
                        if (x > from->val) { from->val = x; maxphl[phl->dir] = phl; }
                    }
                    for (int d = 0; d < HS_NOD; ++d) {
                        const HsRvpdj* phl = maxphl[d];
                        if (!phl) continue;
                        HsRvpd* from = hf[d];
                        if (FORWARD) {
                            const int inner = vadd(m, phl->jnc + phs, phl->ptr);
                            from->ptr = vadd(m, n, inner);
                        }
                        from->dir = (phl->dir == 0 ? D_DIAG : (phl->dir == 1 ? D_HORI : D_VERT)) | D_SPIN;
                        if (from->val > mx->val) mx = from;
                    }
                    if ((fl & 4) && phs == -1) { phs = 1; s3 = (int) (short) ((unsigned) col.y >> 16); continue; }   // AGAG
                    break;
                }
            }

            // optimal path
            y = h->val;
            if (h != mx) *h = *mx;
            else if (Local && y > hq.val) {
                if (LocalL && hq.dir == 0 && !(h->dir & D_SPIN)) h->ptr = vadd(m - 1, n - 3, 0);
                else if (LocalR && y > maxh_val) { maxh_val = y; maxh_p = h->ptr; maxh_m = m; maxh_n = n; }
            }
            if (LocalL && h->val <= 0) h->val = h->dir = 0;
            else if (FORWARD && h->dir == D_NEWD) h->ptr = vadd(m - 1, n - 3, h->ptr);

            // intron 5' boundary: flags bits 3-4 = phase + 2, bit 5: phase +1 as well
            if (spj && ((fl >> 3) & 3)) {
                int phs = (int) ((fl >> 3) & 3) - 2;
                for (;;) {
                    const int nb = n - phs;
                    const int sigJ = aux[nb].w;
                    const int hd = hs_dir2nod(mx->dir);
                    for (int k = (hd == 0 || phs == 1) ? 0 : 1; k < HS_NOD; ++k) {
                        const bool crossspj = phs == 1 && k == 0;
                        const HsRvpd* src = crossspj ? &hq : hf[k];
                        if (!src->dir || (src->dir & D_SPIN)) continue;     // no orphan exon
                        if (!crossspj && k != hd && hd >= 0) {
                            y = mx->val;
                            if (hd == 0 || (k - hd) % 2) y += (k / 2 == 1) ? gop : 0;   // GOP[k / 2]
                            if (src->val <= y) continue;                    // prune
                        }
                        x = src->val + sigJ;
                        HsRvpdj* phl = hl[phs + 1];
                        int* pnx = nx[phs + 1];
                        int& nc = ncand[phs + 1];
                        int l = nc < HS_NCAND ? ++nc : HS_NCAND;
                        while (--l >= 0) {
                            if (x >= phl[pnx[l]].val) { const int t = pnx[l]; pnx[l] = pnx[l + 1]; pnx[l + 1] = t; }
                            else break;
                        }
                        if (++l < HS_NCAND) {
                            phl += pnx[l];
                            phl->val = x; phl->jnc = nb; phl->dir = k; phl->ptr = src->ptr;
                        } else --nc;
                    }
                    if ((fl & 32) && phs == -1) { phs = 1; continue; }      // GTGT
                    break;
                }
            }
        }
    }

    DevResultH R;
    R.score = HS_NEV; R.mr = ar; R.nr = br; R.maxt = 0; R.maxr = 0; R.pad[0] = R.pad[1] = R.pad[2] = 0;
    int ptr = 0;
    if (!LocalR || maxh_m == ar) {                          // ---- lastH_ng
        int glen[3] = {0, 0, 0};
        int rw = lw;
        const int m3 = 3 * ar;
        int rf = bl - m3;
        if (rf > rw) rw = rf; else rf = rw;
        HsRvpd* h = hh0 + rw;
        HsRvpd* const h9 = hh0 + br - m3;
        HsRvpd* mx = h9;
        int bb = rw + m3;
        bool done = false;
        if (a_exgr) {
            for (int ph = 0; h <= h9; ++h, ++bb, ++rf, ph = (ph == 2 ? 0 : ph + 1)) {
                glen[ph] += 3;
                int c0 = h->val, c1 = HS_NEV, c2 = HS_NEV;
                if (rf - rw >= 3 && h[-3].dir != D_DEAD) {
                    c1 = h[-3].val + aux[bb - 2].z;
                    if (!(a_exgr & 2)) c1 += gext3(glen[ph]);
                    if (!(a_exgr & 1) && glen[ph] == 3) c1 += gop;
                    if (aux[bb - 2].y > 0 && !(h->dir & D_SPIN)) c2 = h[-3].val + aux[bb - 2].y;
                }
                const int sig5 = (Local && aux[bb].w > 0) ? aux[bb].w : 0;
                c0 += sig5;
                c1 += sig5;
                int k = 0, best = c0;
                if (c1 > best) { k = 1; best = c1; }
                if (c2 > best) { k = 2; best = c2; }
                if (k == 0) { if (!hs_ishori(h->dir)) glen[ph] = 0; }
                else if (k == 1) { *h = h[-3]; h->dir = D_HORI; h->val = best - sig5; }
                else {
                    *h = h[-3];
                    h->dir = D_DEAD;
                    h->val = best;
                    if (FORWARD && h->val > mx->val) h->ptr = vadd(ar, rf + m3 - 3, h->ptr);
                }
                if (h->val > mx->val) mx = h;
            }
        } else {
            bb += (int) (h9 - h);
            const int y = h9[-3].val + aux[bb - 2].y;
            if (y > h9->val) { *h9 = h9[-3]; h9->val = y; h9->dir = D_HORI; }
        }
        if (b_exgr == 1) {
            rw = min(up, br - 3 * al);
            int g[3] = {HS_NEV, HS_NEV, HS_NEV};
            h = hh0 + rw - 3;
            for (int ph = 0; h >= h9; --h) {
                int x = h[3].val;
                if (!(b_exgr & 1)) x += gop;
                if (x > g[ph]) g[ph] = x;
                if (!(b_exgr & 2)) g[ph] += gep;
                if (h->val > g[ph]) g[ph] = HS_NEV;
                else if (g[ph] > mx->val) { mx = h; mx->val = g[ph]; }
                if (++ph == 3) ph = 0;
            }
        } else if (b_exgr == 2) {
            mx = hh1 + br - m3;
            mx->ptr = vadd(ar, br, mx->ptr);
            done = true;
        }
        if (!done) {
            int pp = (int) (mx - h9);
            rf = ar;
            rw = br;
            if (pp > 0) { rf -= (pp + 2) / 3; if (pp %= 3) rw -= (3 - pp); }
            else if (pp < 0) rw += pp;
            mx->ptr = vadd(rf, rw, mx->ptr);
        }
        R.score = mx->val;
        ptr = mx->ptr;
    } else {
        R.score = maxh_val;
        ptr = vadd(maxh_m, maxh_n, maxh_p);
    }
    A.res[pi] = R;
    if (!FORWARD) return;

    // Vmf::traceback(ptr) + the fix-up of trcbkalignH_ng
    int2* out = A.skl + (int64_t) pi * A.skl_cap;
    int cnt = 0, status = vover ? -3 : 0;
    if (ptr && !vover) {
        int3 sv = vrec[ptr];
        int lm = 0, ln = 0;
        for (;;) {
            if (cnt < A.skl_cap) out[cnt] = make_int2(sv.x, sv.y); else status = -1;
            lm = sv.x; ln = sv.y; ++cnt;
            if (!sv.z) break;
            sv = vrec[sv.z];
        }
        const int rd = Local ? 0 : ((ln - 3 * lm) - bl + 3 * al);
        if (rd) {
            const int2 rec = rd > 0 ? make_int2(al, bl + rd) : make_int2(al - rd / 3, bl);
            if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
            ++cnt;
        }
    }
    A.n_skl[pi] = status ? status : cnt;
}

extern "C" hipError_t spdh_launch_scalar(int forward, const HScalarArgs* a, hipStream_t stream)
{
    HScalarArgs A = *a;
    // few problems: one per wave (a wave of divergent one-thread problems runs them one after another)
    const int per = A.n_probs <= 8192 ? 1 : 64;
    const dim3 grd((A.n_probs + per - 1) / per), blk(per);
    if (forward) hipLaunchKernelGGL(spdh_scalar<true>, grd, blk, 0, stream, A);
    else hipLaunchKernelGGL(spdh_scalar<false>, grd, blk, 0, stream, A);
    return hipGetLastError();
}

// ---- scalar unidirectional Hirschberg ---------------------------------------------------------
//   spdh_scalar_udh   Aln2h1::hirschbergH_ng + hinitH_ng / hlastH_ng   src/fwd2h1.cc:1085-1520, 941-1083
//                     with UdhIntermediate(lub = true)                  src/udh_intermediate.h:29-66
// The -A0 linear-space engine of the protein path: the recurrence above with every state carrying
// the diagonal range visited since the last intermediate row, its start row and a link to where it
// crossed the previous intermediate; the same thread walks the links back into cpos rows ([8] / [9] =
// diagonal bounds of the slab below a row: the slab's window under -A0).  Boundary rules are
// hinitH_ng's / hlastH_ng's, which are not forwardH_ng's (one `jnc` for all frames of the leading gap,
// termination codons gated by algmode.lcl & 2, `dir & DIAG` as the diagonal-continuation test).
// res.pad[0] = -3 flags inputs on which the reference itself indexes outside its arrays.
struct HsRvdwml { int val, dir, upr, lwr, ml, ulk; };
struct HsRvdwmlj { int val, dir, upr, lwr, ml, ulk, jnc; };

__global__ void spdh_scalar_udh(HScalarArgs A)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= A.n_probs) return;
    const DevProblemH P = A.probs[pi];
    const DevScoringH* sc = A.sc;
    const int EOU = 0x7fffffff - 2;
    int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, n_im = P.n_im;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, lgep = sc->lgep, codonk1 = sc->codonk1;
    const int gw1 = sc->g1, gw2 = sc->g2, gw3 = sc->g3, ge1 = A.gape1, ge2 = A.gape2;
    const bool spj = sc->spj;
    const int minl = A.minl;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int4* cols = A.cols + P.col_off;
    const short4* aux = A.aux + P.col_off;
    HsRvdwml* const wbuf = reinterpret_cast<HsRvdwml*>(A.work + P.bnd_off);     // 2 * width + 8 states
    HsRvdwml* const hh0 = wbuf - lw + 3;
    HsRvdwml* const hh1 = hh0 + width;
    int* const imd_base = A.imd + P.imd_off;
    const int64_t us = 2 * (int64_t) width;
    auto IM = [&](int i, int arr, int k, int r) -> int& { return imd_base[(int64_t) i * 4 * us + arr * us + (int64_t) k * width + (r - lw + 1)]; };
    enum { HLNK = 0, VLNK = 1, LWRB = 2, UPRB = 3 };
    int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    auto gext3 = [&](int i) { return i > codonk1 ? lgep : gep; };
    auto acode = [&](int i) -> int { return (i < 0 || i >= P.a_len) ? HS_AMB : acod[i]; };
    auto bcode = [&](int i) -> int { return (i < 0 || i > P.b_len) ? HS_AMB : ((cols[i + 2].x >> 16) & 0xff); };
    auto mtx = [&](int aa, int tron) -> int { return sc->mtx[aa * 32 + tron]; };
    auto ipen = [&](int len) -> int {
        if (len < 0) return -32768;
        if (len >= A.intpen_len) len = A.intpen_len - 1;
        return A.intpen[len];
    };
    auto spjscr = [&](int jnc, int nb, int s3) -> int {
        return ipen(nb - jnc) + s3 + A.t53[16 * ((cols[jnc].w >> 4) & 15) + (cols[nb].w & 15)];
    };
    auto spjseq = [&](int n5, int n3, int& c0, int& c1) {
        c0 = c1 = HS_AMB;
        if (n5 < P.b_left || n3 >= P.b_right) return;
        const int t0 = bcode(n5 - 2), t1 = bcode(n5 - 1), t2 = bcode(n3), t3 = bcode(n3 + 1);
        if (t0 >= 32 || t1 >= 32 || t2 >= 32 || t3 >= 32) return;
        const int w0 = A.mid[t0], w1 = A.mid[t1], w2 = A.mid[t2], w3 = A.mid[t3];
        if (w0 > 3 || w1 > 3 || w2 > 3 || w3 > 3) return;
        c0 = A.tron_of[16 * w0 + 4 * w1 + w2];
        c1 = A.tron_of[16 * w1 + 4 * w2 + w3];
    };
    auto mi_of = [&](int i) { return P.a_left + (i + 1) * P.imd_intvl; };
    for (int i = 0; i <= n_im; ++i) for (int c = 0; c < 10; ++c) CPOS(i, c) = EOU;
    int r = bl - 3 * ar;
    const HsRvdwml black = {HS_NEV, 0, r, r, 0, EOU};
    for (int i = 0; i < 2 * width + 8; ++i) wbuf[i] = black;
    for (int i = 0; i < n_im; ++i)
        for (int64_t k = 0; k < us; ++k) {
            imd_base[(int64_t) i * 4 * us + k] = EOU;
            imd_base[(int64_t) i * 4 * us + us + k] = EOU;
            imd_base[(int64_t) i * 4 * us + 2 * us + k] = 0x7fffffff;
            imd_base[(int64_t) i * 4 * us + 3 * us + k] = (int) 0x80000000;
        }

    // ---- hinitH_ng
    {
        int n = bl;
        r = bl - 3 * al;
        const int r0 = r;
        int rr = br - 3 * al;
        const int dir = a_exgl ? D_DEAD : D_DIAG;
        int bb = n + 1;
        HsRvdwml* h = hh0 + r;
        h->val = (a_exgl && aux[bb].x > 0) ? aux[bb].x : 0;
        h->dir = dir;
        h->lwr = h->upr = h->ulk = r0;
        h->ml = al;
        if (a_exgl) {
            if (up < rr) rr = up;
            int jnc = n;
            for (int i = 1; ++r <= rr; ++i) {
                ++h; ++bb; ++n;
                const int sS = aux[bb].x > 0 ? aux[bb].x : 0;
                if (i < 3) { h->val = sS; h->dir = dir; h->lwr = h->ulk = r; h->ml = al; }
                else {
                    *h = h[-3];
                    const int d = n - jnc;
                    if (!(a_exgl & 1) && d == 3) h->val += gop;
                    if (!(a_exgl & 2)) h->val += gext3(d);
                    h->val += aux[bb - 3].z;
                    h->dir = D_HORI;
                    int x = h[-1].val + gw1;
                    if (x > h->val) { *h = h[-1]; h->val = x; h->dir = D_HOR1; }
                    x = h[-2].val + gw2;
                    if (x > h->val) { *h = h[-2]; h->val = x; h->dir = D_HOR2; }
                }
                if (h->val < sS) { h->val = sS; h->dir = D_DEAD; jnc = n; h->lwr = h->ulk = r; }
                h->upr = r;
            }
        }
        r = r0;
        rr = bl - 3 * ar;
        if (lw > rr) rr = lw;
        h = hh0 + r - 1;
        for (int i = 1; --r >= rr; ++i, --h) {
            if (b_exgl == 1) {
                h->val = 0; h->dir = D_DEAD;
                h->upr = h->lwr = h->ulk = r;
                h->ml = al + i / 3;
            } else if (i <= 3) {
                *h = h[i];
                if (!(b_exgl & 2)) h->val += gep;
                if (!(b_exgl & 1)) h->val += gop;
                if (i < 3) h->val += A.extragop;
                h->dir = D_VERT;
                h->ml += i / 3;
                h->lwr = h->ulk = r;
            } else {
                *h = h[3];
                if (!(b_exgl & 2)) h->val += gext3(i);
                h->lwr = h->ulk = r;
                ++h->ml;
            }
        }
    }

    int ii = 0;
    int mm = mi_of(0);
    int rlst[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int maxh_val = HS_NEV, maxh_upr = 0, maxh_lwr = 0, maxh_ml = al, maxh_ulk = 0, maxh_mr = ar, maxh_nr = br;
    int m = al;
    if (!a_exgl) --m;
    int n1 = 3 * m + lw - 1;
    int n2 = 3 * m + up;
    for ( ; ++m <= ar; ) {
        n1 += 3; n2 += 3;
        const int n0 = max(n1, bl);
        const int n9 = min(n2, br);
        const bool is_imd = m == mm;
        int n = n0;
        r = n - 3 * m;
        HsRvdwml e1[HS_NQUE] = {black, black, black};
        if (!b_exgl && m == al) { e1[2] = hh0[r]; e1[2].val += gw3; }
        HsRvdwml* h = hh0 + r;
        HsRvdwml* f = hh1 + r;
        const int aa0 = acode(m - 1), aa1 = acode(m);
        HsRvdwmlj hl[3][HS_NCAND + 1];
        int nx[3][HS_NCAND + 1];
        for (int ph = 0; ph < 3; ++ph)
            for (int l = 0; l <= HS_NCAND; ++l) {
                hl[ph][l].val = HS_NEV; hl[ph][l].dir = 0; hl[ph][l].upr = (int) 0x80000000; hl[ph][l].lwr = 0x7fffffff;
                hl[ph][l].ml = 0; hl[ph][l].ulk = EOU; hl[ph][l].jnc = 0;
                nx[ph][l] = l;
            }
        int ncand[3] = {-1, -1, -1};
        int q = 0;
        for ( ; n <= n9; ++n, ++r, ++h, ++f) {
            int x, y;
            const int4 col = cols[n];
            const int sigE = (n > bl && n >= 2) ? aux[n - 2].z : 0;
            HsRvdwml* const eq1 = e1 + q;
            HsRvdwml* hf[HS_NOD] = {h, eq1, f};
            const HsRvdwml hq = *h;
            HsRvdwml* mx = h;
            if (m != al) {
                if (n < bl + 3) *h = black;
                else {
                    h->val += mtx(aa0, bcode(n - 2)) + sigE;
                    h->dir = (hq.dir & D_DIAG) ? D_DIAG : D_NEWD;
                }
                y = f[3].val + gep;
                x = h[1].val + (hs_isvert(h[1].dir) ? ge1 : gw1);
                if (x > y) { *f = h[1]; f->val = x; f->dir = D_SLA2; }
                else f->val = y;
                x = h[2].val + (hs_isvert(h[2].dir) ? ge2 : gw2);
                if (x > f->val) { *f = h[2]; f->val = x; f->dir = D_SLA1; }
                x = h[3].val + gw3;
                if (x >= f->val) { *f = h[3]; f->val = x; f->dir = D_VERT; }
                else if (y >= f->val) { *f = f[3]; f->val = y; f->dir = D_VERT; }
                if (f->val >= mx->val) mx = f;
            }
            if (n > n0 + 2) {
                x = h[-3].val + gw3;
                y = eq1->val += gep;
                if (x > y) { *eq1 = h[-3]; eq1->val = x; }
                eq1->val += sigE;
                eq1->dir = (eq1->dir & D_SPIN) + D_HORI;
            }
            if (n > n0 + 1) {
                x = h[-2].val + gw2;
                if (x > eq1->val) { *eq1 = h[-2]; eq1->val = x; eq1->dir = D_HOR2; }
            }
            x = h[-1].val + gw1;
            if (x > eq1->val) { *eq1 = h[-1]; eq1->val = x; eq1->dir = D_HOR1; }
            if (eq1->val > mx->val) mx = eq1;
            if (++q == HS_NQUE) q = 0;

            const unsigned fl = (unsigned) col.x >> 24;
            bool spj3 = false;
            if (spj && (fl & 3)) {
                int phs = (int) (fl & 3) - 2;
                int s3 = (int) (short) (col.y & 0xffff);
                for (;;) {
                    const int nb = n - phs;
                    const int* pnx = nx[phs + 1];
                    const HsRvdwmlj* maxphl[HS_NOD] = {nullptr, nullptr, nullptr};
                    for (int l = 0; l <= ncand[phs + 1]; ++l) {
                        const HsRvdwmlj* phl = hl[phs + 1] + pnx[l];
                        if (phs == 1 && phl->dir == 2) continue;
                        if (nb - phl->jnc < minl) continue;
                        x = phl->val + spjscr(phl->jnc, nb, s3);
                        if (phl->dir == 0 && phs) {
                            int c0, c1;
                            spjseq(phl->jnc, nb, c0, c1);
                            if (phs == 1) x += mtx(aa0, c0);
                            else x += mtx(aa1, c1) - mtx(aa1, bcode(n + 1)) - aux[n + 1].z;
                        }
                        HsRvdwml* from = hf[phl->dir];
                        if (x > from->val) { from->val = x; maxphl[phl->dir] = phl; }
                    }
                    int maxk = HS_NOD;
                    for (int k = 0; k < HS_NOD; ++k) {
                        const HsRvdwmlj* phl = maxphl[k];
                        if (!phl) continue;
                        HsRvdwml* from = hf[k];
                        from->dir = (phl->dir == 0 ? D_DIAG : (phl->dir == 1 ? D_HORI : D_VERT)) | D_SPIN;
                        from->upr = max(phl->upr, r);
                        from->lwr = min(phl->lwr, r);
                        from->ml = phl->ml;
                        from->ulk = phl->ulk;
                        if (from->val >= mx->val) { maxk = k; mx = from; }
                    }
                    if (is_imd && maxk < HS_NOD) {
                        const HsRvdwmlj* phl = maxphl[maxk];
                        IM(ii, HLNK, 0, r) = phl->ulk;
                        mx->ulk = rlst[q] = r;
                        spj3 = true;
                        if (maxk == 0) {
                            if ((phl = maxphl[1]) && hf[1]->val > mx->val + gop) {
                                hf[1]->ulk = r + width;
                                IM(ii, HLNK, 1, r) = phl->ulk;
                            }
                            if (maxphl[2] && hf[2]->val > mx->val + gop) hf[2]->ulk = r + width;
                        }
                    }
                    if ((fl & 4) && phs == -1) { phs = 1; s3 = (int) (short) ((unsigned) col.y >> 16); continue; }
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

            const int hd = hs_dir2nod(mx->dir);
            if (spj && ((fl >> 3) & 3)) {
                int phs = (int) ((fl >> 3) & 3) - 2;
                for (;;) {
                    const int nb = n - phs;
                    const int sigJ = aux[nb].w;
                    for (int k = (hd == 0 || phs == 1) ? 0 : 1; k < HS_NOD; ++k) {
                        const bool crossspj = phs == 1 && k == 0;
                        const HsRvdwml* src = crossspj ? &hq : hf[k];
                        if (!src->dir || (src->dir & D_SPIN)) continue;
                        if (k != hd && !crossspj && hd >= 0) {
                            y = mx->val;
                            if (hd == 0 || (k - hd) % 2) y += (k / 2 == 1) ? gop : 0;
                            if (src->val <= y) continue;
                        }
                        x = src->val + sigJ;
                        HsRvdwmlj* phl = hl[phs + 1];
                        int* pnx = nx[phs + 1];
                        int& nc = ncand[phs + 1];
                        int l = nc < HS_NCAND ? ++nc : HS_NCAND;
                        while (--l >= 0) {
                            if (x >= phl[pnx[l]].val) { const int t = pnx[l]; pnx[l] = pnx[l + 1]; pnx[l + 1] = t; }
                            else break;
                        }
                        if (++l < HS_NCAND) {
                            phl += pnx[l];
                            phl->val = x; phl->jnc = nb; phl->dir = k;
                            phl->upr = src->upr; phl->lwr = src->lwr; phl->ml = src->ml;
                            if (is_imd) {
                                if (k == 1) IM(ii, HLNK, 0, r) = rlst[q];
                                phl->ulk = r;
                            } else phl->ulk = src->ulk;
                        } else --nc;
                    }
                    if ((fl & 32) && phs == -1) { phs = 1; continue; }
                    break;
                }
            }

            if (is_imd) {
                if (hd == 0) rlst[q] = r;
                else if (!spj3 && hd % 2) IM(ii, HLNK, 0, r) = rlst[q];
                for (int k = 0; k < 2; ++k) {
                    HsRvdwml* g = hf[2 * k];
                    IM(ii, VLNK, k, r) = g->ulk;
                    IM(ii, LWRB, k, r) = min(r, g->lwr);
                    IM(ii, UPRB, k, r) = max(r, g->upr);
                    g->lwr = g->upr = r;
                    g->ulk = r + k * width;
                }
            }
        }
        if (is_imd && ++ii < n_im) mm = mi_of(ii);
    }

    int flag = 0;
    const int rr = br - 3 * ar;
    if (LocalR) {
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        ar = maxh_mr; br = maxh_nr;
        if (i < 0) i = 0;
        CPOS(i, 8) = maxh_lwr;
        CPOS(i, 9) = maxh_upr;
    } else {        // ---- hlastH_ng
        int glen[3] = {0, 0, 0};
        const int m3 = 3 * ar;
        int rw = lw;
        int rf = bl - m3;
        if (rf > rw) rw = rf; else rf = rw;
        HsRvdwml* h = hh0 + rw;
        HsRvdwml* const h9 = hh0 + br - m3;
        HsRvdwml* mx = h9;
        int bb = rw + m3;
        if (a_exgr) {
            for (int ph = 0; h <= h9; ++h, ++bb, ++rf, ph = (ph == 2 ? 0 : ph + 1)) {
                glen[ph] += 3;
                int c0 = h->val, c1 = HS_NEV, c2 = HS_NEV;
                if (rf - rw >= 3 && h[-3].dir != D_DEAD) {
                    c1 = h[-3].val + aux[bb - 2].z;
                    if (!(a_exgr & 2)) c1 += gext3(glen[ph]);
                    if (glen[ph] == 3 && !(a_exgr & 1)) c1 += gop;
                    if (sc->term_codon && !(h->dir & D_SPIN)) c2 = h[-3].val + aux[bb - 2].y;
                }
                const int sig5 = (Local && aux[bb].w > 0) ? aux[bb].w : 0;
                c0 += sig5;
                c1 += sig5;
                int k = 0, best = c0;
                if (c1 > best) { k = 1; best = c1; }
                if (c2 > best) { k = 2; best = c2; }
                if (k == 0) { if (!hs_ishori(h->dir)) glen[ph] = 0; }
                else if (k == 1) { *h = h[-3]; h->dir = D_HORI; h->val = best - sig5; }
                else { *h = h[-3]; h->dir = D_DEAD; h->val = best; h->upr = max(rf, h->upr); }
                if (h->val > mx->val) mx = h;
            }
        } else {
            bb += (int) (h9 - h);
            const int y = h9[-3].val + aux[bb - 2].y;
            if (y > h9->val) { *h9 = h9[-3]; h9->val = y; h9->dir = D_HORI; h9->upr = max(br - m3, h9->upr); }
        }
        if (b_exgr == 1) {
            rw = min(up, br - 3 * al);
            for (h = hh0 + rw; h > h9; --h, --rw) {
                const int x = h->val + ((rw % 3) ? A.extragop : 0);
                if (x > mx->val) { mx = h; mx->val = x; }
            }
        } else if (b_exgr == 2)
            mx = hh1 + br - m3;
        maxh_val = mx->val; maxh_lwr = mx->lwr; maxh_upr = mx->upr; maxh_ulk = mx->ulk; maxh_ml = mx->ml;
        r = (int) (mx - hh0);
        if (b_exgr && rr < r) ar = (br - r) / 3;
        if (a_exgr && rr > r) br = 3 * ar + r;
    }
    int i = n_im;
    while (--i >= 0 && mi_of(i) > ar) ;
    if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
    r = br - 3 * ar;
    CPOS(i + 1, 8) = min(maxh_lwr, r);
    CPOS(i + 1, 9) = max(maxh_upr, r);
    r = maxh_ulk;
    for ( ; i >= 0 && mi_of(i) > maxh_ml; --i) {
        int c = 0, d = 0;
        for ( ; r > up; r -= width) ++d;
        if (d > 1 || r < lw - 1) { flag = -3; break; }
        const int mi = mi_of(i);
        if (IM(i, VLNK, d, r) < EOU) {
            CPOS(i, c++) = mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            const int mm3 = 3 * mi;
            for (int rp = IM(i, HLNK, d, r); lw <= rp && rp < up && r != rp; rp = IM(i, HLNK, 0, r = rp)) {
                if (c >= 6) { flag = -3; break; }
                CPOS(i, c++) = r + mm3;
            }
            if (flag) break;
            CPOS(i, c++) = r + mm3;
            CPOS(i, c) = EOU;
            CPOS(i, 8) = IM(i, LWRB, d, r);
            CPOS(i, 9) = IM(i, UPRB, d, r);
            r = IM(i, VLNK, d, r);
            if (r == EOU) break;
        } else
            CPOS(i, 0) = EOU;
    }
    if (!flag) {
        for ( ; r > up; r -= width) ;
        if (LocalL) { al = maxh_ml; bl = r + 3 * maxh_ml; }
        else {
            const int rl = bl - 3 * al;
            if (b_exgl && rl > r) {
                al = (bl - r) / 3;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = EOU;
            }
            if (a_exgl && rl < r) bl = 3 * al + r;
        }
        ++i;
        if ((i < n_im && mi_of(i) < al) || CPOS(i, 2) < bl) maxh_val = HS_NEV;
        else if (CPOS(i, 8) == EOU || CPOS(i, 9) == EOU) flag = -3;
        else {
            r = bl - 3 * al;
            CPOS(i, 8) = min(r, CPOS(i, 8));
            CPOS(i, 9) = max(r, CPOS(i, 9));
        }
    }
#undef CPOS
    A.scores[pi] = maxh_val;
    A.ranges[4 * pi] = al; A.ranges[4 * pi + 1] = ar; A.ranges[4 * pi + 2] = bl; A.ranges[4 * pi + 3] = br;
    DevResultH R;
    R.score = maxh_val; R.mr = ar; R.nr = br; R.maxt = 0; R.maxr = 0; R.pad[0] = flag; R.pad[1] = R.pad[2] = 0;
    A.res[pi] = R;
}

extern "C" hipError_t spdh_launch_scalar_udh(const HScalarArgs* a, hipStream_t stream)
{
    HScalarArgs A = *a;
    const int per = A.n_probs <= 8192 ? 1 : 64;
    const dim3 grd((A.n_probs + per - 1) / per), blk(per);
    hipLaunchKernelGGL(spdh_scalar_udh, grd, blk, 0, stream, A);
    return hipGetLastError();
}
