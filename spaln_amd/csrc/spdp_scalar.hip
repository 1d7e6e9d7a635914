// spdp_scalar.hip -- the reference's scalar exact-intron-length engines on the GPU.
//
//   spdp_scalar_fwd    Aln2s1::forwardS_ng + initS_ng / lastS_ng     src/fwd2s1.cc:217-444, 142-215
//                      + Vmf::traceback + the record fix-up of trcbkalignS_ng   src/vmf.cc:125, src/fwd2s1.cc:1690-1707
//   spdp_scalar_score  Aln2s1::scorealoneS_ng + sinitS_ng / slastS_ng           src/fwd2s1.cc:1163-1336, 1112-1161
//   spdp_scalar_udh    Aln2s1::hirschbergS_ng + hinitS_ng / hlastS_ng           src/fwd2s1.cc:762-1104, 701-760
//
// These are the reference's -A0 engines (int32, row by row, top-NCAND donor list per
// row, exact intron-length penalty).  The -A2/-A3 dispatch needs them for
// sub-problems with fewer than 8 query rows (trcbkalignS_ng, src/fwd2s1.cc:1677),
// which is what they are used for here: a handful of tiny problems per batch, so
// the mapping is simply one thread per problem with its rows in global memory.
// They are correct for any size (tests run them on whole fixtures), not fast.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

#define SC_NCAND 4
#define SC_NOD 3
#define SC_NEV (INT32_MIN / 16 * 7)
__device__ static const unsigned char sc_psp_bit[5] = {4, 1, 8, 2, 16};

struct ScCand { int val, dir, jnc, ptr; };
struct ScRvp { int val, ptr; };

__device__ __forceinline__ int sc_intpen(const ScalarArgs& A, int len)
{
    if (len < 0) return -32768;
    if (len >= A.intpen_len) len = A.intpen_len - 1;
    return A.intpen[len];
}
__device__ __forceinline__ int sc_spjscr(const ScalarArgs& A, const uint8_t* aux, const int2* cols, int jnc, int n)
{
    const int s3 = cols[n].x >> 16;
    return sc_intpen(A, n - jnc) + s3 + A.t53[16 * (aux[2 * jnc + 1] >> 4) + (aux[2 * n + 1] & 15)];
}

template <bool FORWARD>
__global__ void spdp_scalar(ScalarArgs A)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= A.n_probs) return;
    const DevProblem P = A.probs[pi];
    const DevScoring* sc = A.sc;
    const int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, spj = sc->spj, llmt = sc->llmt, ipen = A.ipen;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int2* cols = A.cols + P.col_off;
    const uint8_t* aux = A.aux + 2 * P.col_off;            // per position {flags, dinc}
    int* work = A.work + P.bnd_off;                          // 2 * 2 * width ints (+ width bytes of dir)
    ScRvp* hh0 = reinterpret_cast<ScRvp*>(work) - lw + 1;
    ScRvp* hh1 = hh0 + width;
    unsigned char* hdir = reinterpret_cast<unsigned char*>(work + 4 * width) - lw + 1;
    int3* vrec = A.vmf + P.tb_off;                           // Vmf records of this problem
    const int vcap = (int) P.imd_off;                        // capacity (records)
    int vn = 0;
    bool vover = false;
    auto vadd = [&](int m, int n, int p) -> int {
        if (vn < vcap) vrec[vn] = make_int3(m, n, p); else vover = true;
        return vn++;
    };
    for (int i = 0; i < 2 * width; ++i) { hh0[lw - 1 + i].val = SC_NEV; hh0[lw - 1 + i].ptr = 0; }
    if (FORWARD) for (int i = 0; i < width; ++i) hdir[lw - 1 + i] = 0;
    // ---- initS_ng / sinitS_ng
    if (FORWARD) vadd(0, 0, 0);
    {
        int r = bl - al, rr = br - al;
        ScRvp* h = hh0 + r;
        h->val = 0;
        if (FORWARD) { hdir[r] = 0; h->ptr = vadd(al, bl, 0); }
        if (a_exgl) {
            if (up < rr) rr = up;
            int q = r;
            while (++q <= rr) { hh0[q].val = 0; hh0[q].ptr = 0; if (FORWARD) hdir[q] = 1; }
        }
        rr = bl - ar;
        if (lw > rr) rr = lw;
        ScRvp* f = hh1 + r;
        for (int i = 1; --r >= rr; ++i) {
            --h; --f;
            if (FORWARD) hdir[r] = 2;
            if (b_exgl) { h->val = 0; h->ptr = 0; }
            else {
                *h = h[1];
                if (i == 1) { h->val += gop + gep; if (!FORWARD) f->val = h->val; }
                else { h->val += gep; if (!FORWARD) f->val = f[1].val + gep; }
            }
        }
    }
    int maxh_val = SC_NEV, maxh_m = al, maxh_n = bl, maxh_p = 0;
    int m = al;
    if (!a_exgl) --m;
    int n1 = m + lw, n2 = m + up + 1;
    for ( ; ++m <= ar; ++n1, ++n2) {
        const bool internal = FORWARD ? (spj && (!a_exgr || m < ar)) : true;
        int n = max(n1, bl);
        const int n9 = min(n2, br);
        int r = n - m;
        ScRvp *h = hh0 + r, *f = hh1 + r;
        unsigned char* dir = hdir + r;
        unsigned psp = 0;
        ScRvp e1; e1.val = SC_NEV; e1.ptr = 0;
        ScCand rcd[SC_NCAND + 1];
        int idx[SC_NCAND + 1];
        for (int l = 0; l <= SC_NCAND; ++l) { rcd[l].val = SC_NEV; rcd[l].dir = rcd[l].jnc = rcd[l].ptr = 0; idx[l] = l; }
        int ncand = -1;
        const int* qprof = sc->mtx + ((m >= 1) ? acod[m - 1] : 0) * 32;
        for ( ; ++n <= n9; ) {
            int x;
            ++dir; ++h; ++f;
            // hf[0] = h, hf[1] = &e1, hf[2] = f
            int mxk = 0;                                     // which of hf[] is the running maximum
            const int diag = h->val;
            if (m != al) {
                h->val += qprof[cols[n].y];
                if (FORWARD) *dir = (*dir % 3) ? 3 : 0;      // Newd = 3
                x = h[1].val + gop;
                if (FORWARD ? (x >= f[1].val) : (x > f[1].val)) { f->val = x; f->ptr = h[1].ptr; }
                else *f = f[1];
                f->val += gep;
                if (f->val > h->val) mxk = 2;
            }
            x = h[-1].val + gop;
            if (FORWARD ? (x >= e1.val) : (x > e1.val)) { e1.val = x; e1.ptr = h[-1].ptr; psp = psp ? 1 : 0; }
            else psp &= 1;
            e1.val += gep;
            {
                const int cur = mxk == 0 ? h->val : f->val;
                if (FORWARD ? (e1.val >= cur) : (e1.val > cur)) mxk = 1;
            }
            auto hfv = [&](int k) -> ScRvp* { return k == 0 ? h : (k == 1 ? &e1 : f); };
            if (internal && (aux[2 * n] & 2)) {               // acceptor
                const ScCand* maxphl[SC_NOD] = {nullptr, nullptr, nullptr};
                for (int l = 0; l <= ncand; ++l) {
                    const ScCand* prd = rcd + idx[l];
                    if (n - prd->jnc < llmt) continue;
                    x = prd->val + sc_spjscr(A, aux, cols, prd->jnc, n);
                    ScRvp* from = hfv(prd->dir);
                    if (FORWARD ? (x >= from->val) : (x > from->val)) { from->val = x; maxphl[prd->dir] = prd; }
                }
                for (int k = 0; k < SC_NOD; ++k) {
                    const ScCand* prd = maxphl[k];
                    if (!prd) continue;
                    ScRvp* from = hfv(k);
                    psp |= sc_psp_bit[k];
                    if (FORWARD) from->ptr = vadd(m, n, vadd(m, prd->jnc, prd->ptr));
                    const int cur = hfv(mxk)->val;
                    if (FORWARD ? (from->val >= cur) : (from->val > cur)) mxk = k;
                }
            }
            int hd = 0;
            if (FORWARD) {
                if (mxk != 0) { *h = *hfv(mxk); hd = mxk; *dir = (unsigned char) hd; }
                else if (Local && h->val > diag) {
                    if (LocalL && diag == 0) h->ptr = vadd(m - 1, n - 1, 0);
                    else if (LocalR && h->val > maxh_val) { maxh_val = h->val; maxh_p = h->ptr; maxh_m = m; maxh_n = n; }
                }
                if (LocalL && h->val <= 0) { h->val = 0; *dir = 1; }
                else if (*dir == 3 && !(psp & sc_psp_bit[0])) h->ptr = vadd(m - 1, n - 1, h->ptr);
            } else {
                const int y = h->val;
                if (mxk != 0) h->val = hfv(mxk)->val;
                else if (LocalR && y > maxh_val) maxh_val = y;
                if (LocalL && h->val < 0) h->val = 0;
                hd = mxk;
            }
            const int mxval = FORWARD ? h->val : hfv(mxk)->val;  // *mx (forward: h was overwritten with *mx)
            if (internal && (aux[2 * n] & 1)) {               // donor
                const int sigJ = (int) (short) (cols[n].x & 0xffff) - ipen;   // sig5[n]
                for (int k = hd == 0 ? 0 : 1; k < SC_NOD; ++k) {
                    ScRvp* from = hfv(k);
                    if (psp & sc_psp_bit[k]) continue;
                    if (k != hd) {
                        int z = FORWARD ? hfv(mxk)->val : mxval;
                        if (hd == 0 || (k - hd) % 2) z += (k / 2 == 1) ? gop : 0;
                        if (from->val <= z) continue;
                    }
                    x = from->val + sigJ;
                    int l = ncand < SC_NCAND ? ++ncand : SC_NCAND;
                    while (--l >= 0) {
                        if (FORWARD ? (x > rcd[idx[l]].val) : (x >= rcd[idx[l]].val)) { const int t = idx[l]; idx[l] = idx[l + 1]; idx[l + 1] = t; }
                        else break;
                    }
                    if (++l < SC_NCAND) {
                        ScCand* prd = rcd + idx[l];
                        prd->val = x; prd->jnc = n; prd->dir = k; prd->ptr = from->ptr;
                    } else --ncand;
                }
            }
        }
    }
    DevResult R;
    R.score = SC_NEV; R.mr = ar; R.nr = br; R.ml = al; R.ulk = 0; R.maxr = 0; R.pad[0] = R.pad[1] = 0;
    int ptr = 0;
    if (!FORWARD) {
        if (LocalR) R.score = maxh_val;
        else {  // slastS_ng
            const int r9 = br - ar;
            int mx = hh0[r9].val;
            if (b_exgr) { const int rw = min(up, br - al); for (int r = rw; r > r9; --r) if (hh0[r].val > mx) mx = hh0[r].val; }
            if (a_exgr) { const int rw = max(lw, bl - ar); for (int r = rw; r < r9; ++r) if (hh0[r].val > mx) mx = hh0[r].val; }
            R.score = mx;
        }
        A.res[pi] = R;
        return;
    }
    if (LocalR) { ptr = vadd(maxh_m, maxh_n, maxh_p); R.score = maxh_val; }
    else {      // lastS_ng
        int rw = lw;
        const int rf = bl - ar;
        if (rf > rw) rw = rf;
        const int r9 = br - ar;
        int mx = r9;
        if (a_exgr) for (int r = rw; r <= r9; ++r) if (hh0[r].val > hh0[mx].val) mx = r;
        if (b_exgr) { rw = min(up, br - al); for (int r = rw; r > r9; --r) if (hh0[r].val > hh0[mx].val) mx = r; }
        const int i = mx - r9;
        int m9 = ar, n9 = br;
        if (i > 0) m9 -= i;
        if (i < 0) n9 += i;
        hh0[mx].ptr = vadd(m9, n9, hh0[mx].ptr);
        R.score = hh0[mx].val; ptr = hh0[mx].ptr;
    }
    // Vmf::traceback(ptr) + fix-up of trcbkalignS_ng
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
        const int rd = Local ? 0 : ((ln - lm) - bl + al);
        if (rd) {
            const int2 rec = rd > 0 ? make_int2(al, bl + rd) : make_int2(al - rd, bl);
            if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
            ++cnt;
        }
    }
    A.n_skl[pi] = status ? status : cnt;
    A.res[pi] = R;
}

extern "C" hipError_t spdp_launch_scalar(int forward, const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    // few problems: one per wave (a wave of divergent one-thread problems runs them one after another)
    const int per = A.n_probs <= 8192 ? 1 : 64;
    const dim3 grd((A.n_probs + per - 1) / per), blk(per);
    if (forward) hipLaunchKernelGGL(spdp_scalar<true>, grd, blk, 0, stream, A);
    else hipLaunchKernelGGL(spdp_scalar<false>, grd, blk, 0, stream, A);
    return hipGetLastError();
}

// ---- scalar unidirectional Hirschberg ---------------------------------------------------------
//   spdp_scalar_udh   Aln2s1::hirschbergS_ng + hinitS_ng / hlastS_ng   src/fwd2s1.cc:762-1104, 701-760
//                     with UdhIntermediate(lub = true)                   src/udh_intermediate.h:29-66
// The -A0 linear-space engine: forwardS_ng's recurrence, every state carrying the diagonal range it
// has visited since the last intermediate row (upr / lwr), its start row (ml) and a link (ulk) to
// where its path crossed the previous intermediate row; the tail walks the links back into the
// cpos rows lspS_ng reads ([8] / [9] = diagonal bounds of the slab below a row, which
// mimd_postwork / rcsv_postwork use as the slab's window under -A0).  One thread per problem.
// res.pad[0] = -3 flags inputs on which the reference itself reads or writes outside its arrays.
struct ScRvwml { int val, upr, lwr, ml, ulk; };
struct ScRvdwmlj { int val, dir, upr, lwr, ml, ulk, jnc; };

__global__ void spdp_scalar_udh(ScalarArgs A)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= A.n_probs) return;
    const DevProblem P = A.probs[pi];
    const DevScoring* sc = A.sc;
    const int EOU = 0x7fffffff - 2;                          // end_of_ulk, src/aln.h:49
    int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, n_im = P.n_im;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, llmt = sc->llmt, ipen = A.ipen;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int2* cols = A.cols + P.col_off;
    const uint8_t* aux = A.aux + 2 * P.col_off;
    ScRvwml* const wbuf = reinterpret_cast<ScRvwml*>(A.work + P.bnd_off);    // 2 * width + 4 states
    ScRvwml* const hh0 = wbuf - lw + 1;
    ScRvwml* const hh1 = hh0 + width;
    // intermediate i: hlnk[2], vlnk[2], lwrb[2], uprb[2], each `width` ints, index r - lw + 1
    int* const imd_base = A.imd + P.imd_off;
    const int64_t us = 2 * (int64_t) width;
    auto IM = [&](int i, int arr, int k, int r) -> int& { return imd_base[(int64_t) i * 4 * us + arr * us + (int64_t) k * width + (r - lw + 1)]; };
    enum { HLNK = 0, VLNK = 1, LWRB = 2, UPRB = 3 };
    int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    for (int i = 0; i <= n_im; ++i) for (int c = 0; c < 10; ++c) CPOS(i, c) = EOU;
    int r = bl - ar;
    const ScRvwml black = {SC_NEV, r, r, 0, EOU};
    for (int i = 0; i < 2 * width + 4; ++i) wbuf[i] = black;
    for (int i = 0; i < n_im; ++i)
        for (int64_t k = 0; k < us; ++k) {
            imd_base[(int64_t) i * 4 * us + k] = EOU;
            imd_base[(int64_t) i * 4 * us + us + k] = EOU;
            imd_base[(int64_t) i * 4 * us + 2 * us + k] = 0x7fffffff;
            imd_base[(int64_t) i * 4 * us + 3 * us + k] = (int) 0x80000000;
        }
    auto mi_of = [&](int i) { return P.a_left + (i + 1) * P.imd_intvl; };
    // hinitS_ng
    {
        int rr = br - al;
        const int r0 = bl - al;
        r = r0;
        ScRvwml* h = hh0 + r;
        h->val = 0; h->lwr = h->upr = h->ulk = r; h->ml = al;
        if (a_exgl) {
            if (up < rr) rr = up;
            while (++r <= rr) { ++h; h->val = 0; h->lwr = h->upr = h->ulk = r; h->ml = al; }
        }
        r = r0;
        rr = bl - ar;
        if (lw > rr) rr = lw;
        h = hh0 + r - 1;
        for (int i = 1; --r >= rr; ++i, --h) {
            if (b_exgl) { h->val = 0; h->lwr = h->upr = h->ulk = r; h->ml = h[1].ml + 1; }
            else {
                *h = h[1];
                ++h->ml;
                h->val += (i == 1) ? gop + gep : gep;
                h->lwr = r;
                h->ulk = r0;
            }
        }
    }
    int ii = 0;                                              // current intermediate
    int mm = mi_of(0);
    int rlst = 0x7fffffff;
    int maxh_val = SC_NEV, maxh_upr = 0, maxh_lwr = 0, maxh_ml = al, maxh_ulk = 0, maxh_mr = ar, maxh_nr = br;
    int m = al;
    if (!a_exgl) --m;
    int n1 = m + lw, n2 = m + up + 1;
    for ( ; ++m <= ar; ++n1, ++n2) {
        int n = max(n1, bl);
        const int n9 = min(n2, br);
        const bool is_imd = m == mm;
        unsigned psp = 0;
        r = n - m;
        ScRvwml *h = hh0 + r, *f = hh1 + r;
        ScRvwml e1 = black;
        ScRvdwmlj rcd[SC_NCAND + 1];
        int idx[SC_NCAND + 1];
        for (int l = 0; l <= SC_NCAND; ++l) {
            rcd[l].val = SC_NEV; rcd[l].dir = 0; rcd[l].upr = (int) 0x80000000; rcd[l].lwr = 0x7fffffff;
            rcd[l].ml = 0; rcd[l].ulk = EOU; rcd[l].jnc = 0;
            idx[l] = l;
        }
        int ncand = -1;
        const int* qprof = sc->mtx + ((m >= 1) ? acod[m - 1] : 0) * 32;
        for ( ; ++n <= n9; ) {
            int x;
            ++r; ++h; ++f;
            ScRvwml* hf[SC_NOD] = {h, &e1, f};
            ScRvwml* mx = h;
            if (m != al) {
                h->val += qprof[cols[n].y];
                x = h[1].val + gop;
                if (x >= f[1].val) { *f = h[1]; f->val = x; }
                else *f = f[1];
                f->val += gep;
                if (f->val >= mx->val) mx = f;
            }
            x = h[-1].val + gop;
            if (x >= e1.val) { e1 = h[-1]; e1.val = x; psp = psp ? 1 : 0; }
            else psp &= 3;
            e1.val += gep;
            if (e1.val >= mx->val) mx = &e1;
            bool spj3 = false;
            if (aux[2 * n] & 2) {                            // acceptor
                const ScRvdwmlj* maxphl[SC_NOD] = {nullptr, nullptr, nullptr};
                for (int l = 0; l <= ncand; ++l) {
                    const ScRvdwmlj* prd = rcd + idx[l];
                    if (n - prd->jnc < llmt) continue;
                    ScRvwml* from = hf[prd->dir];
                    x = prd->val + sc_spjscr(A, aux, cols, prd->jnc, n);
                    if (x > from->val) { from->val = x; maxphl[prd->dir] = prd; }
                }
                int maxk = SC_NOD;
                for (int k = 0; k < SC_NOD; ++k) {
                    const ScRvdwmlj* prd = maxphl[k];
                    if (!prd) continue;
                    psp |= sc_psp_bit[k];
                    if (!k) spj3 = true;
                    ScRvwml* from = hf[k];
                    from->upr = max(prd->upr, r);
                    from->lwr = min(prd->lwr, r);
                    from->ml = prd->ml;
                    from->ulk = prd->ulk;
                    if (from->val > mx->val) { maxk = k; mx = from; }
                }
                if (is_imd && maxk < SC_NOD) {
                    const ScRvdwmlj* phl = maxphl[maxk];
                    IM(ii, HLNK, 0, r) = phl->ulk;
                    mx->ulk = rlst = r;
                    if (maxk == 0) {
                        if ((phl = maxphl[1]) && hf[1]->val > mx->val + gop) {
                            hf[1]->ulk = r + width;
                            IM(ii, HLNK, 1, r) = phl->ulk;
                        }
                        if (maxphl[2] && hf[2]->val > mx->val + gop) hf[2]->ulk = r + width;
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
            if (aux[2 * n] & 1) {                            // donor
                const int sigJ = (int) (short) (cols[n].x & 0xffff) - ipen;
                for (int k = (hd == 0) ? 0 : 1; k < SC_NOD; ++k) {
                    ScRvwml* from = hf[k];
                    if (psp & sc_psp_bit[k]) continue;
                    if (k != hd) {
                        int y = mx->val;
                        if (hd == 0 || (k - hd) % 2) y += (k / 2 == 1) ? gop : 0;
                        if (from->val <= y) continue;
                    }
                    x = from->val + sigJ;
                    int l = ncand < SC_NCAND ? ++ncand : SC_NCAND;
                    while (--l >= 0) {
                        if (x > rcd[idx[l]].val) { const int t = idx[l]; idx[l] = idx[l + 1]; idx[l + 1] = t; }
                        else break;
                    }
                    if (++l < SC_NCAND) {
                        ScRvdwmlj* prd = rcd + idx[l];
                        prd->val = x; prd->jnc = n; prd->dir = k;
                        prd->upr = from->upr; prd->lwr = from->lwr; prd->ml = from->ml;
                        if (is_imd) {
                            if (k == 1) IM(ii, HLNK, 0, r) = rlst;
                            prd->ulk = r;
                        } else prd->ulk = from->ulk;
                    } else --ncand;
                }
            }
            if (is_imd) {
                if (hd == 0) rlst = r;
                else if (!spj3 && hd % 2) IM(ii, HLNK, 0, r) = rlst;
                for (int k = 0; k < 2; ++k) {
                    ScRvwml* g = hf[2 * k];
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
    const int rr = br - ar;
    if (LocalR) {
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        ar = maxh_mr; br = maxh_nr;
        if (i < 0) i = 0;
        CPOS(i, 8) = maxh_lwr;
        CPOS(i, 9) = maxh_upr;
    } else {    // hlastS_ng
        const int r9 = br - ar;
        int mxr = r9;
        if (b_exgr) { const int rw = min(up, br - al); for (int q = rw; q > r9; --q) if (hh0[q].val > hh0[mxr].val) mxr = q; }
        if (a_exgr) { const int rw = max(lw, bl - ar); for (int q = rw; q < r9; ++q) if (hh0[q].val > hh0[mxr].val) mxr = q; }
        const ScRvwml mxs = hh0[mxr];
        maxh_val = mxs.val; maxh_lwr = mxs.lwr; maxh_upr = mxs.upr; maxh_ulk = mxs.ulk; maxh_ml = mxs.ml;
        r = mxr;
        if (b_exgr && rr < r) ar = br - r;
        if (a_exgr && rr > r) br = ar + r;
    }
    int i = n_im;
    while (--i >= 0 && mi_of(i) > ar) ;
    if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
    r = br - ar;
    CPOS(i + 1, 8) = min(maxh_lwr, r);
    CPOS(i + 1, 9) = max(maxh_upr, r);
    r = maxh_ulk;
    for ( ; i >= 0 && mi_of(i) > maxh_ml; --i) {
        int c = 0, d = 0;
        for ( ; r > up; r -= width) ++d;
        if (d > 1 || r < lw - 1) { flag = -3; break; }       // outside the link arrays
        const int mi = mi_of(i);
        if (IM(i, VLNK, d, r) < EOU) {
            CPOS(i, c++) = mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            for (int rp = IM(i, HLNK, d, r); lw <= rp && rp < up && r != rp; rp = IM(i, HLNK, 0, r = rp)) {
                if (c >= 6) { flag = -3; break; }            // the terminator would land on [8]
                CPOS(i, c++) = r + mi;
            }
            if (flag) break;
            CPOS(i, c++) = r + mi;
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
        if (LocalL) { al = maxh_ml; bl = r + maxh_ml; }
        else {
            const int rl = bl - al;
            if (b_exgl && rl > r) {
                al = bl - r;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = EOU;
            }
            if (a_exgl && rl < r) bl = al + r;
        }
        ++i;
        if (i >= n_im) flag = -3;                            // the reference dereferences udhimds[n_im]
        else if (mi_of(i) < al || CPOS(i, 2) < bl) maxh_val = SC_NEV;
        else if (CPOS(i, 8) == EOU || CPOS(i, 9) == EOU) flag = -3;     // bounds the reference never set
        else {
            const int rl = bl - al;
            CPOS(i, 8) = min(rl, CPOS(i, 8));
            CPOS(i, 9) = max(rl, CPOS(i, 9));
        }
    }
#undef CPOS
    A.scores[pi] = maxh_val;
    A.ranges[4 * pi] = al; A.ranges[4 * pi + 1] = ar; A.ranges[4 * pi + 2] = bl; A.ranges[4 * pi + 3] = br;
    DevResult R;
    R.score = maxh_val; R.mr = ar; R.nr = br; R.ml = al; R.ulk = 0; R.maxr = 0; R.pad[0] = flag; R.pad[1] = 0;
    A.res[pi] = R;
}

extern "C" hipError_t spdp_launch_scalar_udh(const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    // few problems: one per wave (a wave of divergent one-thread problems runs them one after another)
    const int per = A.n_probs <= 8192 ? 1 : 64;
    const dim3 grd((A.n_probs + per - 1) / per), blk(per);
    hipLaunchKernelGGL(spdp_scalar_udh, grd, blk, 0, stream, A);
    return hipGetLastError();
}
