// spdp_local_udh.hip -- hirschbergS1_wip with local ends (-LS), src/fwd2s1_wip_simd.h:497-812.
//
// The production linear-space sweep (spdp_sweep<FL_UDH>) covers the non-local form.  With local ends the
// reference also carries the left-end row (`ml`) on H / E / F / the row's donor, restarts paths at zero, tracks
// the best cell as the right end -- and depends on lane state it never re-initialises: the link planes
// (hc_a, fc_a) survive from one stripe to the next (`vec_clear(hb_a[0], ..)`, :524, stops short of them), and on
// stripes that hold an intermediate row the substitution-score lanes are overwritten with the direction codes
// (`Store(pv_a, pb_v)`, :676), which out-of-range lanes then add as if they were scores.  With H clamped at zero
// those lanes are reachable, so the results depend on it (fixture s1_local_cut).  This kernel keeps the
// reference's lane state literally: 16 lanes = one stripe of one problem, four problems per wave, stripes one
// after the other, boundary rows by diagonal in memory.  Local linear-space runs are rare (a long query with
// -LS); this is the exactness path for them, not a throughput path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

#define LN 16
#define LNEV SPDP_NEV16
#define L_EOU (0x7fffffff - 2)

__device__ __forceinline__ int l_sadd(int a, int b) { return max(a + b, SPDP_FLOOR16); }
__device__ __forceinline__ int l_up(int v) { return __shfl_up(v, 1, LN); }

__global__ void __launch_bounds__(64) spdp_local_udh(ScalarArgs A)
{
    const int k = threadIdx.x & 15;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 4);
    if (pi >= A.n_probs) return;
    const DevProblem P = A.probs[pi];
    const DevScoring* sc = A.sc;
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, B = P.buf_size;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool local = sc->local;
    const bool LocalL = local && a_exgl && b_exgl, LocalR = local && a_exgr && b_exgr;
    const bool spj = sc->spj;
    const int ge = sc->gep, gn = sc->gep + sc->gop, gop = sc->gop;
    const int llmt = sc->llmt, nquant = sc->nquant;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int2* cols = A.cols + P.col_off;       // .x = (sig5 + ipen) | sig3 << 16, .y = b[n - 1]
    int* hv = A.work + P.bnd_off - lw + 1;
    int* fv = hv + B;
    int* hb = fv + B;
    int* fb = hb + B;
    int* hc = fb + B;
    int* fc = hc + B;
    int* imd0 = A.imd + P.imd_off;
    auto LNK = [&](int i, int which, int d, int r) -> int& { return imd0[((int64_t) i * 4 + which * 2 + d) * width + (r - lw + 1)]; };
    const int n_im = P.n_im;
    const int imd_step = (a_right - a_left + n_im) / (n_im + 1);
    auto qpen = [&](int hil) -> int {
        int pv = sc->qm_pen[0];
        for (int j = 1; j < nquant; ++j) if (hil > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
        return pv;
    };

    // ---- fhinitS1 with links (:68-140 of fwd2s1_simd.cc's form)
    {
        const int rl = b_left - a_left;
        const int ru = up + 2 * LN;
        const int rr = min(b_right - a_left, up);
        int rr_g = rr;
        if (!a_exgl && ge) rr_g = min(rr, (LNEV - gop) / ge + rl);
        for (int e = k; e < B; e += LN) {
            const int r = e + lw - 1;
            int h = LNEV;
            if (b_exgl && r >= lw && r < rl) h = 0;
            if (a_exgl) { if (r >= rl && r <= rr) h = 0; }
            else {
                if (r == rl) h = 0;
                else if (r == rl + 1) h = gop + ge;
                else if (ge) { if (r > rl + 1 && r < rr_g) h = gop + ge + (r - rl - 1) * ge; }
                else if (r > rl + 1 && r < rr) h = gop;
            }
            hv[r] = h; fv[r] = LNEV;
            int c = 0;
            if (r >= rl) { if (a_exgl) c = (r < ru) ? r : 0; else c = (r <= ru) ? rl : 0; }
            else c = b_exgl ? r : rl;
            if (r == rl) c = rl;
            hc[r] = c; fc[r] = c;
            int bm = a_left;
            if (b_exgl && r <= rl && r >= lw) bm = a_left + (rl - r);
            hb[r] = bm; fb[r] = a_left;
        }
        for (int e = k; e < n_im * 4 * width; e += LN) imd0[e] = L_EOU;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    int max_val = LNEV, max_ulk = L_EOU, max_ml = a_left, max_mr = a_right, max_nr = b_right;
    int imd_i = 0, rlst = 0x7fffffff;
    // link lanes that survive from stripe to stripe: by the parity of the step that stored them
    int C0 = 0, C1 = 0, FC = 0;                  // hc_a[0], hc_a[1], fc_a (entry k + 1)
    for (int ml = a_left; ml < a_right; ml += LN) {
        const int j9 = min(LN, a_right - ml);
        const int j8 = j9 - 1;
        int n = max(b_left, lw + ml);
        const int n9 = min(b_right, up + (ml + j9) + 1) + j9;
        int r = n - (ml + 1);
        int donor_r = r;
        int mm_ = 0, k9 = 0, k8 = -1;
        bool is_imd_ = false;
        if (imd_i < n_im) {
            const int mi = a_left + (imd_i + 1) * imd_step;
            mm_ = a_left + (mi - a_left - 1) / LN * LN;
            k9 = mi - mm_; k8 = k9 - 1;
            is_imd_ = ml == mm_;
        }
        (void) k9;
        int H0 = LNEV, H1 = LNEV, F = LNEV;      // hv_a[0], hv_a[1], fv_a: what this lane stored (entry k + 1)
        int B0 = 0, B1 = 0, FB = 0;
        int E = LNEV, EB = 0, EC = 0, D = LNEV, DB = 0, DC = 0, hil = 0;     // E lane, the row's best donor (hv2)
        int s5 = 0, s3 = 0, pv = 0, is_acc = 0, is_don = 0;
        const int* mrow = sc->mtx + ((k < j9) ? acod[ml + k] : 0) * 32;
        for (int pp = 0; n < n9; ++n, ++r, pp ^= 1) {
            const int r0 = r - 2 * j8;
            const int rj = r - 2 * k8;
            const int kb = max(0, n - b_right);
            const int ke = min(j9, n - b_left);
            const bool is_imd = is_imd_ && rj >= lw && rj <= up;
            // Hq = plane of the previous step, Hp = plane of the step before (overwritten now)
            const int Hq = pp ? H0 : H1, Hp = pp ? H1 : H0;
            const int Bq = pp ? B0 : B1, Bp = pp ? B1 : B0;
            const int Cq = pp ? C0 : C1, Cp = pp ? C1 : C0;
            int uHq = l_up(Hq), uF = l_up(F), uHp = l_up(Hp);
            int uBq = l_up(Bq), uFB = l_up(FB), uBp = l_up(Bp);
            int uCq = l_up(Cq), uFC = l_up(FC), uCp = l_up(Cp);
            if (k == 0) {
                uHq = hv[r + 1]; uF = fv[r + 1]; uHp = hv[r];
                // the reference feeds the `ml` lanes from the boundary rows only with local left ends; otherwise
                // entry 0 keeps the zero of the per-stripe clear
                if (LocalL) { uBq = hb[r + 1]; uFB = fb[r + 1]; uBp = hb[r]; } else { uBq = uFB = uBp = 0; }
                uCq = hc[r + 1]; uFC = fc[r + 1]; uCp = hc[r];
            }
            if (kb) pv = 0;
            const int nj = n - k;
            int2 col = make_int2(0, 0);
            if (nj >= 0 && nj <= b_right + 1) col = cols[nj];
            if (k >= kb && k < ke) pv = mrow[col.y];
            if (spj) {
                const int u3 = l_up(s3), u5 = l_up(s5);
                if (k == 0) {
                    const int2 c0 = cols[min(n, b_right + 1)];
                    s3 = kb ? 0 : (c0.x >> 16);
                    s5 = kb ? 0 : (int) (short) (c0.x & 0xffff);
                } else { s3 = u3; s5 = u5; }
            }
            // horizontal
            {
                const int opn = l_sadd(Hq, gn), ext = l_sadd(E, ge);
                if (ext > opn) E = ext;
                else { E = opn; EB = Bq; EC = Cq; }
            }
            // vertical
            int f, fbk, fck;
            {
                const int fext = l_sadd(uF, ge), fopn = l_sadd(uHq, gn);
                if (fext > fopn) { f = fext; fbk = uFB; fck = uFC; }
                else { f = fopn; fbk = uBq; fck = uCq; }
            }
            // diagonal, best of three, acceptor
            int h = l_sadd(pv, uHp), hbk = uBp, hck = uCp, pb3 = 0;
            if (f > h) { h = f; pb3 = 2; hbk = fbk; hck = fck; }
            if (E > h) { h = E; pb3 = 1; hbk = EB; hck = EC; }
            is_acc = 0;
            if (spj) {
                int x = l_sadd(l_sadd(D, s3), qpen(hil));
                if (!(hil > llmt)) x = LNEV;
                if (x > h) { h = x; is_acc = 1; hbk = DB; hck = DC; }
            }
            if (LocalL && 0 > h) h = 0;
            is_don = 0;
            if (spj) {
                const int qd = l_sadd(h, s5);
                if (qd > D) { D = qd; is_don = 1; DB = hbk; DC = hck; hil = 0; }
                hil = min(hil + 1, 32767);
            }
            // the vector stores
            int nH = h, nB = LocalL ? hbk : Bp, nC = hck;
            F = f; FC = fck;
            if (LocalL) FB = fbk;
            if (is_imd) pv = pb3;                             // Store(pv_a, pb_v)
            // scalar bookkeeping after the stores
            if (is_imd && spj && k == k8 && is_acc) {
                LNK(imd_i, 0, 0, rj) = donor_r;
                LNK(imd_i, 0, 1, rj) = donor_r + width;
                rlst = rj;
            }
            if (LocalL && k >= kb && k < ke && nH == 0) { nB = ml + k + 1; nC = r - 2 * k; }
            if (LocalR) {
                int mx = (k < j9) ? nH : INT32_MIN, mk = k;
                for (int off = 8; off; off >>= 1) {
                    const int ov = __shfl_xor(mx, off, LN), ok = __shfl_xor(mk, off, LN);
                    if (ov > mx || (ov == mx && ok < mk)) { mx = ov; mk = ok; }
                }
                if (mx > max_val) {
                    max_val = mx;
                    max_ml = __shfl(nB, mk, LN); max_ulk = __shfl(nC, mk, LN);
                    max_mr = ml + mk + 1; max_nr = n - mk;
                }
            }
            if (is_imd && k == k8) {
                if (spj && is_don) donor_r = rj;
                if (pb3 == 0) rlst = rj;
                if (pb3 == 1) LNK(imd_i, 0, 0, rj) = rlst;
                LNK(imd_i, 1, 0, rj) = nC; nC = rj;
                LNK(imd_i, 1, 1, rj) = FC; FC = rj + width;
            }
            if (pp) { H1 = nH; B1 = nB; C1 = nC; } else { H0 = nH; B0 = nB; C0 = nC; }
            if (k == j8 && j9 == ke && lw <= r0 && r0 <= up) {
                hv[r0] = nH; fv[r0] = F;
                if (LocalL) { hb[r0] = nB; fb[r0] = FB; }
                hc[r0] = nC; fc[r0] = FC;
            }
        }
        if (is_imd_) { rlst = __shfl(rlst, k8, LN); ++imd_i; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (k) return;

    // ---- the end cell: the tracked maximum, or fhlastS1
    DevResult R;
    R.score = max_val; R.mr = max_mr; R.nr = max_nr; R.ml = max_ml; R.ulk = max_ulk; R.maxr = 0; R.pad[0] = R.pad[1] = 0;
    if (!LocalR) {
        const int rr = b_right - a_right;
        int maxr = rr;
        if (a_exgr) {
            const int r1 = max(lw, b_left - a_right);
            int best = r1;
            for (int i = r1 + 1; i < rr; ++i) if (hv[i] > hv[best]) best = i;
            maxr = best;
        }
        if (b_exgr) {
            const int r2 = min(up - 1, b_right - a_left);
            int best = rr;
            for (int i = rr + 1; i < r2; ++i) if (hv[i] > hv[best]) best = i;
            if (hv[best] > hv[maxr]) maxr = best;
        }
        R.score = hv[maxr];
        R.maxr = maxr;
        R.mr = a_right; R.nr = b_right;
        if (maxr > rr) R.mr = b_right - maxr; else R.nr = a_right + maxr;
        R.ulk = hc[maxr];
        R.ml = LocalL ? hb[maxr] : a_left;
    }
    A.res[pi] = R;
}

extern "C" hipError_t spdp_launch_local_udh(const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    hipLaunchKernelGGL(spdp_local_udh, dim3((A.n_probs + 3) / 4), dim3(64), 0, stream, A);
    return hipGetLastError();
}
