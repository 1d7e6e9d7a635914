// spdp_exact.hip -- the reference's -A1 ("rigorous") cDNA engines on the GPU.
//
//   spdp_exact_score   SimdAln2s1::scoreonlyS1                           src/fwd2s1_simd.cc:288-478
//                      Sjsites::get / put (exact intron lists per lane)  src/fwd2s1_simd.cc:40-159
//                      from_spj / to_spj                                 src/fwd2s1_simd.cc:264-284
//                      fhinitS1 / fhlastS1                               src/fwd2s1_simd.cc:163-262
//
// Same 16-row stripes chained through the per-diagonal boundary arrays as the `_wip` engines (the
// results depend on that geometry), but the intron model of the scalar engines: every lane keeps the
// top-NCAND donor candidates of its row; an acceptor column tries them with the exact IntPen(len) and
// the sig53 pair score and raises H / E / F of that one cell; "post-splice" flags keep a spliced state
// from donating again before a match.  In the reference the lists hang off the vector loop as scalar
// calls per queued column (donor_q / accep_q); a queued column n_j is visited by lane j = n - n_j
// exactly once, at step n, so here every lane simply looks at its own column.
// Mapping: 16 lanes = one stripe of one problem, four problems per wave; lanes exchange H / F with
// row_shr-style shuffles inside their 16-lane group; stripes run one after the other.  First form of
// this engine: correct, not yet tiled like spdp_sweep.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

#define XN 16                                    // rows per stripe (SPDP_NELEM)
#define XNEV SPDP_NEV16
__device__ static const unsigned char x_psp_bit[3] = {4, 1, 8};

__device__ __forceinline__ int x_sadd(int a, int b) { return max(a + b, SPDP_FLOOR16); }
__device__ __forceinline__ int x_up(int v) { return __shfl_up(v, 1, XN); }      // lane k <- lane k - 1 of its group

__global__ void __launch_bounds__(64) spdp_exact_score(ScalarArgs A)
{
    const int k = threadIdx.x & 15;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 4);
    if (pi >= A.n_probs) return;                 // a whole 16-lane group leaves together
    const DevProblem P = A.probs[pi];
    const DevScoring* sc = A.sc;
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool local = sc->local;
    const bool LocalL = local && a_exgl && b_exgl, LocalR = local && a_exgr && b_exgr;
    const bool spj = sc->spj;
    const int ge = sc->gep, gn = sc->gep + sc->gop, gop = sc->gop;
    const int minl = A.minl, ipen = A.ipen;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int2* cols = A.cols + P.col_off;       // .x = (sig5 + ipen) | sig3 << 16, .y = b[n - 1]
    const uint8_t* aux = A.aux + 2 * P.col_off;  // {bit0 donor | bit1 acceptor, dinc5 << 4 | dinc3}
    int* hv = A.work + P.bnd_off - lw + 1;       // by diagonal, in place like the reference's hv / fv
    int* fv = hv + P.buf_size;
    const int n_ent = P.buf_size;

    // ---- fhinitS1
    {
        const int rl = b_left - a_left;
        const int rr = min(b_right - a_left, up);
        int rr_g = rr;
        if (!a_exgl && ge) rr_g = min(rr, (XNEV - gop) / ge + rl);
        for (int e = k; e < n_ent; e += XN) {
            const int r = e + lw - 1;
            int h = XNEV;
            if (b_exgl && r >= lw && r < rl) h = 0;
            if (a_exgl) { if (r >= rl && r <= rr) h = 0; }
            else {
                if (r == rl) h = 0;
                else if (r == rl + 1) h = gop + ge;
                else if (ge) { if (r > rl + 1 && r < rr_g) h = gop + ge + (r - rl - 1) * ge; }
                else if (r > rl + 1 && r < rr) h = gop;
            }
            hv[r] = h; fv[r] = XNEV;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    int maxh = XNEV;
    for (int ml = a_left; ml < a_right; ml += XN) {
        const int j9 = min(XN, a_right - ml);
        const int j8 = j9 - 1;
        int n = max(b_left, lw + ml);
        const int n_first = n;
        const int n9 = min(b_right, up + (ml + j9) + 1) + j9;
        int r = n - (ml + 1);
        // per-lane state: H of the last two steps, F, E, flags, the candidate list of my row
        int H1 = XNEV, H2 = XNEV, F1 = XNEV, E = XNEV, ps = 0;
        int c_val[5], c_jnc[5], c_dir[5], idx[5], ncand = -1;
#pragma unroll
        for (int i = 0; i < 5; ++i) { c_val[i] = XNEV; c_jnc[i] = c_dir[i] = 0; idx[i] = i; }
        const int m = ml + 1 + k;                             // my row
        const int* mrow = sc->mtx + ((k < j9) ? acod[ml + k] : 0) * 32;
        for ( ; n < n9; ++n, ++r) {
            const int r0 = r - 2 * j8;
            const int ke = min(j9, n - b_left);
            const int nj = n - k;                             // my column
            // boundary feeds of lane 0 (previous stripe's bottom row, by diagonal)
            int bH1 = 0, bF1 = 0, bH2 = 0;
            if (k == 0) {
                bH1 = __builtin_nontemporal_load(&hv[r + 1]);
                bF1 = __builtin_nontemporal_load(&fv[r + 1]);
                bH2 = __builtin_nontemporal_load(&hv[r]);
            }
            int upH1 = x_up(H1), upF1 = x_up(F1), upH2 = x_up(H2);
            if (k == 0) { upH1 = bH1; upF1 = bF1; upH2 = bH2; }
            // insertion, deletion, diagonal
            {
                const int open = x_sadd(H1, gn), ext = x_sadd(E, ge);
                E = ext > open ? ext : open;
            }
            int F;
            {
                const int open = x_sadd(upH1, gn), ext = x_sadd(upF1, ge);
                F = ext > open ? ext : open;
            }
            int pv = 0;
            const bool incell = nj <= b_right && nj > b_left && k < j9;     // kb <= k < ke
            int2 col = make_int2(0, 0);
            if (nj >= 0 && nj <= b_right + 1) col = cols[nj];
            if (incell) pv = mrow[col.y];
            int H = x_sadd(pv, upH2);
            int hb = 0;
            if (F > H) { H = F; hb = 2; }
            if (E > H) { H = E; hb = 1; }
            if (spj) ps &= hb;
            if (!local) { if (!(H > XNEV)) H = XNEV; }
            else if (LocalL) { if (0 > H) H = 0; }
            if (LocalR) {
                int mx = (k < j9) ? H : INT32_MIN;
                for (int off = 8; off; off >>= 1) mx = max(mx, __shfl_xor(mx, off, XN));
                maxh = max(maxh, mx);
            }
            // the exact intron lists: only columns that were queued (pushed at step n_j of THIS stripe, n_j <= b_right)
            const bool queued = spj && k < j9 && nj >= n_first && nj <= b_right;
            const int rj = nj - m;
            if (queued && rj >= lw && rj < up) {
                const unsigned fl = aux[2 * nj];
                if (fl & 2) {                                 // acceptor: Sjsites::get
                    const int s3 = col.x >> 16;
                    const int d3 = aux[2 * nj + 1] & 15;
                    for (int l = 0; l <= ncand; ++l) {
                        const int ci = idx[l];
                        const int d = c_dir[ci], don = c_jnc[ci];
                        if (nj - don < minl) continue;
                        int len = nj - don;
                        if (len >= A.intpen_len) len = A.intpen_len - 1;
                        const int x = c_val[ci] + A.intpen[len] + s3 + A.t53[16 * (aux[2 * don + 1] >> 4) + d3];
                        int cur = d == 0 ? H : (d == 1 ? E : F);
                        if (x <= cur) continue;
                        cur = (int) (short) x;
                        if (d == 0) H = cur; else if (d == 1) E = cur; else F = cur;
                        ps |= x_psp_bit[d];
                        if (d && cur > H) H = cur;
                    }
                }
                if (fl & 1) {                                 // donor: Sjsites::put
                    const int sigJ = (int) (short) (col.x & 0xffff) - ipen;
                    for (int kk = hb ? 1 : 0; kk < 3; ++kk) {
                        if (ps & x_psp_bit[kk]) continue;
                        const int from = kk == 0 ? H : (kk == 1 ? E : F);
                        if (kk && from <= H + gop) continue;
                        const int x = from + sigJ;
                        if (x <= XNEV) continue;
                        int l = ncand < 4 ? ++ncand : 4;
                        while (--l >= 0) {
                            if (x >= c_val[idx[l]]) { const int t = idx[l]; idx[l] = idx[l + 1]; idx[l + 1] = t; }
                            else break;
                        }
                        if (++l < 4) { const int ci = idx[l]; c_val[ci] = (int) (short) x; c_jnc[ci] = nj; c_dir[ci] = kk; }
                        else --ncand;
                    }
                }
            }
            // bottom row of the stripe -> boundary arrays
            if (k == j8 && j9 == ke && lw <= r0 && r0 <= up) { hv[r0] = H; fv[r0] = F; }
            H2 = H1; H1 = H; F1 = F;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    // ---- fhlastS1 unless a local right end was tracked
    if (k == 0) {
        DevResult R;
        R.score = XNEV; R.mr = a_right; R.nr = b_right; R.ml = a_left; R.ulk = 0; R.maxr = 0; R.pad[0] = R.pad[1] = 0;
        if (LocalR) R.score = maxh;
        else {
            const int rr = b_right - a_right;
            int maxr = rr;
            if (a_exgr) {
                const int r1 = max(lw, b_left - a_right);
                int best = r1;
                for (int i = r1 + 1; i < rr; ++i) if (__builtin_nontemporal_load(&hv[i]) > __builtin_nontemporal_load(&hv[best])) best = i;
                maxr = best;
            }
            if (b_exgr) {
                const int r2 = min(up - 1, b_right - a_left);
                int best = rr;
                for (int i = rr + 1; i < r2; ++i) if (__builtin_nontemporal_load(&hv[i]) > __builtin_nontemporal_load(&hv[best])) best = i;
                if (__builtin_nontemporal_load(&hv[best]) > __builtin_nontemporal_load(&hv[maxr])) maxr = best;
            }
            R.score = __builtin_nontemporal_load(&hv[maxr]);
            R.maxr = maxr;
        }
        A.res[pi] = R;
    }
}

extern "C" hipError_t spdp_launch_exact_score(const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    hipLaunchKernelGGL(spdp_exact_score, dim3((A.n_probs + 3) / 4), dim3(64), 0, stream, A);
    return hipGetLastError();
}
