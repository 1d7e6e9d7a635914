// spdp_h_exact.hip -- the reference's -A1 ("full-precision intron-length distribution") protein engines.
//
//   spdh_exact<false>   SimdAln2h1::forwardH1 (modes 3 / 5)             src/fwd2h1_simd.h:820-1096
//   spdh_exact<true>    SimdAln2h1::hirschbergH1 (modes 2 / 4)          src/fwd2h1_simd.h:1100-1470
//                       fhinitH1 / fhlastH1                             src/fwd2h1_simd.h:546-791
//                       Sjsites::get / put, from_spj / to_spj           src/fwd2h1_simd.h:388-543, 793-815
//                       Vmf::traceback + the fix-up of trcbkalignH_ng   src/vmf.cc:125, src/fwd2h1.cc:2019-2036
//
// 16 int16 lanes per stripe chained through per-diagonal boundary rows as in the `_wip` engines (the results
// depend on that geometry), six codon-phase planes of H / F (step mod 6) and three of E and the side lanes
// (step mod 3); the intron model is the scalar engines': every lane keeps the top-4 donor candidates of its
// row (value, junction, state, phase), an acceptor column re-scores them with the exact IntPen(len), the
// pair signal and the codon the junction spells, and raises H / E / F of the cells one to three steps back.
// The reference hangs the lists off the vector loop as scalar calls per queued column (donor_q / accep_q,
// one queue per frame); a queued column is met by lane j exactly once, at step n_j + 3 j, so here every lane
// looks at its own column.  A Vmf pointer (forward) or the link to the previous intermediate row (linear
// space) rides on H / E / F; forward appends its records through a per-problem atomic counter (record
// numbers differ from the reference's, the chains do not).
// Mapping: 16 lanes = one stripe of one problem, four problems per wave; a lane's planes live in its own
// LDS column (indexed by the step's phase), lanes exchange rows with 16-wide shuffles, stripes run one
// after the other.  This is the exactness engine, not a throughput path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

#define XN 16
#define XNEV (-32768 + 1024)
#define X_EOU (0x7fffffff - 2)                   // end_of_ulk

// LDS slots of one lane
enum { S_HV = 0, S_FV = 6, S_HB = 12, S_FB = 18, S_HC = 24, S_FC = 30,
       S_EV = 36, S_QV = 39, S_PS = 42, S_PV = 45, S_EB = 48, S_QB = 51, S_EC = 54, S_QC = 57, S_CP = 60,
       S_CVAL = 63, S_CULK = 68, S_CJNC = 73, S_CML = 78, S_CDIR = 83, S_CPHS = 88, S_CIDX = 93, S_NC = 98, S_END = 99 };

__device__ __forceinline__ int xh_add(int a, int b) { return min(max(a + b, -32768), 32767); }
__device__ __forceinline__ int xh_w16(int x) { return (int) (short) x; }
__device__ __forceinline__ int xh_up(int v) { return __shfl_up(v, 1, XN); }
__device__ __forceinline__ int xh_mod6(int x) { x %= 6; return x < 0 ? x + 6 : x; }

// G = 16-lane groups (problems) per wave: a wave issues the same instructions for one group as for four, so a launch
// with fewer problems than the chip has wave slots runs one problem per wave (four times the waves in flight)
template <bool UDH, int G>
__global__ void __launch_bounds__(16 * G) spdh_exact(HScalarArgs A)
{
    __shared__ int L[S_END][16 * G];
    __shared__ int Lmtx[32 * 32];                // the substitution matrix (aa x tron, row stride 32)
    __shared__ int Lvcnt[G];                     // Vmf record counters: only this group appends to its problem's list
    const int t = threadIdx.x;
    const int k = t & 15;
    const int pi = blockIdx.x * G + (t >> 4);
    const DevScoringH* sc = A.sc;
    for (int e = t; e < 32 * 32; e += 16 * G) Lmtx[e] = sc->mtx[e];
    __syncthreads();
    if (pi >= A.n_probs) return;                 // a whole 16-lane group leaves together
    const DevProblemH P = A.probs[pi];
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, B = P.buf_size;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const bool local = sc->local;
    const bool LocalL = local && a_exgl && b_exgl, LocalR = local && a_exgr && b_exgr;
    const bool spj = sc->spj;
    const int gop = sc->gop, gep = sc->gep, lgep = sc->lgep, codonk1 = sc->codonk1;
    const int g1 = sc->g1, g2 = sc->g2, g3 = sc->g3;
    const int minl = A.minl;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int4* cols = A.cols + P.col_off;       // .x: cp | tron of n - 2 << 16 | flags << 24, .y: sig3 candidates, .w: dinc
    const short4* aux = A.aux + P.col_off;       // {sigS, sigT, sigE, sig5} raw
    int* hv = A.work + P.bnd_off - lw + 3;       // by diagonal, in place like the reference's hv / fv ...
    int* fv = hv + B;
    int* hb = fv + B;
    int* fb = hb + B;
    int* hc = fb + B;
    int* fc = hc + B;
    int* vcount = &Lvcnt[t >> 4];
    int3* vrec = A.vmf + (UDH ? 0 : P.tb_off);
    const int vcap = UDH ? 0 : (int) P.imd_off;
    auto vadd = [&](int mm, int nn, int pp) -> int {
        const int i = atomicAdd(vcount, 1);
        if (i < vcap) vrec[i] = make_int3(mm, nn, pp);
        return i;
    };
    int* imd0 = A.imd + (UDH ? P.imd_off : 0);   // hlnk[2], vlnk[2] per intermediate row, `width` ints each
    auto LNK = [&](int i, int which, int d, int r) -> int& { return imd0[((int64_t) i * 4 + which * 2 + d) * width + (r - lw + 1)]; };
    const int n_im = UDH ? P.n_im : 0;
    const int imd_step = UDH ? (a_right - a_left + n_im) / (n_im + 1) : 0;
    auto gext3 = [&](int i) { return i > codonk1 ? lgep : gep; };
    auto acode = [&](int i) -> int { return i < 0 ? 2 : (i >= P.a_len ? P.a_pad : acod[i]); };      // a_pad: SpdpProblemH
    auto bcode = [&](int i) -> int { return (i < 0 || i > P.b_len) ? 2 : ((cols[i + 2].x >> 16) & 0xff); };
    auto mtx = [&](int aa, int tron) -> int { return Lmtx[aa * 32 + tron]; };
    auto ipen = [&](int len) -> int {
        if (len < 0) return -32768;
        if (len >= A.intpen_len) len = A.intpen_len - 1;
        return A.intpen[len];
    };
    auto sig3_at = [&](int acc) -> int {         // raw sig3[acc], read from the column record of acc + 1 (phs3 > 0 there)
        const int4 c = cols[acc + 1];
        return (((unsigned) c.x >> 24) & 4) ? (int) (short) ((unsigned) c.y >> 16) : (int) (short) (c.y & 0xffff);
    };
    auto spjscr = [&](int don, int acc) -> int {
        return ipen(acc - don) + sig3_at(acc) + A.t53[16 * ((cols[don].w >> 4) & 15) + (cols[acc].w & 15)];
    };
    auto spjseq = [&](int n5, int n3, int& c0, int& c1) {
        c0 = c1 = 2;
        if (n5 < b_left || n3 >= b_right) return;
        const int t0 = bcode(n5 - 2), t1 = bcode(n5 - 1), t2 = bcode(n3), t3 = bcode(n3 + 1);
        if (t0 >= 32 || t1 >= 32 || t2 >= 32 || t3 >= 32) return;
        const int w0 = A.mid[t0], w1 = A.mid[t1], w2 = A.mid[t2], w3 = A.mid[t3];
        if (w1 > 3 || w2 > 3) return;                   // a codon is defined when its own three bases are
        if (w0 <= 3) c0 = A.tron_of[16 * w0 + 4 * w1 + w2];
        if (w3 <= 3) c1 = A.tron_of[16 * w1 + 4 * w2 + w3];
    };
#define LV(slot) L[(slot)][t]
    // slot of H / E / F / the diagonal predecessor (d = 0..3) in plane qq: hfesv / hfesb / hfesc (:301-325)
    auto vslot = [&](int qq, int d) -> int { return d == 0 ? S_HV + qq : d == 1 ? S_EV + qq % 3 : d == 2 ? S_FV + qq : S_QV + qq % 3; };
    auto bslot = [&](int qq, int d) -> int { return d == 0 ? S_HB + qq : d == 1 ? S_EB + qq % 3 : d == 2 ? S_FB + qq : S_QB + qq % 3; };
    auto cslot = [&](int qq, int d) -> int { return d == 0 ? S_HC + qq : d == 1 ? S_EC + qq % 3 : d == 2 ? S_FC + qq : S_QC + qq % 3; };

    // ---- fhinitH1 (:546-689): bulk fills by all lanes, the sequential parts by lane 0
    const int rl = b_left - 3 * a_left;
    for (int e = k; e < 2 * B; e += XN) {
        (hv + lw - 3)[e] = XNEV;
        (hb + lw - 3)[e] = UDH ? a_left : 0;
        (hc + lw - 3)[e] = 0;
    }
    if constexpr (UDH)
        for (int e = k; e < n_im * 4 * width; e += XN) imd0[e] = X_EOU;
    for (int s = 0; s < S_END; ++s) LV(s) = 0;   // the reference's lane planes start out uninitialised; zero here
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (k == 0) {
        *vcount = 0;
        if constexpr (!UDH) {
            int ptr = vadd(0, 0, 0);
            if (!(a_exgl && b_exgl)) ptr = vadd(a_left, b_left, ptr);
            for (int r = rl; r < up; ++r) hc[r] = a_exgl ? 0 : ptr;
            for (int r = lw; r < rl; ++r) hc[r] = b_exgl ? 0 : ptr;
            if (b_exgl == 2) fc[rl] = ptr;
        } else {
            const int re = a_exgl ? rl : up;
            for (int r = lw; r < re; ++r) hc[r] = r;
            for (int i = 0, r = rl; r >= lw; --r) hb[r] = a_left + (i++ / 3);
        }
        if (b_exgl == 1) { for (int r = lw; r < rl; ++r) hv[r] = 0; }
        else if (b_exgl == 2) { fv[rl] = 0; fc[rl] = rl; }
        int rr = b_right - 3 * a_left;
        if (up < rr) rr = up;
        int r = rl;
        if (!a_exgl) {
            if (b_exgl) { fv[r] = 0; fc[r] = hc[r]; }
            hv[r++] = 0;
            hv[r++] = xh_w16(g1);
            hv[r++] = xh_w16(g2);
            hv[r++] = xh_w16(g3);
            if (gep) {
                const int x = (XNEV - g3) / gep + r;
                if (x < rr) rr = x;
                for ( ; r < rr; ++r) hv[r] = xh_w16(hv[r - 3] + gep);
            } else if (rr > r)
                for (const int v = hv[r - 1]; r < rr; ++r) hv[r] = v;
        } else {
            int n = b_left;
            int lend[3] = {r, r + 1, r + 2};
            int bb = n + 1;
            for (int f = 0; f < 3; ++f, ++r, ++n, ++bb) {
                hv[r] = aux[bb].x > 0 ? aux[bb].x : 0;
                if constexpr (!UDH) { hc[r] = vadd(a_left, n, 0); hb[r] = 1; }
                else hc[r] = r;
            }
            for (int f = 0; r < rr; ++r, ++n, ++bb, f = (f + 1) % 3) {
                int h = hv[r - 3];
                hc[r] = hc[r - 3];
                const int gl = r - lend[f];
                if (!(a_exgl & 1) && gl == 3) h = xh_w16(h + gop);
                if (!(a_exgl & 2)) h = xh_w16(h + gext3(gl));
                h = xh_w16(h + aux[bb - 3].z);
                hv[r] = h;
                if (h < XNEV) break;
                int x = xh_w16(hv[r - 1] + g1);
                if (x > h) { hv[r] = h = x; hc[r] = hc[r - 1]; }
                x = xh_w16(hv[r - 2] + g2);
                if (x > h) { hv[r] = h = x; hc[r] = hc[r - 2]; }
                x = aux[bb].x > 0 ? aux[bb].x : 0;
                if (x > h) {
                    hv[r] = x; lend[f] = r;
                    if constexpr (!UDH) { hc[r] = vadd(a_left, n, 0); hb[r] = 1; }
                    else hc[r] = r;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    int max_val = XNEV, max_ulk = X_EOU, max_ml = a_left, max_mr = a_right, max_nr = b_right;
    int rlst[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};       // hb1.rlst; only the lane of the intermediate row uses it
    int imd_i = 0;
    for (int ml = a_left; ml < a_right; ml += XN) {
        const int j9 = min(XN, a_right - ml);
        const int j8 = j9 - 1;
        int n = max(b_left, lw + 3 * ml);
        const int n_first = n;
        const int n9 = min(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;
        int q = xh_mod6(n + 3 * (ml + 1));
        int r = n - 3 * (ml + 1);
        // stripe reset (:849-860): score planes to nevsel, flag planes / side lanes / candidate lists cleared;
        // the link planes keep what the previous stripe left
        for (int i = 0; i < 6; ++i) { LV(S_HV + i) = XNEV; LV(S_FV + i) = XNEV; LV(S_HB + i) = 0; LV(S_FB + i) = 0; }
        for (int i = 0; i < 3; ++i) { LV(S_EV + i) = XNEV; LV(S_EB + i) = 0; LV(S_PS + i) = 0; LV(S_PV + i) = 0; LV(S_CP + i) = 0; }
        for (int i = 0; i < 5; ++i) {
            LV(S_CVAL + i) = XNEV; LV(S_CULK + i) = 0; LV(S_CJNC + i) = 0; LV(S_CML + i) = 0; LV(S_CDIR + i) = 0;
            LV(S_CPHS + i) = -2; LV(S_CIDX + i) = i;
        }
        LV(S_NC) = -1;
        int sm = 0;
        const int m = ml + k;                                 // hb1's `mj`: my row is a[m]
        int mm_ = 0, k9 = 0, k8 = -1, mi = 0;
        bool is_imd_ = false;
        if constexpr (UDH) {
            if (imd_i < n_im) {
                mi = a_left + (imd_i + 1) * imd_step;
                mm_ = a_left + (mi - a_left - 1) / XN * XN;
                k9 = mi - mm_; k8 = k9 - 1;
                is_imd_ = ml == mm_;
            }
        }
        const int mm3 = 3 * mi;
        const bool imd_lane = UDH && imd_i < n_im && (m + 1) == mi;      // Sjsites' is_imd for my row
        const bool site_lane = spj && k <= j8 && (m + 1) < a_right;
        const int* mrow = Lmtx + ((k < j9) ? acod[ml + k] : 0) * 32;
        // loads one step ahead of their use: my column's record, and (lane 0) the one new entry per boundary row that
        // slides into the window hv / hc [r .. r + 3], fv / fc [r + 3], hb [r] -- this stripe writes those rows only
        // behind lane 0's read position (r - 6 j8 and further back), so an entry read a step early is the entry
        const int c_hi = P.b_len + 2;
        int cx_next = cols[min(max(n - 3 * k, 0), c_hi)].x;
        int wH0 = 0, wH1 = 0, wH2 = 0, wH3 = 0, wC0 = 0, wC1 = 0, wC2 = 0, wC3 = 0, wF = 0, wFC = 0, wB = 0;
        if (k == 0) {
            wH0 = hv[r]; wH1 = hv[r + 1]; wH2 = hv[r + 2]; wH3 = hv[r + 3];
            wC0 = hc[r]; wC1 = hc[r + 1]; wC2 = hc[r + 2]; wC3 = hc[r + 3];
            wF = fv[r + 3]; wFC = fc[r + 3];
            if (!UDH || LocalL) wB = hb[r];
        }
        for ( ; n < n9; ++n, ++r, q = xh_mod6(q + 1)) {
            const int f3 = q % 3;
            const int nb = max(0, n - b_right + 1);
            const int kb = (nb - 1) / 3;
            const int ke = min(j9, (n - b_left) / 3);
            const int q1 = xh_mod6(q - 1), q2 = xh_mod6(q - 2), q3 = xh_mod6(q - 3), q4 = xh_mod6(q - 4), q5 = xh_mod6(q - 5);
            const int c = n - 3 * k;                          // my column
            const int cx = cx_next;
            cx_next = cols[min(max(c + 1, 0), c_hi)].x;
            int pH = 0, pC = 0, pF = 0, pFC = 0, pB = 0;
            if (k == 0 && n + 1 < n9) {
                pH = hv[r + 4]; pC = hc[r + 4]; pF = fv[r + 4]; pFC = fc[r + 4];
                if (!UDH || LocalL) pB = hb[r + 1];
            }
            // coding-potential pipe (:870-877)
            if (k == 0 && spj && !nb) LV(S_CP + f3) = (int) (short) (cx & 0xffff);
            const int cv = LV(S_CP + f3);
            {
                const int upcv = xh_up(cv);
                if (k) LV(S_CP + f3) = upcv;
            }
            // rows of the lane above; lane 0 reads the boundary rows the previous stripe left
            int uH3 = xh_up(LV(S_HV + q3)), uF3 = xh_up(LV(S_FV + q3)), uH4 = xh_up(LV(S_HV + q4)), uH5 = xh_up(LV(S_HV + q5));
            int uC3 = xh_up(LV(S_HC + q3)), uFC3 = xh_up(LV(S_FC + q3)), uC4 = xh_up(LV(S_HC + q4)), uC5 = xh_up(LV(S_HC + q5));
            int uH0 = xh_up(LV(S_HV + q)), uC0 = xh_up(LV(S_HC + q)), uB0 = xh_up(LV(S_HB + q));
            int uB3 = 0, uFB3 = 0, uB4 = 0, uB5 = 0;
            if (UDH && LocalL) { uB3 = xh_up(LV(S_HB + q3)); uFB3 = xh_up(LV(S_FB + q3)); uB4 = xh_up(LV(S_HB + q4)); uB5 = xh_up(LV(S_HB + q5)); }
            if (k == 0) {
                uF3 = wF; uFC3 = wFC;
                uH3 = wH3; uC3 = wC3;
                uH4 = wH2; uC4 = wC2;
                uH5 = wH1; uC5 = wC1;
                uH0 = wH0; uC0 = wC0;
                if (!UDH || LocalL) uB0 = wB;
                if (UDH && LocalL) { uFB3 = fb[r + 3]; uB3 = hb[r + 3]; uB4 = hb[r + 2]; uB5 = hb[r + 1]; }
            }
            // insertion: frame shifts, codon insertion, extension (:879-915)
            int ev, eb, ec;
            {
                int h = xh_add(LV(S_HV + q1), g1), hbb = LV(S_HB + q1), hcc = LV(S_HC + q1);
                int x = xh_add(LV(S_HV + q2), g2);
                bool mk = h > x;
                h = mk ? h : x; hbb = mk ? hbb : LV(S_HB + q2); hcc = mk ? hcc : LV(S_HC + q2);
                x = xh_add(xh_add(LV(S_HV + q3), g3), cv);
                mk = h > x;
                h = mk ? h : x; hbb = mk ? hbb : LV(S_HB + q3); hcc = mk ? hcc : LV(S_HC + q3);
                x = xh_add(xh_add(LV(S_EV + f3), gep), cv);
                mk = x > h;
                ev = mk ? x : h; eb = mk ? LV(S_EB + f3) : hbb; ec = mk ? LV(S_EC + f3) : hcc;
            }
            LV(S_EV + f3) = ev; LV(S_EC + f3) = ec;
            if (UDH && LocalL) LV(S_EB + f3) = eb;
            // deletion (:917-965)
            int fvv, fbv, fcv;
            {
                int f = xh_add(uF3, gep), fbb = uFB3, fcc = uFC3;
                int x = xh_add(uH3, g3);
                bool mk = f > x;
                f = mk ? f : x; fbb = mk ? fbb : uB3; fcc = mk ? fcc : uC3;
                x = xh_add(uH4, g2);
                mk = f > x;
                f = mk ? f : x; fbb = mk ? fbb : uB4; fcc = mk ? fcc : uC4;
                x = xh_add(uH5, g1);
                mk = f > x;
                f = mk ? f : x; fbb = mk ? fbb : uB5; fcc = mk ? fcc : uC5;
                fvv = f; fbv = fbb; fcv = fcc;
            }
            LV(S_FV + q) = fvv; LV(S_FC + q) = fcv;
            if (UDH && LocalL) LV(S_FB + q) = fbv;
            // diagonal (:967-1027)
            if (nb) sm = 0;
            if (k >= kb && k < ke) sm = xh_w16(mrow[(cx >> 16) & 0xff]);
            int qb = 0;
            {
                const int qv = uH0, qc = uC0, qbb = uB0;
                int h = xh_add(xh_add(sm, qv), cv);
                LV(S_QV + f3) = qv; LV(S_QC + f3) = qc;
                if (UDH && LocalL) LV(S_QB + f3) = qbb;
                bool mk = fvv > h;
                h = mk ? fvv : h;
                int hcc = mk ? fcv : qc, hbb = mk ? fbv : qbb, code = mk ? 2 : 0;
                mk = ev > h;
                h = mk ? ev : h; hcc = mk ? ec : hcc; hbb = mk ? eb : hbb; code = mk ? 1 : code;
                LV(S_PV + f3) = code;
                LV(S_PS + f3) &= code;
                if (!local) { if (!(h > XNEV)) h = XNEV; }
                else if (LocalL) {
                    if (0 > h) { h = 0; if (!UDH) { code = 1; hcc = 0; } }
                }
                if constexpr (!UDH) {
                    const int diag = code == 0;
                    qb = diag & ~qbb & 1;
                    hbb = diag;
                    LV(S_QB + f3) = qb;
                }
                LV(S_HV + q) = h; LV(S_HC + q) = hcc;
                if (!UDH || LocalL) LV(S_HB + q) = hbb;
            }
            if (UDH && LocalL && k >= kb && k < ke && LV(S_HV + q) == 0) { LV(S_HB + q) = ml + k; LV(S_HC + q) = r - 6 * k; }
            if (LocalR) {                                     // first maximum over lanes 0 .. j8 (vmax)
                int bv = (k < j9) ? LV(S_HV + q) : -0x7fffffff, bk = k;
                for (int o = 1; o < XN; o <<= 1) {
                    const int ov = __shfl_xor(bv, o, XN), ok = __shfl_xor(bk, o, XN);
                    if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
                }
                if (bv > max_val) {
                    max_val = bv;
                    max_ulk = __shfl(LV(S_HC + q), bk, XN);
                    if constexpr (UDH) { max_ml = __shfl(LV(S_HB + q), bk, XN); max_mr = ml + bk + 2; max_nr = n - 3 * (bk + 1); }
                    else { max_mr = ml + bk + 1; max_nr = n - 3 * bk; }
                }
            }
            if constexpr (!UDH)
                if (k >= kb && k < ke && qb) LV(S_HC + q) = vadd(ml + k, n - 3 * (k + 1), LV(S_HC + q));

            // intron 3' boundary (:1049-1054, Sjsites::get :388-494): my column is a queued acceptor column
            const bool in_q = site_lane && c >= n_first && c < b_right;
            unsigned fl = 0;
            if (in_q) fl = (unsigned) cx >> 24;
            if (in_q && (((fl & 7) == 3) || (fl & 4))) {
                const int acc = c - 1;
                const int rr0 = acc - 3 * (m + 1);
                const int gq = q - 1;
                int mx_idx[3] = {-1, -1, -1}, mx_val[3] = {0, 0, 0}, b_idx = -1, b_val = 0;   // maxprd / brd (by candidate slot)
                const int nc = LV(S_NC);
                for (int l = 0; l <= nc; ++l) {
                    const int ci = LV(S_CIDX + l);
                    const int phs = LV(S_CPHS + ci), d = LV(S_CDIR + ci), don = LV(S_CJNC + ci);
                    const int rr = rr0 + phs;
                    if (rr < lw || rr >= up) continue;
                    if (d == 2 && phs == 1) continue;
                    if (acc - don < minl) continue;
                    int x = LV(S_CVAL + ci) + spjscr(don, acc);
                    if (A.cip && P.cip_off >= 0) x += A.cip[P.cip_off + 3 * (m + 1) - phs];      // Cip_score::cip_score, fwd2h1_simd.h:407
                    if (d == 0 && phs) {
                        int c0, c1;
                        spjseq(don, acc, c0, c1);
                        if (phs == 1) x += mtx(acode(m), c0);
                        else x += mtx(acode(m + 1), c1) - mtx(acode(m + 1), bcode(acc)) - aux[acc].z;
                    }
                    const int qq = xh_mod6(gq + phs);
                    const int vs = vslot(qq, d);
                    if (x <= LV(vs)) continue;
                    const int cval = LV(S_CVAL + ci);
                    if (mx_idx[d] < 0 || x > mx_val[d]) {
                        mx_idx[d] = ci; mx_val[d] = cval;
                        if (b_idx < 0 || x > b_val) { b_idx = ci; b_val = cval; }
                    }
                    LV(vs) = xh_w16(x);
                    LV(S_PS + qq % 3) |= (d == 0 ? 4 : d == 1 ? 1 : 8);
                    const int bs = bslot(qq, d), cs = cslot(qq, d);
                    LV(bs) = LV(S_CML + ci);
                    if constexpr (!UDH) {
                        const int inner = vadd(m + 1, don + phs, LV(S_CULK + ci));
                        LV(cs) = vadd(m + 1, acc + phs, inner);
                    } else
                        LV(cs) = LV(S_CULK + ci);
                    if (d && LV(vs) > LV(S_HV + qq)) { LV(S_HV + qq) = LV(vs); LV(S_HB + qq) = LV(bs); LV(S_HC + qq) = LV(cs); }
                    if (k + 1 == XN) {
                        hv[rr] = LV(S_HV + qq);
                        if (imd_lane) LNK(imd_i, 0, 0, rr) = LV(S_CULK + ci);
                        else { hb[rr] = LV(S_HB + qq); hc[rr] = LV(S_HC + qq); }
                        if (d == 2) {
                            fv[rr] = LV(vs);
                            if (imd_lane) LNK(imd_i, 0, 1, rr) = LV(S_CULK + ci);
                            else { fb[rr] = LV(bs); fc[rr] = LV(cs); }
                        }
                    }
                }
                if (imd_lane && b_idx >= 0) {
                    const int maxd = LV(S_CDIR + b_idx);
                    const int pi_ = mx_idx[maxd];
                    const int qq = xh_mod6(gq + LV(S_CPHS + pi_));
                    const int lstr = acc + LV(S_CPHS + pi_) - mm3;
                    rlst[qq % 3] = lstr;
                    LNK(imd_i, 0, 0, lstr) = LV(S_CULK + pi_);
                    LV(cslot(qq, maxd)) = lstr;
                    LV(S_PV + qq % 3) = maxd;
                    if (maxd) LV(S_HC + qq) = LV(cslot(qq, maxd));
                    else {
                        if (mx_idx[1] >= 0 && LV(S_EV + qq % 3) > LV(S_HV + qq) + gop) {
                            LNK(imd_i, 0, 1, lstr) = LV(S_CULK + mx_idx[1]);
                            LV(S_EC + qq % 3) = lstr + width;
                        }
                        if (mx_idx[2] >= 0 && LV(S_FV + qq) > LV(S_HV + qq) + gop) LV(S_FC + qq) = lstr + width;
                    }
                }
            }
            // intron 5' boundary (:1056-1061, Sjsites::put :496-543)
            if (in_q && ((((fl >> 3) & 7) == 3) || ((fl >> 3) & 4))) {
                const int don = c - 1;
                const int sigJ = aux[don].w;
                int nn = c, pq = q;
                for (int phs = 1; phs > -2; --nn, --phs, pq = xh_mod6(pq - 1)) {
                    const int rr = don - 3 * (m + 1) + phs;
                    if (rr < lw || rr >= up) continue;
                    const int pf = pq % 3;
                    const int h = LV(S_PV + pf);
                    const int thr = LV(S_HV + pq) + gop;
                    for (int kk = (h && phs < 1) ? 1 : 0; kk < 3; ++kk) {
                        if (LV(S_PS + pf) & (kk == 0 ? 4 : kk == 1 ? 1 : 8)) continue;
                        const int cross = (phs == 1 && kk == 0) ? 3 : kk;
                        const int from = LV(vslot(pq, cross));
                        if (kk && from <= thr) continue;
                        const int x = from + sigJ;
                        if (x <= XNEV) continue;
                        int nc = LV(S_NC);
                        int l = nc < 4 ? ++nc : 4;
                        while (--l >= 0) {
                            const int il = LV(S_CIDX + l);
                            if (x >= LV(S_CVAL + il)) { LV(S_CIDX + l) = LV(S_CIDX + l + 1); LV(S_CIDX + l + 1) = il; }
                            else break;
                        }
                        if (++l < 4) {
                            const int ci = LV(S_CIDX + l);
                            LV(S_CVAL + ci) = xh_w16(x);
                            LV(S_CML + ci) = LV(bslot(pq, kk));
                            const int rq = nn - mm3;
                            if (imd_lane) {
                                if (kk == 1) LNK(imd_i, 0, 0, rq) = rlst[pf];
                                LV(S_CULK + ci) = rq;
                            } else
                                LV(S_CULK + ci) = LV(cslot(pq, cross));
                            LV(S_CJNC + ci) = don; LV(S_CDIR + ci) = kk; LV(S_CPHS + ci) = phs;
                        } else --nc;
                        LV(S_NC) = nc;
                    }
                }
            }
            // intermediate row (:1375-1384)
            if constexpr (UDH) {
                const int rj = r - 6 * k8;
                if (is_imd_ && rj >= lw && rj <= up) {
                    // lane k8 holds the row's side lanes, lane k9 - 1 == k8 its H / F planes ([k9] in the reference's layout)
                    if (k == k8) {
                        if (LV(S_PV + f3) == 0) rlst[f3] = rj;
                        if (LV(S_PV + f3) == 1) LNK(imd_i, 0, 0, rj) = rlst[f3];
                        LNK(imd_i, 1, 0, rj) = LV(S_HC + q);
                        LV(S_HC + q) = rj;
                        LNK(imd_i, 1, 1, rj) = LV(S_FC + q);
                        LV(S_FC + q) = rj + width;
                    }
                }
            }
            // hand the bottom row to the next stripe (:1063-1073 / :1386-1397)
            const int r0 = r - 6 * j8;
            if (k == j8 && j9 == ke && lw <= r0 && (UDH ? r0 < up : r0 <= up)) {
                hv[r0] = LV(S_HV + q); hc[r0] = LV(S_HC + q);
                fv[r0] = LV(S_FV + q); fc[r0] = LV(S_FC + q);
                if constexpr (!UDH) hb[r0] = LV(S_HB + q);
                else if (LocalL) { hb[r0] = LV(S_HB + q); fb[r0] = LV(S_FB + q); }
            }
            wH0 = wH1; wH1 = wH2; wH2 = wH3; wH3 = pH;
            wC0 = wC1; wC1 = wC2; wC2 = wC3; wC3 = pC;
            wF = pF; wFC = pFC; wB = pB;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if constexpr (UDH) { if (is_imd_) ++imd_i; }
        // rlst lives on whichever lane held the intermediate row; the next one may sit on another lane, and the
        // reference keeps one array for all: pass it on
        if constexpr (UDH) {
            if (is_imd_) for (int i = 0; i < 3; ++i) rlst[i] = __shfl(rlst[i], k8, XN);
        }
    }
    if (k) return;

    // ---- fhlastH1 (:691-791) by lane 0
    int ptr = 0, maxt = 0;
    const bool by_last = UDH ? !(LocalR && max_mr < a_right) : (!LocalR || max_mr == a_right);
    if (by_last) {
        int glen[3] = {0, 0, 0};
        bool tcdn[3] = {false, false, false};
        const int m3 = 3 * a_right;
        int rw = lw;
        int rf = b_left - m3;
        if (rf > rw) rw = rf; else rf = rw;
        const int rr = b_right - m3;
        int maxr = rr, mx = rr;
        int bb = rw + m3;
        if (a_exgr) {
            int f = 0;
            for (int h = rw; h <= rr; ++h, ++rf, ++bb, f = (f + 1) % 3) {
                glen[f] += 3;
                int cand[3] = {hv[h], XNEV, XNEV};
                if (rf - rw >= 3 && !tcdn[f]) {
                    cand[1] = hv[h - 3] + aux[bb - 2].z;
                    if (!(a_exgr & 2)) cand[1] += gext3(glen[f]);
                    if (!(a_exgr & 1) && glen[f] == 3) cand[1] += gop;
                    if (sc->term_codon) cand[2] = hv[h - 3] + aux[bb - 2].y;
                }
                if (rf - rw >= 3) tcdn[f] = tcdn[f] || aux[bb - 2].y > 0;
                const int s5 = (local && aux[bb].w > 0) ? aux[bb].w : 0;
                cand[0] += s5; cand[1] += s5;
                int kk = 0;
                if (cand[1] > cand[kk]) kk = 1;
                if (cand[2] > cand[kk]) kk = 2;
                if (kk == 0) { glen[f] = 0; tcdn[f] = false; }
                else if (kk == 1) hv[h] = xh_w16(cand[1] - s5);
                else hv[h] = xh_w16(cand[2]);
                if (hv[h] > hv[mx]) { mx = h; maxr = rf - (kk == 2 ? 3 : 0); }
            }
        } else {
            const int y = xh_w16(hv[rr - 3] + aux[bb + (rr - rw)].y);
            if (y > hv[rr]) { hv[rr] = y; maxr = rr - 3; }
        }
        if (b_exgr) {
            rw = min(up - 1, b_right - 3 * a_left);
            int g[3] = {XNEV, XNEV, XNEV};
            int f = 0;
            for (int h = rw - 3; h > rr; --h, f = (f + 1) % 3) {
                int x = hv[h + 3];
                if (!(b_exgr & 1)) x = xh_w16(x + gop);
                if (x > g[f]) g[f] = x;
                if (!(b_exgr & 2)) g[f] = xh_w16(g[f] + gep);
                if (hv[h] > g[f]) g[f] = XNEV;
                else if (g[f] > hv[mx]) { mx = h; hv[h] = g[f]; }
            }
        }
        maxt = mx;
        if constexpr (UDH) hb[maxt] = hb[maxr];
        max_ulk = hc[maxr];
        int qd = maxr - rr;
        if constexpr (!UDH) {
            int m9 = a_right, n9 = b_right;
            if (qd > 0) { m9 -= (qd + 2) / 3; if (qd %= 3) n9 -= 3 - qd; }
            else if (qd < 0) n9 += qd;
            max_ulk = vadd(m9, n9, max_ulk);
            if (maxr != maxt) max_ulk = vadd(a_right, maxt + m3, max_ulk);
        } else {
            if (qd > 0) max_mr = (b_right - maxr) / 3;
            else        max_nr = maxt + m3;
        }
        ptr = max_ulk;
    } else if constexpr (!UDH)
        ptr = vadd(max_mr, max_nr, max_ulk);

    DevResultH R;
    R.score = max_val; R.mr = max_mr; R.nr = max_nr; R.maxt = maxt; R.maxr = 0; R.pad[0] = max_ulk; R.pad[1] = R.pad[2] = 0;
    A.res[pi] = R;

    if constexpr (!UDH) {
        // Vmf::traceback(ptr) + the fix-up of trcbkalignH_ng
        int2* out = A.skl + (int64_t) pi * A.skl_cap;
        const int vn = *vcount;
        int cnt = 0, status = vn > vcap ? -3 : 0;
        {   // mode 3 keeps the record pointer in one int16 lane (undefined in the reference beyond 32767 records)
            const int mq = a_right - a_left;
            float cvol = (float) (lw - b_left + 3 * a_right);
            cvol = (float) mq * (float) (b_right - b_left) - cvol * cvol / 3;
            if (!status && cvol < 65535.f && vn > 32767) status = -4;
        }
        if (ptr && !status) {
            int3 sv = vrec[ptr];
            int lm = 0, ln = 0;
            for (;;) {
                if (cnt < A.skl_cap) out[cnt] = make_int2(sv.x, sv.y); else status = -1;
                lm = sv.x; ln = sv.y; ++cnt;
                if (!sv.z) break;
                sv = vrec[sv.z];
            }
            const int rd = local ? 0 : ((ln - 3 * lm) - b_left + 3 * a_left);
            if (rd) {
                const int2 rec = rd > 0 ? make_int2(a_left, b_left + rd) : make_int2(a_left - rd / 3, b_left);
                if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
                ++cnt;
            }
        }
        A.n_skl[pi] = status ? status : cnt;
    } else {
        // the tail of hirschbergH1 (:1419-1469): walk the links back through the intermediate rows
        int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
        for (int i = 0; i < A.cpos_stride; ++i) cpos[i] = X_EOU;
        auto mi_of = [&](int i) { return a_left + (i + 1) * imd_step; };
#define CPOS(i, c) cpos[(i) * 10 + (c)]
        int al = a_left, ar = a_right, bl = b_left, br = b_right;
        if (by_last) max_ml = LocalL ? hb[maxt] : a_left;
        ar = max_mr; br = max_nr;
        int val = max_val;
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
        int r = max_ulk;
        for ( ; i >= 0 && mi_of(i) > max_ml; --i) {
            int c = 0, d = 0;
            for ( ; r > up; r -= width) ++d;
            if (LNK(i, 1, d, r) < X_EOU) {
                CPOS(i, c++) = mi_of(i);
                CPOS(i, c++) = (d > 0) ? 1 : 0;
                const int m3 = 3 * mi_of(i);
                for (int rp = LNK(i, 0, d, r); lw <= rp && rp < up && r != rp && c < 8; rp = LNK(i, 0, d, r = rp))
                    CPOS(i, c++) = r + m3;
                CPOS(i, c++) = r + m3;
                CPOS(i, c) = X_EOU;
                r = LNK(i, 1, d, r);
                if (r == X_EOU) break;
            } else
                CPOS(i, 0) = X_EOU;
        }
        for ( ; r > up; r -= width) ;
        if (LocalL) { al = max_ml; bl = r + 3 * al; }
        else {
            const int rl2 = bl - 3 * al;
            if (b_exgl && rl2 > r) {
                al = (bl - r) / 3;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = X_EOU;
            }
            if (a_exgl && rl2 < r) bl = 3 * al + r;
        }
        ++i;
        if ((i >= 0 && i < n_im && mi_of(i) < al) || CPOS(i, 2) < bl) val = INT32_MIN / 16 * 7;
#undef CPOS
        A.scores[pi] = val;
        int* rg = A.ranges + 4 * (int64_t) pi;
        rg[0] = al; rg[1] = ar; rg[2] = bl; rg[3] = br;
    }
#undef LV
}

// ---- hirschbergH1_wip with local ends (-LS), src/fwd2h1_wip_simd.h:338-773 ---------------------------
// The production linear-space sweep (spdh_udh, spdp_h_udh.hip) covers the non-local form.  With local ends the
// reference carries the left-end row on H / E / F and the three phase donors as well, restarts paths at zero
// and tracks the best cell -- and, as in the cDNA engine, reads lane state it does not re-initialise: the link
// planes survive from stripe to stripe, and on stripes that hold an intermediate row the substitution lane is
// overwritten with flag vectors (`Store(sm_a, ..)`, :601 / :676 / :707) that out-of-range lanes then add as
// scores.  Same literal 16-lanes-per-problem form as spdh_exact above (planes in the lane's LDS column).
enum { W_HV = 0, W_FV = 6, W_HB = 12, W_FB = 18, W_HC = 24, W_FC = 30, W_EV = 36, W_EB = 39, W_EC = 42, W_CP = 45,
       W_S3 = 48, W_P3 = 54, W_S5 = 60, W_P5 = 66, W_IV = 72, W_IL = 75, W_IC = 78, W_IB = 81, W_END = 84 };

__global__ void __launch_bounds__(64) spdh_local_udh(HScalarArgs A)
{
    __shared__ int L[W_END][64];
    const int t = threadIdx.x;
    const int k = t & 15;
    const int grp = t >> 4;
    const int pi = blockIdx.x * 4 + grp;
    if (pi >= A.n_probs) return;
    const DevProblemH P = A.probs[pi];
    const DevScoringH* sc = A.sc;
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, B = P.buf_size;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const bool local = sc->local;
    const bool LocalL = local && a_exgl && b_exgl, LocalR = local && a_exgr && b_exgr;
    const bool spj = sc->spj;
    const int gop = sc->gop, gep = sc->gep, lgep = sc->lgep, codonk1 = sc->codonk1;
    const int g1 = sc->g1, g2 = sc->g2, g3 = sc->g3;
    const int llmt = sc->llmt, nquant = sc->nquant;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int4* cols = A.cols + P.col_off;
    const short4* aux = A.aux + P.col_off;
    int* hv = A.work + P.bnd_off - lw + 3;
    int* fv = hv + B;
    int* hb = fv + B;
    int* fb = hb + B;
    int* hc = fb + B;
    int* fc = hc + B;
    int* imd0 = A.imd + P.imd_off;
    auto LNK = [&](int i, int which, int d, int r) -> int& { return imd0[((int64_t) i * 4 + which * 2 + d) * width + (r - lw + 1)]; };
    const int n_im = P.n_im;
    const int imd_step = (a_right - a_left + n_im) / (n_im + 1);
    auto gext3 = [&](int i) { return i > codonk1 ? lgep : gep; };
    auto qpen = [&](int hil) -> int {
        int pv = sc->qm_pen[0];
        for (int j = 1; j < nquant; ++j) if (hil > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
        return pv;
    };
#define LV(slot) L[(slot)][t]
    const int rl = b_left - 3 * a_left;
    for (int e = k; e < 2 * B; e += XN) {
        (hv + lw - 3)[e] = XNEV;
        (hb + lw - 3)[e] = a_left;
        (hc + lw - 3)[e] = 0;
    }
    for (int e = k; e < n_im * 4 * width; e += XN) imd0[e] = X_EOU;
    for (int s = 0; s < W_END; ++s) LV(s) = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (k == 0) {                                        // fhinitH1, Hirschberg form (:588-689)
        const int re = a_exgl ? rl : up;
        for (int r = lw; r < re; ++r) hc[r] = r;
        for (int i = 0, r = rl; r >= lw; --r) hb[r] = a_left + (i++ / 3);
        if (b_exgl == 1) { for (int r = lw; r < rl; ++r) hv[r] = 0; }
        else if (b_exgl == 2) { fv[rl] = 0; fc[rl] = rl; }
        int rr = b_right - 3 * a_left;
        if (up < rr) rr = up;
        int r = rl;
        if (!a_exgl) {
            if (b_exgl) { fv[r] = 0; fc[r] = hc[r]; }
            hv[r++] = 0;
            hv[r++] = xh_w16(g1);
            hv[r++] = xh_w16(g2);
            hv[r++] = xh_w16(g3);
            if (gep) {
                const int x = (XNEV - g3) / gep + r;
                if (x < rr) rr = x;
                for ( ; r < rr; ++r) hv[r] = xh_w16(hv[r - 3] + gep);
            } else if (rr > r)
                for (const int v = hv[r - 1]; r < rr; ++r) hv[r] = v;
        } else {
            int lend[3] = {r, r + 1, r + 2};
            int bb = b_left + 1;
            for (int f = 0; f < 3; ++f, ++r, ++bb) { hv[r] = aux[bb].x > 0 ? aux[bb].x : 0; hc[r] = r; }
            for (int f = 0; r < rr; ++r, ++bb, f = (f + 1) % 3) {
                int h = hv[r - 3];
                hc[r] = hc[r - 3];
                const int gl = r - lend[f];
                if (!(a_exgl & 1) && gl == 3) h = xh_w16(h + gop);
                if (!(a_exgl & 2)) h = xh_w16(h + gext3(gl));
                h = xh_w16(h + aux[bb - 3].z);
                hv[r] = h;
                if (h < XNEV) break;
                int x = xh_w16(hv[r - 1] + g1);
                if (x > h) { hv[r] = h = x; hc[r] = hc[r - 1]; }
                x = xh_w16(hv[r - 2] + g2);
                if (x > h) { hv[r] = h = x; hc[r] = hc[r - 2]; }
                x = aux[bb].x > 0 ? aux[bb].x : 0;
                if (x > h) { hv[r] = x; lend[f] = r; hc[r] = r; }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    int max_val = XNEV, max_ulk = X_EOU, max_ml = a_left, max_mr = a_right, max_nr = b_right;
    int rlst[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int imd_i = 0;
    int sm = 0;                                          // sm_a: kept across stripes except for the per-stripe clear
    for (int ml = a_left; ml < a_right; ml += XN) {
        const int j9 = min(XN, a_right - ml);
        const int j8 = j9 - 1;
        int n = max(b_left, lw + 3 * ml);
        const int n9 = min(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;
        int q = xh_mod6(n + 3 * (ml + 1));
        int r = n - 3 * (ml + 1);
        int donor_r[3] = {r, r, r};
        for (int i = 0; i < 6; ++i) {
            LV(W_HV + i) = XNEV; LV(W_FV + i) = XNEV; LV(W_HB + i) = 0; LV(W_FB + i) = 0;
            LV(W_S3 + i) = 0; LV(W_P3 + i) = 0; LV(W_S5 + i) = 0; LV(W_P5 + i) = 0;
        }
        for (int i = 0; i < 3; ++i) {
            LV(W_EV + i) = XNEV; LV(W_EB + i) = 0; LV(W_CP + i) = 0;
            LV(W_IV + i) = XNEV; LV(W_IL + i) = 0; LV(W_IC + i) = 0; LV(W_IB + i) = 0;
        }
        sm = 0;
        int mm_ = 0, k9 = 0, k8 = -1;
        bool is_imd_ = false;
        if (imd_i < n_im) {
            const int mi = a_left + (imd_i + 1) * imd_step;
            mm_ = a_left + (mi - a_left - 1) / XN * XN;
            k9 = mi - mm_; k8 = k9 - 1;
            is_imd_ = ml == mm_;
        }
        (void) k9;
        const int* mrow = sc->mtx + ((k < j9) ? acod[ml + k] : 0) * 32;
        for ( ; n < n9; ++n, ++r, q = xh_mod6(q + 1)) {
            const int f3 = q % 3;
            const int rj = r - 6 * k8;
            const int nb = max(0, n - b_right + 1);
            const int kb = (nb - 1) / 3;
            const int ke = min(j9, (n - b_left) / 3);
            const bool is_imd = is_imd_ && rj >= lw && rj <= up;
            const int q1 = xh_mod6(q - 1), q2 = xh_mod6(q - 2), q3 = xh_mod6(q - 3), q4 = xh_mod6(q - 4), q5 = xh_mod6(q - 5);
            const int c = n - 3 * k;
            int4 col0 = make_int4(0, 0, 0, 0);           // the column record of this step (lane 0 feeds the pipes)
            if (n < P.col_len) col0 = cols[n];
            const unsigned fl0 = nb ? 0u : ((unsigned) col0.x >> 24);
            // coding potential pipe (unconditional here, :121-125)
            if (k == 0) LV(W_CP + f3) = (int) (short) (col0.x & 0xffff);
            const int cv = LV(W_CP + f3);
            { const int u = xh_up(cv); if (k) LV(W_CP + f3) = u; }
            int uH3 = xh_up(LV(W_HV + q3)), uF3 = xh_up(LV(W_FV + q3)), uH4 = xh_up(LV(W_HV + q4)), uH5 = xh_up(LV(W_HV + q5));
            int uC3 = xh_up(LV(W_HC + q3)), uFC3 = xh_up(LV(W_FC + q3)), uC4 = xh_up(LV(W_HC + q4)), uC5 = xh_up(LV(W_HC + q5));
            int uB3 = xh_up(LV(W_HB + q3)), uFB3 = xh_up(LV(W_FB + q3)), uB4 = xh_up(LV(W_HB + q4)), uB5 = xh_up(LV(W_HB + q5));
            int uH0 = xh_up(LV(W_HV + q)), uC0 = xh_up(LV(W_HC + q)), uB0 = xh_up(LV(W_HB + q));
            if (k == 0) {
                uF3 = fv[r + 3]; uFC3 = fc[r + 3]; uH3 = hv[r + 3]; uC3 = hc[r + 3];
                uH4 = hv[r + 2]; uC4 = hc[r + 2]; uH5 = hv[r + 1]; uC5 = hc[r + 1];
                uH0 = hv[r]; uC0 = hc[r];
                // the `ml` feeds exist only with local left ends; entry 0 otherwise keeps the per-stripe zero
                if (LocalL) { uFB3 = fb[r + 3]; uB3 = hb[r + 3]; uB4 = hb[r + 2]; uB5 = hb[r + 1]; uB0 = hb[r]; }
                else { uFB3 = uB3 = uB4 = uB5 = uB0 = 0; }
            }
            // insertion
            int ev, eb, ec;
            {
                int h = xh_add(LV(W_HV + q1), g1), cc = LV(W_HC + q1), bb2 = LV(W_HB + q1);
                int x = xh_add(LV(W_HV + q2), g2);
                bool mk = h > x;
                h = mk ? h : x; cc = mk ? cc : LV(W_HC + q2); bb2 = mk ? bb2 : LV(W_HB + q2);
                x = xh_add(xh_add(LV(W_HV + q3), g3), cv);
                mk = h > x;
                h = mk ? h : x; cc = mk ? cc : LV(W_HC + q3); bb2 = mk ? bb2 : LV(W_HB + q3);
                x = xh_add(xh_add(LV(W_EV + f3), gep), cv);
                mk = x > h;
                ev = mk ? x : h; ec = mk ? LV(W_EC + f3) : cc; eb = mk ? LV(W_EB + f3) : bb2;
            }
            LV(W_EV + f3) = ev; LV(W_EC + f3) = ec;
            if (LocalL) LV(W_EB + f3) = eb;
            // deletion
            int fvv, fcc, fbb;
            {
                int h = xh_add(uF3, gep), cc = uFC3, bb2 = uFB3;
                int x = xh_add(uH3, g3);
                bool mk = h > x;
                h = mk ? h : x; cc = mk ? cc : uC3; bb2 = mk ? bb2 : uB3;
                x = xh_add(uH4, g2);
                mk = h > x;
                h = mk ? h : x; cc = mk ? cc : uC4; bb2 = mk ? bb2 : uB4;
                x = xh_add(uH5, g1);
                mk = h > x;
                fvv = mk ? h : x; fcc = mk ? cc : uC5; fbb = mk ? bb2 : uB5;
            }
            LV(W_FV + q) = fvv; LV(W_FC + q) = fcc;
            if (LocalL) LV(W_FB + q) = fbb;
            // diagonal, best of three
            if (nb) sm = 0;
            if (k >= kb && k < ke) sm = xh_w16(mrow[(cols[c].x >> 16) & 0xff]);
            const int dv = uH0;
            int hx, hcx, hbx, pb, ab = 0;
            {
                int h = xh_add(xh_add(sm, dv), cv);
                int cc = uC0, bb2 = uB0;
                bool mk = fvv > h;
                h = mk ? fvv : h; cc = mk ? fcc : cc; bb2 = mk ? fbb : bb2;
                pb = mk ? 2 : 0;
                mk = ev > h;
                h = mk ? ev : h; cc = mk ? ec : cc; bb2 = mk ? eb : bb2;
                pb = mk ? 1 : pb;
                hx = h; hcx = cc; hbx = bb2;
            }
            const int pv_k8 = __shfl(pb, max(k8, 0), XN);    // PV[f3][k8] of this step
            // intron 3' boundary: two legs (phase of the site, and +1 when both phases are sites)
            if (spj) {
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int pk = 2 * f3 + k2;
                    int f_s = SPDH_MIN_SSV, f_p = 0;
                    const unsigned a3 = fl0 & 7u;            // (phase + 2) | 4 when phs3 == 2
                    if (!k2) { if (a3 & 3u) { f_s = (int) (short) (col0.y & 0xffff); f_p = (int) (a3 & 3u); } }
                    else if (a3 & 4u) { f_s = (int) (short) ((unsigned) col0.y >> 16); f_p = 3; }
                    if (k == 0) { LV(W_S3 + pk) = f_s; LV(W_P3 + pk) = f_p; }
                    const int ss = LV(W_S3 + pk), ph = LV(W_P3 + pk);
                    { const int us = xh_up(ss), upp = xh_up(ph); if (k) { LV(W_S3 + pk) = us; LV(W_P3 + pk) = upp; } }
                    const unsigned long long bal = __ballot(ph != 0);
                    if (!((bal >> (grp * 16)) & 0xffffull)) continue;
                    for (int f = 0; f < 3; ++f) {
                        int x = xh_add(LV(W_IV + f), ss);
                        x = xh_add(x, qpen(LV(W_IL + f)));
                        x = (ph == f + 1) ? x : XNEV;
                        x = (LV(W_IL + f) > llmt) ? x : XNEV;
                        const bool mk = x > hx;
                        hx = mk ? x : hx;
                        hcx = mk ? LV(W_IC + f) : hcx;
                        if (LocalL) hbx = mk ? LV(W_IB + f) : hbx;
                        ab |= mk ? 1 : 0;
                        if (is_imd) {
                            sm = mk ? 1 : 0;                 // Store(sm_a, qv_v)
                            if (k == k8 && mk) {
                                LNK(imd_i, 0, 0, rj) = donor_r[f];
                                LNK(imd_i, 0, 1, rj) = donor_r[f] + width;
                                rlst[f3] = rj;
                            }
                        }
                    }
                }
            }
            if (LocalL && 0 > hx) hx = 0;
            LV(W_HV + q) = hx; LV(W_HC + q) = hcx;
            if (LocalL) LV(W_HB + q) = hbx;
            if (LocalL && k >= kb && k < ke && hx == 0) { LV(W_HB + q) = xh_w16(ml + k); LV(W_HC + q) = r - 6 * k; }
            if (LocalR) {
                int bv = (k < j9) ? LV(W_HV + q) : -0x7fffffff, bk = k;
                for (int o = 1; o < XN; o <<= 1) {
                    const int ov = __shfl_xor(bv, o, XN), ok = __shfl_xor(bk, o, XN);
                    if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
                }
                if (bv > max_val) {
                    max_val = bv;
                    max_ml = __shfl(LV(W_HB + q), bk, XN); max_ulk = __shfl(LV(W_HC + q), bk, XN);
                    max_mr = ml + bk + 2; max_nr = n - 3 * (bk + 1);
                }
            }
            // intron 5' boundary
            if (spj) {
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int pk = 2 * f3 + k2;
                    int f_s = SPDH_MIN_SSV, f_p = 0;
                    const unsigned a5 = (fl0 >> 3) & 7u;
                    if (!k2) { if (a5 & 3u) { f_s = (int) (short) (col0.z & 0xffff); f_p = (int) (a5 & 3u); } }
                    else if (a5 & 4u) { f_s = (int) (short) ((unsigned) col0.z >> 16); f_p = 3; }
                    if (k == 0) { LV(W_S5 + pk) = f_s; LV(W_P5 + pk) = f_p; }
                    const int ss = LV(W_S5 + pk), ph = LV(W_P5 + pk);
                    { const int us = xh_up(ss), upp = xh_up(ph); if (k) { LV(W_S5 + pk) = us; LV(W_P5 + pk) = upp; } }
                    const unsigned long long bal = __ballot(ph != 0);
                    if (!((bal >> (grp * 16)) & 0xffffull)) continue;
                    for (int f = k2 ? 2 : 0; f < 3; ++f) {
                        int x = (f == 2) ? xh_add(dv, ss) : xh_add(hx, ss);
                        x = (ab == 0) ? x : XNEV;
                        x = (ph == f + 1) ? x : XNEV;
                        const bool mk = x > LV(W_IV + f);
                        LV(W_IV + f) = mk ? x : LV(W_IV + f);
                        LV(W_IL + f) = xh_add(mk ? 0 : LV(W_IL + f), 1);
                        LV(W_IC + f) = mk ? hcx : LV(W_IC + f);
                        if (LocalL) LV(W_IB + f) = mk ? hbx : LV(W_IB + f);
                        if (is_imd) {
                            sm = mk ? 1 : 0;                 // Store(sm_a, pv_v)
                            if (k == k8 && mk) donor_r[f] = rj;
                        }
                    }
                }
            }
            if (is_imd) {
                sm = ab;
                if (k == k8) {
                    if (pv_k8 == 0) rlst[f3] = rj;
                    if (!ab && pv_k8 == 1) LNK(imd_i, 0, 0, rj) = rlst[f3];
                    LNK(imd_i, 1, 0, rj) = LV(W_HC + q); LV(W_HC + q) = rj;
                    LNK(imd_i, 1, 1, rj) = LV(W_FC + q); LV(W_FC + q) = rj + width;
                }
            }
            const int r0 = r - 6 * j8;
            if (k == j8 && j9 == ke && lw <= r0 && r0 <= up) {
                hv[r0] = LV(W_HV + q); hc[r0] = LV(W_HC + q);
                fv[r0] = LV(W_FV + q); fc[r0] = LV(W_FC + q);
                if (LocalL) { hb[r0] = LV(W_HB + q); fb[r0] = LV(W_FB + q); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (is_imd_) {
            for (int i = 0; i < 3; ++i) rlst[i] = __shfl(rlst[i], k8, XN);
            ++imd_i;
        }
    }
    if (k) return;

    // ---- fhlastH1 (unless a local right end inside the matrix was tracked), then the link walk (:735-773)
    int maxt = 0;
    const bool by_last = !(LocalR && max_mr < a_right);
    if (by_last) {
        int glen[3] = {0, 0, 0};
        bool tcdn[3] = {false, false, false};
        const int m3 = 3 * a_right;
        int rw = lw;
        int rf = b_left - m3;
        if (rf > rw) rw = rf; else rf = rw;
        const int rr = b_right - m3;
        int maxr = rr, mx = rr;
        int bb = rw + m3;
        if (a_exgr) {
            int f = 0;
            for (int h = rw; h <= rr; ++h, ++rf, ++bb, f = (f + 1) % 3) {
                glen[f] += 3;
                int cand[3] = {hv[h], XNEV, XNEV};
                if (rf - rw >= 3 && !tcdn[f]) {
                    cand[1] = hv[h - 3] + aux[bb - 2].z;
                    if (!(a_exgr & 2)) cand[1] += gext3(glen[f]);
                    if (!(a_exgr & 1) && glen[f] == 3) cand[1] += gop;
                    if (sc->term_codon) cand[2] = hv[h - 3] + aux[bb - 2].y;
                }
                if (rf - rw >= 3) tcdn[f] = tcdn[f] || aux[bb - 2].y > 0;
                const int s5 = (local && aux[bb].w > 0) ? aux[bb].w : 0;
                cand[0] += s5; cand[1] += s5;
                int kk = 0;
                if (cand[1] > cand[kk]) kk = 1;
                if (cand[2] > cand[kk]) kk = 2;
                if (kk == 0) { glen[f] = 0; tcdn[f] = false; }
                else if (kk == 1) hv[h] = xh_w16(cand[1] - s5);
                else hv[h] = xh_w16(cand[2]);
                if (hv[h] > hv[mx]) { mx = h; maxr = rf - (kk == 2 ? 3 : 0); }
            }
        } else {
            const int y = xh_w16(hv[rr - 3] + aux[bb + (rr - rw)].y);
            if (y > hv[rr]) { hv[rr] = y; maxr = rr - 3; }
        }
        if (b_exgr) {
            rw = min(up - 1, b_right - 3 * a_left);
            int g[3] = {XNEV, XNEV, XNEV};
            int f = 0;
            for (int h = rw - 3; h > rr; --h, f = (f + 1) % 3) {
                int x = hv[h + 3];
                if (!(b_exgr & 1)) x = xh_w16(x + gop);
                if (x > g[f]) g[f] = x;
                if (!(b_exgr & 2)) g[f] = xh_w16(g[f] + gep);
                if (hv[h] > g[f]) g[f] = XNEV;
                else if (g[f] > hv[mx]) { mx = h; hv[h] = g[f]; }
            }
        }
        maxt = mx;
        hb[maxt] = hb[maxr];
        max_ulk = hc[maxr];
        const int qd = maxr - rr;
        if (qd > 0) max_mr = (b_right - maxr) / 3;
        else        max_nr = maxt + m3;
        max_ml = LocalL ? hb[maxt] : a_left;
    }
    {
        int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
        for (int i = 0; i < A.cpos_stride; ++i) cpos[i] = X_EOU;
        auto mi_of = [&](int i) { return a_left + (i + 1) * imd_step; };
#define CPOS(i, c) cpos[(i) * 10 + (c)]
        int al = a_left, bl = b_left;
        const int ar = max_mr, br = max_nr;
        int val = max_val;
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
        int r = max_ulk;
        for ( ; i >= 0 && mi_of(i) > max_ml; --i) {
            int c = 0, d = 0;
            for ( ; r > up; r -= width) ++d;
            if (LNK(i, 1, d, r) < X_EOU) {
                CPOS(i, c++) = mi_of(i);
                CPOS(i, c++) = (d > 0) ? 1 : 0;
                const int m3 = 3 * mi_of(i);
                for (int rp = LNK(i, 0, d, r); lw <= rp && rp < up && r != rp && c < 8; rp = LNK(i, 0, d, r = rp))
                    CPOS(i, c++) = r + m3;
                CPOS(i, c++) = r + m3;
                CPOS(i, c) = X_EOU;
                r = LNK(i, 1, d, r);
                if (r == X_EOU) break;
            } else
                CPOS(i, 0) = X_EOU;
        }
        for ( ; r > up; r -= width) ;
        if (LocalL) { al = max_ml; bl = r + 3 * al; }
        else {
            const int rl2 = bl - 3 * al;
            if (b_exgl && rl2 > r) {
                al = (bl - r) / 3;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = X_EOU;
            }
            if (a_exgl && rl2 < r) bl = 3 * al + r;
        }
        ++i;
        if ((i >= 0 && i < n_im && mi_of(i) < al) || CPOS(i, 2) < bl) val = INT32_MIN / 16 * 7;
#undef CPOS
        A.scores[pi] = val;
        int* rg = A.ranges + 4 * (int64_t) pi;
        rg[0] = al; rg[1] = ar; rg[2] = bl; rg[3] = br;
    }
#undef LV
}

extern "C" hipError_t spdh_launch_local_udh(const HScalarArgs* a, hipStream_t stream)
{
    HScalarArgs A = *a;
    hipLaunchKernelGGL(spdh_local_udh, dim3((A.n_probs + 3) / 4), dim3(64), 0, stream, A);
    return hipGetLastError();
}

extern "C" hipError_t spdh_launch_exact(int udh, const HScalarArgs* a, hipStream_t stream)
{
    HScalarArgs A = *a;
    // groups per wave: one while the launch has fewer problems than ~8 waves per CU would hold, then two, then four
    int g = A.n_probs <= 8192 ? 1 : (A.n_probs <= 16384 ? 2 : 4);
    if (const char* e = getenv("SPDP_HX_GROUPS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) g = v; }
    const dim3 grd((A.n_probs + g - 1) / g), blk(16 * g);
#define SPDH_EXACT_GO(U, GG) hipLaunchKernelGGL((spdh_exact<U, GG>), grd, blk, 0, stream, A)
    if (udh) { if (g == 1) SPDH_EXACT_GO(true, 1); else if (g == 2) SPDH_EXACT_GO(true, 2); else SPDH_EXACT_GO(true, 4); }
    else     { if (g == 1) SPDH_EXACT_GO(false, 1); else if (g == 2) SPDH_EXACT_GO(false, 2); else SPDH_EXACT_GO(false, 4); }
#undef SPDH_EXACT_GO
    return hipGetLastError();
}
