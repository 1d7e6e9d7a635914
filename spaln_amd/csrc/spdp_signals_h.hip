// spdp_signals_h.hip -- the protein-side signal precompute on the device: Exinon::intron53_c +
// intron53_p (src/codepot.cc:435-476, 524-611) for a tron genomic window.
//
//   spdh_signals       one thread per position: the position weight matrix scans of PatMat::calcPatMat (src/utilseq.cc:
//                      905-1000: Markov order 2 for the splice sites and the stop context, order <= 1 for the start
//                      context), the 5th-order coding potential of ExinPot::calcScr_3 (:1423-1460) with the stop-codon
//                      rules, the dinucleotide classes and canonical-site levels: sig5, sig3, sigS, sigT, sigE, dinc, cano
//   spdh_signal_phases one thread per window: the phase arrays phs5 / phs3 (a sequential rule: a canonical site marks its
//                      neighbours, and what it writes depends on what the previous site wrote)
//   spdh_signal_pack   one thread per position: the int4 column records + short4 raw signals the protein engines read
//                      (layout: spdp_h_dev.h; the same packing HStore::upload does on the host for supplied arrays)
// The tron sequence is read through tnredctab (src/seq.cc:41: tron -> the middle base of its codon).  Float additions in
// the reference's order (built with -ffp-contract=off).  The coding-potential hash of a position depends only on the run
// of good bases it sits in (it restarts at the range start and after every base that is not A C G T), so it is
// recomputed per position from at most eight bases instead of being carried along the sequence.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

#define SGH_TPB  256
#define SGH_HALO 64

__device__ __constant__ unsigned char sgh_tnred[32] = {4, 4, 4, 1, 2, 0, 0, 2, 0, 0, 2, 0, 3, 3, 0, 3, 3, 1, 1, 1, 2, 0, 3, 2, 2, 0, 4, 4, 4, 4, 4, 4};

template <class X>
__device__ __forceinline__ float pm_scan(const SigPatMatDev& pm, const float* __restrict__ mtx, int pos, int len, X x)
{
    int n = pos - pm.offset;
    int col = 0;
    int q = (n + pm.cols >= len) ? 1 : 0;
    if (n < 0) { col = -n; n = 0; }
    const int last = min(n + (pm.cols - col), len - pm.order);
    float fit = 0.f;
    bool first = true;
    if (pm.order == 2) {
        for (int s = n; s < last; ++s, ++col) {
            const float* row = mtx + col * pm.rows;
            const int i0 = x(s), i1 = x(s + 1), i2 = x(s + 2);
            int k = i0;
            if (i0 > 3) ++q;
            if (first && q == 0) fit += row[k];
            if (i1 > 3) ++q;
            else if (q == 0) { k = 4 * k + i1; if (first) fit += row[k + 4]; }
            if (i2 > 3) ++q;
            else if (q == 0) { k = 4 * k + i2; fit += row[k + 20]; }
            first = false;
        }
        if (q) fit = (float) pm.cols * pm.min_elem;
    } else {
        for (int s = n; s < last; ++s, ++col) {          // a bad base ends the sum: the remaining columns add nothing
            const float* row = mtx + col * pm.rows;
            int k = x(s);
            if (k > 3) ++q;
            if (pm.order && !q) {
                if (first) fit += row[k];
                const int j = x(s + 1);
                if (j > 3) ++q;
                k = 4 * k + j + 4;
            }
            if (!q) fit += row[k];
            first = false;
        }
    }
    return fit + pm.tonic;
}

__global__ __launch_bounds__(SGH_TPB)
void spdh_signals(SignalArgsH A)
{
    extern __shared__ float s_mtx[];                     // pm5, pm3, pmI, pmT, pmB back to back
    __shared__ uint8_t s_b[SGH_TPB + 2 * SGH_HALO];      // tron codes of positions p0 - HALO ..
    const SigJobH J = A.jobs[blockIdx.y];
    const int p0 = blockIdx.x * SGH_TPB;
    const int N = J.b_len + 3;
    if (p0 >= N) return;
    const SigModelHDev& M = *A.model;
    const int o5 = 0, o3 = M.pm5.rows * M.pm5.cols, oI = o3 + M.pm3.rows * M.pm3.cols, oT = oI + M.pmI.rows * M.pmI.cols,
              oB = oT + M.pmT.rows * M.pmT.cols, tot = oB + M.pmB.rows * M.pmB.cols;
    for (int i = threadIdx.x; i < tot; i += SGH_TPB) s_mtx[i] = A.mtx[i];
    const uint8_t* __restrict__ codes = A.codes + J.b_off;
    for (int i = threadIdx.x; i < SGH_TPB + 2 * SGH_HALO; i += SGH_TPB) {
        const int g = p0 - SGH_HALO + i;
        s_b[i] = (g >= 0 && g < J.b_len) ? codes[g] : 0;
    }
    __syncthreads();
    const int pos = p0 + (int) threadIdx.x;
    if (pos >= N) return;
    const int base = p0 - SGH_HALO;
    auto tron = [&](int i) { return (int) s_b[i - base]; };
    auto xs = [&](int i) { return (int) sgh_tnred[tron(i) & 31]; };
    auto xc = [&](int i) { const int c = xs(i); return c > 3 ? 1 : c; };
    auto nc = [&](int i) { return (((i == J.left ? 1 : xc(i - 1)) << 2) + xc(i)) & 0xf; };
    int v5 = 0, v3 = 0, vS = 0, vT = 0, vE = 0, d5 = 0, d3 = 0, c5 = 0, c3 = 0, vB = INT32_MIN;
    const int any = M.any & 3;
    const int jac[4] = {0, 2, 3, 1}, jgt[4] = {0, 0, 3, 1};
    const int ac = jac[any], gt = jgt[any], dflt = any == 3 ? 1 : 0;
    if (pos + 1 >= J.left && pos + 1 < J.right) {
        d5 = nc(pos + 1);
        c5 = d5 == 3 ? 2 : (d5 == 9 || d5 == 11) ? 3 : (d5 == 7 || d5 == 8 || d5 == 10 || d5 == 15) ? gt : dflt;
    }
    if (pos - 1 >= J.left && pos - 1 < J.right) {
        d3 = nc(pos - 1);
        c3 = d3 == 1 ? 2 : d3 == 2 ? 3 : (d3 == 0 || d3 == 3) ? ac : (d3 == 6 || d3 == 10 || d3 == 14) ? gt : dflt;
    }
    if (pos >= J.left && pos < J.right) {
        const int len = J.b_len;
        if (M.dvsp && M.pmI.rows) vS = (int16_t) (int) (M.fT * pm_scan(M.pmI, s_mtx + oI, pos, len, xs));
        if (M.dvsp && M.pmT.rows) vT = (int16_t) (int) (M.fT * pm_scan(M.pmT, s_mtx + oT, pos, len, xs));
        if (M.pot_ndata) {
            // calcScr_3: the value of position pos is what the rolling loop computes at pos + 5, inside [start, stop)
            const int start = max(J.left - 1, 0), stop = min(J.right + 1, len);
            float val = 0.f;
            bool six = pos + 5 < stop;
            for (int i = 0; i < 6 && six; ++i) six = xs(pos + i) < 4;
            if (six) {
                const bool g1 = pos - 1 >= start && xs(pos - 1) < 4;          // does the run of good bases reach further back
                const bool g2 = g1 && pos - 2 >= start && xs(pos - 2) < 4;
                int w0 = 0, w1 = 0, w2 = 0;                                   // hashes ending at pos + 5, + 4, + 3
                for (int i = 0; i < 6; ++i) w0 = 4 * w0 + xs(pos + i);
                for (int i = g1 ? -1 : 0; i < 5; ++i) w1 = 4 * w1 + xs(pos + i);
                for (int i = g2 ? -2 : (g1 ? -1 : 0); i < 4; ++i) w2 = 4 * w2 + xs(pos + i);
                const int nd = M.pot_ndata;
                val += A.pot[3 * (w2 % nd) + 2];
                val += A.pot[3 * (w1 % nd)];
                val += A.pot[3 * (w0 % nd) + 1];
            }
            float e = M.fE * val;
            const int t0 = tron(pos);
            if (t0 == M.trm || t0 == M.trm2) e += M.fO;
            else if (pos + 3 < J.right && pos + 3 < len) { const int t3 = tron(pos + 3); if (t3 == M.trm || t3 == M.trm2) e = 0.f; }
            vE = (int16_t) (int) e;
        }
        if (M.pmB.rows) {                                // a branch site stronger than the threshold (:588-591)
            const float sb = pm_scan(M.pmB, s_mtx + oB, pos, len, xs);
            if (sb > M.thB) vB = (int16_t) (int) (M.fB * sb);
        }
        const float f5 = pm_scan(M.pm5, s_mtx + o5, pos, len, xs);
        const float f3 = pm_scan(M.pm3, s_mtx + o3, pos, len, xs);
        v5 = (int16_t) ((int16_t) (int) (M.fs * f5) + M.tab5[d5]);
        v3 = (int16_t) ((int16_t) (int) (M.fs * f3) + M.tab3[d3]);
    }
    const int64_t o = J.out_off + pos;
    A.sig5[o] = (int16_t) v5; A.sig3[o] = (int16_t) v3; A.sigS[o] = (int16_t) vS; A.sigT[o] = (int16_t) vT; A.sigE[o] = (int16_t) vE;
    if (A.sb) A.sb[o] = vB;
    A.cano[o] = (uint8_t) (c5 | c3 << 4);
    A.dinc[o] = (uint8_t) (d5 << 4 | d3);
}

__global__ void spdh_signal_phases(SignalArgsH A, int n_jobs)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_jobs) return;
    const SigJobH J = A.jobs[j];
    const SigModelHDev& M = *A.model;
    const int N = J.b_len + 3;
    int8_t* p5 = A.phs5 + J.out_off;
    int8_t* p3 = A.phs3 + J.out_off;
    const int16_t* s5 = A.sig5 + J.out_off;
    int16_t* s3 = A.sig3 + J.out_off;
    const uint8_t* cn = A.cano + J.out_off;
    const int32_t* sb = A.sb ? A.sb + J.out_off : nullptr;
    int sigB = 0, posB = -1;                             // the branch-point carry (src/codepot.cc:567-568, 586-597)
    for (int i = 0; i < N; ++i) { p5[i] = -2; p3[i] = -2; }
    const int th5 = (int16_t) (int) (M.fS * M.tonic5), th3 = (int16_t) (int) (M.fS * M.tonic3);
    for (int pos = J.left; pos < J.right; ++pos) {
        const int c5 = cn[pos] & 15, c3 = cn[pos] >> 4;
        if (sb) {
            s3[pos] = (int16_t) (s3[pos] + sigB);
            if (sb[pos] != INT32_MIN) { sigB = sb[pos]; posB = pos; }
            if (posB >= 0 && pos - posB > M.maxb3d) { sigB = 0; posB = -1; }
        }
        if (p5[pos] == -2 && ((M.any == 2 && s5[pos] > th5) || c5)) {
            p5[pos] = 0;
            if (c5 > 1) { p5[pos + 1] = 1; if (pos >= 1) p5[pos - 1] = (p5[pos - 1] == 1) ? 2 : -1; }      // GTGT
        }
        if (p3[pos] == -2 && ((M.any == 2 && s3[pos] > th3) || c3)) {
            p3[pos] = 0;
            if (c3 > 1) { p3[pos + 1] = 1; if (pos >= 1) p3[pos - 1] = (p3[pos - 1] == 1) ? 2 : -1; }      // AGAG
        }
    }
}

__global__ __launch_bounds__(SGH_TPB)
void spdh_signal_pack(SignalArgsH A)
{
    const SigJobH J = A.jobs[blockIdx.y];
    const int x = blockIdx.x * SGH_TPB + (int) threadIdx.x;
    const int N = J.b_len + 3;
    if (x >= N) return;
    const int16_t* s5 = A.sig5 + J.out_off;
    const int16_t* s3 = A.sig3 + J.out_off;
    const int16_t* sE = A.sigE + J.out_off;
    const int8_t* p5 = A.phs5 + J.out_off;
    const int8_t* p3 = A.phs3 + J.out_off;
    const uint8_t* codes = A.codes + J.b_off;
    auto good = [&](int i) { return J.left - 1 <= i && i < J.right; };
    auto at = [&](const int16_t* v, int i) -> int { return (i >= 0 && i < N) ? v[i] : 0; };
    const int cp = (x - 2 >= 0 && good(x - 2)) ? sE[x - 2] : 0;
    const int tron = (x - 2 >= 0 && x - 2 <= J.b_len) ? codes[x - 2] : 0;
    unsigned fl = 0;
    int s3_0 = SPDH_MIN_SSV, s3_1 = SPDH_MIN_SSV, s5_0 = SPDH_MIN_SSV, s5_1 = SPDH_MIN_SSV;
    const int ph3 = p3[x], ph5 = p5[x];
    if (ph3 > -2) {
        const int phase = (ph3 == 2) ? -1 : ph3;
        fl |= (unsigned) (phase + 2);
        s3_0 = at(s3, x - phase);
        if (ph3 == 2) { fl |= 4u; s3_1 = at(s3, x - 1); }
    }
    if (ph5 > -2) {
        const int phase = (ph5 == 2) ? -1 : ph5;
        fl |= (unsigned) (phase + 2) << 3;
        s5_0 = (int16_t) (at(s5, x - phase) + A.ipen);
        if (ph5 == 2) { fl |= 32u; s5_1 = (int16_t) (at(s5, x - 1) + A.ipen); }
    }
    int4 rec;
    rec.x = (int) ((unsigned) (uint16_t) (int16_t) cp | ((unsigned) (tron > 31 ? SPDH_ZCODE : tron) << 16) | (fl << 24));
    rec.y = (int) ((unsigned) (uint16_t) (int16_t) s3_0 | ((unsigned) (uint16_t) (int16_t) s3_1 << 16));
    rec.z = (int) ((unsigned) (uint16_t) (int16_t) s5_0 | ((unsigned) (uint16_t) (int16_t) s5_1 << 16));
    rec.w = (x <= J.b_len) ? (int) A.dinc[J.out_off + x] : 0;
    A.cols[J.col_off + x] = rec;
    A.aux[J.col_off + x] = make_short4(A.sigS[J.out_off + x], A.sigT[J.out_off + x], sE[x], s5[x]);
}

extern "C" hipError_t spdh_launch_signals(const SignalArgsH* a, int n_jobs, int max_len, int lds_floats, int pack, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    const dim3 grid((unsigned) ((max_len + 3 + SGH_TPB - 1) / SGH_TPB), (unsigned) n_jobs);
    hipLaunchKernelGGL(spdh_signals, grid, dim3(SGH_TPB), (size_t) lds_floats * sizeof(float), s, *a);
    hipLaunchKernelGGL(spdh_signal_phases, dim3((n_jobs + 63) / 64), dim3(64), 0, s, *a, n_jobs);
    if (pack) hipLaunchKernelGGL(spdh_signal_pack, grid, dim3(SGH_TPB), 0, s, *a);
    return hipGetLastError();
}
