// spdp_signals.hip -- the splice-signal precompute on the device.
//
// What the reference does on the CPU before every alignment (Exinon::intron53_c + intron53_n, src/codepot.cc:435-520,
// with PatMat::calcPatMat, src/utilseq.cc:905-1000, for the two second-order Markov position weight matrices of the
// species' splice-site model): per genomic position n
//     dinc5 / dinc3 / cano5 / cano3   class of the dinucleotide after / before n and whether it is a canonical site
//     sig5[n] = (short)(fs * P5(n)) + tab5[dinc5[n]],  sig3[n] = (short)(fs * P3(n)) + tab3[dinc3[n]]
// P(n) = tonic + the sum over the `cols` columns of the matrix, column m scored by the trinucleotide at
// n - offset + m (the first column also by its mono- and dinucleotide terms) -- float additions in that order, which
// this kernel keeps (no reassociation, no fma: the file is built with -ffp-contract=off), so that the truncated
// shorts are the reference's bit for bit.
//
// Mapping: one thread per position, 16 chunks of 256 positions per block; the two matrices (2 x 84 x 24 floats = 16 KiB) and the
// block's window of reduced codes (256 + halo) live in LDS.  Per position 2 x (cols + 2) dependent LDS reads and
// float adds: HBM traffic is the 1 B/position read and the 8 (+2) B/position column records written -- the records
// the sweeps read (spdp_dev.h `cols`, `aux`), so a batch uploaded as plain codes never carries its signals over PCIe.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_internal.h"

#define SIG_TPB   256
#define SIG_CHUNKS 16           // 256-position chunks a block walks through with one copy of the matrices
#define SIG_HALO  64            // >= max(offset, cols - offset) + 2 of either matrix (checked by the launcher)

__device__ __forceinline__ int red_strict(int code)      // ncredctab: A C G T = 2 3 5 9 -> 0..3, everything else "bad"
{
    return code == 2 ? 0 : code == 3 ? 1 : code == 5 ? 2 : code == 9 ? 3 : 4;
}

// calcPatMat's value at sequence position `pos`; x(i) = strict reduced code of base i (0 <= i < len)
template <class X>
__device__ __forceinline__ float pat_scan(const float* __restrict__ mtx, int rows, int cols, int offset, float tonic,
                                          float min_elem, int pos, int len, X x)
{
    int n = pos - offset;
    int col = 0;
    int q = (n + cols >= len) ? 1 : 0;
    if (n < 0) { col = -n; n = 0; }                      // columns with no base under them are skipped
    const int last = min(n + (cols - col), len - 2);
    float fit = 0.f;
    bool first = true;
    for (int s = n; s < last; ++s, ++col) {
        const float* row = mtx + col * rows;
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
    if (q) fit = (float) cols * min_elem;
    return fit + tonic;
}

__global__ __launch_bounds__(SIG_TPB)
void spdp_signals(SignalArgs A)
{
    extern __shared__ float s_mtx[];                     // mtx5 then mtx3
    __shared__ uint8_t s_x[SIG_TPB + 2 * SIG_HALO];      // codes of bases p0 - HALO .. p0 + TPB + HALO

    const SigJob J = A.jobs[blockIdx.y];
    if ((int) blockIdx.x * SIG_TPB * SIG_CHUNKS > J.b_len) return;
    const SigModelDev& M = *A.model;
    const int n5 = M.rows * M.cols5, n3 = M.rows * M.cols3;
    // the matrices (16 KB) once per block, for SIG_CHUNKS x 256 positions (one chunk per block made the kernel a copy of
    // matrices: 64 B of L2 -> LDS traffic per position)
    for (int i = threadIdx.x; i < n5; i += SIG_TPB) s_mtx[i] = A.mtx5[i];
    for (int i = threadIdx.x; i < n3; i += SIG_TPB) s_mtx[n5 + i] = A.mtx3[i];
    const uint8_t* __restrict__ codes = A.codes + J.b_off;
    int m5 = INT32_MIN, m3 = INT32_MIN;
  for (int chunk = 0; chunk < SIG_CHUNKS; ++chunk) {
    const int p0 = ((int) blockIdx.x * SIG_CHUNKS + chunk) * SIG_TPB;
    if (p0 > J.b_len) break;                                 // (block-uniform)
    __syncthreads();                                         // the previous chunk's readers are done with s_x
    for (int i = threadIdx.x; i < SIG_TPB + 2 * SIG_HALO; i += SIG_TPB) {
        const int g = p0 - SIG_HALO + i;
        s_x[i] = (g >= 0 && g < J.b_len) ? codes[g] : 0;
    }
    __syncthreads();
    const int pos = p0 + (int) threadIdx.x;
    int v5 = 0, v3 = 0, d5 = 0, d3 = 0, c5 = 0, c3 = 0;
    if (pos <= J.b_len) {
        const int base = p0 - SIG_HALO;
        auto raw = [&](int i) { return (int) s_x[i - base]; };
        auto xs = [&](int i) { return red_strict(raw(i)); };
        auto xc = [&](int i) { const int c = red_strict(raw(i)); return c > 3 ? 1 : c; };     // classes: bad -> 'C'
        // class of the dinucleotide ending at base i of the range [left, right): the chain starts from 'C'
        auto nc = [&](int i) { return (((i == J.left ? 1 : xc(i - 1)) << 2) + xc(i)) & 0xf; };
        const int any = M.any & 3;
        static const uint8_t jac[4] = {0, 2, 3, 1}, jgt[4] = {0, 0, 3, 1};
        const int ac = jac[any], gt = jgt[any], dflt = any == 3 ? 1 : 0;
        if (pos + 1 >= J.left && pos + 1 < J.right) {            // base pos + 1 writes the donor cell of pos
            d5 = nc(pos + 1);
            c5 = d5 == 3 ? 2 : (d5 == 9 || d5 == 11) ? 3 : (d5 == 7 || d5 == 8 || d5 == 10 || d5 == 15) ? gt : dflt;
            if (M.both_ori && d5 == 1) c5 = 1;
        }
        if (pos - 1 >= J.left && pos - 1 < J.right) {            // base pos - 1 writes the acceptor cell of pos
            d3 = nc(pos - 1);
            c3 = d3 == 1 ? 2 : d3 == 2 ? 3 : (d3 == 0 || d3 == 3) ? ac : (d3 == 6 || d3 == 10 || d3 == 14) ? gt : dflt;
            if (M.both_ori && (d3 == 7 || d3 == 11)) c3 = 1;
        }
        if (pos >= J.left && pos < J.right) {
            const float f5 = pat_scan(s_mtx, M.rows, M.cols5, M.off5, M.tonic5, M.min5, pos, J.b_len, xs);
            const float f3 = pat_scan(s_mtx + n5, M.rows, M.cols3, M.off3, M.tonic3, M.min3, pos, J.b_len, xs);
            v5 = (int16_t) ((int16_t) (int) (M.fs * f5) + M.tab5[d5]);
            v3 = (int16_t) ((int16_t) (int) (M.fs * f3) + M.tab3[d3]);
        }
        if (A.sig5) {                                            // plain arrays (spdp_splice_signals)
            const int64_t o = J.out_off + pos;
            A.sig5[o] = (int16_t) v5; A.sig3[o] = (int16_t) v3;
            A.cano5[o] = (uint8_t) c5; A.cano3[o] = (uint8_t) c3; A.dinc[o] = (uint8_t) (d5 << 4 | d3);
        }
        if (A.cols) {                                            // the sweeps' column records (spdp_dev.h)
            const uint16_t s5 = (uint16_t) (int16_t) (v5 + A.ipen);
            const uint32_t pack = A.spj ? ((uint32_t) s5 | ((uint32_t) (uint16_t) v3 << 16)) : 0u;
            A.cols[J.col_off + pos] = make_int2((int) pack, pos > 0 ? raw(pos - 1) : 0);
            if (A.aux) A.aux[J.col_off + pos] = make_uchar2((uint8_t) ((c5 ? 1 : 0) | (c3 ? 2 : 0)), (uint8_t) (d5 << 4 | d3));
            v5 = (int16_t) s5;
        }
    } else {
        v5 = v3 = INT32_MIN;
    }
    m5 = max(m5, v5); m3 = max(m3, v3);
  }
    if (A.maxes) {                                               // bounds for the fp32 sweeps' range guard
        for (int o = 32; o; o >>= 1) { m5 = max(m5, __shfl_xor(m5, o)); m3 = max(m3, __shfl_xor(m3, o)); }
        if ((threadIdx.x & 63) == 0) { atomicMax(A.maxes, m5); atomicMax(A.maxes + 1, m3); }
    }
}

extern "C" hipError_t spdp_launch_signals(const SignalArgs* a, int n_jobs, int max_len, int lds_floats, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned) ((max_len + 1 + SIG_TPB * SIG_CHUNKS - 1) / (SIG_TPB * SIG_CHUNKS)), (unsigned) n_jobs);
    hipLaunchKernelGGL(spdp_signals, grid, dim3(SIG_TPB), (size_t) lds_floats * sizeof(float), s, *a);
    return hipGetLastError();
}
