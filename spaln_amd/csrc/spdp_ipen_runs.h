// spdp_ipen_runs.h -- IntPen(len) beyond the 4096 lengths the -A0 / -A1 kernels keep in LDS as they are.
// The table (IntronPenalty::Penalty, src/codepot.cc) is flat there but for a step now and then (62 steps up to
// 29 k nt with the default parameters), so the kernels keep the steps instead of going to memory for every long
// intron: run starts, run values and, per 64 lengths, the run its first length lies in.  Valid when the table ends
// below 65536, has at most 255 steps beyond 4096 and no two inside one span of 64 (spdp_intpen_runs says so).
#ifndef SPDP_IPEN_RUNS_H
#define SPDP_IPEN_RUNS_H
#include <stdint.h>

#define SPDP_IPR_BASE 4096
#define SPDP_IPR_RUNS 256
#define SPDP_IPR_SPANS 960          // (65536 - 4096) / 64
#define SPDP_IPR_WORDS (SPDP_IPR_RUNS + 1 + SPDP_IPR_RUNS + SPDP_IPR_SPANS / 2)      // uint16 starts, int16 values, uint8 spans
bool spdp_intpen_runs(const int16_t* intpen, int len, int16_t* out);                 // spdp_api.cpp; false: not representable

#ifdef __HIPCC__
struct IpenRuns {
    unsigned short start[SPDP_IPR_RUNS + 1];
    short val[SPDP_IPR_RUNS];
    unsigned char span[SPDP_IPR_SPANS];
};
// by all threads of the block, before a barrier; src = SPDP_IPR_WORDS words or null
__device__ __forceinline__ void ipen_runs_load(IpenRuns& R, const int16_t* src)
{
    if (!src) return;
    const uint16_t* st = reinterpret_cast<const uint16_t*>(src);
    const uint8_t* sp = reinterpret_cast<const uint8_t*>(src + SPDP_IPR_RUNS + 1 + SPDP_IPR_RUNS);
    for (int i = threadIdx.x; i <= SPDP_IPR_RUNS; i += blockDim.x) R.start[i] = st[i];
    for (int i = threadIdx.x; i < SPDP_IPR_RUNS; i += blockDim.x) R.val[i] = src[SPDP_IPR_RUNS + 1 + i];
    for (int i = threadIdx.x; i < SPDP_IPR_SPANS; i += blockDim.x) R.span[i] = sp[i];
}
// IntPen(len) for len >= SPDP_IPR_BASE
__device__ __forceinline__ int ipen_runs_get(const IpenRuns& R, int len, int intpen_len)
{
    const int l = max(min(len, intpen_len - 1), SPDP_IPR_BASE);
    int j = R.span[(l - SPDP_IPR_BASE) >> 6];
    j += l >= (int) R.start[j + 1];
    return R.val[min(j, SPDP_IPR_RUNS - 1)];        // (a table with 255 steps that ends at 65535: the sentinel is not a run)
}
#endif
#endif
