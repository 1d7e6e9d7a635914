// spdp_pipe.h -- two helpers of the wavefront kernels whose tiles / stripes run as pipelines of waves
// (spdp_rowwave.hip, spdp_h_rowwave.hip): round 5.
#ifndef SPDP_PIPE_H
#define SPDP_PIPE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// a problem record every lane has read from the same address: its words, said to be wave-uniform, live in SGPRs -- and so
// does everything computed from them (ranges, array bases, loop bounds)
template <class T> __device__ __forceinline__ T wave_uniform(const T& t)
{
    static_assert(sizeof(T) % 4 == 0, "words");
    T r;
    const int* src = reinterpret_cast<const int*>(&t);
    int* dst = reinterpret_cast<int*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return r;
}

// The diagonal arrays of a pipelined problem cross CUs; the agent-scope atomic load the compiler emits for each entry is
// followed by a wait of its own -- a refill of ten planes was ten memory round trips in a row.  The same loads (sc1: the
// memory side, past the non-coherent L2s) issued together, one wait.  The compiler does not know these loads are
// asynchronous: nothing may look at v[] before the wait, which the empty statements behind it see to.
template <bool X, int N>
__device__ __forceinline__ void gld_n(const int* const (&base)[N], int e, int (&v)[N])      // v[i] = base[i][e], base[] wave-uniform
{
    if constexpr (X) {
        const unsigned off = (unsigned) e * 4u;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(v[i]) : "v"(off), "s"(base[i]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __builtin_nontemporal_load(base[i] + e);
    }
}

#endif
