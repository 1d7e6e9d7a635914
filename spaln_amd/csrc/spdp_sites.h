// spdp_sites.h -- the splice-site code of the -A0 wavefront kernels (spdp_rowwave.hip): the per-row list of donor
// candidates and what an acceptor does with it, written so that a step of the wave is ONE straight run of
// instructions (round 5; DESIGN.md section 6f has the measurement that asked for it).
//
// What it restates (ogotoh/spaln v3.0.7): the candidate bookkeeping of Aln2s1::forwardS_ng / scorealoneS_ng /
// hirschbergS_ng -- src/fwd2s1.cc:330-372 (acceptor: every record of the row's list is priced with
// IntPen(n - jnc) + the junction-pair score and may raise the state it left from), :386-421 (donor: the states of
// the cell enter the list, kept sorted by value, NCAND + 1 slots, a full list whose last kept entry holds against
// the newcomer is cut back to NCAND).
//
// Layout (ours).  Lane = query row, so a lane keeps its row's list in registers:
//   v[l], j[l], q[l]   value, donor column, {state it left from | dinc5 << 5 | slot << 12}, SORTED (l = 0 best);
//                      a free position holds v = DEADV (below every score) and j = BIGJ (so that its intron length
//                      is negative): no count is kept, validity is in the values
//   r[f][s]            riders (Vmf pointer; or upr / lwr / ml / ulk of the linear-space engine), by SLOT: a slot
//                      number stays with its entry while the sorted part shifts, so a rider is written once, when
//                      the entry is made, and read when the entry wins an acceptor
// Pricing reads IntPen from LDS by one clamped look-up per table (the first 4096 lengths as they are, the rest as
// {value, first length of the next value, that value} per span of 64: spdp_ipen_runs.h guarantees at most one step
// per span) and selects; there is no per-lane branch between the top of a step and its end.
#ifndef SPDP_SITES_H
#define SPDP_SITES_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_internal.h"

namespace sites {

constexpr int NC = 5;                                   // NCAND + 1 positions
constexpr int DEADV = INT32_MIN;                        // value of a free position: every score beats it
constexpr int BIGJ = 0x3fffffff;                        // column of a free position: n - BIGJ < llmt for every n
constexpr int IPEN_LDS = SPDP_IPR_BASE;                 // IntPen(len) for len < this lives in LDS as it is

// read-only tables every wave of a block uses
struct Tables {
    short mtx[32 * 32];
    short ipen[IPEN_LDS];
    short t53[256];
    int   span[SPDP_IPR_SPANS + 1];                     // per 64 lengths: value | first length of the next value << 16 (that value: the next span's)
};
__device__ __forceinline__ void load_tables(Tables& T, const ScalarArgs& A, const DevScoring* sc)
{
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) T.mtx[i] = (short) sc->mtx[i];
    for (int i = threadIdx.x; i < IPEN_LDS; i += blockDim.x) T.ipen[i] = A.intpen[min(i, A.intpen_len - 1)];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) T.t53[i] = A.t53[i];
    if (A.ipen_runs) {
        const uint16_t* st = reinterpret_cast<const uint16_t*>(A.ipen_runs);
        const int16_t* val = A.ipen_runs + SPDP_IPR_RUNS + 1;
        const uint8_t* sp = reinterpret_cast<const uint8_t*>(A.ipen_runs + SPDP_IPR_RUNS + 1 + SPDP_IPR_RUNS);
        for (int i = threadIdx.x; i <= SPDP_IPR_SPANS; i += blockDim.x) {
            const int j = sp[min(i, SPDP_IPR_SPANS - 1)];
            const int nxt = st[j + 1];                  // 65535: none
            const int thr = nxt < IPEN_LDS + 64 * (i + 1) ? nxt : 65535;
            // (at most one step per span: the value behind it is what the next span starts with; the entry behind the last
            //  span repeats the last value)
            T.span[i] = ((int) val[i < SPDP_IPR_SPANS ? j : min(j + 1, SPDP_IPR_RUNS - 1)] & 0xffff) | (thr << 16);
        }
    }
    __syncthreads();                                    // the only block-wide barrier: every wave reaches it
}
// IntPen(len), any len (a negative one reads entry 0: its candidate is discarded by the caller)
template <bool RUNS>
__device__ __forceinline__ int intpen_of(const Tables& T, const ScalarArgs& A, int len)
{
    if constexpr (RUNS) {
        const int lc = max(0, min(len, A.intpen_len - 1));
        const int ip1 = T.ipen[min(lc, IPEN_LDS - 1)];
        const int si = (unsigned) max(lc - IPEN_LDS, 0) >> 6;
        const int w0 = T.span[si], w1 = T.span[si + 1];
        const int ip2 = (short) ((lc >= (int) ((unsigned) w0 >> 16) ? w1 : w0) & 0xffff);
        return lc < IPEN_LDS ? ip1 : ip2;
    } else                                              // a table the spans cannot hold: every length goes to memory
        return A.intpen[max(0, min(len, A.intpen_len - 1))];
}
// keeps a value (and the loads it comes from) where it stands: without it the compiler sinks the look-ups of a candidate
// into an exec-masked region of its own and a step becomes a chain of short branches again
#define SPDP_PIN(x) asm volatile("" : "+v"(x))

// the list of one row.  NR riders; GE: a newcomer passes entries of its own value (scorealoneS_ng's `>=`,
// src/fwd2s1.cc:1311; forwardS_ng and hirschbergS_ng insert behind them, :405 / :1042)
template <int NR, bool GE>
struct Cands {
    int v[NC], j[NC], q[NC];
    int r[NR > 0 ? NR : 1][NC];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int l = 0; l < NC; ++l) { v[l] = DEADV; j[l] = BIGJ; q[l] = l << 12; }
#pragma unroll
        for (int f = 0; f < (NR > 0 ? NR : 1); ++f)
#pragma unroll
            for (int l = 0; l < NC; ++l) r[f][l] = 0;
    }
    __device__ __forceinline__ bool any() const { return v[0] != DEADV; }
    // one state of one cell asks for a place (src/fwd2s1.cc:397-420): `t` per lane; x its value, n the column, k the
    // state, dn5 the donor's dinucleotide class, rid the riders.  Returns true where the entry was made.
    __device__ __forceinline__ bool insert(bool t, int x, int n, int k, int dn5, const int (&rid)[NR > 0 ? NR : 1])
    {
        // a full list whose last kept entry holds against the newcomer: the list is NC - 1 long afterwards
        const bool weak = t && (GE ? x < v[NC - 2] : x <= v[NC - 2]);       // (a free v[NC - 2] never holds)
        v[NC - 1] = weak ? DEADV : v[NC - 1];
        j[NC - 1] = weak ? BIGJ : j[NC - 1];
        const bool ins = t && !weak;
        bool c[NC];                                     // x goes in front of position l
#pragma unroll
        for (int l = 0; l < NC - 1; ++l) c[l] = ins && (GE ? x >= v[l] : x > v[l]);
        c[NC - 1] = ins;                                // the last position is free, or falls off the list
        const int nq = k | (dn5 << 5) | (q[NC - 1] & (7 << 12));            // the slot of whoever leaves
        const int slot = (q[NC - 1] >> 12) & 7;
#pragma unroll
        for (int l = NC - 1; l >= 1; --l) {
            v[l] = c[l - 1] ? v[l - 1] : (c[l] ? x : v[l]);
            j[l] = c[l - 1] ? j[l - 1] : (c[l] ? n : j[l]);
            q[l] = c[l - 1] ? q[l - 1] : (c[l] ? nq : q[l]);
        }
        v[0] = c[0] ? x : v[0]; j[0] = c[0] ? n : j[0]; q[0] = c[0] ? nq : q[0];
        if constexpr (NR > 0) {
#pragma unroll
            for (int s = 0; s < NC; ++s) {
                const bool w = ins && slot == s;
#pragma unroll
                for (int f = 0; f < NR; ++f) r[f][s] = w ? rid[f] : r[f][s];
            }
        }
        return ins;
    }
    // rider f of the entry whose q word is `qw`
    __device__ __forceinline__ int rider(int f, int qw) const
    {
        const int s = (qw >> 12) & 7;
        int x = r[f][0];
#pragma unroll
        for (int l = 1; l < NC; ++l) { x = s == l ? r[f][l] : x; SPDP_PIN(x); }      // (pinned: a chain of selects, not an indexed load from a copy of the list in scratch memory)
        return x;
    }
};

// prices every entry of the list at an acceptor (src/fwd2s1.cc:337-345): x[l] = DEADV where the lane is not on an
// acceptor, the position is free or the intron would be shorter than llmt.  All look-ups of the five entries are in
// flight together: addresses, then the loads, then the arithmetic (one LDS round trip per acceptor, not five).
template <bool RUNS, int NR, bool GE>
__device__ __forceinline__ void price(const Cands<NR, GE>& C, const Tables& T, const ScalarArgs& A, bool acc, int n, int base,
                                      int dn3, int llmt, int (&x)[NC])
{
    int ip[NC], jp[NC];
    if constexpr (RUNS) {
        const int top = A.intpen_len - 1;
        // (two rounds, three entries and two: the look-ups of a round are in flight together, and a round's temporaries are
        //  gone before the next one's exist -- the registers of a step decide how many waves a SIMD holds)
#pragma unroll
        for (int l0 = 0; l0 < NC; l0 += 3) {
            int lc[3], w0[3], w1[3];
#pragma unroll
            for (int l = l0; l < min(NC, l0 + 3); ++l) lc[l - l0] = max(0, min(n - C.j[l], top));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = l0; l < min(NC, l0 + 3); ++l) {
                ip[l] = T.ipen[min(lc[l - l0], IPEN_LDS - 1)];
                const int si = (unsigned) max(lc[l - l0] - IPEN_LDS, 0) >> 6;
                w0[l - l0] = T.span[si]; w1[l - l0] = T.span[si + 1];
                jp[l] = *reinterpret_cast<const short*>(reinterpret_cast<const char*>(T.t53) + ((C.q[l] & (15 << 5)) | (dn3 << 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = l0; l < min(NC, l0 + 3); ++l) {
                const int ip2 = (short) ((lc[l - l0] >= (int) ((unsigned) w0[l - l0] >> 16) ? w1[l - l0] : w0[l - l0]) & 0xffff);
                ip[l] = (lc[l - l0] < IPEN_LDS ? ip[l] : ip2) + jp[l];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int l = 0; l < NC; ++l) {
            ip[l] = intpen_of<false>(T, A, n - C.j[l]);
            jp[l] = *reinterpret_cast<const short*>(reinterpret_cast<const char*>(T.t53) + ((C.q[l] & (15 << 5)) | (dn3 << 1)));
            ip[l] += jp[l];
        }
    }
#pragma unroll
    for (int l = 0; l < NC; ++l) {
        int y = C.v[l] + ip[l] + base;
        SPDP_PIN(y);
        x[l] = (acc && n - C.j[l] >= llmt) ? y : DEADV;
    }
    __builtin_amdgcn_sched_barrier(0);
}

}   // namespace sites
#endif
