// spdp_sweep_fp.hip -- second generation of the `_wip` sweeps (score-only and linear-space flavours,
// non-local ends): the recurrence of spdp_kernels.hip, laid out for how gfx950 actually issues VALU work
// (profiles/r02_valu_ubench.txt):
//
//   * a SIMD issues one VALU instruction per ~2.2 cycles, but a max / compare / 3-source form occupies
//     the integer side for 4 cycles unless an fp32 add / fma (or a plain move) issues beside it, and a
//     DPP form costs ~5 cycles and pairs with nothing;
//   * so scores are carried as fp32 (every value is an integer below 2^23 in magnitude: exact), which
//     puts all additions on the FP side in the shadow of the max / compare forms; links stay int32 bits
//     (they are only moved and selected);
//   * the vertical-gap candidate of a cell, max(F + gep, H + gop + gep), is formed by the lane that
//     owns F and H (the row above) right after its own cell -- it shares H + gop + gep with that lane's
//     own horizontal gap -- and travels down as one value (+ one link), so the boundary rows hold
//     {H, Fcand(, Hlink, Fcandlink)};
//   * "no acceptor yet" (hil <= llmt) is folded into the intron-penalty table: an entry is {A, C} and
//     the candidate is max(hv2 + sig3 + A, floor) + C with {A, C} = {-2^22, nevsel - floor} for the
//     short lengths, which yields exactly `nevsel` there, as the reference's blend does;
//   * the bottom row leaves the stripe through LDS: its lane writes the step's {H, Fcand, links} into slot `step` of a
//     16-entry block under a one-lane exec mask (a ds_write beside the VALU stream) and lane j reads slot j back at the
//     flush (round 2 used one 64-bit row_newbcast DPP per pair of values and step; the int kernels a rotate + shift per value).
//
// Reference recurrence: src/fwd2s1_wip_simd.h:97-202 (score-only), :555-758 (linear space),
// boundary set-up / end selection src/fwd2s1_simd.cc:163-262.  Geometry (16-row stripes = DPP rows, four
// stripes of a pass per wave, multi-wave and cross-CU pass pipelines, prefetch, feed / column rings) is
// that of spdp_kernels.hip; results are bit-identical to it (tests/test_gpu_fp_sweep.py runs both).
// Local ends (-LS), the traceback flavour, penalty tables beyond SPDP_FPEN_TAB entries and queries whose
// score could leave the exact fp32 range stay with spdp_kernels.hip.

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

namespace {

enum { FL_SCORE = 0, FL_FORWARD = 1, FL_UDH = 2 };

#define END_OF_ULK (INT32_MAX - 2)
#define SPDP_FPEN_TAB 992                       // entries of the {A, C} penalty table in LDS

constexpr float NEVF = (float) SPDP_NEV16;
constexpr float FLOORF = (float) SPDP_FLOOR16;
constexpr float BIGF = 4194304.f;               // 2^22: pushes a candidate below the floor, sums stay exact

__device__ __forceinline__ float as_f(int x) { return __int_as_float(x); }
__device__ __forceinline__ int as_i(float x) { return __float_as_int(x); }

// lane i of every 16-lane row <- lane i-1; lane 0 of the row keeps `old`
__device__ __forceinline__ int row_shr1(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ float row_shr1(float old, float src) { return as_f(row_shr1(as_i(old), as_i(src))); }

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int4 ld_nt4(const int* p)
{
    const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int2 ld_nt2(const int* p)
{
    const v2i_t v = __builtin_nontemporal_load(reinterpret_cast<const v2i_t*>(p));
    return make_int2(v.x, v.y);
}
// boundary entries: L1-bypassing inside one CU, memory-side (sc1) when a problem spans CUs -- see spdp_kernels.hip
template <bool X> __device__ __forceinline__ int ld_b1(const int* p)
{
    if constexpr (X) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __builtin_nontemporal_load(p);
}
template <bool X> __device__ __forceinline__ int4 ld_b4(const int* p)
{
    if constexpr (X) return make_int4(ld_b1<true>(p), ld_b1<true>(p + 1), ld_b1<true>(p + 2), ld_b1<true>(p + 3));
    else return ld_nt4(p);
}
template <bool X> __device__ __forceinline__ int2 ld_b2(const int* p)
{
    if constexpr (X) return make_int2(ld_b1<true>(p), ld_b1<true>(p + 1));
    else return ld_nt2(p);
}
// ... as plain (non-atomic) buffer loads with the same scope bit: the agent-scope ATOMIC load above is followed by a wait of
// its own, so the "prefetch" of a cross-CU pass was four memory round trips the wave sat out, one after the other, per
// block of sixteen steps (profiles/r04_wave_pmc.txt: 61 % of C5's resident wave cycles in s_waitcnt).  A buffer load with
// sc1 goes the same way to the memory side, stays in flight, and the compiler counts it (round 5).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t bnd_rsrc(const int* base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(base), 0, 0x7ffffffc, 0x00020000);
}
template <bool X> __device__ __forceinline__ int4 ldx_b4(rsrc_t r, const int* base, int64_t idx)
{
    if constexpr (X) { const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int) (idx * 4), 0, 16); return make_int4(v.x, v.y, v.z, v.w); }
    else return ld_nt4(base + idx);
}
template <bool X> __device__ __forceinline__ int2 ldx_b2(rsrc_t r, const int* base, int64_t idx)
{
    if constexpr (X) { const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int) (idx * 4), 0, 16); return make_int2(v.x, v.y); }
    else return ld_nt2(base + idx);
}
template <bool X> __device__ __forceinline__ void st_b1(int* p, int v)
{
    if constexpr (X) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool X> __device__ __forceinline__ void st_b4(int* p, int4 v)
{
    if constexpr (X) { st_b1<true>(p, v.x); st_b1<true>(p + 1, v.y); st_b1<true>(p + 2, v.z); st_b1<true>(p + 3, v.w); }
    else *reinterpret_cast<int4*>(p) = v;
}
template <bool X> __device__ __forceinline__ void st_b2(int* p, int2 v)
{
    if constexpr (X) { st_b1<true>(p, v.x); st_b1<true>(p + 1, v.y); }
    else *reinterpret_cast<int2*>(p) = v;
}

template <bool B> struct BoolTag { static constexpr bool value = B; };
#define WAVE_ORDER() asm volatile("" ::: "memory")
// keeps the read-ahead LDS loads of a step where they were written: without it the scheduler sinks them towards
// their use (fewer live registers) and the waves wait for LDS again
#ifndef SPDP_NO_PIN
#define SPDP_PIN_LOADS() __builtin_amdgcn_sched_barrier(0)
#else
#define SPDP_PIN_LOADS()
#endif

}  // namespace

// ---------------------------------------------------------------------------
// WPB / CROSS / work mapping / progress words: exactly as spdp_sweep (spdp_kernels.hip).
// SPJ: splice signals on (PwdB::DvsP != 0): with it off there is no donor / acceptor state at all.
template <int FL, int WPB, bool CROSS, bool SPJ>
__global__ __launch_bounds__(WPB * 64) void spdp_sweep_fp(SweepArgs A)
{
    // FL_FORWARD (round 4): the traceback flavour -- one code byte per cell in the layout spdp_walk reads (spdp_kernels.hip:
    // 256 bytes per block of 16 steps, 16 per lane) -- on this kernel's step; local ends stay with spdp_kernels.hip
    constexpr bool UDH = FL == FL_UDH;
    constexpr bool FWD = FL == FL_FORWARD;
    enum { TB_DIAG = 1, TB_HORI = 2, TB_VERT = 8, TB_ACCR = 14, TB_NHOR = 16, TB_NVER = 32, TB_DONR = 128 };      // as spdp_kernels.hip:36
    constexpr int BW = UDH ? 4 : 2;                     // dwords per boundary entry
    // LDS layout, bank by bank (64 banks of 4 B; ds_read_b32 sees 32): SQ_LDS_BANK_CONFLICT was 56 % of the LDS
    // cycles of the first version of this kernel, and the LDS array was busy 71 % of the time.
    //  * substitution matrix: residue codes are renumbered so that A C G T (2 3 5 9) become 0 1 2 3 and a row is
    //    36 floats: the 16 common (query, genome) pairs then sit in 16 different banks (rows of 32 put a whole
    //    column in ONE bank: up to 4 addresses per bank and instruction);
    //  * column records as two rings (signals 8 B, matrix column offset 4 B) instead of one 16-byte record: the 16
    //    lanes of a row read 16 consecutive slots = 32 (16) consecutive banks, and the next row of the wave, 16
    //    slots behind, the other half;
    //  * the feed of a row is padded by one entry, so the entries the rows of a wave broadcast in one
    //    instruction lie in different banks.
    __shared__ float  s_mtx[32 * 36];
    __shared__ int    s_perm[32];
    __shared__ float2 s_pen[SPDP_FPEN_TAB];
    // per wave, per DPP row: 2 x 48-slot rings of column records (each record written twice, 48 slots apart: any
    // 31-column window is contiguous) and the 16 boundary entries of the block
    __shared__ float2 s_sig[WPB][4][96];                // {sig5 + ipen, sig3}
    __shared__ int    s_bof[WPB][4][96];                // byte offset of the base's matrix column
    __shared__ int    s_feed[WPB][4][16 * BW + BW];
    // bottom-row results of a block's 16 steps, written by the bottom lane of a stripe, read back by lane = step at the
    // flush (a ds_write under a one-lane exec mask issues beside the VALU stream; the 64-bit DPP collectors it
    // replaces were two of the six most expensive instructions of the step, and sixteen registers)
    __shared__ int    s_out[WPB][4][16 * BW + BW];
    __shared__ int    s_prog_lds[WPB];

    const DevScoring* __restrict__ sc = A.sc;
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        int below = 0;                                  // codes below c that are not one of 2 3 5 9
        for (int x = 0; x < c; ++x) below += (x != 2 && x != 3 && x != 5 && x != 9);
        s_perm[c] = c == 2 ? 0 : c == 3 ? 1 : c == 5 ? 2 : c == 9 ? 3 : 4 + below;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) s_mtx[s_perm[i >> 5] * 36 + s_perm[i & 31]] = (float) sc->mtx[i];
    const int nquant = sc->nquant, llmt = sc->llmt;
    // index of the last table entry: every hil >= pen_cap prices alike and is an acceptor candidate
    const int pen_cap = max(nquant > 1 ? sc->qm_len[nquant - 2] + 1 : 0, llmt + 1);
    for (int h = threadIdx.x; SPJ && h <= pen_cap; h += blockDim.x) {
        // pen(hil) = qm_pen[j] for the last j with hil > qm_len[j-1]  (fwd2s1_wip_simd.h:163-167); no candidate
        // unless hil > llmt: the blend leaves `nevsel` there
        int pv = sc->qm_pen[0];
        for (int j = 1; j < nquant; ++j) if (h > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
        s_pen[h] = (h > llmt) ? make_float2(0.f, (float) pv) : make_float2(-BIGF, NEVF - FLOORF);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;                    // DPP row = stripe slot of the pass
    const int k = lane & 15;                    // lane within the stripe
    const float gef = (float) sc->gep, gnf = (float) (sc->gep + sc->gop);
    const int ge = sc->gep;
    const int cap8 = pen_cap * 8;               // hil is carried as the byte offset of its table entry

    const int wv = threadIdx.x >> 6;
    const int G = CROSS ? A.cross_g : 1;            // blocks cooperating on my problem
    const bool multi = CROSS || (int) blockIdx.x < A.n_multi;
    const int W = multi ? WPB * G : 1;              // waves cooperating on my problem
    const int w = multi ? (CROSS ? ((int) blockIdx.x % G) * WPB + wv : wv) : 0;     // my position among them
    int pi = CROSS ? (int) blockIdx.x / G
                   : (multi ? (int) blockIdx.x : A.n_multi + ((int) blockIdx.x - A.n_multi) * WPB + wv);
    pi = __builtin_amdgcn_readfirstlane(pi);
    const bool active = pi < A.n_probs;
    if (threadIdx.x < WPB) s_prog_lds[threadIdx.x] = 0;
    int* const s_prog = CROSS ? A.gprog + (int64_t) pi * (G * WPB + 2) : s_prog_lds;
    constexpr auto PSCOPE = CROSS ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP;
    if (!active) pi = A.n_probs - 1;                // keep the addressing valid until the barrier
    const DevProblem P = A.probs[pi];
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    int* __restrict__ bnd = A.bnd + P.bnd_off * BW;
    const int2* __restrict__ cols = A.cols + P.col_off;
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    const int n_ent = P.buf_size + SPDP_BND_PAD;
#define BIDX(r) ((r) - lw + 1)

    // ---- fhinitS1 (src/fwd2s1_simd.cc:163-239): boundary values by diagonal; F is `nevsel` there, so the entry
    // carries the vertical-gap candidate that follows from it
    {
        const int rl = b_left - a_left;
        int rr = min(b_right - a_left, up);
        int rr_g = rr;                                   // global: ramp stops where it reaches nevsel
        if (!a_exgl && ge) rr_g = min(rr, (SPDP_NEV16 - sc->gop) / ge + rl);
        const int ru = up + 2 * SPDP_NELEM;
        for (int e = lane + 64 * w; e < n_ent && active; e += 64 * W) {
            const int r = e + lw - 1;
            int h = SPDP_NEV16;
            if (b_exgl && r >= lw && r < rl) h = 0;
            if (a_exgl) { if (r >= rl && r <= rr) h = 0; }
            else {
                if (r == rl) h = 0;
                else if (r == rl + 1) h = sc->gop + ge;
                else if (ge) { if (r > rl + 1 && r < rr_g) h = sc->gop + ge + (r - rl - 1) * ge; }
                else if (r > rl + 1 && r < rr) h = sc->gop;
            }
            const float hf = (float) h;
            const float fcand = fmaxf(NEVF + gef, fmaxf(hf + gnf, FLOORF));
            if constexpr (UDH) {
                int c;                                   // link = diagonal where the path starts
                if (r >= rl) c = a_exgl ? ((r < ru) ? r : 0) : rl;
                else         c = b_exgl ? r : rl;
                if (r > ru) c = 0;
                st_b4<CROSS>(bnd + (int64_t) e * 4, make_int4(as_i(hf), as_i(fcand), c, c));
            } else {
                st_b2<CROSS>(bnd + (int64_t) e * 2, make_int2(as_i(hf), as_i(fcand)));
            }
        }
        if constexpr (UDH) {
            int* imd = A.imd + P.imd_off;
            const int tot = P.n_im * 4 * width;
            for (int e = lane + 64 * w; e < tot && active; e += 64 * W) st_b1<CROSS>(imd + e, END_OF_ULK);
        }
        if constexpr (CROSS) {
            // all blocks of the problem have initialised their share before any pass starts.  One word decides:
            // arrivals count in the low bits; a block that waited too long sets GIVEUP and leaves, and so does every
            // block that sees the bit on arrival or while waiting (one that got through just before ends at the bounded
            // wait for its producer); the host then repeats the launch with one CU per problem.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                constexpr int GIVEUP = 1 << 30;
                int* bar = s_prog + G * WPB;
                int seen = __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                long spins = 0;
                while (!(seen & GIVEUP) && (seen & (GIVEUP - 1)) < G) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1l << 22)) {         // a block of the group is not resident (the GPU is shared)
                        seen = __hip_atomic_fetch_or(bar, GIVEUP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | GIVEUP;
                        break;
                    }
                    seen = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const bool quit = (seen & GIVEUP) != 0;
                if (quit) __hip_atomic_store(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prog_lds[0] = quit ? 1 : 0;
            }
            __syncthreads();
            if (s_prog_lds[0]) return;
            WAVE_ORDER();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();                            // the only block-wide barrier (all waves reach it)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (!active) return;

    // UDH: intermediate rows (src/fwd2s1_wip_simd.h:503-508)
    const int n_im = UDH ? P.n_im : 0;
    const int imd_step = UDH ? (a_right - a_left + n_im) / (n_im + 1) : 0;
    int imd_cur = 0;

    const int n_stripes = (a_right - a_left + SPDP_NELEM - 1) / SPDP_NELEM;
    const int n_passes = (n_stripes + 3) >> 2;
    int64_t tb_base = P.tb_off;                         // forward: byte offset of the pass' first stripe
    constexpr int BIGB = 1 << 20;                       // progress word = pass * BIGB + blocks done
    const int prod = (w + W - 1) % W;                   // wave running the pass before mine
    bool dead = false;                                  // CROSS: a producer never showed up
    for (int s0 = 0; s0 < n_stripes && !dead; s0 += 4) {
        const int pass = s0 >> 2;
        const bool mine = (pass % W) == w;
        // ---- geometry of my stripe (row g of the wave)
        const int s = s0 + g;
        const int ml = a_left + s * SPDP_NELEM;
        const bool has = s < n_stripes;
        const int j9 = has ? min(SPDP_NELEM, a_right - ml) : 0;
        const int j8 = j9 - 1;
        const int n_start = max(b_left, lw + ml);
        const int n9 = min(b_right, up + (ml + j9) + 1) + j9;
        const int n_end = FWD ? n9 + 1 : n9;
        const int len = has ? max(0, n_end - n_start) : 0;
        const int nb = (len + 15) >> 4;
        // blocks this pass runs: row g is active for blocks [LAG*g, LAG*g + nb)
        const int nb0 = __builtin_amdgcn_readlane(nb, 0), nb1 = __builtin_amdgcn_readlane(nb, 16),
                  nb2 = __builtin_amdgcn_readlane(nb, 32), nb3 = __builtin_amdgcn_readlane(nb, 48);
        const int tot = max(max(nb0, SPDP_GROUP_LAG + nb1),
                            max(2 * SPDP_GROUP_LAG + nb2, 3 * SPDP_GROUP_LAG + nb3));
        int64_t my_tb = 0;
        if constexpr (FWD) {
            my_tb = tb_base + 256ll * ((g > 0 ? nb0 : 0) + (g > 1 ? nb1 : 0) + (g > 2 ? nb2 : 0));
            tb_base += 256ll * (nb0 + nb1 + nb2 + nb3);
        }
        // UDH: the reference walks its intermediates in order and tests, per stripe, only the
        // current one (src/fwd2s1_wip_simd.h:527,806-811): replay that pointer over my 4 stripes
        int imd_i = -1, k8 = -1;
        bool pass_imd = false;
        if constexpr (UDH) {
            for (int gg = 0; gg < 4; ++gg) {
                const int ml_gg = a_left + (s0 + gg) * SPDP_NELEM;
                if (imd_cur < n_im && ml_gg < a_right) {
                    const int cand = a_left + (imd_cur + 1) * imd_step;
                    const int mm = a_left + (cand - a_left - 1) / SPDP_NELEM * SPDP_NELEM;
                    if (mm == ml_gg) {
                        if (gg == g) { imd_i = imd_cur; k8 = cand - mm - 1; }
                        ++imd_cur; pass_imd = true;
                    }
                }
            }
        }
        const bool imd_row = UDH && imd_i >= 0;
        int* imd_p = nullptr;
        if constexpr (UDH) if (imd_row) imd_p = A.imd + P.imd_off + (int64_t) imd_i * 4 * width;
        if (!mine) continue;                            // bookkeeping above ran; the sweep is another wave's
        // only the last stripe of a problem can be partial (fewer than 16 rows)
        const bool pass_partial = (s0 + 4 >= n_stripes) && ((a_right - a_left) & 15);

        // my residue's row of the substitution matrix (byte offset into s_mtx)
        const int acode = (k < j9) ? acod[ml + k] : 0;
        const char* const mrow = reinterpret_cast<const char*>(s_mtx) + s_perm[acode & 31] * 144;

        // per-lane DP state: my H of the previous step and the two gap candidates that follow from it (Hg: gap
        // opened from H, shared by my E and by the F of the row below; Fm / FCm: the row below's vertical-gap
        // candidate and its link), E, the diagonal neighbour, the donor state
        float Hs = NEVF, E = NEVF, Hd = NEVF, hv2 = NEVF;
        float Hg = fmaxf(NEVF + gnf, FLOORF), Fm = fmaxf(NEVF + gef, Hg);
        // hil is carried as the byte offset of its table entry; the entry itself is read one step ahead (for
        // hil + 1, i.e. assuming no donor fires in between: a donor resets hil to 1, whose candidate is `nevsel`
        // whatever the donor score, llmt >= 1) so that no LDS round trip sits inside the cell-to-cell recurrence
        int hil8 = 0;
        float2 ptc = SPJ ? s_pen[0] : make_float2(0.f, 0.f);
        bool don_prev = false;
        int Cs = 0, FCm = 0, Cd = 0, ec = 0, hc2 = 0;              // UDH links
        int donor_r = 0, rlst = INT32_MAX;                         // UDH, lane k8 only
        int* const outb = &s_out[wv][g][0];
        const bool is_bottom = k == max(j8, 0);                    // (a partial last stripe: its last real row)
        float2* const sigring = &s_sig[wv][g][0];
        int*    const bofring = &s_bof[wv][g][0];
        int*    const feed = &s_feed[wv][g][0];

        auto run_pass = [&](auto partial_tag, auto imd_tag) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            constexpr bool IMD = decltype(imd_tag)::value;
            // registers holding the NEXT block's boundary entry / column record of this lane
            int4 nx_b = make_int4(0, 0, 0, 0);
            int2 nx_c = make_int2(0, 0);
            const rsrc_t brs = bnd_rsrc(bnd);                                 // (CROSS: the boundary array of this problem as a buffer)
            // a row's blocks are prefetched in order (block 0 once, then lb + 1 from block lb), so the two addresses and
            // the flush address below are carried along, 16 entries per block, instead of being rebuilt from lb
            int64_t pf_b = (int64_t) BIDX(n_start + k - ml) * BW;            // (index into bnd) sweep step n_start + 16 lbn + k of this lane
            const int2* pf_c = cols + (n_start + k);
            int* st_p = bnd + (int64_t) BIDX(n_start + k - (ml + 1) - 2 * j8) * BW;
            // the reference's write condition (fwd2s1_wip_simd.h:205-209) as a range of the sweep step n0 + k
            const int fl_lo = max(b_left + j9, lw + (ml + 1) + 2 * j8);
            const int fl_hi = j9 > 0 ? min(up + (ml + 1) + 2 * j8 + 1, n_end) : INT32_MIN;
            int lbm = 0;                                                // (16 lb) mod 48: where the block sits in the 48-slot rings
            auto prefetch = [&](int) {
                if constexpr (UDH) nx_b = ldx_b4<CROSS>(brs, bnd, pf_b);
                else { const int2 v = ldx_b2<CROSS>(brs, bnd, pf_b); nx_b.x = v.x; nx_b.y = v.y; }
                nx_c = *pf_c;                                           // raw record: no use here, the load must stay in flight
                pf_b += 16 * BW; pf_c += 16;
            };
            // multi-wave: row 0 reads boundary entries of the previous pass, produced by wave `prod`;
            // local block lbn of row 0 needs its absolute blocks <= lbn + 15 flushed (3 rows x LAG + 3)
            // CROSS: the word lives in memory, a look at it is a round trip the wave sits out.  The last value seen is kept, and
            // a wave that has to look waits until the producer is SLACK blocks further than it needs: a look per SLACK + 1
            // blocks instead of one per block (a pass trails its producer by up to that much more; the producer's last word,
            // pass * BIGB, ends every wait)
            // (only where the pipeline's skew, W x the lag between passes, is short against a pass: a launch of few tall narrow
            // problems is bound by that chain and every block of lag counts)
#ifndef SPDP_CROSS_SLACK
#define SPDP_CROSS_SLACK 8
#endif
            const int SLACK = (CROSS && W * (16 + SPDP_CROSS_SLACK) <= nb0) ? SPDP_CROSS_SLACK : 0;
            int seen = INT32_MIN;
            auto wait_for = [&](int lbn) {
                if (W == 1 || pass == 0) return;
                const int need = (pass - 1) * BIGB + lbn + 16;
                if (CROSS && seen >= need) return;
                long spins = 0;
                while ((seen = __hip_atomic_load(&s_prog[prod], __ATOMIC_RELAXED, PSCOPE)) < need + SLACK) {
                    __builtin_amdgcn_s_sleep(4);
                    if constexpr (CROSS) {
                        // a producer that never comes (it left at the barrier): mark the problem, the host re-runs it
                        if (++spins > (1l << 24)) {
                            __hip_atomic_store(s_prog + G * WPB + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            dead = true; return;
                        }
                    }
                }
                if constexpr (CROSS) WAVE_ORDER();      // the entries are read with memory-side loads
                else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            };
            wait_for(0);
            if (g == 0 && nb > 0) prefetch(0);
            for (int blk = 0; blk < tot && !dead; ++blk) {
                if (blk + 1 < nb0) wait_for(blk + 1);
                const int lb = blk - SPDP_GROUP_LAG * g;               // my local block number
                uint32_t code4[4] = {0, 0, 0, 0};
                if (lb == -1 && nb > 0) prefetch(0);                    // one block ahead of first use
                if (lb >= 0 && lb < nb) {
                    const int n0 = n_start + lb * 16;                   // sweep step of j = 0
                    if (lb == 0) {
                        // stripe start: the signal pipes are empty (zero), the residue pipe holds what in-window
                        // lanes need, and lane 0's up-left neighbour comes from the boundary array
                        const int c = n_start - 1 - k;
                        const int bof0 = s_perm[((c > b_left && c <= b_right) ? cols[c].y : 0) & 31] * 4;
                        if constexpr (SPJ) { sigring[15 - k] = make_float2(0.f, 0.f); sigring[15 - k + 48] = make_float2(0.f, 0.f); }
                        bofring[15 - k] = bof0; bofring[15 - k + 48] = bof0;
                        const int r = n_start - (ml + 1);
                        donor_r = r;
                        if (k == 0) {
                            Hd = as_f(ld_b1<CROSS>(&bnd[(int64_t) BIDX(r) * BW]));
                            if constexpr (UDH) Cd = ld_b1<CROSS>(&bnd[(int64_t) BIDX(r) * BW + 2]);
                        }
                    }
                    // ---- this block's chunk (prefetched one block ago) goes to LDS ...
                    if constexpr (UDH) reinterpret_cast<int4*>(feed)[k] = nx_b;
                    else reinterpret_cast<int2*>(feed)[k] = make_int2(nx_b.x, nx_b.y);
                    {
                        // column records are stored per absolute position of the parent sequence; the window edges
                        // are applied here: nothing beyond b_right, no residue at b_left
                        const int nn = n0 + k;
                        int sg = nx_c.x, bs = nx_c.y;
                        if (nn > b_right) { sg = 0; bs = 0; }
                        if (nn <= b_left) bs = 0;
                        const float2 srec = make_float2((float) (short) sg, (float) (sg >> 16));
                        const int bof = s_perm[bs & 31] * 4;
                        const int slot = lbm + k + 16 >= 48 ? lbm + k + 16 - 48 : lbm + k + 16;      // (16 lb + k + 16) mod 48
                        if constexpr (SPJ) { sigring[slot] = srec; sigring[slot + 48] = srec; }
                        bofring[slot] = bof; bofring[slot + 48] = bof;
                    }
                    // ---- ... and the next block's loads are issued now, to land while this one computes
                    if (lb + 1 < nb) prefetch(lb + 1);
                    WAVE_ORDER();
                    // lane k reads column n0 + J - k at step J: one contiguous run of 16 ring slots
                    const int myslot = lbm - k + 16 == 48 ? 0 : lbm - k + 16;             // (16 lb - k + 16) mod 48
                    const float2* const mysig = sigring + myslot;
                    const int* const mybof = bofring + myslot;
                    // intermediate row (lane k8 of its stripe): the steps of this block whose cell lies on it inside the band,
                    // its diagonal at step 0, and the four link planes at that diagonal -- once per block, not per step
                    int imd_jlo = 99, imd_jhi = -1, imd_r0 = 0, imd_r0w = 0;
                    int *imd_h0 = nullptr, *imd_h1 = nullptr, *imd_v = nullptr, *imd_f = nullptr;
                    if constexpr (UDH && IMD) {
                        if (imd_row && k == k8) {
                            imd_r0 = n0 - (ml + 1) - 2 * k;
                            imd_r0w = imd_r0 + width;
                            imd_jlo = max(0, lw - imd_r0);
                            imd_jhi = min(15, min(up - imd_r0, n_end - n0 - 1));
                            imd_h0 = imd_p + BIDX(imd_r0);
                            imd_h1 = imd_h0 + width; imd_v = imd_h1 + width; imd_f = imd_v + width;
                        }
                    }

                    // LDS operands are read ahead of the step that uses them: the matrix column offset three steps, the
                    // substitution score two, signals, feed entry and penalty entry one
                    int bofv[16]; float pvv[16]; float2 sgv[16]; int4 fdv[16];
                    code4[0] = code4[1] = code4[2] = code4[3] = 0;  // FWD: the 16 code bytes of my steps of this block
                    auto ld_feed = [&](int j) {
                        if constexpr (UDH) return reinterpret_cast<const int4*>(feed)[j];
                        else { const int2 v = reinterpret_cast<const int2*>(feed)[j]; return make_int4(v.x, v.y, 0, 0); }
                    };
                    auto ld_pv = [&](int bof) { return *reinterpret_cast<const float*>(mrow + bof); };
                    bofv[0] = mybof[0]; bofv[1] = mybof[1]; bofv[2] = mybof[2];
                    fdv[0] = ld_feed(0);
                    if constexpr (SPJ) sgv[0] = mysig[0];
                    pvv[0] = ld_pv(bofv[0]); pvv[1] = ld_pv(bofv[1]);

#define STEP(J)                                                                                  \
                    {                                                                                        \
                        if constexpr (J + 3 < 16) bofv[J + 3] = mybof[J + 3];                                \
                        if constexpr (J + 2 < 16) pvv[J + 2] = ld_pv(bofv[J + 2]);                           \
                        if constexpr (J + 1 < 16) { fdv[J + 1] = ld_feed(J + 1); if constexpr (SPJ) sgv[J + 1] = mysig[J + 1]; } \
                        int hil_n = 0; float2 pt_n = make_float2(0.f, 0.f);                                  \
                        if constexpr (SPJ) {                                                                 \
                            hil_n = min(hil8 + 8, cap8);                                                     \
                            pt_n = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(s_pen) + hil_n); \
                        }                                                                                    \
                        SPDP_PIN_LOADS();                                                                    \
                        /* neighbour exchange: lane 0 of the row takes the boundary entry of this step */    \
                        const int4 fd = fdv[J];                                                              \
                        const float upH = row_shr1(as_f(fd.x), Hs), fin = row_shr1(as_f(fd.y), Fm);          \
                        int upC = 0, fcin = 0;                                                               \
                        if constexpr (UDH) { upC = row_shr1(fd.z, Cs); fcin = row_shr1(fd.w, FCm); }         \
                        const float pv = pvv[J];                                                             \
                        /* horizontal gap: extend, or open from my H of the previous step */                 \
                        const float ee = E + gef;                                                            \
                        if constexpr (UDH) ec = (ee > Hg) ? ec : Cs;                                         \
                        unsigned code = 0;                                                                   \
                        if constexpr (FWD) code = (ee > Hg) ? 0u : (unsigned) TB_NHOR;                       \
                        E = fmaxf(ee, Hg);                                                                   \
                        float h; int hc = Cd, pb3 = 0;                                                       \
                        if constexpr (FWD) {                                                                 \
                            /* my F was opened (not extended) iff it equals the gap opened from the H above */ \
                            code |= (fin > fmaxf(upH + gnf, FLOORF)) ? 0u : (unsigned) TB_NVER;              \
                            h = fmaxf(Hd + pv, FLOORF);                                                      \
                            const bool c1 = fin > h;                                                         \
                            h = fmaxf(h, fin);                                                               \
                            const bool c2 = E > h;                                                           \
                            h = fmaxf(h, E);                                                                 \
                            code |= c2 ? (unsigned) TB_HORI : (c1 ? (unsigned) TB_VERT : (unsigned) TB_DIAG); \
                        } else if constexpr (UDH) {                                                          \
                            h = fmaxf(Hd + pv, FLOORF);                                                      \
                            const bool c1 = fin > h;                                                         \
                            hc = c1 ? fcin : hc; h = fmaxf(h, fin);                                          \
                            const bool c2 = E > h;                                                           \
                            hc = c2 ? ec : hc; h = fmaxf(h, E);                                              \
                            if constexpr (IMD) pb3 = c2 ? 1 : (c1 ? 2 : 0);                                  \
                        } else {                                                                             \
                            h = fmaxf(fmaxf(Hd + pv, fin), E);                                               \
                        }                                                                                    \
                        bool is_acc = false, is_don = false;                                                 \
                        if constexpr (SPJ) {                                                                 \
                            const float2 sg2 = sgv[J];                                                       \
                            float x = fmaxf(hv2 + sg2.y + ptc.x, FLOORF) + ptc.y;                            \
                            x = don_prev ? NEVF : x;                                                         \
                            if constexpr (UDH) { is_acc = x > h; hc = is_acc ? hc2 : hc; }                   \
                            if constexpr (FWD) { is_acc = x > h; code = is_acc ? ((code & ~15u) | (unsigned) TB_ACCR) : code; } \
                            h = fmaxf(h, x);                                                                 \
                            /* (forward flavour: a cell entered through an acceptor is no donor, spdp_kernels.hip) */ \
                            const float qd = (FWD && is_acc) ? NEVF : h + sg2.x;                             \
                            is_don = qd > hv2;                                                               \
                            hv2 = fmaxf(hv2, qd);                                                            \
                            if constexpr (UDH) hc2 = is_don ? hc : hc2;                                      \
                            hil8 = is_don ? 8 : hil_n;                                                       \
                            ptc = pt_n; don_prev = is_don;                                                   \
                            if constexpr (FWD) code |= is_don ? (unsigned) TB_DONR : 0u;                     \
                        }                                                                                    \
                        if constexpr (FWD) {                                                                 \
                            if constexpr (PARTIAL) { if (k >= j9) code = 0; }                                \
                            /* packed at once: left to the compiler the sixteen bytes are combined after the last step and */ \
                            /* everything they depend on stays live till then (212 registers instead of 96) */ \
                            asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(code4[J >> 2]) : "v"(code), "n"(8 * (J & 3))); \
                        }                                                                                    \
                        int fl = fcin;                                  /* link of my F */                   \
                        if constexpr (UDH && IMD) {                                                          \
                            /* scalar bookkeeping of the intermediate row (lane k8 of its stripe) */         \
                            if (J >= imd_jlo && J <= imd_jhi) {                                              \
                                const int rj = imd_r0 + J;                      /* my cell's diagonal */     \
                                if (SPJ && is_acc) { st_b1<CROSS>(imd_h0 + J, donor_r); st_b1<CROSS>(imd_h1 + J, donor_r + width); } \
                                rlst = ((SPJ && is_acc) || pb3 == 0) ? rj : rlst;                            \
                                if (SPJ) donor_r = is_don ? rj : donor_r;                                    \
                                if (pb3 == 1) st_b1<CROSS>(imd_h0 + J, rlst);                                \
                                st_b1<CROSS>(imd_v + J, hc); hc = rj;                                        \
                                st_b1<CROSS>(imd_f + J, fl); fl = imd_r0w + J;                               \
                            }                                                                                \
                        }                                                                                    \
                        Hd = upH; Hs = h;                                                                    \
                        if constexpr (UDH) { Cd = upC; Cs = hc; }                                            \
                        /* what follows from my cell: the gap opened from H, the row below's F candidate */  \
                        Hg = fmaxf(h + gnf, FLOORF);                                                         \
                        const float fe = fin + gef;                                                          \
                        if constexpr (UDH) FCm = (fe > Hg) ? fl : hc;                                        \
                        Fm = fmaxf(fe, Hg);                                                                  \
                        /* bottom lane of the stripe -> slot J of the row's output block */                  \
                        if (is_bottom) {                                                                     \
                            if constexpr (UDH) reinterpret_cast<int4*>(outb)[J] = make_int4(as_i(Hs), as_i(Fm), Cs, FCm); \
                            else reinterpret_cast<int2*>(outb)[J] = make_int2(as_i(Hs), as_i(Fm));           \
                        }                                                                                    \
                    }
                    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
                    STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
                }
                // CROSS: the progress word tells the waves of other CUs which blocks have LANDED in memory.  Waiting for this
                // block's stores right after issuing them stalled the wave for a memory round trip per block (66 % of the
                // resident cycles of C5's sweeps were SQ_WAIT_ANY); the word now runs one block behind: here, after the sixteen
                // steps, the stores of the previous block have long landed and the wait is free
                if constexpr (CROSS) {
                    if (W > 1 && blk > 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(&s_prog[w], pass * BIGB + blk, __ATOMIC_RELAXED, PSCOPE);
                    }
                }
                if (lb >= 0 && lb < nb) {
                    // ---- flush: lane i takes the bottom-row result of step j = i; it goes to the boundary array under the
                    // reference's write condition (fwd2s1_wip_simd.h:205-209)
                    {
                        const int n = n_start + lb * 16 + k;
                        if (n >= fl_lo && n < fl_hi) {
                            if constexpr (UDH) st_b4<CROSS>(st_p, reinterpret_cast<const int4*>(outb)[k]);
                            else st_b2<CROSS>(st_p, reinterpret_cast<const int2*>(outb)[k]);
                        }
                        st_p += 16 * BW;
                        lbm = lbm == 32 ? 0 : lbm + 16;
                    }
                    if constexpr (FWD) {
                        uint4* dst = reinterpret_cast<uint4*>(A.tb + my_tb + 256ll * lb + 16 * k);
                        *dst = make_uint4(code4[0], code4[1], code4[2], code4[3]);
                    }
                }
                // boundary entries are exchanged between the rows of this wave through memory: a load issued after a
                // store of the same wave to the same address observes it (in-order vector memory path, loads bypass
                // L1), so only the compiler needs a fence here
                WAVE_ORDER();
                if constexpr (!CROSS) {
                    if (W > 1) {                                        // publish: this block's stores are done
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0)
                            __hip_atomic_store(&s_prog[w], pass * BIGB + blk + 1, __ATOMIC_RELAXED, PSCOPE);
                    }
                }
            }
            if (W > 1) {
                if constexpr (CROSS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0)
                    __hip_atomic_store(&s_prog[w], (pass + 1) * BIGB, __ATOMIC_RELAXED, PSCOPE);
            }
        };
        if (pass_partial) {
            if (pass_imd) run_pass(BoolTag<true>{}, BoolTag<true>{});
            else          run_pass(BoolTag<true>{}, BoolTag<false>{});
        } else {
            if (pass_imd) run_pass(BoolTag<false>{}, BoolTag<true>{});
            else          run_pass(BoolTag<false>{}, BoolTag<false>{});
        }
    }
    if (dead) {
        // keep the waves behind me from waiting for the full time-out each
        if (lane == 0) __hip_atomic_store(&s_prog[w], INT32_MAX, __ATOMIC_RELAXED, PSCOPE);
        return;
    }

    if (W > 1) {
        if ((n_passes - 1) % W != w) return;            // the wave of the last pass finishes the problem
        for (int x = 0; x < W; ++x) {
            if (x == w) continue;
            if (n_passes - 1 - x < 0) continue;
            const int last_own = ((n_passes - 1 - x) / W) * W + x;      // last pass of wave x
            long spins = 0;
            while (__hip_atomic_load(&s_prog[x], __ATOMIC_RELAXED, PSCOPE) < (last_own + 1) * BIGB) {
                __builtin_amdgcn_s_sleep(4);
                if constexpr (CROSS) {
                    if (++spins > (1l << 24)) {
                        __hip_atomic_store(s_prog + G * WPB + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                }
            }
        }
        if constexpr (CROSS) WAVE_ORDER();
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // ---- fhlastS1 (src/fwd2s1_simd.cc:241-262)
    DevResult R;
    R.score = SPDP_NEV16; R.mr = a_right; R.nr = b_right; R.ml = a_left; R.ulk = END_OF_ULK; R.maxr = 0;
    R.pad[0] = R.pad[1] = 0;
    {
        const int rr = b_right - a_right;
        // first maximum over [lo, hi): returns index (lo if the range is empty)
        auto argmax_first = [&](int lo, int hi) {
            float bv = -3.0e38f; int bi = INT32_MAX;
            for (int r = lo + lane; r < hi; r += 64) {
                const float v = as_f(ld_b1<CROSS>(&bnd[(int64_t) BIDX(r) * BW]));
                if (v > bv) { bv = v; bi = r; }
            }
            for (int off = 32; off; off >>= 1) {
                const float ov = __shfl_xor(bv, off); const int oi = __shfl_xor(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            return (bi == INT32_MAX) ? lo : bi;
        };
        int maxr = rr;
        if (a_exgr) maxr = argmax_first(max(lw, b_left - a_right), rr);
        if (b_exgr) {
            const int r2 = min(up - 1, b_right - a_left);
            int mv = argmax_first(rr, r2);
            if (r2 - rr < 1) mv = rr;
            if (as_f(ld_b1<CROSS>(&bnd[(int64_t) BIDX(mv) * BW])) > as_f(ld_b1<CROSS>(&bnd[(int64_t) BIDX(maxr) * BW]))) maxr = mv;
        }
        R.score = (int) as_f(ld_b1<CROSS>(&bnd[(int64_t) BIDX(maxr) * BW]));
        if (maxr > rr) R.mr = b_right - maxr; else R.nr = a_right + maxr;
        if constexpr (UDH) R.ulk = ld_b1<CROSS>(&bnd[(int64_t) BIDX(maxr) * BW + 2]);
        R.maxr = maxr;
    }
    if (lane == 0) A.res[pi] = R;
#undef BIDX
}

// ---------------------------------------------------------------------------
template <int FL, bool SPJ>
static hipError_t launch_fp(dim3 grd, int wpb, hipStream_t stream, const SweepArgs& A)
{
    if constexpr (FL == FL_UDH || FL == FL_FORWARD) {
        if (A.cross_g > 0) {
            // one problem over several CUs: every block must be resident, so this is a cooperative launch
            SweepArgs Ac = A;
            void* kargs[] = {&Ac};
            const void* fn = wpb == 16 ? (const void*) spdp_sweep_fp<FL, 16, true, SPJ> : (const void*) spdp_sweep_fp<FL, 4, true, SPJ>;
            return hipLaunchCooperativeKernel(fn, grd, dim3(wpb == 16 ? 1024 : 256), kargs, 0, stream);
        }
    }
    // SPDP_LDS_PAD=<bytes>: extra dynamic LDS per block, i.e. fewer resident blocks per CU (occupancy experiments)
    static const int lds_pad = getenv("SPDP_LDS_PAD") ? atoi(getenv("SPDP_LDS_PAD")) : 0;
    if (wpb == 16) hipLaunchKernelGGL((spdp_sweep_fp<FL, 16, false, SPJ>), grd, dim3(1024), 0, stream, A);
    else           hipLaunchKernelGGL((spdp_sweep_fp<FL, 4, false, SPJ>), grd, dim3(256), lds_pad, stream, A);
    return hipGetLastError();
}

// does this kernel serve such a run (else it stays with spdp_kernels.hip)
extern "C" int spdp_sweep_fp_serves(int local, int spj, int nquant, int pen_cap, int llmt)
{
    if (local) return 0;
    const int cap = nquant > 1 ? pen_cap : 0;
    return !(spj && (llmt < 1 || (cap > llmt + 1 ? cap : llmt + 1) >= SPDP_FPEN_TAB));
}

// returns hipErrorNotSupported when the launch has to stay with spdp_kernels.hip
extern "C" hipError_t spdp_launch_sweep_fp(int flavour, int local, int spj, int nquant, int pen_cap, int llmt,
                                           const SweepArgs* args, int grid, int wpb, hipStream_t stream)
{
    if (!spdp_sweep_fp_serves(local, spj, nquant, pen_cap, llmt)) return hipErrorNotSupported;
    const int blk = wpb == 16 ? 16 : 4;
    const dim3 grd(grid);
    if (flavour == FL_SCORE) return spj ? launch_fp<FL_SCORE, true>(grd, blk, stream, *args) : launch_fp<FL_SCORE, false>(grd, blk, stream, *args);
    if (flavour == FL_UDH) return spj ? launch_fp<FL_UDH, true>(grd, blk, stream, *args) : launch_fp<FL_UDH, false>(grd, blk, stream, *args);
    if (flavour == FL_FORWARD) return spj ? launch_fp<FL_FORWARD, true>(grd, blk, stream, *args) : launch_fp<FL_FORWARD, false>(grd, blk, stream, *args);
    return hipErrorInvalidValue;
}
