// spdp_kernels.hip -- CDNA4 (gfx950) kernels for the spliced-alignment DP of
// ogotoh/spaln's `_wip` engines (reference: src/fwd2s1_wip_simd.h:42 / 233 / 476,
// boundary set-up src/fwd2s1_simd.cc:163-262).  Written for wave64 / DPP rows;
// no CUDA idiom, no MFMA (integer recurrence).
//
// Mapping.  The reference sweeps stripes of 16 query rows (AVX2 int16 lanes),
// lane k of a stripe holding cell (m = ml+1+k, n-k) at sweep step n, and chains
// stripes through per-diagonal arrays hv/fv.  A DPP row is exactly 16 lanes, so
// one wave64 runs FOUR consecutive stripes of one problem at once: row g of
// the wave is stripe 4*pass+g, started SPDP_GROUP_LAG blocks (64 columns) after
// row g-1, so that every boundary value it needs has already been produced.
// The up / up-left neighbour exchange is a single `row_shr:1` DPP move per
// value; lane 0 of each row is fed from a 16-entry register chunk of the
// boundary array by `row_shl:j`; the bottom lane's results are collected in a
// register shift chain (`row_ror:1` + `row_shr:1`) and written back coalesced
// every 16 steps.  Stripes keep the reference's exact geometry (band staircase,
// out-of-window lanes, first-column rule), so results are those of the AVX2
// build bit for bit -- with int32 scores instead of re-based int16 ones (adds
// saturate at SHRT_MIN as _mm256_adds_epi16 does on the low side).
//
// One wave = one problem (four problems per 256-thread block); the hardware dispatcher
// balances the load.
// HBM traffic per 64 rows x 1 column: 8 B column record (+3 L2 re-reads),
// 8/16 B boundary read + 8/16 B boundary write; forward adds 1 B / cell of
// traceback codes.

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

enum { FL_SCORE = 0, FL_FORWARD = 1, FL_UDH = 2 };

// TraceBackCode values (src/rhomb_coord.h:36-61)
enum { TB_DIAG = 1, TB_HORI = 2, TB_VERT = 8, TB_ACCR = 14, TB_NHOR = 16, TB_NVER = 32, TB_DONR = 128 };

#define END_OF_ULK (INT32_MAX - 2)

#define DPP_ROW_SL(n) (0x100 + (n))
#define DPP_ROW_SR(n) (0x110 + (n))
#define DPP_ROW_RR(n) (0x120 + (n))

// lane i of every 16-lane row <- lane i-1; lane 0 of the row keeps `old`
__device__ __forceinline__ int row_shr1(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, DPP_ROW_SR(1), 0xf, 0xf, false);
}
// lane 0 of every row <- lane J of that row (other lanes: don't care)
template <int J>
__device__ __forceinline__ int row_pick(int src)
{
    if constexpr (J == 0) return src;
    else return __builtin_amdgcn_mov_dpp(src, DPP_ROW_SL(J), 0xf, 0xf, true);
}
// lane 0 of every row <- lane 15 of that row
__device__ __forceinline__ int row_ror1(int src)
{
    return __builtin_amdgcn_mov_dpp(src, DPP_ROW_RR(1), 0xf, 0xf, true);
}

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int sadd16(int a, int b) { return max(a + b, SPDP_FLOOR16); }

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
// L1-bypassing loads for data another row of this wave stored a few blocks ago
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

// Boundary entries that cross CUs (CROSS kernels): the per-XCD L2s are not coherent with each other, so every
// access to the boundary array goes to the memory side -- agent-scope relaxed atomics compile to sc1
// loads / stores (write-through, no allocation of stale lines); a flag published after the stores have
// drained (s_waitcnt vmcnt(0)) orders them for the consumer.
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

// Ordering inside ONE wave needs no hardware fence: a wave's LDS instructions execute in order, and so
// do its vector-memory instructions (a load issued after a store of the same wave to the same address
// observes it; the rows of a pass are >= SPDP_GROUP_LAG blocks apart anyway).  Only the compiler must
// not move accesses across these points.  (A wavefront-scope __builtin_amdgcn_fence would do, but it
// makes the compiler emit s_waitcnt vmcnt(0) right after the prefetch loads -- no prefetch left.)
#define WAVE_ORDER() asm volatile("" ::: "memory")

// intron-length penalty modes: flat (nquant == 1, the -A3 model), LDS table, select chain
enum { NQ_FLAT = 0, NQ_TABLE = 1, NQ_CHAIN = 2 };
#define SPDP_PEN_TAB 2048

// ---------------------------------------------------------------------------
// WPB: waves per block.  4 by default; 16 when a launch holds only a few huge problems, each of which
// then owns a whole CU (a 16-wave pass pipeline instead of a 4-wave one).
// CROSS: one problem is spread over A.cross_g blocks (CUs): global wave c * WPB + wv runs passes
// c * WPB + wv, + cross_g * WPB, ..; progress words live in global memory and the boundary array is
// accessed with memory-side coherent loads / stores.  All blocks of the launch must be resident (grid
// <= number of CUs; 16-wave blocks take a whole CU each).
template <int FL, bool LOCAL, int NQM, int WPB, bool CROSS = false>
__global__ __launch_bounds__(WPB * 64) void spdp_sweep(SweepArgs A)
{
    constexpr int BW = (FL == FL_UDH) ? 4 : 2;          // ints per boundary entry
    __shared__ int s_mtx[32 * 32];
    __shared__ short s_pen[NQM == NQ_TABLE ? SPDP_PEN_TAB : 2];
    // per wave, per DPP row: a 2 x 48-entry ring of column records (every record is written
    // twice, 48 slots apart, so that any 31-column window is contiguous) and the 16 boundary
    // entries of the current block
    __shared__ int2 s_col[WPB][4][96];
    __shared__ int  s_feed[WPB][4][16 * BW];
    // bottom-row results of a block's 16 steps: the bottom lane of a stripe writes slot J, lane 15 - j reads slot j at the
    // flush (replaces a rotate + shift DPP pair per value and step: spdp_sweep_fp.hip has the measurements)
    __shared__ int  s_out[WPB][4][16 * BW + BW];

    const DevScoring* __restrict__ sc = A.sc;
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) s_mtx[i] = sc->mtx[i];
    const int nquant = sc->nquant;
    const int pen_cap = (NQM == NQ_TABLE) ? sc->qm_len[nquant - 2] + 1 : 0;
    if constexpr (NQM == NQ_TABLE) {
        // pen(hil) = qm_pen[j] for the last j with hil > qm_len[j-1]  (fwd2s1_wip_simd.h:163-167)
        for (int h = threadIdx.x; h <= pen_cap; h += blockDim.x) {
            int pv = sc->qm_pen[0];
            for (int j = 1; j < nquant; ++j) if (h > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
            s_pen[h] = (short) pv;
        }
    }
    __syncthreads();
    // select-chain form of pen(hil): thresholds and penalties in scalar registers, unused steps never fire
    int ql[7], qp[7];
    #pragma unroll
    for (int j = 0; j < 7; ++j) {
        const bool on = (NQM == NQ_CHAIN) && j + 1 < nquant;
        ql[j] = on ? sc->qm_len[j] : 0x7fffffff;
        qp[j] = on ? sc->qm_pen[j + 1] : 0;
    }

    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;                    // DPP row = stripe slot of the pass
    const int k = lane & 15;                    // lane within the stripe
    const int ge = sc->gep, gn = sc->gep + sc->gop;
    const int spj = sc->spj, llmt = sc->llmt;
    const int p0 = sc->qm_pen[0];

    // Work mapping.  The first n_multi problems (the largest) own a whole 4-wave block each: wave w
    // runs passes w, w+4, w+8, .. (a pass = 4 stripes = 64 query rows) behind wave w-1, which it
    // follows at a distance through progress words in LDS -- a wavefront pipeline inside the CU.
    // All other problems take one wave each (4 per block).  The hardware dispatcher balances the
    // load (blocks are launched as CUs free up), so no software queue is needed.
    __shared__ int s_prog_lds[WPB];
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
    // progress words: LDS inside one block, global memory (zeroed by the host) across blocks
    int* const s_prog = CROSS ? A.gprog + (int64_t) pi * (G * WPB + 2) : s_prog_lds;
    constexpr auto PSCOPE = CROSS ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP;
    if (!active) pi = A.n_probs - 1;                // keep the addressing valid until the barrier
    const DevProblem P = A.probs[pi];
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool LocalL = LOCAL && a_exgl && b_exgl;
    const bool LocalR = LOCAL && a_exgr && b_exgr;
    int* __restrict__ bnd = A.bnd + P.bnd_off * BW;
    const int2* __restrict__ cols = A.cols + P.col_off;
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    const int n_ent = P.buf_size + SPDP_BND_PAD;
#define BIDX(r) ((r) - lw + 1)

    // ---- fhinitS1 (src/fwd2s1_simd.cc:163-239): boundary values by diagonal
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
            if constexpr (FL == FL_UDH) {
                int c;                                   // link = diagonal where the path starts
                if (r >= rl) c = a_exgl ? ((r < ru) ? r : 0) : rl;
                else         c = b_exgl ? r : rl;
                if (r > ru) c = 0;
                st_b4<CROSS>(bnd + (int64_t) e * 4, make_int4(h, SPDP_NEV16, c, c));
            } else {
                st_b2<CROSS>(bnd + (int64_t) e * 2, make_int2(h, SPDP_NEV16));
            }
        }
        if constexpr (FL == FL_UDH) {
            int* imd = A.imd + P.imd_off;
            const int tot = P.n_im * 4 * width;
            // (CROSS: memory-side stores -- a dirty line left in one XCD's L2 would later overwrite the links
            //  another XCD's wave stores into the same line)
            for (int e = lane + 64 * w; e < tot && active; e += 64 * W) st_b1<CROSS>(imd + e, END_OF_ULK);
        }
        if constexpr (CROSS) {
            // all blocks of the problem have initialised their share before any pass starts
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                // One word decides: arrivals count in the low bits; a block that waited too long sets GIVEUP and leaves,
                // and so does every block that sees the bit on arrival or while waiting (one that got through just
                // before ends at the bounded wait for its producer); the host then repeats the launch with one CU
                // per problem (DevRun::sync).
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
    const int n_im = (FL == FL_UDH) ? P.n_im : 0;
    const int imd_step = (FL == FL_UDH) ? (a_right - a_left + n_im) / (n_im + 1) : 0;
    int imd_cur = 0;

    // running maximum for local right ends: value, then first (stripe, step, lane)
    int best_val = SPDP_NEV16, best_mr = a_right, best_nr = b_right, best_ml = a_left, best_ulk = END_OF_ULK;
    unsigned long long best_key = ~0ull;

    int64_t tb_base = P.tb_off;                         // forward: byte offset of the pass' first stripe
    const int n_stripes = (a_right - a_left + SPDP_NELEM - 1) / SPDP_NELEM;
    const int n_passes = (n_stripes + 3) >> 2;
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
        const int n_end = (FL == FL_FORWARD) ? n9 + 1 : n9;
        const int len = has ? max(0, n_end - n_start) : 0;
        const int nb = (len + 15) >> 4;
        // blocks this pass runs: row g is active for blocks [LAG*g, LAG*g + nb)
        const int nb0 = __builtin_amdgcn_readlane(nb, 0), nb1 = __builtin_amdgcn_readlane(nb, 16),
                  nb2 = __builtin_amdgcn_readlane(nb, 32), nb3 = __builtin_amdgcn_readlane(nb, 48);
        const int tot = max(max(nb0, SPDP_GROUP_LAG + nb1),
                            max(2 * SPDP_GROUP_LAG + nb2, 3 * SPDP_GROUP_LAG + nb3));
        int64_t my_tb = 0;
        if constexpr (FL == FL_FORWARD) {
            my_tb = tb_base + 256ll * ((g > 0 ? nb0 : 0) + (g > 1 ? nb1 : 0) + (g > 2 ? nb2 : 0));
            tb_base += 256ll * (nb0 + nb1 + nb2 + nb3);
        }
        // UDH: the reference walks its intermediates in order and tests, per stripe, only the
        // current one (src/fwd2s1_wip_simd.h:527,806-811): replay that pointer over my 4 stripes
        int imd_i = -1, k8 = -1;
        bool pass_imd = false;
        if constexpr (FL == FL_UDH) {
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
        const bool imd_row = (FL == FL_UDH) && imd_i >= 0;
        int* imd_p = nullptr;
        if constexpr (FL == FL_UDH) if (imd_row) imd_p = A.imd + P.imd_off + (int64_t) imd_i * 4 * width;
        if (!mine) continue;                            // bookkeeping above ran; the sweep is another wave's
        // only the last stripe of a problem can be partial (fewer than 16 rows)
        const bool pass_partial = (s0 + 4 >= n_stripes) && ((a_right - a_left) & 15);

        // my residue row of the substitution matrix
        const int acode = (k < j9) ? acod[ml + k] : 0;
        const int* mrow = s_mtx + acode * 32;

        // per-lane DP state
        int Hs = SPDP_NEV16, Fs = SPDP_NEV16, E = SPDP_NEV16, Hd = SPDP_NEV16;
        int hv2 = SPDP_NEV16, hil = 0;
        int Cs = 0, FCs = 0, Cd = 0, ec = 0, hc2 = 0;              // UDH links
        int donor_r = 0, rlst = INT32_MAX;                         // UDH, lane k8 only
        int* const outb = &s_out[threadIdx.x >> 6][g][0];
        const bool is_bottom = k == max(j8, 0);                    // (a partial last stripe: its last real row)
        int2* const colring = &s_col[wv][g][0];
        int*  const feed = &s_feed[wv][g][0];

        auto run_pass = [&](auto partial_tag, auto imd_tag) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
            constexpr bool IMD = decltype(imd_tag)::value;
            // registers holding the NEXT block's boundary entry / column record of this lane
            int4 nx_b = make_int4(0, 0, 0, 0);
            int2 nx_c = make_int2(0, 0);
            auto prefetch = [&](int lbn) {
                const int nn = n_start + lbn * 16 + k;                  // sweep step this lane loads for
                if constexpr (FL == FL_UDH) nx_b = ld_b4<CROSS>(bnd + (int64_t) BIDX(nn - ml) * 4);
                else { const int2 v = ld_b2<CROSS>(bnd + (int64_t) BIDX(nn - ml) * 2); nx_b.x = v.x; nx_b.y = v.y; }
                // raw record: nothing may depend on the loaded value here, or the compiler has to wait
                // for the load on the spot and the prefetch is gone (nn < b_len + SPDP_COL_PAD always)
                nx_c = cols[nn];
            };
            // multi-wave: row 0 reads boundary entries of the previous pass, produced by wave `prod`;
            // local block lbn of row 0 needs its absolute blocks <= lbn + 15 flushed (3 rows x LAG + 3)
            auto wait_for = [&](int lbn) {
                if (W == 1 || pass == 0) return;
                const int need = (pass - 1) * BIGB + lbn + 16;
                long spins = 0;
                while (__hip_atomic_load(&s_prog[prod], __ATOMIC_RELAXED, PSCOPE) < need) {
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
                if (lb == -1 && nb > 0) prefetch(0);                    // one block ahead of first use
                if (lb >= 0 && lb < nb) {
                    const int n0 = n_start + lb * 16;                   // sweep step of j = 0
                    if (lb == 0) {
                        // stripe start: the signal pipes are empty (zero), the residue pipe holds
                        // what in-window lanes need, and lane 0's up-left neighbour comes from the
                        // boundary array
                        const int c = n_start - 1 - k;
                        const int2 rec0 = make_int2(0, (c > b_left && c <= b_right) ? cols[c].y : 0);
                        colring[15 - k] = rec0; colring[15 - k + 48] = rec0;
                        const int r = n_start - (ml + 1);
                        donor_r = r;
                        if (k == 0) {
                            Hd = ld_b1<CROSS>(&bnd[(int64_t) BIDX(r) * BW]);
                            if constexpr (FL == FL_UDH) Cd = ld_b1<CROSS>(&bnd[(int64_t) BIDX(r) * BW + 2]);
                        }
                    }
                    // ---- this block's chunk (prefetched one block ago) goes to LDS ...
                    if constexpr (FL == FL_UDH) reinterpret_cast<int4*>(feed)[k] = nx_b;
                    else reinterpret_cast<int2*>(feed)[k] = make_int2(nx_b.x, nx_b.y);
                    {
                        // column records are stored per absolute position of the parent sequence; the
                        // window edges are applied here: nothing beyond b_right, no residue at b_left
                        const int nn = n0 + k;
                        int2 crec = nx_c;
                        if (nn > b_right) crec = make_int2(0, 0);
                        if (!spj) crec.x = 0;
                        if (nn <= b_left) crec.y = 0;
                        const int slot = (lb * 16 + k + 16) % 48;
                        colring[slot] = crec; colring[slot + 48] = crec;
                    }
                    // ---- ... and the next block's loads are issued now, to land while this one computes
                    if (lb + 1 < nb) prefetch(lb + 1);
                    WAVE_ORDER();
                    // lane k reads column n0 + J - k at step J: one contiguous run of 16 ring slots
                    const int2* const mycol = colring + ((lb * 16 - k + 16 + 48) % 48);
                    uint32_t code4[4] = {0, 0, 0, 0};

#define STEP(J)                                                                                  \
                    {                                                                                        \
                        /* neighbour exchange: lane 0 of the row takes the boundary entry of this step */    \
                        int upH, upF, upC = 0, upFC = 0;                                                     \
                        if constexpr (FL == FL_UDH) {                                                        \
                            const int4 fd = reinterpret_cast<const int4*>(feed)[J];                          \
                            upH = row_shr1(fd.x, Hs); upF = row_shr1(fd.y, Fs);                              \
                            upC = row_shr1(fd.z, Cs); upFC = row_shr1(fd.w, FCs);                            \
                        } else {                                                                             \
                            const int2 fd = reinterpret_cast<const int2*>(feed)[J];                          \
                            upH = row_shr1(fd.x, Hs); upF = row_shr1(fd.y, Fs);                              \
                        }                                                                                    \
                        const int2 cr = mycol[J];                                                            \
                        const int sigp = cr.x;                                                               \
                        const int pv = mrow[cr.y];                                                           \
                        int h, f, hc = 0, fc = 0;                                                            \
                        unsigned code = 0; int pb3 = 0;                                                      \
                        if constexpr (FL == FL_SCORE) {                                                      \
                            E = max3i(E + ge, Hs + gn, SPDP_FLOOR16);                                        \
                            f = max3i(upF + ge, upH + gn, SPDP_FLOOR16);                                     \
                            h = max3i(Hd + pv, f, E);                                                        \
                        } else {                                                                             \
                            const int eo = sadd16(Hs, gn), ee = E + ge;                                      \
                            const bool ext = ee > eo;            /* == sadd16(E, ge) > eo */                 \
                            E = ext ? ee : eo;                                                               \
                            if constexpr (FL == FL_UDH) ec = ext ? ec : Cs;                                  \
                            const int fo = sadd16(upH, gn), fe = upF + ge;                                   \
                            const bool fext = fe > fo;                                                       \
                            f = fext ? fe : fo;                                                              \
                            if constexpr (FL == FL_UDH) fc = fext ? upFC : upC;                              \
                            if constexpr (FL == FL_FORWARD) code = (ext ? 0u : (unsigned) TB_NHOR) | (fext ? 0u : (unsigned) TB_NVER); \
                            h = sadd16(Hd, pv); hc = Cd;                                                     \
                            unsigned dir = TB_DIAG;                                                          \
                            if (f > h) { h = f; hc = fc; dir = TB_VERT; pb3 = 2; }                           \
                            if (E > h) { h = E; hc = ec; dir = TB_HORI; pb3 = 1; }                           \
                            code |= dir;                                                                     \
                        }                                                                                    \
                        bool is_acc = false, is_don = false;                                                 \
                        if (spj) {                                                                           \
                            const int s3 = sigp >> 16;                                                       \
                            const int s5 = (int) (short) sigp;                                               \
                            int pen = p0;                                                                    \
                            if constexpr (NQM == NQ_TABLE) pen = s_pen[min(hil, pen_cap)];                   \
                            if constexpr (NQM == NQ_CHAIN) {                                                 \
                                _Pragma("unroll")                                                            \
                                for (int jq = 0; jq < 7; ++jq) pen = (hil > ql[jq]) ? qp[jq] : pen;          \
                            }                                                                                \
                            int x = sadd16(hv2, s3) + pen;                                                   \
                            x = (hil > llmt) ? x : SPDP_NEV16;                                               \
                            if constexpr (FL == FL_SCORE) { h = max(h, x); }                                 \
                            else if (x > h) {                                                                \
                                h = x; is_acc = true; hc = hc2;                                              \
                                if constexpr (FL == FL_FORWARD) code = (code & ~15u) | TB_ACCR;              \
                            }                                                                                \
                            if (LOCAL && LocalL && h < 0) { h = 0; if constexpr (FL == FL_FORWARD) code &= 15u; } \
                            int qd = h + s5;                                                                 \
                            if constexpr (FL == FL_FORWARD) qd = is_acc ? SPDP_NEV16 : qd;                   \
                            is_don = qd > hv2;                                                               \
                            hv2 = is_don ? qd : hv2;                                                         \
                            if constexpr (FL == FL_UDH) hc2 = is_don ? hc : hc2;                             \
                            hil = is_don ? 1 : hil + 1;                                                      \
                            if constexpr (FL == FL_FORWARD) code |= is_don ? TB_DONR : 0u;                   \
                        } else if (LOCAL && LocalL && h < 0) {                                               \
                            h = 0; if constexpr (FL == FL_FORWARD) code &= 15u;                              \
                        }                                                                                    \
                        Hd = upH; Hs = h; Fs = f;                                                            \
                        if constexpr (FL == FL_UDH) { Cd = upC; Cs = hc; FCs = fc; }                         \
                        if constexpr (FL == FL_FORWARD) {                                                    \
                            if constexpr (PARTIAL) { if (k >= j9) code = 0; }                                \
                            code4[J >> 2] |= (code & 0xffu) << (8 * (J & 3));                                \
                        }                                                                                    \
                        if constexpr (FL == FL_UDH && IMD) {                                                 \
                            /* scalar bookkeeping of the intermediate row (lane k8 of its stripe) */         \
                            const int n = n0 + J;                                                            \
                            const int rj = n - (ml + 1) - 2 * k;            /* my cell's diagonal */         \
                            if (imd_row && k == k8 && rj >= lw && rj <= up && n < n_end) {                   \
                                int* hl0 = imd_p + BIDX(rj);                                                 \
                                if (spj && is_acc) { st_b1<CROSS>(hl0, donor_r); st_b1<CROSS>(hl0 + width, donor_r + width); rlst = rj; } \
                                if (spj && is_don) donor_r = rj;                                             \
                                if (pb3 == 0) rlst = rj;                                                     \
                                if (pb3 == 1) st_b1<CROSS>(hl0, rlst);                                       \
                                st_b1<CROSS>(hl0 + 2 * width, Cs);  Cs = rj;                                 \
                                st_b1<CROSS>(hl0 + 3 * width, FCs); FCs = rj + width;                        \
                            }                                                                                \
                        }                                                                                    \
                        if (LOCAL && LocalR) {                                                               \
                            const int n = n0 + J;                                                            \
                            if (k < j9 && n < n_end && h >= best_val) {                                      \
                                const unsigned long long key =                                               \
                                    ((unsigned long long) s << 40) | ((unsigned long long) (n - n_start) << 8) | k; \
                                if (h > best_val || key < best_key) {                                        \
                                    best_val = h; best_key = key; best_mr = ml + k + 1; best_nr = n - k;     \
                                    if constexpr (FL == FL_UDH) { best_ulk = Cs; }                           \
                                }                                                                            \
                            }                                                                                \
                        }                                                                                    \
                        /* bottom lane of the stripe -> slot J of the row's output block */                  \
                        if (is_bottom) {                                                                     \
                            if constexpr (FL == FL_UDH) reinterpret_cast<int4*>(outb)[J] = make_int4(Hs, Fs, Cs, FCs); \
                            else reinterpret_cast<int2*>(outb)[J] = make_int2(Hs, Fs);                       \
                        }                                                                                    \
                    }
                    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
                    STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
                    // ---- flush: lane i holds the bottom-row result of step j = 15 - i; it goes to the
                    // boundary array under the reference's write condition (fwd2s1_wip_simd.h:205-209)
                    {
                        const int j = 15 - k;
                        const int n = n0 + j;
                        const int r0 = n - (ml + 1) - 2 * j8;
                        if (n - b_left >= j9 && r0 >= lw && r0 <= up && n < n_end && j9 > 0) {
                            if constexpr (FL == FL_UDH)
                                st_b4<CROSS>(bnd + (int64_t) BIDX(r0) * 4, reinterpret_cast<const int4*>(outb)[j]);
                            else
                                st_b2<CROSS>(bnd + (int64_t) BIDX(r0) * 2, reinterpret_cast<const int2*>(outb)[j]);
                        }
                    }
                    if constexpr (FL == FL_FORWARD) {
                        uint4* dst = reinterpret_cast<uint4*>(A.tb + my_tb + 256ll * lb + 16 * k);
                        *dst = make_uint4(code4[0], code4[1], code4[2], code4[3]);
                    }
                }
                // boundary entries are exchanged between the rows of this wave through memory: a
                // load issued after a store of the same wave to the same address observes it (in-order
                // vector memory path, loads bypass L1), so only the compiler needs a fence here
                WAVE_ORDER();
                if (W > 1) {                                            // publish: this block's stores are done
                    if constexpr (CROSS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0)
                        __hip_atomic_store(&s_prog[w], pass * BIGB + blk + 1, __ATOMIC_RELAXED, PSCOPE);
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
            const int last_own = ((n_passes - 1 - x) / W) * W + x;      // last pass of wave x (< 0: none)
            if (n_passes - 1 - x < 0) continue;
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
    // ---- fhlastS1 (src/fwd2s1_simd.cc:241-262) unless a local right end was tracked
    DevResult R;
    R.score = SPDP_NEV16; R.mr = a_right; R.nr = b_right; R.ml = a_left; R.ulk = END_OF_ULK; R.maxr = 0;
    R.pad[0] = R.pad[1] = 0;
    if (LOCAL && LocalR) {
        // reduce (value desc, key asc) over the wave
        for (int off = 32; off; off >>= 1) {
            const int ov = __shfl_xor(best_val, off);
            const unsigned long long ok = __shfl_xor(best_key, off);
            const int omr = __shfl_xor(best_mr, off), onr = __shfl_xor(best_nr, off);
            const int oml = __shfl_xor(best_ml, off), oul = __shfl_xor(best_ulk, off);
            if (ov > best_val || (ov == best_val && ok < best_key)) {
                best_val = ov; best_key = ok; best_mr = omr; best_nr = onr; best_ml = oml; best_ulk = oul;
            }
        }
        R.score = best_val; R.mr = best_mr; R.nr = best_nr; R.ml = best_ml; R.ulk = best_ulk;
    } else {
        const int rr = b_right - a_right;
        // first maximum over [lo, hi): returns index (lo if the range is empty)
        auto argmax_first = [&](int lo, int hi) {
            int bv = INT32_MIN, bi = INT32_MAX;
            for (int r = lo + lane; r < hi; r += 64) {
                const int v = ld_b1<CROSS>(&bnd[(int64_t) BIDX(r) * BW]);
                if (v > bv) { bv = v; bi = r; }
            }
            for (int off = 32; off; off >>= 1) {
                const int ov = __shfl_xor(bv, off), oi = __shfl_xor(bi, off);
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
            if (ld_b1<CROSS>(&bnd[(int64_t) BIDX(mv) * BW]) > ld_b1<CROSS>(&bnd[(int64_t) BIDX(maxr) * BW])) maxr = mv;
        }
        R.score = ld_b1<CROSS>(&bnd[(int64_t) BIDX(maxr) * BW]);
        if (maxr > rr) R.mr = b_right - maxr; else R.nr = a_right + maxr;
        if constexpr (FL == FL_UDH) R.ulk = ld_b1<CROSS>(&bnd[(int64_t) BIDX(maxr) * BW + 2]);
        R.maxr = maxr;
    }
    if (lane == 0) A.res[pi] = R;
#undef BIDX
}

// ---------------------------------------------------------------------------
// host-callable launchers (used by spdp_api.cpp)
template <int FL, bool LOCAL>
static void launch_nq(int nqm, dim3 grd, int wpb, hipStream_t stream, const SweepArgs& A)
{
    if constexpr (!LOCAL && FL == FL_UDH) {
        if (A.cross_g > 0) {
            // one problem over several CUs: every block must be resident, so this is a cooperative launch
            // (the runtime checks the grid against residency)
            SweepArgs Ac = A;
            void* kargs[] = {&Ac};
            const void* fn;
            if (wpb == 16)
                fn = nqm == NQ_FLAT  ? (const void*) spdp_sweep<FL, LOCAL, NQ_FLAT, 16, true>
                   : nqm == NQ_TABLE ? (const void*) spdp_sweep<FL, LOCAL, NQ_TABLE, 16, true>
                                     : (const void*) spdp_sweep<FL, LOCAL, NQ_CHAIN, 16, true>;
            else
                fn = nqm == NQ_FLAT  ? (const void*) spdp_sweep<FL, LOCAL, NQ_FLAT, 4, true>
                   : nqm == NQ_TABLE ? (const void*) spdp_sweep<FL, LOCAL, NQ_TABLE, 4, true>
                                     : (const void*) spdp_sweep<FL, LOCAL, NQ_CHAIN, 4, true>;
            (void) hipLaunchCooperativeKernel(fn, grd, dim3(wpb == 16 ? 1024 : 256), kargs, 0, stream);
            return;
        }
    }
    if constexpr (!LOCAL) {
        if (wpb == 16) {
            const dim3 blk(1024);
            switch (nqm) {
            case NQ_FLAT:  hipLaunchKernelGGL((spdp_sweep<FL, LOCAL, NQ_FLAT, 16>), grd, blk, 0, stream, A); break;
            case NQ_TABLE: hipLaunchKernelGGL((spdp_sweep<FL, LOCAL, NQ_TABLE, 16>), grd, blk, 0, stream, A); break;
            default:       hipLaunchKernelGGL((spdp_sweep<FL, LOCAL, NQ_CHAIN, 16>), grd, blk, 0, stream, A); break;
            }
            return;
        }
    }
    const dim3 blk(256);
    switch (nqm) {
    case NQ_FLAT:  hipLaunchKernelGGL((spdp_sweep<FL, LOCAL, NQ_FLAT, 4>), grd, blk, 0, stream, A); break;
    case NQ_TABLE: hipLaunchKernelGGL((spdp_sweep<FL, LOCAL, NQ_TABLE, 4>), grd, blk, 0, stream, A); break;
    default:       hipLaunchKernelGGL((spdp_sweep<FL, LOCAL, NQ_CHAIN, 4>), grd, blk, 0, stream, A); break;
    }
}

extern "C" hipError_t spdp_launch_sweep(int flavour, int local, int nquant, int pen_cap, const SweepArgs* args,
                                        int grid, int wpb, hipStream_t stream)
{
    SweepArgs A = *args;
    dim3 grd(grid);
    const int blk = (wpb == 16 && !local) ? 16 : 4;
    int nqm = nquant <= 1 ? NQ_FLAT : (pen_cap < SPDP_PEN_TAB ? NQ_TABLE : NQ_CHAIN);
    if (const char* e = getenv("SPDP_NQM")) { if (nqm != NQ_FLAT && atoi(e) == NQ_CHAIN) nqm = NQ_CHAIN; }
    switch (flavour * 2 + (local ? 1 : 0)) {
    case 0: launch_nq<FL_SCORE, false>(nqm, grd, blk, stream, A); break;
    case 1: launch_nq<FL_SCORE, true>(nqm, grd, blk, stream, A); break;
    case 2: launch_nq<FL_FORWARD, false>(nqm, grd, blk, stream, A); break;
    case 3: launch_nq<FL_FORWARD, true>(nqm, grd, blk, stream, A); break;
    case 4: launch_nq<FL_UDH, false>(nqm, grd, blk, stream, A); break;
    case 5: launch_nq<FL_UDH, true>(nqm, grd, blk, stream, A); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Traceback walk over the code buffer written by spdp_sweep<FL_FORWARD>:
// Anti_rhomb_coord<CHAR>::traceback / go_back (src/rhomb_coord.h:141-235) on
// our own layout -- per stripe, per 16-step block, 16 lanes x 16 codes.  One
// thread per problem; emits the reference's Mfile records (end -> start).

struct TbView {
    const uint8_t* tb; int64_t tb_off;
    int a_left, a_right, b_left, b_right, lw, up;
    bool a_exgl;
    // cached stripe
    int cs; int64_t cbase; int c_nstart, c_n9;
    __device__ void stripe_geom(int s, int& n_start, int& n9, int& nb) const {
        const int ml = a_left + s * SPDP_NELEM;
        const int j9 = min(SPDP_NELEM, a_right - ml);
        n_start = max(b_left, lw + ml);
        n9 = min(b_right, up + (ml + j9) + 1) + j9;
        const int len = max(0, n9 + 1 - n_start);
        nb = (len + 15) >> 4;
    }
    __device__ void seek(int s) {
        if (s == cs) return;
        int ns, n9, nb;
        if (cs >= 0 && s == cs - 1) {
            stripe_geom(s, ns, n9, nb);
            cbase -= 256ll * nb;
        } else {
            int64_t base = tb_off;
            for (int t = 0; t < s; ++t) { stripe_geom(t, ns, n9, nb); base += 256ll * nb; }
            stripe_geom(s, ns, n9, nb);
            cbase = base;
        }
        cs = s; c_nstart = ns; c_n9 = n9;
    }
    // code of window-relative cell (mp, np), mp >= 0, np >= 0
    __device__ unsigned at(int mp, int np) {
        if (mp == 0) return (!a_exgl && np >= 1) ? (unsigned) TB_HORI : 0u;
        const int s = (mp - 1) >> 4, k = (mp - 1) & 15;
        seek(s);
        const int n = np + b_left + k;               // sweep step that produced the cell
        if (n < c_nstart || n > c_n9) return 0u;
        const int tau = n - c_nstart;
        // a lane's 16 codes of one block are contiguous: keep the last 16-byte group in registers,
        // so walks along a row (introns, gaps) cost one load per 16 cells
        const int64_t grp = cbase + 256ll * (tau >> 4) + 16 * k;
        if (grp != c_grp) {
            c_w = *reinterpret_cast<const uint4*>(tb + grp);
            c_grp = grp;
        }
        const int j = tau & 15;
        const unsigned w = (j < 8) ? ((j < 4) ? c_w.x : c_w.y) : ((j < 12) ? c_w.z : c_w.w);
        return (w >> (8 * (j & 3))) & 0xffu;
    }
    int64_t c_grp; uint4 c_w;
};

// One WAVE per problem: all 64 lanes run the same walk on the same addresses (the loads
// broadcast), so a long intron scan of one problem never stalls 63 others the way
// one-thread-per-problem did in a divergent wave.
__global__ void spdp_walk(WalkArgs A)
{
    const int pi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pi >= A.n_probs) return;
    const bool writer = (threadIdx.x & 63) == 0;
    const bool seq_walk = A.seq != 0;
    const DevProblem P = A.probs[pi];
    TbView V;
    V.tb = A.tb; V.tb_off = P.tb_off;
    V.a_left = P.a_left; V.a_right = P.a_right; V.b_left = P.b_left; V.b_right = P.b_right;
    V.lw = P.lw; V.up = P.up; V.a_exgl = P.flags & 1; V.cs = -1; V.cbase = 0; V.c_nstart = V.c_n9 = 0;
    V.c_grp = -1; V.c_w = make_uint4(0, 0, 0, 0);
    int2* out = A.skl + (int64_t) pi * A.skl_cap;
    int cnt = 0, status = 0;
    int m = A.res[pi].mr - P.a_left, n = A.res[pi].nr - P.b_left;   // window-relative
    auto emit = [&]() {
        if (cnt < A.skl_cap) { if (writer) out[cnt] = make_int2(m + P.a_left, n + P.b_left); }
        else status = -1;
        ++cnt;
    };
    // to_left / to_upper of Anti_rhomb_coord (cur_m, cur_n track m, n)
    auto to_left = [&](int s) -> unsigned {
        n -= s;
        if (n < 0) { n = 0; return 0u; }
        return V.at(m, n);
    };
    auto to_upper = [&](int s) -> unsigned {
        --m; n -= s;
        if (m < 0) { m = 0; n += s; return 0u; }
        if (n < 0) { if (s > 0) m -= n / s; n = 0; return 0u; }
        return V.at(m, n);
    };
    unsigned code = (m >= 0 && n >= 0) ? V.at(m, n) : 0u;
    long guard = 4l * ((long) (P.a_right - P.a_left) + (P.b_right - P.b_left)) + 64;
    while (code && guard-- > 0) {
        emit();
        switch (code & 15u) {
        case TB_DIAG:
            // a diagonal run: one step changes row and column, so every step is a load of its own.  Inside a stripe
            // the next steps are known in advance (row k - j, sweep step n_step - 2 j): 16 lanes fetch them at once
            // and the first one that ends the run is taken; stripe changes and the matrix edges stay with the
            // one-step form.
            do {
                const int kk = (m - 1) & 15;
                const int cnt = (m >= 1 && !seq_walk) ? min(kk, n) : 0;      // steps that stay in my stripe and inside n >= 0
                if (cnt >= 2) {
                    V.seek((m - 1) >> 4);
                    const int j = (int) (threadIdx.x & 15);
                    const int tau = n - (j + 1) + V.b_left + (kk - (j + 1)) - V.c_nstart;
                    unsigned cj = 0u;
                    if (j < cnt && tau >= 0 && tau <= V.c_n9 - V.c_nstart)
                        cj = V.tb[V.cbase + 256ll * (tau >> 4) + 16 * (kk - (j + 1)) + (tau & 15)];
                    const bool stop = j < cnt && (cj == 0u || (cj & 15u) != TB_DIAG);
                    const unsigned hit = (unsigned) (__ballot(stop) & 0xffffull);
                    const int js = hit ? __builtin_ctz(hit) : cnt - 1;
                    m -= js + 1; n -= js + 1;
                    code = (unsigned) __shfl((int) cj, js, 16);
                } else
                    code = to_upper(1);
            } while (code && (code & 15u) == TB_DIAG);
            break;
        case TB_HORI: {
            bool dead = false;
            while (!(code & TB_NHOR)) { code = to_left(1); if (!code) { dead = true; break; } }
            if (!dead) code = to_left(1);
            break;
        }
        case TB_VERT: {
            bool dead = false;
            while (!(code & TB_NVER)) { code = to_upper(0); if (!code) { dead = true; break; } }
            if (!dead) code = to_upper(0);
            break;
        }
        case TB_ACCR: {
            // across an intron: left along row m to the first cell that carries the donor mark (or is dead).  One load
            // per 16 cells of a 20 kb intron is a chain of 1250 dependent loads; the 64 lanes fetch the next 64 groups
            // (1024 cells) at once instead and the nearest stop among them is taken.  Groups that reach beyond the
            // stripe's first step or the window's first column stay with the one-step form below.
            bool found = false;
            while (!seq_walk && m >= 1) {
                const int k = (m - 1) & 15;
                V.seek((m - 1) >> 4);
                const int t_here = n + V.b_left + k - V.c_nstart;
                const int t_min = max(0, V.b_left + k - V.c_nstart);          // n >= 0 and inside the stripe
                if (t_here > V.c_n9 - V.c_nstart + 1 || t_here - 1 < t_min + 16) break;
                const int g0 = (t_here - 1) >> 4, jmax = (t_here - 1) & 15;
                const int L = (int) (threadIdx.x & 63);
                const int g = g0 - L;
                const bool valid = g >= 0 && 16 * g >= t_min;
                uint4 w4 = make_uint4(0, 0, 0, 0);
                if (valid) w4 = *reinterpret_cast<const uint4*>(V.tb + V.cbase + 256ll * g + 16 * k);
                int best = -1; unsigned bcode = 0;
                const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const unsigned w = ws[wi];
                    unsigned sb = (~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu)) | (w & 0x80808080u);   // byte == 0, or donor mark
                    if (L == 0) {
                        if (4 * wi > jmax) sb = 0u;
                        else if (4 * wi + 3 > jmax) sb &= (1u << (8 * ((jmax & 3) + 1))) - 1u;
                    }
                    if (valid && sb) { const int bj = (31 - __builtin_clz(sb)) >> 3; best = 4 * wi + bj; bcode = (w >> (8 * bj)) & 0xffu; }
                }
                const unsigned long long stop_b = __ballot(best >= 0), inval_b = __ballot(!valid);
                const int first_stop = stop_b ? __builtin_ctzll(stop_b) : 64;
                const int first_inval = inval_b ? __builtin_ctzll(inval_b) : 64;
                if (first_stop < first_inval) {
                    const int bj = __shfl(best, first_stop, 64);
                    code = (unsigned) __shfl((int) bcode, first_stop, 64);
                    n = 16 * (g0 - first_stop) + bj - V.b_left - k + V.c_nstart;
                    found = true;
                    break;
                }
                // nothing in the valid groups: go to the leftmost cell looked at (alive, no mark) and look again from there
                const int last = first_inval - 1;
                if (last < 0) break;
                n = 16 * (g0 - last) - V.b_left - k + V.c_nstart;
            }
            if (!found) do { code = to_left(1); } while (code && !(code & TB_DONR));
            else { V.c_grp = -1; }
            break;
        }
        default:
            status = -2; code = 0;
            break;
        }
    }
    emit();
    if (writer) A.n_skl[pi] = status ? status : cnt;
}

// ---------------------------------------------------------------------------
// Back-walk over the UDH links: tail of hirschbergS1_wip, src/fwd2s1_wip_simd.h:814-863
// (non-local ends).  One thread per problem.

__global__ void spdp_udh_cpos(CposArgs A)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= A.n_probs) return;
    const DevProblem P = A.probs[pi];
    const DevResult R = A.res[pi];
    const int n_im = P.n_im, lw = P.lw, up = P.up, width = P.width;
    const int step = (P.a_right - P.a_left + n_im) / (n_im + 1);
    const int* imd0 = A.imd + P.imd_off;
    int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
#define MI(i) (P.a_left + ((i) + 1) * step)
#define LNK(i, which, d, r) imd0[((int64_t) (i) * 4 + (which) * 2 + (d)) * width + ((r) - lw + 1)]
    for (int i = 0; i <= n_im; ++i) {
        for (int c = 0; c < 10; ++c) CPOS(i, c) = 0;
        CPOS(i, 0) = END_OF_ULK; CPOS(i, 2) = END_OF_ULK;
    }
    int a_left = P.a_left, b_left = P.b_left;
    const int a_right = R.mr, b_right = R.nr;
    const int max_ml = A.local ? R.ml : P.a_left;
    int val = R.score;
    int i = n_im;
    while (--i >= 0 && MI(i) > a_right) ;
    if (i < 0 && MI(0) > a_right) CPOS(0, 2) = b_right;
    int r = R.ulk;
    int edge = 0;
    for ( ; i >= 0 && MI(i) > max_ml; --i) {
        int c = 0, d = 0;
        if (A.strict) { for ( ; r > up; r -= width) ++d; }
        else { for ( ; r >= up; r -= width) ++d; }
        const int vl = LNK(i, 1, d, r);
        if (r + MI(i) <= P.b_left) edge = 1;            // the crossing lies on the boundary column or left of it
        if ((A.strict ? lw <= vl : lw < vl) && vl < up) {
            CPOS(i, c++) = MI(i);
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            // (a pipelined sweep leaves a marker where it stored hs1.rlst of the rows above: what they ended with)
            auto hl = [&](int ii, int dd, int rr) {
                int v = LNK(ii, 0, dd, rr);
                if (A.pipe && v == SPDP_RLST_INHERITED) {
                    const int* rlf = A.pipe + (size_t) pi * A.pipe_stride + A.rlf_off;
                    v = 0x7fffffff;
                    for (int j = ii - 1; j >= 0; --j) if (rlf[j] != SPDP_RLST_INHERITED) { v = rlf[j]; break; }
                }
                return v;
            };
            for (int rp = hl(i, d, r); lw <= rp && rp < up && r != rp; rp = hl(i, d, r = rp)) {
                if (c < 8) CPOS(i, c++) = r + MI(i); else ++c;
            }
            if (c < 9) { CPOS(i, c++) = r + MI(i); CPOS(i, c) = END_OF_ULK; }
            r = LNK(i, 1, d, r);
            if (r == END_OF_ULK) break;
        } else
            CPOS(i, 0) = END_OF_ULK;
    }
    for ( ; r > up; r -= width) ;
    if (A.local && (P.flags & 1) && (P.flags & 4)) {    // LocalL: the path's own left end
        a_left = max_ml;
        b_left = r + a_left;
    } else {
        const int rl = b_left - a_left;
        const bool a_exgl = P.flags & 1, b_exgl = P.flags & 4;
        if (b_exgl && rl > r) {
            a_left = b_left - r;
            for (int j = 0; j < n_im && MI(j) < a_left; ++j) CPOS(j, 0) = END_OF_ULK;
        }
        if (a_exgl && rl < r) b_left = a_left + r;
    }
    ++i;
    if ((i < n_im && MI(i) < a_left) || CPOS(i, 2) < b_left) val = INT32_MIN / 16 * 7;
    if (a_left >= a_right || b_left >= b_right) edge = 1;      // an empty optimum
    if (A.edge) A.edge[pi] = edge;
    A.scores[pi] = val;
    int* rg = A.ranges + 4 * pi;
    rg[0] = a_left; rg[1] = a_right; rg[2] = b_left; rg[3] = b_right;
#undef CPOS
#undef MI
#undef LNK
}

extern "C" hipError_t spdp_launch_walk(const WalkArgs* a, hipStream_t stream)
{
    WalkArgs A = *a;
    hipLaunchKernelGGL(spdp_walk, dim3((A.n_probs + 3) / 4), dim3(256), 0, stream, A);
    return hipGetLastError();
}
extern "C" hipError_t spdp_launch_cpos(const CposArgs* a, hipStream_t stream)
{
    CposArgs A = *a;
    hipLaunchKernelGGL(spdp_udh_cpos, dim3((A.n_probs + 63) / 64), dim3(64), 0, stream, A);
    return hipGetLastError();
}

// gathers the per-problem record slots of spdp_walk into one contiguous array
__global__ void spdp_pack_skl(const int2* skl, int skl_cap, const int* n_skl, const int64_t* off,
                              int2* packed, int n_probs)
{
    const int pi = blockIdx.x;
    if (pi >= n_probs) return;
    const int cnt = n_skl[pi];
    const int2* src = skl + (int64_t) pi * skl_cap;
    int2* dst = packed + off[pi];
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) dst[i] = src[i];
}
extern "C" hipError_t spdp_launch_pack(const int2* skl, int skl_cap, const int* n_skl, const int64_t* off,
                                       int2* packed, int n_probs, hipStream_t stream)
{
    hipLaunchKernelGGL(spdp_pack_skl, dim3(n_probs), dim3(64), 0, stream, skl, skl_cap, n_skl, off, packed, n_probs);
    return hipGetLastError();
}
