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
// row_shr-style shuffles inside their 16-lane group.  The stripes of a problem run as a pipeline of waves
// (spdp_exact<MODE, true>, below) or, <MODE, false>, one after the other in one group.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"

#define XN 16                                    // rows per stripe (SPDP_NELEM)
#define XNEV SPDP_NEV16
// post-splice flag of a state (src/aln.h:56): H 4, E 1, F 8 -- arithmetic, not a table in memory
__device__ __forceinline__ int x_psp_bit_of(int d) { return d == 0 ? 4 : (d == 1 ? 1 : (d == 2 ? 8 : (d == 3 ? 2 : 16))); }    // H, E, F, E2, F2

__device__ __forceinline__ int x_sadd(int a, int b) { return max(a + b, SPDP_FLOOR16); }
__device__ __forceinline__ int x_up(int v) { return __shfl_up(v, 1, XN); }      // lane k <- lane k - 1 of its group

// FORWARD: forwardS1 (src/fwd2s1_simd.cc:481-773) -- the same sweep with a Vmf pointer riding on H / E / F (one
// int here; modes 3 / 5 of the reference split it over int16 lanes), a diagonal flag per cell, a record at
// the start of every diagonal run and two per accepted intron (appended through a per-problem atomic
// counter: record numbers differ from the reference's, the chains do not), then Vmf::traceback and the
// fix-up of trcbkalignS_ng by lane 0.
// MODE 2: hirschbergS1 (src/fwd2s1_simd.cc:775-1150, non-local) -- the pointer lanes carry links (the diagonal at
// which the path crossed the previous intermediate row); the lane holding an intermediate row stores and
// restarts them; spdp_udh_cpos (strict form) walks them back afterwards.
// PIPE: the stripes of a problem run as a pipeline.  A work item is (four problems, stripe): group g of the wave
// sweeps that stripe of problem 4 q + g, the wave of the next item follows ~50 steps behind.  Items are drawn from
// a ticket counter in dispatch order, so a stripe's predecessor is always resident or done.  The boundary arrays
// cross CUs (agent-scope accesses, memory side); a stripe publishes, once its stores have drained, the diagonal
// up to which its bottom row is out (prog[stripe], + 2^20; INT_MAX = finished), and reads its predecessor's word
// before every block of 16 steps it stages.  What the one-wave form carries from stripe to stripe in registers goes
// through memory: the local maximum (per stripe, first maximum in stripe order), the intermediate-row counter
// (recomputed), and hs1.rlst -- only ever stored, so a stripe starts from a marker (XINH) and the link walk
// (spdp_udh_cpos) replaces it by what the intermediate rows above left (rlf[]).
#define XINH SPDP_RLST_INHERITED
#define XPROG0 (1 << 28)
#define XVCH SPDP_VMF_LANE_CHUNK
#define XWPB 4                                   // waves per block: they share the read-only tables in LDS
template <bool X> __device__ __forceinline__ int x_ld(const int* p)
{
    if constexpr (X) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __builtin_nontemporal_load(p);
}
template <bool X> __device__ __forceinline__ void x_st(int* p, int v)
{
    if constexpr (X) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// DAGP: double affine gaps (PwdB::Noll = 3, -yl3; round 5): the second horizontal / vertical gap states E2 / F2 priced with
// LongGOP / LongGEP (src/fwd2s1_simd.cc:347-352, 368-378 / 556-569, 592-611), the better gap of each pair competes for the cell
// (:399-404 / :641-654), five states a donor candidate can leave from (hfesv[..][0..4], src/fwd2s1_simd.h:270-274) and NCAND + 2
// candidates per lane (:197), a third boundary array (fv2, and fc2 with pointers).  Score-only and forward engines: the
// reference's own hirschbergS1 is not usable under -yl3, the ladder refuses that branch.
template <int MODE, bool PIPE, bool DAGP = false>
__global__ void __launch_bounds__(64 * XWPB) __attribute__((amdgpu_waves_per_eu(3))) spdp_exact(ScalarArgs A)
{
    static_assert(!DAGP || MODE != 2, "hirschbergS1 with double affine gaps is not defined by the reference");
    constexpr int NCX = DAGP ? 6 : 4;           // Ncand: the list holds NCX + 1 entries
    constexpr int NOD = DAGP ? 5 : 3;           // states a candidate can leave from
    constexpr bool FORWARD = MODE == 1;         // Vmf records, diagonal flags
    constexpr bool UDH = MODE == 2;             // links, intermediate rows
    constexpr bool PTR = MODE != 0;             // a pointer / link rides on H, E, F
    // what a step needs from memory is staged through LDS a block of 16 steps ahead (per 16-lane group): the column
    // records and class bytes of the block's columns, and the boundary entries lane 0 feeds from (the previous stripe's
    // bottom row, by diagonal) -- a step used to wait for its own loads
    __shared__ int s_mtx[32 * 32];
    __shared__ int2 s_col[4 * XWPB][64];
    __shared__ unsigned short s_ax[4 * XWPB][64];
    enum { FD_HV, FD_FV, FD_HC, FD_FC, FD_HB, FD_FB, FD_FV2, FD_FC2, FD_N };
    __shared__ int s_fd[4 * XWPB][FD_N][20];
    // the tables an acceptor prices its candidates with: a read from memory inside that loop stalled the whole wave
    // (some lane of 64 sits on an acceptor column at almost every step)
    __shared__ short s_ipen[4096];
    __shared__ IpenRuns s_runs;                 // IntPen beyond s_ipen (spdp_ipen_runs.h)
    __shared__ short s_t53[256];
    for (int i = threadIdx.x; i < 32 * 32; i += 64 * XWPB) s_mtx[i] = A.sc->mtx[i];
    for (int i = threadIdx.x; i < 4096; i += 64 * XWPB) s_ipen[i] = A.intpen[min(i, A.intpen_len - 1)];
    ipen_runs_load(s_runs, A.ipen_runs);
    for (int i = threadIdx.x; i < 256; i += 64 * XWPB) s_t53[i] = A.t53[i];
    __shared__ int s_gain[2];                   // max IntPen, max junction-pair score: what an acceptor can add at most
    if (threadIdx.x < 2) s_gain[threadIdx.x] = INT32_MIN;
    __syncthreads();
    {
        int pm = INT32_MIN, tm = INT32_MIN;
        for (int i = threadIdx.x; i < A.intpen_len; i += 64 * XWPB) pm = max(pm, (int) A.intpen[i]);
        for (int i = threadIdx.x; i < 256; i += 64 * XWPB) tm = max(tm, (int) A.t53[i]);
        atomicMax(&s_gain[0], pm); atomicMax(&s_gain[1], tm);
    }
    __syncthreads();                            // (before any group leaves)
    const int k = threadIdx.x & 15;
    const int grp = (threadIdx.x & 63) >> 4;     // my 16-lane group in the wave
    const int g16 = threadIdx.x >> 4;            // ... in the block (its staging rings)
    int pi = (blockIdx.x * XWPB + (threadIdx.x >> 6)) * 4 + grp;
    int my_stripe = -1;                          // PIPE: the one stripe this group sweeps
    if (PIPE) {
        int tk = 0;
        if ((threadIdx.x & 63) == 0) tk = __hip_atomic_fetch_add(A.pipe + A.pipe_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= A.n_items) return;
        const int2 it = A.items[tk];
        pi = it.x * 4 + grp; my_stripe = it.y;
    }
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
    const int ge2 = DAGP ? A.lgep : 0, gn2 = DAGP ? A.lgep + A.lgop : 0, lgop = DAGP ? A.lgop : 0;
    const int minl = A.minl, ipen = A.ipen;
    const uint8_t* acod = A.a_codes + P.a_off;
    const int2* cols = A.cols + P.col_off;       // .x = (sig5 + ipen) | sig3 << 16, .y = b[n - 1]
    const uint8_t* aux = A.aux + 2 * P.col_off;  // {bit0 donor | bit1 acceptor, dinc5 << 4 | dinc3}
    int* hv = A.work + P.bnd_off - lw + 1;       // by diagonal, in place like the reference's hv / fv
    int* fv = hv + P.buf_size;
    const int width = P.width;
    int* imd0 = A.imd + (UDH ? P.imd_off : 0);   // udh: hlnk[2], vlnk[2] per intermediate, `width` ints each
    auto LNK = [&](int i, int which, int d, int r) -> int& { return imd0[((int64_t) i * 4 + which * 2 + d) * width + (r - lw + 1)]; };
    const int n_im = UDH ? P.n_im : 0;
    const int imd_step = UDH ? (a_right - a_left + n_im) / (n_im + 1) : 0;
    int* hb = fv + P.buf_size;                   // forward: diagonal flag, pointers of H and F
    int* hc = hb + P.buf_size;
    int* fc = hc + P.buf_size;
    int* vcount = A.work + P.bnd_off + 5 * (int64_t) P.buf_size;          // forward: records appended so far
    int* fb = fc + P.buf_size;                   // udh with local left ends: left-end row (`ml`) of F (hb: of H)
    int* fv2 = hv + 6 * (int64_t) P.buf_size;    // DAGP: the second vertical gap by diagonal, and its pointer
    int* fc2 = hv + 7 * (int64_t) P.buf_size;
    int3* vrec = A.vmf + P.tb_off;
    int* vraw = reinterpret_cast<int*>(vrec);
    const int vcap = (int) P.imd_off;
    // every lane takes its record numbers XVCH at a time from the problem's counter (an atomic per record made the
    // forward sweep wait for memory at every diagonal run it started)
    // (the next chunk is asked for while half of the current one is still there: the counter's round trip runs under the steps
    //  in between)
    int v_next = 0, v_left = 0, v_pend = 0;
    bool v_asked = false;
    auto vadd = [&](int mm, int nn, int pp) -> int {
        if (v_left == 0) {
            if (!v_asked) v_pend = atomicAdd(vcount, XVCH);
            v_next = v_pend; v_left = XVCH; v_asked = false;
        }
        const int i = v_next++;
        --v_left;
        if (v_left == XVCH / 2 && !v_asked) { v_pend = atomicAdd(vcount, XVCH); v_asked = true; }
        if (i < vcap) {
            if (PIPE) { x_st<true>(vraw + 3 * i, mm); x_st<true>(vraw + 3 * i + 1, nn); x_st<true>(vraw + 3 * i + 2, pp); }
            else vrec[i] = make_int3(mm, nn, pp);
        }
        return i;
    };
    const int n_ent = P.buf_size;
    const int n_stripes = max(1, (a_right - a_left + XN - 1) / XN);     // (DevRun::build lists the same count)
    if (PIPE && my_stripe >= n_stripes) return;  // (a shorter problem of the four)
    // PIPE: what the stripes of the problem share: prog[max_tiles], best[max_tiles][6], rlf[n_im]
    int* sy = PIPE ? A.pipe + (size_t) pi * A.pipe_stride : nullptr;
    int* prog = PIPE ? sy + 2 : nullptr;
    int* tbest = PIPE ? sy + 2 + A.max_tiles : nullptr;
    int* rlf = PIPE ? sy + 2 + 7 * A.max_tiles : nullptr;
    bool stalled = false;
    // waits until stripe t has published at least `req` (per lane: a group waits for its own problem)
    auto wait_for = [&](int t, int req) {
        long spins = 0;
        while (!stalled && __hip_atomic_load(prog + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < req) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1l << 22)) {              // (cannot happen with the ticket order; bounds every spin)
                __hip_atomic_store(A.pipe + A.pipe_ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stalled = true;
            }
        }
    };
    auto publish = [&](int t, int v) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (k == 0) __hip_atomic_store(prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- fhinitS1
    if (!PIPE || my_stripe == 0) {
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
            x_st<PIPE>(&hv[r], h); x_st<PIPE>(&fv[r], XNEV);
            if constexpr (DAGP) x_st<PIPE>(&fv2[r], XNEV);
            if constexpr (FORWARD) {                 // the Vmf part of fhinitS1 (:185-205): records 0 (dummy) and 1 (start)
                const int ru = up + 2 * XN;
                int c = 0;
                if (r == rl) c = 1;
                else if (r > rl && r <= ru) c = a_exgl ? 0 : 1;
                else if (r < rl) c = b_exgl ? 0 : 1;
                x_st<PIPE>(&hb[r], 0); x_st<PIPE>(&hc[r], c); x_st<PIPE>(&fc[r], c);
                if constexpr (DAGP) x_st<PIPE>(&fc2[r], c);              // (src/fwd2s1_simd.cc:236)
            }
            if constexpr (UDH) {                     // the Hirschberg part (:206-227): link = diagonal where the path starts
                const int ru = up + 2 * XN;
                int c = 0;
                if (r >= rl) { if (a_exgl) c = (r < ru) ? r : 0; else c = (r <= ru) ? rl : 0; }
                else c = b_exgl ? r : rl;
                x_st<PIPE>(&hc[r], c); x_st<PIPE>(&fc[r], c);
                int bm = a_left;                     // bbuf = a_left; the free left column counts rows upwards (:209-224)
                if (b_exgl && r <= rl && r >= lw) bm = a_left + (rl - r);
                x_st<PIPE>(&hb[r], bm); x_st<PIPE>(&fb[r], a_left);
            }
        }
        if constexpr (UDH) {
            if (!PIPE) for (int e = k; e < n_im * 4 * width; e += XN) imd0[e] = 0x7fffffff - 2;    // end_of_ulk
        }
        if constexpr (FORWARD) {
            if (k == 0) {
                x_st<PIPE>(vraw, 0); x_st<PIPE>(vraw + 1, 0); x_st<PIPE>(vraw + 2, 0);
                x_st<PIPE>(vraw + 3, a_left); x_st<PIPE>(vraw + 4, b_left); x_st<PIPE>(vraw + 5, 0);
                x_st<PIPE>(vcount, 2);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    int maxh = XNEV, max_ulk = 0, max_mr = a_right, max_nr = b_right, max_ml = a_left;
    int imd_i = 0, rlst = 0x7fffffff;            // udh: current intermediate, hs1.rlst
    const int ml_first = PIPE ? a_left + XN * my_stripe : a_left;
    const int ml_end = PIPE ? min(a_right, ml_first + XN) : a_right;
    if (PIPE && UDH) {
        // the counter as the stripes above would have left it: one step per stripe that held the then current row
        for (int mq = a_left; mq < ml_first && imd_i < n_im; mq += XN) {
            const int mi = a_left + (imd_i + 1) * imd_step;
            if (mq == a_left + (mi - a_left - 1) / XN * XN) ++imd_i;
        }
    }
    for (int ml = ml_first; ml < ml_end; ml += XN) {
        const int j9 = min(XN, a_right - ml);
        const int j8 = j9 - 1;
        int n = max(b_left, lw + ml);
        const int n_first = n;
        const int n9 = min(b_right, up + (ml + j9) + 1) + j9;
        int r = n - (ml + 1);
        // per-lane state: H of the last two steps, F, E, flags, the candidate list of my row
        int H1 = XNEV, H2 = XNEV, F1 = XNEV, E = XNEV, ps = 0;
        int B1 = 0, B2 = 0, C1 = 0, C2 = 0, FC1 = 0, EC = 0, EB = 0, FB = 0;    // forward: flags / pointers of H (two steps), F, E
        int E2 = XNEV, EC2 = 0, EB2 = 0, F21 = XNEV, FC21 = 0, FB2 = 0;        // DAGP: the long gaps (F21 / FC21: of my last step, for the lane below)
        // the donor candidates of my row, best first, in registers: slots are moved, never indexed by a variable (a
        // run-time index into five register arrays costs a select chain per access, and the wave pays for it at every step
        // on which ANY of its 64 lanes sits on a donor or acceptor column)
        int c_val[NCX + 1], c_jnc[NCX + 1], c_dir[NCX + 1], c_ml[NCX + 1], c_ulk[NCX + 1], c_dn5[NCX + 1], ncand = -1;      // c_dn5: dinc5 of the donor column
#pragma unroll
        for (int i = 0; i <= NCX; ++i) { c_val[i] = XNEV; c_jnc[i] = c_dir[i] = c_ml[i] = c_ulk[i] = c_dn5[i] = 0; }
        const int m = ml + 1 + k;                             // my row
        const int sigJ_cip = (A.cip && P.cip_off >= 0 && m <= P.a_right) ? A.cip[P.cip_off + m] : 0;     // Cip_score::cip_score(m), fwd2s1_simd.cc:50
        // udh: is the current intermediate row in this stripe, and on which lane
        int mm_ = 0, k9 = 0, k8 = -1;
        bool is_imd_ = false;
        if constexpr (UDH) {
            if (imd_i < n_im) {
                const int mi = a_left + (imd_i + 1) * imd_step;
                mm_ = a_left + (mi - a_left - 1) / XN * XN;
                k9 = mi - mm_; k8 = k9 - 1;
                is_imd_ = ml == mm_;
            }
        }
        const bool imd_here = UDH && is_imd_ && k == k8;     // my row is the intermediate row (m == imd->mi)
        (void) k9;
        if (PIPE && UDH && is_imd_) {
            for (int e = k; e < 4 * width; e += XN) x_st<true>(imd0 + (int64_t) imd_i * 4 * width + e, 0x7fffffff - 2);
            rlst = imd_i == 0 ? 0x7fffffff : XINH;
        }
        const int st = PIPE ? my_stripe : 0;
        // PIPE: entries up to diagonal `rq` of the stripe above must be out before they are staged
        auto ready = [&](int rq) { if (PIPE && st > 0) wait_for(st - 1, rq + XPROG0); };
        const int* mrow = s_mtx + ((k < j9) ? acod[ml + k] : 0) * 32;
        // ---- staging (see the top of the kernel): registers hold the NEXT block's loads while a block runs
        int2* const ring = s_col[g16];
        unsigned short* const ringx = s_ax[g16];
        int (*const fd)[20] = s_fd[g16];
        const int e_last = lw - 1 + P.buf_size - 1;                       // last entry of the boundary arrays
        auto ld_col = [&](int c, int2& cc, unsigned& ax) {
            const bool in = c >= 0 && c <= b_right + 1;
            cc = in ? cols[c] : make_int2(0, 0);
            ax = in ? reinterpret_cast<const unsigned short*>(aux)[c] : 0u;
        };
        int2 pc = make_int2(0, 0); unsigned pax = 0; int pfd[FD_N] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto prefetch = [&](int nb_, int rb_) {                           // block starting at step nb_, diagonal rb_
            ld_col(nb_ + k, pc, pax);
            const int e = min(rb_ + 1 + k, e_last);
            pfd[FD_HV] = x_ld<PIPE>(&hv[e]); pfd[FD_FV] = x_ld<PIPE>(&fv[e]);
            if constexpr (PTR) { pfd[FD_HC] = x_ld<PIPE>(&hc[e]); pfd[FD_FC] = x_ld<PIPE>(&fc[e]); }
            if constexpr (FORWARD) pfd[FD_HB] = x_ld<PIPE>(&hb[e]);
            if constexpr (UDH) { if (LocalL) { pfd[FD_HB] = x_ld<PIPE>(&hb[e]); pfd[FD_FB] = x_ld<PIPE>(&fb[e]); } }
            if constexpr (DAGP) { pfd[FD_FV2] = x_ld<PIPE>(&fv2[e]); if constexpr (PTR) pfd[FD_FC2] = x_ld<PIPE>(&fc2[e]); }
        };
        auto commit = [&](int nb_) {                                      // the staged block becomes the current one
            ring[(nb_ + k) & 63] = pc; ringx[(nb_ + k) & 63] = (unsigned short) pax;
#pragma unroll
            for (int a = 0; a < FD_N; ++a) {
                const int carry = fd[a][16];                              // entry rb of the new block = entry rb + 16 of the old one
                fd[a][k + 1] = pfd[a];
                if (k == 0) fd[a][0] = carry;
            }
        };
        {
            // the 16 columns left of the first step (lane k of step n_first looks at n_first - k), then block 0
            int2 c0; unsigned a0;
            ready(r + 16);
            ld_col(n_first - 16 + k, c0, a0);
            ring[(n_first - 16 + k) & 63] = c0; ringx[(n_first - 16 + k) & 63] = (unsigned short) a0;
            if (k == 0) {
                const int e = min(r, e_last);
                fd[FD_HV][16] = x_ld<PIPE>(&hv[e]); fd[FD_FV][16] = x_ld<PIPE>(&fv[e]);
                if constexpr (PTR) { fd[FD_HC][16] = x_ld<PIPE>(&hc[e]); fd[FD_FC][16] = x_ld<PIPE>(&fc[e]); }
                if constexpr (FORWARD) fd[FD_HB][16] = x_ld<PIPE>(&hb[e]);
                if constexpr (UDH) { if (LocalL) { fd[FD_HB][16] = x_ld<PIPE>(&hb[e]); fd[FD_FB][16] = x_ld<PIPE>(&fb[e]); } }
                if constexpr (DAGP) { fd[FD_FV2][16] = x_ld<PIPE>(&fv2[e]); if constexpr (PTR) fd[FD_FC2][16] = x_ld<PIPE>(&fc2[e]); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            prefetch(n, r);
        }
        // PIPE: a stripe is finished the moment its last step is done, not when the wave leaves the loop -- the other
        // groups of the wave sweep other problems and may go on for thousands of steps, and the stripes below this one
        // wait for the word (tools/ladder_case.py: two problems in one launch ran as slowly as unpipelined before)
        auto finish_stripe = [&]() {
            if (LocalR && k == 0) {
                int* b = tbest + 6 * st;
                x_st<true>(b, maxh); x_st<true>(b + 1, max_ulk); x_st<true>(b + 2, max_mr); x_st<true>(b + 3, max_nr); x_st<true>(b + 4, max_ml);
            }
            if (st > 0) wait_for(st - 1, INT32_MAX);                      // finished = all stripes up to this one are
            publish(st, INT32_MAX);
        };
        if (PIPE && n >= n9) finish_stripe();
        int jb = 16;                                                      // step within the block; 16 = a new block starts
        for ( ; n < n9; ++n, ++r) {
            if (jb == 16) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (this wave's reads of the old block are done)
                commit(n);
                if (PIPE) { publish(st, r - 1 - 2 * j8 + XPROG0); ready(r + 32); }
                prefetch(n + 16, r + 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                jb = 0;
            }
            const int j = jb++;
            const int r0 = r - 2 * j8;
            const int ke = min(j9, n - b_left);
            const int nj = n - k;                             // my column
            // boundary feeds of lane 0 (previous stripe's bottom row, by diagonal): entry r is fd[..][j], r + 1 is fd[..][j + 1]
            int bH1 = 0, bF1 = 0, bH2 = 0, bC1 = 0, bFC1 = 0, bB2 = 0, bC2 = 0, bB1 = 0, bFB1 = 0, bF21 = 0, bFC21 = 0;
            if (k == 0) {
                bH1 = fd[FD_HV][j + 1]; bF1 = fd[FD_FV][j + 1]; bH2 = fd[FD_HV][j];
                if constexpr (DAGP) { bF21 = fd[FD_FV2][j + 1]; if constexpr (PTR) bFC21 = fd[FD_FC2][j + 1]; }
                if constexpr (PTR) { bC1 = fd[FD_HC][j + 1]; bFC1 = fd[FD_FC][j + 1]; bC2 = fd[FD_HC][j]; }
                if constexpr (FORWARD) bB2 = fd[FD_HB][j];
                if constexpr (UDH) { if (LocalL) { bB2 = fd[FD_HB][j]; bB1 = fd[FD_HB][j + 1]; bFB1 = fd[FD_FB][j + 1]; } }
            }
            int upH1 = x_up(H1), upF1 = x_up(F1), upH2 = x_up(H2);
            int upC1 = 0, upFC1 = 0, upB2 = 0, upC2 = 0;
            if constexpr (PTR) { upC1 = x_up(C1); upFC1 = x_up(FC1); upC2 = x_up(C2); }
            if constexpr (FORWARD) upB2 = x_up(B2);
            int upB1 = 0, upFB1 = 0;                          // udh, local left ends: `ml` of the lane above
            if constexpr (UDH) { if (LocalL) { upB2 = x_up(B2); upB1 = x_up(B1); upFB1 = x_up(FB); } }
            int upF21 = XNEV, upFC21 = 0;
            if constexpr (DAGP) { upF21 = x_up(F21); if constexpr (PTR) upFC21 = x_up(FC21); }
            if (k == 0) { upH1 = bH1; upF1 = bF1; upH2 = bH2; upC1 = bC1; upFC1 = bFC1; upB2 = bB2; upC2 = bC2; upB1 = bB1; upFB1 = bFB1; upF21 = bF21; upFC21 = bFC21; }
            // insertion, deletion, diagonal
            {
                const int open = x_sadd(H1, gn), ext = x_sadd(E, ge);
                const bool m_ = ext > open;
                E = m_ ? ext : open;
                if constexpr (PTR) EC = m_ ? EC : C1;
                if constexpr (UDH) { if (LocalL) EB = m_ ? EB : B1; }
            }
            if constexpr (DAGP) {                             // the long horizontal gap (:347-352 / :556-569)
                const int open = x_sadd(H1, gn2), ext = x_sadd(E2, ge2);
                const bool m_ = ext > open;
                E2 = m_ ? ext : open;
                if constexpr (PTR) EC2 = m_ ? EC2 : C1;
            }
            int F, FC = 0;
            {
                const int open = x_sadd(upH1, gn), ext = x_sadd(upF1, ge);
                const bool m_ = ext > open;
                F = m_ ? ext : open;
                if constexpr (PTR) FC = m_ ? upFC1 : upC1;
                if constexpr (UDH) { if (LocalL) FB = m_ ? upFB1 : upB1; }
            }
            int F2 = XNEV, FC2 = 0;
            if constexpr (DAGP) {                             // the long vertical gap (:368-375 / :592-608)
                const int open = x_sadd(upH1, gn2), ext = x_sadd(upF21, ge2);
                const bool m_ = ext > open;
                F2 = m_ ? ext : open;
                if constexpr (PTR) FC2 = m_ ? upFC21 : upC1;
            }
            int pv = 0;
            const bool incell = nj <= b_right && nj > b_left && k < j9;     // kb <= k < ke
            const int2 col = ring[nj & 63];
            const unsigned axj = ringx[nj & 63];
            if (incell) pv = mrow[col.y];
            int H = x_sadd(pv, upH2);
            int HC = upC2;
            int code = 0;                                     // diag: 0, hori: 1, vert: 2 (pv_a)
            int HBu = upB2;                                   // udh, local left ends: `ml` of H
            if constexpr (DAGP) {                             // the better gap of each pair competes for the cell (:376-378, 399-404)
                const bool f2 = F2 > F, e2 = E2 > E;
                const int Fb = f2 ? F2 : F, FCb = f2 ? FC2 : FC, Eb = e2 ? E2 : E, ECb = e2 ? EC2 : EC;
                if (Fb > H) { H = Fb; HC = FCb; code = 2; }
                if (Eb > H) { H = Eb; HC = ECb; code = 1; }
            } else {
                if (F > H) { H = F; HC = FC; HBu = FB; code = 2; }
                if (E > H) { H = E; HC = EC; HBu = EB; code = 1; }
            }
            int hb_pv = code;
            if (spj) ps &= code;
            if (!local) { if (!(H > XNEV)) H = XNEV; }
            else if (LocalL) { if (0 > H) { H = 0; if constexpr (FORWARD) { code = 1; HC = 0; } } }
            int HB = 0;
            if constexpr (UDH) {
                if (LocalL) {
                    HB = HBu;
                    if (incell && H == 0) { HB = (int) (short) (ml + k + 1); HC = r - 2 * k; }   // left end of a local path
                }
            }
            if constexpr (FORWARD) {
                HB = code == 0;                               // diag: 1, others: 0
                if (HB && !(upB2 & 1) && incell) HC = vadd(ml + k, n - 1 - k, HC);   // a diagonal run starts here
            }
            if (LocalR) {
                int mx = (k < j9) ? H : INT32_MIN;
                int mk = k;                                   // first maximum over the lanes (vmax)
                for (int off = 8; off; off >>= 1) {
                    const int ov = __shfl_xor(mx, off, XN), ok = __shfl_xor(mk, off, XN);
                    if (ov > mx || (ov == mx && ok < mk)) { mx = ov; mk = ok; }
                }
                if (mx > maxh) {
                    maxh = mx;
                    if constexpr (FORWARD) { max_ulk = __shfl(HC, mk, XN); max_mr = ml + mk + 1; max_nr = n - mk; }
                    if constexpr (UDH) { max_ulk = __shfl(HC, mk, XN); max_ml = __shfl(HB, mk, XN); max_mr = ml + mk + 1; max_nr = n - mk; }
                }
            }
            // the exact intron lists: only columns that were queued (pushed at step n_j of THIS stripe, n_j <= b_right)
            const bool queued = spj && k < j9 && nj >= n_first && nj <= b_right;
            const int rj = nj - m;
            if (queued && rj >= lw && rj < up) {
                const unsigned fl = axj & 0xffu;
                // acceptor: Sjsites::get -- unless the best candidate, priced as high as anything can be, cannot beat the lowest of
                // the three states it may raise (every update below is behind `x > state`)
                if ((fl & 2) && ncand >= 0 && c_val[0] + sigJ_cip + s_gain[0] + s_gain[1] + (col.x >> 16) > (DAGP ? min(min(H, min(E, F)), min(E2, F2)) : min(H, min(E, F)))) {
                    const int s3 = col.x >> 16;
                    const int d3 = (axj >> 8) & 15;
                    // udh: maxprd[d], brd -- the best candidate per state and overall: its value and link (and state)
                    bool mx_on[3] = {false, false, false}; int mx_v[3] = {0, 0, 0}, mx_lk[3] = {0, 0, 0};
                    bool br_on = false; int br_v = 0, br_d = 0;
#pragma unroll
                    for (int ci = 0; ci <= NCX; ++ci) {
                        if (ci > ncand) continue;
                        const int d = c_dir[ci], don = c_jnc[ci];
                        if (nj - don < minl) continue;
                        int len = nj - don;
                        const int pen = len < 4096 ? (int) s_ipen[len]
                                      : (A.ipen_runs ? ipen_runs_get(s_runs, len, A.intpen_len) : (int) A.intpen[min(len, A.intpen_len - 1)]);
                        const int x = c_val[ci] + sigJ_cip + pen + s3 + s_t53[16 * c_dn5[ci] + d3];
                        int cur = d == 0 ? H : (d == 1 ? E : (d == 2 || !DAGP ? F : (d == 3 ? E2 : F2)));
                        if (x <= cur) continue;
                        cur = (int) (short) x;
                        if (d == 0) H = cur; else if (d == 1) E = cur; else if (d == 2 || !DAGP) F = cur; else if (d == 3) E2 = cur; else F2 = cur;
                        ps |= x_psp_bit_of(d);
                        if constexpr (FORWARD) {
                            const int inner = vadd(m, don, c_ulk[ci]);
                            const int ptr = vadd(m, nj, inner);
                            const int bml = c_ml[ci];
                            if (d == 0) { HB = bml; HC = ptr; } else if (d == 1) { EB = bml; EC = ptr; } else if (d == 2 || !DAGP) { FB = bml; FC = ptr; }
                            else if (d == 3) { EB2 = bml; EC2 = ptr; } else { FB2 = bml; FC2 = ptr; }
                            if (d && cur > H) { HB = bml; HC = ptr; }
                        }
                        if constexpr (UDH) {
#pragma unroll
                            for (int dd = 0; dd < 3; ++dd)
                                if (dd == d && (!mx_on[dd] || x > mx_v[dd])) {
                                    // NB the reference compares x with the stored candidate's own VALUE (prd->val), as here
                                    mx_on[dd] = true; mx_v[dd] = c_val[ci]; mx_lk[dd] = c_ulk[ci];
                                    if (!br_on || x > br_v) { br_on = true; br_v = c_val[ci]; br_d = d; }
                                }
                            const int lk = c_ulk[ci], bml = c_ml[ci];
                            if (d == 0) { HC = lk; HB = bml; } else if (d == 1) { EC = lk; EB = bml; } else { FC = lk; FB = bml; }
                            if (d && cur > H) { HC = lk; HB = bml; }
                        }
                        if (d && cur > H) H = cur;
                    }
                    if constexpr (UDH) {
                        if (imd_here && br_on) {                  // the acceptor sits on the intermediate row (:91-110)
                            const int maxd = br_d;
                            const int lk = maxd == 0 ? mx_lk[0] : (maxd == 1 ? mx_lk[1] : mx_lk[2]);
                            x_st<PIPE>(&LNK(imd_i, 0, 0, rj), lk); rlst = rj;
                            if (maxd == 0) HC = rj; else if (maxd == 1) EC = rj; else FC = rj;
                            hb_pv = maxd;
                            if (maxd != 0) HC = rj;
                            else {
                                if (mx_on[1] && E > H + gop) EC = rj + width;
                                if (mx_on[2] && F > H + gop) { x_st<PIPE>(&LNK(imd_i, 0, 1, rj), lk); FC = rj + width; }
                            }
                        }
                    }
                }
                if (fl & 1) {                                 // donor: Sjsites::put
                    const int sigJ = (int) (short) (col.x & 0xffff) - ipen;
                    for (int kk = hb_pv ? 1 : 0; kk < NOD; ++kk) {
                        if (ps & x_psp_bit_of(kk)) continue;
                        const int from = kk == 0 ? H : (kk == 1 ? E : (kk == 2 ? F : (kk == 3 ? E2 : F2)));
                        if (kk && from <= H + (kk <= 2 ? gop : lgop)) continue;      // GOP[(k + 1) / 2]
                        const int x = from + sigJ;
                        if (x <= XNEV) continue;
                        if (ncand >= NCX - 1 && c_val[NCX - 1] > x) { ncand = NCX - 1; continue; }      // a full list whose last kept entry beats x: it only gets shorter
                        // the free slot starts below the list and moves up past every entry x ties or beats
                        int pos = ncand < NCX ? ncand + 1 : NCX;
                        if (ncand < NCX) ++ncand;
#pragma unroll
                        for (int l = NCX; l >= 1; --l)
                            if (pos == l && x >= c_val[l - 1]) {
                                c_val[l] = c_val[l - 1]; c_jnc[l] = c_jnc[l - 1]; c_dir[l] = c_dir[l - 1]; c_dn5[l] = c_dn5[l - 1];
                                c_ml[l] = c_ml[l - 1]; c_ulk[l] = c_ulk[l - 1];
                                pos = l - 1;
                            }
                        if (pos < NCX) {
                            int n_ml = 0, n_ulk = 0;
                            if constexpr (FORWARD) {
                                n_ml = kk == 0 ? HB : (kk == 1 ? EB : (kk == 2 ? FB : (kk == 3 ? EB2 : FB2)));
                                n_ulk = kk == 0 ? HC : (kk == 1 ? EC : (kk == 2 ? FC : (kk == 3 ? EC2 : FC2)));
                            }
                            if constexpr (UDH) {
                                if (imd_here) { if (kk & 1) x_st<PIPE>(&LNK(imd_i, 0, 0, rj), rlst); n_ulk = rj; }
                                else n_ulk = kk == 0 ? HC : (kk == 1 ? EC : FC);
                                n_ml = kk == 0 ? HB : (kk == 1 ? EB : FB);
                            }
#pragma unroll
                            for (int l = 0; l < NCX; ++l)
                                if (l == pos) {
                                    c_val[l] = (int) (short) x; c_jnc[l] = nj; c_dir[l] = kk; c_dn5[l] = (int) (axj >> 12) & 15;
                                    c_ml[l] = n_ml; c_ulk[l] = n_ulk;
                                }
                        }
                        else --ncand;
                    }
                }
            }
            if constexpr (UDH) {                              // intermediate row: lane k8 of its stripe (:1043-1052)
                const int rq = r - 2 * k8;
                if (is_imd_ && k == k8 && rq >= lw && rq <= up) {
                    if (hb_pv == 0) rlst = rq;
                    if (hb_pv == 1) x_st<PIPE>(&LNK(imd_i, 0, 0, rq), rlst);
                    x_st<PIPE>(&LNK(imd_i, 1, 0, rq), HC); HC = rq;
                    x_st<PIPE>(&LNK(imd_i, 1, 1, rq), FC); FC = rq + width;
                }
            }
            // bottom row of the stripe -> boundary arrays
            if (k == j8 && j9 == ke && lw <= r0 && r0 <= up) {
                x_st<PIPE>(&hv[r0], H); x_st<PIPE>(&fv[r0], F);
                if constexpr (DAGP) { x_st<PIPE>(&fv2[r0], F2); if constexpr (PTR) x_st<PIPE>(&fc2[r0], FC2); }
                if constexpr (FORWARD) x_st<PIPE>(&hb[r0], HB);
                if constexpr (PTR) { x_st<PIPE>(&hc[r0], HC); x_st<PIPE>(&fc[r0], FC); }
                if constexpr (UDH) { if (LocalL) { x_st<PIPE>(&hb[r0], HB); x_st<PIPE>(&fb[r0], FB); } }
            }
            H2 = H1; H1 = H; F1 = F;
            if constexpr (DAGP) { F21 = F2; FC21 = FC2; }
            if constexpr (FORWARD || UDH) { B2 = B1; B1 = HB; }
            if constexpr (PTR) { C2 = C1; C1 = HC; FC1 = FC; }
            if (PIPE && n == n9 - 1) finish_stripe();
        }
        if constexpr (UDH) {
            if (is_imd_) {
                rlst = __shfl(rlst, k8, XN);                              // hs1.rlst is one variable for all lanes
                if (PIPE && k == 0) x_st<true>(rlf + imd_i, rlst);
                ++imd_i;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (PIPE) {
        // finished = every entry holds what the stripes up to this one leave: the stripe above must be finished too
        const int st = my_stripe;
        if (ml_first >= ml_end) {                                         // (a problem without rows: no stripe loop ran)
            if (st > 0) wait_for(st - 1, INT32_MAX);
            publish(st, INT32_MAX);
        }
        if (st != n_stripes - 1) return;
        if (LocalR) {                                                     // stripes in order: the first maximum wins
            maxh = XNEV; max_ulk = 0; max_mr = a_right; max_nr = b_right; max_ml = a_left;
            for (int t = 0; t < n_stripes; ++t) {
                const int* b = tbest + 6 * t;
                const int v = x_ld<true>(b);
                if (v > maxh) { maxh = v; max_ulk = x_ld<true>(b + 1); max_mr = x_ld<true>(b + 2); max_nr = x_ld<true>(b + 3); max_ml = x_ld<true>(b + 4); }
            }
        }
    }
    // ---- fhlastS1 unless a local right end was tracked; forward: the end record, Vmf::traceback, fix-up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (k == 0) {
        DevResult R;
        R.score = XNEV; R.mr = a_right; R.nr = b_right; R.ml = a_left; R.ulk = 0; R.maxr = 0; R.pad[0] = R.pad[1] = 0;
        int end_ulk = max_ulk;
        if (LocalR) { R.score = maxh; R.mr = max_mr; R.nr = max_nr; if constexpr (UDH) R.ml = max_ml; }
        else {
            const int rr = b_right - a_right;
            int maxr = rr;
            if (a_exgr) {
                const int r1 = max(lw, b_left - a_right);
                int best = r1;
                for (int i = r1 + 1; i < rr; ++i) if (x_ld<PIPE>(&hv[i]) > x_ld<PIPE>(&hv[best])) best = i;
                maxr = best;
            }
            if (b_exgr) {
                const int r2 = min(up - 1, b_right - a_left);
                int best = rr;
                for (int i = rr + 1; i < r2; ++i) if (x_ld<PIPE>(&hv[i]) > x_ld<PIPE>(&hv[best])) best = i;
                if (x_ld<PIPE>(&hv[best]) > x_ld<PIPE>(&hv[maxr])) maxr = best;
            }
            R.score = x_ld<PIPE>(&hv[maxr]);
            R.maxr = maxr;
            if (maxr > rr) R.mr = b_right - maxr; else R.nr = a_right + maxr;
            if constexpr (PTR) end_ulk = x_ld<PIPE>(&hc[maxr]);
            if constexpr (UDH) { if (LocalL) R.ml = x_ld<PIPE>(&hb[maxr]); }
        }
        if constexpr (FORWARD) {
            const int ptr = vadd(R.mr, R.nr, end_ulk);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int used = x_ld<PIPE>(vcount);
            int2* out = A.skl + (int64_t) pi * A.skl_cap;
            int cnt = 0, status = used > vcap ? -3 : 0;
            if (!status) {
                const int* vr = reinterpret_cast<const int*>(vrec);
                int cur = ptr, lm = 0, ln = 0;
                for (;;) {
                    const int sm = x_ld<PIPE>(vr + 3 * (int64_t) cur);
                    const int sn = x_ld<PIPE>(vr + 3 * (int64_t) cur + 1);
                    const int sp = x_ld<PIPE>(vr + 3 * (int64_t) cur + 2);
                    if (cnt < A.skl_cap) out[cnt] = make_int2(sm, sn); else status = -1;
                    lm = sm; ln = sn; ++cnt;
                    if (!sp) break;
                    cur = sp;
                }
                const int rd = local ? 0 : ((ln - lm) - b_left + a_left);
                if (rd) {
                    const int2 rec = rd > 0 ? make_int2(a_left, b_left + rd) : make_int2(a_left - rd, b_left);
                    if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
                    ++cnt;
                }
            }
            A.n_skl[pi] = status ? status : cnt;
        }
        if constexpr (UDH) R.ulk = end_ulk;
        A.res[pi] = R;
    }
}

extern "C" hipError_t spdp_launch_exact(int mode, const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    const dim3 blk(64 * XWPB);
    const bool dagp = A.noll == 3;               // double affine gaps: score-only and forward engines (DevRun::build refuses mode 2)
    if (dagp && mode == 2) return hipErrorNotSupported;
    if (A.pipe) {                                // one wave per (four problems, stripe)
        const dim3 grd((A.n_items + XWPB - 1) / XWPB);
        if (mode == 2) hipLaunchKernelGGL((spdp_exact<2, true>), grd, blk, 0, stream, A);
        else if (mode == 1) { if (dagp) hipLaunchKernelGGL((spdp_exact<1, true, true>), grd, blk, 0, stream, A); else hipLaunchKernelGGL((spdp_exact<1, true>), grd, blk, 0, stream, A); }
        else { if (dagp) hipLaunchKernelGGL((spdp_exact<0, true, true>), grd, blk, 0, stream, A); else hipLaunchKernelGGL((spdp_exact<0, true>), grd, blk, 0, stream, A); }
        return hipGetLastError();
    }
    const dim3 grd((A.n_probs + 4 * XWPB - 1) / (4 * XWPB));
    if (mode == 2) hipLaunchKernelGGL((spdp_exact<2, false>), grd, blk, 0, stream, A);
    else if (mode == 1) { if (dagp) hipLaunchKernelGGL((spdp_exact<1, false, true>), grd, blk, 0, stream, A); else hipLaunchKernelGGL((spdp_exact<1, false>), grd, blk, 0, stream, A); }
    else { if (dagp) hipLaunchKernelGGL((spdp_exact<0, false, true>), grd, blk, 0, stream, A); else hipLaunchKernelGGL((spdp_exact<0, false>), grd, blk, 0, stream, A); }
    return hipGetLastError();
}
