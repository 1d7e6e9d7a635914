// spdp_h_exact.hip -- the reference's -A1 ("full-precision intron-length distribution") protein engines.
//
//   spdh_exact<false, .>   SimdAln2h1::forwardH1 (modes 3 / 5)          src/fwd2h1_simd.h:820-1096
//   spdh_exact<true, .>    SimdAln2h1::hirschbergH1 (modes 2 / 4)       src/fwd2h1_simd.h:1100-1470
//                          fhinitH1 / fhlastH1                          src/fwd2h1_simd.h:546-791
//                          Sjsites::get / put, from_spj / to_spj        src/fwd2h1_simd.h:388-543, 793-815
//                          Vmf::traceback + the fix-up of trcbkalignH_ng  src/vmf.cc:125, src/fwd2h1.cc:2019-2036
//
// 16 int16 lanes per stripe chained through per-diagonal boundary rows as in the `_wip` engines (the results
// depend on that geometry); the intron model is the scalar engines': every lane keeps the top-4 donor
// candidates of its row (value, junction, state, phase), an acceptor column re-scores them with the exact
// IntPen(len), the pair signal and the codon the junction spells, and raises H / E / F of the cells of the
// current and the two previous steps (one per codon phase).  The reference hangs the lists off the vector loop
// as scalar calls per queued column (donor_q / accep_q, one queue per frame); a queued column is met by lane j
// exactly once, at step n_j + 3 j, so here every lane looks at its own column.  A Vmf pointer (forward) or
// the link to the previous intermediate row (linear space) rides on H / E / F.
//
// Mapping (round 4; the first form kept 99 plane words per lane in LDS, indexed by the step's phase, read its
// columns and tables from memory inside the candidate loops and ran the stripes of a problem one after the
// other in one 16-lane group -- 1 GCUPS):
//  * 16 lanes = one stripe of one problem, four problems per wave, four waves per block sharing the tables;
//  * the reference's six codon-phase planes of H / F and three of E are "the cell of this lane 1, 2, 3 steps
//    ago": registers that rotate by one per step, so every access is to a named register;
//  * the lane above hands down ONE row per step (its cell of three steps ago, a row_shr:1 DPP move; lane 0
//    takes the previous stripe's bottom row from an LDS feed instead) and the lane keeps the last three it
//    received -- the reference reads the same cells from the planes of steps n - 3 .. n - 6;
//  * column records (residue, coding potential, site flags, signals, junction class) come through a 128-slot
//    LDS ring per group, a block of 16 steps ahead; boundary rows through a 16-entry feed per row, loaded a
//    block ahead; IntPen (4096 lengths + the run table of spdp_ipen_runs.h), the junction-pair table and the
//    genetic-code tables in LDS; what an acceptor needs of a donor's column (junction class, the two bases
//    before it) is packed into the candidate when it is made;
//  * the candidate list is five register slots kept in order by moving entries (no index indirection);
//  * PIPE: the stripes of a problem run as a pipeline of waves.  A work item is (four problems, stripe), drawn
//    from a ticket counter in dispatch order, so a stripe's predecessor is always resident or done.  The boundary
//    rows cross CUs (agent-scope accesses); a stripe publishes, once its stores have drained, the diagonal up to
//    which its bottom row is final (an acceptor may still raise the entries of the last two steps) and reads its
//    predecessor's word before it loads a block of feed entries.  hb1.rlst is only ever stored (into the links
//    of an intermediate row): a stripe starts from a marker per frame and the link walk replaces it by what the
//    intermediate rows above left (rlf[]); the local maximum is reduced per stripe and combined in stripe order;
//    Vmf record numbers are reserved eight at a time per lane.  The reference's link planes are not
//    re-initialised from stripe to stripe; only cells at nevsel ever read the stale words, and no result
//    (score, records, cpos) can see them, so a stripe starts them at zero.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

#define XN 16
#define XNEV (-32768 + 1024)
#define X_EOU (0x7fffffff - 2)                   // end_of_ulk
#define HXWPB 4                                  // waves per block
#define HXRING 128                               // column records per group in LDS
#define HXVCH 16                                 // Vmf record numbers a lane reserves at a time (the next chunk is asked for at half)
#define HXINH 0x7ffffff0                         // SPDP_RLST_INHERITED (spdp_internal.h) + frame: "rlst as the rows above left it"
#define HXPROG0 (1 << 28)
#define HXPOISON 0x7fffff00                      // pipelined form: a link word the reference would have inherited from the previous stripe

__device__ __forceinline__ int xh_add(int a, int b) { return min(max(a + b, -32768), 32767); }
__device__ __forceinline__ int xh_w16(int x) { return (int) (short) x; }
__device__ __forceinline__ int xh_up(int v) { return __shfl_up(v, 1, XN); }
__device__ __forceinline__ int xh_mod6(int x) { x %= 6; return x < 0 ? x + 6 : x; }
// lane i of every 16-lane row <- lane i - 1; lane 0 of the row keeps `old`
__device__ __forceinline__ int xh_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false); }
__device__ __forceinline__ int xh_psp_bit(int d) { return d == 0 ? 4 : (d == 1 ? 1 : 8); }
template <bool X> __device__ __forceinline__ int xh_ld(const int* p)
{
    if constexpr (X) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool X> __device__ __forceinline__ void xh_st(int* p, int v)
{
    if constexpr (X) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
#define SEL3(a, x0, x1, x2) ((a) == 0 ? (x0) : ((a) == 1 ? (x1) : (x2)))

enum { FD_HV, FD_FV, FD_HC, FD_FC, FD_HB, FD_FB, FD_N };

template <bool UDH, bool PIPE>
__global__ void __launch_bounds__(64 * HXWPB) __attribute__((amdgpu_waves_per_eu(2, 2))) spdh_exact(HScalarArgs A)
{
    __shared__ int s_mtx[32 * 32];               // the substitution matrix (aa x tron, row stride 32)
    __shared__ int4 s_col[4 * HXWPB][HXRING];    // {cp | tron << 16 | flags << 24, sig3 candidates, dinc5 << 4 | dinc3, sigE | sig5 << 16}
    __shared__ int s_fd[4 * HXWPB][FD_N][16];    // the previous stripe's bottom row, entries rb + 3 .. rb + 18 of the running block
    __shared__ short s_ipen[4096];
    __shared__ IpenRuns s_runs;                  // IntPen beyond s_ipen
    __shared__ short s_t53[256];
    __shared__ unsigned char s_mid[32], s_tron[64];
    __shared__ int s_bnd[3];                     // max IntPen, max junction-pair score, max |mtx|: what a candidate can gain at most
    const int tid = threadIdx.x;
    if (tid < 3) s_bnd[tid] = tid == 2 ? 0 : -0x7fffffff;
    __syncthreads();
    {
        int pm = -0x7fffffff, tm = -0x7fffffff, mm = 0;
        for (int i = tid; i < A.intpen_len; i += 64 * HXWPB) pm = max(pm, (int) A.intpen[i]);
        for (int i = tid; i < 256; i += 64 * HXWPB) tm = max(tm, (int) A.t53[i]);
        for (int i = tid; i < 32 * 32; i += 64 * HXWPB) mm = max(mm, abs(A.sc->mtx[i]));
        atomicMax(&s_bnd[0], pm); atomicMax(&s_bnd[1], tm); atomicMax(&s_bnd[2], mm);
    }
    for (int i = tid; i < 32 * 32; i += 64 * HXWPB) s_mtx[i] = A.sc->mtx[i];
    for (int i = tid; i < 4096; i += 64 * HXWPB) s_ipen[i] = A.intpen[min(i, A.intpen_len - 1)];
    ipen_runs_load(s_runs, A.ipen_runs);
    for (int i = tid; i < 256; i += 64 * HXWPB) s_t53[i] = A.t53[i];
    if (tid < 32) s_mid[tid] = A.mid[tid];
    if (tid < 64) s_tron[tid] = A.tron_of[tid];
    __syncthreads();                             // (before any group leaves)
    const int k = tid & 15;
    const int grp = (tid & 63) >> 4;             // my 16-lane group in the wave
    const int g16 = tid >> 4;                    // ... in the block (its ring and feed)
    const int G = A.item_probs;                  // problems per wave: four, or fewer while the launch is small (more waves in flight)
    if (grp >= G) return;
    int pi = (blockIdx.x * HXWPB + (tid >> 6)) * G + grp;
    int my_stripe = -1;                          // PIPE: the one stripe this group sweeps
    if (PIPE) {
        int tk = 0;
        if ((tid & 63) == 0) tk = __hip_atomic_fetch_add(A.pipe + A.pipe_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= A.n_items) return;
        const int2 it = A.items[tk];
        pi = it.x * G + grp; my_stripe = it.y;
    }
    if (pi >= A.n_probs) return;                 // a whole 16-lane group leaves together
    const DevProblemH P = A.probs[pi];
    const DevScoringH* sc = A.sc;
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, B = P.buf_size;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const bool local = sc->local;
    const bool LocalL = local && a_exgl && b_exgl, LocalR = local && a_exgr && b_exgr;
    const bool spj = sc->spj;
    const bool useB = !UDH || LocalL;            // the `ml` / diagonal-flag planes are read at all
    const bool UL = UDH && LocalL;
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
    int* vcount = A.work + P.bnd_off + 6 * (int64_t) B;      // Vmf records appended (reserved) so far
    int3* vrec = A.vmf + (UDH ? 0 : P.tb_off);
    int* vraw = reinterpret_cast<int*>(vrec);
    const int vcap = UDH ? 0 : (int) P.imd_off;
    // a lane's next chunk of numbers is asked for while half of the current one is still there: the counter's round trip
    // (memory side, a microsecond) runs under the steps in between instead of stopping the wave at every eighth record of
    // every lane
    int v_next = 0, v_left = 0, v_pend = 0;
    bool v_asked = false;
    auto vadd = [&](int mm, int nn, int pp) -> int {
        if (v_left == 0) {
            if (!v_asked) v_pend = __hip_atomic_fetch_add(vcount, HXVCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v_next = v_pend; v_left = HXVCH; v_asked = false;
        }
        const int i = v_next++;
        --v_left;
        if (v_left == HXVCH / 2 && !v_asked) { v_pend = __hip_atomic_fetch_add(vcount, HXVCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v_asked = true; }
        if (i < vcap) { xh_st<PIPE>(vraw + 3 * i, mm); xh_st<PIPE>(vraw + 3 * i + 1, nn); xh_st<PIPE>(vraw + 3 * i + 2, pp); }
        return i;
    };
    int* imd0 = A.imd + (UDH ? P.imd_off : 0);   // hlnk[2], vlnk[2] per intermediate row, `width` ints each
    auto LNK = [&](int i, int which, int d, int r) -> int* { return imd0 + (((int64_t) i * 4 + which * 2 + d) * width + (r - lw + 1)); };
    const int n_im = UDH ? P.n_im : 0;
    const int imd_step = UDH ? (a_right - a_left + n_im) / (n_im + 1) : 0;
    auto gext3 = [&](int i) { return i > codonk1 ? lgep : gep; };
    auto acode = [&](int i) -> int { return i < 0 ? 2 : (i >= P.a_len ? P.a_pad : acod[i]); };      // a_pad: SpdpProblemH
    const int c_hi = P.b_len + 2;
    // no candidate of value v can reach more than v + sig3 + gain_max (+ |sigE| of the acceptor for a split codon)
    const int gain_max = s_bnd[0] + s_bnd[1] + 2 * s_bnd[2];
    const bool has_cip = A.cip && P.cip_off >= 0;
    const int n_stripes = max(1, (a_right - a_left + XN - 1) / XN);
    if (PIPE && my_stripe >= n_stripes) return;  // (a shorter problem of the four)
    // PIPE: what the stripes of the problem share: prog[max_tiles], best[max_tiles][6], rlf[n_im][3]
    int* sy = PIPE ? A.pipe + (size_t) pi * A.pipe_stride : nullptr;
    int* prog = PIPE ? sy + 2 : nullptr;
    int* tbest = PIPE ? sy + 2 + A.max_tiles : nullptr;
    int* rlf = PIPE ? sy + 2 + 7 * A.max_tiles : nullptr;
    bool stalled = false;
    auto wait_for = [&](int t, int req) {        // until stripe t has published at least `req` (per lane: a group waits for its own problem)
        long spins = 0;
        while (!stalled && __hip_atomic_load(prog + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < req) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1l << 22)) {          // (cannot happen with the ticket order; bounds every spin)
                __hip_atomic_store(A.pipe + A.pipe_ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stalled = true;
            }
        }
    };
    auto publish = [&](int t, int v) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (k == 0) __hip_atomic_store(prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- fhinitH1 (:546-689): bulk fills by all lanes, the sequential parts by lane 0
    const int rl = b_left - 3 * a_left;
    if (!PIPE || my_stripe == 0) {
        for (int e = k; e < 2 * B; e += XN) {
            xh_st<PIPE>(hv + lw - 3 + e, XNEV);
            xh_st<PIPE>(hb + lw - 3 + e, UDH ? a_left : 0);
            xh_st<PIPE>(hc + lw - 3 + e, 0);
        }
        if constexpr (UDH && !PIPE)
            for (int e = k; e < n_im * 4 * width; e += XN) imd0[e] = X_EOU;
        if (k == 0) xh_st<PIPE>(vcount, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int sc_on = 0, sc_r = 0, sc_n = 0, sc_bb = 0, sc_rr = 0;        // the first-row scan, where it stands (lane 0 -> group)
        int sh1 = 0, sh2 = 0, sh3 = 0, sc1 = 0, sc2 = 0, sc3 = 0, sl0 = 0, sl1 = 0, sl2 = 0;
        // the link / `ml` rows as fhinitH1 leaves them: by all lanes of the group
        int ptr0 = 0;
        if constexpr (!UDH) {
            if (k == 0) {
                ptr0 = vadd(0, 0, 0);
                if (!(a_exgl && b_exgl)) ptr0 = vadd(a_left, b_left, ptr0);
            }
            ptr0 = __shfl(ptr0, 0, XN);
            for (int r = rl + k; r < up; r += XN) xh_st<PIPE>(hc + r, a_exgl ? 0 : ptr0);
            for (int r = lw + k; r < rl; r += XN) xh_st<PIPE>(hc + r, b_exgl ? 0 : ptr0);
        } else {
            const int re = a_exgl ? rl : up;
            for (int r = lw + k; r < re; r += XN) xh_st<PIPE>(hc + r, r);
            for (int i = k, r = rl - k; r >= lw; r -= XN, i += XN) xh_st<PIPE>(hb + r, a_left + i / 3);
        }
        if (b_exgl == 1) { for (int r = lw + k; r < rl; r += XN) xh_st<PIPE>(hv + r, 0); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (k == 0) {
            if constexpr (!UDH) { if (b_exgl == 2) xh_st<PIPE>(fc + rl, ptr0); }
            if (b_exgl == 2) { xh_st<PIPE>(fv + rl, 0); xh_st<PIPE>(fc + rl, rl); }
            int rr = b_right - 3 * a_left;
            if (up < rr) rr = up;
            int r = rl;
            if (!a_exgl) {
                if (b_exgl) { xh_st<PIPE>(fv + r, 0); xh_st<PIPE>(fc + r, UDH ? rl : ptr0); }       // (= hc[rl] as set above)
                xh_st<PIPE>(hv + r++, 0);
                xh_st<PIPE>(hv + r++, xh_w16(g1));
                xh_st<PIPE>(hv + r++, xh_w16(g2));
                xh_st<PIPE>(hv + r++, xh_w16(g3));
                int h1 = xh_w16(g3), h2 = xh_w16(g2), h3 = xh_w16(g1);      // hv[r - 1], [r - 2], [r - 3]
                if (gep) {
                    const int x = (XNEV - g3) / gep + r;
                    if (x < rr) rr = x;
                    for ( ; r < rr; ++r) { const int h = xh_w16(h3 + gep); xh_st<PIPE>(hv + r, h); h3 = h2; h2 = h1; h1 = h; }
                } else if (rr > r)
                    for (const int v = h1; r < rr; ++r) xh_st<PIPE>(hv + r, v);
            } else {
                int n = b_left;
                int le0 = r, le1 = r + 1, le2 = r + 2;            // lend[f], rotated with f
                int bb = n + 1;
                int h1 = 0, h2 = 0, h3 = 0, c1 = 0, c2 = 0, c3 = 0;          // hv / hc [r - 1], [r - 2], [r - 3]
                for (int f = 0; f < 3; ++f, ++r, ++n, ++bb) {
                    const int s = aux[bb].x;
                    const int h = s > 0 ? s : 0;
                    int c;
                    if constexpr (!UDH) { c = vadd(a_left, n, 0); xh_st<PIPE>(hb + r, 1); }
                    else c = r;
                    xh_st<PIPE>(hv + r, h); xh_st<PIPE>(hc + r, c);
                    h3 = h2; h2 = h1; h1 = h; c3 = c2; c2 = c1; c1 = c;
                }
                sc_on = 1; sc_r = r; sc_n = n; sc_bb = bb; sc_rr = rr;
                sh1 = h1; sh2 = h2; sh3 = h3; sc1 = c1; sc2 = c2; sc3 = c3; sl0 = le0; sl1 = le1; sl2 = le2;
            }
        }
        // the max-plus scan over the free first row (:640-689): one entry after the other on lane 0, its signals staged
        // through LDS by the whole group, 128 entries at a time (a load per entry from memory made this scan -- as long
        // as the window -- the longest phase of the stripe that owns it)
        sc_on = __shfl(sc_on, 0, XN); sc_r = __shfl(sc_r, 0, XN); sc_n = __shfl(sc_n, 0, XN); sc_bb = __shfl(sc_bb, 0, XN); sc_rr = __shfl(sc_rr, 0, XN);
        {
            int* const sI = reinterpret_cast<int*>(s_col[g16]);
            while (sc_on && sc_r < sc_rr) {
                const int cnt = min(128, sc_rr - sc_r);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int i = k; i < cnt + 3; i += XN) {
                    const short4 a = aux[max(sc_bb - 3 + i, 0)];
                    sI[i] = (int) ((unsigned) (unsigned short) a.x | ((unsigned) (unsigned short) a.z << 16));      // sigS | sigE << 16
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                int stop = 0;
                if (k == 0) {
                    int r = sc_r, n = sc_n;
                    for (int i = 0; i < cnt; ++i, ++r, ++n) {
                        int h = sh3, c = sc3;
                        const int gl = r - sl0;
                        if (!(a_exgl & 1) && gl == 3) h = xh_w16(h + gop);
                        if (!(a_exgl & 2)) h = xh_w16(h + gext3(gl));
                        h = xh_w16(h + (sI[i] >> 16));                        // aux[bb - 3].z
                        if (h < XNEV) { xh_st<PIPE>(hv + r, h); xh_st<PIPE>(hc + r, c); stop = 1; break; }
                        int x = xh_w16(sh1 + g1);
                        if (x > h) { h = x; c = sc1; }
                        x = xh_w16(sh2 + g2);
                        if (x > h) { h = x; c = sc2; }
                        const int sS = (int) (short) (sI[i + 3] & 0xffff);    // aux[bb].x
                        x = sS > 0 ? sS : 0;
                        if (x > h) {
                            h = x; sl0 = r;
                            if constexpr (!UDH) { c = vadd(a_left, n, 0); xh_st<PIPE>(hb + r, 1); }
                            else c = r;
                        }
                        xh_st<PIPE>(hv + r, h); xh_st<PIPE>(hc + r, c);
                        sh3 = sh2; sh2 = sh1; sh1 = h; sc3 = sc2; sc2 = sc1; sc1 = c;
                        const int t_ = sl0; sl0 = sl1; sl1 = sl2; sl2 = t_;
                    }
                }
                if (__shfl(stop, 0, XN)) break;
                sc_r += cnt; sc_n += cnt; sc_bb += cnt;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    int max_val = XNEV, max_ulk = X_EOU, max_ml = a_left, max_mr = a_right, max_nr = b_right;
    int rl0 = 0x7fffffff, rl1 = 0x7fffffff, rl2 = 0x7fffffff;       // hb1.rlst by frame; only the lane of the intermediate row uses it
    int imd_i = 0;
    const int ml_first = PIPE ? a_left + XN * my_stripe : a_left;
    const int ml_end = PIPE ? min(a_right, ml_first + XN) : a_right;
    if (PIPE && UDH) {
        // the counter as the stripes above would have left it: one step per stripe that held the then current row
        for (int mq = a_left; mq < ml_first && imd_i < n_im; mq += XN) {
            const int mi = a_left + (imd_i + 1) * imd_step;
            if (mq == a_left + (mi - a_left - 1) / XN * XN) ++imd_i;
        }
    }
    int4* const ring = s_col[g16];
    int (*const fd)[16] = s_fd[g16];
    // the link planes of H / F (six) and E (three) by age.  The reference does not re-initialise them from stripe to
    // stripe: a stripe's first steps read what the previous stripe left in the plane of that phase.  Only cells at nevsel
    // start out with those words, so they show only where the final walk leaves the alignment proper (rows above a
    // free left end, a problem without a path) -- the one-group form reproduces them (the planes re-indexed by the new
    // stripe's phase); the pipelined form cannot: it starts them at HXPOISON.  Links are only ever copied, never
    // computed on, so every other link word is the reference's, and a result whose walk meets the poison is marked
    // for a second run in the one-group form (DevResultH::pad[1]).
    int HC[7] = {0, 0, 0, 0, 0, 0, 0}, FC[7] = {0, 0, 0, 0, 0, 0, 0}, EC[4] = {0, 0, 0, 0};
    int q_end = 0;                                                 // the phase after the last step of the previous stripe
    const int e_lo = lw - 3, e_hi = lw - 3 + 2 * B - 1;            // entries of a row pair (hv | fv, ...) that exist
    for (int ml = ml_first; ml < ml_end; ml += XN) {
        const int j9 = min(XN, a_right - ml);
        const int j8 = j9 - 1;
        int n = max(b_left, lw + 3 * ml);
        const int n_first = n;
        const int n9 = min(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;
        int q = xh_mod6(n + 3 * (ml + 1));
        int r = n - 3 * (ml + 1);
        const int st = PIPE ? my_stripe : 0;
        auto ready = [&](int rq) { if (PIPE && st > 0) wait_for(st - 1, rq + HXPROG0); };
        // the cell of this lane by age: [0] this step, [1] .. [3] one to three steps ago (the reference's planes
        // q, q - 1, .. mod 6 / mod 3).  Stripe reset (:849-860): scores to nevsel, flags / side lanes / lists cleared.
        int HV[4] = {XNEV, XNEV, XNEV, XNEV}, FV[4] = {XNEV, XNEV, XNEV, XNEV}, EV[4] = {XNEV, XNEV, XNEV, XNEV};
        int HB[4] = {0, 0, 0, 0}, EB[4] = {0, 0, 0, 0}, FB[7] = {0, 0, 0, 0, 0, 0, 0};
        if (PIPE || ml == a_left) {
            const int w0 = ml == a_left ? 0 : HXPOISON;
#pragma unroll
            for (int i = 0; i < 7; ++i) { HC[i] = w0; FC[i] = w0; }
            EC[0] = EC[1] = EC[2] = EC[3] = w0;
        } else {
            // age i of this stripe is the plane of phase q - i: age ((q_end - q + i - 1) mod 6) + 1 of the last stripe
            const int s6 = xh_mod6(q_end - q), s3 = s6 % 3;
            int oh[7], of[7], oe[4];
#pragma unroll
            for (int i = 0; i < 7; ++i) { oh[i] = HC[i]; of[i] = FC[i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) oe[i] = EC[i];
#pragma unroll
            for (int i = 1; i <= 6; ++i) {
                int a = s6 + i - 1; if (a >= 6) a -= 6;
#pragma unroll
                for (int o = 0; o < 6; ++o) if (a == o) { HC[i] = oh[o + 1]; FC[i] = of[o + 1]; }
            }
#pragma unroll
            for (int i = 1; i <= 3; ++i) {
                int a = s3 + i - 1; if (a >= 3) a -= 3;
#pragma unroll
                for (int o = 0; o < 3; ++o) if (a == o) EC[i] = oe[o + 1];
            }
        }
        int PS[4] = {0, 0, 0, 0}, PV[3] = {0, 0, 0}, CP[4] = {0, 0, 0, 0};
        // what came down from the lane above 1, 2, 3 steps ago (its cells of 4, 5, 6 steps ago)
        int uH[4] = {XNEV, XNEV, XNEV, XNEV}, uC[4] = {0, 0, 0, 0}, uB[4] = {0, 0, 0, 0};
        if (!PIPE) { uC[1] = xh_shr1(0, HC[4]); uC[2] = xh_shr1(0, HC[5]); uC[3] = xh_shr1(0, HC[6]); }
        else if (ml != a_left) uC[1] = uC[2] = uC[3] = HXPOISON;
        // the donor candidates of my row, best first: value, donor position, `ml` / link of the state it left, and what an
        // acceptor needs of the donor's column (c_dk)
        int c_val[5], c_jnc[5], c_ml[5], c_ulk[5], c_dk[5], ncand = -1;      // c_dk: bits 0-11 donor column pack, 12-13 state, 14-15 phase + 1
#pragma unroll
        for (int i = 0; i < 5; ++i) { c_val[i] = XNEV; c_jnc[i] = 0; c_ml[i] = 0; c_ulk[i] = 0; c_dk[i] = 0; }
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
        (void) k9;
        const int mm3 = 3 * mi;
        const bool imd_lane = UDH && imd_i < n_im && (m + 1) == mi;      // Sjsites' is_imd for my row
        const bool site_lane = spj && k <= j8 && (m + 1) < a_right;
        const int* mrow = s_mtx + ((k < j9) ? acod[ml + k] : 0) * 32;
        const int* mrow_m = s_mtx + acode(m) * 32;            // a codon split by an intron is scored against these rows
        const int* mrow_m1 = s_mtx + acode(m + 1) * 32;
        if (PIPE && UDH && is_imd_) {
            for (int e = k; e < 4 * width; e += XN) xh_st<true>(imd0 + (int64_t) imd_i * 4 * width + e, X_EOU);
            if (imd_i) { rl0 = HXINH; rl1 = HXINH + 1; rl2 = HXINH + 2; }
        }
        // ---- staging: the ring holds columns [nb - 48, nb + 32) while the block that starts at step nb runs; the feed
        // holds entries rb + 3 .. rb + 18 of the six boundary rows; registers hold the NEXT block's loads
        auto ld_col = [&](int c) -> int4 {
            const int cc = min(max(c, 0), c_hi);
            const int4 v = cols[cc];
            const short4 a = aux[cc];
            return make_int4(v.x, v.y, v.w & 0xff, (int) ((unsigned) (unsigned short) a.z | ((unsigned) (unsigned short) a.w << 16)));
        };
        int4 pc = make_int4(0, 0, 0, 0);
        int pfd[FD_N] = {0, 0, 0, 0, 0, 0};
        auto ld_feed = [&](int e, int* o) {                    // entry e of each row (clamped to what exists; beyond: never used)
            const int ee = min(max(e, e_lo), e_hi - B);
            o[FD_HV] = xh_ld<PIPE>(hv + ee); o[FD_FV] = xh_ld<PIPE>(fv + ee);
            o[FD_HC] = xh_ld<PIPE>(hc + ee); o[FD_FC] = xh_ld<PIPE>(fc + ee);
            o[FD_HB] = useB ? xh_ld<PIPE>(hb + ee) : 0;
            o[FD_FB] = UL ? xh_ld<PIPE>(fb + ee) : 0;
        };
        {
            ready(r + 3 + 15 + 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 5; ++t) ring[(n_first - 48 + 16 * t + k) & (HXRING - 1)] = ld_col(n_first - 48 + 16 * t + k);
            int f0[FD_N];
            ld_feed(r + 3 + k, f0);
#pragma unroll
            for (int a = 0; a < FD_N; ++a) fd[a][k] = f0[a];
            // lane 0's history of the first step: entries r, r + 1, r + 2 (what it would have received 3, 2, 1 steps ago)
            if (k == 0) {
                uH[3] = xh_ld<PIPE>(hv + r); uH[2] = xh_ld<PIPE>(hv + r + 1); uH[1] = xh_ld<PIPE>(hv + r + 2);
                uC[3] = xh_ld<PIPE>(hc + r); uC[2] = xh_ld<PIPE>(hc + r + 1); uC[1] = xh_ld<PIPE>(hc + r + 2);
                if (useB) { uB[3] = xh_ld<PIPE>(hb + r); uB[2] = xh_ld<PIPE>(hb + r + 1); uB[1] = xh_ld<PIPE>(hb + r + 2); }
            }
            pc = ld_col(n_first + 32 + k);
            ld_feed(r + 3 + 16 + k, pfd);
        }
        auto finish_stripe = [&]() {
            if (LocalR && k == 0) {
                int* b = tbest + 6 * st;
                xh_st<true>(b, max_val); xh_st<true>(b + 1, max_ulk); xh_st<true>(b + 2, max_mr); xh_st<true>(b + 3, max_nr); xh_st<true>(b + 4, max_ml);
            }
            if (UDH && is_imd_) {                                             // (lane k8 holds them)
                if (k == k8) { xh_st<true>(rlf + 3 * imd_i, rl0); xh_st<true>(rlf + 3 * imd_i + 1, rl1); xh_st<true>(rlf + 3 * imd_i + 2, rl2); }
            }
            if (st > 0) wait_for(st - 1, INT32_MAX);                          // finished = all stripes up to this one are
            publish(st, INT32_MAX);
        };
        if (PIPE && n >= n9) finish_stripe();
        int jb = 0;                                                           // step within the block
        for ( ; n < n9; ++n, ++r, q = (q == 5 ? 0 : q + 1)) {
            if (jb == 16) {                                                   // a new block: its loads become current, the next one's start
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ring[(n + 16 + k) & (HXRING - 1)] = pc;
#pragma unroll
                for (int a = 0; a < FD_N; ++a) fd[a][k] = pfd[a];
                if (PIPE) { publish(st, r - 1 - 6 * 15 - 2 + HXPROG0); ready(r + 3 + 15 + 16); }
                pc = ld_col(n + 32 + k);
                ld_feed(r + 3 + 16 + k, pfd);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                jb = 0;
            }
            const int j = jb++;
            const int f3 = q >= 3 ? q - 3 : q;
            const int nb = max(0, n - b_right + 1);
            const int kb = (nb - 1) / 3;
            const int ke = min(j9, (n - b_left) / 3);
            const int c = n - 3 * k;                          // my column
            const int4 col = ring[c & (HXRING - 1)];
            const int cx = col.x;
            // coding-potential pipe (:870-877): lane 0 feeds its column's value while the window lasts, every lane hands
            // the value it used to the lane below, which uses it three steps later
            int cv = CP[3];
            if (k == 0 && spj && !nb) cv = (int) (short) (cx & 0xffff);
            CP[0] = xh_shr1(cv, cv);
            // the row of the lane above (its cell of three steps ago); lane 0: the previous stripe's bottom row
            uH[0] = xh_shr1(fd[FD_HV][j], HV[3]);
            uC[0] = xh_shr1(fd[FD_HC][j], HC[3]);
            const int uF3 = xh_shr1(fd[FD_FV][j], FV[3]);
            const int uFC3 = xh_shr1(fd[FD_FC][j], FC[3]);
            uB[0] = useB ? xh_shr1(fd[FD_HB][j], HB[3]) : 0;
            const int uFB3 = UL ? xh_shr1(fd[FD_FB][j], FB[3]) : 0;
            // insertion: frame shifts, codon insertion, extension (:879-915)
            int ev, eb, ec;
            {
                int h = xh_add(HV[1], g1), hbb = HB[1], hcc = HC[1];
                int x = xh_add(HV[2], g2);
                bool mk = h > x;
                h = mk ? h : x; hbb = mk ? hbb : HB[2]; hcc = mk ? hcc : HC[2];
                x = xh_add(xh_add(HV[3], g3), cv);
                mk = h > x;
                h = mk ? h : x; hbb = mk ? hbb : HB[3]; hcc = mk ? hcc : HC[3];
                x = xh_add(xh_add(EV[3], gep), cv);
                mk = x > h;
                ev = mk ? x : h; eb = mk ? EB[3] : hbb; ec = mk ? EC[3] : hcc;
            }
            EV[0] = ev; EC[0] = ec;
            EB[0] = UL ? eb : EB[3];                          // (the plane keeps its word unless the `ml` lanes are carried)
            // deletion (:917-965)
            int fvv, fbv, fcv;
            {
                int f = xh_add(uF3, gep), fbb = uFB3, fcc = uFC3;
                int x = xh_add(uH[0], g3);
                bool mk = f > x;
                f = mk ? f : x; fbb = mk ? fbb : (UL ? uB[0] : 0); fcc = mk ? fcc : uC[0];
                x = xh_add(uH[1], g2);
                mk = f > x;
                f = mk ? f : x; fbb = mk ? fbb : (UL ? uB[1] : 0); fcc = mk ? fcc : uC[1];
                x = xh_add(uH[2], g1);
                mk = f > x;
                f = mk ? f : x; fbb = mk ? fbb : (UL ? uB[2] : 0); fcc = mk ? fcc : uC[2];
                fvv = f; fbv = fbb; fcv = fcc;
            }
            FV[0] = fvv; FC[0] = fcv;
            FB[0] = UL ? fbv : FB[6];
            // diagonal (:967-1027)
            if (nb) sm = 0;
            if (k >= kb && k < ke) sm = xh_w16(mrow[(cx >> 16) & 0xff]);
            const int qv = uH[3], qc = uC[3], qbb = uB[3];    // the diagonal predecessor: the lane above, six steps ago
            int qb = 0;
            {
                int h = xh_add(xh_add(sm, qv), cv);
                bool mk = fvv > h;
                h = mk ? fvv : h;
                int hcc = mk ? fcv : qc, hbb = mk ? fbv : qbb, code = mk ? 2 : 0;
                mk = ev > h;
                h = mk ? ev : h; hcc = mk ? ec : hcc; hbb = mk ? eb : hbb; code = mk ? 1 : code;
                PV[0] = code;
                PS[0] = PS[3] & code;
                if (!local) { if (!(h > XNEV)) h = XNEV; }
                else if (LocalL) {
                    if (0 > h) { h = 0; if (!UDH) { code = 1; hcc = 0; } }
                }
                if constexpr (!UDH) {
                    const int diag = code == 0;
                    qb = diag & ~qbb & 1;
                    hbb = diag;
                }
                HV[0] = h; HC[0] = hcc;
                HB[0] = useB ? hbb : 0;
            }
            if (UL && k >= kb && k < ke && HV[0] == 0) { HB[0] = ml + k; HC[0] = r - 6 * k; }
            if (LocalR) {                                     // first maximum over lanes 0 .. j8 (vmax)
                int bv = (k < j9) ? HV[0] : -0x7fffffff, bk = k;
                for (int o = 1; o < XN; o <<= 1) {
                    const int ov = __shfl_xor(bv, o, XN), ok = __shfl_xor(bk, o, XN);
                    if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
                }
                if (bv > max_val) {
                    max_val = bv;
                    max_ulk = __shfl(HC[0], bk, XN);
                    if constexpr (UDH) { max_ml = __shfl(HB[0], bk, XN); max_mr = ml + bk + 2; max_nr = n - 3 * (bk + 1); }
                    else { max_mr = ml + bk + 1; max_nr = n - 3 * bk; }
                }
            }
            if constexpr (!UDH)
                if (k >= kb && k < ke && qb) HC[0] = vadd(ml + k, n - 3 * (k + 1), HC[0]);

            const bool in_q = site_lane && c >= n_first && c < b_right;
            const unsigned fl = in_q ? (unsigned) cx >> 24 : 0u;
            bool is_acc = (((fl & 7) == 3) || (fl & 4)) && ncand >= 0;
            if (is_acc && !has_cip) {
                // screen: the best candidate, priced as high as anything can be, against the lowest of the nine cells
                // a candidate may raise (every update below is behind `x > cell`)
                const int s3 = (fl & 4) ? (int) (short) ((unsigned) col.y >> 16) : (int) (short) (col.y & 0xffff);
                const int sE = (int) (short) (ring[(c - 1) & (HXRING - 1)].w & 0xffff);
                const int lo = min(min(min(HV[0], HV[1]), min(HV[2], EV[0])), min(min(EV[1], EV[2]), min(min(FV[0], FV[1]), FV[2])));
                is_acc = c_val[0] + s3 + gain_max + abs(sE) > lo;
            }
            const bool is_don = (((fl >> 3) & 7) == 3) || ((fl >> 3) & 4);
            if (is_acc || is_don) {
                const int4 colm1 = ring[(c - 1) & (HXRING - 1)];          // the site itself: acc = don = c - 1
                // intron 3' boundary (:1049-1054, Sjsites::get :388-494): my column is a queued acceptor column
                if (is_acc) {
                    const int acc = c - 1;
                    const int rr0 = acc - 3 * (m + 1);
                    const int s3 = (fl & 4) ? (int) (short) ((unsigned) col.y >> 16) : (int) (short) (col.y & 0xffff);     // sig3[acc]
                    const int d3 = colm1.z & 15;
                    const int sigE_acc = (int) (short) (colm1.w & 0xffff);
                    // the bases behind the acceptor, for a codon the intron splits (SpJunc::spjseq)
                    const int t2 = acc > P.b_len ? 2 : ((ring[(c + 1) & (HXRING - 1)].x >> 16) & 0xff);
                    const int t3 = acc + 1 > P.b_len ? 2 : ((ring[(c + 2) & (HXRING - 1)].x >> 16) & 0xff);
                    const int w2 = t2 < 32 ? (int) s_mid[t2] : 7, w3 = t3 < 32 ? (int) s_mid[t3] : 7;
                    const bool acc_ok = acc < b_right && w2 <= 3 && w3 != 7;
                    const int cip_base = has_cip ? P.cip_off + 3 * (m + 1) : -1;
                    bool mx_on[3] = {false, false, false}; int mx_val[3] = {0, 0, 0}, mx_phs[3] = {0, 0, 0}, mx_ulk[3] = {0, 0, 0};
                    bool b_on = false; int b_val = 0, b_dir = 0;
#pragma unroll
                    for (int l = 0; l < 5; ++l) {
                        if (l > ncand) continue;
                        const int d = (c_dk[l] >> 12) & 3, phs = ((c_dk[l] >> 14) & 3) - 1, don = c_jnc[l];
                        const int rr = rr0 + phs;
                        if (rr < lw || rr >= up) continue;
                        if (d == 2 && phs == 1) continue;
                        const int len = acc - don;
                        if (len < minl) continue;
                        const int pen = len < 0 ? -32768 : (len < 4096 ? (int) s_ipen[len]
                                      : (A.ipen_runs ? ipen_runs_get(s_runs, len, A.intpen_len) : (int) A.intpen[min(len, A.intpen_len - 1)]));
                        const int dk = c_dk[l];
                        int x = c_val[l] + pen + s3 + s_t53[16 * ((dk >> 8) & 15) + d3];
                        if (cip_base >= 0) x += A.cip[cip_base - phs];      // Cip_score::cip_score, fwd2h1_simd.h:407
                        if (d == 0 && phs) {
                            int c0 = 2, c1 = 2;
                            const int w0 = dk & 7, w1 = (dk >> 3) & 7;
                            if ((dk & 64) && acc_ok && w0 != 7 && w1 <= 3) {
                                if (w0 <= 3) c0 = s_tron[16 * w0 + 4 * w1 + w2];
                                if (w3 <= 3) c1 = s_tron[16 * w1 + 4 * w2 + w3];
                            }
                            if (phs == 1) x += mrow_m[c0];
                            else x += mrow_m1[c1] - mrow_m1[t2] - sigE_acc;
                        }
                        const int a = 1 - phs;                  // the cell this candidate reaches: this step's, or one / two steps ago
                        const int cur = d == 0 ? SEL3(a, HV[0], HV[1], HV[2]) : (d == 1 ? SEL3(a, EV[0], EV[1], EV[2]) : SEL3(a, FV[0], FV[1], FV[2]));
                        if (x <= cur) continue;
                        const int cval = c_val[l], culk = c_ulk[l], cml = c_ml[l];
#pragma unroll
                        for (int dd = 0; dd < 3; ++dd)
                            if (dd == d && (!mx_on[dd] || x > mx_val[dd])) {
                                mx_on[dd] = true; mx_val[dd] = cval; mx_phs[dd] = phs; mx_ulk[dd] = culk;
                                if (!b_on || x > b_val) { b_on = true; b_val = cval; b_dir = d; }
                            }
                        const int w = xh_w16(x);
                        int lk = culk;
                        if constexpr (!UDH) {
                            const int inner = vadd(m + 1, don + phs, culk);
                            lk = vadd(m + 1, acc + phs, inner);
                        }
                        int hnow = 0, hbnow = 0, hcnow = 0;     // H of that cell afterwards (for the boundary row)
#pragma unroll
                        for (int aa = 0; aa < 3; ++aa)
                            if (aa == a) {
                                if (d == 0) { HV[aa] = w; HB[aa] = cml; HC[aa] = lk; }
                                else if (d == 1) { EV[aa] = w; EB[aa] = cml; EC[aa] = lk; }
                                else { FV[aa] = w; FB[aa] = cml; FC[aa] = lk; }
                                PS[aa] |= xh_psp_bit(d);
                                if (d && w > HV[aa]) { HV[aa] = w; HB[aa] = cml; HC[aa] = lk; }
                                hnow = HV[aa]; hbnow = HB[aa]; hcnow = HC[aa];
                            }
                        if (k + 1 == XN) {
                            xh_st<PIPE>(hv + rr, hnow);
                            if (imd_lane) xh_st<PIPE>(LNK(imd_i, 0, 0, rr), culk);
                            else { xh_st<PIPE>(hb + rr, hbnow); xh_st<PIPE>(hc + rr, hcnow); }
                            if (d == 2) {
                                xh_st<PIPE>(fv + rr, w);
                                if (imd_lane) xh_st<PIPE>(LNK(imd_i, 0, 1, rr), culk);
                                else { xh_st<PIPE>(fb + rr, cml); xh_st<PIPE>(fc + rr, lk); }
                            }
                        }
                    }
                    if constexpr (UDH) {
                        if (imd_lane && b_on) {
                            const int maxd = b_dir;
                            const int p_phs = SEL3(maxd, mx_phs[0], mx_phs[1], mx_phs[2]);
                            const int p_ulk = SEL3(maxd, mx_ulk[0], mx_ulk[1], mx_ulk[2]);
                            const int a = 1 - p_phs;
                            int fr = f3 - a; if (fr < 0) fr += 3;     // (q - a) mod 3: the frame of that cell
                            const int lstr = acc + p_phs - mm3;
                            if (fr == 0) rl0 = lstr; else if (fr == 1) rl1 = lstr; else rl2 = lstr;
                            xh_st<PIPE>(LNK(imd_i, 0, 0, lstr), p_ulk);
#pragma unroll
                            for (int aa = 0; aa < 3; ++aa)
                                if (aa == a) {
                                    if (maxd == 0) HC[aa] = lstr; else if (maxd == 1) EC[aa] = lstr; else FC[aa] = lstr;
                                    PV[aa] = maxd;
                                    if (maxd) HC[aa] = lstr;
                                    else {
                                        if (mx_on[1] && EV[aa] > HV[aa] + gop) { xh_st<PIPE>(LNK(imd_i, 0, 1, lstr), mx_ulk[1]); EC[aa] = lstr + width; }
                                        if (mx_on[2] && FV[aa] > HV[aa] + gop) FC[aa] = lstr + width;
                                    }
                                }
                        }
                    }
                }
                // intron 5' boundary (:1056-1061, Sjsites::put :496-543)
                if (is_don) {
                    const int don = c - 1;
                    const int sigJ = colm1.w >> 16;           // sig5[don]
                    // what an acceptor will need of this column: junction class, the two bases before the donor
                    int dk;
                    {
                        const int t0 = (don - 2 < 0 || don - 2 > P.b_len) ? 2 : ((colm1.x >> 16) & 0xff);     // bcode(don - 2)
                        const int t1 = (don - 1 < 0 || don - 1 > P.b_len) ? 2 : ((cx >> 16) & 0xff);          // bcode(don - 1)
                        const int w0 = t0 < 32 ? (int) s_mid[t0] : 7, w1 = t1 < 32 ? (int) s_mid[t1] : 7;
                        dk = w0 | (w1 << 3) | ((don >= b_left) ? 64 : 0) | (((colm1.z >> 4) & 15) << 8);
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {             // phs = 1, 0, -1: the cell of this step, one step, two steps ago
                        const int phs = 1 - a;
                        const int rr = don - 3 * (m + 1) + phs;
                        if (rr < lw || rr >= up) continue;
                        const int h = PV[a];
                        const int thr = HV[a] + gop;
                        int frm = f3 - a; if (frm < 0) frm += 3;
#pragma unroll
                        for (int kk = 0; kk < 3; ++kk) {
                            if (kk == 0 && h && phs < 1) continue;
                            if (PS[a] & xh_psp_bit(kk)) continue;
                            const bool cross = phs == 1 && kk == 0;
                            const int from = cross ? qv : (kk == 0 ? HV[a] : (kk == 1 ? EV[a] : FV[a]));
                            if (kk && from <= thr) continue;
                            const int x = from + sigJ;
                            if (x <= XNEV) continue;
                            // a full list whose fourth entry beats x: the free slot stays below it, and the list is four long after
                            if (ncand >= 3 && c_val[3] > x) { ncand = 3; continue; }
                            // the free slot starts below the list and moves up past every entry x ties or beats
                            int pos = ncand < 4 ? ncand + 1 : 4;
                            if (ncand < 4) ++ncand;
#pragma unroll
                            for (int l = 4; l >= 1; --l)
                                if (pos == l && x >= c_val[l - 1]) {
                                    c_val[l] = c_val[l - 1]; c_jnc[l] = c_jnc[l - 1]; c_dk[l] = c_dk[l - 1];
                                    c_ml[l] = c_ml[l - 1]; c_ulk[l] = c_ulk[l - 1];
                                    pos = l - 1;
                                }
                            if (pos < 4) {
                                const int n_ml = kk == 0 ? HB[a] : (kk == 1 ? EB[a] : FB[a]);
                                int n_ulk;
                                if (imd_lane) {
                                    const int rq = c - a - mm3;
                                    if (kk == 1) xh_st<PIPE>(LNK(imd_i, 0, 0, rq), SEL3(frm, rl0, rl1, rl2));
                                    n_ulk = rq;
                                } else
                                    n_ulk = cross ? qc : (kk == 0 ? HC[a] : (kk == 1 ? EC[a] : FC[a]));
#pragma unroll
                                for (int l = 0; l < 4; ++l)
                                    if (l == pos) {
                                        c_val[l] = xh_w16(x); c_jnc[l] = don; c_dk[l] = dk | (kk << 12) | ((phs + 1) << 14);
                                        c_ml[l] = n_ml; c_ulk[l] = n_ulk;
                                    }
                            } else --ncand;
                        }
                    }
                }
            }
            // intermediate row (:1375-1384)
            if constexpr (UDH) {
                const int rj = r - 6 * k8;
                if (is_imd_ && rj >= lw && rj <= up && k == k8) {
                    // lane k8 holds the row's side lanes, lane k9 - 1 == k8 its H / F planes ([k9] in the reference's layout)
                    if (PV[0] == 0) { if (f3 == 0) rl0 = rj; else if (f3 == 1) rl1 = rj; else rl2 = rj; }
                    if (PV[0] == 1) xh_st<PIPE>(LNK(imd_i, 0, 0, rj), SEL3(f3, rl0, rl1, rl2));
                    xh_st<PIPE>(LNK(imd_i, 1, 0, rj), HC[0]);
                    HC[0] = rj;
                    xh_st<PIPE>(LNK(imd_i, 1, 1, rj), FC[0]);
                    FC[0] = rj + width;
                }
            }
            // hand the bottom row to the next stripe (:1063-1073 / :1386-1397)
            const int r0 = r - 6 * j8;
            if (k == j8 && j9 == ke && lw <= r0 && (UDH ? r0 < up : r0 <= up)) {
                xh_st<PIPE>(hv + r0, HV[0]); xh_st<PIPE>(hc + r0, HC[0]);
                xh_st<PIPE>(fv + r0, FV[0]); xh_st<PIPE>(fc + r0, FC[0]);
                if constexpr (!UDH) xh_st<PIPE>(hb + r0, HB[0]);
                else if (LocalL) { xh_st<PIPE>(hb + r0, HB[0]); xh_st<PIPE>(fb + r0, FB[0]); }
            }
            // one step older
#pragma unroll
            for (int a = 3; a >= 1; --a) {
                HV[a] = HV[a - 1]; FV[a] = FV[a - 1]; EV[a] = EV[a - 1];
                HB[a] = HB[a - 1]; EB[a] = EB[a - 1];
                EC[a] = EC[a - 1];
                PS[a] = PS[a - 1]; CP[a] = CP[a - 1];
                uH[a] = uH[a - 1]; uC[a] = uC[a - 1]; uB[a] = uB[a - 1];
            }
#pragma unroll
            for (int a = 6; a >= 1; --a) FB[a] = FB[a - 1];
#pragma unroll
            for (int a = PIPE ? 3 : 6; a >= 1; --a) { HC[a] = HC[a - 1]; FC[a] = FC[a - 1]; }
            PV[2] = PV[1]; PV[1] = PV[0];
            if (PIPE && n == n9 - 1) finish_stripe();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        q_end = q;
        if constexpr (UDH) {
            if (is_imd_) {
                // rlst lives on the lane that held the intermediate row; the next one may sit on another lane, and the
                // reference keeps one array for all: pass it on
                rl0 = __shfl(rl0, k8, XN); rl1 = __shfl(rl1, k8, XN); rl2 = __shfl(rl2, k8, XN);
                ++imd_i;
            }
        }
    }
    if (PIPE) {
        const int st = my_stripe;
        if (ml_first >= ml_end) {                                             // (a problem without rows: no stripe loop ran)
            if (st > 0) wait_for(st - 1, INT32_MAX);
            publish(st, INT32_MAX);
        }
        if (st != n_stripes - 1) return;
        if (LocalR) {                                                         // stripes in order: the first maximum wins
            max_val = XNEV; max_ulk = X_EOU; max_ml = a_left; max_mr = a_right; max_nr = b_right;
            for (int t = 0; t < n_stripes; ++t) {
                const int* b = tbest + 6 * t;
                const int v = xh_ld<true>(b);
                if (v > max_val) { max_val = v; max_ulk = xh_ld<true>(b + 1); max_mr = xh_ld<true>(b + 2); max_nr = xh_ld<true>(b + 3); max_ml = xh_ld<true>(b + 4); }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- fhlastH1 (:691-791): the scans over the last row / the last column one entry after the other on lane 0,
    // their inputs (boundary entries, signals) staged through LDS by the whole group, 128 entries at a time
    auto HVr = [&](int i) -> int { return xh_ld<PIPE>(hv + i); };
    int ptr = 0, maxt = 0;
    const bool by_last = UDH ? !(LocalR && max_mr < a_right) : (!LocalR || max_mr == a_right);
    const int m3 = 3 * a_right;
    const int rr = b_right - m3;
    int maxr = rr, mx = rr, hmx = 0;
    if (by_last) {
        int* const sH = reinterpret_cast<int*>(s_col[g16]);        // 512 ints: 128 entries, 130 x {sigT | sigE << 16}, 130 x sig5
        int* const sYZ = sH + 128;
        int* const sW = sH + 260;
        int gl0 = 0, gl1 = 0, gl2 = 0;                   // glen[f], tcdn[f], rotated with f
        bool tc0 = false, tc1 = false, tc2 = false;
        int rw = lw;
        int rf = b_left - m3;
        if (rf > rw) rw = rf; else rf = rw;
        const int rw0 = rw;
        hmx = k == 0 ? HVr(mx) : 0;                      // hv[mx]
        int bb = rw + m3;
        if (a_exgr) {
            int p1 = 0, p2 = 0, p3 = 0;                  // hv[h - 1], [h - 2], [h - 3] as they stand (after this loop's writes)
            for (int h0 = rw; h0 <= rr; h0 += 128, bb += 128) {
                const int cnt = min(128, rr - h0 + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int i = k; i < cnt; i += XN) sH[i] = HVr(h0 + i);
                for (int i = k; i < cnt + 2; i += XN) {
                    const short4 a = aux[max(bb - 2 + i, 0)];
                    sYZ[i] = (int) ((unsigned) (unsigned short) a.y | ((unsigned) (unsigned short) a.z << 16));
                    sW[i] = a.w;
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (k == 0) {
                    for (int i = 0; i < cnt; ++i, ++rf) {
                        const int h = h0 + i;
                        gl0 += 3;
                        int hvh = sH[i];
                        int c0 = hvh, c1 = XNEV, c2 = XNEV;
                        const int yz = sYZ[i];                        // aux[bb - 2]: sigT, sigE
                        const int a2y = (int) (short) (yz & 0xffff), a2z = yz >> 16;
                        if (rf - rw0 >= 3 && !tc0) {
                            c1 = p3 + a2z;
                            if (!(a_exgr & 2)) c1 += gext3(gl0);
                            if (!(a_exgr & 1) && gl0 == 3) c1 += gop;
                            if (sc->term_codon) c2 = p3 + a2y;
                        }
                        if (rf - rw0 >= 3) tc0 = tc0 || a2y > 0;
                        const int s5r = sW[i + 2];                    // aux[bb].w
                        const int s5 = (local && s5r > 0) ? s5r : 0;
                        c0 += s5; c1 += s5;
                        int kk = 0, cb = c0;
                        if (c1 > cb) { kk = 1; cb = c1; }
                        if (c2 > cb) { kk = 2; cb = c2; }
                        if (kk == 0) { gl0 = 0; tc0 = false; }
                        else if (kk == 1) { hvh = xh_w16(c1 - s5); xh_st<PIPE>(hv + h, hvh); }
                        else { hvh = xh_w16(c2); xh_st<PIPE>(hv + h, hvh); }
                        if (h == mx) hmx = hvh;
                        if (hvh > hmx) { mx = h; hmx = hvh; maxr = rf - (kk == 2 ? 3 : 0); }
                        p3 = p2; p2 = p1; p1 = hvh;
                        { const int t_ = gl0; gl0 = gl1; gl1 = gl2; gl2 = t_; }
                        { const bool t_ = tc0; tc0 = tc1; tc1 = tc2; tc2 = t_; }
                    }
                }
            }
        } else if (k == 0) {
            const int y = xh_w16(HVr(rr - 3) + aux[bb + (rr - rw)].y);
            if (y > HVr(rr)) { xh_st<PIPE>(hv + rr, y); maxr = rr - 3; if (mx == rr) hmx = y; }
        }
        if (b_exgr) {
            rw = min(up - 1, b_right - 3 * a_left);
            int ga = XNEV, gb_ = XNEV, gc = XNEV;        // g[f], rotated with f
            int n3 = 0, n2 = 0, n1 = 0;                  // hv[h + 3], [h + 2], [h + 1] as they stand
            if (k == 0 && rw - 3 > rr) { n3 = HVr(rw); n2 = HVr(rw - 1); n1 = HVr(rw - 2); }
            for (int h0 = rw - 3; h0 > rr; h0 -= 256) {
                const int cnt = min(256, h0 - rr);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int i = k; i < cnt; i += XN) sH[i] = HVr(h0 - i);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (k == 0) {
                    for (int i = 0; i < cnt; ++i) {
                        const int h = h0 - i;
                        int x = n3;
                        if (!(b_exgr & 1)) x = xh_w16(x + gop);
                        if (x > ga) ga = x;
                        if (!(b_exgr & 2)) ga = xh_w16(ga + gep);
                        int hvh = sH[i];
                        if (hvh > ga) ga = XNEV;
                        else if (ga > hmx) { mx = h; hvh = ga; hmx = ga; xh_st<PIPE>(hv + h, ga); }
                        n3 = n2; n2 = n1; n1 = hvh;
                        { const int t_ = ga; ga = gb_; gb_ = gc; gc = t_; }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (k) return;
    if (by_last) {
        maxt = mx;
        if constexpr (UDH) xh_st<PIPE>(hb + maxt, xh_ld<PIPE>(hb + maxr));
        max_ulk = xh_ld<PIPE>(hc + maxr);
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    DevResultH R;
    R.score = max_val; R.mr = max_mr; R.nr = max_nr; R.maxt = maxt; R.maxr = 0; R.pad[0] = max_ulk; R.pad[2] = 0;
    R.pad[1] = 0;                                 // 1: the walk below met a poisoned link (run again, one group)
    A.res[pi] = R;
    bool poisoned = false;

    if constexpr (!UDH) {
        // Vmf::traceback(ptr) + the fix-up of trcbkalignH_ng
        int2* out = A.skl + (int64_t) pi * A.skl_cap;
        const int vn = xh_ld<PIPE>(vcount);
        int cnt = 0, status = vn > vcap ? -3 : 0;
        {   // mode 3 keeps the record pointer in one int16 lane (undefined in the reference beyond 32767 records)
            const int mq = a_right - a_left;
            float cvol = (float) (lw - b_left + 3 * a_right);
            cvol = (float) mq * (float) (b_right - b_left) - cvol * cvol / 3;
            if (!status && cvol < 65535.f && vn > 32767) status = -4;
        }
        if (ptr && !status) {
            int cur = ptr, lm = 0, ln = 0;
            for (;;) {
                if (PIPE && cur == HXPOISON) { poisoned = true; break; }
                const int sm_ = xh_ld<PIPE>(vraw + 3 * (int64_t) cur), sn_ = xh_ld<PIPE>(vraw + 3 * (int64_t) cur + 1);
                const int sp_ = xh_ld<PIPE>(vraw + 3 * (int64_t) cur + 2);
                if (cnt < A.skl_cap) out[cnt] = make_int2(sm_, sn_); else status = -1;
                lm = sm_; ln = sn_; ++cnt;
                if (!sp_) break;
                cur = sp_;
            }
            const int rd = local ? 0 : ((ln - 3 * lm) - b_left + 3 * a_left);
            if (rd) {
                const int2 rec = rd > 0 ? make_int2(a_left, b_left + rd) : make_int2(a_left - rd / 3, b_left);
                if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
                ++cnt;
            }
        }
        A.n_skl[pi] = poisoned ? 0 : (status ? status : cnt);
    } else {
        // the tail of hirschbergH1 (:1419-1469): walk the links back through the intermediate rows
        int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
        for (int i = 0; i < A.cpos_stride; ++i) cpos[i] = X_EOU;
        auto mi_of = [&](int i) { return a_left + (i + 1) * imd_step; };
        // a horizontal link that stands for "rlst of this frame as the intermediate rows above left it"
        auto hlnk = [&](int ii, int d, int rr_) -> int {
            int v = xh_ld<PIPE>(LNK(ii, 0, d, rr_));
            if (PIPE && v == HXPOISON) poisoned = true;
            if (PIPE && v >= HXINH && v < HXINH + 3) {
                const int fr = v - HXINH;
                v = 0x7fffffff;
                for (int j = ii - 1; j >= 0; --j) { const int w = xh_ld<true>(rlf + 3 * j + fr); if (w != HXINH + fr) { v = w; break; } }
            }
            return v;
        };
#define CPOS(i, c) cpos[(i) * 10 + (c)]
        int al = a_left, ar = a_right, bl = b_left, br = b_right;
        if (by_last) max_ml = LocalL ? xh_ld<PIPE>(hb + maxt) : a_left;
        ar = max_mr; br = max_nr;
        int val = max_val;
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
        int r = max_ulk;
        for ( ; i >= 0 && mi_of(i) > max_ml; --i) {
            int c = 0, d = 0;
            if (PIPE && (r == HXPOISON || poisoned)) { poisoned = true; break; }
            for ( ; r > up; r -= width) ++d;
            if (xh_ld<PIPE>(LNK(i, 1, d, r)) < X_EOU) {
                CPOS(i, c++) = mi_of(i);
                CPOS(i, c++) = (d > 0) ? 1 : 0;
                const int m3 = 3 * mi_of(i);
                for (int rp = hlnk(i, d, r); lw <= rp && rp < up && r != rp && c < 8; rp = hlnk(i, d, r = rp))
                    CPOS(i, c++) = r + m3;
                CPOS(i, c++) = r + m3;
                CPOS(i, c) = X_EOU;
                r = xh_ld<PIPE>(LNK(i, 1, d, r));
                if (r == X_EOU) break;
            } else
                CPOS(i, 0) = X_EOU;
        }
        if (PIPE && (r == HXPOISON || poisoned)) { poisoned = true; r = up; }
        for ( ; r > up; r -= width) ;
        if (LocalL) { al = max_ml; bl = r + 3 * al; }
        else {
            const int rl2_ = bl - 3 * al;
            if (b_exgl && rl2_ > r) {
                al = (bl - r) / 3;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = X_EOU;
            }
            if (a_exgl && rl2_ < r) bl = 3 * al + r;
        }
        ++i;
        if ((i >= 0 && i < n_im && mi_of(i) < al) || CPOS(i, 2) < bl) val = INT32_MIN / 16 * 7;
#undef CPOS
        A.scores[pi] = val;
        int* rg = A.ranges + 4 * (int64_t) pi;
        rg[0] = al; rg[1] = ar; rg[2] = bl; rg[3] = br;
    }
    if (PIPE && poisoned) A.res[pi].pad[1] = 1;
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
    const dim3 blk(64 * HXWPB);
    if (A.item_probs < 1 || A.item_probs > 4) A.item_probs = 4;
    if (A.pipe) {                                // one wave per (item_probs problems, stripe)
        const dim3 grd((A.n_items + HXWPB - 1) / HXWPB);
        if (udh) hipLaunchKernelGGL((spdh_exact<true, true>), grd, blk, 0, stream, A);
        else hipLaunchKernelGGL((spdh_exact<false, true>), grd, blk, 0, stream, A);
        return hipGetLastError();
    }
    const int per_block = HXWPB * A.item_probs;
    const dim3 grd((A.n_probs + per_block - 1) / per_block);
    if (udh) hipLaunchKernelGGL((spdh_exact<true, false>), grd, blk, 0, stream, A);
    else hipLaunchKernelGGL((spdh_exact<false, false>), grd, blk, 0, stream, A);
    return hipGetLastError();
}
