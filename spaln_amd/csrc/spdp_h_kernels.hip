// spdp_h_kernels.hip -- CDNA4 (gfx950) kernels for the aa x genome `_wip` engine of ogotoh/spaln
// (reference: SimdAln2h1::forwardH1_wip, src/fwd2h1_wip_simd.h:50-334; boundary set-up and end
// selection fhinitH1 / fhlastH1, src/fwd2h1_simd.h:546-785; traceback Anti_rhomb_coord<SHORT>
// with step 3, src/rhomb_coord.h:65-235).
//
// Mapping (same idea as spdp_kernels.hip): the reference sweeps stripes of 16 query rows, lane k of a
// stripe holding cell (m = ml+1+k, n-3k) at sweep step n.  One wave64 runs FOUR consecutive stripes
// of one problem, one per 16-lane DPP row, row g started SPDH_LAG blocks of 16 steps after row g-1 so
// that the boundary entries it reads (hv / fv by diagonal r = n - 3m) have been written.  A lane
// keeps its own last three H / F / E values (the reference's six rotating phase buffers hold
// exactly that) and pulls the upper row's value of three steps ago with one `row_shr:1` DPP move.
// Per-column inputs come from an LDS ring of packed records; lane k reads the record of its own
// column n - 3k.  All score arithmetic is int16-saturating (the value sits in the upper half of a 32-bit register and
// `v_add_i32 ... clamp` saturates it where `v_add_i16 ... clamp` would), i.e. exactly the
// reference's _mm256_adds_epi16 lanes -- including what the lanes outside the DP matrix compute,
// which the reference's end-cell selection can observe.
//
// Traceback codes go to HBM in the reference's own skewed layout (2 B per cell); the walk kernel
// replays Anti_rhomb_coord::go_back on it.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

// TraceBackCode values (src/rhomb_coord.h:36-61)
enum { C_DIAG = 1, C_HORI = 2, C_HORL = 3, C_HOR1 = 4, C_HOR2 = 5, C_VERT = 8, C_VERL = 9, C_VER1 = 10, C_VER2 = 11,
       C_ACCM = 13, C_ACCZ = 14, C_ACCP = 15, C_NHOR = 16, C_NVER = 32, C_NHOL = 64, C_DONM = 64,
       C_NVEL = 128, C_DONZ = 128, C_DONP = 256 };

typedef short s16;
__device__ __forceinline__ s16 sadd(s16 a, s16 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s16 smax(s16 a, s16 b) { return a > b ? a : b; }
// the sweep keeps its int16 scores in the UPPER half of 32-bit registers (value * 65536): `v_add_i32 ... clamp` then
// saturates exactly where `v_add_i16 ... clamp` does, at less than half the issue cost (profiles/r02_valu_ubench.txt:
// 7.7 cycles per wave-instruction for the 16-bit VOP3 form, 4 for the 32-bit one), and order comparisons are unchanged
typedef int q16;
#define Q16(x) ((q16) ((unsigned) (x) << 16))
__device__ __forceinline__ q16 qadd(q16 a, q16 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ q16 qmax(q16 a, q16 b) { return a > b ? a : b; }

#define DPP_ROW_SR(n) (0x110 + (n))
#define DPP_ROW_RR(n) (0x120 + (n))
// lane i of every 16-lane row <- lane i-1; lane 0 of the row keeps `old`
__device__ __forceinline__ int row_shr1(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, DPP_ROW_SR(1), 0xf, 0xf, false);
}
// lane 0 of every row <- lane 15 of that row
__device__ __forceinline__ int row_ror1(int src)
{
    return __builtin_amdgcn_mov_dpp(src, DPP_ROW_RR(1), 0xf, 0xf, true);
}

typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
// L1-bypassing loads for data another row of this wave stored a few blocks ago
__device__ __forceinline__ int2 ld_nt2(const int2* p)
{
    const v2i_t v = __builtin_nontemporal_load(reinterpret_cast<const v2i_t*>(p));
    return make_int2(v.x, v.y);
}
__device__ __forceinline__ int ld_nt1(const int* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ unsigned ld_nt_u16(const uint16_t* p) { return __builtin_nontemporal_load(p); }

#define SPDH_PEN_TAB 2048

// ---------------------------------------------------------------------------
template <bool B> struct BoolTag { static constexpr bool value = B; };

// SPJ: splice-aware (b->inex.intr); TAB: the intron-length penalty steps fit the LDS table
template <bool SPJ, bool TAB, bool LOCAL>
#ifndef SPDH_MINBLK
#define SPDH_MINBLK 5          // blocks per CU the register budget is set for (A/B: -DSPDH_MINBLK=4)
#endif
__global__ __launch_bounds__(256, SPDH_MINBLK) void spdh_sweep(HSweepArgs A)
{
    __shared__ int   s_mtx[32 * 32];
    __shared__ int   s_pen[SPDH_PEN_TAB];          // q16
    __shared__ int   s_qlen[8], s_qpen[8];
    __shared__ int4  s_ring[4][4][64];
    __shared__ int2  s_feed[4][4][16];
    __shared__ int2  s_out[4][4][17];           // bottom-row results of a block's steps (spdp_sweep_fp.hip: why not a DPP chain)

    const DevScoringH* __restrict__ sc = A.sc;
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) s_mtx[i] = Q16(sc->mtx[i]);      // q16
    if (threadIdx.x < 8) { s_qlen[threadIdx.x] = sc->qm_len[threadIdx.x]; s_qpen[threadIdx.x] = sc->qm_pen[threadIdx.x]; }
    const int nquant = sc->nquant;
    const int pen_cap = (nquant > 1) ? min(sc->qm_len[nquant - 2] + 1, SPDH_PEN_TAB - 1) : 0;
    // pen(hil) = qm_pen[j] for the last j with hil > qm_len[j-1]  (fwd2h1_wip_simd.h:229-233)
    for (int h = threadIdx.x; h <= pen_cap; h += blockDim.x) {
        int pv = sc->qm_pen[0];
        for (int j = 1; j < nquant; ++j) if (h > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
        s_pen[h] = Q16(pv);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane >> 4;                    // DPP row = stripe slot of the pass
    const int k = lane & 15;                    // lane within the stripe
    const int pi = __builtin_amdgcn_readfirstlane((int) blockIdx.x * 4 + wv);
    if (pi >= A.n_probs) return;                // one wave = one problem; no barrier below

    const DevProblemH P = A.probs[pi];
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const int m_width = P.m_width, n_width = P.n_width;
    const q16 ge = Q16(sc->gep), g1 = Q16(sc->g1), g2 = Q16(sc->g2), g3 = Q16(sc->g3);
    const int gop = sc->gop, gep = sc->gep;
    const int llmt = sc->llmt;
    int2* __restrict__ bnd = A.bnd + P.bnd_off;
    const int4* __restrict__ cols = A.cols + P.col_off;
    const short4* __restrict__ aux = A.aux + P.col_off;
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    uint16_t* __restrict__ tb = A.tb + P.tb_off;
    const int n_ent = P.buf_size + SPDH_BND_PAD;
    const int col_len = P.col_len;
#define BIDX(r) ((r) - lw + 3)
    auto gap_ext3 = [&](int i) { return i > sc->codonk1 ? sc->lgep : gep; };

    // =====================================================================================
    // fhinitH1 (src/fwd2h1_simd.h:546-689): boundary row by diagonal + codes of bitmap row 0
    // =====================================================================================
    {
        const int rl = b_left - 3 * a_left;
        int rr = min(b_right - 3 * a_left, up);
        int rr_g = rr;                                   // global: the ramp stops where it reaches nevsel
        if (!a_exgl && gep) rr_g = min(rr, (SPDH_NEV - sc->g3) / gep + rl + 4);
        for (int e = lane; e < n_ent; e += 64) {
            const int r = e + lw - 3;
            int h = SPDH_NEV, f = SPDH_NEV;
            if (b_exgl == 1 && r >= lw && r < rl) h = 0;
            if (b_exgl == 2 && r == rl) f = 0;
            if (!a_exgl) {
                const int i = r - rl;
                if (b_exgl && i == 0) f = 0;
                if (i == 0) h = 0;
                else if (i >= 1 && i <= 3) h = (i == 1) ? sc->g1 : (i == 2 ? sc->g2 : sc->g3);
                else if (i >= 4) {
                    const int gb = ((i - 1) % 3 == 0) ? sc->g1 : ((i - 1) % 3 == 1 ? sc->g2 : sc->g3);
                    if (gep) { if (r < rr_g) h = gb + ((i - 1) / 3) * gep; }
                    else if (r < rr) h = sc->g3;
                }
            }
            bnd[e] = make_int2(h, f);
        }
        // row 0 of the bitmap: initialize_m0(4) for global left ends, zero otherwise
        for (int c = lane; c < n_width; c += 64)
            tb[(int64_t) c * m_width] = (uint16_t) ((!a_exgl && c >= 1) ? 4 : 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (a_exgl) {
            // semi-global: best of "start here" (sigS) and "extend the leading gap", a sequential scan
            // -- run uniformly by the whole wave on 64-entry chunks of prefetched signals
            auto sigS0 = [&](int pos) { const int v = aux[pos].x; return v > 0 ? v : 0; };
            int hm3 = sigS0(b_left + 1), hm2 = sigS0(b_left + 2), hm1 = sigS0(b_left + 3);
            if (lane < 3 && rl + lane < rl + 3) {
                // the three frame starts are stored even when they lie beyond rr (as the reference does)
                const int v = lane == 0 ? hm3 : (lane == 1 ? hm2 : hm1);
                bnd[BIDX(rl + lane)].x = v;
            }
            int l0 = rl, l1 = rl + 1, l2 = rl + 2;       // lend[] rotated: l0 is the current frame's
            bool stopped = false;
            for (int r0 = rl + 3; r0 < rr && !stopped; r0 += 64) {
                const int rj = r0 + lane;
                const int bb = b_left + (rj - rl) + 1;    // position the SGPT6 pointer is at
                int vS = 0, vE = 0;
                if (rj < rr) { vS = aux[bb].x; vE = aux[bb - 3].z; }
                int myh = 0, mycode = 0; bool mine = false, nocode = false;
                const int cnt = min(64, rr - r0);
                for (int j = 0; j < cnt; ++j) {
                    const int r = r0 + j;
                    const int sS = __builtin_amdgcn_readlane(vS, j), sE = __builtin_amdgcn_readlane(vE, j);
                    const int gl = r - l0;
                    int h = hm3;
                    if (!(a_exgl & 1) && gl == 3) h += gop;
                    if (!(a_exgl & 2)) h += gap_ext3(gl);
                    h = (s16) (h + sE);
                    int code = 0;
                    const bool stop = h < SPDH_NEV;
                    if (!stop) {
                        int x = (s16) (hm1 + sc->g1);
                        if (x > h) { h = x; code = C_HOR1; }
                        x = (s16) (hm2 + sc->g2);
                        if (x > h) { h = x; code = C_HOR2; }
                        x = sS > 0 ? sS : 0;
                        if (x > h) { h = x; l0 = r; }
                        else code = C_HORI;
                    }
                    if (lane == j) { myh = h; mycode = code; mine = true; nocode = stop; }
                    hm3 = hm2; hm2 = hm1; hm1 = h;
                    const int t = l0; l0 = l1; l1 = l2; l2 = t;
                    if (stop) { stopped = true; break; }
                }
                if (mine) {
                    bnd[BIDX(rj)].x = myh;
                    if (!nocode) tb[(int64_t) (rj - rl) * m_width] = (uint16_t) mycode;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // =====================================================================================
    // the sweep (src/fwd2h1_wip_simd.h:101-331)
    // =====================================================================================
    const int n_stripes = (a_right - a_left + SPDH_NELEM - 1) / SPDH_NELEM;
    const bool LocalL = LOCAL && a_exgl && b_exgl;
    const bool LocalR = LOCAL && a_exgr && b_exgr;
    // running maximum for local right ends: value, then the first (stripe, step, lane) holding it
    q16 best_val = Q16(SPDH_NEV); int best_mr = a_right, best_nr = b_right;       // (q16 until the reduction below)
    unsigned long long best_key = ~0ull;
    int4* const ring = &s_ring[wv][g][0];
    int2* const feed = &s_feed[wv][g][0];
    auto pen_of = [&](int hil) -> q16 {
        if constexpr (TAB) return s_pen[min(hil, pen_cap)];
        int pv = s_qpen[0];
        for (int jq = 1; jq < nquant; ++jq) pv = (hil > s_qlen[jq - 1]) ? s_qpen[jq] : pv;
        return Q16(pv);
    };

    for (int s0 = 0; s0 < n_stripes; s0 += 4) {
        // ---- geometry of my stripe (row g of the wave)
        const int s = s0 + g;
        const int ml = a_left + s * SPDH_NELEM;
        const bool has = s < n_stripes;
        const int j9 = has ? min(SPDH_NELEM, a_right - ml) : 0;
        const int j8 = j9 - 1;
        const int n_start = max(b_left, lw + 3 * ml);
        const int n9 = min(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;
        const int len = has ? max(0, n9 + 1 - n_start) : 0;
        const int nb = (len + 15) >> 4;
        const int nb0 = __builtin_amdgcn_readlane(nb, 0), nb1 = __builtin_amdgcn_readlane(nb, 16),
                  nb2 = __builtin_amdgcn_readlane(nb, 32), nb3 = __builtin_amdgcn_readlane(nb, 48);
        const int tot = max(max(nb0, SPDH_LAG + nb1), max(2 * SPDH_LAG + nb2, 3 * SPDH_LAG + nb3));
        const int mp1 = ml + 1;
        const int e_base = -3 * ml - lw + 3;              // entry read as hv[r + 3] at step n: n + e_base
        const bool partial = j9 < SPDH_NELEM;

        const int acode = (k < j9) ? acod[ml + k] : SPDH_ZCODE;
        const int* mrow = s_mtx + acode * 32;

        // per-lane DP state (q16: the int16 value in the upper half of the register)
        q16 h1 = Q16(SPDH_NEV), h2 = Q16(SPDH_NEV), h3 = Q16(SPDH_NEV);          // my H one, two, three steps ago
        q16 f1 = Q16(SPDH_NEV), f2 = Q16(SPDH_NEV), f3 = Q16(SPDH_NEV);          // my F
        q16 e1 = Q16(SPDH_NEV), e2 = Q16(SPDH_NEV), e3 = Q16(SPDH_NEV);          // my E, by frame
        q16 u4 = Q16(SPDH_NEV), u5 = Q16(SPDH_NEV), u6 = Q16(SPDH_NEV);          // upper row's H four, five, six steps ago
        q16 hiv0 = Q16(SPDH_NEV), hiv1 = Q16(SPDH_NEV), hiv2 = Q16(SPDH_NEV);    // best donor so far, by phase
        int hil0 = 0, hil1 = 0, hil2 = 0;                          // columns since that donor
        int2* const outb = &s_out[wv][g][0];
        const bool is_bottom = k == ((j9 < SPDH_NELEM && j9 > 0) ? j8 : 15);     // (a partial last stripe: its last real row)
        // bitmap offset of my cell at step n_start (advances by m_width per step)
        int tb_off = (3 * (mp1 - a_left) + (n_start - b_left)) * m_width + (mp1 - a_left) + k;

        int2 nx_b = make_int2(0, 0);
        int4 nx_c = make_int4(0, 0, 0, 0);
        // column record as lanes see it: window edges applied
        // column record as lanes see it: window edges applied
        auto mask_col = [&](int4 rec, int c) -> int4 {
            if (c < 0 || c >= col_len) rec = make_int4(0, 0, 0, 0);
            if (c >= b_right) rec.x &= 0x00ffffff;                          // nothing splices at n >= b_right
            if (c < b_left + 3 || c > b_right + 2) rec.x = (rec.x & (int) 0xff00ffffu) | (SPDH_ZCODE << 16);
            return rec;
        };
        auto load_col = [&](int c) -> int4 { return mask_col(cols[min(max(c, 0), col_len - 1)], c); };
        auto prefetch = [&](int lbn) {
            const int nn = n_start + lbn * 16 + k;                  // sweep step this lane loads for
            const int e = min(nn + e_base, n_ent - 1);
            nx_b = ld_nt2(bnd + e);
            // raw record (masks are applied when it is consumed: nothing may depend on the loaded value
            // here, or the compiler waits for the load on the spot and the prefetch is gone)
            nx_c = cols[min(max(nn, 0), col_len - 1)];
        };
        auto run_pass = [&](auto partial_tag) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial_tag)::value;
        if (g == 0 && nb > 0) prefetch(0);
        for (int blk = 0; blk < tot; ++blk) {
            const int lb = blk - SPDH_LAG * g;                       // my local block number
            if (lb == -1 && nb > 0) prefetch(0);                    // one block ahead of first use
            if (lb >= 0 && lb < nb) {
                const int n0 = n_start + lb * 16;                   // sweep step of J = 0
                if (lb == 0) {
                    // stripe start: the pipes hold nothing yet (cp = 0, no signals) but the codons of
                    // in-window columns left of n_start; lane 0's upper neighbours come from the boundary
                    for (int i = 0; i < 3; ++i) {
                        const int c = n_start - 1 - k - 16 * i;
                        int4 rec = load_col(c);
                        rec.x &= 0x00ff0000; rec.y = 0; rec.z = 0;
                        ring[c & 63] = rec;
                    }
                    if (k == 0) {
                        const int e = n_start + e_base;
                        u4 = Q16(ld_nt1(&bnd[e - 1].x));
                        u5 = Q16(ld_nt1(&bnd[e - 2].x));
                        u6 = Q16(ld_nt1(&bnd[e - 3].x));
                    }
                }
                feed[k] = make_int2(Q16(nx_b.x), Q16(nx_b.y));
                ring[(n0 + k) & 63] = mask_col(nx_c, n0 + k);
                if (lb + 1 < nb) prefetch(lb + 1);
                asm volatile("" ::: "memory");      // wave-internal ordering: see spdp_kernels.hip WAVE_ORDER

#pragma unroll
                for (int J = 0; J < 16; ++J) {
                    const int n = n0 + J;
                    const int4 rec = ring[(n - 3 * k) & 63];
                    const q16 cv = Q16(rec.x);
                    const int tron = (rec.x >> 16) & 0xff;
                    const unsigned fl = (unsigned) rec.x >> 24;
                    // ---- horizontal: 1-nt / 2-nt frame shift, new codon insertion, extension
                    const q16 a1 = qadd(h1, g1), a2 = qadd(h2, g2);
                    bool m = a1 > a2;
                    q16 eh = qmax(a1, a2);
                    int eb = m ? C_HOR1 : C_HOR2;
                    const q16 a3 = qadd(qadd(h3, g3), cv);
                    m = eh > a3;
                    eh = qmax(eh, a3);
                    eb = m ? eb : C_HORI;
                    q16 ee = qadd(qadd(e3, ge), cv);
                    m = ee > eh;
                    ee = qmax(ee, eh);
                    int hb = m ? 0 : C_NHOR;
                    eb = m ? C_HORI : eb;
                    // ---- vertical: extension, codon deletion, 2-nt / 1-nt frame shift
                    const int2 fd = feed[J];
                    const q16 u3 = row_shr1(fd.x, h3);
                    const q16 uf = row_shr1(fd.y, f3);
                    q16 ff = qadd(uf, ge);
                    const q16 b3 = qadd(u3, g3), b2 = qadd(u4, g2);
                    m = b3 > b2;
                    q16 fh = qmax(b3, b2);
                    int pb = m ? C_VERT : C_VER1;
                    const q16 b1 = qadd(u5, g1);
                    m = fh > b1;
                    fh = qmax(fh, b1);
                    pb = m ? pb : C_VER2;
                    m = ff > fh;
                    ff = qmax(ff, fh);
                    hb |= m ? 0 : C_NVER;
                    pb = m ? C_VERT : pb;
                    // ---- diagonal
                    const q16 sm = mrow[tron];
                    const q16 dg = qadd(qadd(sm, u6), cv);
                    m = ff > dg;
                    q16 h = m ? ff : dg;
                    pb = m ? pb : C_DIAG;
                    m = ee > h;
                    h = m ? ee : h;
                    pb = m ? eb : pb;
                    bool ab = false;
                    if constexpr (SPJ) {
                        // ---- intron 3' boundary: candidate 0 (phase -1 / 0 / +1), candidate 1 (phase +1)
                        const unsigned c0 = fl & 3u;
                        const q16 s3_0 = Q16(rec.y), s3_1 = (q16) ((unsigned) rec.y & 0xffff0000u);
                        // (selects on copies: a ?: on the captured variables themselves would select
                        //  between their addresses and pin all of them to memory)
                        const q16 cv0 = hiv0, cv1 = hiv1, cv2 = hiv2;
                        const int cl0 = hil0, cl1 = hil1, cl2 = hil2;
                        const q16 shiv = (c0 == 1u) ? cv0 : ((c0 == 2u) ? cv1 : cv2);
                        const int shil = (c0 == 1u) ? cl0 : ((c0 == 2u) ? cl1 : cl2);
                        const q16 x0 = qadd(qadd(shiv, s3_0), pen_of(shil));
                        const q16 x1 = qadd(qadd(cv2, s3_1), pen_of(cl2));
                        // A score already below `nevsel` is LIFTED to it by the reference's non-matching
                        // blends (`Blend(qv, ninf, ..)` then `Cmp_gt(qv, hv)`, :237-243) whenever the pipe
                        // is not skipped as a whole (`AllZero(ph_v)`, :224).  Only dead cells can be that
                        // low, so the three-blend form runs only when the wave holds one.
                        if (__builtin_amdgcn_ballot_w64(h < Q16(SPDH_NEV)) == 0ull) {
                            m = (c0 != 0u) && (shil > llmt) && (x0 > h);
                            h = m ? x0 : h;
                            pb = m ? (int) (12u + c0) : pb;
                            ab = m;
                            m = (fl & 4u) && (cl2 > llmt) && (x1 > h);
                            h = m ? x1 : h;
                            pb = m ? C_ACCP : pb;
                            ab = ab || m;
                        } else {
                            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(c0 != 0u);
                            const unsigned long long b1 = __builtin_amdgcn_ballot_w64((fl & 4u) != 0u);
                            const unsigned long long rowm = 0xffffull << (16 * g);
                            const bool any0 = (b0 & rowm) != 0ull, any1 = (b1 & rowm) != 0ull;
#pragma unroll
                            for (int f = 0; f < 3; ++f) {
                                const int lf = f == 0 ? cl0 : (f == 1 ? cl1 : cl2);
                                const q16 cand = (c0 == (unsigned) (f + 1) && lf > llmt) ? x0 : Q16(SPDH_NEV);
                                m = any0 && cand > h;
                                h = m ? cand : h;
                                pb = m ? (C_ACCM + f) : pb;
                                ab = ab || (m && c0 != 0u);
                            }
                            const q16 cand = ((fl & 4u) && cl2 > llmt) ? x1 : Q16(SPDH_NEV);
                            m = any1 && cand > h;
                            h = m ? cand : h;
                            pb = m ? C_ACCP : pb;
                            ab = ab || (m && (fl & 4u));
                        }
                    }
                    if constexpr (LOCAL) {
                        // local left end (:243-247; accscr stays 0 without re-basing), right end (:250-258)
                        if (LocalL && h < 0) { h = 0; hb = 0; }
                        if (LocalR && k < j9 && n <= n9 && h >= best_val) {
                            const unsigned long long key =
                                ((unsigned long long) s << 40) | ((unsigned long long) (n - n_start) << 8) | (unsigned) k;
                            if (h > best_val || key < best_key) {
                                best_val = h; best_key = key; best_mr = ml + k + 1; best_nr = n - 3 * k;
                            }
                        }
                    }
                    if constexpr (SPJ) {
                        // ---- intron 5' boundary
                        const unsigned d0 = (fl >> 3) & 3u;
                        const q16 s5_0 = Q16(rec.z), s5_1 = (q16) ((unsigned) rec.z & 0xffff0000u);
                        const q16 pvH = ab ? Q16(SPDH_NEV) : qadd(h, s5_0);
                        const q16 pvD = ab ? Q16(SPDH_NEV) : qadd(u6, s5_0);
                        const q16 pvD1 = ab ? Q16(SPDH_NEV) : qadd(u6, s5_1);
                        m = (d0 == 1u) && (pvH > hiv0);
                        hiv0 = m ? pvH : hiv0; hil0 = m ? 0 : hil0; hb |= m ? C_DONM : 0;
                        m = (d0 == 2u) && (pvH > hiv1);
                        hiv1 = m ? pvH : hiv1; hil1 = m ? 0 : hil1; hb |= m ? C_DONZ : 0;
                        m = (d0 == 3u) && (pvD > hiv2);
                        hiv2 = m ? pvD : hiv2; hil2 = m ? 0 : hil2; hb |= m ? C_DONP : 0;
                        m = (fl & 32u) && (pvD1 > hiv2);
                        hiv2 = m ? pvD1 : hiv2; hil2 = m ? 0 : hil2; hb |= m ? C_DONP : 0;
                        // (the reference's int16 counters saturate at 32767; every threshold they are
                        //  compared with is far below, so plain increments decide identically)
                        ++hil0; ++hil1; ++hil2;
                    }
                    // ---- rotate the histories
                    h3 = h2; h2 = h1; h1 = h;
                    f3 = f2; f2 = f1; f1 = ff;
                    e3 = e2; e2 = e1; e1 = ee;
                    u6 = u5; u5 = u4; u4 = u3;
                    // ---- traceback code (the last stripe's lanes past a_right are masked, :308)
                    if (k < j9 && n <= n9) tb[tb_off] = (uint16_t) (hb | pb);
                    tb_off += m_width;
                    // ---- bottom lane of the stripe -> slot J of the row's output block (read back at the flush)
                    if (is_bottom) outb[J] = make_int2(h, ff);
                    // keep the unrolled steps apart: interleaving them only inflates register pressure
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- flush: lane i holds the bottom-row result of step j = 15 - i; it becomes the
                // boundary entry of its diagonal under the reference's condition (:301-305)
                {
                    const int j = 15 - k;
                    const int n = n0 + j;
                    const int r0 = n - 3 * mp1 - 6 * j8;
                    if (n <= n9 && n - b_left >= 3 * j9 && r0 >= lw && r0 <= up && j9 > 0)
                        { const int2 o = outb[j]; bnd[BIDX(r0)] = make_int2(o.x >> 16, o.y >> 16); }
                }
            }
            // boundary entries are exchanged between the rows of this wave through memory: a load
            // issued after a store of the same wave to the same address observes it (in-order vector
            // memory path, loads bypass L1), so only the compiler needs a fence here
            asm volatile("" ::: "memory");
        }
        };
        // only the last stripe of a problem can be partial (fewer than 16 rows)
        if ((s0 + 4 >= n_stripes) && ((a_right - a_left) & 15)) run_pass(BoolTag<true>{});
        else run_pass(BoolTag<false>{});
    }

    // =====================================================================================
    // fhlastH1 (src/fwd2h1_simd.h:691-785): end cell + edits of the last row's codes
    // =====================================================================================
    DevResultH R;
    R.score = SPDH_NEV; R.mr = a_right; R.nr = b_right; R.maxt = 0; R.maxr = 0;
    R.pad[0] = R.pad[1] = R.pad[2] = 0;
    bool run_last = true;
    if constexpr (LOCAL) {
        if (LocalR) {
            // reduce (value desc, key asc) over the wave
            for (int off = 32; off; off >>= 1) {
                const int ov = __shfl_xor(best_val, off);
                const unsigned long long ok = __shfl_xor(best_key, off);
                const int omr = __shfl_xor(best_mr, off), onr = __shfl_xor(best_nr, off);
                if (ov > best_val || (ov == best_val && ok < best_key)) {
                    best_val = ov; best_key = ok; best_mr = omr; best_nr = onr;
                }
            }
            R.score = best_val >> 16; R.mr = best_mr; R.nr = best_nr;
            run_last = best_mr == a_right;             // `if (!LocalR || maxh.mr == a->right)`, :330
        }
    }
    if (run_last) {
        const int m3 = 3 * a_right;
        const int rw = max(lw, b_left - m3);
        const int rr = b_right - m3;
        int maxr = rr, mx = rr;
        int mxval = (s16) ld_nt1(&bnd[BIDX(rr)].x);
        const int64_t rowbase = (int64_t) 3 * (m_width - 1) * m_width + (m_width - 1);   // + cur_n * m_width
        if (a_exgr) {
            int gl0 = 0, gl1 = 0, gl2 = 0;                 // glen[] / tcdn[] rotated with the frame
            bool tc0 = false, tc1 = false, tc2 = false;
            int hq1 = 0, hq2 = 0, hq3 = 0;                  // the (edited) values of the three previous diagonals
            for (int i0 = 0; rw + i0 <= rr; i0 += 64) {
                const int hj = rw + i0 + lane;
                const int bb = hj + m3;                     // genomic position of the cell
                int vH = 0, vE = 0, vT = 0, vC = 0, v5 = 0;
                const bool in = hj <= rr;
                const int64_t cidx = rowbase + (int64_t) (bb - b_left) * m_width;
                if (in) {
                    vH = (s16) ld_nt1(&bnd[BIDX(hj)].x);
                    if (bb - 2 >= 0) { const short4 ax = aux[bb - 2]; vE = ax.z; vT = ax.y; }
                    if constexpr (LOCAL) v5 = aux[bb].w;
                    vC = ld_nt_u16(&tb[cidx]);
                }
                int mycode = vC; bool changed = false;
                const int cnt = min(64, rr - (rw + i0) + 1);
                for (int j = 0; j < cnt; ++j) {
                    const int i = i0 + j;
                    const int hd = rw + i;
                    const int cand0 = __builtin_amdgcn_readlane(vH, j);
                    const int sE = __builtin_amdgcn_readlane(vE, j), sT = __builtin_amdgcn_readlane(vT, j);
                    gl0 += 3;
                    int cand1 = SPDH_NEV, cand2 = SPDH_NEV;
                    if (i >= 3 && !tc0) {
                        cand1 = hq3 + sE;
                        if (!(a_exgr & 2)) cand1 += gap_ext3(gl0);
                        if (!(a_exgr & 1) && gl0 == 3) cand1 += gop;
                        if (sc->term_codon) cand2 = hq3 + sT;
                    }
                    if (i >= 3) tc0 = tc0 || sT > 0;
                    int s5l = 0;
                    if constexpr (LOCAL) { s5l = __builtin_amdgcn_readlane(v5, j); s5l = s5l > 0 ? s5l : 0; }
                    int kk = 0, best = cand0 + s5l;
                    cand1 += s5l;
                    if (cand1 > best) { kk = 1; best = cand1; }
                    if (cand2 > best) { kk = 2; best = cand2; }
                    const int newh = (kk == 0) ? cand0 : (int) (s16) (kk == 1 ? best - s5l : best);
                    if (kk == 0) { gl0 = 0; tc0 = false; }
                    if (hd == mx) mxval = newh;
                    else if (newh > mxval) { mx = hd; mxval = newh; maxr = hd - (kk == 2 ? 3 : 0); }
                    if (lane == j) {
                        if (kk != 0) { mycode = C_HORI; changed = true; }
                        if (gl0 == 3) { mycode |= C_NHOR; changed = true; }
                    }
                    hq3 = hq2; hq2 = hq1; hq1 = newh;
                    { const int t = gl0; gl0 = gl1; gl1 = gl2; gl2 = t; }
                    { const bool t = tc0; tc0 = tc1; tc1 = tc2; tc2 = t; }
                }
                if (in && changed) tb[cidx] = (uint16_t) mycode;
            }
        } else {
            const int y = (s16) ((s16) ld_nt1(&bnd[BIDX(rr - 3)].x) + aux[b_right].y);
            if (y > mxval) { mxval = y; maxr = rr - 3; }
        }
        if (b_exgr) {
            const int rw2 = min(up - 1, b_right - 3 * a_left);
            int ga = SPDH_NEV, gb = SPDH_NEV, gc = SPDH_NEV;  // g[] rotated with the frame
            const int hs = rw2 - 3;
            if (hs > rr) {
                int hp1 = (s16) ld_nt1(&bnd[BIDX(hs + 1)].x), hp2 = (s16) ld_nt1(&bnd[BIDX(hs + 2)].x),
                    hp3 = (s16) ld_nt1(&bnd[BIDX(hs + 3)].x);
                for (int h0 = hs; h0 > rr; h0 -= 64) {
                    const int hj = h0 - lane;
                    int vH = 0;
                    if (hj > rr) vH = (s16) ld_nt1(&bnd[BIDX(hj)].x);
                    const int cnt = min(64, h0 - rr);
                    for (int j = 0; j < cnt; ++j) {
                        const int hd = h0 - j;
                        int cur = __builtin_amdgcn_readlane(vH, j);
                        int x = hp3;
                        if (!(b_exgr & 1)) x = (s16) (x + gop);
                        if (x > ga) ga = x;
                        if (!(b_exgr & 2)) ga = (s16) (ga + gep);
                        if (cur > ga) ga = SPDH_NEV;
                        else if (ga > mxval) { mx = hd; mxval = ga; cur = ga; }
                        hp3 = hp2; hp2 = hp1; hp1 = cur;
                        const int t = ga; ga = gb; gb = gc; gc = t;
                    }
                }
            }
        }
        R.maxt = mx; R.maxr = maxr;
        if (maxr - rr > 0) R.mr = (b_right - maxr) / 3;
        else R.nr = mx + m3;
    }
    if (lane == 0) A.res[pi] = R;
#undef BIDX
}

// ---------------------------------------------------------------------------
// Traceback walk: Anti_rhomb_coord<SHORT>::traceback / go_back with step = 3 (src/rhomb_coord.h:141-235)
// on the reference's own layout.  One wave per problem, all lanes walking the same path (the loads
// broadcast).  A cell the sweep never wrote reads as 0, as in the reference's zero-filled bitmap.
__global__ void spdh_walk(HWalkArgs A)
{
    const int pi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pi >= A.n_probs) return;
    const bool writer = (threadIdx.x & 63) == 0;
    const DevProblemH P = A.probs[pi];
    const uint16_t* __restrict__ tb = A.tb + P.tb_off;
    const int m_width = P.m_width;
    int2* out = A.skl + (int64_t) pi * A.skl_cap;
    int cnt = 0, status = 0;
    auto at = [&](int cm, int cn) -> unsigned {
        if (cm > 0) {
            const int s = (cm - 1) >> 4, kq = (cm - 1) & 15;
            const int ml = P.a_left + s * SPDH_NELEM;
            const int j9 = min(SPDH_NELEM, P.a_right - ml);
            const int n_start = max(P.b_left, P.lw + 3 * ml);
            const int n9 = min(P.b_right, P.up + 3 * (ml + j9) + 1) + 3 * j9;
            const int step = cn + P.b_left + 3 * kq;
            if (step < n_start || step > n9) return 0u;
        } else if (cn >= P.n_width) return 0u;
        return tb[(int64_t) (3 * cm + cn) * m_width + cm];
    };
    int cm = A.res[pi].mr - P.a_left, cn = A.res[pi].nr - P.b_left;   // cursor (cur_m, cur_n)
    int m = cm, n = cn;                                                // what go_back reports
    auto emit = [&]() {
        if (cnt < A.skl_cap) { if (writer) out[cnt] = make_int2(m + P.a_left, n + P.b_left); }
        else status = -1;
        ++cnt;
    };
    auto to_left = [&](int s) -> unsigned {
        m = cm; n = cn -= s;
        if (n < 0) { cn = n = 0; return 0u; }
        return at(cm, cn);
    };
    auto to_upper = [&](int s) -> unsigned {
        m = --cm; n = cn -= s;
        if (m < 0) { cm = m = 0; cn = n += s; return 0u; }
        if (n < 0) { if (s > 0) cm = m -= n / s; cn = n = 0; return 0u; }
        return at(cm, cn);
    };
    unsigned code = 0;
    {
        // a start cell outside the reference's allocation is an out-of-bounds read there: flagged
        const int64_t sp = (int64_t) (3 * cm + cn) * m_width + cm;
        if (cm < 0 || cn < 0 || sp >= P.tb_size) status = -3;
        else code = at(cm, cn);
    }
    long guard = 4l * ((long) (P.a_right - P.a_left) + (P.b_right - P.b_left)) + 256;
    while (code && guard-- > 0) {
        emit();
        const unsigned dir = code & 15u;
        if (dir == C_DIAG) {
            do { code = to_upper(3); } while (code && (code & 15u) == C_DIAG);
        } else if (dir == C_HORI || dir == C_HORL) {
            const unsigned flag = (dir == C_HORI) ? C_NHOR : C_NHOL;
            bool dead = false;
            while (!(code & flag)) { code = to_left(3); if (!code) { dead = true; break; } }
            if (!dead) { const unsigned d = code & 15u; if (d != C_HOR1 && d != C_HOR2) code = to_left(3); }
        } else if (dir == C_VERT || dir == C_VERL) {
            const unsigned flag = (dir == C_VERT) ? C_NVER : C_NVEL;
            bool dead = false;
            while (!(code & flag)) { code = to_upper(0); if (!code) { dead = true; break; } }
            if (!dead) { const unsigned d = code & 15u; if (d != C_VER1 && d != C_VER2) code = to_upper(0); }
        } else if (dir == C_ACCZ) {
            do { code = to_left(1); } while (code && !(code & C_DONZ));
        } else if (dir == C_ACCM) {
            do { code = to_left(1); } while (code && !(code & C_DONM));
        } else if (dir == C_ACCP) {
            bool dead = false;
            do { code = to_left(1); if (!code) { dead = true; break; } } while (!(code & C_DONP));
            if (!dead) { code = to_upper(3); ++m; n += 3; }
        } else if (dir == C_HOR1) code = to_left(1);
        else if (dir == C_HOR2) code = to_left(2);
        else if (dir == C_VER1) code = to_upper(1);
        else if (dir == C_VER2) code = to_upper(2);
        else { status = -2; code = 0; break; }           // fatal("Unexpected dir") in the reference
    }
    if (status != -2) emit();
    if (writer) A.n_skl[pi] = status ? status : cnt;
    if (writer && status == -3) A.n_skl[pi] = -3;
}

extern "C" hipError_t spdh_launch_sweep(const HSweepArgs* a, int spj, int pen_cap, int local, hipStream_t stream)
{
    HSweepArgs A = *a;
    const dim3 grd((A.n_probs + 3) / 4), blk(256);
    const bool tab = pen_cap < SPDH_PEN_TAB;
    if (local) {
        if (!spj)     hipLaunchKernelGGL((spdh_sweep<false, true, true>), grd, blk, 0, stream, A);
        else if (tab) hipLaunchKernelGGL((spdh_sweep<true, true, true>), grd, blk, 0, stream, A);
        else          hipLaunchKernelGGL((spdh_sweep<true, false, true>), grd, blk, 0, stream, A);
    } else {
        if (!spj)     hipLaunchKernelGGL((spdh_sweep<false, true, false>), grd, blk, 0, stream, A);
        else if (tab) hipLaunchKernelGGL((spdh_sweep<true, true, false>), grd, blk, 0, stream, A);
        else          hipLaunchKernelGGL((spdh_sweep<true, false, false>), grd, blk, 0, stream, A);
    }
    return hipGetLastError();
}
extern "C" hipError_t spdh_launch_walk(const HWalkArgs* a, hipStream_t stream)
{
    HWalkArgs A = *a;
    hipLaunchKernelGGL(spdh_walk, dim3((A.n_probs + 3) / 4), dim3(256), 0, stream, A);
    return hipGetLastError();
}
