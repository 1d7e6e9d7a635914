// spdp_h_udh.hip -- linear-space engine of the aa x genome path: SimdAln2h1::hirschbergH1_wip
// (src/fwd2h1_wip_simd.h:338-773) for gfx950.  Same wave / DPP-row / LDS-ring structure and the same
// int16 saturating lanes as spdh_sweep (spdp_h_kernels.hip); instead of traceback codes every score
// carries a LINK -- the diagonal on which its path crossed the previous intermediate row -- through
// the same selects.  On the `n_im` intermediate rows the lane holding that row records, per
// diagonal, the vertical links (`vlnk`), the horizontal links the row's intron acceptors create
// (`hlnk`, exact in place) and a 3-bit event code from which spdh_udh_cpos replays the reference's
// `rlst[]` bookkeeping in its sequential stripe order (the rows of a pass run concurrently here, the
// reference's stripes do not) before walking the links back into the `cpos` rows lspH_ng consumes.
//
// Reference quirks reproduced: `hil` (intron length so far) advances only on steps where some lane
// of the stripe has a donor candidate in the pipe (`if (AllZero(ph_v)) continue;` skips the
// increment too, :656-668), and twice for phase +1 when the second pipe is busy; a score below
// `nevsel` is lifted to it by a non-matching acceptor blend (:575-580).  Not reproduced: local mode
// (`hb` lanes) -- rejected by the host; the reference leaves `sm_a` holding flag vectors on
// intermediate-row stripes (:586, 673, 681), which adds 0/1 to the diagonal term of lanes that have
// not reached the window yet; those cells lie left of the first codon and never carry a path.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

typedef short s16;
// int16 scores in the upper half of 32-bit registers: see spdp_h_kernels.hip
typedef int q16;
#define Q16(x) ((q16) ((unsigned) (x) << 16))
__device__ __forceinline__ q16 qadd(q16 a, q16 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ q16 qmax(q16 a, q16 b) { return a > b ? a : b; }
__device__ __forceinline__ s16 sadd(s16 a, s16 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s16 smax(s16 a, s16 b) { return a > b ? a : b; }

#define DPP_ROW_SR(n) (0x110 + (n))
#define DPP_ROW_RR(n) (0x120 + (n))
__device__ __forceinline__ int row_shr1(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, DPP_ROW_SR(1), 0xf, 0xf, false);
}
__device__ __forceinline__ int row_ror1(int src)
{
    return __builtin_amdgcn_mov_dpp(src, DPP_ROW_RR(1), 0xf, 0xf, true);
}
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ld_nt4(const int4* p)
{
    const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int ld_nt1(const int* p) { return __builtin_nontemporal_load(p); }

#define SPDH_PEN_TAB 2048
#define END_OF_ULK (INT32_MAX - 2)
template <bool B> struct BoolTag { static constexpr bool value = B; };

// ---------------------------------------------------------------------------
template <bool SPJ, bool TAB>
__global__ __launch_bounds__(256, 4) void spdh_sweep_udh(HUdhArgs A)
{
    __shared__ int   s_mtx[32 * 32];
    __shared__ int   s_pen[SPDH_PEN_TAB];          // q16
    __shared__ int   s_qlen[8], s_qpen[8];
    __shared__ int4  s_ring[4][4][64];
    __shared__ int4  s_feed[4][4][16];
    __shared__ int4  s_out[4][4][17];           // bottom-row results of a block's steps (spdp_sweep_fp.hip: why not a DPP chain)

    const DevScoringH* __restrict__ sc = A.sc;
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) s_mtx[i] = Q16(sc->mtx[i]);      // q16
    if (threadIdx.x < 8) { s_qlen[threadIdx.x] = sc->qm_len[threadIdx.x]; s_qpen[threadIdx.x] = sc->qm_pen[threadIdx.x]; }
    const int nquant = sc->nquant;
    const int pen_cap = (nquant > 1) ? min(sc->qm_len[nquant - 2] + 1, SPDH_PEN_TAB - 1) : 0;
    for (int h = threadIdx.x; h <= pen_cap; h += blockDim.x) {
        int pv = sc->qm_pen[0];
        for (int j = 1; j < nquant; ++j) if (h > sc->qm_len[j - 1]) pv = sc->qm_pen[j];
        s_pen[h] = Q16(pv);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int k = lane & 15;
    const int pi = __builtin_amdgcn_readfirstlane((int) blockIdx.x * 4 + wv);
    if (pi >= A.n_probs) return;

    const DevProblemH P = A.probs[pi];
    const int a_left = P.a_left, a_right = P.a_right, b_left = P.b_left, b_right = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const q16 ge = Q16(sc->gep), g1 = Q16(sc->g1), g2 = Q16(sc->g2), g3 = Q16(sc->g3);
    const int gop = sc->gop, gep = sc->gep;
    const int llmt = sc->llmt;
    int4* __restrict__ bnd = A.bnd + P.bnd_off;
    const int4* __restrict__ cols = A.cols + P.col_off;
    const short4* __restrict__ aux = A.aux + P.col_off;
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    int* __restrict__ imd0 = A.imd + P.imd_off;
    const int n_ent = P.buf_size + SPDH_BND_PAD;
    const int col_len = P.col_len;
    const int n_im = P.n_im;
#define BIDX(r) ((r) - lw + 3)
    auto gap_ext3 = [&](int i) { return i > sc->codonk1 ? sc->lgep : gep; };

    // ---- fhinitH1 with link lanes (mode 2 / 4: src/fwd2h1_simd.h:594-603 and the `c` pointers below)
    {
        const int rl = b_left - 3 * a_left;
        int rr = min(b_right - 3 * a_left, up);
        int rr_g = rr;
        if (!a_exgl && gep) rr_g = min(rr, (SPDH_NEV - sc->g3) / gep + rl + 4);
        const int re = a_exgl ? rl : up;                 // hc[r] = r below this diagonal
        for (int e = lane; e < n_ent; e += 64) {
            const int r = e + lw - 3;
            int h = SPDH_NEV, f = SPDH_NEV, c = 0, fc = 0;
            if (r >= lw && r < re) c = r;
            if (b_exgl == 1 && r >= lw && r < rl) h = 0;
            if (b_exgl == 2 && r == rl) { f = 0; fc = rl; }
            if (!a_exgl) {
                const int i = r - rl;
                if (b_exgl && i == 0) { f = 0; fc = c; }
                if (i == 0) h = 0;
                else if (i >= 1 && i <= 3) h = (i == 1) ? sc->g1 : (i == 2 ? sc->g2 : sc->g3);
                else if (i >= 4) {
                    const int gb = ((i - 1) % 3 == 0) ? sc->g1 : ((i - 1) % 3 == 1 ? sc->g2 : sc->g3);
                    if (gep) { if (r < rr_g) h = gb + ((i - 1) / 3) * gep; }
                    else if (r < rr) h = sc->g3;
                }
            }
            bnd[e] = make_int4(h, f, c, fc);
        }
        const int tot_imd = n_im * 5 * width;
        for (int e = lane; e < tot_imd; e += 64) imd0[e] = ((e / width) % 5 == 4) ? -1 : END_OF_ULK;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (a_exgl) {
            auto sigS0 = [&](int pos) { const int v = aux[pos].x; return v > 0 ? v : 0; };
            int hm3 = sigS0(b_left + 1), hm2 = sigS0(b_left + 2), hm1 = sigS0(b_left + 3);
            int cm3 = rl, cm2 = rl + 1, cm1 = rl + 2;
            if (lane < 3) {
                const int v = lane == 0 ? hm3 : (lane == 1 ? hm2 : hm1);
                int4 t = bnd[BIDX(rl + lane)];
                t.x = v; t.z = rl + lane;
                bnd[BIDX(rl + lane)] = t;
            }
            int l0 = rl, l1 = rl + 1, l2 = rl + 2;
            bool stopped = false;
            for (int r0 = rl + 3; r0 < rr && !stopped; r0 += 64) {
                const int rj = r0 + lane;
                const int bb = b_left + (rj - rl) + 1;
                int vS = 0, vE = 0;
                if (rj < rr) { vS = aux[bb].x; vE = aux[bb - 3].z; }
                int myh = 0, myc = 0; bool mine = false;
                const int cnt = min(64, rr - r0);
                for (int j = 0; j < cnt; ++j) {
                    const int r = r0 + j;
                    const int sS = __builtin_amdgcn_readlane(vS, j), sE = __builtin_amdgcn_readlane(vE, j);
                    const int gl = r - l0;
                    int h = hm3, c = cm3;
                    if (!(a_exgl & 1) && gl == 3) h += gop;
                    if (!(a_exgl & 2)) h += gap_ext3(gl);
                    h = (s16) (h + sE);
                    const bool stop = h < SPDH_NEV;
                    if (!stop) {
                        int x = (s16) (hm1 + sc->g1);
                        if (x > h) { h = x; c = cm1; }
                        x = (s16) (hm2 + sc->g2);
                        if (x > h) { h = x; c = cm2; }
                        x = sS > 0 ? sS : 0;
                        if (x > h) { h = x; l0 = r; c = r; }
                    }
                    if (lane == j) { myh = h; myc = c; mine = true; }
                    hm3 = hm2; hm2 = hm1; hm1 = h;
                    cm3 = cm2; cm2 = cm1; cm1 = c;
                    const int t = l0; l0 = l1; l1 = l2; l2 = t;
                    if (stop) { stopped = true; break; }
                }
                if (mine) {
                    int4 t = bnd[BIDX(rj)];
                    t.x = myh; t.z = myc;
                    bnd[BIDX(rj)] = t;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- the sweep
    const int n_stripes = (a_right - a_left + SPDH_NELEM - 1) / SPDH_NELEM;
    const int imd_step = (a_right - a_left + n_im) / (n_im + 1);
    int imd_cur = 0;
    int4* const ring = &s_ring[wv][g][0];
    int4* const feed = &s_feed[wv][g][0];
    auto pen_of = [&](int hil) -> q16 {
        if constexpr (TAB) return s_pen[min(hil, pen_cap)];
        int pv = s_qpen[0];
        for (int jq = 1; jq < nquant; ++jq) pv = (hil > s_qlen[jq - 1]) ? s_qpen[jq] : pv;
        return Q16(pv);
    };
    // "does any lane of my 16-lane row ..." from a wave ballot
    const int row_sh = 16 * (g & 1);
    auto row_any = [&](bool pred) -> bool {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(pred);
        const unsigned half = (g & 2) ? (unsigned) (b >> 32) : (unsigned) b;
        return ((half >> row_sh) & 0xffffu) != 0u;
    };

    for (int s0 = 0; s0 < n_stripes; s0 += 4) {
        const int s = s0 + g;
        const int ml = a_left + s * SPDH_NELEM;
        const bool has = s < n_stripes;
        const int j9 = has ? min(SPDH_NELEM, a_right - ml) : 0;
        const int j8 = j9 - 1;
        const int n_start = max(b_left, lw + 3 * ml);
        const int n9 = min(b_right, up + 3 * (ml + j9) + 1) + 3 * j9;      // exclusive here (:398)
        const int len = has ? max(0, n9 - n_start) : 0;
        const int nb = (len + 15) >> 4;
        const int nb0 = __builtin_amdgcn_readlane(nb, 0), nb1 = __builtin_amdgcn_readlane(nb, 16),
                  nb2 = __builtin_amdgcn_readlane(nb, 32), nb3 = __builtin_amdgcn_readlane(nb, 48);
        const int tot = max(max(nb0, SPDH_LAG + nb1), max(2 * SPDH_LAG + nb2, 3 * SPDH_LAG + nb3));
        const int mp1 = ml + 1;
        const int e_base = -3 * ml - lw + 3;
        const bool partial = j9 < SPDH_NELEM;
        // the reference walks its intermediates in order and tests, per stripe, only the current one
        // (:372-375, 698-703): replay that pointer over my 4 stripes
        int imd_i = -1, k8 = 0;
        bool pass_imd = false;
        for (int gg = 0; gg < 4; ++gg) {
            const int ml_gg = a_left + (s0 + gg) * SPDH_NELEM;
            if (imd_cur < n_im && ml_gg < a_right) {
                const int cand = a_left + (imd_cur + 1) * imd_step;
                const int mm = a_left + (cand - a_left - 1) / SPDH_NELEM * SPDH_NELEM;
                if (mm == ml_gg) {
                    if (gg == g) { imd_i = imd_cur; k8 = cand - mm - 1; }
                    ++imd_cur; pass_imd = true;
                }
            }
        }
        const bool imd_row = imd_i >= 0;
        int* const imd_p = imd0 + (int64_t) max(imd_i, 0) * 5 * width;
#define IIDX(r) ((r) - lw + 1)

        const int acode = (k < j9) ? acod[ml + k] : SPDH_ZCODE;
        const int* mrow = s_mtx + acode * 32;

        q16 h1 = Q16(SPDH_NEV), h2 = Q16(SPDH_NEV), h3 = Q16(SPDH_NEV);
        q16 f1 = Q16(SPDH_NEV), f2 = Q16(SPDH_NEV), f3 = Q16(SPDH_NEV);
        q16 e1 = Q16(SPDH_NEV), e2 = Q16(SPDH_NEV), e3 = Q16(SPDH_NEV);
        q16 u4 = Q16(SPDH_NEV), u5 = Q16(SPDH_NEV), u6 = Q16(SPDH_NEV);
        int c1 = 0, c2 = 0, c3 = 0, fc1 = 0, fc2 = 0, fc3 = 0, ec1 = 0, ec2 = 0, ec3 = 0;   // their links
        int uc4 = 0, uc5 = 0, uc6 = 0;
        q16 hiv0 = Q16(SPDH_NEV), hiv1 = Q16(SPDH_NEV), hiv2 = Q16(SPDH_NEV);
        int hil0 = 0, hil1 = 0, hil2 = 0, hic0 = 0, hic1 = 0, hic2 = 0;
        int dr0 = n_start - 3 * mp1, dr1 = dr0, dr2 = dr0;                  // donor_r[], lane k8 only
        int4* const outb = &s_out[wv][g][0];
        const bool is_bottom = k == ((j9 < SPDH_NELEM && j9 > 0) ? j8 : 15);     // (a partial last stripe: its last real row)

        int4 nx_b = make_int4(0, 0, 0, 0);
        int4 nx_c = make_int4(0, 0, 0, 0);
        // column record as lanes see it: window edges applied
        auto mask_col = [&](int4 rec, int c) -> int4 {
            if (c < 0 || c >= col_len) rec = make_int4(0, 0, 0, 0);
            if (c >= b_right) rec.x &= 0x00ffffff;                          // nothing splices at n >= b_right
            if (c < b_left + 3 || c > b_right + 2) rec.x = (rec.x & (int) 0xff00ffffu) | (SPDH_ZCODE << 16);
            return rec;
        };
        auto load_col = [&](int c) -> int4 { return mask_col(cols[min(max(c, 0), col_len - 1)], c); };
        auto prefetch = [&](int lbn) {
            const int nn = n_start + lbn * 16 + k;
            const int e = min(nn + e_base, n_ent - 1);
            nx_b = ld_nt4(bnd + e);
            // raw record (masks are applied when it is consumed: nothing may depend on the loaded value
            // here, or the compiler waits for the load on the spot and the prefetch is gone)
            nx_c = cols[min(max(nn, 0), col_len - 1)];
        };
        auto run_pass = [&](auto partial_tag, auto imd_tag) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial_tag)::value;
        constexpr bool IMD = decltype(imd_tag)::value;
        if (g == 0 && nb > 0) prefetch(0);
        for (int blk = 0; blk < tot; ++blk) {
            const int lb = blk - SPDH_LAG * g;
            if (lb == -1 && nb > 0) prefetch(0);
            if (lb >= 0 && lb < nb) {
                const int n0 = n_start + lb * 16;
                if (lb == 0) {
                    for (int i = 0; i < 3; ++i) {
                        const int c = n_start - 1 - k - 16 * i;
                        int4 rec = load_col(c);
                        rec.x &= 0x00ff0000; rec.y = 0; rec.z = 0;
                        ring[c & 63] = rec;
                    }
                    if (k == 0) {
                        const int e = n_start + e_base;
                        const int4 t1 = ld_nt4(bnd + e - 1), t2 = ld_nt4(bnd + e - 2), t3 = ld_nt4(bnd + e - 3);
                        u4 = Q16(t1.x); uc4 = t1.z;
                        u5 = Q16(t2.x); uc5 = t2.z;
                        u6 = Q16(t3.x); uc6 = t3.z;
                    }
                }
                feed[k] = make_int4(Q16(nx_b.x), Q16(nx_b.y), nx_b.z, nx_b.w);
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
                    // ---- insertion (E)
                    const q16 a1 = qadd(h1, g1), a2 = qadd(h2, g2);
                    bool m = a1 > a2;
                    q16 eh = qmax(a1, a2);
                    int ehc = m ? c1 : c2;
                    const q16 a3 = qadd(qadd(h3, g3), cv);
                    m = eh > a3;
                    eh = qmax(eh, a3);
                    ehc = m ? ehc : c3;
                    q16 ee = qadd(qadd(e3, ge), cv);
                    m = ee > eh;
                    ee = qmax(ee, eh);
                    const int eec = m ? ec3 : ehc;
                    // ---- deletion (F): extension first, then the three opens (:465-521)
                    const int4 fd = feed[J];
                    const q16 u3 = row_shr1(fd.x, h3);
                    const q16 uf = row_shr1(fd.y, f3);
                    const int uc3 = row_shr1(fd.z, c3);
                    const int ufc = row_shr1(fd.w, fc3);
                    q16 ff = qadd(uf, ge);
                    int ffc = ufc;
                    const q16 b3 = qadd(u3, g3);
                    m = ff > b3; ff = qmax(ff, b3); ffc = m ? ffc : uc3;
                    const q16 b2 = qadd(u4, g2);
                    m = ff > b2; ff = qmax(ff, b2); ffc = m ? ffc : uc4;
                    const q16 b1 = qadd(u5, g1);
                    m = ff > b1; ff = qmax(ff, b1); ffc = m ? ffc : uc5;
                    // ---- diagonal
                    const q16 sm = mrow[tron];
                    const q16 dg = qadd(qadd(sm, u6), cv);
                    m = ff > dg;
                    q16 h = m ? ff : dg;
                    int hc = m ? ffc : uc6;
                    int pv = m ? 2 : 0;
                    m = ee > h;
                    h = m ? ee : h;
                    hc = m ? eec : hc;
                    pv = m ? 1 : pv;
                    bool ab = false;
                    int acc_dr = 0; bool acc_hit = false;          // lane k8: donor_r of the last acceptor taken
                    if constexpr (SPJ) {
                        const unsigned c0 = fl & 3u;
                        const q16 s3_0 = Q16(rec.y), s3_1 = (q16) ((unsigned) rec.y & 0xffff0000u);
                        const q16 cv0 = hiv0, cv1 = hiv1, cv2 = hiv2;
                        const int cl0 = hil0, cl1 = hil1, cl2 = hil2;
                        const int cc0 = hic0, cc1 = hic1, cc2 = hic2;
                        const int cd0 = dr0, cd1 = dr1, cd2 = dr2;
                        // the three blends of each pipe in the reference's order; a pipe whose 16 lanes
                        // hold no candidate is skipped as a whole (:569)
                        const bool any0 = row_any(c0 != 0u), any1 = row_any((fl & 4u) != 0u);
                        const q16 x0 = qadd(qadd((c0 == 1u) ? cv0 : ((c0 == 2u) ? cv1 : cv2), s3_0),
                                            pen_of((c0 == 1u) ? cl0 : ((c0 == 2u) ? cl1 : cl2)));
                        const q16 x1 = qadd(qadd(cv2, s3_1), pen_of(cl2));
#pragma unroll
                        for (int f = 0; f < 3; ++f) {
                            const int lf = f == 0 ? cl0 : (f == 1 ? cl1 : cl2);
                            const int lc = f == 0 ? cc0 : (f == 1 ? cc1 : cc2);
                            const int ld = f == 0 ? cd0 : (f == 1 ? cd1 : cd2);
                            const q16 cand = (c0 == (unsigned) (f + 1) && lf > llmt) ? x0 : Q16(SPDH_NEV);
                            m = any0 && cand > h;
                            h = m ? cand : h; hc = m ? lc : hc; ab = ab || m;
                            if constexpr (IMD) { acc_dr = m ? ld : acc_dr; acc_hit = acc_hit || m; }
                        }
#pragma unroll
                        for (int f = 0; f < 3; ++f) {
                            const int lf = f == 0 ? cl0 : (f == 1 ? cl1 : cl2);
                            const int lc = f == 0 ? cc0 : (f == 1 ? cc1 : cc2);
                            const int ld = f == 0 ? cd0 : (f == 1 ? cd1 : cd2);
                            const q16 cand = (f == 2 && (fl & 4u) && lf > llmt) ? x1 : Q16(SPDH_NEV);
                            m = any1 && cand > h;
                            h = m ? cand : h; hc = m ? lc : hc; ab = ab || m;
                            if constexpr (IMD) { acc_dr = m ? ld : acc_dr; acc_hit = acc_hit || m; }
                        }
                    }
                    const int rj = n - 3 * mp1 - 6 * k8;            // my cell's diagonal when I hold the imd row
                    const bool on_imd = IMD && imd_row && k == k8 && rj >= lw && rj <= up && n < n9;
                    if constexpr (SPJ) {
                        // ---- intron 5' boundary; the length counters advance only with a busy pipe
                        const unsigned d0 = (fl >> 3) & 3u;
                        const q16 s5_0 = Q16(rec.z), s5_1 = (q16) ((unsigned) rec.z & 0xffff0000u);
                        const bool dny0 = row_any(d0 != 0u), dny1 = row_any((fl & 32u) != 0u);
                        const q16 pvH = ab ? Q16(SPDH_NEV) : qadd(h, s5_0);
                        const q16 pvD = ab ? Q16(SPDH_NEV) : qadd(u6, s5_0);
                        const q16 pvD1 = ab ? Q16(SPDH_NEV) : qadd(u6, s5_1);
                        m = dny0 && (d0 == 1u) && (pvH > hiv0);
                        hiv0 = m ? pvH : hiv0; hic0 = m ? hc : hic0; hil0 = m ? 0 : hil0; hil0 += dny0 ? 1 : 0;
                        if constexpr (IMD) dr0 = (m && on_imd) ? rj : dr0;
                        m = dny0 && (d0 == 2u) && (pvH > hiv1);
                        hiv1 = m ? pvH : hiv1; hic1 = m ? hc : hic1; hil1 = m ? 0 : hil1; hil1 += dny0 ? 1 : 0;
                        if constexpr (IMD) dr1 = (m && on_imd) ? rj : dr1;
                        m = dny0 && (d0 == 3u) && (pvD > hiv2);
                        hiv2 = m ? pvD : hiv2; hic2 = m ? hc : hic2; hil2 = m ? 0 : hil2; hil2 += dny0 ? 1 : 0;
                        if constexpr (IMD) dr2 = (m && on_imd) ? rj : dr2;
                        m = dny1 && (fl & 32u) && (pvD1 > hiv2);
                        hiv2 = m ? pvD1 : hiv2; hic2 = m ? hc : hic2; hil2 = m ? 0 : hil2; hil2 += dny1 ? 1 : 0;
                        if constexpr (IMD) dr2 = (m && on_imd) ? rj : dr2;
                    }
                    // ---- intermediate row (:672-683): links to the previous intermediate, event code
                    if constexpr (IMD) {
                        if (on_imd) {
                            int* hl0 = imd_p + IIDX(rj);
                            if (acc_hit) { hl0[0] = acc_dr; hl0[width] = acc_dr + width; }
                            hl0[4 * width] = pv | (ab ? 4 : 0);
                            hl0[2 * width] = hc;  hc = rj;
                            hl0[3 * width] = ffc; ffc = rj + width;
                        }
                    }
                    // ---- rotate the histories
                    h3 = h2; h2 = h1; h1 = h;     c3 = c2; c2 = c1; c1 = hc;
                    f3 = f2; f2 = f1; f1 = ff;    fc3 = fc2; fc2 = fc1; fc1 = ffc;
                    e3 = e2; e2 = e1; e1 = ee;    ec3 = ec2; ec2 = ec1; ec1 = eec;
                    u6 = u5; u5 = u4; u4 = u3;    uc6 = uc5; uc5 = uc4; uc4 = uc3;
                    // ---- bottom lane of the stripe -> slot J of the row's output block (read back at the flush)
                    if (is_bottom) outb[J] = make_int4(h, ff, hc, ffc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    const int j = 15 - k;
                    const int n = n0 + j;
                    const int r0 = n - 3 * mp1 - 6 * j8;
                    if (n < n9 && n - b_left >= 3 * j9 && r0 >= lw && r0 <= up && j9 > 0)
                        { const int4 o = outb[j]; bnd[BIDX(r0)] = make_int4(o.x >> 16, o.y >> 16, o.z, o.w); }
                }
            }
            asm volatile("" ::: "memory");
        }
        };
        const bool pass_partial = (s0 + 4 >= n_stripes) && ((a_right - a_left) & 15);
        if (pass_partial) {
            if (pass_imd) run_pass(BoolTag<true>{}, BoolTag<true>{});
            else          run_pass(BoolTag<true>{}, BoolTag<false>{});
        } else {
            if (pass_imd) run_pass(BoolTag<false>{}, BoolTag<true>{});
            else          run_pass(BoolTag<false>{}, BoolTag<false>{});
        }
#undef IIDX
    }

    // ---- fhlastH1 without a bitmap (mode 2 / 4): end cell and its link
    DevResultH R;
    R.score = SPDH_NEV; R.mr = a_right; R.nr = b_right; R.maxt = 0; R.maxr = 0;
    R.pad[0] = END_OF_ULK; R.pad[1] = R.pad[2] = 0;
    {
        const int m3 = 3 * a_right;
        const int rw = max(lw, b_left - m3);
        const int rr = b_right - m3;
        int maxr = rr, mx = rr;
        int mxval = (s16) ld_nt1(&bnd[BIDX(rr)].x);
        if (a_exgr) {
            int gl0 = 0, gl1 = 0, gl2 = 0;
            bool tc0 = false, tc1 = false, tc2 = false;
            int hq1 = 0, hq2 = 0, hq3 = 0;
            for (int i0 = 0; rw + i0 <= rr; i0 += 64) {
                const int hj = rw + i0 + lane;
                const int bb = hj + m3;
                int vH = 0, vE = 0, vT = 0;
                if (hj <= rr) {
                    vH = (s16) ld_nt1(&bnd[BIDX(hj)].x);
                    if (bb - 2 >= 0) { const short4 ax = aux[bb - 2]; vE = ax.z; vT = ax.y; }
                }
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
                    int kk = 0, best = cand0;
                    if (cand1 > best) { kk = 1; best = cand1; }
                    if (cand2 > best) { kk = 2; best = cand2; }
                    const int newh = (kk == 0) ? cand0 : (int) (s16) best;
                    if (kk == 0) { gl0 = 0; tc0 = false; }
                    if (hd == mx) mxval = newh;
                    else if (newh > mxval) { mx = hd; mxval = newh; maxr = hd - (kk == 2 ? 3 : 0); }
                    hq3 = hq2; hq2 = hq1; hq1 = newh;
                    { const int t = gl0; gl0 = gl1; gl1 = gl2; gl2 = t; }
                    { const bool t = tc0; tc0 = tc1; tc1 = tc2; tc2 = t; }
                }
            }
        } else {
            const int y = (s16) ((s16) ld_nt1(&bnd[BIDX(rr - 3)].x) + aux[b_right].y);
            if (y > mxval) { mxval = y; maxr = rr - 3; }
        }
        if (b_exgr) {
            const int rw2 = min(up - 1, b_right - 3 * a_left);
            int ga = SPDH_NEV, gb = SPDH_NEV, gc = SPDH_NEV;
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
        R.pad[0] = ld_nt1(&bnd[BIDX(maxr)].z);          // maxh.ulk
        if (maxr - rr > 0) R.mr = (b_right - maxr) / 3;
        else R.nr = mx + m3;
    }
    if (lane == 0) A.res[pi] = R;
#undef BIDX
}

// ---------------------------------------------------------------------------
// Tail of hirschbergH1_wip (:718-772): replay of the rlst[] bookkeeping over the recorded events,
// then the link back-walk into cpos rows and the written-back ranges.  One thread per problem.
__global__ void spdh_udh_cpos(HCposArgs A)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= A.n_probs) return;
    const DevProblemH P = A.probs[pi];
    const DevResultH R = A.res[pi];
    const int n_im = P.n_im, lw = P.lw, up = P.up, width = P.width;
    const int step = (P.a_right - P.a_left + n_im) / (n_im + 1);
    int* imd0 = A.imd + P.imd_off;
    int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
#define MI(i) (P.a_left + ((i) + 1) * step)
#define LNK(i, which, r) imd0[((int64_t) (i) * 5 + (which)) * width + ((r) - lw + 1)]
    for (int i = 0; i <= n_im; ++i)
        for (int c = 0; c < 10; ++c) CPOS(i, c) = END_OF_ULK;
    // rlst[p]: last diagonal of frame p at which a path entered an intermediate row diagonally (or
    // through an acceptor); a horizontal move on the row links back to it (:674-677).  The array
    // lives across stripes and intermediates in the reference, so it is replayed in that order.
    int rlst[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
    for (int i = 0; i < n_im; ++i)
        for (int r = lw; r <= up; ++r) {
            const int ev = LNK(i, 4, r);
            if (ev < 0) continue;
            const int p = ((r % 3) + 3) % 3;
            const int pv = ev & 3;
            const bool ab = ev & 4;
            if (ab) rlst[p] = r;
            if (pv == 0) rlst[p] = r;
            if (!ab && pv == 1) LNK(i, 0, r) = rlst[p];
        }
    int a_left = P.a_left, b_left = P.b_left;
    const int a_right = R.mr, b_right = R.nr;
    const int max_ml = P.a_left;                       // non-local
    int val = R.score;
    int i = n_im;
    while (--i >= 0 && MI(i) > a_right) ;
    if (i < 0 && MI(0) > a_right) CPOS(0, 2) = b_right;
    int r = R.pad[0];
    for ( ; i >= 0 && MI(i) > max_ml; --i) {
        int c = 0, d = 0;
        for ( ; r > up; r -= width) ++d;
        const int vl = (r >= lw - 1 && d < 2) ? LNK(i, 2 + d, r) : END_OF_ULK;
        if (vl < END_OF_ULK) {
            CPOS(i, c++) = MI(i);
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            const int mm3 = 3 * MI(i);
            for (int rp = LNK(i, d, r); lw <= rp && rp < up && r != rp; rp = LNK(i, d, r = rp)) {
                if (c < 8) CPOS(i, c++) = r + mm3; else ++c;
            }
            if (c < 9) { CPOS(i, c++) = r + mm3; CPOS(i, c) = END_OF_ULK; }
            r = LNK(i, 2 + d, r);                      // re-read: the horizontal walk moved r to the run's start
            if (r == END_OF_ULK) break;
        } else
            CPOS(i, 0) = END_OF_ULK;
    }
    while (r > up) r -= width;
    {
        const int rl = b_left - 3 * a_left;
        if (P.b_exgl && rl > r) {
            a_left = (b_left - r) / 3;
            for (int j = 0; j < n_im && MI(j) < a_left; ++j) CPOS(j, 0) = END_OF_ULK;
        }
        if (P.a_exgl && rl < r) b_left = 3 * a_left + r;
    }
    ++i;
    if ((i >= 0 && i < n_im && MI(i) < a_left) || CPOS(i, 2) < b_left) val = INT32_MIN / 16 * 7;
    A.scores[pi] = val;
    int* rg = A.ranges + 4 * pi;
    rg[0] = a_left; rg[1] = a_right; rg[2] = b_left; rg[3] = b_right;
#undef CPOS
#undef MI
#undef LNK
}

extern "C" hipError_t spdh_launch_udh(const HUdhArgs* a, int spj, int pen_cap, hipStream_t stream)
{
    HUdhArgs A = *a;
    const dim3 grd((A.n_probs + 3) / 4), blk(256);
    const bool tab = pen_cap < SPDH_PEN_TAB;
    if (!spj)     hipLaunchKernelGGL((spdh_sweep_udh<false, true>), grd, blk, 0, stream, A);
    else if (tab) hipLaunchKernelGGL((spdh_sweep_udh<true, true>), grd, blk, 0, stream, A);
    else          hipLaunchKernelGGL((spdh_sweep_udh<true, false>), grd, blk, 0, stream, A);
    return hipGetLastError();
}
extern "C" hipError_t spdh_launch_cpos(const HCposArgs* a, hipStream_t stream)
{
    HCposArgs A = *a;
    hipLaunchKernelGGL(spdh_udh_cpos, dim3((A.n_probs + 63) / 64), dim3(64), 0, stream, A);
    return hipGetLastError();
}
