// spdp_rowwave.hip -- the exact-intron-length ("-A0") cDNA engines as wavefront kernels.
//
// What they compute (ogotoh/spaln v3.0.7; restated for the checker in oracle/spdp_oracle_scalar.c):
//   MODE 0  Aln2s1::scorealoneS_ng + sinitS_ng / slastS_ng                 src/fwd2s1.cc:1163-1336, 1112-1161
//   MODE 1  Aln2s1::forwardS_ng + initS_ng / lastS_ng, Vmf::traceback and the record fix-up of
//           trcbkalignS_ng                                                 src/fwd2s1.cc:217-444, 142-215, 1690-1707
// int32 scores, affine gaps, a sorted list of the best donor candidates per query row, acceptors priced with the
// exact IntPen(length) + the junction table, "orphan exon" flags.  The reference walks the matrix row by row
// with its state in arrays indexed by DIAGONAL (H, F, and in mode 1 a Vmf pointer per state and a direction
// byte); results depend on what those arrays hold at the band edges, so the arrays are kept as they are.
//
// Mapping (ours).  One wave per 64-row tile of a problem (the four waves of a block share the read-only tables in
// LDS): the tiles of a problem run as a pipeline of waves, or -- <., false> -- one after the other in one wave.
// Lane k of the wave owns query row m0 + k of the tile and
// all per-row state (the horizontal gap, the orphan-exon flags, the candidate list) lives in its registers.  The
// wave sweeps anti-diagonals: at step S lane k is at column S - m, i.e. on array entry r = S - 2m; it reads
// entries r - 1 (its own left neighbour), r (the cell above-left) and r + 1 (the cell above) and writes r.  Every
// entry a lane reads was written one or two steps earlier by the lane above it -- the order the reference's
// row-by-row loop guarantees as well -- so the arrays can be shared through an LDS window of 256 diagonals that
// slides with the sweep: 64 entries at a time stream in from / out to the arrays in global memory (coalesced),
// which carry the state from one tile of rows to the next.  Vmf records are appended through a counter that is
// uniform in the wave (ballot + prefix count; pipelined tiles reserve their numbers 512 at a time from the
// problem's counter); record numbers differ from the reference's, the chains do not.  Lane 0 of the last tile's
// wave walks the chain back at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_dev.h"
#include "spdp_internal.h"
#include "spdp_sites.h"
#include "spdp_pipe.h"

namespace {

constexpr int NEV = INT32_MIN / 16 * 7;                 // NEVSEL, src/cmn.h:79
constexpr int RING = 256;                               // diagonals resident in LDS
constexpr int CHUNK = 32;                               // steps between two refills of the window
constexpr int WPB = 4;                                  // waves (= problems) per block: they share the tables below
using sites::NC; using sites::Tables; using sites::load_tables; using sites::DEADV;
// the linear-space kernel keeps the tables as round 2 laid them out (IntPen beyond 4096 as runs, spdp_ipen_runs.h): with them
// three four-wave blocks fill a CU's LDS to within one allocation granule; the span form of spdp_sites.h is 2 KB larger
constexpr int IPEN_LDS = SPDP_IPR_BASE;                 // IntPen(len) for len < this lives in LDS as it is, longer ones as runs

// read-only tables every wave of a block uses
struct TablesU {
    short mtx[32 * 32];
    short ipen[IPEN_LDS];
    short t53[256];
    IpenRuns runs;                                      // IntPen beyond the table above (spdp_ipen_runs.h)
    int gain[2];                                        // max IntPen, max junction-pair score: what an acceptor can add at most
};
__device__ __forceinline__ void load_tables_u(TablesU& T, const ScalarArgs& A, const DevScoring* sc)
{
    if (threadIdx.x < 2) T.gain[threadIdx.x] = INT32_MIN;
    __syncthreads();
    {
        int pm = INT32_MIN, tm = INT32_MIN;
        for (int i = threadIdx.x; i < A.intpen_len; i += blockDim.x) pm = max(pm, (int) A.intpen[i]);
        for (int i = threadIdx.x; i < 256; i += blockDim.x) tm = max(tm, (int) A.t53[i]);
        atomicMax(&T.gain[0], pm); atomicMax(&T.gain[1], tm);
    }
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) T.mtx[i] = (short) sc->mtx[i];
    for (int i = threadIdx.x; i < IPEN_LDS; i += blockDim.x) T.ipen[i] = A.intpen[min(i, A.intpen_len - 1)];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) T.t53[i] = A.t53[i];
    ipen_runs_load(T.runs, A.ipen_runs);
    __syncthreads();                                    // the only block-wide barrier: every wave reaches it
}
__device__ __forceinline__ int intpen_of_u(const TablesU& T, const ScalarArgs& A, int len)
{
    if (len < IPEN_LDS) return T.ipen[len];
    if (A.ipen_runs) return ipen_runs_get(T.runs, len, A.intpen_len);          // (kernel-uniform)
    return len >= A.intpen_len ? A.intpen[A.intpen_len - 1] : A.intpen[len];
}
// LDS traffic inside ONE wave needs no barrier (its DS instructions execute in order); the compiler must keep it so
#define WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// the three states a candidate can splice from / into: the cell's diagonal value, the horizontal and the vertical gap
enum { K_H = 0, K_E = 1, K_F = 2, K_E2 = 3, K_F2 = 4 };   // hf[] of the reference: DIAG, HORI, VERT, HORL, VERL (the last two with Noll = 3)
__device__ __forceinline__ int psp_bit(int k) { return k == 0 ? 4 : (k == 1 ? 1 : (k == 2 ? 8 : (k == 3 ? 2 : 16))); }    // src/aln.h:56

struct Lds2 {
    int hv[RING], fv[RING];
    int hp[RING], fp[RING], dr[RING];                   // forward only
};
struct Lds3 : Lds2 { int f2v[RING], f2p[RING]; };       // + the second vertical-gap state (Noll = 3)
template <bool DAGP> struct Lds_of { using type = Lds2; };
template <> struct Lds_of<true> { using type = Lds3; };
template <bool DAGP> using Lds = typename Lds_of<DAGP>::type;

}   // namespace

// PIPE: the tiles of one problem run as separate waves, each a few dozen anti-diagonals behind the tile above it
// (ScalarArgs::items lists (problem, tile) in dispatch order; a wave draws the next one from a ticket counter, so a
// tile's predecessor is always resident or done).  The arrays then cross CUs: their accesses go to the memory side
// (agent-scope atomics, the per-XCD L2s are not coherent with each other) and a tile publishes, after its stores have
// drained, the anti-diagonal up to which it has handed its entries back (prog[tile], +1; INT_MAX = finished).
template <bool X> __device__ __forceinline__ int gld(const int* p)
{
    if constexpr (X) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __builtin_nontemporal_load(p);
}
template <bool X> __device__ __forceinline__ void gst(int* p, int v)
{
    if constexpr (X) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
#define STORES_DRAINED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// ---- an experiment kept behind SPDP_COLFEED (default off: measured slower).  The column records.  A lane loaded the record of its own column (two steps ahead), but any wait for such a load is a
//     wait for EVERYTHING the wave has in flight (one in-order counter for loads and stores on gfx9), and the compiler, with
//     stores under branches in between, waits for zero outstanding: a step paid a full memory latency.  Now the records ride
//     down the lanes: lane k is on column c0 - k, the column lane k - 1 was on a step earlier, so each step lane 0 takes the
//     record of the column it enters -- ONE scalar load, issued a step ahead, out of the vector memory counter altogether --
//     and every other lane takes what the lane above it held (two DPP wave shifts).
#ifndef SPDP_COLFEED
#define SPDP_COLFEED 0
#endif
struct ColFeed {
    int x = 0, ya = 0;                                  // my column: cols[n].x, and cols[n].y & 0xff | aux[n] << 8
    unsigned long long rc = 0; unsigned ra = 0; int sh = 0;
    // the record of column c (wave-uniform) is asked for ...
    __device__ __forceinline__ void issue(const int2* cols, const uint8_t* aux, int c)
    {
        auto uni = [](uintptr_t a) -> uintptr_t {              // (wave-uniform by construction; said so that it sits in SGPRs)
            return (uintptr_t) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) a) |
                   (uintptr_t) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (a >> 32)) << 32;
        };
        const uintptr_t pc = uni(reinterpret_cast<uintptr_t>(cols + c));
        const uintptr_t pa = uni(reinterpret_cast<uintptr_t>(aux) + 2 * (intptr_t) c);
        const uintptr_t pq = pa & ~(uintptr_t) 3;
        sh = (int) (pa & 2) * 8;
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(rc) : "s"(pc));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(ra) : "s"(pq));
    }
    // ... has arrived (before anything looks at rc / ra: the compiler does not know these loads are asynchronous) ...
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rc), "+s"(ra)); }
    // ... and enters at lane 0 while every record moves one lane down
    __device__ __forceinline__ void shift()
    {
        const int nx = (int) (unsigned) rc;
        const int ny = ((int) (rc >> 32) & 0xff) | (int) (((ra >> sh) & 0xffffu) << 8);
        x = __builtin_amdgcn_update_dpp(nx, x, 0x138, 0xf, 0xf, false);          // wave_shr:1, lane 0 keeps `old` = the new record
        ya = __builtin_amdgcn_update_dpp(ny, ya, 0x138, 0xf, 0xf, false);
    }
};
// CUT: forwardS_ng with a cut range (`cutrng`, src/fwd2s1.cc:217, 423-430; shortcutS_ng :1899-1930): a row that reaches
// genomic column cut_l charges its horizontal gap for the cut_len columns behind it, leaves {that gap, nothing} in the
// diagonal arrays and goes on behind the cut -- while the arrays simply keep counting, so a lane's array entry follows
// its "virtual" column v (the one it would be on without the jump) and everything that names a genomic position
// (column records, intron lengths, path records) its real column n = v + cut_len once it is past the cut.  Rows that
// start behind cut_l never jump, as in the reference.  One wave per problem.
// DAGP: double affine gaps (PwdB::Noll = 3, -yl3; src/fwd2s1.cc:219, 297-305, 320-330): a second vertical and a second
// horizontal gap state priced with LongGOP / LongGEP, five states a candidate can leave from, a third array by diagonal.
template <int MODE, bool PIPE, bool CUT = false, bool DAGP = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void spdp_rowwave(ScalarArgs A)
{
    constexpr bool FWD = MODE == 1;
    constexpr int NODK = DAGP ? 5 : 3;                  // states (Nod)
    constexpr int NEWD = 8;                             // the file-local Newd of src/fwd2s1.cc:48 (a direction is a state number or this)
    static_assert(!CUT || (FWD && !PIPE), "the cut range exists for the forward engine only");
    __shared__ Lds<DAGP> Lw[WPB];
    __shared__ Tables T;
    const DevScoring* sc = A.sc;
    load_tables(T, A, sc);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (scalar: what follows from it -- the problem, its ranges, its arrays -- stays in SGPRs)
    Lds<DAGP>& L = Lw[wv];
    const int lane = threadIdx.x & 63;
    int pi = blockIdx.x * WPB + wv;
    int t_lo = 0, t_hi = INT32_MAX;                     // tiles of the problem this wave sweeps
    if (PIPE) {
        int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(A.pipe + A.pipe_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= A.n_items) return;
        const int2 it = A.items[tk];
        pi = __builtin_amdgcn_readfirstlane(it.x); t_lo = __builtin_amdgcn_readfirstlane(it.y); t_hi = t_lo + 1;
    }
    if (pi >= A.n_probs) return;
    const DevProblem P = wave_uniform(A.probs[pi]);
    const int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width;
    const int cut_l = CUT ? P.cut_l : 0, cut_len = CUT ? P.cut_len : 0;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, spj = sc->spj, llmt = sc->llmt, ipen = A.ipen;
    const int lgop = DAGP ? A.lgop : 0, lgep = DAGP ? A.lgep : 0, codonk1 = DAGP ? A.codonk1 : INT32_MAX;
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    const int2* __restrict__ cols = A.cols + P.col_off;            // {(sig5 + ipen) | sig3 << 16, base}
    const uint8_t* __restrict__ aux = A.aux + 2 * P.col_off;        // {bit0 donor | bit1 acceptor, dinc5 << 4 | dinc3}
    // the reference's arrays, by entry e = r - (lw - 1), 0 <= e < width
    int* __restrict__ gHv = A.work + P.bnd_off;
    int* __restrict__ gFv = gHv + width;
    int* __restrict__ gHp = gFv + width;
    int* __restrict__ gFp = gHp + width;
    int* __restrict__ gDr = gFp + width;
    int* __restrict__ gF2v = (FWD ? gDr : gFv) + width;              // Noll = 3: behind the arrays of the affine form
    int* __restrict__ gF2p = gF2v + width;
    int3* __restrict__ vrec = A.vmf + P.tb_off;
    int* __restrict__ vraw = reinterpret_cast<int*>(vrec);
    const int vcap = (int) P.imd_off;
    // PIPE: what the tiles of the problem share: {records appended, overflow}, prog[max_tiles], best[max_tiles][4]
    int* __restrict__ sy = PIPE ? A.pipe + (size_t) pi * A.pipe_stride : nullptr;
    int* __restrict__ prog = PIPE ? sy + 2 : nullptr;
    int* __restrict__ tbest = PIPE ? sy + 2 + A.max_tiles : nullptr;
    int vcount = 2;                                     // wave-uniform; records 0 (dummy) and 1 (start) below
    int vleft = 0;                                      // PIPE: numbers left of the chunk this wave holds
    bool vover = false;
    // appends one record for every lane that asks; returns its number (garbage for the others)
    auto vadd = [&](bool need, int mm, int nn, int pp) -> int {
        const unsigned long long mask = __ballot(need);
        if (!mask) return 0;
        const int cnt = __popcll(mask);
        if (PIPE && cnt > vleft) {                      // numbers come from the problem's counter SPDP_VMF_CHUNK at a time:
            int b = 0;                                  // they differ from the one-wave order, the chains do not
            if (lane == 0) b = __hip_atomic_fetch_add(sy, SPDP_VMF_CHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vcount = 2 + __builtin_amdgcn_readfirstlane(b);
            vleft = SPDP_VMF_CHUNK;
        }
        const int my = vcount + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0));
        vcount += cnt; vleft -= cnt;
        if (need) {
            if (my < vcap) {
                if (PIPE) { gst<true>(vraw + 3 * my, mm); gst<true>(vraw + 3 * my + 1, nn); gst<true>(vraw + 3 * my + 2, pp); }
                else vrec[my] = make_int3(mm, nn, pp);
            } else vover = true;
        }
        return my;
    };
    // PIPE: waits until tile `t` has published at least `req`; false when it never does (a wave of the problem is not
    // resident: cannot happen with the ticket order, kept as a bound on every spin)
    auto wait_for = [&](int t, int req) -> bool {
        long spins = 0;
        while (__hip_atomic_load(prog + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < req) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1l << 22)) {
                if (lane == 0) __hip_atomic_store(A.pipe + A.pipe_ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        return true;
    };
    auto publish = [&](int t, int v) {
        STORES_DRAINED();
        if (lane == 0) __hip_atomic_store(prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- the arrays as vset / initS_ng (sinitS_ng) leave them
    if (t_lo == 0) {
        const int r0 = bl - al;
        const int r_hi = a_exgl ? min(up, br - al) : r0;            // free start columns of the first row
        const int r_lo = max(lw, bl - ar);                          // first column, rows below the first
        for (int e = lane; e < width; e += 64) {
            const int r = e + lw - 1;
            int hv = NEV, hp = 0, fv = NEV, dr = 0;
            if (r == r0) { hv = 0; hp = 1; }
            else if (r > r0 && r <= r_hi) { hv = 0; dr = 1; }
            else if (r < r0 && r >= r_lo) {
                dr = 2;
                if (b_exgl) hv = 0;
                else {
                    // GapPenalty(1) + GapExtPen(2) + .. + GapExtPen(i), i = r0 - r (src/aln.h:275-282; beyond codonk1 with LongGEP)
                    const int i = r0 - r;
                    hv = gop + (DAGP ? min(i, codonk1) * gep + max(0, i - codonk1) * lgep : i * gep); hp = 1;
                    if (!FWD) fv = gop + i * gep;                   // (sinitS_ng: F runs on with BasicGEP)
                }
            }
            gst<PIPE>(gHv + e, hv); gst<PIPE>(gFv + e, fv);
            if (FWD) { gst<PIPE>(gHp + e, hp); gst<PIPE>(gFp + e, 0); gst<PIPE>(gDr + e, dr); }
            if (DAGP) { gst<PIPE>(gF2v + e, NEV); if (FWD) gst<PIPE>(gF2p + e, 0); }
        }
        if (FWD && lane == 0) {
            gst<PIPE>(vraw + 0, 0); gst<PIPE>(vraw + 1, 0); gst<PIPE>(vraw + 2, 0);
            gst<PIPE>(vraw + 3, al); gst<PIPE>(vraw + 4, bl); gst<PIPE>(vraw + 5, 0);
        }
    }

    // running maximum of a local right end: first maximum in row-major order
    int best_v = NEV, best_m = al, best_n = bl, best_p = 0;

    const int R0 = al + (a_exgl ? 1 : 0);                          // first row the reference's loop visits
    const int n_tiles = max(1, (ar - R0 + 64) / 64);               // (DevRun::prepare lists the same count)
    bool stalled = false;
    for (int ti = t_lo; ti < t_hi; ++ti) {
        const int m0 = R0 + 64 * ti;
        if (m0 > ar) break;
        // what the previous tile wrote back must be what this one reads
        if (!PIPE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        const int m = m0 + lane;
        const bool row = m <= ar;
        const int n_first = max(m - 1 + lw, bl) + 1, n_last = min(m + up, br);
        const bool any = row && n_first <= n_last;
        // CUT: does my row reach column cut_l, and the last virtual column it visits
        const bool jumps = CUT && n_first <= cut_l && cut_l <= n_last;
        const int v_last = jumps ? max(cut_l, n_last - cut_len) : n_last;
        auto real_col = [&](int v) { return (CUT && jumps && v > cut_l) ? v + cut_len : v; };
        // anti-diagonals this tile sweeps
        int s_lo = any ? n_first + m : INT32_MAX, s_hi = any ? v_last + m : INT32_MIN;
        for (int off = 32; off; off >>= 1) { s_lo = min(s_lo, __shfl_xor(s_lo, off)); s_hi = max(s_hi, __shfl_xor(s_hi, off)); }
        s_lo = __builtin_amdgcn_readfirstlane(s_lo); s_hi = __builtin_amdgcn_readfirstlane(s_hi);      // (the sweep's loop is scalar)
        if (s_lo > s_hi) continue;                                  // (PIPE: published as finished below)
        const int acode = (row && m >= 1) ? acod[m - 1] : 0;
        const short* qprof = T.mtx + acode * 32;
        const bool internal = FWD ? (spj && (!a_exgr || m < ar)) : true;
        const int sigB = (row && A.cip && P.cip_off >= 0) ? A.cip[P.cip_off + m] : 0;      // Cip_score::cip_score(m)

        // per-row state
        int e1v = NEV, e1p = 0;
        int e2v = NEV, e2p = 0;                                     // the second horizontal-gap state (Noll = 3)
        unsigned psp = 0;
        sites::Cands<FWD ? 2 : 0, !FWD> C;                          // the row's donor candidates (spdp_sites.h); riders: column, Vmf pointer
        C.clear();

        // window of entries resident in LDS: [res_lo, res_hi)
        auto need_lo = [&](int S) { return S - 2 * (m0 + 63) - 1 - (lw - 1); };
        auto need_hi = [&](int S) { return S - 2 * m0 + 1 - (lw - 1); };
        int res_lo = max(0, need_lo(s_lo)), res_hi = res_lo;
        auto refill = [&](int S) {
            // entries no lane will touch again go back to memory, those of the next CHUNK steps come in
            const int dead = min(max(0, need_lo(S)), width);
            for (int e = res_lo + lane; e < dead; e += 64) {
                const int q = e & (RING - 1);
                gst<PIPE>(gHv + e, L.hv[q]); gst<PIPE>(gFv + e, L.fv[q]);
                if (FWD) { gst<PIPE>(gHp + e, L.hp[q]); gst<PIPE>(gFp + e, L.fp[q]); gst<PIPE>(gDr + e, L.dr[q]); }
                if constexpr (DAGP) { gst<PIPE>(gF2v + e, L.f2v[q]); if (FWD) gst<PIPE>(gF2p + e, L.f2p[q]); }
            }
            res_lo = max(res_lo, dead);
            const int want = min(width, need_hi(S + CHUNK - 1) + 1);
            if (PIPE) {
                // the tile below may read everything under `dead`; what I am about to read, the tile above must have
                // handed back: its need_lo(S') >= want, S' published as S' + 1
                if (ti > 0 && !stalled) stalled = !wait_for(ti - 1, want + 2 * (m0 - 1) + 1 + (lw - 1) + 1);
                publish(ti, S + 1);
            }
            for (int e = max(res_hi, res_lo) + lane; e < want; e += 64) {
                const int q = e & (RING - 1);
                constexpr int NP = (FWD ? 5 : 2) + (DAGP ? (FWD ? 2 : 1) : 0);
                const int* src[NP]; int val[NP];
                src[0] = gHv; src[1] = gFv;
                if constexpr (FWD) { src[2] = gHp; src[3] = gFp; src[4] = gDr; }
                if constexpr (DAGP) { src[FWD ? 5 : 2] = gF2v; if constexpr (FWD) src[6] = gF2p; }
                gld_n<PIPE>(src, e, val);
                L.hv[q] = val[0]; L.fv[q] = val[1];
                if constexpr (FWD) { L.hp[q] = val[2]; L.fp[q] = val[3]; L.dr[q] = val[4]; }
                if constexpr (DAGP) { L.f2v[q] = val[FWD ? 5 : 2]; if constexpr (FWD) L.f2p[q] = val[6]; }
            }
            res_hi = max(res_hi, want);
            WAVE_SYNC();
        };
        // the column records of my next two cells are already on their way when a step starts
        auto ld_col = [&](int vv, int2& c, int& a2) {
            if (any && vv >= n_first && vv <= v_last) { const int nn = real_col(vv); c = cols[nn]; a2 = reinterpret_cast<const unsigned short*>(aux)[nn]; }
        };
        int2 col1 = make_int2(0, 0), col2 = make_int2(0, 0); int aux1 = 0, aux2 = 0;
        ColFeed feed;                                               // (without a cut range: the records ride down the lanes)
        constexpr bool OWN = CUT || !SPDP_COLFEED;                  // every lane loads its own column's record
        if constexpr (OWN) {
            ld_col(s_lo - m, col1, aux1);
            ld_col(s_lo + 1 - m, col2, aux2);
        } else { feed.issue(cols, aux, s_lo - m0); feed.arrived(); }

        for (int S = s_lo; S <= s_hi; ++S) {
            if (((S - s_lo) & (CHUNK - 1)) == 0) refill(S);
            const int v = S - m;                                    // the column without the jump: array entry r = v - m
            const int n = real_col(v);
            const bool on = any && v >= n_first && v <= v_last;
            int2 col; int ax, adn;
            if constexpr (OWN) {
                col = col1; ax = aux1 & 0xff; adn = aux1 >> 8;
                col1 = col2; aux1 = aux2;
                ld_col(v + 2, col2, aux2);
            } else {
                feed.shift();                                       // lane k now holds column (S - m0) - k = S - m
                feed.issue(cols, aux, S + 1 - m0);                  // lane 0's next column: at most br + 63, inside the padded records
                col = make_int2(feed.x, feed.ya & 0xff); ax = (feed.ya >> 8) & 0xff; adn = (feed.ya >> 16) & 0xff;
            }
            if (__ballot(on) == 0) { if constexpr (!OWN) feed.arrived(); continue; }
            const int r = v - m;
            const int q = (r - (lw - 1)) & (RING - 1), ql = (q - 1) & (RING - 1), qu = (q + 1) & (RING - 1);
            int hv = L.hv[q], hp = FWD ? L.hp[q] : 0, dir = FWD ? L.dr[q] : 0;     // entry r: the cell above-left
            const int uhv = L.hv[qu], uhp = FWD ? L.hp[qu] : 0;                    // entry r + 1: H of the cell above
            const int ufv = L.fv[qu], ufp = FWD ? L.fp[qu] : 0;                    //              F of the cell above
            const int lhv = L.hv[ql], lhp = FWD ? L.hp[ql] : 0;                    // entry r - 1: my left neighbour
            int fv = L.fv[q], fp = FWD ? L.fp[q] : 0;
            int f2v = NEV, f2p = 0, uf2v = NEV, uf2p = 0;
            if constexpr (DAGP) { f2v = L.f2v[q]; uf2v = L.f2v[qu]; if (FWD) { f2p = L.f2p[q]; uf2p = L.f2p[qu]; } }
            const int diag = hv;
            int mxk = K_H;                                          // which state holds the running maximum
            // the value / pointer of state k (five-way once the long-gap states exist)
            auto val_k = [&](int k) {
                if constexpr (DAGP) return k == K_H ? hv : (k == K_E ? e1v : (k == K_F ? fv : (k == K_E2 ? e2v : f2v)));
                else return k == K_H ? hv : (k == K_E ? e1v : fv);
            };
            auto ptr_k = [&](int k) {
                if constexpr (DAGP) return k == K_H ? hp : (k == K_E ? e1p : (k == K_F ? fp : (k == K_E2 ? e2p : f2p)));
                else return k == K_H ? hp : (k == K_E ? e1p : fp);
            };
            if (m != al) {                                          // (the row of a global left end has no cell above)
                hv += qprof[col.y];
                if (FWD) dir = (dir % NEWD) ? NEWD : 0;             // a diagonal step that follows a non-diagonal one: NEWD
                const int x = uhv + gop;
                if (FWD ? (x >= ufv) : (x > ufv)) { fv = x; fp = uhp; } else { fv = ufv; fp = ufp; }
                fv += gep;
                if (fv > hv) mxk = K_F;
                if constexpr (DAGP) {                               // Vertical2 (:297-305 / :1227-1231)
                    const int x2 = uhv + lgop;
                    if (FWD ? (x2 >= uf2v) : (x2 > uf2v)) { f2v = x2; f2p = uhp; } else { f2v = uf2v; f2p = uf2p; }
                    f2v += lgep;
                    if (f2v > val_k(mxk)) mxk = K_F2;
                }
            }
            if (on) {                                               // (row state: only cells of the row may touch it)
                const int x = lhv + gop;
                const unsigned prev_psp = psp;
                if (FWD ? (x >= e1v) : (x > e1v)) { e1v = x; e1p = lhp; psp = psp ? 1u : 0u; } else psp &= 1u;
                e1v += gep;
                const int cur = DAGP ? val_k(mxk) : (mxk == K_H ? hv : fv);
                if (FWD ? (e1v >= cur) : (e1v > cur)) mxk = K_E;
                if constexpr (DAGP) {                               // Horizontal2 (:320-330 / :1245-1254)
                    const int x2 = lhv + lgop;
                    if (FWD ? (x2 >= e2v) : (x2 > e2v)) { e2v = x2; e2p = lhp; if (prev_psp) psp |= 2u; } else psp |= (prev_psp & 2u);
                    e2v += lgep;
                    const int cur2 = val_k(mxk);
                    if (FWD ? (e2v >= cur2) : (e2v > cur2)) mxk = K_E2;
                }
            }
            __builtin_amdgcn_sched_barrier(0);                         // (the sections of a step stay apart: what one holds in registers the next need not)
            // ---- acceptor: every candidate of my row may raise the state it left from (src/fwd2s1.cc:330-372 / :1256-1283)
            const bool acc = on && internal && (ax & 2) && C.any();
            if (__ballot(acc)) {
                int x[NC];
                if (A.ipen_runs) sites::price<true>(C, T, A, acc, n, sigB + (col.x >> 16), adn & 15, llmt, x);
                else sites::price<false>(C, T, A, acc, n, sigB + (col.x >> 16), adn & 15, llmt, x);
                // who raised which state: the q word of the entry (FWD: the last one that reaches the value), -1 none
                int wh = -1, we = -1, wf = -1, we2 = -1, wf2 = -1;
#pragma unroll
                for (int l = 0; l < NC; ++l) {
                    const int cd = C.q[l] & 7;
#define SPDP_TRY(K, S, W) { const bool b_ = cd == K && (FWD ? x[l] >= S : x[l] > S); S = b_ ? x[l] : S; W = b_ ? C.q[l] : W; }
                    SPDP_TRY(K_H, hv, wh) SPDP_TRY(K_E, e1v, we)
                    if constexpr (DAGP) { SPDP_TRY(K_F, fv, wf) SPDP_TRY(K_E2, e2v, we2) SPDP_TRY(K_F2, f2v, wf2) }
                    else { const bool b_ = cd >= K_F && (FWD ? x[l] >= fv : x[l] > fv); fv = b_ ? x[l] : fv; wf = b_ ? C.q[l] : wf; }
#undef SPDP_TRY
                }
                // K_H, K_E, K_F in this order, as the reference's loop over its states; the two path records of an accepted
                // intron (donor side, acceptor side) are appended under one wave-uniform test per state
#define SPDP_TAKE(K, W, P)                                                                                      \
                if (__ballot(W >= 0)) {                                                                         \
                    const bool t = W >= 0;                                                                      \
                    if (t) psp |= psp_bit(K);                                                                   \
                    if (FWD) { const int p1 = vadd(t, m, C.rider(0, W), C.rider(1, W)); const int p2 = vadd(t, m, n, p1); if (t) P = p2; } \
                    if (t) { const int cur = val_k(mxk); if (FWD ? (val_k(K) >= cur) : (val_k(K) > cur)) mxk = K; }    \
                }
                SPDP_TAKE(K_H, wh, hp) SPDP_TAKE(K_E, we, e1p) SPDP_TAKE(K_F, wf, fp)
                if constexpr (DAGP) { SPDP_TAKE(K_E2, we2, e2p) SPDP_TAKE(K_F2, wf2, f2p) }
#undef SPDP_TAKE
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the cell's value: the best state
            const int hd = mxk;
            const int mxv = val_k(mxk);                                  // *mx
            const int mxp = ptr_k(mxk);
            int hval_raw = hv;                                      // the diagonal state's own value (candidate source K_H)
            int hptr_raw = hp;
            if (FWD) {
                bool newrec = false; int nr_p = 0;
                if (hd != K_H) { hv = mxv; hp = mxp; dir = hd; }
                else if (Local && hv > diag) {
                    if (LocalL && diag == 0) { newrec = true; nr_p = 0; }
                    else if (LocalR && on && hv > best_v) { best_v = hv; best_p = hp; best_m = m; best_n = n; }
                }
                const int p_local = vadd(on && newrec, m - 1, n - 1, nr_p);
                if (on && newrec) hp = p_local;
                // (vadd keeps a counter every lane must advance: it is never called under a per-lane condition)
                const bool reset = LocalL && hv <= 0;
                if (reset) { hv = 0; dir = 1; }
                const bool t = on && !reset && dir == NEWD && !(psp & psp_bit(K_H));
                const int pn = vadd(t, m - 1, n - 1, hp);
                if (t) hp = pn;
                hval_raw = hv; hptr_raw = hp;                       // (the reference's donor loop reads *h after these updates)
            } else {
                const int y = hv;
                if (hd != K_H) hv = mxv;
                else if (LocalR && on && y > best_v) best_v = y;
                if (LocalL && hv < 0) hv = 0;
                hval_raw = hv;
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- donor: the states of this cell enter my row's candidate list (src/fwd2s1.cc:386-421 / :1297-1329)
            const bool don = on && internal && (ax & 1);
            if (__ballot(don)) {
                const int sigJ = (int) (short) (col.x & 0xffff) - ipen;
                const int dn5 = adn >> 4;
                const int mx_now = hd == K_H ? hval_raw : val_k(hd);         // *mx as it stands now
                unsigned pend = 0;                                  // the states that ask for a place, lowest first
#pragma unroll
                for (int k = 0; k < NODK; ++k) {
                    const int sv = k == K_H ? hval_raw : val_k(k);
                    bool t = don && k >= (hd == K_H ? 0 : 1) && !(psp & psp_bit(k));
                    int z = mx_now;
                    if (hd == K_H || ((k - hd) % 2)) z += (k / 2 == 1) ? gop : (k / 2 == 2 ? lgop : 0);      // GOP[k / 2]
                    if (k != hd && sv <= z) t = false;              // cannot become the better path
                    pend |= t ? 1u << k : 0u;
                }
                while (__ballot(pend != 0)) {
                    const bool t = pend != 0;
                    const int k = __ffs(pend) - 1;
                    pend &= pend - 1;
                    const int sv = k <= K_H ? hval_raw : val_k(k);
                    const int sp = k <= K_H ? hptr_raw : ptr_k(k);
                    if constexpr (FWD) { const int rid[2] = {n, sp}; C.insert(t, sv + sigJ, n, k, dn5, rid); }
                    else { const int rid[1] = {0}; C.insert(t, sv + sigJ, n, k, dn5, rid); }
                }
            }
            if (CUT && on && jumps && v == cut_l) {                  // the gap runs on over the cut: {gap, nothing} stay behind
                e1v += gep * cut_len;
                if (DAGP) { e2v += lgep * cut_len; hv = e2v; hp = e2p; }            // (*h = dagp ? e2 : e1; F2 stays)
                else { hv = e1v; hp = e1p; }
                fv = NEV; fp = 0;
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- entry r takes the cell
            if (on) {
                L.hv[q] = hv; L.fv[q] = fv;
                if (FWD) { L.hp[q] = hp; L.fp[q] = fp; L.dr[q] = dir; }
                if constexpr (DAGP) { L.f2v[q] = f2v; if (FWD) L.f2p[q] = f2p; }
            }
            if constexpr (!OWN) feed.arrived();
        }
        // everything still resident goes back
        {
            WAVE_SYNC();
            for (int e = res_lo + lane; e < res_hi; e += 64) {
                const int q = e & (RING - 1);
                gst<PIPE>(gHv + e, L.hv[q]); gst<PIPE>(gFv + e, L.fv[q]);
                if (FWD) { gst<PIPE>(gHp + e, L.hp[q]); gst<PIPE>(gFp + e, L.fp[q]); gst<PIPE>(gDr + e, L.dr[q]); }
                if constexpr (DAGP) { gst<PIPE>(gF2v + e, L.f2v[q]); if (FWD) gst<PIPE>(gF2p + e, L.f2p[q]); }
            }
        }
    }
    if (PIPE) {
        // finished = every entry holds what the tiles up to this one leave: the tile above must be finished too
        const int ti = t_lo;
        if (LocalR) {
            // first maximum in row-major order over the tile's lanes, handed to the wave of the last tile
            for (int off = 32; off; off >>= 1) {
                const int ov = __shfl_xor(best_v, off), om = __shfl_xor(best_m, off), on_ = __shfl_xor(best_n, off), op = __shfl_xor(best_p, off);
                if (ov > best_v || (ov == best_v && (om < best_m || (om == best_m && on_ < best_n)))) { best_v = ov; best_m = om; best_n = on_; best_p = op; }
            }
            if (lane == 0) { gst<true>(tbest + 4 * ti, best_v); gst<true>(tbest + 4 * ti + 1, best_m);
                             gst<true>(tbest + 4 * ti + 2, best_n); gst<true>(tbest + 4 * ti + 3, best_p); }
        }
        if (__any(vover) && lane == 0) gst<true>(sy + 1, 1);
        if (ti > 0 && !stalled) stalled = !wait_for(ti - 1, INT32_MAX);
        publish(ti, INT32_MAX);
        if (ti != n_tiles - 1) return;
        // the wave of the last tile ends the problem
        vover = __hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (LocalR) {
            best_v = NEV; best_m = al; best_n = bl; best_p = 0;
            for (int t = lane; t < n_tiles; t += 64) {              // tiles in row order: the first maximum wins
                const int ov = gld<true>(tbest + 4 * t), om = gld<true>(tbest + 4 * t + 1);
                const int on_ = gld<true>(tbest + 4 * t + 2), op = gld<true>(tbest + 4 * t + 3);
                if (ov > best_v) { best_v = ov; best_m = om; best_n = on_; best_p = op; }
            }
        }
    } else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");

    DevResult R;
    R.score = NEV; R.mr = ar; R.nr = br; R.ml = al; R.ulk = 0; R.maxr = 0; R.pad[0] = R.pad[1] = 0;
    auto GH = [&](int r) { return gld<PIPE>(gHv + (r - (lw - 1))); };
    // first maximum of H over [lo, hi] walked upwards (dir = +1) or downwards (dir = -1), only where it beats `start`
    auto scan_best = [&](int lo, int hi, int step, int start_r, int start_v) {
        // candidates strictly greater than the running maximum take over in walk order: the result is the first position
        // (in walk order) that holds the range's maximum, provided that maximum beats start_v
        int bv = start_v, bk = INT32_MAX;                           // bk: position in walk order
        const int cnt = hi - lo + 1;
        for (int i = lane; i < cnt; i += 64) {
            const int r = step > 0 ? lo + i : hi - i;
            const int v = GH(r);
            if (v > bv) { bv = v; bk = i; }
        }
        for (int off = 32; off; off >>= 1) {
            const int ov = __shfl_xor(bv, off), ok = __shfl_xor(bk, off);
            if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        if (bk == INT32_MAX) return start_r;
        return step > 0 ? lo + bk : hi - bk;
    };
    if (!FWD) {
        if (LocalR) {
            for (int off = 32; off; off >>= 1) best_v = max(best_v, __shfl_xor(best_v, off));
            R.score = best_v;
        } else {                                                    // slastS_ng: the best end value
            const int r9 = br - ar;
            int mx = GH(r9);
            if (b_exgr) { const int rw = min(up, br - al); if (rw > r9) mx = max(mx, GH(scan_best(r9 + 1, rw, -1, r9, mx))); }
            if (a_exgr) { const int rw = max(lw, bl - ar); if (rw < r9) mx = max(mx, GH(scan_best(rw, r9 - 1, +1, r9, mx))); }
            R.score = mx;
        }
        if (lane == 0) A.res[pi] = R;
        return;
    }
    int ptr = 0;
    if (LocalR) {
        // first maximum in row-major order over all lanes
        for (int off = 32; off; off >>= 1) {
            const int ov = __shfl_xor(best_v, off), om = __shfl_xor(best_m, off), on_ = __shfl_xor(best_n, off), op = __shfl_xor(best_p, off);
            if (ov > best_v || (ov == best_v && (om < best_m || (om == best_m && on_ < best_n)))) { best_v = ov; best_m = om; best_n = on_; best_p = op; }
        }
        ptr = vadd(lane == 0, best_m, best_n, best_p);
        ptr = __shfl(ptr, 0);
        R.score = best_v;
    } else {                                                        // lastS_ng (entries behind the cut sit cut_len lower)
        const int r9 = br - ar - cut_len;
        int mx = r9;
        if (a_exgr) { const int rw = max(lw, (CUT ? cut_l : bl) - ar) - cut_len; if (rw <= r9) mx = scan_best(rw, r9, +1, mx, GH(mx)); }
        if (b_exgr) { const int rw = min(up, br - al) - cut_len; if (rw > r9) mx = scan_best(r9 + 1, rw, -1, mx, GH(mx)); }
        const int i = mx - r9;
        int m9 = ar, n9 = br;
        if (i > 0) m9 -= i;
        if (i < 0) n9 += i;
        const int e = mx - (lw - 1);
        ptr = vadd(lane == 0, m9, n9, gld<PIPE>(gHp + e));
        ptr = __shfl(ptr, 0);
        R.score = GH(mx);
    }
    int vtotal = vcount;                                            // numbers handed out
    if (PIPE) { STORES_DRAINED(); vtotal = 2 + __hip_atomic_load(sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    const bool over_any = __any(vover);
    // Vmf::traceback(ptr) + the boundary record of trcbkalignS_ng, by one lane
    if (lane == 0) {
        int2* out = A.skl + (int64_t) pi * A.skl_cap;
        int cnt = 0, status = over_any ? -3 : 0;
        auto rec = [&](int i) {
            if (PIPE) return make_int3(gld<true>(vraw + 3 * i), gld<true>(vraw + 3 * i + 1), gld<true>(vraw + 3 * i + 2));
            return vrec[i];
        };
        if (ptr > 0 && ptr < vtotal && !status) {
            int3 sv = rec(ptr);
            int lm = 0, ln = 0;
            for (;;) {
                if (cnt < A.skl_cap) out[cnt] = make_int2(sv.x, sv.y); else status = -1;
                lm = sv.x; ln = sv.y; ++cnt;
                if (!sv.z) break;
                if (sv.z < 0 || sv.z >= vtotal || cnt > vtotal) { status = -2; break; }     // not a chain: never follow it
                sv = rec(sv.z);
            }
            const int rd = Local ? 0 : ((ln - lm) - bl + al);
            if (rd) {
                const int2 rec = rd > 0 ? make_int2(al, bl + rd) : make_int2(al - rd, bl);
                if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
                ++cnt;
            }
        }
        A.n_skl[pi] = status ? status : cnt;
        A.res[pi] = R;
    }
}

// ---------------------------------------------------------------------------------------------------------
// MODE 2: Aln2s1::hirschbergS_ng + hinitS_ng / hlastS_ng with UdhIntermediate(lub = true)
//         src/fwd2s1.cc:762-1104, 701-760; src/udh_intermediate.h:29-66
// The -A0 linear-space engine: the recurrence above with every state carrying the range of diagonals it has
// visited since the last intermediate row (upr, lwr), the row its path started on (ml) and a link (ulk) to where
// it crossed the previous intermediate row.  Same mapping: lane = row, entries {val, upr, lwr, ml, ulk} of H and F
// by diagonal in the sliding LDS window.  The lane that holds an intermediate row writes that row's link / bound
// arrays (global memory) and keeps `rlst`, the one value the reference carries from one intermediate row to the
// next: a tile is never taller than the distance between intermediates, so it holds at most one and `rlst` goes
// from tile to tile by broadcast.  Lane 0 walks the links back into the cpos rows lspS_ng reads.
namespace {
constexpr int EOU = 0x7fffffff - 2;                     // end_of_ulk, src/aln.h:49
constexpr int INH = 0x7ffffff0;                         // PIPE: "the rlst the row above me ends with", resolved by the link walk
// (the affine form must not grow by a byte: three blocks of it fill the CU's LDS to within one allocation granule)
struct LdsU2 {
    int hv[RING], hu[RING], hl[RING], hm[RING], hk[RING];
    int fv[RING], fu[RING], fl[RING], fm[RING], fk[RING];
};
struct LdsU3 : LdsU2 { int gv[RING], gu[RING], gl[RING], gm[RING], gk[RING]; };     // + the second vertical-gap state (Noll = 3)
template <bool DAGP> struct LdsU_of { using type = LdsU2; };
template <> struct LdsU_of<true> { using type = LdsU3; };
template <bool DAGP> using LdsU = typename LdsU_of<DAGP>::type;
struct St { int v, u, l, m, k; };                       // value, upr, lwr, ml, ulk
// one of three state records, FIELD BY FIELD: `c ? a : b` on the records themselves is an lvalue -- a pointer is selected and
// the record copied from memory, which parks H, E and F in scratch memory for the whole sweep (round 4)
__device__ __forceinline__ St st_sel3(int k, const St& h, const St& e, const St& f)
{
    St r;
    r.v = k == K_H ? h.v : (k == K_E ? e.v : f.v); r.u = k == K_H ? h.u : (k == K_E ? e.u : f.u);
    r.l = k == K_H ? h.l : (k == K_E ? e.l : f.l); r.m = k == K_H ? h.m : (k == K_E ? e.m : f.m);
    r.k = k == K_H ? h.k : (k == K_E ? e.k : f.k);
    return r;
}
// ... of five (Noll = 3: H, E1, F1, E2, F2)
__device__ __forceinline__ St st_sel5(int k, const St& h, const St& e, const St& f, const St& e2, const St& f2)
{
    St r;
#define SEL5(x) (k == K_H ? h.x : (k == K_E ? e.x : (k == K_F ? f.x : (k == K_E2 ? e2.x : f2.x))))
    r.v = SEL5(v); r.u = SEL5(u); r.l = SEL5(l); r.m = SEL5(m); r.k = SEL5(k);
#undef SEL5
    return r;
}
}   // namespace

// PIPE as above.  `rlst` is the one value that would tie a tile to the END of the intermediate row above it; it is only
// ever stored (into HLNK), so a tile starts from the marker INH, every intermediate row leaves the value it ends
// with in rlf[], and the link walk replaces the marker by what the rows above left.
// DAGP: double affine gaps (Noll = 3): the states E2 / F2 (HORL / VERL), a third plane of entries by diagonal and a third
// link plane per intermediate row (src/fwd2s1.cc:764, 847-880, 917-927, 1016-1023)
template <bool DAGP> constexpr int WPBU = 4;
template <bool PIPE, bool DAGP = false>
__global__ __launch_bounds__(64 * WPBU<DAGP>) __attribute__((amdgpu_waves_per_eu(DAGP ? 2 : 3, DAGP ? 2 : 3))) void spdp_rowwave_udh(ScalarArgs A)
{
    constexpr int WPB = WPBU<DAGP>;
    constexpr int NOL = DAGP ? 3 : 2, NODK = 2 * NOL - 1, NA = 5 * NOL;
    __shared__ LdsU<DAGP> Lw[WPB];
    __shared__ TablesU T;
    const DevScoring* sc = A.sc;
    load_tables_u(T, A, sc);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LdsU<DAGP>& L = Lw[wv];
    const int lane = threadIdx.x & 63;
    int pi = blockIdx.x * WPB + wv;
    int t_lo = 0, t_hi = INT32_MAX;
    if (PIPE) {
        int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(A.pipe + A.pipe_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= A.n_items) return;
        const int2 it = A.items[tk];
        pi = __builtin_amdgcn_readfirstlane(it.x); t_lo = __builtin_amdgcn_readfirstlane(it.y); t_hi = t_lo + 1;
    }
    if (pi >= A.n_probs) return;
    const DevProblem P = wave_uniform(A.probs[pi]);
    int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, n_im = P.n_im, intvl = P.imd_intvl;
    const bool a_exgl = P.flags & 1, a_exgr = P.flags & 2, b_exgl = P.flags & 4, b_exgr = P.flags & 8;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, llmt = sc->llmt, ipen = A.ipen;
    const int lgop = A.lgop, lgep = A.lgep, codonk1 = A.codonk1;
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    const int2* __restrict__ cols = A.cols + P.col_off;
    const uint8_t* __restrict__ aux = A.aux + 2 * P.col_off;
    int* const g0 = A.work + P.bnd_off;                             // 5 * Noll arrays of `width` entries
    auto G = [&](int arr) { return g0 + (int64_t) arr * width; };  // 0..4 H {v,u,l,m,k}, 5..9 F, 10..14 F2
    // intermediate i: hlnk[Noll], vlnk[Noll], lwrb[Noll], uprb[Noll], `width` ints each, entry r - lw + 1
    int* const imd_base = A.imd + P.imd_off;
    const int64_t us = NOL * (int64_t) width;
    enum { HLNK = 0, VLNK = 1, LWRB = 2, UPRB = 3 };
    auto IM = [&](int i, int arr, int k, int r) -> int* { return imd_base + (int64_t) i * 4 * us + arr * us + (int64_t) k * width + (r - lw + 1); };
    auto mi_of = [&](int i) { return P.a_left + (i + 1) * intvl; };
    int* cpos = A.cpos + (int64_t) pi * A.cpos_stride;
#define CPOS(i, c) cpos[(i) * 10 + (c)]
    // PIPE: what the tiles of the problem share: prog[max_tiles], best[max_tiles][8], rlf[n_im]
    int* __restrict__ sy = PIPE ? A.pipe + (size_t) pi * A.pipe_stride : nullptr;
    int* __restrict__ prog = PIPE ? sy + 2 : nullptr;
    int* __restrict__ tbest = PIPE ? sy + 2 + A.max_tiles : nullptr;
    int* __restrict__ rlf = PIPE ? sy + 2 + 9 * A.max_tiles : nullptr;
    auto wait_for = [&](int t, int req) -> bool {
        long spins = 0;
        while (__hip_atomic_load(prog + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < req) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1l << 22)) {
                if (lane == 0) __hip_atomic_store(A.pipe + A.pipe_ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        return true;
    };
    auto publish = [&](int t, int v) {
        STORES_DRAINED();
        if (lane == 0) __hip_atomic_store(prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto imd_init = [&](int i) {
        int* b = imd_base + (int64_t) i * 4 * us;
        for (int64_t q = lane; q < us; q += 64) {
            gst<PIPE>(b + q, EOU); gst<PIPE>(b + us + q, EOU); gst<PIPE>(b + 2 * us + q, 0x7fffffff); gst<PIPE>(b + 3 * us + q, (int) 0x80000000);
        }
    };

    // ---- the arrays as hinitS_ng leaves them; the link / bound arrays of the intermediates
    if (t_lo == 0) {
        const int r0 = bl - al, rb = bl - ar;
        const int r_hi = a_exgl ? min(up, br - al) : r0;
        const int r_lo = max(lw, bl - ar);
        for (int e = lane; e < width; e += 64) {
            const int r = e + lw - 1;
            St h = {NEV, rb, rb, 0, EOU};
            if (r >= r0 && r <= r_hi) h = {0, r, r, al, r};
            else if (r < r0 && r >= r_lo) {
                if (b_exgl) h = {0, r, r, al + (r0 - r), r};
                else if (DAGP) h = {gop + min(r0 - r, codonk1) * gep + max(0, r0 - r - codonk1) * lgep, r0, r, al + (r0 - r), r0};   // GapPenalty(1) + GapExtPen(2 ..)
                else h = {gop + (r0 - r) * gep, r0, r, al + (r0 - r), r0};
            }
            gst<PIPE>(G(0) + e, h.v); gst<PIPE>(G(1) + e, h.u); gst<PIPE>(G(2) + e, h.l); gst<PIPE>(G(3) + e, h.m); gst<PIPE>(G(4) + e, h.k);
            gst<PIPE>(G(5) + e, NEV); gst<PIPE>(G(6) + e, rb); gst<PIPE>(G(7) + e, rb); gst<PIPE>(G(8) + e, 0); gst<PIPE>(G(9) + e, EOU);
            if (DAGP) { gst<PIPE>(G(10) + e, NEV); gst<PIPE>(G(11) + e, rb); gst<PIPE>(G(12) + e, rb); gst<PIPE>(G(13) + e, 0); gst<PIPE>(G(14) + e, EOU); }
        }
        if (!PIPE) for (int i = 0; i < n_im; ++i) imd_init(i);      // (PIPE: by the tile that holds the row)
    }

    // local right end: first maximum in row-major order
    St best = {NEV, 0, 0, al, 0}; int best_mr = ar, best_nr = br;
    int rlst = 0x7fffffff;
    const int R0 = al + (a_exgl ? 1 : 0);
    const int TH = max(1, min(64, intvl));                          // tile height: at most one intermediate row per tile
    const int n_tiles = max(1, (ar - R0 + TH) / TH);               // (DevRun::prepare lists the same count)
    bool stalled = false;
    for (int ti = t_lo; ti < t_hi; ++ti) {
        const int m0 = R0 + TH * ti;
        if (m0 > ar) break;
        if (!PIPE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        const int m = m0 + lane;
        const bool row = lane < TH && m <= ar;
        const int n_first = max(m - 1 + lw, bl) + 1, n_last = min(m + up, br);
        const bool any = row && n_first <= n_last;
        int s_lo = any ? n_first + m : INT32_MAX, s_hi = any ? n_last + m : INT32_MIN;
        for (int off = 32; off; off >>= 1) { s_lo = min(s_lo, __shfl_xor(s_lo, off)); s_hi = max(s_hi, __shfl_xor(s_hi, off)); }
        s_lo = __builtin_amdgcn_readfirstlane(s_lo); s_hi = __builtin_amdgcn_readfirstlane(s_hi);
        // is my row an intermediate row, and which
        const int iq = (m - P.a_left) / max(1, intvl) - 1;
        const bool is_imd = row && intvl > 0 && (m - P.a_left) % intvl == 0 && iq >= 0 && iq < n_im;
        const unsigned long long imd_mask = __ballot(is_imd);
        if (PIPE && imd_mask) {
            const int i0 = __shfl(iq, __ffsll((long long) imd_mask) - 1);
            imd_init(i0);
            rlst = i0 == 0 ? 0x7fffffff : INH;
        }
        if (s_lo <= s_hi) {
            const int acode = (row && m >= 1) ? acod[m - 1] : 0;
            const short* qprof = T.mtx + acode * 32;
            const int sigB = (row && A.cip && P.cip_off >= 0) ? A.cip[P.cip_off + m] : 0;  // Cip_score::cip_score(m)
            St E = {NEV, bl - ar, bl - ar, 0, EOU};
            St E2 = E;
            unsigned psp = 0;
            int cv[NC], cj[NC], cd[NC], cu[NC], cl[NC], cm[NC], ck[NC], cx[NC];
#pragma unroll
            for (int l = 0; l < NC; ++l) { cv[l] = NEV; cj[l] = 0; cd[l] = 0; cu[l] = (int) 0x80000000; cl[l] = 0x7fffffff; cm[l] = 0; ck[l] = EOU; cx[l] = 0; }
            int ncand = -1;
            auto need_lo = [&](int S) { return S - 2 * (m0 + 63) - 1 - (lw - 1); };
            auto need_hi = [&](int S) { return S - 2 * m0 + 1 - (lw - 1); };
            int res_lo = max(0, need_lo(s_lo)), res_hi = res_lo;
            int* lds[NA] = {L.hv, L.hu, L.hl, L.hm, L.hk, L.fv, L.fu, L.fl, L.fm, L.fk};
            if constexpr (DAGP) { lds[10] = L.gv; lds[11] = L.gu; lds[12] = L.gl; lds[13] = L.gm; lds[14] = L.gk; }
            auto refill = [&](int S) {
                const int dead = min(max(0, need_lo(S)), width);
                for (int e = res_lo + lane; e < dead; e += 64) {
                    const int q = e & (RING - 1);
#pragma unroll
                    for (int a = 0; a < NA; ++a) gst<PIPE>(G(a) + e, lds[a][q]);
                }
                res_lo = max(res_lo, dead);
                const int want = min(width, need_hi(S + CHUNK - 1) + 1);
                if (PIPE) {
                    if (ti > 0 && !stalled) stalled = !wait_for(ti - 1, want + 2 * (m0 - TH + 63) + 1 + (lw - 1) + 1);
                    publish(ti, S + 1);
                }
                for (int e = max(res_hi, res_lo) + lane; e < want; e += 64) {
                    const int q = e & (RING - 1);
                    const int* src[NA]; int val[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) src[a] = G(a);
                    gld_n<PIPE>(src, e, val);
#pragma unroll
                    for (int a = 0; a < NA; ++a) lds[a][q] = val[a];
                }
                res_hi = max(res_hi, want);
                WAVE_SYNC();
            };
            auto ld_col = [&](int nn, int2& c, int& a2) {
                if (any && nn >= n_first && nn <= n_last) { c = cols[nn]; a2 = reinterpret_cast<const unsigned short*>(aux)[nn]; }
            };
            int2 col1 = make_int2(0, 0), col2 = make_int2(0, 0); int aux1 = 0, aux2 = 0;
            ColFeed feed;                                           // (SPDP_COLFEED: the column records ride down the lanes)
            constexpr bool OWN = !SPDP_COLFEED;
            if constexpr (OWN) { ld_col(s_lo - m, col1, aux1); ld_col(s_lo + 1 - m, col2, aux2); }
            else { feed.issue(cols, aux, s_lo - m0); feed.arrived(); }
            for (int S = s_lo; S <= s_hi; ++S) {
                if (((S - s_lo) & (CHUNK - 1)) == 0) refill(S);
                const int n = S - m;
                const bool on = any && n >= n_first && n <= n_last;
                int2 col; int ax, adn;
                if constexpr (OWN) {
                    col = col1; ax = aux1 & 0xff; adn = aux1 >> 8;
                    col1 = col2; aux1 = aux2;
                    ld_col(n + 2, col2, aux2);
                } else {
                    feed.shift();                                   // lane k now holds column (S - m0) - k = S - m
                    feed.issue(cols, aux, S + 1 - m0);              // lane 0's next column: at most br + 63, inside the padded records
                    col = make_int2(feed.x, feed.ya & 0xff); ax = (feed.ya >> 8) & 0xff; adn = (feed.ya >> 16) & 0xff;
                }
                if (__ballot(on) == 0) { if constexpr (!OWN) feed.arrived(); continue; }
                const int r = n - m;
                const int q = (r - (lw - 1)) & (RING - 1), ql = (q - 1) & (RING - 1), qu = (q + 1) & (RING - 1);
                St H = {L.hv[q], L.hu[q], L.hl[q], L.hm[q], L.hk[q]};
                St F = {L.fv[q], L.fu[q], L.fl[q], L.fm[q], L.fk[q]};
                const St uH = {L.hv[qu], L.hu[qu], L.hl[qu], L.hm[qu], L.hk[qu]};
                const St uF = {L.fv[qu], L.fu[qu], L.fl[qu], L.fm[qu], L.fk[qu]};
                const St lH = {L.hv[ql], L.hu[ql], L.hl[ql], L.hm[ql], L.hk[ql]};
                St F2 = F, uF2 = F;
                if constexpr (DAGP) {
                    F2 = {L.gv[q], L.gu[q], L.gl[q], L.gm[q], L.gk[q]};
                    uF2 = {L.gv[qu], L.gu[qu], L.gl[qu], L.gm[qu], L.gk[qu]};
                }
                auto val_of = [&](int k) {
                    if (DAGP) return k == K_H ? H.v : (k == K_E ? E.v : (k == K_F ? F.v : (k == K_E2 ? E2.v : F2.v)));
                    return k == K_H ? H.v : (k == K_E ? E.v : F.v);
                };
                int mxk = K_H;
                if (m != P.a_left) {
                    H.v += qprof[col.y];
                    const int x = uH.v + gop;
                    if (x >= uF.v) { F = uH; F.v = x; } else F = uF;
                    F.v += gep;
                    if (F.v >= H.v) mxk = K_F;
                    if (DAGP) {                                     // Vertical2
                        const int x2 = uH.v + lgop;
                        if (x2 >= uF2.v) { F2 = uH; F2.v = x2; } else F2 = uF2;
                        F2.v += lgep;
                        if (F2.v >= val_of(mxk)) mxk = K_F2;
                    }
                }
                if (on) {
                    const unsigned prev_psp = psp;
                    const int x = lH.v + gop;
                    if (x >= E.v) { E = lH; E.v = x; psp = psp ? 1u : 0u; } else psp &= 3u;
                    E.v += gep;
                    if (E.v >= (DAGP ? val_of(mxk) : (mxk == K_H ? H.v : F.v))) mxk = K_E;
                    if (DAGP) {                                     // Horizontal2
                        const int x2 = lH.v + lgop;
                        if (x2 >= E2.v) { E2 = lH; E2.v = x2; if (prev_psp) psp |= 2u; } else psp |= (prev_psp & 2u);
                        E2.v += lgep;
                        if (E2.v >= val_of(mxk)) mxk = K_E2;
                    }
                }
                // ---- acceptor
                bool spj3 = false;
                const bool acc = on && (ax & 2) && ncand >= 0;
                if (__ballot(acc)) {
                    int sel_h = -1, sel_e = -1, sel_f = -1, sel_e2 = -1, sel_f2 = -1;
                    const int s3 = col.x >> 16, dn3 = adn & 15;
#pragma unroll
                    for (int l = 0; l < NC; ++l) {
                        const int len = n - cj[l];
                        if (acc && l <= ncand && len >= llmt) {
                            const int x = cv[l] + sigB + intpen_of_u(T, A, len) + s3 + T.t53[16 * cx[l] + dn3];
                            if (cd[l] == K_H) { if (x > H.v) { H.v = x; sel_h = l; } }
                            else if (cd[l] == K_E) { if (x > E.v) { E.v = x; sel_e = l; } }
                            else if (!DAGP || cd[l] == K_F) { if (x > F.v) { F.v = x; sel_f = l; } }
                            else if (cd[l] == K_E2) { if (x > E2.v) { E2.v = x; sel_e2 = l; } }
                            else { if (x > F2.v) { F2.v = x; sel_f2 = l; } }
                        }
                    }
                    auto pick = [&](int sel, const int* arr) { int v = 0; _Pragma("unroll") for (int l = 0; l < NC; ++l) if (l == sel) v = arr[l]; return v; };
                    int maxk = NODK;
                    if (sel_h >= 0) {
                        psp |= psp_bit(K_H); spj3 = true;
                        H.u = max(pick(sel_h, cu), r); H.l = min(pick(sel_h, cl), r); H.m = pick(sel_h, cm); H.k = pick(sel_h, ck);
                        if (H.v > val_of(mxk)) { maxk = K_H; mxk = K_H; }
                    }
                    if (sel_e >= 0) {
                        psp |= psp_bit(K_E);
                        E.u = max(pick(sel_e, cu), r); E.l = min(pick(sel_e, cl), r); E.m = pick(sel_e, cm); E.k = pick(sel_e, ck);
                        if (E.v > val_of(mxk)) { maxk = K_E; mxk = K_E; }
                    }
                    if (sel_f >= 0) {
                        psp |= psp_bit(K_F);
                        F.u = max(pick(sel_f, cu), r); F.l = min(pick(sel_f, cl), r); F.m = pick(sel_f, cm); F.k = pick(sel_f, ck);
                        if (F.v > val_of(mxk)) { maxk = K_F; mxk = K_F; }
                    }
                    if (DAGP && sel_e2 >= 0) {
                        psp |= psp_bit(K_E2);
                        E2.u = max(pick(sel_e2, cu), r); E2.l = min(pick(sel_e2, cl), r); E2.m = pick(sel_e2, cm); E2.k = pick(sel_e2, ck);
                        if (E2.v > val_of(mxk)) { maxk = K_E2; mxk = K_E2; }
                    }
                    if (DAGP && sel_f2 >= 0) {
                        psp |= psp_bit(K_F2);
                        F2.u = max(pick(sel_f2, cu), r); F2.l = min(pick(sel_f2, cl), r); F2.m = pick(sel_f2, cm); F2.k = pick(sel_f2, ck);
                        if (F2.v > val_of(mxk)) { maxk = K_F2; mxk = K_F2; }
                    }
                    if (is_imd && acc && maxk < NODK) {
                        int sel = maxk == K_H ? sel_h : (maxk == K_E ? sel_e : sel_f);
                        if (DAGP && maxk > K_F) sel = maxk == K_E2 ? sel_e2 : sel_f2;
                        gst<PIPE>(IM(iq, HLNK, 0, r), pick(sel, ck));
                        rlst = r;
                        if (maxk == K_H) H.k = r; else if (maxk == K_E) E.k = r; else if (!DAGP || maxk == K_F) F.k = r;
                        else if (maxk == K_E2) E2.k = r; else F2.k = r;
                        if (maxk == K_H) {
                            if (sel_e >= 0 && E.v > H.v + gop) { E.k = r + width; gst<PIPE>(IM(iq, HLNK, 1, r), pick(sel_e, ck)); }
                            if (sel_f >= 0 && F.v > H.v + gop) F.k = r + width;
                            if (DAGP) {
                                if (sel_e2 >= 0 && E2.v > H.v + lgop) { E2.k = r + 2 * width; gst<PIPE>(IM(iq, HLNK, 2, r), pick(sel_e2, ck)); }
                                if (sel_f2 >= 0 && F2.v > H.v + lgop) F2.k = r + 2 * width;
                            }
                        }
                    }
                }
                // ---- the cell takes the best state
                const int hd = mxk;
                const St MX = DAGP ? st_sel5(mxk, H, E, F, E2, F2) : st_sel3(mxk, H, E, F);      // *mx
                if (hd == K_H) {
                    if (LocalR && on && H.v > best.v) { best = H; best_mr = m; best_nr = n; }
                } else {
                    H = MX;
                    if (H.u < r) H.u = r;
                    if (H.l > r) H.l = r;
                }
                if (LocalL && H.v <= 0) { H.v = 0; H.m = m; H.k = H.u = H.l = r; }
                // ---- donor
                const bool don = on && (ax & 1);
                if (__ballot(don)) {
                    const int sigJ = (int) (short) (col.x & 0xffff) - ipen;
                    const int dn5 = adn >> 4;
                    const int mx_now = hd == K_H ? H.v : MX.v;      // *mx: E / F keep their own value when they won
#pragma unroll
                    for (int k = 0; k < NODK; ++k) {
                        const St src = DAGP ? st_sel5(k, H, E, F, E2, F2) : st_sel3(k, H, E, F);
                        bool t = don && k >= (hd == K_H ? 0 : 1) && !(psp & psp_bit(k));
                        if (t && k != hd) {
                            int z = mx_now;
                            if (hd == K_H || ((k - hd) % 2)) z += (k / 2 == 1) ? gop : ((k / 2 == 2) ? lgop : 0);
                            if (src.v <= z) t = false;
                        }
                        {
                            const bool weak = t && ncand >= NC - 2 && !(src.v + sigJ > cv[NC - 2]);
                            if (weak) ncand = NC - 2;
                            t = t && !weak;
                        }
                        if (__ballot(t)) {
                            const int x = src.v + sigJ;
                            int pos = ncand < NC - 1 ? ncand + 1 : NC - 1;
                            if (t && ncand < NC - 1) ++ncand;
#pragma unroll
                            for (int l = NC - 1; l >= 1; --l) {
                                const bool mv = t && pos == l && x > cv[l - 1];
                                if (mv) { cv[l] = cv[l - 1]; cj[l] = cj[l - 1]; cd[l] = cd[l - 1]; cu[l] = cu[l - 1]; cl[l] = cl[l - 1];
                                          cm[l] = cm[l - 1]; ck[l] = ck[l - 1]; cx[l] = cx[l - 1]; pos = l - 1; }
                            }
                            if (t) {
                                if (pos < NC - 1) {
#pragma unroll
                                    for (int l = 0; l < NC - 1; ++l)
                                        if (l == pos) { cv[l] = x; cj[l] = n; cd[l] = k; cu[l] = src.u; cl[l] = src.l; cm[l] = src.m;
                                                        ck[l] = is_imd ? r : src.k; cx[l] = dn5; }
                                    if (is_imd && k == K_E) gst<PIPE>(IM(iq, HLNK, 0, r), rlst);
                                } else --ncand;
                            }
                        }
                    }
                }
                // ---- an intermediate row records where the paths cross it and restarts ranges and links
                if (is_imd && on) {
                    if (hd == K_H) rlst = r;
                    else if (!spj3 && (hd % 2)) gst<PIPE>(IM(iq, HLNK, 0, r), rlst);
                    gst<PIPE>(IM(iq, VLNK, 0, r), H.k); gst<PIPE>(IM(iq, LWRB, 0, r), min(r, H.l)); gst<PIPE>(IM(iq, UPRB, 0, r), max(r, H.u));
                    H.l = H.u = r; H.k = r;
                    gst<PIPE>(IM(iq, VLNK, 1, r), F.k); gst<PIPE>(IM(iq, LWRB, 1, r), min(r, F.l)); gst<PIPE>(IM(iq, UPRB, 1, r), max(r, F.u));
                    F.l = F.u = r; F.k = r + width;
                    if (DAGP) {
                        gst<PIPE>(IM(iq, VLNK, 2, r), F2.k); gst<PIPE>(IM(iq, LWRB, 2, r), min(r, F2.l)); gst<PIPE>(IM(iq, UPRB, 2, r), max(r, F2.u));
                        F2.l = F2.u = r; F2.k = r + 2 * width;
                    }
                }
                if (on) {
                    L.hv[q] = H.v; L.hu[q] = H.u; L.hl[q] = H.l; L.hm[q] = H.m; L.hk[q] = H.k;
                    L.fv[q] = F.v; L.fu[q] = F.u; L.fl[q] = F.l; L.fm[q] = F.m; L.fk[q] = F.k;
                    if constexpr (DAGP) { L.gv[q] = F2.v; L.gu[q] = F2.u; L.gl[q] = F2.l; L.gm[q] = F2.m; L.gk[q] = F2.k; }
                }
                if constexpr (!OWN) feed.arrived();
            }
            WAVE_SYNC();
            for (int e = res_lo + lane; e < res_hi; e += 64) {
                const int q = e & (RING - 1);
#pragma unroll
                for (int a = 0; a < NA; ++a) gst<PIPE>(G(a) + e, lds[a][q]);
            }
        }
        // `rlst` of this tile's intermediate row is what the next one starts from
        if (PIPE) { if (is_imd) gst<true>(rlf + iq, rlst); }
        else if (imd_mask) rlst = __shfl(rlst, __ffsll((long long) imd_mask) - 1);
    }
    if (PIPE) {
        const int ti = t_lo;
        if (LocalR) {
            for (int off = 32; off; off >>= 1) {
                St o; o.v = __shfl_xor(best.v, off); o.u = __shfl_xor(best.u, off); o.l = __shfl_xor(best.l, off);
                o.m = __shfl_xor(best.m, off); o.k = __shfl_xor(best.k, off);
                const int om = __shfl_xor(best_mr, off), on_ = __shfl_xor(best_nr, off);
                if (o.v > best.v || (o.v == best.v && (om < best_mr || (om == best_mr && on_ < best_nr)))) { best = o; best_mr = om; best_nr = on_; }
            }
            if (lane == 0) {
                int* b = tbest + 8 * ti;
                gst<true>(b, best.v); gst<true>(b + 1, best.u); gst<true>(b + 2, best.l); gst<true>(b + 3, best.m);
                gst<true>(b + 4, best.k); gst<true>(b + 5, best_mr); gst<true>(b + 6, best_nr);
            }
        }
        if (ti > 0 && !stalled) stalled = !wait_for(ti - 1, INT32_MAX);
        publish(ti, INT32_MAX);
        if (ti != n_tiles - 1) return;
        if (LocalR) {                                               // tiles in row order: the first maximum wins
            best = {NEV, 0, 0, al, 0}; best_mr = ar; best_nr = br;
            for (int t = lane; t < n_tiles; t += 64) {
                const int* b = tbest + 8 * t;
                const int ov = gld<true>(b);
                if (ov > best.v) { best = {ov, gld<true>(b + 1), gld<true>(b + 2), gld<true>(b + 3), gld<true>(b + 4)};
                                   best_mr = gld<true>(b + 5); best_nr = gld<true>(b + 6); }
            }
        }
    } else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    for (int i = lane; i < 10 * (n_im + 1); i += 64) cpos[i] = EOU;
    STORES_DRAINED();

    // ---- the end cell (hlastS_ng) or the tracked local maximum
    auto GL = [&](int arr, int r) { return gld<PIPE>(G(arr) + (r - (lw - 1))); };
    auto scan_best = [&](int lo, int hi, int step, int start_r, int start_v) {
        int bv = start_v, bk = INT32_MAX;
        const int cnt = hi - lo + 1;
        for (int i = lane; i < cnt; i += 64) {
            const int r = step > 0 ? lo + i : hi - i;
            const int v = GL(0, r);
            if (v > bv) { bv = v; bk = i; }
        }
        for (int off = 32; off; off >>= 1) {
            const int ov = __shfl_xor(bv, off), ok = __shfl_xor(bk, off);
            if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        if (bk == INT32_MAX) return start_r;
        return step > 0 ? lo + bk : hi - bk;
    };
    St mxs = best;
    int flag = 0;
    const int rr = br - ar;
    if (LocalR) {
        for (int off = 32; off; off >>= 1) {
            St o; o.v = __shfl_xor(best.v, off); o.u = __shfl_xor(best.u, off); o.l = __shfl_xor(best.l, off);
            o.m = __shfl_xor(best.m, off); o.k = __shfl_xor(best.k, off);
            const int om = __shfl_xor(best_mr, off), on_ = __shfl_xor(best_nr, off);
            if (o.v > best.v || (o.v == best.v && (om < best_mr || (om == best_mr && on_ < best_nr)))) { best = o; best_mr = om; best_nr = on_; }
        }
        mxs = best;
    } else {
        const int r9 = br - ar;
        int mxr = r9;
        if (b_exgr) { const int rw = min(up, br - al); if (rw > r9) mxr = scan_best(r9 + 1, rw, -1, mxr, GL(0, mxr)); }
        if (a_exgr) { const int rw = max(lw, bl - ar); if (rw < r9) mxr = scan_best(rw, r9 - 1, +1, mxr, GL(0, mxr)); }
        mxs = {GL(0, mxr), GL(1, mxr), GL(2, mxr), GL(3, mxr), GL(4, mxr)};
        if (b_exgr && rr < mxr) ar = br - mxr;
        if (a_exgr && rr > mxr) br = ar + mxr;
    }
    if (lane != 0) return;
    int score = mxs.v;
    if (LocalR) {
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        ar = best_mr; br = best_nr;
        if (i < 0) i = 0;
        CPOS(i, 8) = mxs.l;
        CPOS(i, 9) = mxs.u;
    }
    // ---- walk the links back: one cpos row per intermediate the path crosses
    // a link that stands for "rlst as the intermediate rows above left it"
    auto hlnk = [&](int ii, int d, int rr_) {
        int v = gld<PIPE>(IM(ii, HLNK, d, rr_));
        if (PIPE && v == INH) {
            v = 0x7fffffff;
            for (int j = ii - 1; j >= 0; --j) { const int w = gld<true>(rlf + j); if (w != INH) { v = w; break; } }
        }
        return v;
    };
    int i = n_im;
    while (--i >= 0 && mi_of(i) > ar) ;
    if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
    int r = br - ar;
    CPOS(i + 1, 8) = min(mxs.l, r);
    CPOS(i + 1, 9) = max(mxs.u, r);
    r = mxs.k;
    for ( ; i >= 0 && mi_of(i) > mxs.m; --i) {
        int c = 0, d = 0;
        for ( ; r > up; r -= width) ++d;
        if (d > NOL - 1 || r < lw - 1) { flag = -3; break; }              // outside the link arrays (undefined in the reference)
        const int mi = mi_of(i);
        if (gld<PIPE>(IM(i, VLNK, d, r)) < EOU) {
            CPOS(i, c++) = mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            for (int rp = hlnk(i, d, r); lw <= rp && rp < up && r != rp; rp = hlnk(i, 0, r = rp)) {
                if (c >= 6) { flag = -3; break; }                   // the terminator would land on [8]
                CPOS(i, c++) = r + mi;
            }
            if (flag) break;
            CPOS(i, c++) = r + mi;
            CPOS(i, c) = EOU;
            CPOS(i, 8) = gld<PIPE>(IM(i, LWRB, d, r));
            CPOS(i, 9) = gld<PIPE>(IM(i, UPRB, d, r));
            r = gld<PIPE>(IM(i, VLNK, d, r));
            if (r == EOU) break;
        } else
            CPOS(i, 0) = EOU;
    }
    if (!flag) {
        for ( ; r > up; r -= width) ;
        if (LocalL) { al = mxs.m; bl = r + mxs.m; }
        else {
            const int rl = bl - al;
            if (b_exgl && rl > r) {
                al = bl - r;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = EOU;
            }
            if (a_exgl && rl < r) bl = al + r;
        }
        ++i;
        if (i >= n_im) flag = -3;                                   // the reference dereferences udhimds[n_im]
        else if (mi_of(i) < al || CPOS(i, 2) < bl) score = NEV;
        else if (CPOS(i, 8) == EOU || CPOS(i, 9) == EOU) flag = -3; // bounds the reference never set
        else {
            const int rl = bl - al;
            CPOS(i, 8) = min(rl, CPOS(i, 8));
            CPOS(i, 9) = max(rl, CPOS(i, 9));
        }
    }
#undef CPOS
    A.scores[pi] = score;
    A.ranges[4 * pi] = al; A.ranges[4 * pi + 1] = ar; A.ranges[4 * pi + 2] = bl; A.ranges[4 * pi + 3] = br;
    DevResult R;
    R.score = score; R.mr = ar; R.nr = br; R.ml = al; R.ulk = 0; R.maxr = 0; R.pad[0] = flag; R.pad[1] = 0;
    A.res[pi] = R;
}

extern "C" hipError_t spdp_launch_rowwave_udh(const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    const int units = A.pipe ? A.n_items : A.n_probs;
    constexpr int W3 = WPBU<true>, W2 = WPBU<false>;
    if (A.noll == 3) {
        if (A.pipe) hipLaunchKernelGGL((spdp_rowwave_udh<true, true>), dim3((units + W3 - 1) / W3), dim3(64 * W3), 0, stream, A);
        else hipLaunchKernelGGL((spdp_rowwave_udh<false, true>), dim3((units + W3 - 1) / W3), dim3(64 * W3), 0, stream, A);
    } else if (A.pipe) hipLaunchKernelGGL(spdp_rowwave_udh<true>, dim3((units + W2 - 1) / W2), dim3(64 * W2), 0, stream, A);
    else hipLaunchKernelGGL(spdp_rowwave_udh<false>, dim3((units + W2 - 1) / W2), dim3(64 * W2), 0, stream, A);
    return hipGetLastError();
}

extern "C" hipError_t spdp_launch_rowwave(int forward, const ScalarArgs* a, hipStream_t stream)
{
    ScalarArgs A = *a;
    const dim3 blk(64 * WPB);
    const bool dagp = A.noll == 3;                      // double affine gaps: the F2 / E2 states
    if (A.pipe) {                                       // one wave per (problem, tile)
        const dim3 grd((A.n_items + WPB - 1) / WPB);
        if (dagp) {
            if (forward) hipLaunchKernelGGL((spdp_rowwave<1, true, false, true>), grd, blk, 0, stream, A);
            else hipLaunchKernelGGL((spdp_rowwave<0, true, false, true>), grd, blk, 0, stream, A);
        } else if (forward) hipLaunchKernelGGL((spdp_rowwave<1, true>), grd, blk, 0, stream, A);
        else hipLaunchKernelGGL((spdp_rowwave<0, true>), grd, blk, 0, stream, A);
        return hipGetLastError();
    }
    const dim3 grd((A.n_probs + WPB - 1) / WPB);
    if (dagp) {
        if (forward == 2) hipLaunchKernelGGL((spdp_rowwave<1, false, true, true>), grd, blk, 0, stream, A);
        else if (forward) hipLaunchKernelGGL((spdp_rowwave<1, false, false, true>), grd, blk, 0, stream, A);
        else hipLaunchKernelGGL((spdp_rowwave<0, false, false, true>), grd, blk, 0, stream, A);
    } else if (forward == 2) hipLaunchKernelGGL((spdp_rowwave<1, false, true>), grd, blk, 0, stream, A);      // every problem with a cut range
    else if (forward) hipLaunchKernelGGL((spdp_rowwave<1, false>), grd, blk, 0, stream, A);
    else hipLaunchKernelGGL((spdp_rowwave<0, false>), grd, blk, 0, stream, A);
    return hipGetLastError();
}
