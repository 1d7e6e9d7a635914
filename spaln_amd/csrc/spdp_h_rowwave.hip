// spdp_h_rowwave.hip -- the exact-intron-length ("-A0") protein x genome engines as wavefront kernels.
//
// What they compute (ogotoh/spaln v3.0.7; restated for the checker in oracle/spdp_oracle_h_scalar.c):
//   MODE 0  Aln2h1::forwardH_ng without a Vmf (HomScoreH_ng)                  src/fwd2h1.cc:294-617, 143-292
//   MODE 1  forwardH_ng + initH_ng / lastH_ng, Vmf::traceback and the record fix-up of trcbkalignH_ng
//                                                                             src/vmf.cc:125, src/fwd2h1.cc:2019-2036
//   MODE 2  Aln2h1::hirschbergH_ng + hinitH_ng / hlastH_ng with UdhIntermediate(lub = true)
//                                                                             src/fwd2h1.cc:1085-1520, 941-1083
// int32 scores; a cell (m, n) pairs residue m with the codon ending at n; three gap states (match H, insertion E in
// a queue of three codon phases, deletion F) with 1- and 2-nt frame shifts; a sorted list of the best donor
// candidates per row AND codon phase; acceptors priced with IntPen(length) + the junction table, and a codon
// split by the intron re-scored from the four bases around it.  The -A2 / -A3 dispatch runs this engine on
// sub-problems below 8 query rows, -A0 runs it on everything.
//
// The reference walks the matrix row by row over arrays indexed by DIAGONAL r = n - 3m that it updates in place; what
// a cell sees at the band edges is whatever those arrays hold, so the arrays stay as they are.  Mapping (ours): one
// wave per problem, lane k owns row m0 + k of a tile of up to 64 rows with all per-row state (the insertion queue,
// three candidate lists) in its registers; at step S it is on column S - m, i.e. on array entry S - 4m: it reads
// entries r - 3 .. r + 3 -- its own last three cells and the four cells of the row above that the lane above wrote one
// to four steps earlier -- and writes r.  The arrays are shared through an LDS window of 512 diagonals sliding with
// the sweep, streamed from / to global memory 64 entries at a time between tiles.  Every state carries riders:
// a Vmf record number (MODE 1) or the diagonal range, start row and intermediate-row link of its path (MODE 2).
// Vmf records are appended through a wave-uniform counter (ballot + prefix count).  The sequential boundary rules
// (leading / trailing gap relaxation along the first and last row) run on one lane over LDS-staged chunks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"
#include "spdp_pipe.h"

namespace {

constexpr int NEV = INT32_MIN / 16 * 7;                 // NEVSEL, src/cmn.h:79
constexpr int EOU = 0x7fffffff - 2;                     // end_of_ulk, src/aln.h:49
constexpr int RING = 512;                               // diagonals resident in LDS
constexpr int CRING = 128;                              // columns resident in LDS
constexpr int CHUNK = 16;                               // steps between two refills of the window (32 until round 4: a tile that has caught up
                                                        // with the one above waits for its next publish, half a chunk on average)
constexpr int NC = 5;                                   // NCAND + 1 slots per candidate list
constexpr int WPB = 4;                                  // waves (= problems) per block
constexpr int IPEN_LDS = 4096;
constexpr int AMB = 2;                                  // the ambiguity code of both alphabets

// TraceBackDir (src/aln.h:30-35): what a state remembers about its last step
enum : int { T_DEAD = 0, T_DIAG = 2, T_NEWD = 3, T_VERT = 4, T_SLA1 = 5, T_SLA2 = 6, T_VERL = 7, T_HORI = 8, T_HOR1 = 9,
             T_HOR2 = 10, T_HORL = 11, T_NEWV = 12, T_NEWH = 13, T_SPIN = 16 };
constexpr unsigned M_DIAG = 1u << T_DIAG | 1u << T_NEWD;
constexpr unsigned M_VERT = 1u << T_VERT | 1u << T_SLA1 | 1u << T_SLA2 | 1u << T_VERL | 1u << T_NEWV;
constexpr unsigned M_HORI = 1u << T_HORI | 1u << T_HOR1 | 1u << T_HOR2 | 1u << T_HORL | 1u << T_NEWH;
__device__ __forceinline__ bool is_kind(int d, unsigned mask) { return (mask >> (d & 15)) & 1u; }
// which of the three states a direction belongs to (dir2nod, src/aln.h:50-56): 0 H, 1 E, 2 F, 3 / 4 the long-gap forms
__device__ __forceinline__ int node_of(int d)
{
    d &= 15;
    if (is_kind(d, M_DIAG)) return 0;
    if (d == T_VERL) return 4;
    if (d == T_HORL) return 3;
    if (is_kind(d, M_VERT)) return 2;
    if (is_kind(d, M_HORI)) return 1;
    return -1;
}

struct Tables {
    int mtx[32 * 32];
    short ipen[IPEN_LDS];
    short t53[256];
    uint8_t mid[32];
    uint8_t tron_of[64];
    IpenRuns runs;                                      // IntPen beyond the table above (spdp_ipen_runs.h)
    int gain[3];                                        // max IntPen, max junction-pair score, max |mtx|: what an acceptor can add at most
};
__device__ __forceinline__ void load_tables(Tables& T, const HScalarArgs& A, const DevScoringH* sc)
{
    if (threadIdx.x < 3) T.gain[threadIdx.x] = threadIdx.x == 2 ? 0 : INT32_MIN;
    __syncthreads();
    {
        int pm = INT32_MIN, tm = INT32_MIN, mm = 0;
        for (int i = threadIdx.x; i < A.intpen_len; i += blockDim.x) pm = max(pm, (int) A.intpen[i]);
        for (int i = threadIdx.x; i < 256; i += blockDim.x) tm = max(tm, (int) A.t53[i]);
        for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) mm = max(mm, abs(sc->mtx[i]));
        atomicMax(&T.gain[0], pm); atomicMax(&T.gain[1], tm); atomicMax(&T.gain[2], mm);
    }
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) T.mtx[i] = sc->mtx[i];
    for (int i = threadIdx.x; i < IPEN_LDS; i += blockDim.x) T.ipen[i] = A.intpen[min(i, A.intpen_len - 1)];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) T.t53[i] = A.t53[i];
    if (threadIdx.x < 32) T.mid[threadIdx.x] = A.mid[threadIdx.x];
    if (threadIdx.x < 64) T.tron_of[threadIdx.x] = A.tron_of[threadIdx.x];
    ipen_runs_load(T.runs, A.ipen_runs);
    __syncthreads();                                    // the only block-wide barrier
}
#define WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// a DP state: value, direction, riders (MODE 1: a = Vmf record; MODE 2: a = upr, b = lwr, c = ml, e = ulk)
struct St { int v, d, a, b, c, e; };
template <int MODE> struct Shape { static constexpr int NF = MODE == 0 ? 2 : (MODE == 1 ? 3 : 6); };
// picks one of two / three state records FIELD BY FIELD: `c ? a : b` on the records themselves is an lvalue -- a pointer is
// selected and the record copied from memory, which parks every record it may name in scratch memory (round 4: h, ea
// and f lived there and every access of the step waited for a round trip, 65 % of the wave cycles in SQ_WAIT_ANY)
__device__ __forceinline__ St st_sel(bool c, const St& a, const St& b)
{
    St r;
    r.v = c ? a.v : b.v; r.d = c ? a.d : b.d; r.a = c ? a.a : b.a; r.b = c ? a.b : b.b; r.c = c ? a.c : b.c; r.e = c ? a.e : b.e;
    return r;
}
__device__ __forceinline__ St st_sel3(int k, const St& s0, const St& s1, const St& s2) { return st_sel(k == 0, s0, st_sel(k == 1, s1, s2)); }
__device__ __forceinline__ St st_sel5(int k, const St& s0, const St& s1, const St& s2, const St& s3, const St& s4)
{
    return st_sel(k < 2, st_sel(k == 0, s0, s1), st_sel(k == 2, s2, st_sel(k == 3, s3, s4)));
}
__device__ __forceinline__ int& fld(St& s, int i) { return i == 0 ? s.v : i == 1 ? s.d : i == 2 ? s.a : i == 3 ? s.b : i == 4 ? s.c : s.e; }

// donor candidates of one codon phase, best first
template <int MODE> struct Cands {
    int v[NC], j[NC], x[NC];                            // value, donor position, packed {state, dinc5, the two bases before the donor}
    int a[NC], b[NC], c[NC], e[NC];                     // riders of the state it left from
    int n;                                              // index of the last one, -1: none
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int l = 0; l < NC; ++l) { v[l] = NEV; j[l] = 0; x[l] = 0; a[l] = MODE == 2 ? INT32_MIN : 0; b[l] = INT32_MAX; c[l] = 0; e[l] = EOU; }
        n = -1;
    }
    // the free slot starts below the list and moves up past every entry the newcomer ties or beats; a full list
    // drops its last entry, and a newcomer that beats nobody in a full list takes that entry with it
    __device__ __forceinline__ bool insert(bool t, int val, int pos_n, int packed, const St& src, int ulk)
    {
        int pos = n < NC - 1 ? n + 1 : NC - 1;
        if (t && n < NC - 1) ++n;
#pragma unroll
        for (int l = NC - 1; l >= 1; --l) {
            const bool mv = t && pos == l && val >= v[l - 1];
            if (mv) { v[l] = v[l - 1]; j[l] = j[l - 1]; x[l] = x[l - 1]; a[l] = a[l - 1];
                      if (MODE == 2) { b[l] = b[l - 1]; c[l] = c[l - 1]; e[l] = e[l - 1]; }
                      pos = l - 1; }
        }
        if (!t) return false;
        if (pos < NC - 1) {
#pragma unroll
            for (int l = 0; l < NC - 1; ++l)
                if (l == pos) { v[l] = val; j[l] = pos_n; x[l] = packed; a[l] = src.a;
                                if (MODE == 2) { b[l] = src.b; c[l] = src.c; e[l] = ulk; } }
            return true;
        }
        --n;
        return false;
    }
};
template <class A> __device__ __forceinline__ int pick(int sel, const A& arr)
{
    int v = 0;
#pragma unroll
    for (int l = 0; l < NC; ++l) if (l == sel) v = arr[l];
    return v;
}

}   // namespace

// PIPE (as spdp_rowwave.hip): the tiles of one problem run as separate waves, each a few dozen anti-diagonals behind
// the tile above it.  HScalarArgs::items lists (problem, tile) in dispatch order and a wave draws the next one from a
// ticket counter, so a tile's predecessor is always resident or done.  The arrays then cross CUs: their accesses
// go to the memory side (agent-scope atomics; the per-XCD L2s are not coherent with each other) and a tile
// publishes, once its stores have drained, the anti-diagonal up to which it has handed its entries back
// (prog[tile], + 1; INT_MAX = finished).  Vmf numbers are reserved SPDP_VMF_CHUNK at a time from the problem's counter.
// MODE 2: `rlst`, which would tie a tile to the END of the intermediate row above it, is only ever stored (into
// HLNK): a tile starts from the marker INH + slot, every intermediate row leaves what it ends with in rlf[], and
// the link walk replaces the marker by what the rows above left.
constexpr int INH = 0x7ffffff0;
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

// DAGP: double affine gaps (PwdB::Noll = 3, -yl3; src/fwd2h1.cc:297, 343, 365, 413-449, 577-598 / 1088, 1140, 1162, 1211-1247,
// 1316-1330, 1412-1440; round 5): a second deletion state F2 (a third group of planes by diagonal) and a second insertion
// queue E2, priced with GapW3L = LongGOP + LongGEP / LongGEP; five states a donor candidate can leave from (its state code
// takes a third bit of the packed word); an intermediate row keeps three planes of links.  One wave per problem, two
// problems per block (the third group of planes does not leave LDS for four).
// CUT (MODE 1, one wave per problem): forwardH_ng with a cut range (shortcutH_ng; src/fwd2h1.cc:308-312, 589-603, lastH_ng
// :210-281).  After column cut_l of a row the three insertion states, charged for the codons of the cut, REPLACE the row's
// last three entries (F: black) and the row goes on at column cut_l + cut_len + 1 on the next entry: a cell is addressed
// by its virtual column v (= n up to cut_l, n - cut_len behind it), the data of a column by its real position.  The
// reference finishes a row before it starts the next, so the next row's cells at cut_l - 2 .. cut_l already see the
// replaced entries -- two columns ahead of what the one-column skew of the wave provides.  Hence three phases per tile:
// the skewed sweep up to column cut_l - 3, the seam (cut_l - 2 .. cut_l and the replacement) row after row on one lane
// at a time, and the skewed sweep again from cut_l + 1.
template <int MODE, bool PIPE, bool CUT = false, bool DAGP = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void spdh_rowwave(HScalarArgs A)
{
    static_assert(!CUT || (MODE == 1 && !PIPE), "the cut range: forward engine, one wave per problem");
    static_assert(!DAGP || !PIPE, "double affine gaps: one wave per problem");
    constexpr bool FWD = MODE == 1, UDH = MODE == 2;
    constexpr int NF = Shape<MODE>::NF;
    constexpr int NP = DAGP ? 3 : 2;                    // groups of planes by diagonal: H, F (, F2)
    constexpr int NODK = DAGP ? 5 : 3;                  // states (Nod): H, E, F (, E2, F2)
    constexpr int NOLL = DAGP ? 3 : 2;
    constexpr int WPBK = DAGP ? 2 : WPB;                // waves (= problems) of a block
    __shared__ int Lw[WPBK][NP * NF][RING];
    // the column records and signal quadruples of the columns the wave is on (its 64 rows sit on 64 + 31 consecutive
    // columns during a chunk of steps, and a cell looks 2 back and 4 ahead): staged 32 columns at a time, coalesced --
    // an acceptor or donor used to wait for four or five dependent reads from memory, and some lane is on one at
    // almost every step
    __shared__ int4 Ccol[WPBK][CRING];
    __shared__ short4 Caux[WPBK][CRING];
    __shared__ Tables T;
    const DevScoringH* sc = A.sc;
    load_tables(T, A, sc);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (scalar: the problem, its ranges and its arrays stay in SGPRs)
    int (*L)[RING] = Lw[wv];                            // L[f] = field f of H, L[NF + f] = field f of F
    int4* const Cc = Ccol[wv];
    short4* const Ca = Caux[wv];
    const int lane = threadIdx.x & 63;
    if (wv >= WPBK) return;                             // (not reached: the launch has WPBK waves per block)
    int pi = blockIdx.x * WPBK + wv;
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
    const DevProblemH P = wave_uniform(A.probs[pi]);
    int al = P.a_left, ar = P.a_right, bl = P.b_left, br = P.b_right;
    const int lw = P.lw, up = P.up, width = P.width, n_im = UDH ? P.n_im : 0, intvl = P.imd_intvl;
    const int a_exgl = P.a_exgl, a_exgr = P.a_exgr, b_exgl = P.b_exgl, b_exgr = P.b_exgr;
    const bool Local = sc->local;
    const bool LocalL = Local && a_exgl && b_exgl, LocalR = Local && a_exgr && b_exgr;
    const int gop = sc->gop, gep = sc->gep, lgep = sc->lgep, codonk1 = sc->codonk1;
    const int gw1 = sc->g1, gw2 = sc->g2, gw3 = sc->g3, ge1 = A.gape1, ge2 = A.gape2;
    const int lgop = DAGP ? A.lgop : 0, gw3l = lgop + lgep;         // PwdB::GapW3L (src/aln2.cc:124)
    const bool spj = sc->spj && !P.nospj;
    const int minl = A.minl;
    const int cutlen = CUT ? P.cut_len : 0, cut_l = CUT ? P.cut_l : INT32_MAX / 2;
    const int xshift = CUT ? max(0, cutlen - 16) : 0;               // staged columns: the two sides of the cut 16 slots apart
    auto XC = [&](int c) -> int { return (!CUT || c <= cut_l + 8) ? c : c - xshift; };     // real column -> staging index
    auto CX = [&](int x) -> int { return (!CUT || x <= cut_l + 8) ? x : x + xshift; };
    auto RV = [&](int v) -> int { return (!CUT || v <= cut_l) ? v : v + cutlen; };         // virtual -> real column
    const uint8_t* __restrict__ acod = A.a_codes + P.a_off;
    const int4* __restrict__ cols = A.cols + P.col_off;             // .x: codon ending at n | flags, .y: sig3 x 2, .w: dinc
    const short4* __restrict__ aux = A.aux + P.col_off;             // {sigS, sigT, sigE, sig5}
    const int W = width + 4;                                        // NP NF arrays of W ints: entry e = r - lw + 3
    int* const g0 = A.work + P.bnd_off;
    auto G = [&](int arr) { return g0 + (int64_t) arr * W; };
    auto gext3 = [&](int i) { return i > codonk1 ? lgep : gep; };
    auto tron_at = [&](int i) -> int { return (i < 0 || i > P.b_len) ? AMB : ((cols[i + 2].x >> 16) & 0xff); };   // the codon starting at i
    auto intpen_of = [&](int len) -> int {
        if (len < 0) return -32768;
        if (len < IPEN_LDS) return T.ipen[len];
        if (A.ipen_runs) return ipen_runs_get(T.runs, len, A.intpen_len);      // (kernel-uniform)
        return A.intpen[min(len, A.intpen_len - 1)];
    };

    // ---- Vmf (MODE 1): record 0 is never used
    int3* __restrict__ vrec = A.vmf + P.tb_off;
    int* __restrict__ vraw = reinterpret_cast<int*>(vrec);
    const int vcap = (int) P.imd_off;
    // PIPE: what the tiles of the problem share: {record numbers handed out, overflow}, prog[max_tiles],
    // best[max_tiles][8], rlf[n_im][3]
    int* __restrict__ sy = PIPE ? A.pipe + (size_t) pi * A.pipe_stride : nullptr;
    int* __restrict__ prog = PIPE ? sy + 2 : nullptr;
    int* __restrict__ tbest = PIPE ? sy + 2 + A.max_tiles : nullptr;
    int* __restrict__ rlf = PIPE ? sy + 2 + 9 * A.max_tiles : nullptr;
    int vcount = 1;
    int vleft = 0;                                      // PIPE: numbers left of the chunk this wave holds
    bool vover = false;
    auto vadd = [&](bool need, int mm, int nn, int pp) -> int {
        if (!FWD) return 0;
        const unsigned long long mask = __ballot(need);
        if (!mask) return 0;
        const int cnt = __popcll(mask);
        if (PIPE && cnt > vleft) {
            int b = 0;
            if (lane == 0) b = __hip_atomic_fetch_add(sy, SPDP_VMF_CHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vcount = 1 + __builtin_amdgcn_readfirstlane(b);
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
    auto wait_for = [&](int t, int req) -> bool {
        long spins = 0;
        while (__hip_atomic_load(prog + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < req) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1l << 22)) {                 // (cannot happen with the ticket order; bounds every spin)
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
    // ---- intermediate rows (MODE 2): hlnk[Noll], vlnk[Noll], lwrb[Noll], uprb[Noll], `width` ints each, entry r - lw + 1
    int* const imd_base = UDH ? A.imd + P.imd_off : nullptr;
    const int64_t us = NOLL * (int64_t) width;
    enum { HLNK = 0, VLNK = 1, LWRB = 2, UPRB = 3 };
    auto IM = [&](int i, int arr, int k, int r) -> int* { return imd_base + (int64_t) i * 4 * us + arr * us + (int64_t) k * width + (r - lw + 1); };
    auto mi_of = [&](int i) { return P.a_left + (i + 1) * intvl; };
    int* cpos = UDH ? A.cpos + (int64_t) pi * A.cpos_stride : nullptr;
#define CPOS(i, c) cpos[(i) * 10 + (c)]

    const int r_corner = bl - 3 * al;                               // the diagonal of the start cell
    const int r_floor = bl - 3 * ar;
    const St black = UDH ? St{NEV, 0, r_floor, r_floor, 0, EOU} : St{NEV, 0, 0, 0, 0, 0};
    auto put = [&](int e, int isF, const St& s) {                  // one entry to global memory
        St t = s;
#pragma unroll
        for (int f = 0; f < NF; ++f) gst<PIPE>(G(isF * NF + f) + e, fld(t, f));
    };
    auto imd_init = [&](int i) {
        int* b = imd_base + (int64_t) i * 4 * us;
        for (int64_t q = lane; q < us; q += 64) {
            gst<PIPE>(b + q, EOU); gst<PIPE>(b + us + q, EOU); gst<PIPE>(b + 2 * us + q, INT32_MAX); gst<PIPE>(b + 3 * us + q, INT32_MIN);
        }
    };

    // ---- the arrays as initH_ng / hinitH_ng leave them
    if (t_lo == 0) {
        if (UDH && !PIPE) for (int i = 0; i < n_im; ++i) imd_init(i);       // (PIPE: by the tile that holds the row)
        // the start cell, and below it the first column (leading gap of the query side): closed form per entry
        const int first_dir = a_exgl ? T_DEAD : T_DIAG;
        const int sS0 = aux[bl + 1].x;
        St corner = black;
        corner.v = (a_exgl && sS0 > 0) ? sS0 : 0;
        corner.d = first_dir;
        const int p0 = vadd(lane == 0, al, bl, 0);
        if (FWD) corner.a = __shfl(p0, 0);
        if (UDH) { corner.a = corner.b = corner.e = r_corner; corner.c = al; }
        const int r_low = max(lw, r_floor);
        for (int e = lane; e < W; e += 64) {
            const int r = e + lw - 3;
            St h = black;
            if (r == r_corner) h = corner;
            else if (r < r_corner && r >= r_low) {
                const int i = r_corner - r;                         // nucleotides of the gap
                if (b_exgl == 1) {
                    h.v = 0; h.d = T_DEAD;
                    if (FWD) h.a = 0;
                    if (UDH) { h.a = h.b = h.e = r; h.c = al + i / 3; }
                } else {
                    const int base = (i - 1) % 3 + 1, steps = (i - base) / 3;       // i = base + 3 steps: whole codons after a 1-, 2- or 3-nt start
                    h = corner;
                    h.d = T_VERT;
                    if (!(b_exgl & 2)) h.v += gep;
                    if (!(b_exgl & 1)) h.v += gop;
                    if (base < 3) h.v += A.extragop;
                    if (!(b_exgl & 2)) {
                        const int cheap = min(steps, max(0, (codonk1 - base) / 3));         // codons still inside the short-gap regime
                        h.v += cheap * gep + (steps - cheap) * lgep;
                    }
                    if (UDH) { h.b = h.e = r; h.c = al + i / 3; }
                }
            }
            put(e, 0, h);
            put(e, 1, black);
            if constexpr (DAGP) put(e, 2, black);
        }
        // the first row: a leading gap on the genomic side may restart wherever a start codon signal beats it --
        // a running maximum with restarts, one lane, history in registers
        if (a_exgl) {
            const int r_top = min(up, br - 3 * al);
            St h1 = corner, h2 = black, h3 = black;                 // the entries 1, 2, 3 below the current one
            int since[3] = {bl, 0, 0};                              // where the gap of each frame (MODE 2: of all frames) started
            for (int r = r_corner + 1; r <= r_top; ++r) {
                const int i = r - r_corner, n = bl + i;
                const int sS = max(0, (int) aux[n + 1].x);
                St h;
                bool fresh = false;
                if (i < 3) { h = corner; h.v = sS; h.d = first_dir; fresh = true; if (!UDH) since[i] = n; }
                else {
                    h = h3;
                    const int k = n - since[UDH ? 0 : i % 3];
                    if (k == 3 && !(a_exgl & 1)) h.v += gop;
                    if (!(a_exgl & 2)) h.v += gext3(k);
                    h.v += aux[n - 2].z;
                    h.d = T_HORI;
                    if (h1.v + gw1 > h.v) { h = h1; h.v += gw1; h.d = T_HOR1; }
                    if (h2.v + gw2 > h.v) { h = h2; h.v += gw2; h.d = T_HOR2; }
                }
                if (h.v < sS) { h.v = sS; h.d = T_DEAD; fresh = true; since[UDH ? 0 : i % 3] = n; }
                const int p = vadd(lane == 0 && fresh, al, n, 0);
                if (FWD && fresh) h.a = p;
                if (UDH) { if (fresh) { h.b = h.e = r; if (i < 3) h.c = al; } h.a = r; }
                if (lane == 0) put(r - lw + 3, 0, h);
                h3 = h2; h2 = h1; h1 = h;
            }
        }
    }

    // running maximum of a local right end: first maximum in row-major order
    St best = black; int best_m = UDH ? ar : al, best_n = UDH ? br : bl;
    best.v = NEV; best.c = al;
    int rl0 = INT32_MAX, rl1 = INT32_MAX, rl2 = INT32_MAX;         // MODE 2: `rlst` per queue slot, carried from one intermediate row to the next

    const int R0 = al + (a_exgl ? 1 : 0);
    const int TH = UDH ? max(1, min(64, intvl)) : 64;              // a tile holds at most one intermediate row
    const int n_tiles = max(1, (ar - R0 + TH) / TH);               // (the host lists the same count)
    bool stalled = false;
    for (int ti = t_lo; ti < t_hi; ++ti) {
        const int m0 = R0 + TH * ti;
        if (m0 > ar) break;
        if (!PIPE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        const int m = m0 + lane;
        const bool row = lane < TH && m <= ar;
        const int n0 = max(3 * m + lw - 1, bl), n9 = min(3 * m + up, br);
        const bool any = row && n0 <= n9;
        const int v0 = n0, v9 = !CUT ? n9 : (n9 <= cut_l ? n9 : (n9 > cut_l + cutlen ? n9 - cutlen : cut_l));    // (the host: n0 <= cut_l)
        int s_lo = any ? v0 + m : INT32_MAX, s_hi = any ? v9 + m : INT32_MIN;
        for (int off = 32; off; off >>= 1) { s_lo = min(s_lo, __shfl_xor(s_lo, off)); s_hi = max(s_hi, __shfl_xor(s_hi, off)); }
        const int iq = UDH ? (m - P.a_left) / max(1, intvl) - 1 : -1;
        const bool is_imd = UDH && row && intvl > 0 && (m - P.a_left) % intvl == 0 && iq >= 0 && iq < n_im;
        const unsigned long long imd_mask = __ballot(is_imd);
        if (UDH && PIPE && imd_mask) {
            const int i0 = __shfl(iq, __ffsll((long long) imd_mask) - 1);
            imd_init(i0);
            rl0 = i0 == 0 ? INT32_MAX : INH; rl1 = i0 == 0 ? INT32_MAX : INH + 1; rl2 = i0 == 0 ? INT32_MAX : INH + 2;
        }
        if (s_lo <= s_hi) {
            const int aa0 = (row && m >= 1 && m - 1 < P.a_len) ? acod[m - 1] : AMB;       // the residue of my row, and the next one
            const int aa1 = row ? (m < P.a_len ? acod[m] : P.a_pad) : AMB;                  // a_pad: what the Seq holds behind the query
            const int* prof0 = T.mtx + aa0 * 32;
            const int* prof1 = T.mtx + aa1 * 32;
            // Cip_score::cip_score(3 m - phs) for the three phases (sigB[phs], src/fwd2h1.cc:352-354)
            const bool has_cip = row && A.cip && P.cip_off >= 0;
            const int cipm = has_cip ? A.cip[P.cip_off + 3 * m + 1] : 0, cip0 = has_cip ? A.cip[P.cip_off + 3 * m] : 0,
                      cipp = has_cip ? A.cip[P.cip_off + 3 * m - 1] : 0;
            // the insertion queue: ea is the slot of the current column's frame
            St ea = black, eb = black, ec = black;
            St e2a = black, e2b = black, e2c = black;               // the second insertion queue (Noll = 3)
            Cands<MODE> cl[3];
#pragma unroll
            for (int ph = 0; ph < 3; ++ph) cl[ph].clear();
            bool seeded = false;                                    // a global query start: the corner seeds the third slot

            auto need_lo = [&](int S) { return S - 4 * (m0 + 63) - 3 - lw + 3; };
            auto need_hi = [&](int S) { return S - 4 * m0 + 3 - lw + 3; };
            int res_lo = max(0, need_lo(s_lo)), res_hi = res_lo;
            int cres = INT32_MIN;                                   // columns below this one are staged
            auto refill = [&](int S) {
                const int dead = min(max(0, need_lo(S)), W);
                for (int e = res_lo + lane; e < dead; e += 64) {
                    const int q = e & (RING - 1);
#pragma unroll
                    for (int a = 0; a < NP * NF; ++a) gst<PIPE>(G(a) + e, L[a][q]);
                }
                res_lo = max(res_lo, dead);
                const int want = min(W, need_hi(S + CHUNK - 1) + 1);
                if (PIPE) {
                    // what I am about to read, the tile above must have handed back: its need_lo(S') >= want
                    if (ti > 0 && !stalled) stalled = !wait_for(ti - 1, want + 4 * (m0 - TH + 63) + lw + 1);
                    publish(ti, S + 1);
                }
                for (int e = max(res_hi, res_lo) + lane; e < want; e += 64) {
                    const int q = e & (RING - 1);
                    const int* src[NP * NF]; int val[NP * NF];             // (all planes in flight together: spdp_pipe.h)
#pragma unroll
                    for (int a = 0; a < NP * NF; ++a) src[a] = G(a);
                    gld_n<PIPE>(src, e, val);
#pragma unroll
                    for (int a = 0; a < NP * NF; ++a) L[a][q] = val[a];
                }
                res_hi = max(res_hi, want);
                // columns [S - (m0 + 63) - 3, S + CHUNK - 1 - m0 + 5] of the next CHUNK steps (CUT: their real positions,
                // staged at XC(.): at most 16 slots more)
                const int c_hi = XC(RV(S + CHUNK - 1 - m0) + 5);
                for (int x = max(cres, XC(RV(S - (m0 + 63)) - 3)) + lane; x <= c_hi; x += 64) {
                    const int c = CX(x);
                    const bool in = c >= 0 && c <= P.b_len + 2;
                    Cc[x & (CRING - 1)] = in ? cols[c] : make_int4(0, 0, 0, 0);
                    Ca[x & (CRING - 1)] = in ? aux[c] : make_short4(0, 0, 0, 0);
                }
                cres = c_hi + 1;
                WAVE_SYNC();
            };
            auto COL = [&](int c) -> int4 { return Cc[XC(c) & (CRING - 1)]; };
            auto AUX = [&](int c) -> short4 { return Ca[XC(c) & (CRING - 1)]; };
            auto tron_l = [&](int i) -> int { return (i < 0 || i > P.b_len) ? AMB : ((COL(i + 2).x >> 16) & 0xff); };   // tron_at from the staged columns
            auto lds_get = [&](int e, int isF) {
                const int q = e & (RING - 1);
                St s = black;
#pragma unroll
                for (int f = 0; f < NF; ++f) fld(s, f) = L[isF * NF + f][q];
                return s;
            };
            auto lds_put = [&](int e, int isF, St s) {
                const int q = e & (RING - 1);
#pragma unroll
                for (int f = 0; f < NF; ++f) L[isF * NF + f][q] = fld(s, f);
            };

            // CUT: steps up to SA_end sweep columns <= cut_l - 3; the next 3 TH steps are the seam (lane k / 3 on column
            // cut_l - 2 + k % 3); from then on the wave is back on the schedule of a sweep that starts at column cut_l + 1
            const int SA_end = CUT ? cut_l - 3 + m0 + TH - 1 : 0, seam = CUT ? 3 * TH : 0;
            const int S_begin = CUT ? min(s_lo, SA_end + 1) : s_lo;
            const int S2_begin = cut_l + 1 + m0, shiftB = CUT ? SA_end + seam + 1 - S2_begin : 0;
            const int S_end = CUT ? max(s_hi + shiftB, SA_end + seam) : s_hi;
            for (int S = S_begin; S <= S_end; ++S) {
                int v = S - m;
                bool act = true;
                if constexpr (CUT) {
                    if (S <= SA_end) {
                        if (((S - S_begin) & (CHUNK - 1)) == 0) refill(S);
                        act = v <= cut_l - 3;
                    } else if (S <= SA_end + seam) {
                        if (S == SA_end + 1) refill(SA_end);        // (covers what the seam reads: entries and columns of all rows around cut_l)
                        const int k = S - SA_end - 1;
                        v = cut_l - 2 + k % 3;
                        act = lane == k / 3;
                    } else {
                        const int S2 = S - shiftB;
                        if (S2 == S2_begin) {                       // hand the window back and take the one of the new schedule
                            WAVE_SYNC();
                            for (int e = res_lo + lane; e < res_hi; e += 64) {
                                const int q = e & (RING - 1);
#pragma unroll
                                for (int a = 0; a < NP * NF; ++a) gst<PIPE>(G(a) + e, L[a][q]);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                            res_lo = res_hi = max(0, need_lo(S2));
                            cres = INT32_MIN;
                        }
                        if (((S2 - S2_begin) & (CHUNK - 1)) == 0) refill(S2);
                        v = S2 - m;
                        act = v > cut_l;
                    }
                } else if (((S - s_lo) & (CHUNK - 1)) == 0) refill(S);
                const int n = RV(v);
                const bool on = any && act && v >= v0 && v <= v9;
                if (__ballot(on) == 0) continue;
                const int4 col = on ? COL(n) : make_int4(0, 0, 0, 0);
                const int sigE = (on && n > bl && n >= 2) ? (int) AUX(n - 2).z : 0;
                const int r = v - 3 * m, e = r - lw + 3;
                St h = lds_get(e, 0), f = lds_get(e, 1);
                const St hq = h;                                    // the entry as the cell found it
                const St u1 = lds_get(e + 1, 0), u2 = lds_get(e + 2, 0), u3 = lds_get(e + 3, 0), fu = lds_get(e + 3, 1);
                const St l1 = lds_get(e - 1, 0), l2 = lds_get(e - 2, 0), l3 = lds_get(e - 3, 0);
                St f2 = black, fu2 = black;
                if constexpr (DAGP) { f2 = lds_get(e, 2); fu2 = lds_get(e + 3, 2); }
                if (on && !seeded && !b_exgl && m == al) {          // (the queue slot of column n0 + 2)
                    ec = hq;
                    if (UDH) ec.v += gw3; else ec.v = gw3;
                    if constexpr (DAGP) { e2c = hq; if (UDH) e2c.v += gw3l; else e2c.v = gw3l; }
                }
                seeded = seeded || on;
                int mxk = 0;                                        // which state holds the running maximum: 0 H, 1 E, 2 F (3 E2, 4 F2)
                auto val_of = [&](int k) {
                    if constexpr (DAGP) return k == 0 ? h.v : (k == 1 ? ea.v : (k == 2 ? f.v : (k == 3 ? e2a.v : f2.v)));
                    else return k == 0 ? h.v : (k == 1 ? ea.v : f.v);
                };
                // (state k of the cell, read and written: spelled out -- a lambda that takes or returns the records by reference
                //  parks all of them in scratch memory, 0 -> 48 bytes per lane and a0 9.3 -> 5.6 GCUPS when round 5 tried)
#define ST_OF(K) (DAGP ? st_sel5((K), h, ea, f, e2a, f2) : st_sel3((K), h, ea, f))
#define ST_SET(K, S) do { if ((K) == 0) h = (S); else if ((K) == 1) ea = (S); else if (!DAGP || (K) == 2) f = (S); \
                          else if ((K) == 3) e2a = (S); else f2 = (S); } while (0)
                if (m != al) {
                    // match: the codon ending here against my residue
                    if (n < bl + 3) h = black;
                    else {
                        h.v += prof0[(col.x >> 16) & 0xff] + sigE;
                        h.d = (UDH ? (hq.d & T_DIAG) != 0 : is_kind(hq.d, M_DIAG)) ? T_DIAG : T_NEWD;
                    }
                    // deletion: extend, or open after a frame shift of 2 / 1 nt or after a whole codon
                    const int stay = fu.v + gep;
                    const int x2 = u1.v + (is_kind(u1.d, M_VERT) ? ge1 : gw1);
                    const int x1 = u2.v + (is_kind(u2.d, M_VERT) ? ge2 : gw2);
                    const int x0 = u3.v + gw3;
                    St nf = fu; nf.v = stay; nf.d = T_VERT;
                    if (x2 > nf.v) { nf = u1; nf.v = x2; nf.d = T_SLA2; }
                    if (x1 > nf.v) { nf = u2; nf.v = x1; nf.d = T_SLA1; }
                    if (x0 >= nf.v) { nf = u3; nf.v = x0; nf.d = T_VERT; }
                    f = nf;
                    if (UDH ? (f.v >= h.v) : (f.v > h.v)) mxk = 2;
                    if constexpr (DAGP) {                           // long deletion: open from the cell a codon above, or extend
                        const int xl = u3.v + gw3l, yl = fu2.v + lgep;
                        St n2 = fu2; n2.v = yl;
                        if (xl >= yl) { n2 = u3; n2.v = xl; n2.d = T_VERL; }
                        f2 = n2;
                        if (UDH ? (f2.v >= val_of(mxk)) : (f2.v > val_of(mxk))) mxk = 4;
                    }
                }
                if (on) {
                    // insertion: a whole codon (extend or open), 2 nt, 1 nt
                    if (n > n0 + 2) {
                        const int x = l3.v + gw3;
                        ea.v += gep;
                        if (x > ea.v) { ea = l3; ea.v = x; }
                        ea.v += sigE;
                        ea.d = (ea.d & T_SPIN) + T_HORI;
                        if constexpr (DAGP) {                       // long insertion (its place among the maxima: before E's frame shifts)
                            const int xl = l3.v + gw3l;
                            e2a.v += lgep;
                            if (xl > e2a.v) { e2a = l3; e2a.v = xl; }
                            e2a.v += sigE;
                            e2a.d = (e2a.d & T_SPIN) + T_HORL;
                            if (e2a.v > val_of(mxk)) mxk = 3;
                        }
                    }
                    if (n > n0 + 1) {
                        const int x = l2.v + gw2;
                        if (x > ea.v) { ea = l2; ea.v = x; ea.d = UDH ? T_HOR2 : (ea.d & T_SPIN) + T_HOR2; }
                    }
                    const int x = l1.v + gw1;
                    if (x > ea.v) { ea = l1; ea.v = x; ea.d = UDH ? T_HOR1 : (ea.d & T_SPIN) + T_HOR1; }
                    if (ea.v > val_of(mxk)) mxk = 1;
                }
                const unsigned fl = (unsigned) col.x >> 24;
                // MODE 2 reads `rlst` of the NEXT queue slot
                const int qslot = on ? (n - n0 + 1) % 3 : 0;
                auto rl_get = [&]() { return qslot == 0 ? rl0 : (qslot == 1 ? rl1 : rl2); };
                auto rl_set = [&](int v) { if (qslot == 0) rl0 = v; else if (qslot == 1) rl1 = v; else rl2 = v; };

                // ---- acceptor: the candidates of this column's phase(s) may raise the state they left from
                bool spj3 = false;
                const bool acc = on && spj && (fl & 3);
                if (__ballot(acc)) {
#pragma unroll 1
                    for (int pass = 0; pass < 2; ++pass) {
                        const int ph0 = (int) (fl & 3) - 2;
                        const bool t = acc && (pass == 0 || ((fl & 4) && ph0 == -1));
                        if (!__ballot(t)) break;
                        const int phs = pass == 0 ? ph0 : 1;
                        const int s3 = pass == 0 ? (int) (short) (col.y & 0xffff) : (int) (short) ((unsigned) col.y >> 16);
                        const int nb = n - phs;
                        int dn3 = 0, w2 = 7, w3 = 7, fix = 0;
                        if (t) {
                            dn3 = COL(nb).w & 15;
                            if (phs) {                              // the two bases after the acceptor, for a codon the intron splits
                                const int t2 = tron_l(nb), t3 = tron_l(nb + 1);
                                w2 = t2 < 32 ? T.mid[t2] : 7; w3 = t3 < 32 ? T.mid[t3] : 7;
                                if (nb >= br) w2 = 7;
                                if (phs == -1) fix = prof1[tron_l(n + 1)] + AUX(n + 1).z;     // what the next row's match will add anyway
                            }
                        }
                        int sel[5] = {-1, -1, -1, -1, -1};
                        // the list of my column's phase, picked once (round 4: the candidate loop used to be written out three
                        // times, once per phase, and a wave ran all three whenever its lanes sat on columns of different
                        // phases -- the acceptor was 48 % of the step's cycles)
                        const int phi = phs + 1;
                        int tv[NC], tj[NC], tx[NC];
#pragma unroll
                        for (int l = 0; l < NC; ++l) {
                            tv[l] = phi == 0 ? cl[0].v[l] : (phi == 1 ? cl[1].v[l] : cl[2].v[l]);
                            tj[l] = phi == 0 ? cl[0].j[l] : (phi == 1 ? cl[1].j[l] : cl[2].j[l]);
                            tx[l] = phi == 0 ? cl[0].x[l] : (phi == 1 ? cl[1].x[l] : cl[2].x[l]);
                        }
                        const int tn = phi == 0 ? cl[0].n : (phi == 1 ? cl[1].n : cl[2].n);
                        const int cipv = phs < 0 ? cipm : (phs == 0 ? cip0 : cipp);
                        // screen: the best candidate of the list, priced as high as anything can be, against the lowest of the
                        // three states it may raise (every update below is behind `x > state`)
                        const int lowest = DAGP ? min(min(h.v, min(ea.v, f.v)), min(e2a.v, f2.v)) : min(h.v, min(ea.v, f.v));
                        const bool tq = t && tn >= 0 && tv[0] + cipv + s3 + T.gain[0] + T.gain[1] + T.gain[2] + abs(fix) > lowest;
                        if (__ballot(tq)) {
#pragma unroll
                            for (int l = 0; l < NC; ++l) {
                                if (!(tq && l <= tn)) continue;
                                const int cd = DAGP ? ((tx[l] & 3) | ((tx[l] >> 12) & 1) << 2) : (tx[l] & 3);
                                if (phs == 1 && cd == 2) continue;
                                if (nb - tj[l] < minl) continue;
                                int x = tv[l] + cipv + intpen_of(nb - tj[l]) + s3 + T.t53[16 * ((tx[l] >> 2) & 15) + dn3];
                                if (cd == 0 && phs) {
                                    const int w0 = (tx[l] >> 6) & 7, w1 = (tx[l] >> 9) & 7;
                                    // a codon is defined when its own three bases are (spj_amb_tron_tab / spj_tron_amb_tab:
                                    // an ambiguous first or last base of the four leaves the other codon standing)
                                    if (phs == 1) x += prof0[(w0 < 4 && w1 < 4 && w2 < 4) ? T.tron_of[16 * w0 + 4 * w1 + w2] : AMB];
                                    else x += prof1[(w1 < 4 && w2 < 4 && w3 < 4) ? T.tron_of[16 * w1 + 4 * w2 + w3] : AMB] - fix;
                                }
                                if (cd == 0) { if (x > h.v) { h.v = x; sel[0] = l; } }
                                else if (cd == 1) { if (x > ea.v) { ea.v = x; sel[1] = l; } }
                                else if (!DAGP || cd == 2) { if (x > f.v) { f.v = x; sel[2] = l; } }
                                else if (cd == 3) { if (x > e2a.v) { e2a.v = x; sel[3] = l; } }
                                else { if (x > f2.v) { f2.v = x; sel[4] = l; } }
                            }
                        }
                        // the winners, in the order H, E, F (, E2, F2)
                        int maxk = NODK;
                        int lnk[5] = {EOU, EOU, EOU, EOU, EOU};
#pragma unroll
                        for (int k = 0; k < NODK; ++k) {
                            const bool w = sel[k] >= 0;
                            if (!__ballot(w)) continue;
                            int cj = pick(sel[k], tj), ca = 0, cb = 0, cc = 0, ce = 0;
#pragma unroll
                            for (int ph = 0; ph < 3; ++ph)
                                if (phi == ph) { ca = pick(sel[k], cl[ph].a);
                                                 if (UDH) { cb = pick(sel[k], cl[ph].b); cc = pick(sel[k], cl[ph].c); ce = pick(sel[k], cl[ph].e); } }
                            St to = ST_OF(k);
                            const int p1 = vadd(w, m, cj + phs, ca);
                            const int p2 = vadd(w, m, n, p1);
                            if (w) {
                                to.d = (k == 0 ? T_DIAG : (k == 1 ? T_HORI : (k == 2 ? T_VERT : (k == 3 ? T_HORL : T_VERL)))) | T_SPIN;    // nod2dir
                                if (FWD) to.a = p2;
                                if (UDH) { to.a = max(ca, r); to.b = min(cb, r); to.c = cc; to.e = ce; lnk[k] = ce; }
                                ST_SET(k, to);
                                if (UDH ? (to.v >= val_of(mxk)) : (to.v > val_of(mxk))) { mxk = k; maxk = k; }
                            }
                        }
                        if (UDH && is_imd && t && maxk < NODK) {
                            gst<PIPE>(IM(iq, HLNK, 0, r), lnk[maxk]);
                            rl_set(r);
                            { St w_ = ST_OF(maxk); w_.e = r; ST_SET(maxk, w_); }
                            spj3 = true;
                            if (maxk == 0) {
                                if (sel[1] >= 0 && ea.v > h.v + gop) { ea.e = r + width; gst<PIPE>(IM(iq, HLNK, 1, r), lnk[1]); }
                                if (sel[2] >= 0 && f.v > h.v + gop) f.e = r + width;
                                if constexpr (DAGP) {               // (c = 2, d = 3: the long pair against GOP[2])
                                    if (sel[3] >= 0 && e2a.v > h.v + lgop) { e2a.e = r + 2 * width; gst<PIPE>(IM(iq, HLNK, 2, r), lnk[3]); }
                                    if (sel[4] >= 0 && f2.v > h.v + lgop) f2.e = r + 2 * width;
                                }
                            }
                        }
                    }
                }

                // ---- the cell takes the best state
                const int y = h.v;
                St mxs = ST_OF(mxk);
                if (FWD || MODE == 0) {
                    bool opened = false;
                    if (mxk != 0) h = mxs;
                    else if (Local && y > hq.v) {
                        if (LocalL && hq.d == 0 && !(h.d & T_SPIN)) opened = true;
                        else if (LocalR && on && y > best.v) { best = h; best_m = m; best_n = n; }
                    }
                    const int p_open = vadd(on && opened, m - 1, n - 3, 0);
                    if (FWD && on && opened) h.a = p_open;
                    const bool reset = LocalL && h.v <= 0;
                    if (reset) { h.v = 0; h.d = 0; }
                    const bool turn = on && !reset && h.d == T_NEWD;
                    const int p_turn = vadd(turn, m - 1, n - 3, h.a);
                    if (FWD && turn) h.a = p_turn;
                } else {
                    if (mxk == 0) {
                        if (LocalR && on && y > best.v) { best = h; best_m = m; best_n = n; }
                    } else {
                        if (mxs.a < r) mxs.a = r;
                        if (mxs.b > r) mxs.b = r;
                        ST_SET(mxk, mxs);
                        h = mxs;
                    }
                    if (LocalL && h.v <= 0) { h.v = 0; h.d = 0; h.c = m; h.e = h.a = h.b = r; }
                }
                if (mxk == 0) mxs = h;                              // (`mx` is the entry itself: it sees the resets)

                // ---- donor: the states of this cell enter the candidate list(s) of the phase(s) this column can start
                const int hd = node_of(mxs.d);
                const bool don = on && spj && ((fl >> 3) & 3);
                if (__ballot(don)) {
#pragma unroll 1
                    for (int pass = 0; pass < 2; ++pass) {
                        const int ph0 = (int) ((fl >> 3) & 3) - 2;
                        const bool t = don && (pass == 0 || ((fl & 32) && ph0 == -1));
                        if (!__ballot(t)) break;
                        const int phs = pass == 0 ? ph0 : 1;
                        const int nb = n - phs;
                        int sigJ = 0, packed = 0;
                        if (t) {
                            sigJ = AUX(nb).w;
                            const int t0 = tron_l(nb - 2), t1 = tron_l(nb - 1);
                            int w0 = t0 < 32 ? T.mid[t0] : 7, w1 = t1 < 32 ? T.mid[t1] : 7;
                            if (nb < P.b_left) w0 = w1 = 7;            // (outside the range: neither codon)
                            packed = ((COL(nb).w >> 4) & 15) << 2 | (w0 & 7) << 6 | (w1 & 7) << 9;
                        }
#pragma unroll
                        for (int k = 0; k < NODK; ++k) {
                            const bool cross = phs == 1 && k == 0;          // the intron cuts the codon of the cell above-left
                            const St own = ST_OF(k);
                            const St src = st_sel(cross, hq, own);
                            bool tk = t && k >= ((hd == 0 || phs == 1) ? 0 : 1) && src.d && !(src.d & T_SPIN);    // (no orphan exon)
                            if (tk && !cross && k != hd && hd >= 0) {
                                int z = mxs.v;
                                if (hd == 0 || ((k - hd) & 1)) z += (k == 2 || k == 3) ? gop : (k == 4 ? lgop : 0);   // GOP[k / 2]
                                if (src.v <= z) tk = false;         // cannot become the better path
                            }
                            {   // a full list whose last kept entry beats the newcomer: the free slot stays below it and the list is
                                // NC - 1 long afterwards (what Cands::insert does with it, without the insertion code)
                                const int phi = phs + 1;
                                const int ln = phi == 0 ? cl[0].n : (phi == 1 ? cl[1].n : cl[2].n);
                                const int vl = phi == 0 ? cl[0].v[NC - 2] : (phi == 1 ? cl[1].v[NC - 2] : cl[2].v[NC - 2]);
                                const bool weak = tk && ln >= NC - 2 && vl > src.v + sigJ;
                                if (weak) { if (phi == 0) cl[0].n = NC - 2; else if (phi == 1) cl[1].n = NC - 2; else cl[2].n = NC - 2; }
                                tk = tk && !weak;
                            }
                            if (!__ballot(tk)) continue;
#pragma unroll
                            for (int ph = 0; ph < 3; ++ph) {
                                const bool tt = tk && phs + 1 == ph;
                                if (!__ballot(tt)) continue;
                                const bool kept = cl[ph].insert(tt, src.v + sigJ, nb, packed | (k & 3) | (k >> 2) << 12, src, is_imd ? r : src.e);
                                // an intermediate row links an insertion that splices out to the last match of its frame
                                if (UDH && is_imd && kept && k == 1) gst<PIPE>(IM(iq, HLNK, 0, r), rl_get());
                            }
                        }
                    }
                }

                // ---- an intermediate row records where the paths cross it and restarts ranges and links
                if (UDH && is_imd && on) {
                    if (hd == 0) rl_set(r);
                    else if (!spj3 && (hd & 1)) gst<PIPE>(IM(iq, HLNK, 0, r), rl_get());
                    gst<PIPE>(IM(iq, VLNK, 0, r), h.e); gst<PIPE>(IM(iq, LWRB, 0, r), min(r, h.b)); gst<PIPE>(IM(iq, UPRB, 0, r), max(r, h.a));
                    h.b = h.a = r; h.e = r;
                    gst<PIPE>(IM(iq, VLNK, 1, r), f.e); gst<PIPE>(IM(iq, LWRB, 1, r), min(r, f.b)); gst<PIPE>(IM(iq, UPRB, 1, r), max(r, f.a));
                    f.b = f.a = r; f.e = r + width;
                    if constexpr (DAGP) {
                        gst<PIPE>(IM(iq, VLNK, 2, r), f2.e); gst<PIPE>(IM(iq, LWRB, 2, r), min(r, f2.b)); gst<PIPE>(IM(iq, UPRB, 2, r), max(r, f2.a));
                        f2.b = f2.a = r; f2.e = r + 2 * width;
                    }
                }
                if (on) {
                    lds_put(e, 0, h);
                    lds_put(e, 1, f);
                    const St t = ea; ea = eb; eb = ec; ec = t;      // the next column is the next frame
                    if constexpr (DAGP) { lds_put(e, 2, f2); const St t2 = e2a; e2a = e2b; e2b = e2c; e2c = t2; }
                }
                if constexpr (CUT) {
                    if (on && v == cut_l) {                         // the insertion states run on over the cut (ea: the next column's frame)
                        const int lg = gep * cutlen / 3;
                        ea.v += lg; eb.v += lg; ec.v += lg;
                        if constexpr (DAGP) {                       // (*h = dagp ? e2 : e1, src/fwd2h1.cc:595-598)
                            const int lg2 = lgep * cutlen / 3;
                            e2a.v += lg2; e2b.v += lg2; e2c.v += lg2;
                            lds_put(e - 2, 0, e2b); lds_put(e - 1, 0, e2c); lds_put(e, 0, e2a);
                            lds_put(e - 2, 2, black); lds_put(e - 1, 2, black); lds_put(e, 2, black);
                        } else { lds_put(e - 2, 0, eb); lds_put(e - 1, 0, ec); lds_put(e, 0, ea); }
                        lds_put(e - 2, 1, black); lds_put(e - 1, 1, black); lds_put(e, 1, black);
                    }
                }
            }
            WAVE_SYNC();
            for (int e = res_lo + lane; e < res_hi; e += 64) {
                const int q = e & (RING - 1);
#pragma unroll
                for (int a = 0; a < NP * NF; ++a) gst<PIPE>(G(a) + e, L[a][q]);
            }
        }
        if (UDH && PIPE) {
            if (is_imd) { gst<true>(rlf + 3 * iq, rl0); gst<true>(rlf + 3 * iq + 1, rl1); gst<true>(rlf + 3 * iq + 2, rl2); }
        } else if (UDH && imd_mask) {
            const int src = __ffsll((long long) imd_mask) - 1;
            rl0 = __shfl(rl0, src); rl1 = __shfl(rl1, src); rl2 = __shfl(rl2, src);
        }
    }
    auto reduce_best = [&]() {                                      // first maximum in row-major order over the lanes
        for (int off = 32; off; off >>= 1) {
            St o;
#pragma unroll
            for (int a = 0; a < 6; ++a) fld(o, a) = __shfl_xor(fld(best, a), off);
            const int om = __shfl_xor(best_m, off), on_ = __shfl_xor(best_n, off);
            if (o.v > best.v || (o.v == best.v && (om < best_m || (om == best_m && on_ < best_n)))) { best = o; best_m = om; best_n = on_; }
        }
    };
    if (PIPE) {
        // finished = every entry holds what the tiles up to this one leave: the tile above must be finished too
        const int ti = t_lo;
        if (LocalR) {
            reduce_best();
            if (lane == 0) {
                int* b = tbest + 8 * ti;
#pragma unroll
                for (int a = 0; a < 6; ++a) gst<true>(b + a, fld(best, a));
                gst<true>(b + 6, best_m); gst<true>(b + 7, best_n);
            }
        }
        if (FWD && __any(vover) && lane == 0) gst<true>(sy + 1, 1);
        if (ti > 0 && !stalled) stalled = !wait_for(ti - 1, INT32_MAX);
        publish(ti, INT32_MAX);
        if (ti != n_tiles - 1) return;
        // the wave of the last tile ends the problem
        if (FWD) vover = __hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (LocalR) {                                               // tiles in row order: the first maximum wins
            St bb = black; bb.v = NEV; bb.c = al;
            int bm = UDH ? ar : al, bn = UDH ? br : bl;
            for (int t = lane; t < n_tiles; t += 64) {
                const int* b = tbest + 8 * t;
                if (gld<true>(b) > bb.v) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) fld(bb, a) = gld<true>(b + a);
                    bm = gld<true>(b + 6); bn = gld<true>(b + 7);
                }
            }
            best = bb; best_m = bm; best_n = bn;
        }
    } else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    if (UDH) { for (int i = lane; i < 10 * (n_im + 1); i += 64) cpos[i] = EOU; STORES_DRAINED(); }

    // ---- the end of the alignment: the tracked local maximum, or lastH_ng / hlastH_ng on the last row
    // staging for the sequential relaxations: entries [e0, e0 + cnt) of H through the LDS arrays, cnt <= RING
    auto stage_in = [&](int e0, int cnt) {
        for (int i = lane; i < cnt; i += 64)
#pragma unroll
            for (int a = 0; a < NF; ++a) L[a][i] = gld<PIPE>(G(a) + e0 + i);
        WAVE_SYNC();
    };
    auto stage_out = [&](int e0, int from, int cnt) {
        WAVE_SYNC();
        for (int i = from + lane; i < cnt; i += 64)
#pragma unroll
            for (int a = 0; a < NF; ++a) gst<PIPE>(G(a) + e0 + i, L[a][i]);
        if (PIPE) STORES_DRAINED(); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    };
    auto at = [&](int i) { St s = black;
#pragma unroll
        for (int a = 0; a < NF; ++a) fld(s, a) = L[a][i];
        return s; };
    auto set = [&](int i, St s) {
#pragma unroll
        for (int a = 0; a < NF; ++a) L[a][i] = fld(s, a); };
    auto gload = [&](int r, int isF) { St s = black;
#pragma unroll
        for (int a = 0; a < NF; ++a) fld(s, a) = gld<PIPE>(G(isF * NF + a) + (r - lw + 3));
        return s; };

    St fin = black;                                                 // the state the alignment ends in
    int fin_r = br - 3 * ar;
    int ptr = 0;
    if (LocalR) reduce_best();
    const bool by_last_row = UDH ? !LocalR : (!LocalR || best_m == ar);
    if (by_last_row) {
        const int m3 = 3 * ar, r9 = br - cutlen - m3;               // (CUT: both ends global, only the entry of br matters)
        const int r_in = max(lw, bl - m3);                          // first entry of the last row
        int mx_r = r9, mx_v = gload(r9, 0).v;
        if (a_exgr) {
            // trailing gap on the genomic side: per frame, every entry may be reached from the one a codon below it --
            // sequential (each entry sees its predecessor as already relaxed), one lane over staged chunks
            int glen0 = 0, glen1 = 0, glen2 = 0;
            constexpr int CH = RING - 3;
            for (int c0 = r_in; c0 <= r9; c0 += CH) {
                const int cnt = min(CH, r9 - c0 + 1);
                const int back = c0 - r_in >= 3 ? 3 : 0;            // relaxed entries below the chunk come along
                stage_in(c0 - back - lw + 3, cnt + back);
                for (int i = 0; i < cnt; ++i) {
                    const int r = c0 + i, t = r - r_in, ph = t % 3, n = r + m3;
                    bool rec = false; int rec_p = 0;
                    if (lane == 0) {
                        int& glen = ph == 0 ? glen0 : (ph == 1 ? glen1 : glen2);
                        glen += 3;
                        St h = at(i + back);
                        int c1 = NEV, c2 = NEV;
                        St below = black;
                        if (t >= 3) {
                            below = at(i + back - 3);
                            if (below.d != T_DEAD) {
                                c1 = below.v + aux[n - 2].z;
                                if (!(a_exgr & 2)) c1 += gext3(glen);
                                if (!(a_exgr & 1) && glen == 3) c1 += gop;
                                const bool stop_ok = UDH ? sc->term_codon != 0 : aux[n - 2].y > 0;
                                if (stop_ok && !(h.d & T_SPIN)) c2 = below.v + aux[n - 2].y;
                            }
                        }
                        const int s5 = (Local && aux[n].w > 0) ? aux[n].w : 0;
                        const int c0v = h.v + s5;
                        c1 += s5;
                        if (c2 > c0v && c2 > c1) {                  // a stop codon ends the alignment here
                            h = below; h.d = T_DEAD; h.v = c2;
                            if (UDH) h.a = max(r, h.a);
                            if (FWD && r != mx_r && h.v > mx_v) { rec = true; rec_p = h.a; }
                        } else if (c1 > c0v) { h = below; h.d = T_HORI; h.v = c1 - s5; }
                        else if (!is_kind(h.d, M_HORI)) glen = 0;
                        set(i + back, h);
                    }
                    const int p = vadd(rec, ar, r + m3 - 3, rec_p);
                    if (lane == 0) {
                        St h = at(i + back);
                        if (rec) { h.a = p; set(i + back, h); }
                        if (r == mx_r) mx_v = h.v;
                        else if (h.v > mx_v) { mx_r = r; mx_v = h.v; }
                    }
                }
                stage_out(c0 - back - lw + 3, back, cnt + back);
            }
        } else {
            // a global genomic end: only a stop codon may follow the last residue
            if (lane == 0) {
                const St below = gload(r9 - 3, 0);
                St h9 = gload(r9, 0);
                const int y = below.v + aux[br - 2].y;
                if (y > h9.v) {
                    h9 = below; h9.v = y; h9.d = T_HORI;
                    if (UDH) h9.a = max(r9, h9.a);
                    put(r9 - lw + 3, 0, h9);
                    mx_v = y;
                }
            }
            if (PIPE) STORES_DRAINED(); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        }
        mx_r = __shfl(mx_r, 0); mx_v = __shfl(mx_v, 0);
        bool from_f = false;
        if (b_exgr == 1) {
            const int r_top = min(up, br - 3 * al);
            if (UDH) {
                // trailing gap on the query side, MODE 2: a frame-shifted end pays once -- the first best in descending order
                int bv = mx_v, bk = INT32_MAX;
                for (int i = lane; i < r_top - r9; i += 64) {
                    const int r = r_top - i;
                    const int x = gload(r, 0).v + ((r % 3) ? A.extragop : 0);
                    if (x > bv) { bv = x; bk = i; }
                }
                for (int off = 32; off; off >>= 1) {
                    const int ov = __shfl_xor(bv, off), ok = __shfl_xor(bk, off);
                    if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
                }
                if (bk != INT32_MAX) { mx_r = r_top - bk; mx_v = bv; }
            } else {
                // MODE 0 / 1: per frame a running gap from the entries above, closed wherever an entry beats it;
                // sequential downwards (it rewrites the values it later reads), one lane over staged chunks
                int g0v = NEV, g1v = NEV, g2v = NEV;
                constexpr int CH = RING - 3;
                for (int c1 = r_top - 3; c1 >= r9; c1 -= CH) {
                    const int cnt = min(CH, c1 - r9 + 1);
                    const int c0 = c1 - cnt + 1;
                    stage_in(c0 - lw + 3, cnt + 3);
                    if (lane == 0) {
                        for (int i = cnt - 1; i >= 0; --i) {
                            const int r = c0 + i, ph = (r_top - 3 - r) % 3;
                            int& g = ph == 0 ? g0v : (ph == 1 ? g1v : g2v);
                            int x = L[0][i + 3];
                            if (!(b_exgr & 1)) x += gop;
                            if (x > g) g = x;
                            if (!(b_exgr & 2)) g += gep;
                            if (L[0][i] > g) g = NEV;
                            else if (g > mx_v) { mx_r = r; mx_v = g; L[0][i] = g; }
                        }
                    }
                    stage_out(c0 - lw + 3, 0, cnt);
                }
                mx_r = __shfl(mx_r, 0); mx_v = __shfl(mx_v, 0);
            }
        } else if (b_exgr == 2) {
            from_f = true; mx_r = r9;
        }
        fin = gload(mx_r, from_f ? 1 : 0);
        if (!from_f) fin.v = mx_v;
        fin_r = from_f ? mx_r + width : mx_r;                        // (the reference measures an F entry from the H array)
        if (FWD) {
            int rf = ar, rw = br;
            if (!from_f) {
                int pp = mx_r - r9;
                if (pp > 0) { rf -= (pp + 2) / 3; if (pp %= 3) rw -= 3 - pp; }
                else if (pp < 0) rw += pp;
            }
            ptr = __shfl(vadd(lane == 0, rf, rw, fin.a), 0);
        }
    } else {
        fin = best;
        if (FWD) ptr = __shfl(vadd(lane == 0, best_m, best_n, best.a), 0);
    }

    if (!UDH) {
        DevResultH R;
        R.score = fin.v; R.mr = ar; R.nr = br; R.maxt = 0; R.maxr = 0; R.pad[0] = R.pad[1] = R.pad[2] = 0;
        if (!FWD) { if (lane == 0) A.res[pi] = R; return; }
        int vtotal = vcount;                                        // numbers handed out
        if (PIPE) { STORES_DRAINED(); vtotal = 1 + __hip_atomic_load(sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        const bool over_any = __any(vover);
        if (lane == 0) {
            // Vmf::traceback(ptr) + the boundary record of trcbkalignH_ng
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
                    if (sv.z < 0 || sv.z >= vtotal || cnt > vtotal) { status = -2; break; }
                    sv = rec(sv.z);
                }
                const int rd = Local ? 0 : ((ln - 3 * lm) - bl + 3 * al);
                if (rd) {
                    const int2 rec = rd > 0 ? make_int2(al, bl + rd) : make_int2(al - rd / 3, bl);
                    if (cnt < A.skl_cap) out[cnt] = rec; else status = -1;
                    ++cnt;
                }
            }
            A.n_skl[pi] = status ? status : cnt;
            A.res[pi] = R;
        }
        return;
    }

    // ---- MODE 2: the end cell, then the links back through the intermediate rows (one cpos row each)
    if (lane != 0) return;
    int flag = 0, score = fin.v;
    if (LocalR) {
        int i = n_im;
        while (--i >= 0 && mi_of(i) > ar) ;
        ar = best_m; br = best_n;
        if (i < 0) i = 0;
        CPOS(i, 8) = fin.b;
        CPOS(i, 9) = fin.a;
    } else {
        const int rr = br - 3 * ar;
        if (b_exgr && rr < fin_r) ar = (br - fin_r) / 3;
        if (a_exgr && rr > fin_r) br = 3 * ar + fin_r;
    }
    // a link that stands for "rlst of this queue slot as the intermediate rows above left it"
    auto hlnk = [&](int ii, int d, int rr_) {
        int v = gld<PIPE>(IM(ii, HLNK, d, rr_));
        if (PIPE && v >= INH && v < INH + 3) {
            const int slot = v - INH;
            v = INT32_MAX;
            for (int j = ii - 1; j >= 0; --j) { const int w = gld<true>(rlf + 3 * j + slot); if (w != INH + slot) { v = w; break; } }
        }
        return v;
    };
    int i = n_im;
    while (--i >= 0 && mi_of(i) > ar) ;
    if (i < 0 && mi_of(0) > ar) CPOS(0, 2) = br;
    int r = br - 3 * ar;
    CPOS(i + 1, 8) = min(fin.b, r);
    CPOS(i + 1, 9) = max(fin.a, r);
    r = fin.e;
    for ( ; i >= 0 && mi_of(i) > fin.c; --i) {
        int c = 0, d = 0;
        for ( ; r > up; r -= width) ++d;                            // links into the F array carry + width
        if (d > NOLL - 1 || r < lw - 1) { flag = -3; break; }       // outside the link arrays (undefined in the reference)
        const int mi = mi_of(i);
        if (gld<PIPE>(IM(i, VLNK, d, r)) < EOU) {
            CPOS(i, c++) = mi;
            CPOS(i, c++) = (d > 0) ? 1 : 0;
            const int mm3 = 3 * mi;
            for (int rp = hlnk(i, d, r); lw <= rp && rp < up && r != rp; rp = hlnk(i, 0, r = rp)) {
                if (c >= 6) { flag = -3; break; }
                CPOS(i, c++) = r + mm3;
            }
            if (flag) break;
            CPOS(i, c++) = r + mm3;
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
        if (LocalL) { al = fin.c; bl = r + 3 * fin.c; }
        else {
            const int rl = bl - 3 * al;
            if (b_exgl && rl > r) {
                al = (bl - r) / 3;
                for (int j = 0; j < n_im && mi_of(j) < al; ++j) CPOS(j, 0) = EOU;
            }
            if (a_exgl && rl < r) bl = 3 * al + r;
        }
        ++i;
        if ((i < n_im && mi_of(i) < al) || CPOS(i, 2) < bl) score = NEV;
        else if (CPOS(i, 8) == EOU || CPOS(i, 9) == EOU) flag = -3;
        else {
            r = bl - 3 * al;
            CPOS(i, 8) = min(r, CPOS(i, 8));
            CPOS(i, 9) = max(r, CPOS(i, 9));
        }
    }
#undef CPOS
    A.scores[pi] = score;
    A.ranges[4 * pi] = al; A.ranges[4 * pi + 1] = ar; A.ranges[4 * pi + 2] = bl; A.ranges[4 * pi + 3] = br;
    DevResultH R;
    R.score = score; R.mr = ar; R.nr = br; R.maxt = 0; R.maxr = 0; R.pad[0] = flag; R.pad[1] = R.pad[2] = 0;
    A.res[pi] = R;
}

extern "C" hipError_t spdh_launch_scalar(int forward, const HScalarArgs* a, hipStream_t stream)
{
    HScalarArgs A = *a;
    const dim3 blk(64 * WPB);
    if (A.pipe) {                                       // one wave per (problem, tile)
        const dim3 grd((A.n_items + WPB - 1) / WPB);
        if (forward) hipLaunchKernelGGL((spdh_rowwave<1, true>), grd, blk, 0, stream, A);
        else hipLaunchKernelGGL((spdh_rowwave<0, true>), grd, blk, 0, stream, A);
        return hipGetLastError();
    }
    if (A.noll == 3) {                                  // double affine gaps: one wave per problem, two problems per block
        const dim3 g2((A.n_probs + 1) / 2), b2(128);
        if (forward == 2) hipLaunchKernelGGL((spdh_rowwave<1, false, true, true>), g2, b2, 0, stream, A);
        else if (forward) hipLaunchKernelGGL((spdh_rowwave<1, false, false, true>), g2, b2, 0, stream, A);
        else hipLaunchKernelGGL((spdh_rowwave<0, false, false, true>), g2, b2, 0, stream, A);
        return hipGetLastError();
    }
    const dim3 grd((A.n_probs + WPB - 1) / WPB);
    if (forward == 2) hipLaunchKernelGGL((spdh_rowwave<1, false, true>), grd, blk, 0, stream, A);      // problems with a cut range
    else if (forward) hipLaunchKernelGGL((spdh_rowwave<1, false>), grd, blk, 0, stream, A);
    else hipLaunchKernelGGL((spdh_rowwave<0, false>), grd, blk, 0, stream, A);
    return hipGetLastError();
}

extern "C" hipError_t spdh_launch_scalar_udh(const HScalarArgs* a, hipStream_t stream)
{
    HScalarArgs A = *a;
    if (A.noll == 3) { hipLaunchKernelGGL((spdh_rowwave<2, false, false, true>), dim3((A.n_probs + 1) / 2), dim3(128), 0, stream, A); return hipGetLastError(); }
    if (A.pipe) hipLaunchKernelGGL((spdh_rowwave<2, true>), dim3((A.n_items + WPB - 1) / WPB), dim3(64 * WPB), 0, stream, A);
    else hipLaunchKernelGGL((spdh_rowwave<2, false>), dim3((A.n_probs + WPB - 1) / WPB), dim3(64 * WPB), 0, stream, A);
    return hipGetLastError();
}
