// spdp_hsp.hip -- the word-seeded HSP search of the block search on the device: one (query, candidate region) per wave.
//
// What is computed: the HSPs Wilip's scan finds at level -1 (ogotoh/spaln v3.0.7 src/wln.cc: Wlp::foldseq / lookup :253-320, dmsnno /
// dmsnno31 / scan_b :554-678, enter :471-500, eval / reeval :358-469), i.e. what FindHsp (src/blksrc.cc:2346) looks at on the
// region of a candidate block pair: words of the level's reduced alphabet shared by query and region, grouped by diagonal, a
// diagonal's words turned into seed segments by a gap-tolerant score, every segment stretched while the classes agree and trimmed
// to its best-scoring window.  The chaining of the HSPs into gene candidates follows on the host (spdp_hsp_chain.h).
//
// The reference streams the region once and keeps a rolling array of diagonal states, its query index a linked list per word.
// Here the same decisions are reached by sorting instead of streaming, which is what a wave can do well:
//   1. the query's words go into an LDS hash (word -> chain of positions), 64 positions at a time;
//   2. the region is cut from the RESIDENT genome where it lies (other strand and translation applied on the fly -- no copy of
//      the region exists anywhere), 64 positions at a time: each lane spells its word, looks it up, and appends a key
//      (diagonal, query position) per shared word to an LDS list;
//   3. the keys are sorted (bitonic, LDS): a diagonal's words now sit together in query order, which is the order the
//      reference's diagonal state sees them in;
//   4. lanes take diagonals (a lane owns the runs that start in its slice) and run the segment score over them;
//   5. lanes take segments and do the stretch-and-trim pass on the sequences themselves.
// Work is proportional to the number of shared words, not to query x region.  Tasks that do not fit the LDS budget (very long
// queries, regions full of repeats) are flagged and served by the host's form of the same search (spdp_hsp_host.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_hsp_dev.h"

namespace {

typedef unsigned long long u64;
constexpr uint32_t EMPTY = 0xffffffffu;
// nucleotide code -> A C G T as 0 .. 3, anything else >= 4; the first element of an (ambiguous) code; the amino acid a codon with an
// unknown first base is read as, by its middle base (src/seq.cc:31-33, src/utilseq.cc:204-225)
__constant__ uint8_t HSP_NCRED[17] = {15, 15, 0, 1, 4, 2, 5, 6, 10, 3, 7, 8, 10, 9, 12, 13, 14};
__constant__ uint8_t HSP_NCELEM[17] = {0, 0, 0, 1, 2, 2, 0, 2, 0, 3, 3, 3, 1, 1, 2, 3, 0};
__constant__ uint8_t HSP_MOST_ABUNDANT[4] = {14, 3, 10, 13};

__device__ __forceinline__ int lane_id() { return (int) threadIdx.x; }
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// ---- the two sequences as the search reads them -----------------------------------------------------------------------------
struct Region {                         // [0, len) of the region in the orientation of the search; translated when bbt == 3
    const uint8_t* g; int len; bool rvs, tron;
    const uint8_t* tron_of;             // (LDS) codon -> tron code
    __device__ __forceinline__ int nuc(int i) const
    {
        if (i < 0 || i >= len) return 0;                        // (the pads of the reference's sequence object)
        const int c = rvs ? g[len - 1 - i] : g[i];
        if (!rvs) return c > 16 ? 16 : c;
        switch (c) { case 2: return 9; case 9: return 2; case 3: return 5; case 5: return 3; default: return c > 16 ? 16 : c; }
    }
    // Seq::nuc2tron (src/seq.cc:774-798): position p stands for the codon (p - 1, p, p + 1)
    __device__ int at(int p) const
    {
        if (!tron) return nuc(p);
        const int c0 = nuc(p - 1), c1 = nuc(p), c2 = nuc(p + 1);
        if (c1 <= 1) return 1;                                  // a gap in the middle
        const int r1 = HSP_NCRED[c1];
        if (r1 >= 4) return 2;                                  // ambiguous
        const int r0 = HSP_NCRED[c0];
        if (r0 >= 4) return HSP_MOST_ABUNDANT[r1];
        return tron_of[16 * r0 + 4 * r1 + HSP_NCELEM[c2]];
    }
};

struct Level {                          // the search level's parameters as this task uses them
    int elem, tpl, mask, width, weight, gain, gain1, cutoff, precutoff, vthr, tplwt;
    bool spaced;
};

struct Shared {                         // the wave's LDS
    int* conv; int* exam; int* mtx; uint8_t* tron_of;
    uint32_t* hkey; uint16_t* hhead; uint16_t* next;            // query words: open-addressing table + chains (position + 1, 0 ends)
    uint32_t* hits; int* raw;                                   // shared words as sort keys; seed segments {lastj, r, maxj, mxscr}
    int* ctr;                                                   // [0] hits, [1] segments, [2] HSPs out, [3] trouble
    uint32_t hmask;
};

// a word of the reduced alphabet at position p of a sequence read through `code(i)`; step 1 (nucleotides, query) or 3 (codons of a
// frame).  EMPTY: some residue of the pattern is not part of the alphabet
template <class F>
__device__ __forceinline__ uint32_t spell(const Level& L, const Shared& S, F code, int p, int step)
{
    uint32_t w = 0;
    for (int k = 0; k < L.weight; ++k) {
        const int c = S.conv[code(p + step * S.exam[k]) & 31];
        if (c >= L.elem) return EMPTY;
        w = w * (uint32_t) L.elem + (uint32_t) c;
    }
    return w;
}

__device__ __forceinline__ uint32_t slot_of(uint32_t w, uint32_t mask) { return (w * 2654435761u >> 7) & mask; }

// ---- the segment score of one diagonal (Wlp::scan_b / enter and the closing of a diagonal in dmsnno) ---------------------------------
struct Diagonal {
    int score = 0, best = 0, prev, first = 0, best_at = 0;      // prev: last shared word; first: where the running segment began
    __device__ Diagonal(int width) : prev(-(width + 1)) {}
};
__device__ __forceinline__ void put_segment(const Shared& S, const Diagonal& d, int r, int cap)
{
    const int k = atomicAdd(&S.ctr[1], 1);
    if (k >= cap) { atomicOr(&S.ctr[3], HSP_TOO_MANY); return; }
    S.raw[4 * k] = d.first; S.raw[4 * k + 1] = r; S.raw[4 * k + 2] = d.best_at; S.raw[4 * k + 3] = d.best;
}
__device__ __forceinline__ void word_on_diagonal(const Level& L, const Shared& S, Diagonal& d, int m, int r, int cap)
{
    const int gap = m - d.prev - L.width;
    if (gap > 0) {
        // words apart: the gap costs; a segment that has fallen too far below its best (or below zero) ends here
        const int floor_ = d.best - L.cutoff;
        d.score -= L.gain * gap;
        if (floor_ > d.score || d.score < 0) {
            if (floor_ > 0) put_segment(S, d, r, cap);
            d.score = L.tplwt + (m < L.width ? L.gain * (L.width - m) : 0);
            d.best = d.score; d.best_at = d.first = m;
        } else d.score += L.tplwt;
    } else d.score += (m - d.first == 1) ? L.gain1 : L.gain;     // (the reference compares with the segment's FIRST word here)
    if (d.score > d.best) { d.best = d.score; d.best_at = m; }
    d.prev = m;
}
__device__ __forceinline__ void close_diagonal(const Level& L, const Shared& S, Diagonal& d, int r, int mm, int cap)
{
    if (d.best <= L.precutoff) return;
    const int over = d.best_at + 2 * L.width - mm;               // a segment at the query's end is not blamed for stopping there
    if (over > 0) d.best += L.gain * over;
    if (d.best > L.cutoff) put_segment(S, d, r, cap);
}

// ---- stretch and trim (Wlp::eval): the segment's diagonal is followed backwards while the residue classes agree, then forwards
// through the segment and on while they agree; the score is the matrix score of the best window (running sum, reset below zero),
// plus the end bonuses.  -> false: not above the level's threshold
__device__ bool stretch_and_trim(const HspArgs& A, const HspTask& T, const Level& L, const Shared& S, const uint8_t* a, const Region& B,
                                 int bbt, int seg_first, int seg_r, int seg_last, int* out)
{
    int jx = T.a_left + seg_first, jy = bbt * seg_first + seg_r, jlen = seg_last - seg_first + L.width;
    int scr = 0;
    int as = jx, bs = jy + (bbt == 3 ? 1 : 0);
    const int seed_a = as, seed_b = bs;
    if (T.a_exgl) { const int lend = L.tpl - jx; if (lend > 0) scr += A.end_bonus * (lend < L.tpl ? lend : L.tpl); }
    // backwards
    while (--as >= 0 && (bs -= bbt) >= 0) {
        if ((as < seed_a || bs < seed_b) && S.conv[a[as] & 31] != S.conv[B.at(bs) & 31]) break;
        --jx; jy -= bbt; ++jlen;
    }
    if (as < 0) bs -= bbt;
    const int a_end = jx + jlen < T.a_right ? jx + jlen : T.a_right;
    const int b_end = jy + bbt * jlen < B.len ? jy + bbt * jlen : B.len;
    const int b_stop = bbt == 3 ? B.len - 1 : B.len;
    const int from = as;
    int len = 0, nid = 0, best = scr, w_from = 0, w_len = 0, w_nid = 0, restart = 0;
    while (++as < T.a_len && (bs += bbt) < b_stop) {
        const int ca = a[as], cb = B.at(bs);
        if ((as >= a_end || bs >= b_end) && S.conv[ca & 31] != S.conv[cb & 31]) break;
        ++len;
        scr += S.mtx[ca * A.mtx_cols + cb];
        if (ca == cb || (ca == A.ser && cb == A.ser2)) ++nid;
        if (scr < 0) { scr = 0; restart = as - from; len = nid = 0; }
        if (scr > best) { best = scr; w_from = restart; w_len = len; w_nid = nid; }
    }
    jx += w_from; jy += bbt * w_from;
    if (A.crs == 0 && bbt == 3) { int mis = w_len - w_nid; if (mis > 3) mis = 3; scr -= mis * L.vthr; }
    const int rend = L.tpl - T.a_right + jx + w_len;
    if (T.a_exgr && rend > 0) scr += A.end_bonus * (rend < L.tpl ? rend : L.tpl);
    else if (w_nid == w_len) scr += A.end_bonus * 4;
    if (scr <= L.vthr) return false;
    out[0] = jx; out[1] = jy; out[2] = w_len; out[3] = w_nid; out[4] = scr; out[5] = seg_r; out[6] = seg_first; out[7] = 0;
    return true;
}

__global__ void __launch_bounds__(64) spdp_hsp_search(HspArgs A)
{
    extern __shared__ uint32_t lds[];
    const int me = lane_id();
    // ---- LDS: tables first
    Shared S;
    uint32_t* l = lds;
    S.ctr = (int*) l; l += 8;
    S.conv = (int*) l; l += 32;
    S.exam = (int*) l; l += 32;
    S.mtx = (int*) l; l += A.mtx_rows * A.mtx_cols;
    S.tron_of = (uint8_t*) l; l += 16;
    S.hkey = l; l += A.hash_slots;
    S.hits = l; l += A.hit_cap;
    S.raw = (int*) l; l += 4 * A.seg_cap;
    S.hhead = (uint16_t*) l; l += A.hash_slots / 2;
    S.next = (uint16_t*) l;
    S.hmask = (uint32_t) A.hash_slots - 1;
    for (int i = me; i < 32; i += 64) { S.conv[i] = A.level.convtab[i]; }
    for (int i = me; i < A.mtx_rows * A.mtx_cols; i += 64) S.mtx[i] = A.mtx[i];
    if (me < 64) S.tron_of[me] = A.tron_of[me];
    // the pattern: where its residues sit (a contiguous word: 0 .. width - 1)
    Level L;
    L.elem = A.level.elem; L.tpl = A.level.tpl; L.mask = A.level.mask; L.width = A.level.width; L.gain = A.level.gain; L.gain1 = A.level.gain1;
    L.spaced = A.level.bitpat_len > 0;
    {
        int wt = 0;
        if (L.spaced) { for (int w = 0; w < A.level.bitpat_len; ++w) if (A.level.bitpat[w]) { if (me == 0) S.exam[wt] = w; ++wt; } }
        else { wt = L.width; if (me < wt) S.exam[me] = me; }
        L.weight = wt;
    }
    L.tplwt = L.tpl * L.gain;
    lds_sync();
    const int bbt = A.bbt;
    for (int t = blockIdx.x; t < A.n_tasks; t += gridDim.x) {
        const HspTask T = A.tasks[t];
        // cut-offs: a query below `shortquery` residues scales them (Wlp::Wlp, src/wln.cc:222)
        L.cutoff = A.level.cutoff; L.vthr = A.level.vthr; L.precutoff = L.cutoff - L.gain * L.tpl;
        if (T.a_len < A.shortquery) {
            L.cutoff = L.cutoff * T.a_len / A.shortquery; L.vthr = L.vthr * T.a_len / A.shortquery; L.precutoff = L.precutoff * T.a_len / A.shortquery;
        }
        const uint8_t* a = A.codes + T.a_off;
        Region B = {A.genome + T.g_off, T.b_len, T.rvs != 0, bbt == 3, S.tron_of};
        const int mm = T.a_right - T.a_left, span_a = L.width - 1, span_b = bbt == 3 ? 3 * L.width - 1 : L.width - 1;
        const int nk = mm - span_a, nn = T.b_len - span_b;
        int* count = A.counts + 2 * (size_t) t;
        if (me < 4) S.ctr[me] = 0;
        int mbits = 1;
        while ((1 << mbits) < nk) ++mbits;
        const int r_bias = bbt * (nk > 0 ? nk - 1 : 0);
        bool fits = nk > 0 && nn > 0 && T.a_right - T.a_left >= L.width && T.b_len >= bbt * L.width;
        const bool empty_task = !fits;
        if (fits && (nk > A.hash_slots / 2 || nk >= 65535 || (u64) (nn + r_bias) << mbits > 0xfffffff0ull)) { fits = false; if (me == 0) S.ctr[3] = HSP_TOO_LONG; }
        if (!fits) {
            lds_sync();
            if (me == 0) { count[0] = 0; count[1] = empty_task ? 0 : S.ctr[3]; }
            lds_sync();
            continue;
        }
        // ---- 1. the query's words
        for (uint32_t i = me; i < (uint32_t) A.hash_slots; i += 64) { S.hkey[i] = EMPTY; S.hhead[i] = 0; }
        lds_sync();
        for (int m0 = 0; m0 < nk; m0 += 64) {
            const int m = m0 + me;
            if (m < nk) {
                const uint32_t w = spell(L, S, [&](int i) { return (int) a[T.a_left + i]; }, m, 1);
                if (w != EMPTY && w < (uint32_t) L.mask) {
                    uint32_t s = slot_of(w, S.hmask);
                    for (;;) {
                        const uint32_t old = atomicCAS(&S.hkey[s], EMPTY, w);
                        if (old == EMPTY || old == w) break;
                        s = (s + 1) & S.hmask;
                    }
                    // chain: the slot's head becomes this position, this position points at the old head (16-bit exchange done on the
                    // 32-bit word that holds it)
                    uint32_t* word32 = (uint32_t*) S.hhead + (s >> 1);
                    const int sh = (s & 1) * 16;
                    uint32_t seen = *word32, want;
                    do { want = (seen & ~(0xffffu << sh)) | ((uint32_t) (m + 1) << sh); } while (!__hip_atomic_compare_exchange_strong(word32, &seen, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    S.next[m] = (uint16_t) ((seen >> sh) & 0xffffu);
                }
            }
        }
        lds_sync();
        // ---- 2. the region's words against them: one key per shared word
        for (int n0 = 0; n0 < nn; n0 += 64) {
            const int n = n0 + me;
            if (n < nn) {
                const uint32_t w = bbt == 3 ? spell(L, S, [&](int i) { return B.at(i); }, n + 1, 3)
                                            : spell(L, S, [&](int i) { return B.at(i); }, n, 1);
                if (w != EMPTY && w < (uint32_t) L.mask) {
                    uint32_t s = slot_of(w, S.hmask);
                    uint32_t k;
                    while ((k = S.hkey[s]) != EMPTY && k != w) s = (s + 1) & S.hmask;
                    if (k == w)
                        for (int m1 = S.hhead[s]; m1; m1 = S.next[m1 - 1]) {
                            const int m = m1 - 1;
                            const int at = atomicAdd(&S.ctr[0], 1);
                            if (at < A.hit_cap) S.hits[at] = ((uint32_t) (n - bbt * m + r_bias) << mbits) | (uint32_t) m;
                        }
                }
            }
        }
        lds_sync();
        int n_hits = S.ctr[0];
        if (n_hits > A.hit_cap) {
            if (me == 0) { count[0] = 0; count[1] = HSP_TOO_MANY; }
            lds_sync();
            continue;
        }
        // ---- 3. sorted by (diagonal, query position)
        int np2 = 64;
        while (np2 < n_hits) np2 <<= 1;
        for (int i = n_hits + me; i < np2; i += 64) S.hits[i] = EMPTY;
        lds_sync();
        for (int k = 2; k <= np2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = me; i < np2; i += 64) {
                    const int x = i ^ j;
                    if (x > i) {
                        const uint32_t u = S.hits[i], v = S.hits[x];
                        if (((i & k) == 0) == (u > v)) { S.hits[i] = v; S.hits[x] = u; }
                    }
                }
                lds_sync();
            }
        // ---- 4. diagonals: a lane owns the runs that start in its slice
        {
            const int per = (n_hits + 63) / 64;
            int i = me * per;
            const int slice_end = i + per < n_hits ? i + per : n_hits;
            while (i < slice_end && i > 0 && (S.hits[i] >> mbits) == (S.hits[i - 1] >> mbits)) ++i;       // (the run belongs to the lane before)
            while (i < slice_end) {
                const uint32_t dkey = S.hits[i] >> mbits;
                const int r = (int) dkey - r_bias;
                Diagonal d(L.width);
                for ( ; i < n_hits && (S.hits[i] >> mbits) == dkey; ++i) word_on_diagonal(L, S, d, (int) (S.hits[i] & ((1u << mbits) - 1)), r, A.seg_cap);
                close_diagonal(L, S, d, r, mm, A.seg_cap);
            }
        }
        lds_sync();
        const int n_seg = S.ctr[1] < A.seg_cap ? S.ctr[1] : A.seg_cap;
        // ---- 5. stretch and trim; what stays above the threshold goes out
        int* out = A.out + (size_t) t * A.out_cap * 8;
        for (int s0 = 0; s0 < n_seg; s0 += 64) {
            const int s = s0 + me;
            if (s < n_seg) {
                int rec[8];
                if (stretch_and_trim(A, T, L, S, a, B, bbt, S.raw[4 * s], S.raw[4 * s + 1], S.raw[4 * s + 2], rec)) {
                    const int k = atomicAdd(&S.ctr[2], 1);
                    if (k < A.out_cap) for (int x = 0; x < 8; ++x) out[8 * k + x] = rec[x];
                }
            }
        }
        wave_sync();
        if (me == 0) {
            int flags = S.ctr[3];
            if (S.ctr[2] > A.out_cap) flags |= HSP_TOO_MANY;
            count[0] = S.ctr[2] < A.out_cap ? S.ctr[2] : A.out_cap; count[1] = flags;
        }
        lds_sync();
    }
}

}   // namespace

extern "C" uint32_t spdp_hsp_lds_bytes(const HspArgs* a)
{
    size_t w = 8 + 32 + 32 + (size_t) a->mtx_rows * a->mtx_cols + 16 + a->hash_slots + a->hit_cap + 4 * (size_t) a->seg_cap + a->hash_slots / 2 + a->hash_slots / 4 + 8;
    return (uint32_t) (4 * w);
}
extern "C" hipError_t spdp_hsp_launch(const HspArgs* a, int n_waves, hipStream_t s)
{
    hipLaunchKernelGGL(spdp_hsp_search, dim3(n_waves), dim3(64), spdp_hsp_lds_bytes(a), s, *a);
    return hipGetLastError();
}
