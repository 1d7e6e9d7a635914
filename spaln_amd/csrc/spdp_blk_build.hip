// spdp_blk_build.hip -- the block-index builder's device passes (include/spdp.h "the index builder").
//
// What the reference does with a per-residue state machine, twice over the genome (MakeBlk::scan_genome / m_scan_genome,
// src/blksrc.cc:1111-1183, 1509-1593; Block::c2w, :448-464; Bitpat_wq::word / flaw, src/bitpat.cc:178-212; Chash::countBlk /
// registBlk, src/blksrc.cc:402-425): every residue ends a word of every bit pattern; the word counts towards the word's
// frequency (tcount) and, at the phases its run of unambiguous residues allows, towards "block b holds word w".  Restated
// without the state: for the word that ends at residue j, seen from a block whose word state started afresh at residue lo,
//     ss       = residues since max(lo, the last ambiguous residue + 1), a 16-bit counter in the reference
//     flawless = the residues the pattern examines all lie at or behind lo and are unambiguous (a contiguous pattern:
//                ss >= its weight)
//     phase    = (unsigned) (ss - width) % Nshift == 0        (the unsigned remainder is the reference's: Nshift is an INT)
// so every residue is independent of every other once the last ambiguous residue before it is known: a prefix maximum.
//
// Passes: blkidx_tile_last (the last ambiguous residue of every 4096-residue tile; the carry between tiles is a host loop
// over the tile records), blkidx_words (one block per tile: the tile + a halo of the widest pattern in LDS, 16 residues per
// thread, the in-tile prefix maximum by wave shuffles; counts by global atomics, (word, block) keys appended through a
// wave-aggregated counter), a radix sort of the keys (rocPRIM), blkidx_heads / blkidx_compact (a key that differs from its
// left neighbour opens a block of a word's list; the kept words' lists written where the host's table says).
// HBM traffic: 1 B / residue read once, 8 B / key written, the sort's passes over the keys; the rest stays on the chip.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "spdp_internal.h"
#include "spdp_blk_build.h"

namespace {

constexpr int TILE = 4096, TPB = 256, PER = TILE / TPB, HALO = 32;

__device__ __forceinline__ int reduced_nt(int code) { return code == 2 ? 0 : code == 3 ? 1 : code == 5 ? 2 : code == 9 ? 3 : 4; }

__global__ __launch_bounds__(TPB) void blkidx_tile_last(const uint8_t* __restrict__ codes, int64_t G, int64_t* __restrict__ tile_last)
{
    const int64_t t0 = (int64_t) blockIdx.x * TILE;
    int64_t last = -1;
    for (int i = threadIdx.x; i < TILE; i += TPB) {
        const int64_t g = t0 + i;
        if (g < G && reduced_nt(codes[g]) == 4) last = g;            // (i grows: the thread's last hit is its largest)
    }
    for (int o = 32; o; o >>= 1) { const int64_t v = __shfl_xor(last, o); last = v > last ? v : last; }
    __shared__ int64_t s[TPB / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = last;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) last = s[w] > last ? s[w] : last;
        tile_last[blockIdx.x] = last;
    }
}

__global__ __launch_bounds__(TPB) void blkidx_words(BlkBuildArgs A)
{
    __shared__ uint8_t s_x[HALO + TILE];                             // reduced codes of residues t0 - HALO .. t0 + TILE
    __shared__ int64_t s_wave[TPB / 64];
    const int64_t t0 = (int64_t) blockIdx.x * TILE;
    for (int i = threadIdx.x; i < HALO + TILE; i += TPB) {
        const int64_t g = t0 - HALO + i;
        s_x[i] = (g >= 0 && g < A.G) ? (uint8_t) reduced_nt(A.codes[g]) : (uint8_t) 4;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p0 = t0 + (int64_t) threadIdx.x * PER;
    // the last ambiguous residue before my first one: tiles before mine, threads before me
    int64_t mine = -1;
    for (int i = 0; i < PER; ++i) if (s_x[HALO + threadIdx.x * PER + i] == 4 && p0 + i < A.G) mine = p0 + i;
    int64_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const int64_t v = __shfl_up(incl, o); if (lane >= o && v > incl) incl = v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int64_t last = A.tile_carry[blockIdx.x];
    for (int w = 0; w < wave; ++w) last = s_wave[w] > last ? s_wave[w] : last;
    { const int64_t prev = __shfl_up(incl, 1); if (lane > 0 && prev > last) last = prev; }

    // my chromosome
    int c = 0;
    {
        int lo = 0, hi = A.n_chr;                                    // the last c with chr_off[c] <= p0
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.chr_off[mid] <= p0) lo = mid; else hi = mid; }
        c = lo;
    }
    int64_t c_lo = A.chr_off[c], c_hi = A.chr_off[c + 1];
    const int64_t s_size = (int64_t) A.margin + A.blklen;
    for (int i = 0; i < PER; ++i) {
        const int64_t g = p0 + i;
        const bool in = g < A.G;
        if (in) while (g >= c_hi && c + 1 < A.n_chr) { ++c; c_lo = c_hi; c_hi = A.chr_off[c + 1]; }
        const int xi = HALO + threadIdx.x * PER + i;
        const int uc = s_x[xi];
        if (in && uc == 4) last = g;
        const bool good = in && uc != 4 && g >= c_lo && g < c_hi;
        const int64_t jc = g - c_lo, L = c_hi - c_lo;
        const int64_t nb = L < s_size ? 1 : 1 + (L - A.margin) / A.blklen;
        // the blocks that see this residue, and where their word state started
        int64_t blk[2], lo[2];
        int n_cand = 0;
        if (good) {
            if (!A.threaded) {
                blk[0] = jc < s_size ? 0 : (jc - A.margin) / A.blklen;
                lo[0] = blk[0] ? blk[0] * A.blklen + A.margin : 0;
                n_cand = 1;
            } else {
                const int64_t b1 = jc / A.blklen;
                if (b1 < nb) { blk[n_cand] = b1; lo[n_cand] = b1 * A.blklen; ++n_cand; }
                if (b1 >= 1 && jc < b1 * A.blklen + A.margin) { blk[n_cand] = b1 - 1; lo[n_cand] = (b1 - 1) * A.blklen; ++n_cand; }
            }
        }
        for (int cd = 0; cd < 2; ++cd) {
            const bool on = cd < n_cand;
            const int64_t lo_g = on ? c_lo + lo[cd] : 0;
            const int64_t run0 = last + 1 > lo_g ? last + 1 : lo_g;
            const int64_t ss = g - run0 + 1;
            const bool counts = on && jc < (blk[cd] + 1) * (int64_t) A.blklen;
            for (int k = 0; k < A.nbit; ++k) {
                const int width = A.width[k];
                bool ok = on && g - width + 1 >= lo_g;
                uint32_t w = 0;
                if (A.spaced[k]) {
                    for (int e = 0; e < A.weight; ++e) {
                        const int x = s_x[xi - width + 1 + A.exam[k][e]];
                        ok = ok && x != 4;
                        w = w * 4 + (uint32_t) (x & 3);
                    }
                } else {
                    ok = ok && ss >= A.weight;
                    for (int e = 0; e < A.weight; ++e) w = w * 4 + (uint32_t) (s_x[xi - A.weight + 1 + e] & 3);
                }
                if (ok && counts) atomicAdd(A.tcount + w, 1u);
                const int nw = (int) (ss & 0xffff) - width;
                const bool emit = ok && (uint32_t) nw % (uint32_t) A.nshift == 0;
                // one counter bump per wave
                const unsigned long long m = __ballot(emit);
                if (m) {
                    unsigned long long base = 0;
                    const int leader = __ffsll((long long) m) - 1;
                    if (lane == leader) base = atomicAdd(A.n_keys, (unsigned long long) __popcll(m));
                    base = ((unsigned long long) (unsigned) __shfl((int) (base >> 32), leader) << 32) | (unsigned) __shfl((int) base, leader);
                    if (emit) {
                        const unsigned long long at = base + __popcll(m & ((1ull << lane) - 1));
                        if (at < A.cap) A.keys[at] = ((unsigned long long) w << 32) | (unsigned) (A.chr_first[c] + (int) blk[cd]);
                    }
                }
            }
        }
    }
}

// ---- the translated index (`spaln -W -KP`; Block::c2w6 / c2w6_pp, src/blksrc.cc:466-532) ---------------------------------------
// A residue completes one codon on each strand; the codon's amino-acid class joins the word of its reading frame.  What the
// reference keeps as state -- the open reading frame's length per frame (ss6), the word's flaw counter, a ring of MinOrf slots
// per strand in which a taken word waits before it reaches the block's hash and from which the words of a frame that closes
// before MinOrf nucleotides are struck -- restated per residue j and strand, for a block whose state started afresh at lo:
//     sss      = codons with a class that end at j, j - 3, ... without a gap, counted from lo at the earliest
//                = min((j - the last class-less codon end of this frame) / 3, (j - 2 - lo) / 3 + 1)         (a prefix maximum per frame)
//     word     = the classes of the last K of them, taken when sss >= K and (sss - K) % Nshift == 0
//     struck   = a frame end at j + 3c (c >= 1, before the block's last residue and at most MinOrf residues on) whose own frame
//                was shorter than MinOrf aims at this slot: the reference's loop over (sp + 1 + i * Nshift) codons back, followed
//                literally (its extra turn when 2 sp >= Nshift included)
//     arrives  = MinOrf residues later in the same block; past the last residue of a chromosome's last block the ring is emptied
//                into it as far as blklen reaches (the rest is lost, and an empty block follows: the host counts it)
constexpr int P_BACK = 32, P_FWD = 128;

struct CodonClasses { uint8_t c[64]; };

__device__ __forceinline__ void tron_classes(const CodonClasses& cc, int x0, int x1, int x2, int& fwd, int& rev)
{
    if (x0 > 3 || x1 > 3 || x2 > 3) { fwd = rev = 255; return; }
    fwd = cc.c[16 * x0 + 4 * x1 + x2];
    rev = cc.c[16 * (3 - x2) + 4 * (3 - x1) + (3 - x0)];
}

// per tile and (residue class mod 3, strand): the last residue whose codon has no class
__global__ __launch_bounds__(TPB) void blkidx_tile_last6(const uint8_t* __restrict__ codes, int64_t G, CodonClasses cc, int nalpha,
                                                         int64_t* __restrict__ tile_last)
{
    __shared__ CodonClasses s_cc;
    __shared__ int64_t s[TPB / 64][6];
    if (threadIdx.x < 64) s_cc.c[threadIdx.x] = cc.c[threadIdx.x];
    __syncthreads();
    const int64_t t0 = (int64_t) blockIdx.x * TILE;
    int64_t last[6] = {-1, -1, -1, -1, -1, -1};
    for (int i = threadIdx.x; i < TILE; i += TPB) {
        const int64_t g = t0 + i;
        if (g >= G) break;
        int f = 255, r = 255;
        if (g >= 2) tron_classes(s_cc, reduced_nt(codes[g - 2]), reduced_nt(codes[g - 1]), reduced_nt(codes[g]), f, r);
        const int cls = (int) (g % 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (cls == k && f >= nalpha) last[2 * k] = g;
            if (cls == k && r >= nalpha) last[2 * k + 1] = g;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
        for (int o = 32; o; o >>= 1) { const int64_t v = __shfl_xor(last[k], o); last[k] = v > last[k] ? v : last[k]; }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; ++k) s[threadIdx.x >> 6][k] = last[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        int64_t m = s[0][threadIdx.x];
        for (int w = 1; w < TPB / 64; ++w) m = s[w][threadIdx.x] > m ? s[w][threadIdx.x] : m;
        tile_last[6 * (int64_t) blockIdx.x + threadIdx.x] = m;
    }
}

__global__ __launch_bounds__(TPB) void blkidx_words_p(BlkBuildArgsP A)
{
    __shared__ uint8_t s_x[P_BACK + TILE + P_FWD];                   // reduced codes of residues t0 - P_BACK .. t0 + TILE + P_FWD
    __shared__ uint8_t s_c[2][P_BACK + TILE + P_FWD];                // the class of the codon that ends there, per strand
    __shared__ int s_wave[TPB / 64][6];
    __shared__ CodonClasses s_cc;
    const int64_t t0 = (int64_t) blockIdx.x * TILE;
    if (threadIdx.x < 64) s_cc.c[threadIdx.x] = A.codon_class[threadIdx.x];
    for (int i = threadIdx.x; i < P_BACK + TILE + P_FWD; i += TPB) {
        const int64_t g = t0 - P_BACK + i;
        s_x[i] = (g >= 0 && g < A.G) ? (uint8_t) reduced_nt(A.codes[g]) : (uint8_t) 4;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P_BACK + TILE + P_FWD; i += TPB) {
        int f = 255, r = 255;
        if (i >= 2) tron_classes(s_cc, s_x[i - 2], s_x[i - 1], s_x[i], f, r);
        s_c[0][i] = (uint8_t) f; s_c[1][i] = (uint8_t) r;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = threadIdx.x * PER;                                // tile-relative
    const int64_t p0 = t0 + i0;
    const int r0 = (int) (p0 % 3);
    // the last class-less codon end before my first residue, per (residue class, strand): tile-relative, far back = none
    constexpr int NONE = -(1 << 24);
    int mine[6] = {NONE, NONE, NONE, NONE, NONE, NONE};
    for (int i = 0; i < PER; ++i) {
        if (p0 + i >= A.G) break;
        const int cls = (r0 + i) % 3;
        const bool bf = s_c[0][P_BACK + i0 + i] >= A.nalpha, br = s_c[1][P_BACK + i0 + i] >= A.nalpha;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (cls == k && bf) mine[2 * k] = i0 + i;
            if (cls == k && br) mine[2 * k + 1] = i0 + i;
        }
    }
    int before[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int incl = mine[k];
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o && v > incl) incl = v; }
        if (lane == 63) s_wave[wave][k] = incl;
        const int prev = __shfl_up(incl, 1);
        before[k] = lane > 0 ? prev : NONE;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int64_t carry = A.tile_carry[6 * (int64_t) blockIdx.x + k];
        int last = (carry < 0 || t0 - carry > (1 << 23)) ? NONE : (int) (carry - t0);
        for (int w = 0; w < wave; ++w) last = s_wave[w][k] > last ? s_wave[w][k] : last;
        before[k] = before[k] > last ? before[k] : last;
    }
    // rotate so that l?[0] belongs to the residue class of my first residue
    int lf[3], lr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int cls = (r0 + k) % 3;
        lf[k] = cls == 0 ? before[0] : cls == 1 ? before[2] : before[4];
        lr[k] = cls == 0 ? before[1] : cls == 1 ? before[3] : before[5];
    }
    int lf0 = lf[0], lf1 = lf[1], lf2 = lf[2], lr0 = lr[0], lr1 = lr[1], lr2 = lr[2];

    int c = 0;
    {
        int lo = 0, hi = A.n_chr;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.chr_off[mid] <= p0) lo = mid; else hi = mid; }
        c = lo;
    }
    int64_t c_lo = A.chr_off[c], c_hi = A.chr_off[c + 1];
    const int64_t s_size = (int64_t) A.margin + A.blklen;
    const int K = A.K, wq = A.minorf, nshift = A.nshift;
    for (int i = 0; i < PER; ++i) {
        const int64_t g = p0 + i;
        const bool in = g < A.G;
        if (in) while (g >= c_hi && c + 1 < A.n_chr) { ++c; c_lo = c_hi; c_hi = A.chr_off[c + 1]; }
        const int ti = i0 + i, xi = P_BACK + ti;
        const int cf = s_c[0][xi], cr = s_c[1][xi];
        if (in && cf >= A.nalpha) lf0 = ti;
        if (in && cr >= A.nalpha) lr0 = ti;
        const bool here = in && g >= c_lo && g < c_hi;
        const int64_t jc = g - c_lo, L = c_hi - c_lo;
        const int64_t nb = L < s_size ? 1 : 1 + (L - A.margin) / A.blklen;
        int64_t blk[2] = {0, 0}, lo[2] = {0, 0};
        int n_cand = 0;
        if (here) {
            if (!A.threaded) {
                blk[0] = jc < s_size ? 0 : (jc - A.margin) / A.blklen;
                lo[0] = blk[0] ? blk[0] * A.blklen + A.margin : 0;
                n_cand = 1;
            } else {
                const int64_t b1 = jc / A.blklen;
                if (b1 < nb) { blk[n_cand] = b1; lo[n_cand] = b1 * A.blklen; ++n_cand; }
                if (b1 >= 1 && jc < b1 * A.blklen + A.margin) { blk[n_cand] = b1 - 1; lo[n_cand] = (b1 - 1) * A.blklen; ++n_cand; }
            }
        }
        for (int cd = 0; cd < 2; ++cd) {
            const bool on = cd < n_cand;
            const int64_t lo_c = lo[cd];                                         // chromosome-relative
            int64_t hi_c = (blk[cd] + 1) * (int64_t) A.blklen + A.margin;
            if (hi_c > L) hi_c = L;
            const int pos0 = (!A.threaded && blk[cd]) ? A.margin : 0;
            const bool counts = on && pos0 + (jc - lo_c) < A.blklen;
            const bool tail = blk[cd] == nb - 1;
            for (int d = 0; d < 2; ++d) {
                const int cls = d ? cr : cf;
                const int lastbad = d ? lr0 : lf0;
                bool ok = on && cls < A.nalpha && jc - 2 >= lo_c;
                int sss = 0;
                if (ok) {
                    const int64_t since = (jc - 2 - lo_c) / 3 + 1;
                    const int run = (ti - lastbad) / 3;
                    sss = (int) (since < run ? since : run);
                    ok = sss >= K;
                }
                uint32_t w = 0;
                if (ok) for (int e = 0; e < K; ++e) w = w * (uint32_t) A.nalpha + s_c[d][xi - 3 * (K - 1 - e)];
                if (ok && counts) atomicAdd(A.tcount + w, 1u);
                bool emit = ok && (sss - K) % nshift == 0;
                uint32_t to_block = 0;
                if (emit) {
                    // is it struck from the ring before it arrives?
                    int s = sss;
                    for (int cc = 1; 3 * cc <= wq && jc + 3 * cc < hi_c; ++cc) {
                        if (s_c[d][xi + 3 * cc] < A.nalpha) { ++s; continue; }
                        if (s >= K && 3 * s < wq) {
                            const int nw = s - K, sp = nw % nshift, turns = (nw + sp) / nshift + 1;
                            const int t = cc - (sp + 1);
                            if (t >= 0 && t % nshift == 0 && t / nshift < turns) { emit = false; break; }
                        }
                        s = 0;
                    }
                    const int64_t jp = jc + wq;
                    if (jp < hi_c) to_block = (uint32_t) (A.chr_first[c] + (int) blk[cd]);
                    else if (tail) {
                        const int64_t n = pos0 + (hi_c - lo_c);
                        const int64_t rest = n > A.blklen ? n - A.blklen : 0;
                        to_block = (uint32_t) (A.chr_first[c] + (int) blk[cd]);
                        if (jp - hi_c >= wq - rest) emit = false;                // the reset that closes the block empties the ring first
                    } else emit = false;
                }
                const unsigned long long m = __ballot(emit);
                if (m) {
                    unsigned long long base = 0;
                    const int leader = __ffsll((long long) m) - 1;
                    if (lane == leader) base = atomicAdd(A.n_keys, (unsigned long long) __popcll(m));
                    base = ((unsigned long long) (unsigned) __shfl((int) (base >> 32), leader) << 32) | (unsigned) __shfl((int) base, leader);
                    if (emit) {
                        const unsigned long long at = base + __popcll(m & ((1ull << lane) - 1));
                        if (at < A.cap) A.keys[at] = ((unsigned long long) w << 32) | to_block;
                    }
                }
            }
        }
        // the next residue belongs to the next residue class
        { const int t = lf0; lf0 = lf1; lf1 = lf2; lf2 = t; }
        { const int t = lr0; lr0 = lr1; lr1 = lr2; lr2 = t; }
    }
}

// flag[i] = key i opens a (word, block) entry; cnt[word] = entries of the word
__global__ void blkidx_heads(const unsigned long long* __restrict__ keys, int64_t n, uint32_t* __restrict__ cnt, uint8_t* __restrict__ flag)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    flag[i] = head ? 1 : 0;
    if (head) atomicAdd(cnt + (uint32_t) (keys[i] >> 32), 1u);
}

// unique key u (the u-th head) -> its place in the kept words' lists
__global__ void blkidx_compact(const unsigned long long* __restrict__ ukeys, int64_t n_u, const uint64_t* __restrict__ uoff,
                               const int32_t* __restrict__ blkp, uint32_t* __restrict__ blkb)
{
    const int64_t u = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_u) return;
    const uint32_t w = (uint32_t) (ukeys[u] >> 32);
    const int32_t p = blkp[w];
    if (p > 0) blkb[(int64_t) p - 1 + (u - (int64_t) uoff[w])] = (uint32_t) ukeys[u];
}

struct Dev {
    void* p = nullptr;
    ~Dev() { if (p) (void) hipFree(p); }
    hipError_t get(size_t bytes) { if (p) { (void) hipFree(p); p = nullptr; } return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T> T* as() const { return (T*) p; }
};

}   // namespace

struct BlkBuildDev {
    Dev codes, chr_off, chr_first, tile_last, tcount, cnt, keys, keys2, n_keys, flag, ukeys, n_u, uoff, tmp, blkp, blkb;
    int64_t n_unique = 0;
    uint32_t tabsize = 0;
};

void spdp_blkidx_free(BlkBuildDev* d) { delete d; }

// the keys of a words pass -> tcount / cnt on the host, the sorted unique keys and where a word's entries start kept on the device
static int blkidx_finish_keys(SpdpContext* ctx, BlkBuildDev* d, unsigned long long n_keys, int key_bits, uint32_t tabsize,
                              std::vector<uint32_t>& tcount, std::vector<uint32_t>& cnt)
{
    hipStream_t st = ctx->stream;
    tcount.resize(tabsize); cnt.assign(tabsize, 0);
    HIPCHK(hipMemcpyAsync(tcount.data(), d->tcount.p, sizeof(uint32_t) * (size_t) tabsize, hipMemcpyDeviceToHost, st));
    d->n_unique = 0;
    if (n_keys) {
        // sort, heads, the unique keys
        HIPCHK(d->keys2.get(sizeof(unsigned long long) * n_keys));
        size_t tmp_bytes = 0;
        HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, d->keys.as<unsigned long long>(), d->keys2.as<unsigned long long>(), (size_t) n_keys, 0u, (unsigned) key_bits, st));
        HIPCHK(d->tmp.get(tmp_bytes));
        HIPCHK(rocprim::radix_sort_keys(d->tmp.p, tmp_bytes, d->keys.as<unsigned long long>(), d->keys2.as<unsigned long long>(), (size_t) n_keys, 0u, (unsigned) key_bits, st));
        HIPCHK(d->flag.get((size_t) n_keys));
        hipLaunchKernelGGL(blkidx_heads, dim3((unsigned) ((n_keys + 255) / 256)), dim3(256), 0, st, d->keys2.as<unsigned long long>(), (int64_t) n_keys,
                           d->cnt.as<uint32_t>(), d->flag.as<uint8_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(d->ukeys.get(sizeof(unsigned long long) * n_keys)); HIPCHK(d->n_u.get(sizeof(size_t)));
        tmp_bytes = 0;
        HIPCHK(rocprim::select(nullptr, tmp_bytes, d->keys2.as<unsigned long long>(), d->flag.as<uint8_t>(), d->ukeys.as<unsigned long long>(), d->n_u.as<size_t>(), (size_t) n_keys, st));
        HIPCHK(d->tmp.get(tmp_bytes));
        HIPCHK(rocprim::select(d->tmp.p, tmp_bytes, d->keys2.as<unsigned long long>(), d->flag.as<uint8_t>(), d->ukeys.as<unsigned long long>(), d->n_u.as<size_t>(), (size_t) n_keys, st));
        size_t nu = 0;
        HIPCHK(hipMemcpyAsync(&nu, d->n_u.p, sizeof nu, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(cnt.data(), d->cnt.p, sizeof(uint32_t) * (size_t) tabsize, hipMemcpyDeviceToHost, st));
        // where a word's entries start among the unique keys
        HIPCHK(d->uoff.get(sizeof(uint64_t) * (size_t) tabsize));
        tmp_bytes = 0;
        HIPCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, d->cnt.as<uint32_t>(), d->uoff.as<uint64_t>(), (uint64_t) 0, (size_t) tabsize, rocprim::plus<uint64_t>(), st));
        HIPCHK(d->tmp.get(tmp_bytes));
        HIPCHK(rocprim::exclusive_scan(d->tmp.p, tmp_bytes, d->cnt.as<uint32_t>(), d->uoff.as<uint64_t>(), (uint64_t) 0, (size_t) tabsize, rocprim::plus<uint64_t>(), st));
        HIPCHK(hipStreamSynchronize(st));
        d->n_unique = (int64_t) nu;
        (void) d->keys.get(0); (void) d->keys2.get(0); (void) d->flag.get(0);              // (the sort's buffers are no longer needed)
    } else HIPCHK(hipStreamSynchronize(st));
    return 0;
}

// pass 1: words -> tcount[tabsize], cnt[tabsize] (host vectors), the sorted unique keys kept on the device
int spdp_blkidx_words(SpdpContext* ctx, const uint8_t* codes, const int64_t* chr_off, const int32_t* chr_first, int n_chr,
                      BlkBuildArgs A, uint32_t tabsize, int key_bits, std::vector<uint32_t>& tcount, std::vector<uint32_t>& cnt,
                      BlkBuildDev** out)
{
    (void) hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    BlkBuildDev* d = new BlkBuildDev;
    struct Guard { BlkBuildDev*& d; bool keep = false; ~Guard() { if (!keep) { delete d; d = nullptr; } } } guard{d};
    const int64_t G = A.G;
    const int64_t n_tiles = (G + TILE - 1) / TILE;
    d->tabsize = tabsize;
    HIPCHK(d->codes.get((size_t) G)); HIPCHK(d->chr_off.get(sizeof(int64_t) * (n_chr + 1))); HIPCHK(d->chr_first.get(sizeof(int32_t) * n_chr));
    HIPCHK(d->tile_last.get(sizeof(int64_t) * n_tiles)); HIPCHK(d->tcount.get(sizeof(uint32_t) * (size_t) tabsize));
    HIPCHK(d->cnt.get(sizeof(uint32_t) * (size_t) tabsize)); HIPCHK(d->n_keys.get(sizeof(unsigned long long)));
    HIPCHK(hipMemcpyAsync(d->codes.p, codes, (size_t) G, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d->chr_off.p, chr_off, sizeof(int64_t) * (n_chr + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d->chr_first.p, chr_first, sizeof(int32_t) * n_chr, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d->tcount.p, 0, sizeof(uint32_t) * (size_t) tabsize, st));
    HIPCHK(hipMemsetAsync(d->cnt.p, 0, sizeof(uint32_t) * (size_t) tabsize, st));
    hipLaunchKernelGGL(blkidx_tile_last, dim3((unsigned) n_tiles), dim3(TPB), 0, st, d->codes.as<uint8_t>(), G, d->tile_last.as<int64_t>());
    HIPCHK(hipGetLastError());
    std::vector<int64_t> tl((size_t) n_tiles);
    HIPCHK(hipMemcpyAsync(tl.data(), d->tile_last.p, sizeof(int64_t) * n_tiles, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t run = -1;                                   // exclusive prefix maximum: what a tile inherits
    for (int64_t t = 0; t < n_tiles; ++t) { const int64_t mine = tl[t]; tl[t] = run; if (mine > run) run = mine; }
    HIPCHK(hipMemcpyAsync(d->tile_last.p, tl.data(), sizeof(int64_t) * n_tiles, hipMemcpyHostToDevice, st));
    A.codes = d->codes.as<uint8_t>(); A.chr_off = d->chr_off.as<int64_t>(); A.chr_first = d->chr_first.as<int32_t>(); A.n_chr = n_chr;
    A.tile_carry = d->tile_last.as<int64_t>(); A.tcount = d->tcount.as<uint32_t>(); A.n_keys = d->n_keys.as<unsigned long long>();
    // keys: about one per Nshift residues and pattern; a run that needs more says so and is repeated with room for them
    unsigned long long cap = (unsigned long long) ((double) G * A.nbit / A.nshift * 1.25) + (1ull << 20);
    unsigned long long n_keys = 0;
    for (int round = 0; round < 2; ++round) {
        HIPCHK(d->keys.get(sizeof(unsigned long long) * cap));
        HIPCHK(hipMemsetAsync(d->n_keys.p, 0, sizeof(unsigned long long), st));
        if (round) HIPCHK(hipMemsetAsync(d->tcount.p, 0, sizeof(uint32_t) * (size_t) tabsize, st));
        A.keys = d->keys.as<unsigned long long>(); A.cap = cap;
        hipLaunchKernelGGL(blkidx_words, dim3((unsigned) n_tiles), dim3(TPB), 0, st, A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&n_keys, d->n_keys.p, sizeof n_keys, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (n_keys <= cap) break;
        if (round) { ctx->err = "spdp_blk_index_build: key count changed between two passes"; return -1; }
        cap = n_keys;
    }
    if (blkidx_finish_keys(ctx, d, n_keys, key_bits, tabsize, tcount, cnt)) return -1;
    guard.keep = true;
    *out = d;
    return 0;
}

// pass 1 of the translated index: as spdp_blkidx_words, the words being amino-acid words of the six reading frames
int spdp_blkidx_words_p(SpdpContext* ctx, const uint8_t* codes, const int64_t* chr_off, const int32_t* chr_first, int n_chr,
                        BlkBuildArgsP A, int key_bits, std::vector<uint32_t>& tcount, std::vector<uint32_t>& cnt, BlkBuildDev** out)
{
    (void) hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    if (A.minorf + 2 > P_FWD || 3 * A.K + 2 > P_BACK) { ctx->err = "spdp_blk_index_build_p: MinOrf or the word length beyond the kernel's halo"; return -1; }
    BlkBuildDev* d = new BlkBuildDev;
    struct Guard { BlkBuildDev*& d; bool keep = false; ~Guard() { if (!keep) { delete d; d = nullptr; } } } guard{d};
    const int64_t G = A.G;
    const int64_t n_tiles = (G + TILE - 1) / TILE;
    const uint32_t tabsize = A.tabsize;
    d->tabsize = tabsize;
    HIPCHK(d->codes.get((size_t) G)); HIPCHK(d->chr_off.get(sizeof(int64_t) * (n_chr + 1))); HIPCHK(d->chr_first.get(sizeof(int32_t) * n_chr));
    HIPCHK(d->tile_last.get(sizeof(int64_t) * 6 * n_tiles)); HIPCHK(d->tcount.get(sizeof(uint32_t) * (size_t) tabsize));
    HIPCHK(d->cnt.get(sizeof(uint32_t) * (size_t) tabsize)); HIPCHK(d->n_keys.get(sizeof(unsigned long long)));
    HIPCHK(hipMemcpyAsync(d->codes.p, codes, (size_t) G, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d->chr_off.p, chr_off, sizeof(int64_t) * (n_chr + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d->chr_first.p, chr_first, sizeof(int32_t) * n_chr, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d->tcount.p, 0, sizeof(uint32_t) * (size_t) tabsize, st));
    HIPCHK(hipMemsetAsync(d->cnt.p, 0, sizeof(uint32_t) * (size_t) tabsize, st));
    CodonClasses cc;
    memcpy(cc.c, A.codon_class, 64);
    hipLaunchKernelGGL(blkidx_tile_last6, dim3((unsigned) n_tiles), dim3(TPB), 0, st, d->codes.as<uint8_t>(), G, cc, A.nalpha, d->tile_last.as<int64_t>());
    HIPCHK(hipGetLastError());
    std::vector<int64_t> tl((size_t) (6 * n_tiles));
    HIPCHK(hipMemcpyAsync(tl.data(), d->tile_last.p, sizeof(int64_t) * 6 * n_tiles, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t run[6] = {-1, -1, -1, -1, -1, -1};         // exclusive prefix maxima: what a tile inherits
    for (int64_t t = 0; t < n_tiles; ++t)
        for (int k = 0; k < 6; ++k) { const int64_t mine = tl[6 * t + k]; tl[6 * t + k] = run[k]; if (mine > run[k]) run[k] = mine; }
    HIPCHK(hipMemcpyAsync(d->tile_last.p, tl.data(), sizeof(int64_t) * 6 * n_tiles, hipMemcpyHostToDevice, st));
    A.codes = d->codes.as<uint8_t>(); A.chr_off = d->chr_off.as<int64_t>(); A.chr_first = d->chr_first.as<int32_t>(); A.n_chr = n_chr;
    A.tile_carry = d->tile_last.as<int64_t>(); A.tcount = d->tcount.as<uint32_t>(); A.n_keys = d->n_keys.as<unsigned long long>();
    // keys: two codons per residue, a word every Nshift codons of an open frame
    unsigned long long cap = (unsigned long long) ((double) G * 2 / A.nshift * 1.1) + (1ull << 20);
    unsigned long long n_keys = 0;
    for (int round = 0; round < 2; ++round) {
        HIPCHK(d->keys.get(sizeof(unsigned long long) * cap));
        HIPCHK(hipMemsetAsync(d->n_keys.p, 0, sizeof(unsigned long long), st));
        if (round) HIPCHK(hipMemsetAsync(d->tcount.p, 0, sizeof(uint32_t) * (size_t) tabsize, st));
        A.keys = d->keys.as<unsigned long long>(); A.cap = cap;
        hipLaunchKernelGGL(blkidx_words_p, dim3((unsigned) n_tiles), dim3(TPB), 0, st, A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&n_keys, d->n_keys.p, sizeof n_keys, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (n_keys <= cap) break;
        if (round) { ctx->err = "spdp_blk_index_build_p: key count changed between two passes"; return -1; }
        cap = n_keys;
    }
    if (blkidx_finish_keys(ctx, d, n_keys, key_bits, tabsize, tcount, cnt)) return -1;
    guard.keep = true;
    *out = d;
    return 0;
}

// pass 2: the kept words' lists, in the order the host's blkp table gives
int spdp_blkidx_lists(SpdpContext* ctx, BlkBuildDev* d, const int32_t* blkp, int64_t word_no, uint32_t* blkb)
{
    (void) hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    if (!word_no || !d->n_unique) return 0;
    HIPCHK(d->blkp.get(sizeof(int32_t) * (size_t) d->tabsize)); HIPCHK(d->blkb.get(sizeof(uint32_t) * (size_t) word_no));
    HIPCHK(hipMemcpyAsync(d->blkp.p, blkp, sizeof(int32_t) * (size_t) d->tabsize, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(blkidx_compact, dim3((unsigned) ((d->n_unique + 255) / 256)), dim3(256), 0, st, d->ukeys.as<unsigned long long>(), d->n_unique,
                       d->uoff.as<uint64_t>(), d->blkp.as<int32_t>(), d->blkb.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(blkb, d->blkb.p, sizeof(uint32_t) * (size_t) word_no, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}
