// spdp_blk_vote.hip -- the vote of the block search on the device: ONE QUERY PER WAVE.
//
// What is computed: the state the reference's block search holds at each of its TestOutput calls (ogotoh/spaln v3.0.7,
// SrchBlk::findblock src/blksrc.cc:2971-3087 with Qwords :2819-2969, Bhit4 :2763-2817, Randbs :2047-2069) and the list of
// candidate block pairs TestOutput makes of it (:2547-2672).  The k-mers of a query are looked up from both ends inwards on
// both strands (four directions x Nshift phases per round); every word votes for the blocks of its posting list; a block that
// keeps being hit by consecutive words of a phase (itself or its neighbour towards the scan) builds up a run score; blocks
// above the random expectation enter a bounded best-of list per direction; the scan stops when enough significant block
// pairs have appeared.
//
// How it is laid out for the machine.  A wave owns a query.  Control flow (rounds, directions, phases, stop rules) is uniform
// and lives in scalar registers; the DATA of a step is a posting list, and its entries are spread over the 64 lanes:
//   * the words of all Nshift phases of a direction are spelled at once, one phase per lane, and their table entries fetched
//     together (one memory latency per direction instead of one per word);
//   * a posting list is read 64 blocks at a time, coalesced;
//   * the run hash -- block -> number of consecutive words that hit it -- lives in LDS.  The reference's results depend on the
//     geometry of its open-addressing table (it writes the "empty" value into live slots, which cuts probe chains), so the
//     table is kept slot for slot; but its 64 updates of a chunk are applied TOGETHER: every lane probes a snapshot, marks the
//     slots it would write in an owner array, re-walks its probe path, and the first lane whose path meets a lower lane's
//     write splits the chunk -- lanes below it commit at once, the rest probe again.  Sequential semantics, a few rounds per
//     chunk;
//   * score slots are {query tag, value} pairs in the wave's slab of HBM: nothing is cleared between queries, nothing is
//     listed for clean-up, a slot of an older query reads as zero;
//   * the bounded queues (min-heaps with a position table, 2 and ~11 entries) sit in LDS; the lanes decide in parallel whose
//     update would change a queue (almost none), and only those are applied, lowest lane first.
// No MFMA (integer look-ups and compares); the bound is memory latency per word, hidden by ~10 waves per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_blk_dev.h"
#include "../../include/spdp.h"

namespace {

typedef unsigned long long u64;
struct KV { uint32_t key; int32_t val; };           // a table slot / a queue entry {block, score}

constexpr int OWN_SLOTS = 256;                      // owner marks, slot number folded (a false meeting only splits a chunk early)
constexpr uint32_t NOBODY = 0xffffffffu;
// status words of the wave (LDS)
enum { ST_SIGN = 0, ST_MMCT = 4, ST_NHIT = 8, ST_MAXS = 12, ST_TESTWORD = 16, ST_MAXBSCR = 20, ST_QA_FRONT = 24, ST_QB_FRONT = 28,
       ST_HH_LEVEL = 32, ST_TROUBLE = 33, ST_Q_LEVEL = 40 /* 8: the position tables' levels */, ST_WORDS = 64 };

__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uniu(uint32_t x) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) x); }
__device__ __forceinline__ int lane_id() { return (int) threadIdx.x; }
__device__ __forceinline__ int first_lane(u64 m) { return __ffsll((long long) m) - 1; }
__device__ __forceinline__ int from_lane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

struct Modulus {                                    // x mod n without a division (n fixed for many x)
    uint32_t n, m;
    __device__ __forceinline__ void set(uint32_t n_) { n = n_; m = 0xffffffffu / n_; }
    __device__ __forceinline__ uint32_t of(uint32_t x) const
    {
        uint32_t r = x - __umulhi(x, m) * n;
        if (r >= n) r -= n;
        if (r >= n) r -= n;
        return r;
    }
};

// ---- score slots: {tag, value}, a slot whose tag is not the query's reads as zero ------------------------------------------
struct Score { u64 hits; u64 run; };                // hits: every word hit of the block (the fallback ranking); run: the run score
__device__ __forceinline__ int slot_get(const u64* p, uint32_t tag)
{
    const u64 x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t) (x >> 32) == tag ? (int) (uint32_t) x : 0;
}
__device__ __forceinline__ void slot_put(u64* p, uint32_t tag, int v)
{
    __hip_atomic_store(p, ((u64) tag << 32) | (uint32_t) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the run hash ----------------------------------------------------------------------------------------------------------
// Slots {key, count}; count 0 = empty.  With epochs (block numbers < 2^24) the top byte of a key names the phase it was written in
// and a slot of another phase is empty: "cleared" costs nothing 254 times out of 255.
struct RunHash {
    KV* lds; KV* glob;                              // level 0 in LDS (or null), the levels in HBM
    KV* g0; KV* grown; uint32_t b_off;                  // level 0 in the slab (when not in LDS); the larger levels: A at grown, B at grown + b_off
    Modulus size; uint32_t step_mod;
    uint32_t epoch; bool epochs;
    int level;
    const int32_t* sizes;
    __device__ __forceinline__ bool in_lds() const { return level == 0 && lds; }
    __device__ __forceinline__ KV get(uint32_t s) const { return in_lds() ? lds[s] : glob[s]; }
    __device__ __forceinline__ void put(uint32_t s, uint32_t key, int val) const
    {
        const KV kv = {epochs ? key | (epoch << 24) : key, val};
        if (in_lds()) lds[s] = kv; else glob[s] = kv;
    }
    __device__ __forceinline__ bool live(const KV& kv) const { return kv.val != 0 && (!epochs || (kv.key >> 24) == epoch); }
    __device__ __forceinline__ uint32_t key_of(const KV& kv) const { return epochs ? kv.key & 0xffffffu : kv.key; }
    __device__ __forceinline__ uint32_t home(uint32_t key) const { return size.of(key); }
    __device__ __forceinline__ uint32_t stride(uint32_t key) const { return step_mod - ((step_mod & (step_mod - 1)) ? key % step_mod : (key & (step_mod - 1))); }
    __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t u) const { s += u; while (s >= size.n) s -= size.n; return s; }
    __device__ __forceinline__ void bind(int lv)
    {
        level = lv;
        size.set((uint32_t) sizes[lv]);
        glob = g0;
        if (lv) glob = grown + ((lv & 1) ? 0u : b_off);
    }
    __device__ __forceinline__ void wipe() const                    // all lanes
    {
        const KV z = {0u, 0};
        for (uint32_t i = lane_id(); i < size.n; i += 64) { if (in_lds()) lds[i] = z; else glob[i] = z; }
    }
    __device__ __forceinline__ void new_phase()                     // all lanes, uniform
    {
        if (epochs && ++epoch < 256) return;
        wipe();
        if (epochs) epoch = 1;
        wave_sync();
    }
};

// the place of a key on a snapshot of the table: the first slot of its probe sequence that is empty or its own.  `hole`: a slot
// this lane has just emptied itself (seen as empty although the snapshot still shows it live).
struct Found { uint32_t slot; int count; bool round; };
__device__ __forceinline__ Found find_place(const RunHash& h, uint32_t key, uint32_t hole)
{
    uint32_t s = h.home(key);
    const uint32_t u = h.stride(key), s0 = s;
    for (;;) {
        if (s == hole) return {s, 0, false};
        const KV kv = h.get(s);
        if (!h.live(kv)) return {s, 0, false};
        if (h.key_of(kv) == key) return {s, kv.val, false};
        s = h.next(s, u);
        if (s == s0) return {s, 0, true};
    }
}
// the same walk, asking the owner marks: does a lower lane write a slot this lane reads?
__device__ __forceinline__ bool path_meets_lower(const RunHash& h, const uint32_t* own, uint32_t key, uint32_t hole, uint32_t me)
{
    uint32_t s = h.home(key);
    const uint32_t u = h.stride(key), s0 = s;
    for (;;) {
        if (own[s & (OWN_SLOTS - 1)] < me) return true;
        if (s == hole) return false;
        const KV kv = h.get(s);
        if (!h.live(kv) || h.key_of(kv) == key) return false;
        s = h.next(s, u);
        if (s == s0) return false;
    }
}

// ---- one lane, one posting entry, the table as it is -- only when a probe came round (the table is full): the reference grows
// the table (next prime >= twice the size, live slots re-entered in slot order) and goes on probing the NEW table from the
// slot number and with the stride it had in the old one (Dhash::map / resize, src/clib.h:298-355)
struct SlowHash {
    RunHash* h; int* st;
    __device__ __forceinline__ bool grow()
    {
        RunHash& H = *h;
        if (H.level + 1 >= SPDP_BLK_HASH_LEVELS) return false;
        RunHash old = H;
        H.bind(H.level + 1);
        for (uint32_t i = 0; i < H.size.n; ++i) H.glob[i] = KV{0u, 0};
        for (uint32_t i = 0; i < old.size.n; ++i) {
            const KV kv = old.get(i);
            if (!old.live(kv)) continue;
            const uint32_t key = old.key_of(kv);
            uint32_t s = H.home(key);
            const uint32_t u = H.stride(key), s0 = s;
            for (;;) {
                const KV c = H.get(s);
                if (!H.live(c) || H.key_of(c) == key) break;
                s = H.next(s, u);
                if (s == s0) return false;
            }
            H.put(s, key, kv.val);
        }
        return true;
    }
    // the slot of a key (claimed if empty), its count after the increment written
    __device__ __forceinline__ uint32_t bump(uint32_t key, int& count)
    {
        RunHash& H = *h;
        uint32_t s = H.home(key);
        const uint32_t u = H.stride(key), s0 = s;
        for (;;) {
            const KV kv = H.get(s);
            if (!H.live(kv)) { count = 1; break; }
            if (H.key_of(kv) == key) { count = kv.val + 1; break; }
            s = H.next(s, u);
            if (s == s0 && !grow()) { st[ST_TROUBLE] |= SPDP_BLK_TABLE | 0x400 | (h->level << 16); count = 1; break; }
        }
        H.put(s, key, count);
        return s;
    }
    // -> the block credited with the word (0xffffffff: none)
    __device__ __forceinline__ uint32_t entry(uint32_t blk, int p, bool towards_up)
    {
        RunHash& H = *h;
        int c;
        uint32_t s = bump(blk, c);
        if (c != p) {
            H.put(s, blk, 0);
            if (towards_up) s = bump(++blk, c);
            else if (blk) s = bump(--blk, c);
            else return NOBODY;
        }
        if (c == p) return blk;
        H.put(s, blk, 0);
        return NOBODY;
    }
};

// ---- a bounded best-of list: min-heap of {block, score} with a table block -> place (the reference's PrQueue_wh<BlkScr> with
// replace = true over Dhash<INT,int>(.., -1): src/clib.h:570-688).  The table is probed like the run hash and entries leave it by
// being marked empty, so a block CAN be lost from sight and entered twice: kept as it is, results depend on it.
struct BestOf {
    KV* heap; KV* place; int cap; Modulus size; uint32_t step_mod; int* front; int* trouble;
    KV* level0; KV* grown; uint32_t b_off; const int32_t* sizes; int* level;      // the table's home in LDS, its larger forms in the slab
    __device__ __forceinline__ uint32_t stride(uint32_t key) const { return step_mod - ((step_mod & (step_mod - 1)) ? key % step_mod : (key & (step_mod - 1))); }
    __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t u) const { s += u; while (s >= size.n) s -= size.n; return s; }
    __device__ __forceinline__ void bind(int lv)
    {
        size.set((uint32_t) sizes[lv]);
        place = level0;
        if (lv) place = grown + ((lv & 1) ? 0u : b_off);
    }
    // where the heap holds the block, as far as the table knows (-1: not; -2: the probe came round -- the table is full and the
    // reference grows it at this very lookup: the caller takes the one-lane path)
    __device__ __forceinline__ int where(uint32_t key) const
    {
        uint32_t s = size.of(key);
        const uint32_t u = stride(key), s0 = s;
        for (;;) {
            const KV kv = place[s];
            if (kv.val == -1) return -1;
            if (kv.key == key) return kv.val;
            s = next(s, u);
            if (s == s0) return -2;
        }
    }
    // one lane: the table of the next size, the live entries re-entered in slot order (Dhash::resize, src/clib.h:341-355)
    __device__ __forceinline__ bool grow()
    {
        if (*level + 1 >= SPDP_BLK_HASH_LEVELS) { *trouble |= SPDP_BLK_TABLE; return false; }
        const KV* old = place;
        const uint32_t n_old = size.n;
        bind(++*level);
        for (uint32_t i = 0; i < size.n; ++i) place[i] = KV{0u, -1};
        for (uint32_t i = 0; i < n_old; ++i) {
            const KV kv = old[i];
            if (kv.val == -1) continue;
            uint32_t s = size.of(kv.key);
            const uint32_t u = stride(kv.key), s0 = s;
            while (place[s].val != -1 && place[s].key != kv.key) { s = next(s, u); if (s == s0) { *trouble |= SPDP_BLK_TABLE; return false; } }
            place[s] = kv;
        }
        return true;
    }
    // one lane: the slot of a key -- its own, or the empty one its probe sequence meets first; a probe that comes round grows the
    // table and goes on in the new one from the slot number and with the stride it had (Dhash::map)
    __device__ __forceinline__ uint32_t slot_for(uint32_t key)
    {
        uint32_t s = size.of(key);
        const uint32_t u = stride(key), s0 = s;
        for (;;) {
            const KV kv = place[s];
            if (kv.val == -1 || kv.key == key) return s;
            s = next(s, u);
            if (s == s0 && !grow()) return s;
        }
    }
    __device__ __forceinline__ void note(uint32_t key, int at) { const uint32_t s = slot_for(key); place[s] = KV{key, at}; }     // one lane
    __device__ __forceinline__ void seat(int k, KV v) { heap[k] = v; note(v.key, k); }
    __device__ __forceinline__ void sink(int k)
    {
        const KV v = heap[k];
        const int n = *front;
        for (int c; (c = 2 * k + 1) < n; k = c) {
            if (c + 1 < n && heap[c + 1].val < heap[c].val) ++c;
            if (!(heap[c].val < v.val)) break;
            seat(k, heap[c]);
        }
        seat(k, v);
    }
    __device__ __forceinline__ void rise(int k)
    {
        const KV v = heap[k];
        while (k > 0) {
            const int up = (k - 1) / 2;
            if (!(v.val < heap[up].val)) break;
            seat(k, heap[up]);
            k = up;
        }
        seat(k, v);
    }
    // would offering {key, score} change anything?  (all lanes, their own offers, on the list as it is)
    __device__ __forceinline__ bool matters(uint32_t key, int score) const
    {
        const int at = where(key);
        if (at == -2) return true;                      // (the lookup itself changes the table)
        if (at < 0 && *front < cap) return true;
        return heap[at < 0 ? 0 : at].val < score;
    }
    __device__ __forceinline__ void offer(uint32_t key, int score)  // one lane
    {
        int at;                                         // (a lookup, not a claim: an empty slot stays as it is)
        { const uint32_t s = slot_for(key); const KV kv = place[s]; at = (kv.val != -1 && kv.key == key) ? kv.val : -1; }
        if (at < 0) {
            if (*front < cap) { const int k = (*front)++; heap[k] = KV{key, score}; rise(k); return; }
            at = 0;
        }
        if (heap[at].val < score) {
            note(heap[at].key, -1);
            heap[at] = KV{key, score};
            sink(at);
        }
    }
    // the offers of the lanes in `who`, in lane order
    __device__ __forceinline__ void offer_in_order(u64 who, uint32_t key, int score)
    {
        const int me = lane_id();
        int from = 0;
        for (;;) {
            const bool mine = ((who >> me) & 1) && me >= from && matters(key, score);
            const u64 m = __ballot(mine);
            if (!m) break;
            const int l = first_lane(m);
            if (me == l) offer(key, score);
            wave_sync();
            bind(*level);                               // (the offer may have grown the table)
            from = l + 1;
        }
    }
    __device__ __forceinline__ void reset()                         // all lanes
    {
        if (lane_id() == 0) { *front = 0; *level = 0; }
        bind(0);
        for (uint32_t i = lane_id(); i < size.n; i += 64) place[i] = KV{0u, -1};
    }
};

// ---- the words of the query --------------------------------------------------------------------------------------------------
struct Word { int score; uint32_t off0, off1; int len0, len1; };      // score < 0: no usable word, 0: a ubiquitous one; up to two posting lists

struct Speller {
    const BlkDev* ix; const uint8_t* q; int q_len, right;
    const uint8_t* res;                                 // the query's residues as the index's alphabet sees them (LDS), or null: too long
    __device__ __forceinline__ uint32_t residue(int i) const
    {
        if (i < 0 || i >= q_len) return 255u;
        if (res) return res[i];
        const int c = q[i];
        return c < ix->convts ? ix->convtab[c] : 255u;
    }
    // the word of pattern k at ss: read left to right (d < 2) or as the other strand sees it (d >= 2).  A word that meets an
    // unusable residue keeps the digits read so far (the reference looks its table entries up all the same)
    __device__ __forceinline__ uint32_t spell(int ss, int d, int k, int& good) const
    {
        const int32_t* bp = ix->bitpat + ix->pat_off[k];
        const int weight = bp[0];
        const int32_t* at = bp + 3 + (d >= 2 ? weight : 0);
        const uint32_t base = (uint32_t) ix->nalpha;
        uint32_t fwd = 0, rev = 0, unit = 1;        // rev: digits in rising order
        int i = 0;
        for ( ; i < weight; ++i) {
            const uint32_t c = residue(ss + at[i]);
            if (c >= base) break;
            fwd = fwd * base + c;
            rev += (ix->drna ? 3u - c : c) * unit;
            unit *= base;
        }
        good = i;
        if (d < 2) return fwd;
        // the digits sit at the top of the word: scaled by base ^ (weight - good)
        return good ? rev * ((uint32_t) ix->tabsize / unit) : 0u;
    }
    __device__ __forceinline__ Word at(int ss, int d) const
    {
        const BlkDev& X = *ix;
        Word w; w.score = -1; w.off0 = w.off1 = 0; w.len0 = w.len1 = 0;
        const uint32_t tab = (uint32_t) X.tabsize;
        if (X.kk == 1) {
            int good;
            const uint32_t x = spell(ss, d, 0, good);
            const int32_t lp = X.blkp[x];
            if (!lp) { w.score = 0; return w; }
            if (good != X.bitpat[X.pat_off[0]]) return w;
            w.score = X.wscr[x];
            w.off0 = (uint32_t) lp - 1; w.len0 = X.nblk[x];
            return w;
        }
        int n_ok = 0, sum = 0;
        for (int k = 0; k < X.kk; ++k) {
            const int32_t* bp = X.bitpat + X.pat_off[k];
            if (ss >= right - bp[1]) break;             // (no room for this pattern, nor for the later ones)
            int good;
            const uint32_t x = spell(ss, d, k, good);
            if (x >= tab || X.wscr[x] < 0 || good < bp[0]) continue;
            const int32_t lp = X.blkp[x];
            if (!lp) continue;
            ++n_ok; sum += X.wscr[x];
            if (k == 0 && X.kk > 1) { w.off0 = (uint32_t) lp - 1; w.len0 = X.nblk[x]; }
            else if (k == 1 && X.kk > 2) { w.off1 = (uint32_t) lp - 1; w.len1 = X.nblk[x]; }     // (the last pattern scores but does not vote)
        }
        if (n_ok) w.score = (int) ((double) sum / X.app_c);
        return w;
    }
};

__device__ __forceinline__ int random_expectation(const BlkDev& X, uint32_t mmc)
{
    if (mmc < 128) return X.rscrtab[mmc];
    if (X.rbscoef == 0) return (int) X.rbscons;
    const double x = (double) (mmc + 1);
    return (int) (X.rbscoef * (X.gdb ? log(x) : sqrt(x)) + X.rbscons);
}

// ---- chromosomes: the one that holds a block ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t first_block(const BlkDev& X, int c) { return (uint32_t) X.chr[2 * c + 1]; }
__device__ __forceinline__ int chromosome_of(const BlkDev& X, uint32_t blk)
{
    // the reference brackets the answer with a linear estimate before it bisects (src/blksrc.cc:1985-2002); the bracket decides
    // which chromosome a block on a boundary of equal first blocks is given to, so it is kept
    int lo = (int) (X.bclw + X.bcce * (blk - 1)) - 1, hi = (int) (X.bcup + X.bcce * (blk - 1)) + 1;
    if (lo < 0) lo = 0;
    if (hi > X.n_chr) hi = X.n_chr;
    if (first_block(X, lo) > blk) lo = 0;
    if (first_block(X, hi) < blk) hi = X.n_chr;
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (first_block(X, mid) > blk) hi = mid;
        else if (first_block(X, mid + 1) > blk) return mid;
        else lo = mid;
    }
    return first_block(X, hi) > blk ? lo : hi;
}

// ---- the wave --------------------------------------------------------------------------------------------------------------
struct Wave {
    const BlkDev* ix;
    int* st; int* as; uint32_t* own;
    // the eight best-of lists: by word hits (h) and by run score (r), a list per direction
    KV* qh_mem; KV* qr_mem; KV* qh_grown; KV* qr_grown;
    int qh_words, qr_words, qh_cap, qr_cap, qh_grown_words, qr_grown_words; uint32_t qh_step, qr_step;
    __device__ __forceinline__ BestOf hits_list(int d) const
    {
        BestOf Q;
        KV* m = qh_mem + (size_t) d * qh_words;
        Q.cap = qh_cap; Q.heap = m; Q.level0 = m + Q.cap + 1; Q.sizes = ix->ha_sizes;
        KV* g = qh_grown + (size_t) d * qh_grown_words;
        Q.grown = g; Q.b_off = (uint32_t) Q.sizes[SPDP_BLK_HASH_LEVELS - 1];
        Q.step_mod = qh_step; Q.front = st + ST_QA_FRONT + d; Q.trouble = st + ST_TROUBLE; Q.level = st + ST_Q_LEVEL + d;
        Q.bind(*Q.level);
        return Q;
    }
    __device__ __forceinline__ BestOf run_list(int d) const
    {
        BestOf Q;
        KV* m = qr_mem + (size_t) d * qr_words;
        Q.cap = qr_cap; Q.heap = m; Q.level0 = m + Q.cap + 1; Q.sizes = ix->hb_sizes;
        KV* g = qr_grown + (size_t) d * qr_grown_words;
        Q.grown = g; Q.b_off = (uint32_t) Q.sizes[SPDP_BLK_HASH_LEVELS - 1];
        Q.step_mod = qr_step; Q.front = st + ST_QB_FRONT + d; Q.trouble = st + ST_TROUBLE; Q.level = st + ST_Q_LEVEL + 4 + d;
        Q.bind(*Q.level);
        return Q;
    }
    RunHash hh;
    Score* score; uint32_t* stage; int32_t* scratch; uint32_t* header;
    uint8_t* res; int res_cap;
    uint32_t* mrg;                                      // two posting lists and their merge (512 words)
    uint32_t* pre;                                      // the first 64 entries of the posting lists of a direction's phases (fetched together)
    uint32_t tag;
};

struct Pair { int32_t bscr, chr; uint32_t lb, rb, ub, db, zl, zr; int32_t rvs; };

// the significant blocks of one strand, both ends, in genome order, joined into spans: a span is a block or a stretch of blocks
// that belong to one gene as far as their distances say (extract_to_work).  -> spans as (first, last), n
__device__ __forceinline__ int spans_of_strand(const Wave& W, int f, uint32_t* sites, uint32_t* first, uint32_t* last)
{
    const BlkDev& X = *W.ix;
    const int d = 2 * f, e = d + 1;
    const int nd = W.st[ST_SIGN + d], ne = W.st[ST_SIGN + e];
    int n = 0;
    for (int i = 0; i < nd; ++i) sites[n++] = W.run_list(d).heap[i].key << 1;
    for (int i = 0; i < ne; ++i) sites[n++] = (W.run_list(e).heap[i].key << 1) | 1u;
    if (!n) return 0;
    for (int a = 1; a < n; ++a) {                       // ascending by (block, end)
        const uint32_t v = sites[a];
        int b = a;
        for ( ; b > 0 && sites[b - 1] > v; --b) sites[b] = sites[b - 1];
        sites[b] = v;
    }
    if (f) {                                            // on the other strand the right-end hit of a block comes first
        uint32_t cur = sites[0] >> 1;
        for (int i = 1; i < n; ++i) {
            if ((sites[i] >> 1) == cur) { const uint32_t t = sites[i - 1]; sites[i - 1] = sites[i]; sites[i] = t; }
            else cur = sites[i] >> 1;
        }
    }
    int n_spans = 0;
    uint32_t open = sites[0] >> 1, prev = open;
    int prev_end = (int) (sites[0] & 1) ^ f, prev_chr = n > 1 ? chromosome_of(X, prev) : 0;
    for (int i = 1; i < n; ++i) {
        const uint32_t b = sites[i] >> 1;
        const int end = (int) (sites[i] & 1) ^ f, c = chromosome_of(X, b);
        const int gap = (int) (b - prev);
        const bool joined = c == prev_chr && (gap < 2 || (!prev_end && end && gap <= X.maxblock) || (prev_end == end && gap <= X.extblock));
        if (!joined) { first[n_spans] = open; last[n_spans++] = prev; open = b; }
        prev = b; prev_end = end; prev_chr = c;
    }
    first[n_spans] = open; last[n_spans++] = prev;
    return n_spans;
}

// TestOutput's list of candidate block pairs: every span grown over the neighbouring blocks that scored at all, its score the
// sum of its blocks' run scores from both ends; best first, at most ncand (one lane)
__device__ __forceinline__ int candidate_pairs(const Wave& W, Pair* pairs)
{
    const BlkDev& X = *W.ix;
    const int nseg = X.nseg, cap = X.ncand;
    uint32_t* sites = (uint32_t*) (pairs + cap + 2);
    uint32_t* first = sites + 2 * cap + 2;
    uint32_t* last = first + 2 * cap + 2;
    int n_pairs = 0;
    const uint32_t reach = (uint32_t) X.extblock;
    for (int f = 0; f < 2; ++f) {
        const int n_spans = spans_of_strand(W, f, sites, first, last);
        const u64* sd = &W.score[(size_t) (2 * f) * nseg].run;
        const u64* se = &W.score[(size_t) (2 * f + 1) * nseg].run;
        auto both = [&](uint32_t r) { return slot_get(sd + 2 * (size_t) r, W.tag) + slot_get(se + 2 * (size_t) r, W.tag); };
        uint32_t taken_to = 0;                          // blocks below belong to the span before
        for (int i = 0; i < n_spans; ++i) {
            Pair P;
            P.rvs = f; P.lb = first[i]; P.rb = last[i];
            const uint32_t next_span = i + 1 < n_spans ? first[i + 1] : 0x3fffffffu;
            P.chr = chromosome_of(X, P.rb);
            P.zl = first_block(X, P.chr); P.zr = first_block(X, P.chr + 1) - 1;
            P.bscr = 0;
            for (uint32_t r = P.lb; r <= P.rb; ++r) P.bscr += both(r);
            uint32_t r = P.lb;
            uint32_t stop = r > reach ? r - reach : 0;
            if (P.zl > stop) stop = P.zl;
            if (taken_to > stop) stop = taken_to;
            while (r && --r >= stop) { const int s = both(r); if (!s) break; P.lb = r; P.bscr += s; }
            P.ub = r > reach ? r - reach : 0;
            if (P.zl > P.ub) P.ub = P.zl;
            r = P.rb;
            stop = r + reach;
            if (P.zr < stop) stop = P.zr;
            if (next_span < stop) stop = next_span;
            while (++r < stop) { const int s = both(r); if (!s) break; P.rb = r; P.bscr += s; }
            P.db = r + reach < P.zr ? r + reach : P.zr;
            taken_to = P.rb + 1;
            // into the list: behind everything that scores at least as much; the list keeps `cap`
            int at = n_pairs < cap ? n_pairs : cap;
            pairs[at] = P;
            for ( ; at > 0 && pairs[at].bscr > pairs[at - 1].bscr; --at) { const Pair t = pairs[at]; pairs[at] = pairs[at - 1]; pairs[at - 1] = t; }
            if (n_pairs < cap) ++n_pairs;
        }
    }
    return n_pairs;
}

struct Record {                                         // one query's record, written by lane 0 unless said otherwise
    int32_t* out; int cap, n; bool cut;
    __device__ __forceinline__ void put(int v) { if (n < cap) out[n] = v; else cut = true; ++n; }
};

// the state at a TestOutput call: counters, the significant blocks per direction in their lists' own order, the candidate pairs,
// the run scores around the pairs on their strands (what FindHsp looks at when it moves a pair's ends, src/blksrc.cc:2408-2460)
__device__ __forceinline__ void write_state(Wave& W, Record& R)
{
    const BlkDev& X = *W.ix;
    const int me = lane_id(), nseg = X.nseg;
    Pair* pairs = (Pair*) W.scratch;
    int np = 0, n = R.n;
    if (me == 0) {
        for (int k = ST_SIGN; k < ST_SIGN + 20; ++k) R.put(W.st[k]);
        for (int d = 0; d < 4; ++d) {
            const int f = W.st[ST_QB_FRONT + d];
            R.put(f);
            for (int i = 0; i < f; ++i) { { const KV kv = W.run_list(d).heap[i]; R.put((int) kv.key); R.put(kv.val); } }
        }
        np = candidate_pairs(W, pairs);
        R.put(np);
        for (int i = 0; i < np; ++i) {
            const Pair& b = pairs[i];
            R.put(b.bscr); R.put(b.chr); R.put((int) b.lb); R.put((int) b.rb); R.put((int) b.ub); R.put((int) b.db);
            R.put((int) b.zl); R.put((int) b.zr); R.put(b.rvs);
        }
        n = R.n;
    }
    wave_sync();
    np = uni(np); n = uni(n);
    const int at_count = n++;
    int n_runs = 0;
    // (FindHsp may move a pair's end up to three times -- a protein query looks again at the region it has grown -- by at most
    // max(ExtBlockL, ExtBlock) blocks each time, and reads the run scores on its way)
    const uint32_t el = 4u * (uint32_t) (X.extblockl > X.extblock ? X.extblockl : X.extblock);
    for (int i = 0; i < np; ++i) {
        const Pair b = pairs[i];
        uint32_t lo = b.lb > el ? b.lb - el : 0, hi = b.rb + el;
        if (lo < b.zl) lo = b.zl;
        if (hi > b.zr) hi = b.zr;
        for (int dd = 2 * b.rvs; dd < 2 * b.rvs + 2; ++dd)
            for (uint32_t r0 = lo; r0 <= hi; r0 += 64) {
                const uint32_t r = r0 + me;
                int v = 0;
                if (r <= hi) {
                    bool seen = false;                  // (a block near two pairs is reported with the first)
                    for (int j = 0; j < i && !seen; ++j) {
                        const Pair o = pairs[j];
                        if (o.rvs != b.rvs) continue;
                        uint32_t olo = o.lb > el ? o.lb - el : 0, ohi = o.rb + el;
                        if (olo < o.zl) olo = o.zl;
                        if (ohi > o.zr) ohi = o.zr;
                        seen = r >= olo && r <= ohi;
                    }
                    if (!seen) v = slot_get(&W.score[(size_t) dd * nseg + r].run, W.tag);
                }
                const u64 m = __ballot(v != 0);
                if (v) {
                    const int k = n + 2 * __popcll(m & ((1ull << me) - 1));
                    if (k + 1 < R.cap) { R.out[k] = (int) (r | ((uint32_t) dd << 28)); R.out[k + 1] = v; }
                }
                n += 2 * __popcll(m); n_runs += __popcll(m);
            }
    }
    if (me == 0) {
        if (at_count < R.cap) R.out[at_count] = n_runs;
        R.n = n;
        if (n > R.cap) R.cut = true;
    }
}

// the posting entries of one word, the lists merged when there are two (ascending, a block of both counted once): where they
// are and how many (one lane merges into the staging area; one list is read where it lies)
__device__ __forceinline__ const uint32_t* entries_of(const Wave& W, const Word& w, int& n)
{
    const BlkDev& X = *W.ix;
    if (!w.len1) { n = w.len0; return X.blkb + w.off0; }
    if (!w.len0) { n = w.len1; return X.blkb + w.off1; }
    const int me = lane_id();
    if (w.len0 <= 128 && w.len1 <= 128) {
        // both lists in LDS (A at 0, B at 128), every entry ranked among the other list's by bisection, the merged sequence (an entry
        // of both lists once) written at 256 and closed up in place
        uint32_t* A = W.mrg; uint32_t* B = W.mrg + 128; uint32_t* M = W.mrg + 256;
        int na = w.len0, nb = w.len1;
        for (int i = me; i < 128; i += 64) {
            A[i] = i < na ? X.blkb[w.off0 + i] : 0xffffffffu;
            B[i] = i < nb ? X.blkb[w.off1 + i] : 0xffffffffu;
        }
        lds_sync();
        for (int i = me; i < 128; i += 64) {            // (a zero ends a list)
            const u64 za = __ballot(i < na && A[i] == 0), zb = __ballot(i < nb && B[i] == 0);
            if (za) na = min(na, i - me + first_lane(za));
            if (zb) nb = min(nb, i - me + first_lane(zb));
        }
        auto below = [](const uint32_t* L, int n, uint32_t x, bool or_equal) {        // entries of L below x (or up to x)
            int lo = 0, hi = n;
            while (lo < hi) { const int mid = (lo + hi) >> 1; const uint32_t v = L[mid]; if (v < x || (or_equal && v == x)) lo = mid + 1; else hi = mid; }
            return lo;
        };
        for (int i = me; i < 256; i += 64) M[i] = 0u;
        lds_sync();
        for (int i = me; i < 128; i += 64) {
            if (i < na) M[i + below(B, nb, A[i], false)] = A[i];
            if (i < nb) {
                const int r = below(A, na, B[i], true);
                const bool twice = r > 0 && A[r - 1] == B[i];
                if (!twice) M[i + r] = B[i];            // (an entry of both lists: A's copy stands, this slot stays empty)
            }
        }
        lds_sync();
        int k = 0;
        for (int i0 = 0; i0 < na + nb; i0 += 64) {
            const uint32_t v = i0 + me < na + nb ? M[i0 + me] : 0u;
            const u64 m = __ballot(v != 0);
            lds_sync();
            if (v) M[k + __popcll(m & ((1ull << me) - 1))] = v;       // (closing up never overtakes a slot still to be read: k <= i0)
            k += __popcll(m);
            lds_sync();
        }
        n = k;
        return M;
    }
    int k = 0;
    if (me == 0) {
        const uint32_t* a = X.blkb + w.off0; const uint32_t* b = X.blkb + w.off1;
        int i = 0, j = 0;
        while (i < w.len0 || j < w.len1) {
            const uint32_t x = i < w.len0 ? a[i] : 0xffffffffu, y = j < w.len1 ? b[j] : 0xffffffffu;
            const uint32_t m = x < y ? x : y;
            if (!m) break;                              // (a zero ends a list)
            W.stage[k++] = m;
            i += x == m; j += y == m;
        }
    }
    wave_sync();
    n = uni(k);
    return W.stage;
}

// One word of direction d, phase sft: all its posting entries.  -> how many of them the word's run went on in
__device__ __forceinline__ int vote_of_word(Wave& W, const Word& w, int d, int sft, int p, int threshold, bool fetched)
{
    const BlkDev& X = *W.ix;
    const int me = lane_id(), nseg = X.nseg;
    const bool up = d & 1;                              // scanning from the right end: the neighbour is the next block
    int n_all;
    const uint32_t* list = entries_of(W, w, n_all);
    Score* sc = W.score + (size_t) d * nseg;
    RunHash& H = W.hh;
    int went_on = 0;
    for (int c0 = 0; c0 < n_all; c0 += 64) {
        bool have = c0 + me < n_all;
        const uint32_t blk = have ? ((fetched && c0 == 0) ? W.pre[sft * 64 + me] : list[c0 + me]) : 0u;
        const u64 zeros = __ballot(have && blk == 0);
        if (zeros) { have = have && me < first_lane(zeros); n_all = 0; }        // a zero ends the list
        // every hit counts for the block's total, and the totals' best-of list sees every one of them
        int total = 0, run_ahead = 0;                   // (the block's run score fetched in the same round trip: it is the one credited, mostly)
        if (have) {
            const u64 xh = __hip_atomic_load(&sc[blk].hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 xr = __hip_atomic_load(&sc[blk].run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total = ((uint32_t) (xh >> 32) == W.tag ? (int) (uint32_t) xh : 0) + w.score;
            run_ahead = (uint32_t) (xr >> 32) == W.tag ? (int) (uint32_t) xr : 0;
            slot_put(&sc[blk].hits, W.tag, total);
        }
        W.hits_list(d).offer_in_order(__ballot(have), blk, total);
        bool first_group = true;
        // the run hash: snapshot, marks, commit the lanes below the first meeting
        u64 pend = __ballot(have);
        while (pend) {
            const bool mine = (pend >> me) & 1;
            Found a = {0, 0, false}, b = {0, 0, false};
            bool ask = false;
            uint32_t nb = blk;
            int ca = 0, cb = 0;
            if (mine) {
                a = find_place(H, blk, NOBODY);
                ca = a.count + 1;
                ask = ca != p && (up || blk != 0);
                if (ask && !a.round) { nb = up ? blk + 1 : blk - 1; b = find_place(H, nb, a.slot); cb = b.count + 1; }
            }
            const u64 full = __ballot(mine && (a.round || b.round));
            const u64 upto = full ? ((1ull << first_lane(full)) - 1) : ~0ull;   // lanes from the first full probe on: later
            const bool in = mine && ((upto >> me) & 1);
            if (in) {
                atomicMin(&W.own[a.slot & (OWN_SLOTS - 1)], (uint32_t) me);
                if (ask) atomicMin(&W.own[b.slot & (OWN_SLOTS - 1)], (uint32_t) me);
            }
            lds_sync();
            bool meets = false;
            if (in) {
                meets = path_meets_lower(H, W.own, blk, NOBODY, (uint32_t) me);
                if (!meets && ask) meets = path_meets_lower(H, W.own, nb, a.slot, (uint32_t) me);
            }
            const u64 met = __ballot(meets);
            lds_sync();
            if (in) { W.own[a.slot & (OWN_SLOTS - 1)] = NOBODY; if (ask) W.own[b.slot & (OWN_SLOTS - 1)] = NOBODY; }
            const u64 below = met ? ((1ull << first_lane(met)) - 1) : ~0ull;
            const bool go = in && ((below >> me) & 1);
            uint32_t credited = NOBODY;
            if (go) {
                if (ca == p) { H.put(a.slot, blk, ca); credited = blk; }
                else {
                    if (!ask || b.slot != a.slot) H.put(a.slot, blk, 0);
                    if (ask) { H.put(b.slot, nb, cb == p ? cb : 0); if (cb == p) credited = nb; }
                }
            }
            u64 done = __ballot(go);
            if (H.in_lds()) lds_sync(); else wave_sync();
            if (!done) {
                // the lowest pending lane met a full table: its entry alone, with the table growing under it
                const int l = first_lane(pend);
                if (me == l) { RunHash T = H; SlowHash S = {&T, W.st}; credited = S.entry(blk, p, up); W.header[1] += 1; W.header[2] += (uint32_t) (T.level - H.level); W.st[ST_HH_LEVEL] = T.level; }
                wave_sync();
                const int lv = uni(W.st[ST_HH_LEVEL]);
                if (lv != H.level) H.bind(lv);              // (every lane alike: the table's place and size stay uniform)
                done = 1ull << l;
            }
            pend &= ~done;
            // the credited blocks of this group (distinct: two lanes on one block meet in the hash): run scores, the best run
            // of the direction, the best-of list of the blocks above the random expectation
            const bool cr = credited != NOBODY && ((done >> me) & 1);
            int run = 0;
            if (cr) {
                // (within a group the credited blocks are distinct; a value fetched ahead is good for the chunk's first group only: a
                // later group may follow a lane that credited this block as ITS neighbour)
                run = ((first_group && credited == blk) ? run_ahead : slot_get(&sc[credited].run, W.tag)) + w.score;
                slot_put(&sc[credited].run, W.tag, run);
            }
            first_group = false;
            const u64 crm = __ballot(cr);
            if (crm) {
                went_on += __popcll(crm);
                int best = cr ? run : (int) 0x80000000;
                for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(best, o); best = t > best ? t : best; }
                if (best > W.st[ST_MAXBSCR + d]) { if (me == 0) { W.st[ST_MAXBSCR + d] = best; W.st[ST_MAXS + d] = sft; } }
                const u64 sig = __ballot(cr && run >= threshold);
                if (sig) {
                    W.run_list(d).offer_in_order(sig, credited, run);
                    if (me == 0) W.st[ST_SIGN + d] = W.st[ST_QB_FRONT + d];
                }
                lds_sync();
            }
        }
    }
    return went_on;
}

// the scan of one query up to its stop_at-th TestOutput call.  -> 0: it ends before that call; 1: reached; 2: reached, and it is
// the call the reference makes behind its scan (TestOutput(1))
__device__ __forceinline__ int scan(Wave& W, const uint8_t* q, int q_len, int left, int right, int stop_at, int& calls)
{
    const BlkDev& X = *W.ix;
    const int me = lane_id(), nshift = X.nshift;
    calls = 0;
    if (me < ST_WORDS) W.st[me] = 0;
    const int width0 = X.bitpat[X.pat_off[0] + 1];
    const int qlen = right - left;
    if (qlen - (nshift + width0) < 1 || nshift > SPDP_BLK_MAX_SHIFT) return 0;
    for (int d = 0; d < 4; ++d) { W.hits_list(d).reset(); W.run_list(d).reset(); }
    if (W.hh.level) W.hh.bind(0);
    if (W.hh.epochs) W.hh.epoch = 255;                  // (the first phase wipes the table)
    // where the phases stand: Nshift consecutive start points at the left end, the last Nshift full words at the right end, the
    // phase of a right-end point chosen so that the two ends of a phase are a multiple of Nshift apart
    if (me < nshift) {
        const int ts = right - (width0 + nshift) - 1;
        const int ph = (((right - (width0 + nshift)) - left) % nshift + me) % nshift;
        W.as[0 * 32 + me] = W.as[2 * 32 + me] = left + me;
        W.as[1 * 32 + ph] = W.as[3 * 32 + ph] = ts + me;
    }
    lds_sync();
    const bool res_fits = q_len <= W.res_cap;
    if (res_fits) for (int i = me; i < q_len; i += 64) { const int c = q[i]; W.res[i] = c < X.convts ? X.convtab[c] : (uint8_t) 255; }
    lds_sync();
    Speller sp = {&X, q, q_len, right, res_fits ? W.res : nullptr};
    const bool is_short = qlen < X.shortquery;
    const int base = X.rscrtab[0];
    int met_ends = 0;                                   // bit f: the two scans of strand f have met
    int nohit = 0, sigpr = 0, notry = 0, c = qlen / (2 * nshift) - 1;
    uint32_t nmmc = 0;
    while (!met_ends) {
        int totalsign = 0;
        const int threshold = random_expectation(X, nmmc);
        for (int d = 0; d < 4; ++d) {
            if (met_ends >> (d >> 1) & 1) continue;
            const bool up = d & 1;
            const int e = d ^ 1;
            // the first word of every phase of this direction, a phase per lane
            Word mine; mine.score = -1; mine.off0 = mine.off1 = 0; mine.len0 = mine.len1 = 0;
            if (me < nshift) mine = sp.at(W.as[d * 32 + me], d);
            // ... and the head of every phase's posting list, all in flight together (one list: nothing to merge)
            const bool fetch = nshift <= SPDP_BLK_PRE_PHASES;
            if (fetch) {
                uint32_t head[SPDP_BLK_PRE_PHASES];
                #pragma unroll
                for (int k = 0; k < SPDP_BLK_PRE_PHASES; ++k) {
                    const int n0 = from_lane(mine.len0, k), n1 = from_lane(mine.len1, k), sc0 = from_lane(mine.score, k);
                    const uint32_t o0 = (uint32_t) from_lane((int) mine.off0, k);
                    head[k] = (k < nshift && sc0 > 0 && !n1 && me < n0) ? X.blkb[o0 + me] : 0u;
                }
                #pragma unroll
                for (int k = 0; k < SPDP_BLK_PRE_PHASES; ++k) if (k < nshift) W.pre[k * 64 + me] = head[k];
                lds_sync();
            }
            int maxp = 0;
            for (int sft = 0; sft < nshift; ++sft) {
                const int ms = is_short ? (up ? left : right) : W.as[e * 32 + sft];
                int cscr = 0, more = 0, p = 0;
                W.hh.new_phase();
                bool first = true;
                do {
                    const int ss = W.as[d * 32 + sft];
                    lds_sync();
                    if (me == 0) W.as[d * 32 + sft] = ss + (up ? -nshift : nshift);
                    lds_sync();
                    if (up ^ (ss >= ms)) { met_ends |= 1 << (d >> 1); break; }
                    Word w;
                    if (first) {
                        w.score = from_lane(mine.score, sft);
                        w.off0 = (uint32_t) from_lane((int) mine.off0, sft); w.len0 = from_lane(mine.len0, sft);
                        w.off1 = (uint32_t) from_lane((int) mine.off1, sft); w.len1 = from_lane(mine.len1, sft);
                    } else {
                        w = sp.at(ss, d);
                        w.score = uni(w.score); w.off0 = uniu(w.off0); w.off1 = uniu(w.off1); w.len0 = uni(w.len0); w.len1 = uni(w.len1);
                    }
                    const bool was_first = first;
                    first = false;
                    if (w.score < 0) break;
                    if (me == 0) W.st[ST_TESTWORD + d] += X.kk;
                    if (w.score == 0) { more = 1; continue; }
                    ++p; cscr += w.score;
                    more = vote_of_word(W, w, d, sft, p, threshold, fetch && was_first && !w.len1);
                } while (more && cscr < base);
                if (p > maxp) maxp = p;
                lds_sync();
                if (W.st[ST_MAXS + d] == sft) nohit = !more;
            }
            lds_sync();
            if (me == 0) { W.st[ST_MMCT + d] += nohit; W.st[ST_NHIT + d] += maxp; }
            lds_sync();
            totalsign += W.st[ST_SIGN + d];
        }
        const bool pair_now = (W.st[ST_SIGN + 0] && W.st[ST_SIGN + 1]) || (W.st[ST_SIGN + 2] && W.st[ST_SIGN + 3]);
        if (pair_now) ++sigpr;
        if ((++nmmc % (uint32_t) X.maxmmc == 0 && totalsign) || sigpr > X.minsigpr) {
            if (calls++ == stop_at) return 1;
            c = 0;
            if (++notry > X.minsigpr) return 0;
        }
    }
    if (!((W.st[ST_SIGN + 0] && W.st[ST_SIGN + 1]) || (W.st[ST_SIGN + 2] && W.st[ST_SIGN + 3]))) {
        // no significant pair: the blocks with most word hits stand in, with their run scores
        c = -1;
        if (me == 0) {
            for (int d = 0; d < 4; ++d)
                for (int i = 0; i < X.nascr && i < W.st[ST_QA_FRONT + d]; ++i) {
                    const uint32_t key = W.hits_list(d).heap[i].key;
                    if (!key) continue;
                    W.run_list(d).offer(key, slot_get(&W.score[(size_t) d * X.nseg + key].run, W.tag));
                    W.st[ST_SIGN + d] = W.st[ST_QB_FRONT + d];
                    W.st[ST_WORDS - 1] = 1;
                }
        }
        wave_sync();
        if (W.st[ST_WORDS - 1]) c = 1;
    }
    if (c != -1 && calls++ == stop_at) return 2;
    return 0;
}

// (four waves per SIMD: the kernel waits for memory three quarters of its time, more waves in flight pay for a few spilled registers)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) spdp_blk_vote_wave(BlkVoteArgs A)
{
    extern __shared__ uint32_t lds[];
    const BlkDev& X = *A.ix;
    const int me = lane_id();
    Wave W;
    W.ix = &X;
    // ---- LDS
    uint32_t* l = lds;
    W.st = (int*) l; l += ST_WORDS;
    W.as = (int*) l; l += 4 * 32;
    W.own = l; l += OWN_SLOTS;
    for (int i = me; i < OWN_SLOTS; i += 64) W.own[i] = NOBODY;
    W.qh_cap = X.nascr; W.qh_words = X.nascr + 1 + X.ha_size1; W.qh_step = (uint32_t) X.ha_size2;
    W.qh_mem = (KV*) l; l += 2 * 4 * (size_t) W.qh_words;
    W.qr_cap = X.ncand; W.qr_words = X.ncand + 1 + X.hb_size1; W.qr_step = (uint32_t) X.hb_size2;
    W.qr_mem = (KV*) l; l += 2 * 4 * (size_t) W.qr_words;
    // ---- the wave's slab
    uint8_t* g = A.slabs + (size_t) blockIdx.x * A.slab_bytes;
    W.header = (uint32_t*) g; g += 16;
    W.score = (Score*) g; g += sizeof(Score) * (4 * (size_t) X.nseg + 2);
    RunHash& H = W.hh;
    H.sizes = X.hh_sizes; H.step_mod = (uint32_t) X.hh_size2;
    W.res = (uint8_t*) l; W.res_cap = A.res_cap; l += A.res_cap / 4;
    W.pre = l; l += 64 * SPDP_BLK_PRE_PHASES;
    W.mrg = l; l += 512;
    H.lds = A.hh_in_lds ? (KV*) l : nullptr;
    H.g0 = (KV*) g; if (!A.hh_in_lds) g += sizeof(KV) * (size_t) X.hh_sizes[0];
    H.grown = (KV*) g; H.b_off = (uint32_t) X.hh_sizes[SPDP_BLK_HASH_LEVELS - 1];
    g += sizeof(KV) * ((size_t) X.hh_sizes[SPDP_BLK_HASH_LEVELS - 1] + (size_t) X.hh_sizes[SPDP_BLK_HASH_LEVELS - 2]);
    H.epochs = X.nseg < (1 << 24); H.epoch = 255;
    H.bind(0);
    // (the order of the slab: the run lists' larger tables after the hit lists', as spdp_blk_vote_slab_bytes counts them)
    W.qh_grown_words = X.ha_sizes[SPDP_BLK_HASH_LEVELS - 1] + X.ha_sizes[SPDP_BLK_HASH_LEVELS - 2];
    W.qh_grown = (KV*) g; g += sizeof(KV) * 4 * (size_t) W.qh_grown_words;
    W.qr_grown_words = X.hb_sizes[SPDP_BLK_HASH_LEVELS - 1] + X.hb_sizes[SPDP_BLK_HASH_LEVELS - 2];
    W.qr_grown = (KV*) g; g += sizeof(KV) * 4 * (size_t) W.qr_grown_words;
    W.stage = (uint32_t*) g; g += 4 * (2 * (size_t) X.maxlist + 2);
    W.scratch = (int32_t*) g;
    uint32_t tag = *W.header;
    wave_sync();
    for (;;) {
        int qi = 0;
        if (me == 0) qi = (int) atomicAdd(A.next, 1u);
        qi = uni(qi);
        if (qi >= A.n) break;
        if (++tag == 0) {                               // (the tags have come round: every slot of the slab reads as new again)
            for (size_t i = me; i < 2 * (4 * (size_t) X.nseg + 2); i += 64) ((u64*) W.score)[i] = 0;
            tag = 1;
            wave_sync();
        }
        W.tag = tag;
        const int64_t o = A.offs[qi];
        const int len = (int) (A.offs[qi + 1] - o);
        int calls = 0;
        const int reached = scan(W, A.codes + o, len, A.left[qi], A.right[qi], A.stop_at ? A.stop_at[qi] : 0, calls);
        wave_sync();
        Record R = {A.out + (size_t) qi * A.out_cap, A.out_cap, 3, false};
        if (reached) write_state(W, R);
        wave_sync();
        if (me == 0) {
            if (A.out_cap > 0) R.out[0] = R.n < R.cap ? R.n : R.cap;
            if (A.out_cap > 1) R.out[1] = calls;
            if (A.out_cap > 2) R.out[2] = (reached ? SPDP_BLK_REACHED : 0) | (R.cut ? SPDP_BLK_CUT : 0) | (W.st[ST_TROUBLE] & ~3) |
                                          (reached == 2 ? SPDP_BLK_FORCED : 0);
        }
        wave_sync();
    }
    if (me == 0) *W.header = tag;
}

}   // namespace

extern "C" uint32_t spdp_blk_vote_lds_bytes(const BlkDev* ix, int hh_in_lds)
{
    size_t w = ST_WORDS + 4 * 32 + OWN_SLOTS + SPDP_BLK_RES_CAP / 4 + 64 * SPDP_BLK_PRE_PHASES + 512;
    w += 4 * (2 * ((size_t) ix->nascr + 1) + 2 * (size_t) ix->ha_size1);
    w += 4 * (2 * ((size_t) ix->ncand + 1) + 2 * (size_t) ix->hb_size1);
    if (hh_in_lds) w += 2 * (size_t) ix->hh_sizes[0];
    return (uint32_t) (4 * w);
}
extern "C" size_t spdp_blk_vote_slab_bytes(const BlkDev* ix, int hh_in_lds)
{
    size_t b = 16 + sizeof(Score) * (4 * (size_t) ix->nseg + 2);
    if (!hh_in_lds) b += sizeof(KV) * (size_t) ix->hh_sizes[0];
    b += sizeof(KV) * ((size_t) ix->hh_sizes[SPDP_BLK_HASH_LEVELS - 1] + (size_t) ix->hh_sizes[SPDP_BLK_HASH_LEVELS - 2]);
    b += sizeof(KV) * 4 * ((size_t) ix->hb_sizes[SPDP_BLK_HASH_LEVELS - 1] + ix->hb_sizes[SPDP_BLK_HASH_LEVELS - 2] +
                           (size_t) ix->ha_sizes[SPDP_BLK_HASH_LEVELS - 1] + ix->ha_sizes[SPDP_BLK_HASH_LEVELS - 2]);
    b += 4 * (2 * (size_t) ix->maxlist + 2);
    b += sizeof(Pair) * ((size_t) ix->ncand + 2) + 4 * 3 * (2 * (size_t) ix->ncand + 2);
    return (b + 255) & ~(size_t) 255;
}
extern "C" hipError_t spdp_blk_vote_launch(const BlkVoteArgs* a, hipStream_t s)
{
    BlkVoteArgs A = *a;
    hipError_t e = hipMemsetAsync(A.next, 0, 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spdp_blk_vote_wave, dim3(A.n_waves), dim3(64), A.lds_bytes, s, A);
    return hipGetLastError();
}
