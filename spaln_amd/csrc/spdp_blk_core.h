// spdp_blk_core.h -- the vote of the block search for one query, as one sequential routine (first slice).
//
// What it computes is what the reference's SrchBlk::findblock computes between its TestOutput calls (ogotoh/spaln v3.0.7,
// src/blksrc.cc:2971-3087, with Qwords :2819-2969, Bhit4 :2763-2817, Randbs :2047-2069, extract_to_work :2547-2603 and the
// block-pair list of TestOutput :2620-2672): the query's k-mers are looked up from both ends inwards, on both strands
// (four directions x Nshift phases), every word votes for the genome blocks of its posting list, runs of consecutive
// words that keep hitting the same (or the neighbouring) block build up that block's score, blocks above the random
// expectation enter a bounded priority queue per direction, and the scan stops when enough significant block pairs have
// appeared.  The routine is inherently sequential per query -- each word's effect depends on what the previous one left
// in a small hash -- so the device runs ONE QUERY PER LANE, tens of thousands at a time (spdp_blk.hip); all of a lane's
// state lives in a private slab of HBM (BlkWork), its traffic is random 4-byte reads of posting lists and score slots:
// latency- and HBM-transaction-bound, no MFMA, no LDS.
//
// Two containers are kept with the reference's exact geometry, because its results depend on it: Dhash (double hashing,
// src/clib.h:192-314) -- findblock writes the "empty" value into live slots, which cuts probe chains, so what a later
// lookup finds depends on table size and probe step -- and PrQueue_wh (a binary min-heap with a position hash,
// src/clib.h:570-688).  Table sizes come with the index (SpdpBlkIndexDesc).
//
// Header only, no allocation, no library calls besides log / sqrt / pow: compiled for the device by spdp_blk.hip; the
// tests' CPU checker (oracle/blk_check.cpp) compiles the same text with the host compiler.
#ifndef SPDP_BLK_CORE_H_
#define SPDP_BLK_CORE_H_

#include <stdint.h>

#ifndef SPDP_HD
#define SPDP_HD inline
#endif

#define SPDP_BLK_MAX_SHIFT 32
#define SPDP_BLK_HASH_LEVELS 4          // a hash may grow three times (x ~8) before a query is reported as SPDP_BLK_TABLE
#define SPDP_BLK_INT_MAX 0x7fffffff

struct BlkDev {                         // the index and the search parameters, device side (pointers into HBM)
    int32_t nalpha, tabsize, nshift, nbitpat, convts, n_chr, kk, drna, maxmmc, nseg, minsigpr, ncand, nascr;
    int32_t maxblock, extblock, extblockl, shortquery, hh_size1, hh_size2, hb_size1, hb_size2, ha_size1, ha_size2, gdb;
    int32_t hh_sizes[SPDP_BLK_HASH_LEVELS];  // hh_size1 and what Dhash::resize makes of it: the next prime >= twice the size, again and again
    int32_t hb_sizes[SPDP_BLK_HASH_LEVELS], ha_sizes[SPDP_BLK_HASH_LEVELS];      // the same for the queues' position hashes
    float rbscoef, rbscons;
    double bclw, bcup, bcce, app_c;
    const uint8_t* convtab;
    const uint16_t* nblk;
    const int16_t* wscr;
    const int32_t* blkp;
    const uint32_t* blkb;
    const int32_t* rscrtab;
    const int32_t* chr;
    const int32_t* bitpat;
    int32_t pat_off[3];                 // where pattern k starts in bitpat
};

// supprime(n), src/supprime.cc:375: the smallest prime >= n (host side: fills BlkDev::hh_sizes)
inline uint32_t blk_next_prime(uint32_t n)
{
    if (n <= 3) return n;
    if (n % 2 == 0) ++n;
    for ( ; ; n += 2) {
        bool prime = true;
        for (uint32_t x = 3; x * x <= n; x += 2) if (n % x == 0) { prime = false; break; }
        if (prime) return n;
    }
}
inline void blk_fill_hash_levels(BlkDev& ix)
{
    ix.hh_sizes[0] = ix.hh_size1; ix.hb_sizes[0] = ix.hb_size1; ix.ha_sizes[0] = ix.ha_size1;
    for (int l = 1; l < SPDP_BLK_HASH_LEVELS; ++l) {
        ix.hh_sizes[l] = (int32_t) blk_next_prime(2u * (uint32_t) ix.hh_sizes[l - 1]);
        ix.hb_sizes[l] = (int32_t) blk_next_prime(2u * (uint32_t) ix.hb_sizes[l - 1]);
        ix.ha_sizes[l] = (int32_t) blk_next_prime(2u * (uint32_t) ix.ha_sizes[l - 1]);
    }
}

struct BlkKV { uint32_t key; int32_t val; };
struct BlkBS { uint32_t key; int32_t bscr; };

// t = the live table; a table that can grow (the run hash of findblock) also has a second buffer and the sizes of its next levels
// epoch != 0: "cleared" is a matter of counting -- a slot belongs to the table as it is now only if the top byte of its key
// holds the current epoch (keys are block numbers, < 2^24); clearing bumps the epoch and touches memory once in 255 times.
// Every reader goes through blk_slot_val / blk_slot_key, so the table behaves exactly as one that is wiped each time.
struct BlkHash { BlkKV* t; uint32_t size1, size2; int32_t undef; BlkKV* spare; const int32_t* sizes; int level; uint32_t epoch; };
struct BlkQueue { BlkBS* data; int capacity, front; BlkHash hpos; };

struct BlkWork {                        // one lane's slab
    int32_t* bscr;                      // 4 x nseg (+2), contiguous: an index one past a row lands in the next row, as in the reference
    int32_t* ascr;
    BlkHash hh;
    BlkQueue qa[4], qb[4];
    int32_t* touched; int touched_cap, n_touched;    // d * nseg + blk of every score slot written (for the clean-up), -1 = overflowed
    int overflow;
};

// bytes of one lane's slab and its carving (the same function sizes the allocation and binds the pointers)
SPDP_HD size_t blk_work_ints(const BlkDev& ix, int touched_cap)
{
    const int top = SPDP_BLK_HASH_LEVELS - 1;
    size_t n = 2 * (4 * (size_t) ix.nseg + 2);
    n += 4 * (size_t) ix.hh_sizes[top];                                  // two buffers of the largest level, two ints per slot
    n += 4 * (2 * ((size_t) ix.nascr + 1) + 4 * (size_t) ix.ha_sizes[top]);
    n += 4 * (2 * ((size_t) ix.ncand + 1) + 4 * (size_t) ix.hb_sizes[top]);
    n += (size_t) touched_cap;
    return (n + 3) & ~(size_t) 3;
}
SPDP_HD int32_t* blk_hash_bind(BlkHash& h, int32_t* p, const int32_t* sizes, int32_t step, int32_t undef, bool epochs = false)
{
    h.epoch = epochs ? 255 : 0;                         // (255: the first clear wipes the buffer for real)
    const size_t top = (size_t) sizes[SPDP_BLK_HASH_LEVELS - 1];
    h.t = (BlkKV*) p; h.spare = (BlkKV*) (p + 2 * top);
    h.size1 = (uint32_t) sizes[0]; h.size2 = (uint32_t) step; h.undef = undef; h.sizes = sizes; h.level = 0;
    return p + 4 * top;
}
SPDP_HD void blk_work_bind(BlkWork& w, const BlkDev& ix, int32_t* slab, int touched_cap)
{
    int32_t* p = slab;
    w.bscr = p; p += 4 * (size_t) ix.nseg + 2;
    w.ascr = p; p += 4 * (size_t) ix.nseg + 2;
    p = blk_hash_bind(w.hh, p, ix.hh_sizes, ix.hh_size2, 0, ix.nseg < (1 << 24));
    for (int d = 0; d < 4; ++d) {
        w.qa[d].data = (BlkBS*) p; p += 2 * ((size_t) ix.nascr + 1);
        w.qa[d].capacity = ix.nascr; w.qa[d].front = 0;
        p = blk_hash_bind(w.qa[d].hpos, p, ix.ha_sizes, ix.ha_size2, -1);
        w.qb[d].data = (BlkBS*) p; p += 2 * ((size_t) ix.ncand + 1);
        w.qb[d].capacity = ix.ncand; w.qb[d].front = 0;
        p = blk_hash_bind(w.qb[d].hpos, p, ix.hb_sizes, ix.hb_size2, -1);
    }
    w.touched = p; w.touched_cap = touched_cap; w.n_touched = 0;
    w.overflow = 0;
}

// ---- Dhash<key, int> ------------------------------------------------------------------------------------------------
SPDP_HD void blk_hash_wipe(BlkHash& h) { for (uint32_t i = 0; i < h.size1; ++i) { h.t[i].key = 0; h.t[i].val = h.undef; } }
SPDP_HD void blk_hash_clear(BlkHash& h)
{
    if (h.epoch && ++h.epoch < 256) return;
    blk_hash_wipe(h);
    if (h.epoch) h.epoch = 1;
}
SPDP_HD int32_t blk_slot_val(const BlkHash& h, const BlkKV* sh) { return (h.epoch && (sh->key >> 24) != h.epoch) ? h.undef : sh->val; }
SPDP_HD uint32_t blk_slot_key(const BlkHash& h, const BlkKV* sh) { return h.epoch ? sh->key & 0xffffffu : sh->key; }
SPDP_HD void blk_slot_claim(const BlkHash& h, BlkKV* sh, uint32_t key) { sh->key = h.epoch ? key | (h.epoch << 24) : key; sh->val = h.undef; }
SPDP_HD void blk_hash_rewind(BlkHash& h)
{
    if (h.level & 1) { BlkKV* t = h.t; h.t = h.spare; h.spare = t; }
    if (h.level && h.epoch) h.epoch = 255;              // another buffer, another size: the next clear wipes it
    h.level = 0; h.size1 = (uint32_t) h.sizes[0];
}
// the probe of one key; returns the slot (a free one it may claim, or the key's own), or null when the probe came round
SPDP_HD BlkKV* blk_hash_probe(BlkHash& h, uint32_t key, uint32_t& v, uint32_t u, uint32_t v0)
{
    BlkKV* sh = h.t + v;
    while (blk_slot_val(h, sh) != h.undef && blk_slot_key(h, sh) != key) {
        v = (v + u) % h.size1;
        if (v == v0) return nullptr;
        sh = h.t + v;
    }
    return sh;
}
// Dhash::resize() (src/clib.h:341-355): the next level's table, live entries re-entered in slot order
SPDP_HD bool blk_hash_grow(BlkHash& h)
{
    if (!h.spare || h.level + 1 >= SPDP_BLK_HASH_LEVELS) return false;
    BlkKV* old = h.t;
    const uint32_t n_old = h.size1;
    h.t = h.spare; h.spare = old;
    h.size1 = (uint32_t) h.sizes[++h.level];
    const uint32_t was = h.epoch;
    blk_hash_wipe(h);                                   // (the new buffer may hold anything: wiped for real, same epoch)
    for (uint32_t i = 0; i < n_old; ++i) {
        const bool live = (!was || (old[i].key >> 24) == was) && old[i].val != h.undef;
        if (!live) continue;
        const uint32_t key = was ? old[i].key & 0xffffffu : old[i].key;
        uint32_t v = key % h.size1;
        BlkKV* sh = blk_hash_probe(h, key, v, h.size2 - key % h.size2, v);
        if (!sh) return false;
        blk_slot_claim(h, sh, key);
        sh->val = old[i].val;
    }
    return true;
}
SPDP_HD BlkKV* blk_hash_map(BlkHash& h, uint32_t key, bool record, int& overflow)
{
    uint32_t v = key % h.size1;
    const uint32_t u = h.size2 - key % h.size2, v0 = v;
    BlkKV* sh;
    while (!(sh = blk_hash_probe(h, key, v, u, v0))) {
        // the probe came round: the reference grows the table and goes on probing the NEW table from the position and with
        // the step it had in the old one
        if (!blk_hash_grow(h)) { overflow = 1; return record ? h.t + v : nullptr; }
        BlkKV* at = h.t + v;
        if (!(blk_slot_val(h, at) != h.undef && blk_slot_key(h, at) != key)) { sh = at; break; }
    }
    if (blk_slot_val(h, sh) == h.undef) { if (record) blk_slot_claim(h, sh, key); else sh = nullptr; }
    return sh;
}
SPDP_HD BlkKV* blk_hash_incr(BlkHash& h, uint32_t key, int& overflow)
{
    BlkKV* sh = blk_hash_map(h, key, true, overflow);
    if (sh->val == h.undef) sh->val = 0;
    sh->val += 1;
    return sh;
}

// ---- PrQueue_wh<BlkScr>: ascending heap (data[0] = the smallest score), replace = true -----------------------------
SPDP_HD void blk_queue_settle(BlkQueue& q, int k, BlkBS v, int& ovf) { q.data[k] = v; blk_hash_map(q.hpos, v.key, true, ovf)->val = k; }
SPDP_HD void blk_queue_down(BlkQueue& q, int k, int& ovf)
{
    const BlkBS v = q.data[k];
    const int kmax = q.front;
    while (k < kmax / 2) {
        int l = 2 * k + 1;
        const int r = l + 1;
        if (r < kmax && q.data[r].bscr < q.data[l].bscr) ++l;
        if (!(q.data[l].bscr < v.bscr)) break;
        blk_queue_settle(q, k, q.data[l], ovf);
        k = l;
    }
    blk_queue_settle(q, k, v, ovf);
}
SPDP_HD void blk_queue_up(BlkQueue& q, int k, int& ovf)
{
    const BlkBS v = q.data[k];
    int h = (k - 1) / 2;
    while (k && v.bscr < q.data[h].bscr) {
        blk_queue_settle(q, k, q.data[h], ovf);
        k = h;
        h = (h - 1) / 2;
    }
    blk_queue_settle(q, k, v, ovf);
}
SPDP_HD void blk_queue_update(BlkQueue& q, BlkBS x, int& ovf)
{
    const BlkKV* kv = blk_hash_map(q.hpos, x.key, false, ovf);
    int p = kv ? kv->val : -1;
    if (p < 0) {
        if (q.front < q.capacity) { q.data[q.front] = x; blk_queue_up(q, q.front++, ovf); return; }
        p = 0;
    }
    if (q.data[p].bscr < x.bscr) {
        blk_hash_map(q.hpos, q.data[p].key, true, ovf)->val = q.hpos.undef;
        q.data[p] = x;
        blk_queue_down(q, p, ovf);
    }
}

// ---- words of the query ------------------------------------------------------------------------------------------------
struct BlkWords {
    uint32_t ww[3]; int xx[3]; uint32_t front[3]; int endss[3];
};

SPDP_HD int blk_randbs(const BlkDev& ix, uint32_t mmc)
{
    if (mmc < 128) return ix.rscrtab[mmc];
    if (ix.rbscoef == 0) return (int) ix.rbscons;
    const double x = (double) (mmc + 1);
    return (int) (ix.rbscoef * (ix.gdb ? log(x) : sqrt(x)) + ix.rbscons);
}

SPDP_HD uint32_t blk_code(const BlkDev& ix, const uint8_t* q, int q_len, int i)
{
    if (i < 0 || i >= q_len) return 255;
    const int c = q[i];
    return c < ix.convts ? ix.convtab[c] : 255;
}

SPDP_HD uint32_t blk_spell(const BlkDev& ix, const uint8_t* q, int q_len, int ss, int d, bool rvs, int k, int& n_good)
{
    const int32_t* bp = ix.bitpat + ix.pat_off[k];
    const int weight = bp[0], wshift = bp[2];
    const int32_t* exam = bp + 3 + (rvs ? weight : 0);
    const uint32_t nalpha = (uint32_t) ix.nalpha, tab = (uint32_t) ix.tabsize;
    uint32_t w = 0;
    int i = 0;
    for ( ; i < weight; ++i) {
        const uint32_t c = blk_code(ix, q, q_len, ss + exam[i]);
        if (c >= nalpha) break;
        if (ix.drna) w = d >= 2 ? (w >> 2) + ((3 - c) << wshift) : (w << 2) + c;
        else         w = d >= 2 ? (tab * c + w) / nalpha : w * nalpha + c;
    }
    n_good = i;
    return w;
}

// Qwords::querywords(ss, d, rvs): > 0 the words' score, 0 a ubiquitous word, < 0 no usable word
SPDP_HD int blk_query_words(const BlkDev& ix, BlkWords& qw, const uint8_t* q, int q_len, int ss, int d, bool rvs)
{
    const uint32_t tab = (uint32_t) ix.tabsize;
    if (ix.kk == 1) {
        int good;
        qw.ww[0] = blk_spell(ix, q, q_len, ss, d, rvs, 0, good);
        qw.xx[0] = 0;
        if (!ix.blkp[qw.ww[0]]) return 0;
        if (ix.wscr[qw.ww[0]] < 0) qw.xx[0] = -1;
        if (good == ix.bitpat[ix.pat_off[0]]) return ix.wscr[qw.ww[0]];
        return -1;
    }
    for (int k = 0; k < ix.kk; ++k) qw.ww[k] = tab;
    for (int k = 0; k < ix.kk; ++k) {
        if (ss >= qw.endss[k]) break;
        int good;
        qw.ww[k] = blk_spell(ix, q, q_len, ss, d, rvs, k, good);
        qw.xx[k] = good < ix.bitpat[ix.pat_off[k]] ? -1 : 0;
    }
    int c = 0, wdscr = 0;
    for (int k = 0; k < ix.kk; ++k) {
        if (qw.ww[k] >= tab || ix.wscr[qw.ww[k]] < 0) qw.xx[k] = -1;
        else if (!qw.xx[k] && ix.blkp[qw.ww[k]]) { ++c; wdscr += ix.wscr[qw.ww[k]]; }
    }
    return c ? (int) ((double) wdscr / ix.app_c) : -1;
}

SPDP_HD uint32_t blk_posting(const BlkDev& ix, uint32_t w, int x) { return ix.blkb[ix.blkp[w] - 1 + x]; }
SPDP_HD void blk_merge_begin(const BlkDev& ix, BlkWords& qw)
{
    for (int k = 0; k < ix.kk; ++k)
        qw.front[k] = (qw.xx[k] >= 0 && qw.ww[k] < (uint32_t) ix.tabsize && ix.blkp[qw.ww[k]]) ? blk_posting(ix, qw.ww[k], qw.xx[k]) : 0;
}
SPDP_HD uint32_t blk_merge_next(const BlkDev& ix, BlkWords& qw)
{
    uint32_t blk = qw.front[0];
    if (ix.kk == 1) {
        qw.front[0] = (++qw.xx[0] < (int) ix.nblk[qw.ww[0]]) ? blk_posting(ix, qw.ww[0], qw.xx[0]) : 0;
        return blk;
    }
    const int last = ix.kk - 1;                        // the reference's merge loop runs over all patterns but the last
    for (int j = 0; j < last; ++j) {
        if (qw.xx[j] < 0) continue;
        if (blk == 0) blk = qw.front[j];
        if (qw.front[j] && qw.front[j] < blk) blk = qw.front[j];
    }
    if (blk == 0) return 0;
    for (int j = 0; j < last; ++j) {
        if (qw.xx[j] < 0) continue;
        if (blk == qw.front[j])
            qw.front[j] = (++qw.xx[j] < (int) ix.nblk[qw.ww[j]]) ? blk_posting(ix, qw.ww[j], qw.xx[j]) : 0;
    }
    return blk;
}

// ---- chromosomes and block pairs -------------------------------------------------------------------------------------
SPDP_HD uint32_t blk_chr_first(const BlkDev& ix, int m) { return (uint32_t) ix.chr[2 * m + 1]; }
SPDP_HD int blk_chr_of(const BlkDev& ix, uint32_t blk)
{
    int lw = (int) (ix.bclw + ix.bcce * (blk - 1)) - 1;
    int up = (int) (ix.bcup + ix.bcce * (blk - 1)) + 1;
    if (lw < 0) lw = 0;
    if (up > ix.n_chr) up = ix.n_chr;
    if (blk_chr_first(ix, lw) > blk) lw = 0;
    if (blk_chr_first(ix, up) < blk) up = ix.n_chr;
    while (up - lw > 1) {
        const int md = (lw + up) / 2;
        if (blk_chr_first(ix, md) > blk) up = md;
        else if (blk_chr_first(ix, md + 1) > blk) return md;
        else lw = md;
    }
    return blk_chr_first(ix, up) > blk ? lw : up;
}

// the significant blocks of one strand (both ends) sorted by position and grouped: sw[] gets block << 1 | "pairs with the next";
// sw must hold 2 * ncand + 2 words
SPDP_HD int blk_group_blocks(const BlkDev& ix, const BlkWork& w, const int* sign, int d, uint32_t* sw)
{
    const int e = d + 1, f = d >> 1;
    if (!sign[d] && !sign[e]) return 0;
    int j = 0;
    for (int i = 0; i < sign[d]; ++i) sw[j++] = w.qb[d].data[i].key << 1;
    for (int i = 0; i < sign[e]; ++i) sw[j++] = (w.qb[e].data[i].key << 1) + 1;
    if (j == 1) { sw[0] &= ~1u; sw[1] = SPDP_BLK_INT_MAX; return 1; }
    for (int a = 1; a < j; ++a) {                       // ascending (the keys are distinct: any sort gives this order)
        const uint32_t v = sw[a];
        int b = a;
        while (b > 0 && sw[b - 1] > v) { sw[b] = sw[b - 1]; --b; }
        sw[b] = v;
    }
    uint32_t p = sw[0] >> 1;
    if (d && j > 1) {
        for (int i = 1; i < j; ++i) {
            const uint32_t q = sw[i] >> 1;
            if (p == q) { const uint32_t t = sw[i - 1]; sw[i - 1] = sw[i]; sw[i] = t; }
            else p = q;
        }
    }
    p = sw[0] >> 1;
    int pr = (int) (sw[0] & 1) ^ f;
    int cp = blk_chr_of(ix, p);
    sw[0] &= ~1u;
    int k = 0, run = 0;
    for (int i = 1; i < j; ++i) {
        const uint32_t q = sw[i] >> 1;
        const int qr = (int) (sw[i] & 1) ^ f;
        const int cq = blk_chr_of(ix, q);
        sw[i] &= ~1u;
        const int st = (int) (q - p);
        if (cp == cq && (st < 2 || (!pr && qr && st <= ix.maxblock) || (pr == qr && st <= ix.extblock))) {
            if (!run++) sw[k++] = sw[i - 1];
        } else {
            sw[k++] = sw[i - 1] | (run ? 1u : 0u);
            run = 0;
        }
        p = q; pr = qr; cp = cq;
    }
    sw[k++] = sw[j - 1] + (run ? 1u : 0u);
    sw[k] = SPDP_BLK_INT_MAX;
    return k;
}

struct BlkPair { int32_t bscr, chr; uint32_t lb, rb, ub, db, zl, zr; int32_t rvs; };     // BPAIR, src/blksrc.h:287-294 (nine ints out)

// TestOutput's list of candidate block pairs, best first; bpair holds ncand + 1 entries, sw 2 x (2 ncand + 2) words
SPDP_HD int blk_build_pairs(const BlkDev& ix, const BlkWork& w, const int* sign, BlkPair* bpair, uint32_t* sw_both)
{
    const int nseg = ix.nseg;
    uint32_t* sigw[2] = {sw_both, sw_both + 2 * ix.ncand + 2};
    int sigm[2];
    BlkPair* cur = bpair;
    BlkPair* const last = bpair + ix.ncand;
    cur->bscr = 0;
    for (int f = 0; f < 2; ++f) sigm[f] = blk_group_blocks(ix, w, sign, f << 1, sigw[f]);
    for (int f = 0; f < 2; ++f) {
        const int32_t* bd = w.bscr + (size_t) (2 * f) * nseg;
        const int32_t* be = w.bscr + (size_t) (2 * f + 1) * nseg;
        uint32_t pu = 0;
        for (int i = 0; i < sigm[f]; ++i) {
            const uint32_t p = sigw[f][i] >> 1;
            uint32_t q = sigw[f][i + 1 < sigm[f] ? i + 1 : i];
            const bool ispair = q & 1;
            q >>= 1;
            if (ispair) ++i; else q = p;
            const uint32_t qd = sigw[f][i + 1] >> 1;
            cur->rvs = f;
            const int c1 = cur->chr = blk_chr_of(ix, q);
            cur->zl = blk_chr_first(ix, c1);
            cur->zr = blk_chr_first(ix, c1 + 1) - 1;
            cur->lb = p; cur->rb = q; cur->bscr = 0;
            for (uint32_t r = cur->lb; r <= cur->rb; ++r) cur->bscr += bd[r] + be[r];
            const uint32_t exb = (uint32_t) ix.extblock;
            uint32_t r = cur->lb;
            uint32_t z = r > exb ? r - exb : 0;
            if (cur->zl > z) z = cur->zl;
            if (pu > z) z = pu;
            while (r && --r >= z && (bd[r] + be[r])) { cur->lb = r; cur->bscr += bd[r] + be[r]; }
            cur->ub = r > exb ? r - exb : 0;
            if (cur->zl > cur->ub) cur->ub = cur->zl;
            r = cur->rb;
            z = r + exb;
            if (cur->zr < z) z = cur->zr;
            if (qd < z) z = qd;
            while (++r < z && (bd[r] + be[r])) { cur->rb = r; cur->bscr += bd[r] + be[r]; }
            cur->db = r + exb < cur->zr ? r + exb : cur->zr;
            pu = cur->rb + 1;
            for (BlkPair* x = cur; --x >= bpair; ) {
                if (x[1].bscr > x->bscr) { const BlkPair t = x[0]; x[0] = x[1]; x[1] = t; }
                else break;
            }
            if (cur < last) ++cur;
        }
    }
    return (int) (cur - bpair);
}

// ---- the vote ------------------------------------------------------------------------------------------------------------
struct BlkVote {                        // what findblock holds in Bhit4 besides the score arrays and queues
    int sign[4], maxs[4], nhit[4], mmct[4], testword[4];
};

SPDP_HD void blk_touch(BlkWork& w, int slot)
{
    if (w.n_touched < 0) return;
    if (w.n_touched >= w.touched_cap) { w.n_touched = -1; return; }
    w.touched[w.n_touched++] = slot;
}

// Runs findblock's scan for the query q[left, right) up to its stop_at-th TestOutput call (earlier calls are taken to have
// answered "nothing found, go on").  Returns 1 when that call is reached (v and the work slab hold the state TestOutput
// sees; 2 when it is the forced call findblock makes behind its scan, TestOutput(1)), 0 when findblock ends before it;
// *calls = TestOutput calls met on the way.  The caller clears the slab
// afterwards (blk_work_reset).
SPDP_HD int blk_vote_run(const BlkDev& ix, BlkWork& w, BlkVote& v, const uint8_t* q, int q_len, int left, int right,
                         int stop_at, int* calls_out)
{
    const int nshift = ix.nshift, nseg = ix.nseg;
    for (int d = 0; d < 4; ++d) { v.sign[d] = v.maxs[d] = v.nhit[d] = v.mmct[d] = v.testword[d] = 0; }
    *calls_out = 0;
    const int width0 = ix.bitpat[ix.pat_off[0] + 1];
    const int qlen = right - left;
    if (qlen - (nshift + width0) < 1 || nshift > SPDP_BLK_MAX_SHIFT) return 0;
    BlkWords qw;
    for (int k = 0; k < 3; ++k) { qw.ww[k] = 0; qw.xx[k] = 0; qw.front[k] = 0; qw.endss[k] = k < ix.kk ? right - ix.bitpat[ix.pat_off[k] + 1] : 0; }
    // every query starts as the first query of a process does: queues empty, every hash at its initial size.  (The
    // reference's run hash is a local of findblock; its queues' position hashes belong to the worker thread and keep the
    // size an earlier query may have grown them to -- a dependence on the thread's history that is not imitated.)
    blk_hash_rewind(w.hh);
    for (int d = 0; d < 4; ++d) {
        w.qa[d].front = 0; blk_hash_rewind(w.qa[d].hpos); blk_hash_clear(w.qa[d].hpos);
        w.qb[d].front = 0; blk_hash_rewind(w.qb[d].hpos); blk_hash_clear(w.qb[d].hpos);
    }
    // scan positions (init4): phases s = 0 .. Nshift - 1 from the left end and from the last full word at the right end
    int as[4][SPDP_BLK_MAX_SHIFT];
    {
        int ss = left, ts = right - (width0 + nshift);
        int ph = (ts-- - ss) % nshift;
        for (int p = 0; p < nshift; ++p) {
            as[0][p] = as[2][p] = ss++;
            as[1][ph] = as[3][ph] = ts++;
            if (++ph == nshift) ph = 0;
        }
    }
    int calls = 0, nohit = 0, sigpr = 0, notry = 0;
    int c = qlen / (nshift + nshift) - 1;
    const bool is_short = qlen < ix.shortquery;
    bool meet[2] = {false, false};
    uint32_t nmmc = 0;
    int maxbscr[4] = {0, 0, 0, 0};
    const int base = ix.rscrtab[0];
    while (!(meet[0] || meet[1])) {
        int totalsign = 0;
        for (int d = 0; d < 4; ++d) {
            if (meet[d / 2]) continue;
            const int prty = d & 1, e = prty ? d - 1 : d + 1;
            const bool rvs = d >= 2;
            int32_t* rscr = w.bscr + (size_t) d * nseg;
            int32_t* acr = w.ascr + (size_t) d * nseg;
            int ms = prty ? left : right;
            int maxp = 0;
            for (int sft = 0; sft < nshift; ++sft) {
                if (!is_short) ms = as[e][sft];
                int cscr = 0, more = 0, p = 0;
                blk_hash_clear(w.hh);
                do {
                    const int ss = as[d][sft];
                    as[d][sft] += prty ? -nshift : nshift;
                    if (prty ^ (ss >= ms)) { meet[d / 2] = true; break; }
                    const int wdscr = blk_query_words(ix, qw, q, q_len, ss, d, rvs);
                    if (wdscr < 0) break;
                    v.testword[d] += ix.kk;
                    if (wdscr == 0) { more = 1; continue; }
                    blk_merge_begin(ix, qw);
                    ++p; more = 0;
                    cscr += wdscr;
                    uint32_t blk;
                    while ((blk = blk_merge_next(ix, qw)) != 0) {
                        BlkKV* h = blk_hash_incr(w.hh, blk, w.overflow);
                        acr[blk] += wdscr;
                        blk_touch(w, 4 * nseg + 2 + d * nseg + (int) blk);
                        { const BlkBS sb = {blk, acr[blk]}; blk_queue_update(w.qa[d], sb, w.overflow); }
                        if (p != h->val) {                          // the run of consecutive hits broke in this block: try its neighbour
                            h->val = 0;
                            if (prty) h = blk_hash_incr(w.hh, ++blk, w.overflow);
                            else if (blk) h = blk_hash_incr(w.hh, --blk, w.overflow);
                        }
                        if (p == h->val) {
                            ++more;
                            rscr[blk] += wdscr;
                            blk_touch(w, d * nseg + (int) blk);
                            if (rscr[blk] > maxbscr[d]) { maxbscr[d] = rscr[blk]; v.maxs[d] = sft; }
                            if (rscr[blk] >= blk_randbs(ix, nmmc)) {
                                const BlkBS sb = {blk, rscr[blk]};
                                blk_queue_update(w.qb[d], sb, w.overflow);
                                v.sign[d] = w.qb[d].front;
                            }
                        } else h->val = 0;
                    }
                } while (more && cscr < base);
                if (p > maxp) maxp = p;
                if (v.maxs[d] == sft) nohit = !more;
            }
            v.mmct[d] += nohit;
            v.nhit[d] += maxp;
            totalsign += v.sign[d];
        }
        if ((v.sign[0] && v.sign[1]) || (v.sign[2] && v.sign[3])) ++sigpr;
        if (((++nmmc % (uint32_t) ix.maxmmc) == 0 && totalsign) || sigpr > ix.minsigpr) {
            if (calls++ == stop_at) { *calls_out = calls; return 1; }
            c = 0;
            if (++notry > ix.minsigpr) { *calls_out = calls; return 0; }
        }
    }
    if (!((v.sign[0] && v.sign[1]) || (v.sign[2] && v.sign[3]))) {
        // no significant pair: the blocks with most word hits of all stand in
        c = -1;
        for (int d = 0; d < 4; ++d)
            for (int i = 0; i < ix.nascr; ++i) {
                const BlkBS bs = i < w.qa[d].front ? w.qa[d].data[i] : BlkBS{0, 0};
                if (bs.key) {
                    const BlkBS sb = {bs.key, w.bscr[(size_t) d * nseg + bs.key]};
                    blk_queue_update(w.qb[d], sb, w.overflow);
                    c = v.sign[d] = w.qb[d].front;
                }
            }
    }
    if (c != -1 && calls++ == stop_at) { *calls_out = calls; return 2; }    // (2: the forced call behind the scan, TestOutput(1))
    *calls_out = calls;
    return 0;
}

// the run scores a caller's FindHsp can ask for: it moves a pair's ends by up to ExtBlockL blocks inside the chromosome and
// looks at the run scores of the pair's own strand on the way (src/blksrc.cc:2408-2460)
SPDP_HD bool blk_near_a_pair(const BlkDev& ix, const BlkPair* bpair, int np, int d, uint32_t blk)
{
    for (int i = 0; i < np; ++i) {
        const BlkPair& b = bpair[i];
        if (b.rvs != (d >> 1)) continue;
        const uint32_t e = (uint32_t) ix.extblockl;
        uint32_t lo = b.lb > e ? b.lb - e : 0, hi = b.rb + e;
        if (lo < b.zl) lo = b.zl;
        if (hi > b.zr) hi = b.zr;
        if (blk >= lo && blk <= hi) return true;
    }
    return false;
}

// One query's result record (int32): [0] ints written incl. this header, [1] TestOutput calls met, [2] flags
// (1 reached the asked call, 2 record cut at the capacity, 4 a hash table of the reference's size ran full), then -- if
// reached -- sign[4] mmct[4] nhit[4] maxs[4] testword[4]; per direction: n, (block, score) x n of the significant blocks in
// the queue's own order; n_pairs, nine ints per candidate block pair (bscr chr lb rb ub db zl zr rvs), best first; n_runs,
// (block | direction << 28, score) x n_runs: the run scores within ExtBlockL blocks of a reported pair, on its strand (unordered).  Also zeroes the score slots it walks.
SPDP_HD int blk_emit_and_reset(const BlkDev& ix, BlkWork& w, const BlkVote& v, int reached, int calls, BlkPair* bpair, uint32_t* sw,
                               int32_t* out, int cap)
{
    const int nseg = ix.nseg;
    int n = 3, cut = 0, np = 0;
#define PUT(x) do { if (n < cap) out[n] = (x); else cut = 1; ++n; } while (0)
    if (reached) {
        for (int d = 0; d < 4; ++d) PUT(v.sign[d]);
        for (int d = 0; d < 4; ++d) PUT(v.mmct[d]);
        for (int d = 0; d < 4; ++d) PUT(v.nhit[d]);
        for (int d = 0; d < 4; ++d) PUT(v.maxs[d]);
        for (int d = 0; d < 4; ++d) PUT(v.testword[d]);
        for (int d = 0; d < 4; ++d) {
            PUT(w.qb[d].front);
            for (int i = 0; i < w.qb[d].front; ++i) { PUT((int32_t) w.qb[d].data[i].key); PUT(w.qb[d].data[i].bscr); }
        }
        np = blk_build_pairs(ix, w, v.sign, bpair, sw);
        PUT(np);
        for (int i = 0; i < np; ++i) {
            const BlkPair& b = bpair[i];
            PUT(b.bscr); PUT(b.chr); PUT((int32_t) b.lb); PUT((int32_t) b.rb); PUT((int32_t) b.ub); PUT((int32_t) b.db);
            PUT((int32_t) b.zl); PUT((int32_t) b.zr); PUT(b.rvs);
        }
    }
    // the run scores, and the clean-up of both score arrays: by the list of touched slots, or -- if that ran over -- in full
    const int at = n;
    if (reached) PUT(0);
    const int row = 4 * nseg + 2;                       // slots < row: run scores (bscr), the others word-hit counts (ascr)
    if (w.n_touched >= 0) {
        for (int i = 0; i < w.n_touched; ++i) {
            const int slot = w.touched[i];
            if (slot >= row) { w.ascr[slot - row] = 0; continue; }
            if (reached && w.bscr[slot]) {              // (a slot can be listed many times: it is reported at its first visit)
                const int d = slot / nseg < 4 ? slot / nseg : 3;
                if (blk_near_a_pair(ix, bpair, np, d, (uint32_t) (slot - d * nseg))) {
                    if (at < cap) out[at] += 1;
                    PUT((slot - d * nseg) | (d << 28)); PUT(w.bscr[slot]);
                }
            }
            w.bscr[slot] = 0;
        }
    } else {
        for (int d = 0; d < 4; ++d)
            for (int x = 0; x < nseg; ++x) {
                const size_t s = (size_t) d * nseg + x;
                if (reached && w.bscr[s] && blk_near_a_pair(ix, bpair, np, d, (uint32_t) x)) { if (at < cap) out[at] += 1; PUT(x | (d << 28)); PUT(w.bscr[s]); }
                w.bscr[s] = 0; w.ascr[s] = 0;
            }
    }
    w.bscr[4 * (size_t) nseg] = w.bscr[4 * (size_t) nseg + 1] = 0;
    w.ascr[4 * (size_t) nseg] = w.ascr[4 * (size_t) nseg + 1] = 0;
    w.n_touched = 0;
#undef PUT
    if (cap > 0) out[0] = n < cap ? n : cap;
    if (cap > 1) out[1] = calls;
    if (cap > 2) out[2] = (reached ? 1 : 0) | (cut ? 2 : 0) | (w.overflow ? 4 : 0) | (reached == 2 ? 8 : 0);
    w.overflow = 0;
    return n;
}

#endif
