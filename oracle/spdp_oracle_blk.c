/* spdp_oracle_blk.c -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * CPU restatement of the voting phase of the reference's block search for a cDNA / EST query against a genome index
 * (SURVEY 8 row f4): SrchBlk::findblock up to each of its TestOutput calls, and the candidate block pairs TestOutput
 * builds before it hands them to FindHsp.  ogotoh/spaln v3.0.7:
 *
 *   SrchBlk::findblock          src/blksrc.cc:2971-3087     blk_vote (the two-sided scan, its stop rules)
 *   SrchBlk::init4              src/blksrc.cc:1940-1964     scan positions as[d][s]
 *   Qwords::querywords(ss,d,rvs) src/blksrc.cc:2880-2936    query_words (k-mer codes, forward and reverse-complement)
 *   Qwords::init_mrglist / next_mrglist  :2938-2969         merge of the posting lists (incl. its k = 1 .. kk-1 loop over
 *                                                           elements 0 .. kk-2)
 *   Bhit4::update_a / update_b  :2804-2817                  PrQueue_wh<BlkScr> with its position hash (src/clib.h:570-688)
 *   Dhash<INT,int>::map / incr / resize  src/clib.h:192-355 double hashing, literally (growth included): findblock writes the "undefined"
 *                                                           value 0 into live slots (h->val = 0), which cuts probe chains;
 *                                                           what a later lookup finds depends on the table geometry
 *   Randbs::randbs              :2064-2069
 *   SrchBlk::extract_to_work    :2547-2603
 *   SrchBlk::TestOutput, the block-pair list :2605-2672
 *   SrchBlk::findChrNo          :1985-2002
 *
 * Pinned to the reference itself: oracle/_ref/spaln_blktap (the reference's CLI with a recorder on these functions,
 * oracle/ref_build/blk_tap.cc) writes the index, the queries and the state at every TestOutput / first FindHsp call;
 * tests/test_oracle_blk.py replays the fixtures (tests/golden/blk_*.spdg) through this file.
 * Output layout of one snapshot = the recorder's: see blk_tap.cc, records -2 and -3.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    /* BlkWcPrm + ContBlk scalars + file statics of blksrc.cc, as the recorder wrote them (blk_prm) */
    int32_t nalpha, ktuple, tabsize, nshift, blklen, nbitpat, convts, n_chr, avrscr, maxblk;
    int32_t kk, drna, maxmmc, nseg, minsigpr, ncand, nascr, maxblock, extblock, shortquery;
    int32_t hh_size1, hh_size2, hb_size1, hb_size2, ha_size1, ha_size2;
    int32_t phase1t, gdb, has_chrid, pad0;
    float rbscoef, rbscons;
    double bclw, bcup, bcce, cfact;
    const uint8_t* convtab;
    const uint16_t* nblk;
    const int16_t* wscr;
    const int32_t* blkp;        /* offset into blkb + 1, 0 = none */
    const uint32_t* blkb;
    const int32_t* rscrtab;     /* 128 */
    const int32_t* chr;         /* (spos, first block) per chromosome, n_chr + 1 entries */
    const int32_t* bitpat;      /* per pattern: weight, width, wshift, exam[2 * weight] */
} BlkIndex;

/* ---- Dhash<key, int>, literally ---------------------------------------------------------------------------------- */
typedef struct { uint32_t key; int32_t val; } KV;
typedef struct { KV* t; uint32_t size1, size2; int32_t undef; int overflow; } DH;

static void dh_clear(DH* h) { for (uint32_t i = 0; i < h->size1; ++i) { h->t[i].key = 0; h->t[i].val = h->undef; } }
static uint32_t next_prime(uint32_t n)          /* supprime(n), src/supprime.cc:375: the smallest prime >= n */
{
    if (n <= 3) return n;
    if (n % 2 == 0) ++n;
    for ( ; ; n += 2) {
        int prime = 1;
        for (uint32_t x = 3; x * x <= n; x += 2) if (n % x == 0) { prime = 0; break; }
        if (prime) return n;
    }
}
static KV* dh_map(DH* h, uint32_t key, int record);
int spdp_oracle_blk_last_forced = 0;            /* the snapshot of the last call is findblock's closing TestOutput(1) (tests: the loci checker) */
int spdp_oracle_blk_grows = 0;                  /* how often a table grew (tests: the fixtures must reach this path) */
/* Dhash::resize() (src/clib.h:341-355): a table of the next prime >= twice the size, live entries re-entered in slot order */
static void dh_grow(DH* h)
{
    KV* old = h->t;
    const uint32_t n_old = h->size1;
    if (n_old > (1u << 24)) { h->overflow = 1; return; }
    ++spdp_oracle_blk_grows;
    h->size1 = next_prime(2 * n_old);
    h->t = (KV*) malloc(sizeof(KV) * h->size1);
    dh_clear(h);
    for (uint32_t i = 0; i < n_old; ++i)
        if (old[i].val != h->undef) dh_map(h, old[i].key, 1)->val = old[i].val;
    free(old);
}
static KV* dh_map(DH* h, uint32_t key, int record)
{
    uint32_t v = key % h->size1, u = h->size2 - key % h->size2, v0 = v;
    KV* sh = h->t + v;
    while (sh->val != h->undef && sh->key != key) {
        v = (v + u) % h->size1;
        if (v == v0) {                                  /* the probe came round: the reference grows the table and goes on */
            dh_grow(h);                                 /* probing the NEW table from the OLD position with the old step */
            if (h->overflow) return record ? sh : 0;
        }
        sh = h->t + v;
    }
    if (sh->val == h->undef) { if (record) sh->key = key; else sh = 0; }
    return sh;
}
static KV* dh_incr(DH* h, uint32_t key)
{
    KV* sh = dh_map(h, key, 1);
    if (sh->val == h->undef) sh->val = 0;
    sh->val += 1;
    return sh;
}
static void dh_assign(DH* h, uint32_t key, int32_t val) { dh_map(h, key, 1)->val = val; }
static void dh_remove(DH* h, uint32_t key) { dh_map(h, key, 1)->val = h->undef; }

/* ---- PrQueue_wh<BlkScr> (ascending: data[0] holds the smallest score; replace = true) ---------------------------- */
typedef struct { uint32_t key; int32_t bscr; } BS;
typedef struct { BS* data; int capacity, front; DH hpos; } PQ;

static void pq_down(PQ* q, int k)
{
    BS v = q->data[k];
    const int kmax = q->front;
    while (k < kmax / 2) {
        int l = 2 * k + 1, r = l + 1;
        if (r < kmax && q->data[r].bscr < q->data[l].bscr) ++l;
        if (!(q->data[l].bscr < v.bscr)) break;
        q->data[k] = q->data[l];
        dh_assign(&q->hpos, q->data[k].key, k);
        k = l;
    }
    q->data[k] = v;
    dh_assign(&q->hpos, q->data[k].key, k);
}
static void pq_up(PQ* q, int k)
{
    BS v = q->data[k];
    int h = (k - 1) / 2;
    while (k && v.bscr < q->data[h].bscr) {
        q->data[k] = q->data[h];
        dh_assign(&q->hpos, q->data[k].key, k);
        k = h;
        h = (h - 1) / 2;
    }
    q->data[k] = v;
    dh_assign(&q->hpos, q->data[k].key, k);
}
static void pq_update(PQ* q, BS x)
{
    KV* kv = dh_map(&q->hpos, x.key, 0);
    int p = kv ? kv->val : -1;
    if (p < 0) {
        if (q->front < q->capacity) { q->data[q->front] = x; pq_up(q, q->front++); return; }
        p = 0;
    }
    if (q->data[p].bscr < x.bscr) {
        dh_remove(&q->hpos, q->data[p].key);
        q->data[p] = x;
        pq_down(q, p);
    }
}

/* ---- the search ------------------------------------------------------------------------------------------------ */
typedef struct {
    const BlkIndex* ix;
    const uint8_t* q; int qlen_total;                   /* query codes, [0, len) */
    uint32_t ww[3]; int xx[3]; uint32_t front[3]; int endss[3];
    const int32_t* pat[3];                              /* -> weight, width, wshift, exam */
    double app_c;
    int* bscr; int* ascr;                               /* 4 x nseg each, contiguous (a block index one past a row lands in the next) */
    PQ qa[4], qb[4];
    int sign[4], maxs[4], nhit[4], mmct[4], testword[4];
    DH hh;
} ST;

static int randbs(const BlkIndex* ix, uint32_t mmc)
{
    if (mmc < 128) return ix->rscrtab[mmc];
    if (ix->rbscoef == 0) return (int) ix->rbscons;
    const double x = (double) (mmc + 1);
    return (int) (ix->rbscoef * (ix->gdb ? log(x) : sqrt(x)) + ix->rbscons);
}

static int code_at(const ST* s, int i)
{
    if (i < 0 || i >= s->qlen_total) return 255;        /* outside the sequence: never a residue */
    const int c = s->q[i];
    return c < s->ix->convts ? s->ix->convtab[c] : 255;
}

/* Qwords::querywords(ss, d, rvs): ss = offset of the word's first position in the query */
static int query_words(ST* s, int ss, int d, int rvs)
{
    const BlkIndex* ix = s->ix;
    const uint32_t nalpha = (uint32_t) ix->nalpha, tab = (uint32_t) ix->tabsize;
    if (ix->kk == 1) {
        const int32_t* bp = s->pat[0];
        const int weight = bp[0], wshift = bp[2];
        const int32_t* exam = bp + 3 + (rvs ? weight : 0);
        int i = 0;
        s->ww[0] = 0; s->xx[0] = 0;
        for ( ; i < weight; ++i) {
            const uint32_t c = (uint32_t) code_at(s, ss + exam[i]);
            if (c >= nalpha) break;
            if (ix->drna) s->ww[0] = d >= 2 ? (s->ww[0] >> 2) + ((3 - c) << wshift) : (s->ww[0] << 2) + c;
            else          s->ww[0] = d >= 2 ? (tab * c + s->ww[0]) / nalpha : s->ww[0] * nalpha + c;
        }
        if (!ix->blkp[s->ww[0]]) return 0;              /* ubiquitous */
        if (ix->wscr[s->ww[0]] < 0) s->xx[0] = -1;
        if (i == weight) return ix->wscr[s->ww[0]];
        return -1;
    }
    for (int k = 0; k < ix->kk; ++k) s->ww[k] = tab;
    for (int k = 0; k < ix->kk; ++k) {
        if (ss >= s->endss[k]) break;
        const int32_t* bp = s->pat[k];
        const int weight = bp[0], wshift = bp[2];
        const int32_t* exam = bp + 3 + (rvs ? weight : 0);
        s->ww[k] = 0; s->xx[k] = 0;
        int i = 0;
        for ( ; i < weight; ++i) {
            const uint32_t c = (uint32_t) code_at(s, ss + exam[i]);
            if (c >= nalpha) break;
            if (ix->drna) s->ww[k] = d >= 2 ? (s->ww[k] >> 2) + ((3 - c) << wshift) : (s->ww[k] << 2) + c;
            else          s->ww[k] = d >= 2 ? (tab * c + s->ww[k]) / nalpha : s->ww[k] * nalpha + c;
        }
        if (i < weight) s->xx[k] = -1;
    }
    int c = 0, wdscr = 0;
    for (int k = 0; k < ix->kk; ++k) {
        if (s->ww[k] >= tab || ix->wscr[s->ww[k]] < 0) s->xx[k] = -1;
        else if (!s->xx[k] && ix->blkp[s->ww[k]]) { ++c; wdscr += ix->wscr[s->ww[k]]; }
    }
    if (c) return (int) ((double) wdscr / s->app_c);
    return -1;
}

static uint32_t list_at(const BlkIndex* ix, uint32_t w, int x) { return ix->blkb[ix->blkp[w] - 1 + x]; }
static void init_merge(ST* s)
{
    for (int k = 0; k < s->ix->kk; ++k)
        s->front[k] = (s->xx[k] >= 0 && s->ww[k] < (uint32_t) s->ix->tabsize && s->ix->blkp[s->ww[k]]) ? list_at(s->ix, s->ww[k], s->xx[k]) : 0;
}
static uint32_t next_merge(ST* s)
{
    const BlkIndex* ix = s->ix;
    uint32_t blk = s->front[0];
    if (ix->kk == 1) {
        s->front[0] = (++s->xx[0] < (int) ix->nblk[s->ww[0]]) ? list_at(ix, s->ww[0], s->xx[0]) : 0;
        return blk;
    }
    for (int j = 0; j + 1 < ix->kk; ++j) {              /* (the reference's k = 1 .. kk - 1 walks elements 0 .. kk - 2) */
        if (s->xx[j] < 0) continue;
        if (blk == 0) blk = s->front[j];
        if (s->front[j] && s->front[j] < blk) blk = s->front[j];
    }
    if (blk == 0) return 0;
    for (int j = 0; j + 1 < ix->kk; ++j) {
        if (s->xx[j] < 0) continue;
        if (blk == s->front[j])
            s->front[j] = (++s->xx[j] < (int) ix->nblk[s->ww[j]]) ? list_at(ix, s->ww[j], s->xx[j]) : 0;
    }
    return blk;
}

static int chrblk(const BlkIndex* ix, int m) { return ix->chr[2 * m + 1]; }
static int find_chr(const BlkIndex* ix, uint32_t blk)
{
    int lw = (int) (ix->bclw + ix->bcce * (blk - 1)) - 1;
    int up = (int) (ix->bcup + ix->bcce * (blk - 1)) + 1;
    if (lw < 0) lw = 0;
    if (up > ix->n_chr) up = ix->n_chr;
    if ((uint32_t) chrblk(ix, lw) > blk) lw = 0;
    if ((uint32_t) chrblk(ix, up) < blk) up = ix->n_chr;
    while (up - lw > 1) {
        const int md = (lw + up) / 2;
        if ((uint32_t) chrblk(ix, md) > blk) up = md;
        else if ((uint32_t) chrblk(ix, md + 1) > blk) return md;
        else lw = md;
    }
    return (uint32_t) chrblk(ix, up) > blk ? lw : up;
}

static int cmp_u32(const void* a, const void* b) { return (int) (*(const uint32_t*) a - *(const uint32_t*) b); }

/* SrchBlk::extract_to_work(d), d = 0 / 2: the significant blocks of both ends of one strand, sorted and paired */
static int extract_to_work(ST* s, int d, uint32_t* sw)
{
    const BlkIndex* ix = s->ix;
    const int e = d + 1, f = d >> 1;
    if (!s->sign[d] && !s->sign[e]) return 0;
    int j = 0;
    while (j < s->sign[d]) { sw[j] = s->qb[d].data[j].key << 1; ++j; }
    for (int k = 0; k < s->sign[e]; ++k) sw[j++] = (s->qb[e].data[k].key << 1) + 1;
    if (j == 1) { sw[0] &= ~1u; sw[1] = INT_MAX; return 1; }
    qsort(sw, j, sizeof(uint32_t), cmp_u32);
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
    int cp = find_chr(ix, p);
    sw[0] &= ~1u;
    int k = 0, c = 0;
    for (int i = 1; i < j; ++i) {
        const uint32_t q = sw[i] >> 1;
        const int qr = (int) (sw[i] & 1) ^ f;
        const int cq = find_chr(ix, q);
        sw[i] &= ~1u;
        const int st = (int) (q - p);
        if (cp == cq && (st < 2 || (!pr && qr && st <= ix->maxblock) || (pr == qr && st <= ix->extblock))) {
            if (!c++) sw[k++] = sw[i - 1];
        } else {
            sw[k++] = sw[i - 1] | (c ? 1u : 0u);
            c = 0;
        }
        p = q; pr = qr; cp = cq;
    }
    sw[k++] = sw[j - 1] + (c ? 1u : 0u);
    sw[k] = INT_MAX;
    return k;
}

typedef struct { int bscr, chr; uint32_t lb, rb, ub, db, zl, zr; int rvs; } BP;

/* the block-pair list of TestOutput (:2620-2672); returns the number of pairs */
static int build_pairs(ST* s, BP* bpair)
{
    const BlkIndex* ix = s->ix;
    const int nseg = ix->nseg;
    uint32_t* sigw[2];
    int sigm[2];
    BP* curbp = bpair;
    BP* const lstbp = bpair + ix->ncand;
    sigw[0] = (uint32_t*) calloc(4 * (size_t) ix->ncand + 4, sizeof(uint32_t));
    sigw[1] = sigw[0] + 2 * ix->ncand + 2;
    curbp->bscr = 0;
    for (int f = 0; f < 2; ++f) sigm[f] = extract_to_work(s, f << 1, sigw[f]);
    for (int f = 0; f < 2; ++f) {
        const int d = f << 1, e = d + 1;
        const int* bd = s->bscr + (size_t) d * nseg;
        const int* be = s->bscr + (size_t) e * nseg;
        uint32_t pu = 0;
        for (int i = 0; i < sigm[f]; ++i) {
            const uint32_t p = sigw[f][i] >> 1;
            uint32_t q = sigw[f][i + 1 < sigm[f] ? i + 1 : i];
            const int ispair = q & 1;
            q >>= 1;
            if (ispair) ++i; else q = p;
            const uint32_t qd = sigw[f][i + 1] >> 1;
            curbp->rvs = f;
            const int c1 = curbp->chr = find_chr(ix, q);
            curbp->zl = (uint32_t) chrblk(ix, c1);
            curbp->zr = (uint32_t) chrblk(ix, c1 + 1) - 1;
            curbp->lb = p;
            curbp->rb = q;
            curbp->bscr = 0;
            for (uint32_t r = curbp->lb; r <= curbp->rb; ++r) curbp->bscr += bd[r] + be[r];
            const uint32_t exb = (uint32_t) ix->extblock;
            uint32_t r = curbp->lb;
            uint32_t z = r > exb ? r - exb : 0;
            if (curbp->zl > z) z = curbp->zl;
            if (pu > z) z = pu;
            while (r && --r >= z && (bd[r] + be[r])) {
                curbp->lb = r;
                curbp->bscr += bd[r] + be[r];
            }
            curbp->ub = r > exb ? r - exb : 0;
            if (curbp->zl > curbp->ub) curbp->ub = curbp->zl;
            r = curbp->rb;
            z = r + exb;
            if (curbp->zr < z) z = curbp->zr;
            if (qd < z) z = qd;
            while (++r < z && (bd[r] + be[r])) {
                curbp->rb = r;
                curbp->bscr += bd[r] + be[r];
            }
            curbp->db = r + exb < curbp->zr ? r + exb : curbp->zr;
            pu = curbp->rb + 1;
            for (BP* w = curbp; --w >= bpair; ) {
                if (w[1].bscr > w->bscr) { const BP t = w[0]; w[0] = w[1]; w[1] = t; }
                else break;
            }
            if (curbp < lstbp) ++curbp;
        }
    }
    free(sigw[0]);
    return (int) (curbp - bpair);
}

static int* emit_vote(const ST* s, int* o)
{
    const int nseg = s->ix->nseg;
    *o++ = -2;
    for (int d = 0; d < 4; ++d) *o++ = s->sign[d];
    for (int d = 0; d < 4; ++d) *o++ = s->mmct[d];
    for (int d = 0; d < 4; ++d) *o++ = s->nhit[d];
    for (int d = 0; d < 4; ++d) *o++ = s->maxs[d];
    for (int d = 0; d < 4; ++d) *o++ = s->testword[d];
    for (int d = 0; d < 4; ++d) {
        *o++ = s->qb[d].front;
        for (int i = 0; i < s->qb[d].front; ++i) { *o++ = (int) s->qb[d].data[i].key; *o++ = s->qb[d].data[i].bscr; }
        *o++ = s->qa[d].front;
        for (int i = 0; i < s->qa[d].front; ++i) { *o++ = (int) s->qa[d].data[i].key; *o++ = s->qa[d].data[i].bscr; }
        for (int pass = 0; pass < 2; ++pass) {
            const int* a = (pass ? s->ascr : s->bscr) + (size_t) d * nseg;
            int* cnt = o++;
            *cnt = 0;
            for (int x = 0; x < nseg; ++x) if (a[x]) { *o++ = x; *o++ = a[x]; ++*cnt; }
        }
    }
    return o;
}

/* findblock's vote on one query.  stop_at: the TestOutput call (0-based) at which to stop; earlier calls are taken to have
 * returned 0 ("nothing found yet, go on").  Writes the state at that call (record -2) and the block pairs it would
 * build (record -3 with the number of pairs first, nine ints each); returns the number of ints written, 0 if findblock
 * ends before that call is reached (the query too short, or the `notry` rule), -1 on a table overflow. */
/* The reference keeps its queues' slot arrays from query to query (PrQueue_wh::reset only rewinds `front`), and the
 * stand-in rule at the end of findblock reads the first Nascr slots of the all-hits queue whether or not this query
 * filled them (src/blksrc.cc:3076-3082): what a query with fewer than Nascr hit blocks in some direction gets depends on
 * the query the same thread searched before.  The queues' position hashes persist too: reset() clears them but a table
 * that has grown (orphaned entries fill it: removing a key whose probe chain was cut leaves its live entry behind) stays
 * grown, and its size decides where keys land.  `carry` = that memory (NULL = a fresh process): 4 x (nascr + 1) slots of the
 * all-hits queues as key / score pairs, then the sizes of the eight position hashes (qa[0..3], qb[0..3]; 0 = initial);
 * read at the start, written back at the end. */
int spdp_oracle_blk_vote_carry(const BlkIndex* ix, const uint8_t* q, int q_len, int left, int right, int stop_at, int* out, int* carry);
int spdp_oracle_blk_vote(const BlkIndex* ix, const uint8_t* q, int q_len, int left, int right, int stop_at, int* out)
{
    return spdp_oracle_blk_vote_carry(ix, q, q_len, left, right, stop_at, out, 0);
}
int spdp_oracle_blk_vote_carry(const BlkIndex* ix, const uint8_t* q, int q_len, int left, int right, int stop_at, int* out, int* carry)
{
    ST s;
    memset(&s, 0, sizeof s);
    s.ix = ix; s.q = q; s.qlen_total = q_len;
    const int nshift = ix->nshift, nseg = ix->nseg;
    {
        const int32_t* p = ix->bitpat;
        for (int k = 0; k < ix->kk; ++k) { s.pat[k] = p; p += 3 + 2 * p[0]; }
    }
    const int qlen = right - left;
    if (qlen - (nshift + s.pat[0][1]) < 1) return 0;
    s.app_c = ix->kk > 1 ? pow((double) ix->nbitpat, ix->cfact) : 1.;
    for (int k = 0; k < ix->kk; ++k) s.endss[k] = right - s.pat[k][1];
    s.bscr = (int*) calloc(4 * (size_t) nseg + 2, sizeof(int));
    s.ascr = (int*) calloc(4 * (size_t) nseg + 2, sizeof(int));
    for (int d = 0; d < 4; ++d) {
        s.qa[d].capacity = ix->nascr; s.qa[d].data = (BS*) calloc(ix->nascr + 1, sizeof(BS));
        if (carry) memcpy(s.qa[d].data, carry + 2 * d * (ix->nascr + 1), sizeof(BS) * (ix->nascr + 1));
        s.qa[d].hpos.size1 = ix->ha_size1; s.qa[d].hpos.size2 = ix->ha_size2; s.qa[d].hpos.undef = -1;
        if (carry && carry[8 * (ix->nascr + 1) + d]) s.qa[d].hpos.size1 = carry[8 * (ix->nascr + 1) + d];
        s.qa[d].hpos.t = (KV*) malloc(sizeof(KV) * s.qa[d].hpos.size1); dh_clear(&s.qa[d].hpos);
        s.qb[d].capacity = ix->ncand; s.qb[d].data = (BS*) calloc(ix->ncand + 1, sizeof(BS));
        s.qb[d].hpos.size1 = ix->hb_size1; s.qb[d].hpos.size2 = ix->hb_size2; s.qb[d].hpos.undef = -1;
        if (carry && carry[8 * (ix->nascr + 1) + 4 + d]) s.qb[d].hpos.size1 = carry[8 * (ix->nascr + 1) + 4 + d];
        s.qb[d].hpos.t = (KV*) malloc(sizeof(KV) * s.qb[d].hpos.size1); dh_clear(&s.qb[d].hpos);
    }
    s.hh.size1 = ix->hh_size1; s.hh.size2 = ix->hh_size2; s.hh.undef = 0;
    s.hh.t = (KV*) malloc(sizeof(KV) * ix->hh_size1);
    /* init4: scan positions, as offsets into the query */
    int as[4][64];
    {
        int ss = left, ts = right - (s.pat[0][1] + nshift);
        int qph = (ts-- - ss) % nshift;
        for (int p = 0; p < nshift; ++p) {
            as[0][p] = as[2][p] = ss++;
            as[1][qph] = as[3][qph] = ts++;
            if (++qph == nshift) qph = 0;
        }
    }
    int n_out = 0, calls = 0;
    spdp_oracle_blk_last_forced = 0;
    int nohit = 0, sigpr = 0;
    int c = qlen / (nshift + nshift) - 1;
    const int is_short = qlen < ix->shortquery;
    const int at = right, ab = left;
    int meet[2] = {0, 0};
    uint32_t nmmc = 0;
    int notry = 0;
    int maxbscr[4] = {0, 0, 0, 0};
    int done = 0;
#define SNAP() do { int* o = emit_vote(&s, out); BP* bp = (BP*) calloc(ix->ncand + 2, sizeof(BP)); const int np = build_pairs(&s, bp); \
        *o++ = -3; *o++ = np; for (int i_ = 0; i_ < np; ++i_) { *o++ = bp[i_].bscr; *o++ = bp[i_].chr; *o++ = (int) bp[i_].lb; *o++ = (int) bp[i_].rb; \
        *o++ = (int) bp[i_].ub; *o++ = (int) bp[i_].db; *o++ = (int) bp[i_].zl; *o++ = (int) bp[i_].zr; *o++ = bp[i_].rvs; } free(bp); n_out = (int) (o - out); } while (0)
    while (!(meet[0] || meet[1]) && !done) {
        int totalsign = 0;
        for (int d = 0; d < 4; ++d) {
            if (meet[d / 2]) continue;
            const int prty = d % 2, e = prty ? d - 1 : d + 1, rvs = d >= 2;
            int* rscr = s.bscr + (size_t) d * nseg;
            int* acr = s.ascr + (size_t) d * nseg;
            int ms = prty ? ab : at;
            int maxp = 0;
            for (int sft = 0; sft < nshift; ++sft) {
                int* ws = &as[d][sft];
                if (!is_short) ms = as[e][sft];
                int cscr = 0, qq = 0, p = 0;
                dh_clear(&s.hh);
                do {
                    const int ss = *ws;
                    if (prty) *ws -= nshift; else *ws += nshift;
                    if (prty ^ (ss >= ms)) { meet[d / 2] = 1; break; }
                    const int wdscr = query_words(&s, ss, d, rvs);
                    if (wdscr < 0) break;
                    s.testword[d] += ix->kk;
                    if (wdscr == 0) { qq = 1; continue; }
                    init_merge(&s);
                    ++p; qq = 0;
                    cscr += wdscr;
                    uint32_t blk;
                    while ((blk = next_merge(&s)) != 0) {
                        KV* h = dh_incr(&s.hh, blk);
                        acr[blk] += wdscr;
                        { BS sb = {blk, acr[blk]}; pq_update(&s.qa[d], sb); }
                        if (p != h->val) {
                            h->val = 0;
                            if (prty) h = dh_incr(&s.hh, ++blk);
                            else if (blk) h = dh_incr(&s.hh, --blk);
                        }
                        if (p == h->val) {
                            ++qq;
                            rscr[blk] += wdscr;
                            if (rscr[blk] > maxbscr[d]) { maxbscr[d] = rscr[blk]; s.maxs[d] = sft; }
                            if (rscr[blk] >= randbs(ix, nmmc)) { BS sb = {blk, rscr[blk]}; pq_update(&s.qb[d], sb); s.sign[d] = s.qb[d].front; }
                        } else h->val = 0;
                    }
                } while (qq && cscr < ix->rscrtab[0]);
                if (p > maxp) maxp = p;
                if (s.maxs[d] == sft) nohit = !qq;
            }
            s.mmct[d] += nohit;
            s.nhit[d] += maxp;
            totalsign += s.sign[d];
        }
        if ((s.sign[0] && s.sign[1]) || (s.sign[2] && s.sign[3])) ++sigpr;
        if (((++nmmc % (uint32_t) ix->maxmmc) == 0 && totalsign) || sigpr > ix->minsigpr) {
            if (calls++ == stop_at) { SNAP(); done = 1; break; }
            c = 0;                                          /* TestOutput(0) returned 0 */
            if (++notry > ix->minsigpr) { done = 2; break; }
        }
    }
    if (!done) {
        if (!((s.sign[0] && s.sign[1]) || (s.sign[2] && s.sign[3]))) {
            c = -1;
            for (int d = 0; d < 4; ++d)
                for (int i = 0; i < ix->nascr; ++i) {
                    const BS bs = s.qa[d].data[i];
                    if (bs.key) {
                        BS sb = {bs.key, s.bscr[(size_t) d * nseg + bs.key]};
                        pq_update(&s.qb[d], sb);
                        c = s.sign[d] = s.qb[d].front;
                    }
                }
        }
        if (c != -1 && calls++ == stop_at) { spdp_oracle_blk_last_forced = 1; SNAP(); }
    }
#undef SNAP
    int overflow = s.hh.overflow;
    for (int d = 0; d < 4; ++d) {
        if (carry) {
            memcpy(carry + 2 * d * (ix->nascr + 1), s.qa[d].data, sizeof(BS) * (ix->nascr + 1));
            carry[8 * (ix->nascr + 1) + d] = (int) s.qa[d].hpos.size1;
            carry[8 * (ix->nascr + 1) + 4 + d] = (int) s.qb[d].hpos.size1;
        }
        overflow |= s.qa[d].hpos.overflow | s.qb[d].hpos.overflow;
        free(s.qa[d].data); free(s.qa[d].hpos.t); free(s.qb[d].data); free(s.qb[d].hpos.t);
    }
    free(s.hh.t); free(s.bscr); free(s.ascr);
    return overflow ? -1 : n_out;
}
