/* spdp_oracle_blkidx.c -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline): never linked into the product.
 *
 * CPU restatement of the reference's block-index builder for a nucleotide genome, `spaln -W -KD` (ogotoh/spaln v3.0.7):
 *   MakeBlk::idxblk                     src/blksrc.cc:1185-1241   two passes over the genome: count, then register
 *   MakeBlk::scan_genome                src/blksrc.cc:1111-1183   the serial block walk (no -t): a block ends after
 *                                                                 margin + blklen residues, the word state starts afresh
 *   MakeBlk::m_scan_genome / worker     src/blksrc.cc:1459-1593   the threaded walk (-t >= 1): every block is scanned from
 *                                                                 a fresh state over its blklen + margin residues, the
 *                                                                 last margin residues of a block open the next one
 *   Block::c2w                          src/blksrc.cc:448-464     residue -> words of every bit pattern, their phase
 *   Bitpat / Bitpat_wq::word / flaw     src/bitpat.cc:109-212
 *   Chash::countBlk / registBlk         src/blksrc.cc:402-425     a word counts once per block
 *   MakeBlk::blkscrtab(segn, blksz)     src/blksrc.cc:944-997     word scores, the cut-off, the posting-list layout
 *   MakeBlk::findChrBbound              src/blksrc.cc:583-596
 * Pinned to the arrays of the reference's own index files (tests/golden/blk_*.spdg: nblk, blkp, blkb, wscr, chr, pb2c of
 * `spaln -W -KD` runs; tests/golden/idx_t4.bkn.gz for the threaded walk): tests/test_oracle_blkidx.py.
 *
 * Residues are the library's codes (A 2, C 3, G 5, T 9); every other code is an ambiguous residue (uc == Nalpha). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t ktuple, nshift, blklen, maxgene, nbitpat, afact;
    uint32_t bitpat, bitpat2;
    int32_t threaded;
} OrcBlkBuildParams;

typedef struct {            /* Bitpat + the state of Bitpat_wq for one frame */
    int weight, width, spaced, reverse;
    int exam[2 * 32];
    uint32_t msb, fstat, queue0;        /* contiguous seed */
    uint8_t ring[32]; int qp;           /* spaced seed: the last `width` residues, 255 = BAD_RES */
    uint32_t tabsize;
} Pat;

static void pat_make(Pat* p, uint32_t npat, int reverse, uint32_t tabsize)
{
    memset(p, 0, sizeof *p);
    for (uint32_t x = npat; x; x >>= 1) { p->weight += x & 1; ++p->width; }
    int wt = 0;
    for (int w = 0; w < p->width; ++w) if (npat & (1u << w)) p->exam[wt++] = w;
    for (int w = 0; w < p->width; ++w) if (npat & (1u << (p->width - 1 - w))) p->exam[wt++] = w;
    p->spaced = p->width > p->weight; p->reverse = reverse;
    p->msb = 1u << (p->weight - 1); p->tabsize = tabsize;
}
static void pat_clear(Pat* p) { memset(p->ring, 255, sizeof p->ring); p->fstat = p->msb; p->queue0 = 0; p->qp = 0; }
static void pat_flaw(Pat* p)
{
    if (p->spaced) { p->ring[p->qp] = 255; if (++p->qp == p->width) p->qp = 0; }
    else { p->queue0 = 0; p->fstat = p->msb; }
}
/* returns the word; *ok = flawless() afterwards */
static uint32_t pat_word(Pat* p, uint32_t c, int* ok)
{
    if (!p->spaced) {
        p->fstat >>= 1;
        p->queue0 = (uint32_t) (((uint64_t) p->queue0 * 4 + c) % p->tabsize);
        *ok = p->fstat == 0;
        return p->queue0;
    }
    p->ring[p->qp] = (uint8_t) c;
    if (++p->qp == p->width) p->qp = 0;
    const int* exm = p->reverse ? p->exam + p->weight : p->exam;
    uint32_t w = 0;
    for (int k = 0; k < p->weight; ++k) {
        int q = p->qp + exm[k];
        if (q >= p->width) q -= p->width;
        if (p->ring[q] == 255) { p->fstat = 1; *ok = 0; return 0; }
        w = w * 4 + p->ring[q];
    }
    p->fstat = 0; *ok = 1;
    return w;
}

static int reduced(uint8_t code) { return code == 2 ? 0 : code == 3 ? 1 : code == 5 ? 2 : code == 9 ? 3 : 4; }

typedef struct { uint32_t word, block; } Hit;
static int hit_cmp(const void* a, const void* b)
{
    const Hit* x = (const Hit*) a; const Hit* y = (const Hit*) b;
    if (x->word != y->word) return x->word < y->word ? -1 : 1;
    return x->block < y->block ? -1 : (x->block > y->block);
}

/* out arrays of the caller: nblk[tabsize], blkp[tabsize] (1-based offset into blkb, 0 = no list), wscr[tabsize],
 * chr[2 * (n_chr + 1)] = {spos, segn}, b2c[3], head[8] = {WordNo, glen, AvrScr, MaxBlk, BytBlk, n_blocks, MinScr, tabsize};
 * *blkb malloc'ed (WordNo block numbers, 1-based).  Returns 0, or -1 (parameters), -2 (a word in more than 65535 blocks: the
 * reference's 16-bit counters wrap there), -3 (out of memory). */
int orc_blk_index_build(const uint8_t* codes, const int64_t* chr_off, int n_chr, const OrcBlkBuildParams* prm,
                        uint16_t* nblk, int32_t* blkp, int16_t* wscr, uint32_t** blkb, int32_t* chr, double* b2c, int64_t* head)
{
    const int K = prm->ktuple, nbit = prm->nbitpat, nshift = prm->nshift, blklen = prm->blklen;
    if (K < 1 || K > 16 || nbit < 1 || nbit > 5 || nshift < 1 || blklen < 1 || n_chr < 1) return -1;
    const uint64_t tab64 = 1ull << (2 * K);
    const uint32_t tabsize = (uint32_t) tab64;
    Pat pat[5];
    /* WordTab::WordTab (src/bitpat.cc:232-253): one pattern = BitPat; several = the contiguous k-mer, then BitPat forward /
     * mirrored, then Bitpat2 forward / mirrored */
    if (nbit == 1) pat_make(&pat[0], prm->bitpat, 0, tabsize);
    else {
        pat_make(&pat[0], K >= 32 ? 0xffffffffu : ((1u << K) - 1), 0, tabsize);
        for (int k = 1; k < nbit; ++k) pat_make(&pat[k], k < 3 ? prm->bitpat : prm->bitpat2, (k - 1) % 2, tabsize);
    }
    int max_width = 0;
    for (int k = 0; k < nbit; ++k) { if (pat[k].weight != K || pat[k].width > 32) return -1; if (pat[k].width > max_width) max_width = pat[k].width; }
    const int margin = max_width - 1;                   /* prelude = margin (MinOrf = 0 for DNA), src/blksrc.cc:441-445 */
    const int64_t s_size = (int64_t) margin + blklen;

    uint32_t* tcount = (uint32_t*) calloc(tabsize, sizeof(uint32_t));
    size_t cap = 1 << 20, n_hit = 0;
    Hit* hits = (Hit*) malloc(cap * sizeof(Hit));
    if (!tcount || !hits) { free(tcount); free(hits); return -3; }

    uint32_t block = 0;                                 /* blocks stored so far; a block's number in the file is 1-based */
    int64_t spos = 0;
    for (int c = 0; c < n_chr; ++c) {
        const uint8_t* s = codes + chr_off[c];
        const int64_t L = chr_off[c + 1] - chr_off[c];
        chr[2 * c] = (int32_t) spos; chr[2 * c + 1] = (int32_t) (block + 1);
        const int64_t nb = L <= 0 ? 0 : (L < s_size ? 1 : 1 + (L - margin) / blklen);
        for (int64_t b = 0; b < nb; ++b) {
            /* residues this block's word state sees, from a fresh state */
            const int64_t lo = prm->threaded ? b * blklen : (b == 0 ? 0 : b * blklen + margin);
            int64_t hi = (b + 1) * blklen + margin;
            if (hi > L) hi = L;
            const int64_t t_hi = (b + 1) * blklen;      /* words ending before it count in tcount (posinblk < blklen) */
            uint16_t ss = 0;
            for (int k = 0; k < nbit; ++k) pat_clear(&pat[k]);
            for (int64_t j = lo; j < hi; ++j) {
                const int uc = reduced(s[j]);
                if (uc == 4) { ss = 0; for (int k = 0; k < nbit; ++k) pat_flaw(&pat[k]); continue; }
                ++ss;                                    /* (a SHORT in the reference: wraps at 65536) */
                for (int k = 0; k < nbit; ++k) {
                    int ok;
                    const uint32_t w = pat_word(&pat[k], (uint32_t) uc, &ok);
                    if (!ok) continue;
                    if (j < t_hi) ++tcount[w];
                    /* `nw % Nshift` with Nshift an unsigned member (src/bitpat.h:121): a word that is flawless before `width`
                     * residues have gone by -- the ambiguous one lay under a '0' of a spaced pattern -- is taken at the
                     * phases of 2^32 + nw, not of nw */
                    const int nw = (int) ss - pat[k].width;
                    if ((uint32_t) nw % (uint32_t) nshift == 0) {
                        if (n_hit == cap) { cap *= 2; Hit* h2 = (Hit*) realloc(hits, cap * sizeof(Hit)); if (!h2) { free(hits); free(tcount); return -3; } hits = h2; }
                        hits[n_hit].word = w; hits[n_hit].block = block + 1; ++n_hit;
                    }
                }
            }
            ++block;
        }
        spos += L;
    }
    chr[2 * n_chr] = (int32_t) spos; chr[2 * n_chr + 1] = (int32_t) (block + 1);
    /* a word counts once per block; lists in block order */
    qsort(hits, n_hit, sizeof(Hit), hit_cmp);
    size_t n_u = 0;
    for (size_t i = 0; i < n_hit; ++i) if (i == 0 || hits[i].word != hits[n_u - 1].word || hits[i].block != hits[n_u - 1].block) hits[n_u++] = hits[i];
    uint32_t* cnt = (uint32_t*) calloc(tabsize, sizeof(uint32_t));
    if (!cnt) { free(hits); free(tcount); return -3; }
    for (size_t i = 0; i < n_u; ++i) ++cnt[hits[i].word];
    for (uint32_t w = 0; w < tabsize; ++w) if (cnt[w] > 65535) { free(cnt); free(hits); free(tcount); return -2; }

    /* blkscrtab(segn, blksz), src/blksrc.cc:944-997 */
    const uint32_t segn = block;
    const uint32_t blksz = segn ? (uint32_t) ((uint32_t) spos / segn) : 0;
    uint32_t m = 0;
    for (uint32_t w = 0; w < tabsize; ++w) if (tcount[w]) ++m;
    const double basescr = log((double) segn);
    short min_scr = (short) -(100 * log((double) prm->afact * blksz / m));
    if (min_scr < 0) min_scr = 0;
    double avr = 0.;
    int64_t word_no = 0; uint32_t max_blk = 0;
    m = 0;
    for (uint32_t w = 0; w < tabsize; ++w) {
        if (cnt[w]) {
            /* (short) of +inf when the word never counted in tcount: cvttsd2si's 0x80000000, whose low half is 0 */
            const short sc = tcount[w] ? (short) (100 * (basescr - log((double) tcount[w] / nbit))) : 0;
            if (sc > min_scr) { ++m; word_no += cnt[w]; wscr[w] = sc; avr += sc; if (cnt[w] > max_blk) max_blk = cnt[w]; }
            else wscr[w] = min_scr;
        } else wscr[w] = -1;
    }
    *blkb = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) (word_no > 0 ? word_no : 1));
    if (!*blkb) { free(cnt); free(hits); free(tcount); return -3; }
    int64_t at = 0;
    size_t hi_ = 0;
    for (uint32_t w = 0; w < tabsize; ++w) {
        const size_t first = hi_;
        hi_ += cnt[w];
        if (cnt[w] && wscr[w] > min_scr) {
            blkp[w] = (int32_t) (at + 1); nblk[w] = (uint16_t) cnt[w];
            for (size_t i = first; i < hi_; ++i) (*blkb)[at++] = hits[i].block;
        } else { blkp[w] = 0; nblk[w] = 0; }
    }
    /* findChrBbound */
    const double B = (double) segn;
    b2c[0] = b2c[1] = b2c[2] = 0.;
    for (int k = 0; k <= n_chr; ++k) {
        const double off = k * B - n_chr * (double) (chr[2 * k + 1] - 1);
        if (off < b2c[0]) b2c[0] = off;
        if (off > b2c[1]) b2c[1] = off;
    }
    b2c[0] /= B; b2c[1] /= B;
    head[0] = word_no; head[1] = spos; head[2] = m ? (int64_t) (uint16_t) (avr / m) : 0; head[3] = max_blk;
    head[4] = segn <= 65535 ? 2 : 4; head[5] = segn; head[6] = min_scr; head[7] = tabsize;
    free(cnt); free(hits); free(tcount);
    return 0;
}

/* ---- the translated index, `spaln -W -KP` (<db>.bkp): amino-acid words of the six reading frames ----------------------------
 *   Block::c2w6 / c2w6_pp               src/blksrc.cc:466-532     a residue completes one codon per strand; its class joins the
 *                                                                 frame's word; words are taken every Nshift codons counted from
 *                                                                 the start of the open reading frame, wait MinOrf residues in a
 *                                                                 ring per strand, and are struck from it when their frame
 *                                                                 closes before MinOrf nucleotides
 *   ReducWord::ReducWord (g2r)          src/bitpat.cc:58-106      codon -> class of the reduced amino-acid alphabet
 *   Bitpat_wq::word / flaw              src/bitpat.cc:178-212     (contiguous words only here: -KP's default, one pattern)
 *   MakeBlk::scan_genome / m_scan_genome / harvest   :1111-1183, :1485-1593   the walks; a chromosome's last block flushes the ring
 *                                                                 up to blklen; what lies behind is lost with the reset that
 *                                                                 closes the block, though a block number is spent on it
 *   WordTab::reset                      src/bitpat.cc:439-461     what a new block starts from
 *   MakeBlk::blkscrtab(segn)            src/blksrc.cc:879-942     word scores with the composition term, the cut-off
 * The composition terms (MakeBlk::prepacomp, :844-877: functions of the reference's matrix tables) are parameters.
 * Pinned to files of `spaln -W -KP` with and without -t: tests/test_oracle_blkidx.py. */
typedef struct {
    OrcBlkBuildParams b;
    int32_t nalpha, minorf;
    double aaafact;
    double acomp[24];
    uint8_t codon_class[64];    /* codon 16 * b1 + 4 * b2 + b3 (A C G T = 0 1 2 3) -> class, >= nalpha: none (stop, Sec) */
} OrcBlkBuildParamsP;

typedef struct {
    uint32_t cc[6], xx[3], ss6[6], fstat[6], word[6];
    int p, w_qp;
    uint32_t* ring;             /* 2 * minorf */
} TronState;

#define TRON_BAD 0xffffffffu

static void tron_reset(TronState* s, int minorf, uint32_t msb)
{
    for (int q = 0; q < 6; ++q) { s->ss6[q] = 0; s->fstat[q] = msb; s->word[q] = 0; }
    for (int q = 0; q < 3; ++q) { s->cc[q] = 0; s->xx[q] = 4; }
    s->w_qp = 0;
    for (int i = 0; i < 2 * minorf; ++i) s->ring[i] = TRON_BAD;
}

typedef struct { Hit* hits; size_t n, cap; int failed; } HitList;
static void hit_add(HitList* h, uint32_t w, uint32_t block)
{
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : (1 << 20);
        Hit* h2 = (Hit*) realloc(h->hits, h->cap * sizeof(Hit));
        if (!h2) { h->failed = 1; return; }
        h->hits = h2;
    }
    h->hits[h->n].word = w; h->hits[h->n].block = block; ++h->n;
}

/* one residue (Block::c2w6); returns -4 when the reference would write outside its ring */
static int tron_residue(TronState* s, uint32_t uc, const OrcBlkBuildParamsP* prm, uint32_t tabsize, uint32_t msb, uint32_t* tcc,
                        HitList* out, uint32_t block)
{
    const int K = prm->b.ktuple, nshift = prm->b.nshift, wq = prm->minorf;
    int p = s->p;
    s->cc[p] = s->cc[p + 3] = 0;
    if (uc == 4) { s->xx[0] = s->xx[1] = s->xx[2] = 4; }
    else for (int q = 0; q < 3; ++q) {
        s->cc[q] = ((s->cc[q] << 2) + uc) & 63;
        s->cc[q + 3] = (((3 - uc) << 4) + (s->cc[q + 3] >> 2)) & 63;
        s->xx[p] >>= 1;
    }
    if (++p == 3) p = 0;
    s->p = p;
    int wqp = s->w_qp, wq_base = 0;
    if (++s->w_qp == wq) s->w_qp = 0;
    for (int q = p; q < 6; q += 3, wqp += wq, wq_base += wq) {
        const int orf_len = 3 * (int) s->ss6[q];
        if (!s->xx[p]) uc = prm->codon_class[s->cc[q]];
        if (!s->xx[p] && uc < (uint32_t) prm->nalpha) s->ss6[q] = (s->ss6[q] + 1) & 0xffff;
        else s->ss6[q] = 0;
        uint32_t w = TRON_BAD;
        if (s->ss6[q]) {
            s->fstat[q] >>= 1;
            s->word[q] = (uint32_t) (((uint64_t) s->word[q] * prm->nalpha + uc) % tabsize);
            if (s->fstat[q] == 0) {
                w = s->word[q];
                if (tcc) ++tcc[w];
                const int nw = (int) s->ss6[q] - K;
                if (nw < 0 || (uint32_t) nw % (uint32_t) nshift) w = TRON_BAD;
            }
        } else {
            s->word[q] = 0; s->fstat[q] = msb;
            int nw = orf_len / 3 - K;
            if (0 <= nw && orf_len < wq) {
                const int sp = nw % nshift;
                nw = (nw + sp) / nshift;
                int qp = wqp - (sp + 1) * 3;
                for ( ; nw-- >= 0; qp -= 3 * nshift) {
                    if (qp < wq_base) qp += wq;
                    if (qp < 0 || qp >= 2 * wq) return -4;
                    s->ring[qp] = TRON_BAD;
                }
            }
        }
        if (wq) { const uint32_t t = s->ring[wqp]; s->ring[wqp] = w; w = t; }
        if (w != TRON_BAD) hit_add(out, w, block);
    }
    return 0;
}
static void tron_flush(TronState* s, int n, int wq, HitList* out, uint32_t block)     /* Block::c2w6_pp */
{
    while (n-- > 0) {
        int wqp = s->w_qp;
        if (++s->w_qp == wq) s->w_qp = 0;
        for (int q = 0; q < 2; ++q, wqp += wq) {
            const uint32_t w = s->ring[wqp];
            s->ring[wqp] = TRON_BAD;
            if (w != TRON_BAD) hit_add(out, w, block);
        }
    }
}

/* as orc_blk_index_build; -4: the reference would write outside its ring with these parameters */
int orc_blk_index_build_tron(const uint8_t* codes, const int64_t* chr_off, int n_chr, const OrcBlkBuildParamsP* prm,
                             uint16_t* nblk, int32_t* blkp, int16_t* wscr, uint32_t** blkb, int32_t* chr, double* b2c, int64_t* head)
{
    const int K = prm->b.ktuple, nshift = prm->b.nshift, blklen = prm->b.blklen, wq = prm->minorf, na = prm->nalpha;
    if (K < 1 || K > 7 || prm->b.nbitpat != 1 || nshift < 1 || blklen < 1 || n_chr < 1 || na < 2 || na > 20 || wq < 1 || wq > 4096) return -1;
    uint64_t tab64 = 1;
    for (int i = 0; i < K; ++i) tab64 *= (uint64_t) na;
    if (tab64 > (1ull << 31)) return -1;
    const uint32_t tabsize = (uint32_t) tab64, msb = 1u << (K - 1);
    const int prelude = 3 * K - 1, margin = prelude + wq;       /* src/blksrc.cc:440-445 */
    const int64_t s_size = (int64_t) margin + blklen;
    uint32_t* tcount = (uint32_t*) calloc(tabsize, sizeof(uint32_t));
    TronState st;
    memset(&st, 0, sizeof st);
    st.ring = (uint32_t*) malloc(sizeof(uint32_t) * 2 * (size_t) wq);
    HitList hl = {0, 0, 0, 0};
    if (!tcount || !st.ring) { free(tcount); free(st.ring); return -3; }
    uint32_t block = 0;
    int64_t spos = 0;
    int rc = 0;
    for (int c = 0; c < n_chr && !rc; ++c) {
        const uint8_t* s = codes + chr_off[c];
        const int64_t L = chr_off[c + 1] - chr_off[c];
        chr[2 * c] = (int32_t) spos; chr[2 * c + 1] = (int32_t) (block + 1);
        const int64_t nb = L <= 0 ? 0 : (L < s_size ? 1 : 1 + (L - margin) / blklen);
        for (int64_t b = 0; b < nb && !rc; ++b) {
            const int64_t lo = prm->b.threaded ? b * blklen : (b == 0 ? 0 : b * blklen + margin);
            int64_t hi = (b + 1) * blklen + margin;
            if (hi > L) hi = L;
            const int pos0 = (!prm->b.threaded && b) ? margin : 0;      /* posinblk of the block's first residue */
            tron_reset(&st, wq, msb);
            for (int64_t j = lo; j < hi && !rc; ++j) {
                const uint32_t uc = (uint32_t) reduced(s[j]);
                rc = tron_residue(&st, uc, prm, tabsize, msb, pos0 + (j - lo) < blklen ? tcount : 0, &hl, block + 1);
            }
            if (b == nb - 1) {                                      /* the chromosome's last block: the ring is emptied */
                const int64_t n = pos0 + (hi - lo);
                const int rest = (int) (n > blklen ? n - blklen : 0);
                tron_flush(&st, wq - rest, wq, &hl, block + 1);
                /* store_blk: the block is closed and the state reset -- the ring with it; c2w6_pp(rest) then finds it empty, but
                 * the block it would have filled has its number */
                if (rest > 0) ++block;
            }
            ++block;
        }
        spos += L;
    }
    free(st.ring);
    if (rc || hl.failed) { free(tcount); free(hl.hits); return rc ? rc : -3; }
    chr[2 * n_chr] = (int32_t) spos; chr[2 * n_chr + 1] = (int32_t) (block + 1);
    Hit* hits = hl.hits; const size_t n_hit = hl.n;
    if (n_hit) qsort(hits, n_hit, sizeof(Hit), hit_cmp);
    size_t n_u = 0;
    for (size_t i = 0; i < n_hit; ++i) if (i == 0 || hits[i].word != hits[n_u - 1].word || hits[i].block != hits[n_u - 1].block) hits[n_u++] = hits[i];
    uint32_t* cnt = (uint32_t*) calloc(tabsize, sizeof(uint32_t));
    if (!cnt) { free(hits); free(tcount); return -3; }
    for (size_t i = 0; i < n_u; ++i) ++cnt[hits[i].word];
    for (uint32_t w = 0; w < tabsize; ++w) if (cnt[w] > 65535) { free(cnt); free(hits); free(tcount); return -2; }

    /* blkscrtab(segn), src/blksrc.cc:879-942 */
    const uint32_t segn = block;
    const double basescr = log((double) segn);
    const double deltaa = prm->acomp[0] - prm->acomp[na - 1];
    double alc = prm->aaafact > 0 ? K * prm->acomp[0] : 0.;
    double avr = 0.;
    uint32_t m = 0;
    for (uint32_t w = 0; w < tabsize; ++w) {
        if (tcount[w]) {
            ++m;
            short sc = (short) (100 * (basescr - log((double) tcount[w] / 1)));
            if (prm->aaafact > 0) sc = (short) (sc + (short) alc);
            wscr[w] = sc;
            avr += sc;
        } else wscr[w] = 0;
        if (prm->aaafact > 0) {
            int p = 0, q = 0;
            for (uint32_t x = w + 1; (q = (int) (x % (uint32_t) na)) == 0; x /= (uint32_t) na) ++p;
            if (p) alc += p * deltaa;
            alc += prm->acomp[q] - prm->acomp[q - 1];
        }
    }
    avr /= m;
    short min_scr = (short) (avr - 100 * (1 + prm->aaafact) * log((double) prm->b.afact));
    if (min_scr < 0) min_scr = 0;
    int64_t word_no = 0; uint32_t max_blk = 0;
    for (uint32_t w = 0; w < tabsize; ++w) {
        if (!cnt[w]) wscr[w] = -1;
        else if (wscr[w] > min_scr) { word_no += cnt[w]; if (cnt[w] > max_blk) max_blk = cnt[w]; }
        else { wscr[w] = 0; }
    }
    *blkb = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) (word_no > 0 ? word_no : 1));
    if (!*blkb) { free(cnt); free(hits); free(tcount); return -3; }
    int64_t at = 0;
    size_t hi_ = 0;
    for (uint32_t w = 0; w < tabsize; ++w) {
        const size_t first = hi_;
        hi_ += cnt[w];
        if (cnt[w] && wscr[w] > min_scr) {
            blkp[w] = (int32_t) (at + 1); nblk[w] = (uint16_t) cnt[w];
            for (size_t i = first; i < hi_; ++i) (*blkb)[at++] = hits[i].block;
        } else { blkp[w] = 0; nblk[w] = 0; }
    }
    const double B = (double) segn;
    b2c[0] = b2c[1] = b2c[2] = 0.;
    for (int k = 0; k <= n_chr; ++k) {
        const double off = k * B - n_chr * (double) (chr[2 * k + 1] - 1);
        if (off < b2c[0]) b2c[0] = off;
        if (off > b2c[1]) b2c[1] = off;
    }
    b2c[0] /= B; b2c[1] /= B;
    head[0] = word_no; head[1] = spos; head[2] = (int64_t) (uint16_t) avr; head[3] = max_blk;
    head[4] = segn <= 65535 ? 2 : 4; head[5] = segn; head[6] = min_scr; head[7] = tabsize;
    free(cnt); free(hits); free(tcount);
    return 0;
}
