// spdp_blk_index_io.cpp -- reads the reference's block index file (<db>.bkn, written by `spaln -W`) and derives the search
// parameters the reference derives when it opens one.  Host only; the result is a SpdpBlkIndexDesc for spdp_blk_index_create.
//
// What it follows (ogotoh/spaln v3.0.7): SrchBlk::ReadBlkInfo + read_pwc + read_blk_dt (src/blksrc.cc:1697-1858: the
// file is BlkWcPrm, ContBlk, Block2Chr, CHROMO[ChrNo + 1], Nblk[TabSize], the posting-list offsets, the lists, wscr[TabSize],
// ConvTab -- struct images of a 64-bit little-endian build), SrchBlk::initialize (:2179-2227: patterns, Randbs, maxmmc,
// MaxBlock / ExtBlock / ExtBlockL, shortquery, Ncand), Bitpat::Bitpat (src/bitpat.cc:109-143), Randbs::Randbs (:2047-2062)
// and the table sizes Dhash picks (src/clib.h:257-267).  Only the current format (version 26, 2- or 4-byte block numbers)
// of a nucleotide index is read; older versions and the 3-byte form are refused.
#include "../../include/spdp.h"
#include "spdp_blk_dev.h"
#include "spdp_gencode.h"
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <new>
#include <vector>

namespace {

#pragma pack(push, 1)
struct FileWcPrm { uint32_t Nalpha, Ktuple, Bitpat2, TabSize, BitPat, Nshift, blklen, MaxGene; int16_t Nbitpat; uint16_t afact; };
#pragma pack(pop)
struct FileContBlk {                            // ContBlk as the reference's compiler lays it out (pointers: whatever the writer held)
    uint32_t ConvTS; uint32_t pad0;
    uint64_t WordNo, WordSz, ChrNo, glen;
    uint16_t AvrScr, MaxBlk, BytBlk, VerNo;
    uint64_t p_Nblk, p_blkp, p_blkb, p_wscr, p_ChrID;
};
struct FileChromo { uint32_t spos, segn; };
static_assert(sizeof(FileWcPrm) == 36 && sizeof(FileContBlk) == 88, "block index header layout");

struct HostIndex {
    SpdpBlkIndexDesc d;
    FileWcPrm wcp; FileContBlk wc; double b2c[3];       // the header as the file holds it (spdp_blk_index_write)
    std::vector<FileChromo> chrid;
    std::vector<uint8_t> convtab;
    std::vector<uint16_t> nblk;
    std::vector<int16_t> wscr;
    std::vector<int32_t> blkp, rscrtab, chr, bitpat;
    std::vector<uint32_t> blkb;
    std::string err;
};

// Bitpat(npat): the positions a pattern examines, forward and mirrored
void add_pattern(std::vector<int32_t>& out, uint32_t npat)
{
    int width = 0, weight = 0;
    for (uint32_t x = npat; x; x >>= 1) { weight += x & 1; ++width; }
    out.push_back(weight); out.push_back(width); out.push_back(2 * (weight - 1));
    for (int w = 0; w < width; ++w) if (npat & (1u << w)) out.push_back(w);
    for (int w = 0; w < width; ++w) if (npat & (1u << (width - 1 - w))) out.push_back(w);
}

// the size Dhash(n, ..) ends up with: supprime(int(1.2f * n)), at least 31
int dhash_size(int n)
{
    const int want = (int) (1.2f * (float) n);
    return want < 31 ? 31 : (int) blk_next_prime((uint32_t) want);
}

bool read_all(FILE* f, void* p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }

// SrchBlk::initialize's part (src/blksrc.cc:2179-2227): what the search derives from the header and the options
bool derive_search_params(HostIndex* h, const SpdpBlkSearchOpts& o, std::string& why)
{
    const FileWcPrm& wcp = h->wcp; const FileContBlk& wc = h->wc; const double* b2c = h->b2c; const std::vector<FileChromo>& chrid = h->chrid;
    h->bitpat.clear(); h->chr.clear();
    SpdpBlkIndexDesc& d = h->d;
    memset(&d, 0, sizeof d);
    d.nalpha = (int32_t) wcp.Nalpha; d.tabsize = (int32_t) wcp.TabSize; d.nshift = (int32_t) wcp.Nshift; d.nbitpat = wcp.Nbitpat;
    d.convts = (int32_t) wc.ConvTS; d.n_chr = (int32_t) wc.ChrNo; d.maxblk = wc.MaxBlk;
    d.kk = wcp.Nbitpat / 2 + 1; d.drna = wcp.Nalpha == 4 ? 1 : 0;
    if (d.kk < 1 || d.kk > 3) { why = ("number of bit patterns out of range"); return false; }
    // a header this code would divide by or shift with: k-mer size 1 .. 16 (a word is 32 bits at Nalpha = 4), every pattern of
    // that weight (Bitpat's weight is the k-mer size for all patterns of an index, src/blksrc.cc:1697-1724)
    auto weight_of = [](uint32_t x) { int w = 0; for (; x; x >>= 1) w += (int) (x & 1); return w; };
    if (wcp.Ktuple < 1 || wcp.Ktuple > 16) { why = ("k-mer size out of range"); return false; }
    if (weight_of(wcp.BitPat) != (int) wcp.Ktuple || (wcp.Nbitpat > 3 && weight_of(wcp.Bitpat2) != (int) wcp.Ktuple)) {
        why = ("bit pattern weight differs from the k-mer size"); return false;
    }
    if (wcp.Nshift < 1 || wcp.blklen < 1) { why = ("shift count / block length out of range"); return false; }
    if (wcp.Nbitpat == 1) add_pattern(h->bitpat, wcp.BitPat);
    else { add_pattern(h->bitpat, (1u << wcp.Ktuple) - 1); add_pattern(h->bitpat, wcp.BitPat); }
    if (wcp.Nbitpat > 3) add_pattern(h->bitpat, wcp.Bitpat2);
    const int weight0 = h->bitpat[0];
    // Randbs(avr = AvrScr * weight / Nshift, gdb)
    const double avr = (double) wc.AvrScr * weight0 / wcp.Nshift;
    const float coef = (float) (o.rbs_fact * avr), cons = (float) (o.rbs_base * avr);
    d.rbscoef = coef; d.rbscons = cons; d.gdb = o.genomic_db ? 1 : 0;
    h->rscrtab.resize(128);
    for (int i = 0; i < 128; ++i) {
        const double x = (double) (i + 1);
        h->rscrtab[i] = (int) (coef * (d.gdb ? log(x) : sqrt(x)) + cons);
    }
    d.maxmmc = (o.max_mmc == 0 || o.max_mmc > INT32_MAX / weight0 || o.local) ? INT32_MAX : weight0 * o.max_mmc / (int) wcp.Nshift;
    d.nseg = (int32_t) chrid[wc.ChrNo].segn;
    d.minsigpr = o.min_sigpr; d.ncand = o.max_out + 10; d.nascr = std::max(1, std::min(o.nascr, d.ncand));
    d.maxblock = (int32_t) (wcp.MaxGene / wcp.blklen);
    d.extblock = o.ext_block > 0 ? o.ext_block : o.max_intron_len / (int) wcp.blklen + 1;
    d.extblockl = d.maxblock / 2 + 1;
    d.shortquery = 8 * (int32_t) wcp.Ktuple;
    d.blklen = (int32_t) wcp.blklen;
    d.hh_size = dhash_size(2 * wc.MaxBlk); d.hb_size = dhash_size(2 * d.ncand); d.ha_size = dhash_size(2 * d.nascr);
    d.hh_step = d.hb_step = d.ha_step = 8;
    d.bclw = b2c[0]; d.bcup = b2c[1]; d.bcce = b2c[2];
    d.cfact = o.cfact;
    for (const FileChromo& c : chrid) { h->chr.push_back((int32_t) c.spos); h->chr.push_back((int32_t) c.segn); }
    d.convtab = h->convtab.data(); d.nblk = h->nblk.data(); d.wscr = h->wscr.data(); d.blkp = h->blkp.data();
    d.blkb = h->blkb.data(); d.n_words = (int64_t) h->blkb.size(); d.rscrtab = h->rscrtab.data(); d.chr = h->chr.data();
    d.bitpat = h->bitpat.data(); d.n_bitpat = (int32_t) h->bitpat.size();
    if (d.nseg < 2 || d.maxmmc < 1) { why = ("derived parameters out of range"); return false; }
    return true;
}

}   // namespace

extern "C" void spdp_blk_search_opts_default(SpdpBlkSearchOpts* o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->max_out = 1; o->max_mmc = 15; o->min_sigpr = 3; o->nascr = 2;
    o->cfact = 0.75; o->rbs_fact = 0.4f; o->rbs_base = 3.f; o->genomic_db = 1;
}

extern "C" SpdpBlkIndexHost* spdp_blk_index_read(const char* path, const SpdpBlkSearchOpts* opts, char* err, int err_cap)
{
    HostIndex* h = new HostIndex;
    auto fail = [&](const std::string& m) -> SpdpBlkIndexHost* {
        if (err && err_cap > 0) snprintf(err, (size_t) err_cap, "%s: %s", path ? path : "(null)", m.c_str());
        delete h;
        return nullptr;
    };
    SpdpBlkSearchOpts o;
    if (opts) o = *opts; else spdp_blk_search_opts_default(&o);
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) return fail("cannot open");
    FileWcPrm wcp; FileContBlk wc; double b2c[3];
    if (!read_all(f, &wcp, sizeof wcp) || !read_all(f, &wc, sizeof wc) || !read_all(f, b2c, sizeof b2c)) { fclose(f); return fail("short header"); }
    if (wc.VerNo != 26) { fclose(f); return fail("only index version 26 is read (spaln 3.0.x)"); }
    if (wc.BytBlk != 2 && wc.BytBlk != 4) { fclose(f); return fail("3-byte block numbers are not read"); }
    // Nalpha = 4: a nucleotide index (-KD, .bkn); otherwise the amino-acid words of a translated genome (-KP, .bkp: protein
    // queries, SrchBlk's DvsP = 1 branch) or of a protein database (-KA); src/blksrc.cc:2184-2187
    if (wcp.Nalpha < 2 || wcp.Nalpha > 32) { fclose(f); return fail("alphabet size out of range"); }
    if (wcp.TabSize == 0 || wcp.TabSize > (1u << 30) || wcp.Nshift == 0 || wcp.Nshift > SPDP_BLK_MAX_SHIFT || wcp.blklen == 0 ||
        wc.ChrNo == 0 || wc.ChrNo > (1u << 24) || wc.WordNo > (1ull << 32) || wc.WordSz > (1ull << 33) || wc.ConvTS == 0 || wc.ConvTS > 256) {
        fclose(f); return fail("header values out of range");
    }
    if (wc.WordNo > (uint64_t) INT32_MAX) { fclose(f); return fail("more postings than a 32-bit list offset (blkp) can address"); }
    if (wcp.Ktuple == wcp.BitPat) wcp.BitPat = (1u << wcp.BitPat) - 1;
    {   // the header's sizes against what the file really holds, BEFORE anything is sized from them
        const long at = ftell(f);
        if (at < 0 || fseek(f, 0, SEEK_END) != 0) { fclose(f); return fail("cannot seek"); }
        const long end = ftell(f);
        if (end < at || fseek(f, at, SEEK_SET) != 0) { fclose(f); return fail("cannot seek"); }
        const uint64_t rest = (uint64_t) (end - at);
        const uint64_t need = ((uint64_t) wc.ChrNo + 1) * sizeof(FileChromo) + (uint64_t) wcp.TabSize * (2 + 4 + 2) +
                              (uint64_t) wc.WordSz * 2 + wc.ConvTS;
        if (need > rest) { fclose(f); return fail("header sizes exceed the file"); }
    }
    std::vector<FileChromo> chrid;
    try {
        chrid.resize(wc.ChrNo + 1);
        h->nblk.resize(wcp.TabSize); h->blkp.resize(wcp.TabSize); h->wscr.resize(wcp.TabSize);
        h->blkb.resize(wc.WordNo); h->convtab.resize(wc.ConvTS);
    } catch (const std::bad_alloc&) { fclose(f); return fail("out of memory for the index tables"); }
    bool ok = read_all(f, chrid.data(), chrid.size() * sizeof(FileChromo)) &&
              read_all(f, h->nblk.data(), (size_t) wcp.TabSize * 2) && read_all(f, h->blkp.data(), (size_t) wcp.TabSize * 4);
    if (ok && wc.BytBlk == 4) ok = wc.WordSz == 2 * wc.WordNo && read_all(f, h->blkb.data(), wc.WordNo * 4);
    else if (ok) {
        ok = wc.WordSz == wc.WordNo;
        std::vector<uint16_t> s(ok ? wc.WordSz : 0);
        ok = ok && read_all(f, s.data(), s.size() * 2);
        for (size_t i = 0; ok && i < s.size(); ++i) h->blkb[i] = s[i];
    }
    ok = ok && read_all(f, h->wscr.data(), (size_t) wcp.TabSize * 2) && read_all(f, h->convtab.data(), wc.ConvTS);
    fclose(f);
    if (!ok) return fail("short or inconsistent file");
    for (uint32_t w = 0; w < wcp.TabSize; ++w)
        if (h->blkp[w] < 0 || (h->blkp[w] && (uint64_t) h->blkp[w] - 1 + h->nblk[w] > wc.WordNo)) return fail("a posting list runs past the end");

    h->wcp = wcp; h->wc = wc; memcpy(h->b2c, b2c, sizeof b2c); h->chrid = chrid;
    std::string why;
    if (!derive_search_params(h, o, why)) return fail(why);
    return (SpdpBlkIndexHost*) h;
}

extern "C" const SpdpBlkIndexDesc* spdp_blk_index_host_desc(const SpdpBlkIndexHost* h) { return h ? &((const HostIndex*) h)->d : nullptr; }
extern "C" void spdp_blk_index_host_free(SpdpBlkIndexHost* h) { delete (HostIndex*) h; }

// ---- the index builder (include/spdp.h "the index builder"): host side -----------------------------------------------------
#include "spdp_internal.h"
#include "spdp_blk_build.h"
#include "spdp_hostcpus.h"
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>

namespace {
// DefBitPat (src/bitpat.cc:47-56): the pairs of spaced patterns `spaln -W -XC<n>` uses for a k-mer weight, leftmost position first
const char* const kDefBitPat[16] = {
    "", "1", "101", "1011,10011", "101011,1000111", "10100111,100101101", "1010011011,1010100111",
    "1001110111,100011011011", "100110110111,1010010111011", "1001110110111,10100101011111",
    "100111001101111,1010011010101111", "1000111101111011,1001110101111011", "101001101011111011,100011100011111111",
    "10101001110100111111,101100011011010011111", "100011110001111110111,1010110010101011111011",
    "101011001001011101101111,1001100001100111101111011"};
// bpcompress (src/bitpat.cc:214-226): '1' at string position i -> bit i; stops at the first character that is not 0 / 1
uint32_t bp_compress(const char* sp, int* ones)
{
    uint32_t c = 0;
    *ones = 0;
    for (uint32_t b = 1; *sp; b <<= 1, ++sp)
        if (*sp == '1') { c |= b; ++*ones; } else if (*sp != '0') break;
    return c;
}
double wall(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
}   // namespace

// setupbitpat(DNA, gnmsz) with no -X option but -XC (src/blksrc.cc:680-737; the defaults wcp_cf of :45)
extern "C" int spdp_blk_build_params_default(int64_t fasta_bytes, int32_t nbitpat, SpdpBlkBuildParams* p)
{
    if (!p || fasta_bytes < 1) return -1;
    memset(p, 0, sizeof *p);
    p->afact = 10;
    p->nbitpat = nbitpat > 1 ? nbitpat : 1;
    const double gs = (double) fasta_bytes;
    int blklen = (int) sqrt(gs);
    blklen = (int) (blklen / 1024 + 1) * 1024;
    if (blklen > 65536) blklen = 65536;
    p->blklen = blklen;
    int k = (int) (log(gs) * 0.59);
    if (k < 3) k = 3;
    if (k > 16) k = 16;
    p->ktuple = k;
    p->maxgene = (int) (38 * sqrt(gs) / 1024 + 1) * 1024;
    if (p->maxgene < 16384) p->maxgene = 16384;
    p->nshift = k;
    if (p->nbitpat > 1) {
        if (k >= 16) return -1;
        const char* sp = kDefBitPat[k];
        int w = 0;
        p->bitpat = bp_compress(sp, &w);
        if (w > p->ktuple) p->ktuple = w;
        if (p->nbitpat > 3) {
            const char* comma = strchr(sp, ',');
            if (comma) { int w2 = 0; p->bitpat2 = bp_compress(comma + 1, &w2); if (w2 > p->ktuple) p->ktuple = w2; }
            else p->nbitpat = 3;
        }
    } else p->bitpat = (1u << k) - 1;
    return 0;
}

static SpdpBlkIndexHost* blk_index_build(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParams* p,
                                         const SpdpBlkSearchOpts* opts, double* seconds);
extern "C" SpdpBlkIndexHost* spdp_blk_index_build(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParams* p,
                                                  const SpdpBlkSearchOpts* opts, double* seconds)
{
    try { return blk_index_build(ctx, genome, p, opts, seconds); }
    catch (const std::bad_alloc&) { if (ctx) ctx->err = "spdp_blk_index_build: out of host memory"; return nullptr; }      // (nothing of C++ crosses the C boundary)
}
static SpdpBlkIndexHost* blk_index_build(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParams* p,
                                         const SpdpBlkSearchOpts* opts, double* seconds)
{
    if (!ctx) return nullptr;
    auto fail = [&](const char* m) -> SpdpBlkIndexHost* { ctx->err = std::string("spdp_blk_index_build: ") + m; return nullptr; };
    if (!genome || !genome->codes || !genome->chr_off || genome->n_chr < 1 || !p) return fail("null argument");
    const int K = p->ktuple, nbit = p->nbitpat;
    if (K < 3 || K > 15 || (nbit != 1 && nbit != 3 && nbit != 5) || p->nshift < 1 || p->nshift > SPDP_BLK_MAX_SHIFT || p->blklen < 64 || p->blklen > 65536 ||
        p->afact < 1 || p->maxgene < p->blklen)
        return fail("parameters out of range (k 3 .. 15, 1 / 3 / 5 bit patterns, blklen 64 .. 65536)");
    const auto t_begin = std::chrono::steady_clock::now();
    const int n_chr = genome->n_chr;
    const int64_t G = genome->chr_off[n_chr] - genome->chr_off[0];
    if (genome->chr_off[0] != 0 || G < 1 || G > (int64_t) UINT32_MAX) return fail("chr_off must start at 0; at most 2^32 - 1 residues (the reference's positions are 32-bit)");
    const uint32_t tabsize = 1u << (2 * K);
    // the patterns as WordTab::WordTab lines them up (src/bitpat.cc:246-252)
    BlkBuildArgs A;
    memset(&A, 0, sizeof A);
    uint32_t pats[5]; int rev[5];
    if (nbit == 1) { pats[0] = p->bitpat; rev[0] = 0; }
    else { pats[0] = (1u << K) - 1; rev[0] = 0; for (int k = 1; k < nbit; ++k) { pats[k] = k < 3 ? p->bitpat : p->bitpat2; rev[k] = (k - 1) % 2; } }
    int max_width = 0;
    for (int k = 0; k < nbit; ++k) {
        int width = 0, weight = 0;
        for (uint32_t x = pats[k]; x; x >>= 1) { weight += x & 1; ++width; }
        if (weight != K || width > 32 || !(pats[k] & 1)) return fail("a bit pattern's weight is not k (or it is wider than 32)");
        A.width[k] = width; A.spaced[k] = width > weight;
        int wt = 0;
        for (int w = 0; w < width; ++w) if (pats[k] & (1u << (rev[k] ? width - 1 - w : w))) A.exam[k][wt++] = w;
        max_width = std::max(max_width, width);
    }
    A.G = G; A.nbit = nbit; A.nshift = p->nshift; A.blklen = p->blklen; A.margin = max_width - 1; A.threaded = p->threaded ? 1 : 0; A.weight = K;
    // blocks per chromosome (the walk of scan_genome: a block ends after margin + blklen residues, then after every blklen)
    const int64_t s_size = (int64_t) A.margin + A.blklen;
    std::vector<int32_t> chr_first(n_chr);
    std::unique_ptr<HostIndex> hold(new HostIndex);         // (freed on every way out but the last)
    HostIndex* h = hold.get();
    h->chrid.resize((size_t) n_chr + 1);
    uint64_t blocks = 0;
    for (int c = 0; c < n_chr; ++c) {
        const int64_t L = genome->chr_off[c + 1] - genome->chr_off[c];
        if (L < 0) { return fail("chr_off decreases"); }
        chr_first[c] = (int32_t) (blocks + 1);
        h->chrid[c] = {(uint32_t) genome->chr_off[c], (uint32_t) (blocks + 1)};
        blocks += L <= 0 ? 0 : (L < s_size ? 1 : 1 + (L - A.margin) / A.blklen);
    }
    h->chrid[n_chr] = {(uint32_t) G, (uint32_t) (blocks + 1)};
    if (blocks < 1 || blocks >= (1ull << 31)) { return fail("no block, or too many"); }
    int key_bits = 32 + 2 * K;
    std::vector<uint32_t> tcount, cnt;
    BlkBuildDev* dev = nullptr;
    if (spdp_blkidx_words(ctx, genome->codes, genome->chr_off, chr_first.data(), n_chr, A, tabsize, key_bits, tcount, cnt, &dev)) { return nullptr; }
    struct DevGuard { BlkBuildDev* d; ~DevGuard() { spdp_blkidx_free(d); } } dev_guard{dev};
    const double t_dev1 = wall(t_begin);
    // ---- blkscrtab(segn, blksz), src/blksrc.cc:944-997
    const auto t_host = std::chrono::steady_clock::now();
    const uint32_t segn = (uint32_t) blocks, blksz = (uint32_t) G / segn;
    try { h->nblk.assign(tabsize, 0); h->blkp.assign(tabsize, 0); h->wscr.assign(tabsize, 0); } catch (const std::bad_alloc&) { return fail("out of memory for the index tables"); }
    const int nt = std::max(1, std::min(spdp_host_cpus(), (int) (tabsize >> 14)));
    auto on_ranges = [&](auto f) {
        std::vector<std::thread> th;
        const uint32_t step = (tabsize + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { f(t, (uint32_t) std::min<uint64_t>(tabsize, (uint64_t) t * step), (uint32_t) std::min<uint64_t>(tabsize, (uint64_t) (t + 1) * step)); });
        for (std::thread& x : th) x.join();
    };
    std::vector<uint64_t> part_m(nt, 0);
    on_ranges([&](int t, uint32_t a, uint32_t b) { uint64_t m = 0; for (uint32_t w = a; w < b; ++w) if (tcount[w]) ++m; part_m[t] = m; });
    uint64_t m_seen = 0;
    for (uint64_t x : part_m) m_seen += x;
    if (!m_seen) { return fail("no word in the genome"); }
    const double basescr = log((double) segn);
    short min_scr = (short) -(100 * log((double) p->afact * blksz / (uint32_t) m_seen));
    if (min_scr < 0) min_scr = 0;
    std::vector<double> part_avr(nt, 0.);
    std::vector<uint64_t> part_kept(nt, 0), part_words(nt, 0), part_max(nt, 0), part_over(nt, 0);
    on_ranges([&](int t, uint32_t a, uint32_t b) {
        double avr = 0.; uint64_t kept = 0, words = 0, mx = 0, over = 0;
        for (uint32_t w = a; w < b; ++w) {
            if (!cnt[w]) { h->wscr[w] = -1; continue; }
            if (cnt[w] > 65535) ++over;
            // a word that never counted in tcount (all its occurrences in a block's last margin residues): the reference takes
            // (short) of +inf there -- cvttsd2si's 0x80000000, whose low half is 0
            const short sc = tcount[w] ? (short) (100 * (basescr - log((double) tcount[w] / nbit))) : (short) 0;
            if (sc > min_scr) { ++kept; words += cnt[w]; h->wscr[w] = sc; avr += sc; mx = std::max<uint64_t>(mx, cnt[w]); }
            else { h->wscr[w] = min_scr; cnt[w] = 0; }
        }
        part_avr[t] = avr; part_kept[t] = kept; part_words[t] = words; part_max[t] = mx; part_over[t] = over;
    });
    double avr = 0.; uint64_t kept = 0, word_no = 0, max_blk = 0, over = 0;
    for (int t = 0; t < nt; ++t) { avr += part_avr[t]; kept += part_kept[t]; word_no += part_words[t]; max_blk = std::max(max_blk, part_max[t]); over += part_over[t]; }
    if (over) { return fail("a word lies in more than 65 535 blocks: the reference's 16-bit counters wrap there (use a longer k)"); }
    if (word_no > (uint64_t) INT32_MAX) { return fail("more postings than a 32-bit list offset (blkp) can address"); }
    uint64_t at = 0;
    for (uint32_t w = 0; w < tabsize; ++w) if (cnt[w]) { h->blkp[w] = (int32_t) (at + 1); h->nblk[w] = (uint16_t) cnt[w]; at += cnt[w]; }
    try { h->blkb.assign((size_t) word_no, 0); } catch (const std::bad_alloc&) { return fail("out of memory for the posting lists"); }
    const double t_host1 = wall(t_host);
    const auto t_dev2 = std::chrono::steady_clock::now();
    if (spdp_blkidx_lists(ctx, dev, h->blkp.data(), (int64_t) word_no, h->blkb.data())) { return nullptr; }
    const double t_dev2s = wall(t_dev2);
    // ---- the header (MakeBlk::idxblk, WriteBlkInfo, findChrBbound)
    memset(&h->wcp, 0, sizeof h->wcp); memset(&h->wc, 0, sizeof h->wc);
    h->wcp.Nalpha = 4; h->wcp.Ktuple = (uint32_t) K; h->wcp.Bitpat2 = p->bitpat2; h->wcp.TabSize = tabsize; h->wcp.BitPat = p->bitpat;
    h->wcp.Nshift = (uint32_t) p->nshift; h->wcp.blklen = (uint32_t) p->blklen; h->wcp.MaxGene = (uint32_t) p->maxgene;
    h->wcp.Nbitpat = (int16_t) nbit; h->wcp.afact = (uint16_t) p->afact;
    h->convtab.assign(17, 4);                           // iConvTab of "A|C|G|T|N" over the nucleotide codes (src/bitpat.cc:58-87)
    h->convtab[0] = h->convtab[1] = 255; h->convtab[2] = 0; h->convtab[3] = 1; h->convtab[5] = 2; h->convtab[9] = 3;
    h->wc.ConvTS = 17; h->wc.WordNo = word_no; h->wc.ChrNo = (uint64_t) n_chr; h->wc.glen = (uint64_t) G;
    h->wc.AvrScr = kept ? (uint16_t) (avr / (double) kept) : 0; h->wc.MaxBlk = (uint16_t) max_blk;
    h->wc.BytBlk = segn <= 65535 ? 2 : 4; h->wc.WordSz = h->wc.BytBlk == 2 ? word_no : 2 * word_no; h->wc.VerNo = 26;
    const double B = (double) segn;
    h->b2c[0] = h->b2c[1] = h->b2c[2] = 0.;
    for (int k = 0; k <= n_chr; ++k) {
        const double off = k * B - n_chr * (double) (h->chrid[k].segn - 1);
        h->b2c[0] = std::min(h->b2c[0], off); h->b2c[1] = std::max(h->b2c[1], off);
    }
    h->b2c[0] /= B; h->b2c[1] /= B;
    SpdpBlkSearchOpts o;
    if (opts) o = *opts; else spdp_blk_search_opts_default(&o);
    std::string why;
    if (!derive_search_params(h, o, why)) { ctx->err = "spdp_blk_index_build: " + why; return nullptr; }
    if (seconds) { seconds[0] = t_dev1 + t_dev2s; seconds[1] = t_host1; seconds[2] = wall(t_begin); }
    return (SpdpBlkIndexHost*) hold.release();
}

// ---- the translated index, `spaln -W -KP` (<db>.bkp) --------------------------------------------------------------------------
// what setupbitpat picks for a translated genome (src/blksrc.cc:680-737: wcp_af; k from 0.36 ln(size), at most 6)
extern "C" int spdp_blk_build_params_default_p(int64_t fasta_bytes, SpdpBlkBuildParamsP* p)
{
    if (!p || fasta_bytes < 1) return -1;
    const double keep_acomp[20] = {p->acomp[0], p->acomp[1], p->acomp[2], p->acomp[3], p->acomp[4], p->acomp[5], p->acomp[6], p->acomp[7], p->acomp[8], p->acomp[9],
                                   p->acomp[10], p->acomp[11], p->acomp[12], p->acomp[13], p->acomp[14], p->acomp[15], p->acomp[16], p->acomp[17], p->acomp[18], p->acomp[19]};
    memset(p, 0, sizeof *p);
    memcpy(p->acomp, keep_acomp, sizeof keep_acomp);    // (the composition terms are the caller's: see include/spdp.h)
    p->b.afact = 10; p->b.nbitpat = 1;
    const double gs = (double) fasta_bytes;
    int blklen = (int) sqrt(gs);
    blklen = (int) (blklen / 1024 + 1) * 1024;
    if (blklen > 65536) blklen = 65536;
    p->b.blklen = blklen;
    int k = (int) (log(gs) * 0.36);
    if (k < 3) k = 3;
    if (k > 6) k = 6;
    p->b.ktuple = k; p->b.nshift = k; p->b.bitpat = (1u << k) - 1;
    p->b.maxgene = (int) (38 * sqrt(gs) / 1024 + 1) * 1024;
    if (p->b.maxgene < 16384) p->b.maxgene = 16384;
    p->nalpha = 20; p->minorf = 30; p->aaafact = 1.;
    // iConvTab of the twenty-letter alphabet over the tron codes (ReducWord::ReducWord, src/bitpat.cc:58-87): A .. V = 3 .. 22 in
    // the order of the reference's amino-acid codes, the AGY serines (23) with the serines, Sec (24) one past the ambiguous class
    p->convts = 27;
    for (int c = 0; c < 27; ++c) p->convtab[c] = 20;
    for (int c = 3; c <= 22; ++c) p->convtab[c] = (uint8_t) (c - 3);
    p->convtab[23] = 15; p->convtab[24] = 21; p->convtab[25] = 0; p->convtab[26] = 20;
    return 0;
}

static SpdpBlkIndexHost* blk_index_build_p(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParamsP* p,
                                           const SpdpBlkSearchOpts* opts, double* seconds)
{
    if (!ctx) return nullptr;
    auto fail = [&](const char* m) -> SpdpBlkIndexHost* { ctx->err = std::string("spdp_blk_index_build_p: ") + m; return nullptr; };
    if (!genome || !genome->codes || !genome->chr_off || genome->n_chr < 1 || !p) return fail("null argument");
    const int K = p->b.ktuple, na = p->nalpha, wq = p->minorf, nshift = p->b.nshift;
    if (K < 3 || K > 7 || p->b.nbitpat != 1 || p->b.bitpat != (1u << K) - 1 || nshift < 1 || nshift > SPDP_BLK_MAX_SHIFT || na < 6 || na > 20 ||
        wq < 3 || wq > 120 || p->b.afact < 1 || p->b.maxgene < p->b.blklen || p->convts < 24 || p->convts > 32 || !(p->aaafact > 0))
        return fail("parameters out of range (contiguous words of 3 .. 7 amino acids, 6 .. 20 classes, MinOrf 3 .. 120)");
    uint64_t tab64 = 1;
    for (int i = 0; i < K; ++i) tab64 *= (uint64_t) na;
    if (tab64 > (1ull << 30)) return fail("Nalpha ^ k beyond 2^30 words");
    const uint32_t tabsize = (uint32_t) tab64;
    const int margin = 3 * K - 1 + wq;                      // prelude + MinOrf, src/blksrc.cc:440-445
    if (p->b.blklen <= margin || p->b.blklen > 65536) return fail("blklen must exceed 3 k - 1 + MinOrf (and be at most 65536)");
    // the reference strikes the words of a short frame from its ring by stepping back from the frame's end; parameters with which a
    // step would pass the ring's length (it would strike younger words there) are refused
    for (int s = K; 3 * s < wq; ++s) {
        const int nw = s - K, sp = nw % nshift, turns = (nw + sp) / nshift + 1;
        if (3 * (sp + 1) + 3 * nshift * (turns - 1) > wq) return fail("k, Nshift and MinOrf for which the reference's delay ring wraps onto itself");
    }
    const auto t_begin = std::chrono::steady_clock::now();
    const int n_chr = genome->n_chr;
    const int64_t G = genome->chr_off[n_chr] - genome->chr_off[0];
    if (genome->chr_off[0] != 0 || G < 1 || G > (int64_t) UINT32_MAX) return fail("chr_off must start at 0; at most 2^32 - 1 residues (the reference's positions are 32-bit)");
    BlkBuildArgsP A;
    memset(&A, 0, sizeof A);
    A.G = G; A.K = K; A.nalpha = na; A.nshift = nshift; A.blklen = p->b.blklen; A.margin = margin; A.minorf = wq; A.threaded = p->b.threaded ? 1 : 0;
    A.tabsize = tabsize;
    {   // g2r (src/bitpat.cc:88-106): codon -> tron code -> class
        uint8_t mid[32], tron_of[64];
        spdp_genetic_code_tables(mid, tron_of);
        for (int g = 0; g < 64; ++g) {
            const int t = tron_of[g];
            A.codon_class[g] = (t >= 3 && t <= 23 && t < p->convts) ? p->convtab[t] : (uint8_t) 255;    // TGA (24) is the class "U" = Nalpha, TAA / TAG none
            if (A.codon_class[g] >= na) A.codon_class[g] = 255;
        }
    }
    const int64_t s_size = (int64_t) margin + A.blklen;
    std::vector<int32_t> chr_first(n_chr);
    std::unique_ptr<HostIndex> hold(new HostIndex);
    HostIndex* h = hold.get();
    h->chrid.resize((size_t) n_chr + 1);
    uint64_t blocks = 0;
    for (int c = 0; c < n_chr; ++c) {
        const int64_t L = genome->chr_off[c + 1] - genome->chr_off[c];
        if (L < 0) { return fail("chr_off decreases"); }
        chr_first[c] = (int32_t) (blocks + 1);
        h->chrid[c] = {(uint32_t) genome->chr_off[c], (uint32_t) (blocks + 1)};
        if (L <= 0) continue;
        const int64_t nb = L < s_size ? 1 : 1 + (L - margin) / A.blklen;
        // the last block: when it is longer than blklen an empty block follows it (scan_genome / harvest close it twice)
        const int64_t b = nb - 1;
        const int64_t lo = A.threaded ? b * A.blklen : (b ? b * A.blklen + margin : 0);
        const int64_t n = ((!A.threaded && b) ? margin : 0) + (L - lo);
        blocks += (uint64_t) nb + (n > A.blklen ? 1 : 0);
    }
    h->chrid[n_chr] = {(uint32_t) G, (uint32_t) (blocks + 1)};
    if (blocks < 1 || blocks >= (1ull << 31)) { return fail("no block, or too many"); }
    int word_bits = 1;
    while ((1ull << word_bits) < tab64) ++word_bits;
    std::vector<uint32_t> tcount, cnt;
    BlkBuildDev* dev = nullptr;
    if (spdp_blkidx_words_p(ctx, genome->codes, genome->chr_off, chr_first.data(), n_chr, A, 32 + word_bits, tcount, cnt, &dev)) { return nullptr; }
    struct DevGuard { BlkBuildDev* d; ~DevGuard() { spdp_blkidx_free(d); } } dev_guard{dev};
    const double t_dev1 = wall(t_begin);
    // ---- blkscrtab(segn), src/blksrc.cc:879-942
    const auto t_host = std::chrono::steady_clock::now();
    const uint32_t segn = (uint32_t) blocks;
    try { h->nblk.assign(tabsize, 0); h->blkp.assign(tabsize, 0); h->wscr.assign(tabsize, 0); } catch (const std::bad_alloc&) { return fail("out of memory for the index tables"); }
    const double basescr = log((double) segn);
    const double deltaa = p->acomp[0] - p->acomp[na - 1];
    // the composition term of every word as the reference's running sum leaves it (a double sum in table order: one thread, nothing but
    // the sum), then the scores -- logarithms -- on all host threads; their total is a sum of integers, exact in any order
    std::vector<int16_t> alc_of;
    try { alc_of.resize(tabsize); } catch (const std::bad_alloc&) { return fail("out of memory for the index tables"); }
    {
        double alc = K * p->acomp[0];
        double step[21];                                // what the sum gains when a digit goes from q - 1 to q
        for (int q = 1; q <= na; ++q) step[q] = p->acomp[q % na] - p->acomp[q - 1];     // (q = na is never taken: the wrap has its own rule below)
        for (uint32_t w0 = 0; w0 < tabsize; w0 += (uint32_t) na) {
            // the words w0 .. w0 + na - 1 differ in their last digit only: na - 1 plain steps, then the carry
            for (int d = 0; d < na - 1; ++d) { alc_of[w0 + d] = (short) alc; alc += step[d + 1]; }
            const uint32_t w = w0 + (uint32_t) na - 1;
            alc_of[w] = (short) alc;
            int z = 0, q = 0;
            for (uint32_t x = w + 1; (q = (int) (x % (uint32_t) na)) == 0; x /= (uint32_t) na) ++z;
            alc += z * deltaa;
            alc += p->acomp[q] - p->acomp[q - 1];
        }
    }
    const int nt = std::max(1, std::min(spdp_host_cpus(), (int) (tabsize >> 14)));
    auto on_ranges = [&](auto f) {
        std::vector<std::thread> th;
        const uint32_t step = (tabsize + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { f(t, (uint32_t) std::min<uint64_t>(tabsize, (uint64_t) t * step), (uint32_t) std::min<uint64_t>(tabsize, (uint64_t) (t + 1) * step)); });
        for (std::thread& x : th) x.join();
    };
    std::vector<int64_t> part_sum(nt, 0);
    std::vector<uint64_t> part_m(nt, 0);
    on_ranges([&](int t, uint32_t lo, uint32_t hi) {
        int64_t sum = 0; uint64_t m = 0;
        for (uint32_t w = lo; w < hi; ++w) {
            if (!tcount[w]) continue;
            ++m;
            short sc = (short) (100 * (basescr - log((double) tcount[w] / 1)));
            sc = (short) (sc + alc_of[w]);
            h->wscr[w] = sc;
            sum += sc;
        }
        part_sum[t] = sum; part_m[t] = m;
    });
    uint64_t m_seen = 0; int64_t total = 0;
    for (int t = 0; t < nt; ++t) { m_seen += part_m[t]; total += part_sum[t]; }
    if (!m_seen) { return fail("no word in the genome"); }
    double avr = (double) total;
    avr /= (double) m_seen;
    short min_scr = (short) (avr - 100 * (1 + p->aaafact) * log((double) p->b.afact));
    if (min_scr < 0) min_scr = 0;
    std::vector<uint64_t> part_words(nt, 0), part_over(nt, 0);
    on_ranges([&](int t, uint32_t lo, uint32_t hi) {
        uint64_t words = 0, ov = 0;
        for (uint32_t w = lo; w < hi; ++w) {
            if (!cnt[w]) { h->wscr[w] = -1; continue; }
            if (cnt[w] > 65535) ++ov;
            if (h->wscr[w] > min_scr) words += cnt[w];
            else { h->wscr[w] = 0; cnt[w] = 0; }
        }
        part_words[t] = words; part_over[t] = ov;
    });
    uint64_t word_no = 0, over = 0;
    for (int t = 0; t < nt; ++t) { word_no += part_words[t]; over += part_over[t]; }
    { std::vector<int16_t>().swap(alc_of); }
    if (over) { return fail("a word lies in more than 65 535 blocks: the reference's 16-bit counters wrap there (use a longer k)"); }
    if (word_no > (uint64_t) INT32_MAX) { return fail("more postings than a 32-bit list offset (blkp) can address"); }
    uint64_t at = 0;
    for (uint32_t w = 0; w < tabsize; ++w) if (cnt[w]) { h->blkp[w] = (int32_t) (at + 1); h->nblk[w] = (uint16_t) cnt[w]; at += cnt[w]; }
    try { h->blkb.assign((size_t) word_no, 0); } catch (const std::bad_alloc&) { return fail("out of memory for the posting lists"); }
    const double t_host1 = wall(t_host);
    const auto t_dev2 = std::chrono::steady_clock::now();
    if (spdp_blkidx_lists(ctx, dev, h->blkp.data(), (int64_t) word_no, h->blkb.data())) { return nullptr; }
    const double t_dev2s = wall(t_dev2);
    memset(&h->wcp, 0, sizeof h->wcp); memset(&h->wc, 0, sizeof h->wc);
    h->wcp.Nalpha = (uint32_t) na; h->wcp.Ktuple = (uint32_t) K; h->wcp.Bitpat2 = p->b.bitpat2; h->wcp.TabSize = tabsize; h->wcp.BitPat = p->b.bitpat;
    h->wcp.Nshift = (uint32_t) nshift; h->wcp.blklen = (uint32_t) p->b.blklen; h->wcp.MaxGene = (uint32_t) p->b.maxgene;
    h->wcp.Nbitpat = 1; h->wcp.afact = (uint16_t) p->b.afact;
    h->convtab.assign(p->convtab, p->convtab + p->convts);
    h->wc.ConvTS = (uint32_t) p->convts; h->wc.WordNo = word_no; h->wc.ChrNo = (uint64_t) n_chr; h->wc.glen = (uint64_t) G;
    h->wc.AvrScr = (uint16_t) avr;
    // ContBlk::MaxBlk is never set on this path of the reference (blkscrtab(segn) only raises it, from whatever the heap held): its
    // files carry 65535, and the search sizes its run table from it (src/blksrc.cc:2983) -- so does this one
    h->wc.MaxBlk = 65535;
    h->wc.BytBlk = segn <= 65535 ? 2 : 4; h->wc.WordSz = h->wc.BytBlk == 2 ? word_no : 2 * word_no; h->wc.VerNo = 26;
    const double B = (double) segn;
    h->b2c[0] = h->b2c[1] = h->b2c[2] = 0.;
    for (int k = 0; k <= n_chr; ++k) {
        const double off = k * B - n_chr * (double) (h->chrid[k].segn - 1);
        h->b2c[0] = std::min(h->b2c[0], off); h->b2c[1] = std::max(h->b2c[1], off);
    }
    h->b2c[0] /= B; h->b2c[1] /= B;
    SpdpBlkSearchOpts o;
    if (opts) o = *opts; else spdp_blk_search_opts_default(&o);
    std::string why;
    if (!derive_search_params(h, o, why)) { ctx->err = "spdp_blk_index_build_p: " + why; return nullptr; }
    if (seconds) { seconds[0] = t_dev1 + t_dev2s; seconds[1] = t_host1; seconds[2] = wall(t_begin); }
    return (SpdpBlkIndexHost*) hold.release();
}
extern "C" SpdpBlkIndexHost* spdp_blk_index_build_p(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParamsP* p,
                                                    const SpdpBlkSearchOpts* opts, double* seconds)
{
    try { return blk_index_build_p(ctx, genome, p, opts, seconds); }
    catch (const std::bad_alloc&) { if (ctx) ctx->err = "spdp_blk_index_build_p: out of host memory"; return nullptr; }
}

// WriteBlkInfo / writeBlkInfo (src/blksrc.cc:598-622): the struct images of the reference's 64-bit build
extern "C" int spdp_blk_index_write(const SpdpBlkIndexHost* hh, const char* path)
{
    const HostIndex* h = (const HostIndex*) hh;
    if (!h || !path) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    FileContBlk wc = h->wc;
    wc.p_Nblk = wc.p_blkp = wc.p_blkb = wc.p_wscr = wc.p_ChrID = 0;
    bool ok = fwrite(&h->wcp, sizeof h->wcp, 1, f) == 1 && fwrite(&wc, sizeof wc, 1, f) == 1 && fwrite(h->b2c, sizeof h->b2c, 1, f) == 1 &&
              fwrite(h->chrid.data(), sizeof(FileChromo), h->chrid.size(), f) == h->chrid.size() &&
              fwrite(h->nblk.data(), 2, h->nblk.size(), f) == h->nblk.size() && fwrite(h->blkp.data(), 4, h->blkp.size(), f) == h->blkp.size();
    if (ok && wc.BytBlk == 2) {
        std::vector<uint16_t> s(h->blkb.size());
        for (size_t i = 0; i < s.size(); ++i) s[i] = (uint16_t) h->blkb[i];
        ok = fwrite(s.data(), 2, s.size(), f) == s.size();
    } else if (ok) ok = fwrite(h->blkb.data(), 4, h->blkb.size(), f) == h->blkb.size();
    ok = ok && fwrite(h->wscr.data(), 2, h->wscr.size(), f) == h->wscr.size() && fwrite(h->convtab.data(), 1, h->convtab.size(), f) == h->convtab.size();
    return (fclose(f) == 0 && ok) ? 0 : -1;
}
