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
#include "spdp_blk_core.h"
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
    if (wcp.Nalpha != 4) { fclose(f); return fail("not a nucleotide index"); }
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

    SpdpBlkIndexDesc& d = h->d;
    memset(&d, 0, sizeof d);
    d.nalpha = (int32_t) wcp.Nalpha; d.tabsize = (int32_t) wcp.TabSize; d.nshift = (int32_t) wcp.Nshift; d.nbitpat = wcp.Nbitpat;
    d.convts = (int32_t) wc.ConvTS; d.n_chr = (int32_t) wc.ChrNo; d.maxblk = wc.MaxBlk;
    d.kk = wcp.Nbitpat / 2 + 1; d.drna = 1;
    if (d.kk < 1 || d.kk > 3) return fail("number of bit patterns out of range");
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
    if (d.nseg < 2 || d.maxmmc < 1) return fail("derived parameters out of range");
    return (SpdpBlkIndexHost*) h;
}

extern "C" const SpdpBlkIndexDesc* spdp_blk_index_host_desc(const SpdpBlkIndexHost* h) { return h ? &((const HostIndex*) h)->d : nullptr; }
extern "C" void spdp_blk_index_host_free(SpdpBlkIndexHost* h) { delete (HostIndex*) h; }
