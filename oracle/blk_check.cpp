// blk_check.cpp -- TEST INFRASTRUCTURE ONLY: the product's host-side block-search logic (spaln_amd/csrc/spdp_loci.h: TestOutput's second
// half and FindHsp as a machine that is advanced with search answers; spdp_hsp_host.h / spdp_hsp_chain.h: the HSP search in its host
// form) built with the host compiler, so that the tests without a GPU can hold it against the reference's recorded runs
// (tests/test_blk_find.py).  The vote itself is device code now; the checker is handed the vote's state by the caller (the oracle's
// vote, oracle/spdp_oracle_blk.c).  Nothing in the product links this.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../spaln_amd/csrc/spdp_loci.h"
#include "../spaln_amd/csrc/spdp_hsp_host.h"
#include "../spaln_amd/csrc/spdp_region.h"

extern "C" {
// One TestOutput call of one query: pairs (9 ints each: bscr chr lb rb ub db zl zr rvs), mmct[4], the run scores near the pairs
// ((block | direction << 28, score) x n_runs), forced (the call is TestOutput(1)), critjscr in / out.  The log has the recorder's layout
// (oracle/ref_build/blk_tap.cc, snap_find): -4, 0, call, critjscr, n_pairs, pairs x 10, n_loci, per locus {chr, sens, base, len, left,
// right, jscr, CdsNo, 0} + (CdsNo + 1) x 5.  prm: find_prm of the fixture; chr_tab: the index's chromosome table.  Returns the ints
// written; *verdict: > 0 loci, 0 go on voting, -1 the search ends
int loci_check_call(const uint8_t* genome, const int64_t* chr_off, int n_chr, const int32_t* chr_tab, const int32_t* rscrtab, float rbscoef,
                    float rbscons, int gdb, const uint8_t* q, int q_len, int left, int right, const int32_t* prm, const int16_t* intpen,
                    int intpen_len, const SpdpWilipModel* model, const int32_t* pairs9, int n_pairs, const int32_t* mmct,
                    const int32_t* runs, int n_runs, int forced, int call, int32_t* critjscr, int32_t* verdict, int32_t* log, int cap)
{
    spdp_loci::Params P;
    P.vthr = prm[0]; memcpy(&P.drop_rate, &prm[1], 4); P.max_out = prm[4]; P.max_out2 = prm[5]; P.bbt = prm[6]; P.min_agap = prm[7];
    P.blklen = prm[8]; P.ext_block = prm[9]; P.ext_block_l = prm[10]; P.phase1t = prm[11]; P.a_exgl = prm[20]; P.a_exgr = prm[21];
    P.dvsp = prm[12]; P.no_retry = prm[2];
    const spdp_loci::Chromosomes G = {chr_off, n_chr, chr_tab};
    spdp_loci::Call c;
    c.P = &P; c.G = &G; c.rnd = {rscrtab, rbscoef, rbscons, gdb};
    c.q = {q_len, left, right};
    for (int i = 0; i < n_pairs; ++i) {
        const int32_t* r = pairs9 + 9 * i;
        c.pairs.push_back({r[0], r[1], 0, (uint32_t) r[2], (uint32_t) r[3], (uint32_t) r[4], (uint32_t) r[5], (uint32_t) r[6], (uint32_t) r[7], r[8]});
    }
    memcpy(c.mmct, mmct, sizeof c.mmct);
    c.forced = forced != 0;
    for (int i = 0; i < n_runs; ++i) c.runs.emplace_back((uint32_t) runs[2 * i], runs[2 * i + 1]);
    c.critjscr = *critjscr;
    c.begin();
    const spdp_hsp::GapCosts gc = {intpen, intpen_len, prm[13], prm[14], prm[15], prm[16], prm[17]};
    int pi; spdp_loci::Region r;
    std::vector<uint8_t> codes;
    std::vector<spdp_hsp::Unit> units;
    while (c.needs(pi, r)) {                            // every search answered on the spot, by the host's form
        spdp_region::materialize(genome, chr_off, r.chr, r.base, r.len, r.rvs != 0, P.bbt == 3, codes);
        const spdp_hsp::Seqs s = {q, q_len, left, right, P.a_exgl, P.a_exgr, codes.data(), r.len, 0, r.len, P.bbt == 3 ? 3 : 1, nullptr, nullptr, nullptr};
        spdp_hsp::search(model, s, gc, -1, units);
        c.take(units);
    }
    *critjscr = c.critjscr; *verdict = c.result;
    int n = 0;
#define PUT(x) do { if (n < cap) log[n] = (x); ++n; } while (0)
    PUT(-4); PUT(0); PUT(call); PUT(c.critjscr); PUT(n_pairs);
    for (const spdp_loci::Pair& b : c.pairs) {
        PUT(b.bscr); PUT(b.chr); PUT((int) b.lb); PUT((int) b.rb); PUT((int) b.ub); PUT((int) b.db); PUT((int) b.zl); PUT((int) b.zr);
        PUT(b.rvs); PUT(b.jscr);
    }
    const int nl = c.result > 0 ? c.result : 0;
    PUT(nl);
    for (int k = 0; k < nl; ++k) {
        const spdp_loci::Locus& g = c.loci[k];
        PUT(g.at.chr); PUT(g.at.rvs ? 3 : 0); PUT(g.at.base); PUT(g.at.len); PUT(g.left); PUT(g.right); PUT(g.jscr); PUT((int) g.hsp.size() - 1); PUT(0);
        for (const spdp_hsp::Hsp& t : g.hsp) { PUT(t.jx); PUT(t.jy); PUT(t.jlen); PUT(t.nid); PUT(t.jscr); }
    }
#undef PUT
    return n;
}

// the product's translation of a region into tron codes (spdp_region.h), in place: held against the tron codes the reference itself
// made of the protein fixtures' windows (tests/test_blk_find.py)
void blk_check_nuc2tron(uint8_t* codes, int len) { spdp_region::to_tron(codes, len); }
}
