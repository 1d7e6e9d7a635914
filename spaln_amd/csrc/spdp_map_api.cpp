// spdp_map_api.cpp -- spdp_map_align_s: block search -> regions + splice signals -> seeded alignment -> rescoring -> the
// locus that stays, for a batch of nucleotide queries in one call (include/spdp.h "map and align").  What the reference's
// per-query driver does around alignS_ng when the genome is searched (src/spaln.cc:880-1010: blkaln / spalign2, genomicseq at
// :913 reading the region and building its Exinon), batched: all loci of a chunk of queries share one signal launch, one
// seeded call and one rescoring call.  Host code only; the device work is that of the entries it calls.
#include "spdp_internal.h"
#include "spdp_h_internal.h"
#include "spdp_region.h"
#include "spdp_hostcpus.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

int spdh_signals_run(SpdpContext* ctx, const SpdpSignalModelH* m, const std::vector<SigJobH>& jobs, SignalArgsH args, int pack);   // spdp_signals_api.cpp

namespace {

struct DevMem {                      // scoped device allocation
    void* p = nullptr;
    ~DevMem() { if (p) (void) hipFree(p); }
    hipError_t get(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T> T* as() const { return (T*) p; }
};

inline uint8_t other_strand(uint8_t c)   // A 2 <-> T 9, C 3 <-> G 5; ambiguity codes stay as the block search's cut leaves them
{
    switch (c) { case 2: return 9; case 9: return 2; case 3: return 5; case 5: return 3; default: return c; }
}

double since(std::chrono::steady_clock::time_point t)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
}

template <class F> void on_host_threads(int n, F f)
{
    std::atomic<int> next{0};
    auto work = [&] { for (int k; (k = next++) < n; ) f(k); };
    const int nt = std::max(1, std::min(spdp_host_cpus(), n));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

}  // namespace

static int map_align_s(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                                const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpSignalModel* sigmodel,
                                const SpdpBlkFindParams* fprm, const SpdpRescoreParams* rp,
                                const uint8_t* codes, const int64_t* offs, int32_t n, int32_t ori,
                                SpdpMapGene* genes, SpdpMapExon** exons, double* seconds)
{
    if (!ctx) return -1;
    if (!ix || !hix || !genome || !sc || !sp || !sigmodel || !fprm || !rp || !codes || !offs || !genes || !exons) {
        ctx->err = "spdp_map_align_s: null argument"; return -1;
    }
    if (ori != 1 && ori != 3) { ctx->err = "spdp_map_align_s: ori must be 1 (the query as given) or 3 (both orientations)"; return -1; }
    if (!sp->wilip) { ctx->err = "spdp_map_align_s: SpdpSeedParams.wilip missing (the HSP searches of this call are the library's own)"; return -1; }
    *exons = nullptr;
    double sec[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) { genes[i].chr = -1; genes[i].rvs = 0; genes[i].q_rev = 0; genes[i].score = SPDP_NEVSEL; genes[i].val = 0; genes[i].n_loci = 0; genes[i].n_exons = 0; genes[i].exon_off = 0; }
    if (n <= 0) return 0;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<int32_t> ql(n, 0), qr(n);
    for (int i = 0; i < n; ++i) qr[i] = (int32_t) (offs[i + 1] - offs[i]);
    SpdpLocus* loci = nullptr; SpdpJuxt* hsps = nullptr; int32_t n_loci = 0;
    if (spdp_blk_find(ctx, ix, hix, genome, sp->wilip, sc, fprm, codes, offs, ql.data(), qr.data(), n, &loci, &n_loci, &hsps, nullptr)) return -1;
    struct Owned { SpdpLocus* l; SpdpJuxt* h; ~Owned() { free(l); free(h); } } owned{loci, hsps};
    sec[0] = since(t0);

    const bool both = ori == 3;
    std::vector<uint8_t> codes_rc;                      // comrev() of every query (ori = 3)
    if (both) {
        codes_rc.resize((size_t) offs[n]);
        on_host_threads(n, [&](int q) {
            const int64_t a0 = offs[q], len = offs[q + 1] - offs[q];
            for (int64_t i = 0; i < len; ++i) codes_rc[a0 + i] = other_strand(codes[a0 + len - 1 - i]);
        });
    }
    SpdpSignalModel sigm = *sigmodel;
    SpdpSeedParams spx = *sp;
    if (both) sigm.both_ori = spx.both_ori = 1;         // Exinon(seq, pwd, ori == 3), src/spaln.cc:1143, 1150
    // positions (NOT bytes) of the loci one chunk may hold; the signal arrays of a chunk take 8 B per position of pinned host memory
    // and as much on the device, allocated as the chunk needs them (the default bound is 16 GiB each; a C4-sized batch of 125 000
    // ESTs x 2 strands x ~50 kb of loci is what reaches it).  SPDP_MAP_CHUNK_MPOS = n: n x 2^20 positions (SPDP_MAP_CHUNK_MB: its old name)
    size_t chunk_positions = (size_t) 2048 << 20;
    for (const char* v : {"SPDP_MAP_CHUNK_MB", "SPDP_MAP_CHUNK_MPOS"})
        if (const char* e = getenv(v)) chunk_positions = (size_t) std::max(1, atoi(e)) << 20;
    std::vector<std::vector<SpdpMapExon>> kept(n);
    int partial = 0;
    (void) hipSetDevice(ctx->device);
    // chunks of loci: as few as the position limit allows, of equal size (a call's time is a chain of request latencies, not
    // device work: DESIGN.md 6g -- so the larger a chunk the better, and a short last chunk costs as much as a full one)
    int64_t all_positions = 0;
    for (int k = 0; k < n_loci; ++k) all_positions += (both ? 2 : 1) * (int64_t) (loci[k].len + 1);
    const int64_t n_chunks = std::max<int64_t>(1, (all_positions + (int64_t) chunk_positions - 1) / (int64_t) chunk_positions);
    const int64_t per_chunk = (all_positions + n_chunks - 1) / n_chunks;
    for (int c0 = 0; c0 < n_loci; ) {
        t0 = std::chrono::steady_clock::now();
        std::vector<int64_t> at;                        // first position of locus c0 + k in the chunk's arrays (len + 1 positions each)
        int64_t tot = 0;
        int c1 = c0;
        while (c1 < n_loci && (c1 == c0 || tot + (both ? 2 : 1) * (int64_t) (loci[c1].len + 1) <= per_chunk + (1 << 17))) {
            at.push_back(tot); tot += (both ? 2 : 1) * (int64_t) (loci[c1].len + 1); ++c1;
        }
        const int m = c1 - c0;
        const int ns = both ? 2 * m : m;                // slots of the chunk's arrays: locus k as the block search gave it, then (ori = 3)
        if (both) { at.resize(ns); for (int k = 0; k < m; ++k) at[m + k] = at[k] + loci[c0 + k].len + 1; }      // its other strand right behind it
        auto slot_left = [&](int j) { const SpdpLocus& L = loci[c0 + (j < m ? j : j - m)]; return j < m ? L.left : L.len - L.right; };
        auto slot_right = [&](int j) { const SpdpLocus& L = loci[c0 + (j < m ? j : j - m)]; return j < m ? L.right : L.len - L.left; };
        for (int k = 0; k < m; ++k) {
            const SpdpLocus& L = loci[c0 + k];
            if (L.chr < 0 || L.chr >= genome->n_chr || L.base < 0 || L.len < 0 || L.left < 0 || L.right > L.len || L.right < L.left ||
                genome->chr_off[L.chr] + L.base + L.len > genome->chr_off[L.chr + 1]) { ctx->err = "spdp_map_align_s: a locus outside its chromosome"; return -1; }
        }
        // ---- the regions as the aligner reads them, then their signals in one launch.  One pinned block of the context (it
        // stays for the next call) and one device block of the same layout: codes | sig5 | sig3 | cano5 | cano3 | dinc
        const int64_t T = (tot + 255) / 256 * 256;
        uint8_t* H = (uint8_t*) ctx->staging(2, (size_t) T * 8);
        if (!H) { ctx->err = "spdp_map_align_s: no pinned host memory for a chunk's regions and signals (SPDP_MAP_CHUNK_MB sets the chunk size)"; return -1; }
        uint8_t* reg = H; int16_t* sig5 = (int16_t*) (H + T); int16_t* sig3 = (int16_t*) (H + 3 * T);
        uint8_t* cano5 = H + 5 * T; uint8_t* cano3 = H + 6 * T; uint8_t* dinc = H + 7 * T;
        on_host_threads(ns, [&](int j) {
            const SpdpLocus& L = loci[c0 + (j < m ? j : j - m)];
            const uint8_t* src = genome->codes + genome->chr_off[L.chr] + L.base;
            uint8_t* dst = reg + at[j];
            if ((L.rvs != 0) == (j < m)) for (int i = 0; i < L.len; ++i) dst[i] = other_strand(src[L.len - 1 - i]);
            else memcpy(dst, src, (size_t) L.len);
            dst[L.len] = 0;
        });
        const double t_regions = since(t0);
        {
            DevMem d;
            HIPCHK(d.get((size_t) T * 8));
            uint8_t* D = d.as<uint8_t>();
            HIPCHK(hipMemcpyAsync(D, reg, tot, hipMemcpyHostToDevice, ctx->stream));
            std::vector<SigJob> jobs(ns);
            for (int j = 0; j < ns; ++j) {
                SigJob& J = jobs[j];
                memset(&J, 0, sizeof J);
                J.b_off = at[j]; J.out_off = at[j]; J.b_len = loci[c0 + (j < m ? j : j - m)].len; J.left = slot_left(j); J.right = slot_right(j);
            }
            SignalArgs A;
            memset(&A, 0, sizeof A);
            A.codes = D;
            A.sig5 = (int16_t*) (D + T); A.sig3 = (int16_t*) (D + 3 * T);
            A.cano5 = D + 5 * T; A.cano3 = D + 6 * T; A.dinc = D + 7 * T;
            if (spdp_signals_run(ctx, &sigm, jobs, A, nullptr, nullptr)) return -1;
            HIPCHK(hipMemcpy(H + T, D + T, (size_t) T * 7, hipMemcpyDeviceToHost));
        }
        if (getenv("SPDP_MAP_VERBOSE")) fprintf(stderr, "[map] regions cut %.3f s, signals made and brought back %.3f s\n", t_regions, since(t0) - t_regions);
        std::vector<SpdpProblem> probs(ns);
        std::vector<const SpdpJuxt*> hl(m);
        std::vector<int32_t> hn(m), low(m, 0);
        for (int j = 0; j < ns; ++j) {
            const int k = j < m ? j : j - m;
            const SpdpLocus& L = loci[c0 + k];
            SpdpProblem& P = probs[j];
            memset(&P, 0, sizeof P);
            P.a = (j < m ? codes : codes_rc.data()) + offs[L.query]; P.a_len = (int32_t) (offs[L.query + 1] - offs[L.query]);
            P.b = reg + at[j]; P.b_len = L.len;
            P.sig5 = sig5 + at[j]; P.sig3 = sig3 + at[j];
            P.cano5 = cano5 + at[j]; P.cano3 = cano3 + at[j]; P.dinc = dinc + at[j];
            P.a_left = 0; P.a_right = P.a_len; P.b_left = slot_left(j); P.b_right = slot_right(j);
            P.a_exgl = P.a_exgr = P.b_exgl = P.b_exgr = 1;
            if (j < m) { hl[k] = hsps + L.hsp_off; hn[k] = L.n_hsp; }
        }
        sec[1] += since(t0);
        // ---- the aligner on every locus, then the printer's scores
        t0 = std::chrono::steady_clock::now();
        std::vector<SpdpAlignment> aln(m);
        std::vector<int32_t> orient(m, 0);
        const int rc = both ? spdp_align_s_seeded_ori3(ctx, sc, &spx, probs.data(), probs.data() + m, m, hl.data(), hn.data(), low.data(), nullptr, aln.data(), orient.data())
                            : spdp_align_s_seeded(ctx, sc, &spx, probs.data(), m, hl.data(), hn.data(), low.data(), nullptr, aln.data());
        if (rc < 0) return -1;
        if (both) for (int k = 0; k < m; ++k) if (orient[k]) probs[k] = probs[m + k];        // rescoring reads the pair that was aligned
        if (rc > 0) ++partial;
        sec[2] += since(t0);
        if (getenv("SPDP_MAP_VERBOSE")) {
            int64_t st[12] = {0};
            spdp_seeded_stats(ctx, st, 11);
            fprintf(stderr, "[map] chunk of %d loci, %.1f M positions: regions + signals %.3f s; seeded call %.3f s (upload %.3f, walks with the device idle %.3f, "
                    "device batches %.3f, handing back %.3f; %lld batches, %lld + %lld DP requests, %lld HSP searches)\n", m, tot / 1e6, sec[1], since(t0),
                    st[6] / 1e6, st[7] / 1e6, st[8] / 1e6, st[9] / 1e6, (long long) st[0], (long long) st[1], (long long) st[2], (long long) st[4]);
        }
        t0 = std::chrono::steady_clock::now();
        std::vector<SpdpRescored> res(m);
        memset(res.data(), 0, sizeof(SpdpRescored) * m);
        if (spdp_skl_rng_s(ctx, sc, rp, probs.data(), m, aln.data(), res.data())) { spdp_free_alignments(aln.data(), m); return -1; }
        for (int k = 0; k < m; ++k) {
            if (aln[k].n_skl < 1) continue;
            const SpdpLocus& L = loci[c0 + k];
            SpdpMapGene& G = genes[L.query];
            ++G.n_loci;
            if (G.chr >= 0 && G.val >= res[k].val) continue;
            const int rvs = orient[k] ? !L.rvs : (L.rvs != 0);                                    // the strand the aligned region lies on
            const int a_len = probs[k].a_len;
            G.chr = L.chr; G.rvs = rvs; G.q_rev = orient[k]; G.score = res[k].score; G.val = res[k].val;
            std::vector<SpdpMapExon>& ex = kept[L.query];
            ex.clear();
            auto site = [&L, rvs](int pos) { return L.base + (rvs ? L.len - pos : pos + 1); };      // Seq::SiteNo
            for (int e = 0; e < res[k].n_exons; ++e) {
                const SpdpExon& x = res[k].exons[e];
                if (x.left > (1 << 30)) continue;                                                 // (the closing record of the list)
                if (orient[k]) ex.push_back({a_len - x.rleft, a_len - x.rright + 1, site(x.left), site(x.right - 1)});     // positions of the query as given
                else ex.push_back({x.rleft + 1, x.rright, site(x.left), site(x.right - 1)});
            }
        }
        spdp_free_rescored(res.data(), m);
        spdp_free_alignments(aln.data(), m);
        sec[3] += since(t0);
        c0 = c1;
    }
    size_t ne = 0;
    for (int i = 0; i < n; ++i) ne += kept[i].size();
    *exons = (SpdpMapExon*) malloc(sizeof(SpdpMapExon) * std::max<size_t>(ne, 1));
    if (!*exons) { ctx->err = "spdp_map_align_s: out of memory"; return -1; }
    size_t o = 0;
    for (int i = 0; i < n; ++i) {
        genes[i].exon_off = (int64_t) o; genes[i].n_exons = (int32_t) kept[i].size();
        if (!kept[i].empty()) memcpy(*exons + o, kept[i].data(), sizeof(SpdpMapExon) * kept[i].size());
        o += kept[i].size();
    }
    if (seconds) memcpy(seconds, sec, sizeof sec);
    if (partial) { ctx->err = "spdp_map_align_s: some walks met a state the seeded path does not serve; those loci come back without an alignment"; return 1; }
    return 0;
}

extern "C" int spdp_map_align_s(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                                const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpSignalModel* sigmodel,
                                const SpdpBlkFindParams* fprm, const SpdpRescoreParams* rp,
                                const uint8_t* codes, const int64_t* offs, int32_t n, int32_t ori,
                                SpdpMapGene* genes, SpdpMapExon** exons, double* seconds)
{
    try { return map_align_s(ctx, ix, hix, genome, sc, sp, sigmodel, fprm, rp, codes, offs, n, ori, genes, exons, seconds); }
    catch (const std::bad_alloc&) {                     // (nothing of C++ crosses the C boundary)
        if (ctx) ctx->err = "spdp_map_align_s: out of host memory (SPDP_MAP_CHUNK_MB sets the size of a chunk)";
        if (exons && *exons) { free(*exons); *exons = nullptr; }
        return -1;
    }
}


// ---- protein queries: spdp_map_align_h ---------------------------------------------------------------------------------------------
// The same chain for amino-acid queries against the translated index (`spaln -W -KP`): spdp_blk_find (the vote on the amino-acid
// words, the HSP search on regions read as tron codes) -> per locus the region as the aligner reads it (other strand, Seq::nuc2tron) and
// its SGPT6 signals, all loci of a chunk in one launch of spdp_signals_h -> spdp_align_h_seeded with the library's own HSP searches ->
// the junction phases the walks chose written back (skl_rngH_ng reads them: spdp_seeded_phase_marks) -> spdp_skl_rng_h -> the locus with
// the highest fstat.val.  What blkaln / genomicseq / spalign2 do per query (src/spaln.cc:846-1010, 1137-1152), for a batch.
static int map_align_h(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                       const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpSignalModelH* sigmodel,
                       const SpdpBlkFindParams* fprm, const SpdpRescoreParamsH* rp,
                       const uint8_t* codes, const int64_t* offs, int32_t n,
                       SpdpMapGene* genes, SpdpMapExon** exons, double* seconds)
{
    if (!ctx) return -1;
    if (!ix || !hix || !genome || !sc || !sp || !sigmodel || !fprm || !rp || !codes || !offs || !genes || !exons) {
        ctx->err = "spdp_map_align_h: null argument"; return -1;
    }
    if (!sp->wilip || sp->wilip->dvsp != 1) { ctx->err = "spdp_map_align_h: SpdpSeedParams.wilip must be the protein model (dvsp = 1)"; return -1; }
    if (!sc->intpen || sc->intpen_len <= 0) { ctx->err = "spdp_map_align_h: SpdpScoringH.intpen missing"; return -1; }
    *exons = nullptr;
    double sec[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) { genes[i].chr = -1; genes[i].rvs = 0; genes[i].q_rev = 0; genes[i].score = SPDP_NEVSEL; genes[i].val = 0; genes[i].n_loci = 0; genes[i].n_exons = 0; genes[i].exon_off = 0; }
    if (n <= 0) return 0;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<int32_t> ql(n, 0), qr(n);
    for (int i = 0; i < n; ++i) qr[i] = (int32_t) (offs[i + 1] - offs[i]);
    SpdpScoring chain_costs;                            // (the gap and intron prices the HSP chaining reads)
    memset(&chain_costs, 0, sizeof chain_costs);
    chain_costs.gop = sc->gop; chain_costs.gep = sc->gep; chain_costs.lgop = sc->lgop; chain_costs.lgep = sc->lgep; chain_costs.codonk1 = sc->codonk1;
    chain_costs.intpen = sc->intpen; chain_costs.intpen_len = sc->intpen_len;
    SpdpLocus* loci = nullptr; SpdpJuxt* hsps = nullptr; int32_t n_loci = 0;
    if (spdp_blk_find(ctx, ix, hix, genome, sp->wilip, &chain_costs, fprm, codes, offs, ql.data(), qr.data(), n, &loci, &n_loci, &hsps, nullptr)) return -1;
    struct Owned { SpdpLocus* l; SpdpJuxt* h; ~Owned() { free(l); free(h); } } owned{loci, hsps};
    sec[0] = since(t0);
    size_t chunk_positions = (size_t) 512 << 20;        // 14 B per position on both sides of the bus
    if (const char* e = getenv("SPDP_MAP_CHUNK_MPOS")) chunk_positions = (size_t) std::max(1, atoi(e)) << 20;
    std::vector<std::vector<SpdpMapExon>> kept(n);
    int partial = 0;
    (void) hipSetDevice(ctx->device);
    for (int k = 0; k < n_loci; ++k) {
        const SpdpLocus& L = loci[k];
        if (L.chr < 0 || L.chr >= genome->n_chr || L.base < 0 || L.len < 0 || L.left < 0 || L.right > L.len || L.right < L.left ||
            genome->chr_off[L.chr] + L.base + L.len > genome->chr_off[L.chr + 1]) { ctx->err = "spdp_map_align_h: a locus outside its chromosome"; return -1; }
    }
    for (int c0 = 0; c0 < n_loci; ) {
        t0 = std::chrono::steady_clock::now();
        std::vector<int64_t> at;                        // first position of locus c0 + k in the chunk's arrays (len + 3 positions each)
        int64_t tot = 0;
        int c1 = c0;
        while (c1 < n_loci && (c1 == c0 || tot + loci[c1].len + 3 <= (int64_t) chunk_positions)) { at.push_back(tot); tot += loci[c1].len + 3; ++c1; }
        const int m = c1 - c0;
        const int64_t T = (tot + 255) / 256 * 256;
        // one host block: tron codes | sig5 sig3 sigS sigT sigE (int16) | phs5 phs3 (int8) | dinc
        uint8_t* Hbuf = (uint8_t*) ctx->staging(2, (size_t) T * 14);       // (pinned, kept by the context)
        if (!Hbuf) { ctx->err = "spdp_map_align_h: no pinned host memory for a chunk's regions and signals (SPDP_MAP_CHUNK_MPOS sets the chunk size)"; return -1; }
        uint8_t* reg = Hbuf;
        int16_t* s16[5]; for (int i = 0; i < 5; ++i) s16[i] = (int16_t*) (Hbuf + T + 2 * T * i);
        int8_t* phs5 = (int8_t*) (Hbuf + 11 * T); int8_t* phs3 = (int8_t*) (Hbuf + 12 * T);
        uint8_t* dinc = Hbuf + 13 * T;
        on_host_threads(m, [&](int j) {
            const SpdpLocus& L = loci[c0 + j];
            spdp_region::materialize_into(genome->codes, genome->chr_off, L.chr, L.base, L.len, L.rvs != 0, true, reg + at[j]);
            reg[at[j] + L.len + 1] = reg[at[j] + L.len + 2] = 0;
        });
        {
            DevMem d;
            HIPCHK(d.get((size_t) T * 15));
            uint8_t* D = d.as<uint8_t>();
            HIPCHK(hipMemcpyAsync(D, reg, tot, hipMemcpyHostToDevice, ctx->stream));
            std::vector<SigJobH> jobs(m);
            for (int j = 0; j < m; ++j) {
                SigJobH& J = jobs[j];
                memset(&J, 0, sizeof J);
                J.b_off = at[j]; J.out_off = at[j]; J.b_len = loci[c0 + j].len; J.left = loci[c0 + j].left; J.right = loci[c0 + j].right;
            }
            SignalArgsH A;
            memset(&A, 0, sizeof A);
            A.codes = D;
            A.sig5 = (int16_t*) (D + T); A.sig3 = (int16_t*) (D + 3 * T); A.sigS = (int16_t*) (D + 5 * T); A.sigT = (int16_t*) (D + 7 * T);
            A.sigE = (int16_t*) (D + 9 * T); A.phs5 = (int8_t*) (D + 11 * T); A.phs3 = (int8_t*) (D + 12 * T); A.dinc = D + 13 * T; A.cano = D + 14 * T;
            if (spdh_signals_run(ctx, sigmodel, jobs, A, 0)) return -1;
            HIPCHK(hipMemcpy(Hbuf + T, D + T, (size_t) T * 13, hipMemcpyDeviceToHost));
        }
        std::vector<SpdpProblemH> probs(m);
        std::vector<const SpdpJuxt*> hl(m);
        std::vector<int32_t> hn(m), low(m, 0);
        for (int j = 0; j < m; ++j) {
            const SpdpLocus& L = loci[c0 + j];
            SpdpProblemH& P = probs[j];
            memset(&P, 0, sizeof P);
            P.a = codes + offs[L.query]; P.a_len = (int32_t) (offs[L.query + 1] - offs[L.query]);
            P.b = reg + at[j]; P.b_len = L.len;
            P.sig5 = s16[0] + at[j]; P.sig3 = s16[1] + at[j]; P.sigS = s16[2] + at[j]; P.sigT = s16[3] + at[j]; P.sigE = s16[4] + at[j];
            P.phs5 = phs5 + at[j]; P.phs3 = phs3 + at[j]; P.dinc = dinc + at[j];
            P.exin_left = L.left; P.exin_right = L.right;
            P.a_left = 0; P.a_right = P.a_len; P.b_left = L.left; P.b_right = L.right;
            P.a_exgl = P.a_exgr = P.b_exgl = P.b_exgr = 1;
            hl[j] = hsps + L.hsp_off; hn[j] = L.n_hsp;
        }
        sec[1] += since(t0);
        t0 = std::chrono::steady_clock::now();
        std::vector<SpdpAlignment> aln(m);
        const int rc = spdp_align_h_seeded(ctx, sc, sp, probs.data(), m, hl.data(), hn.data(), low.data(), nullptr, aln.data());
        if (rc < 0) return -1;
        if (rc > 0) ++partial;
        struct Alns { SpdpAlignment* a; int n; ~Alns() { spdp_free_alignments(a, n); } } alns{aln.data(), m};
        // the phases of the junctions the walks chose themselves: where the reference's walk writes into its Exinon
        for (int j = 0; j < m; ++j) {
            const SpdpPhaseMark* mk = nullptr;
            const int nm = spdp_seeded_phase_marks(ctx, j, &mk);
            for (int i = 0; i < nm; ++i) {
                if (mk[i].n < 0 || mk[i].n > loci[c0 + j].len + 2) continue;
                (mk[i].side == 5 ? phs5 : phs3)[at[j] + mk[i].n] = mk[i].value;
            }
        }
        sec[2] += since(t0);
        t0 = std::chrono::steady_clock::now();
        std::vector<SpdpRescored> res(m);
        memset(res.data(), 0, sizeof(SpdpRescored) * m);
        if (spdp_skl_rng_h(ctx, sc, rp, probs.data(), m, aln.data(), res.data())) return -1;
        for (int j = 0; j < m; ++j) {
            if (aln[j].n_skl < 1) continue;
            const SpdpLocus& L = loci[c0 + j];
            SpdpMapGene& G = genes[L.query];
            ++G.n_loci;
            if (G.chr >= 0 && G.val >= res[j].val) continue;
            G.chr = L.chr; G.rvs = L.rvs != 0; G.q_rev = 0; G.score = res[j].score; G.val = res[j].val;
            std::vector<SpdpMapExon>& ex = kept[L.query];
            ex.clear();
            const int rvs = L.rvs != 0;
            auto site = [&L, rvs](int pos) { return L.base + (rvs ? L.len - pos : pos + 1); };      // Seq::SiteNo
            for (int e = 0; e < res[j].n_exons; ++e) {
                const SpdpExon& x = res[j].exons[e];
                if (x.left > (1 << 30)) continue;                                                 // (the closing record of the list)
                ex.push_back({x.rleft + 1, x.rright, site(x.left), site(x.right - 1)});
            }
        }
        spdp_free_rescored(res.data(), m);
        sec[3] += since(t0);
        c0 = c1;
    }
    size_t ne = 0;
    for (int i = 0; i < n; ++i) ne += kept[i].size();
    *exons = (SpdpMapExon*) malloc(sizeof(SpdpMapExon) * std::max<size_t>(ne, 1));
    if (!*exons) { ctx->err = "spdp_map_align_h: out of memory"; return -1; }
    size_t o = 0;
    for (int i = 0; i < n; ++i) {
        genes[i].exon_off = (int64_t) o; genes[i].n_exons = (int32_t) kept[i].size();
        if (!kept[i].empty()) memcpy(*exons + o, kept[i].data(), sizeof(SpdpMapExon) * kept[i].size());
        o += kept[i].size();
    }
    if (seconds) memcpy(seconds, sec, sizeof sec);
    if (partial) { ctx->err = "spdp_map_align_h: some walks met a state the seeded path does not serve; those loci come back without an alignment"; return 1; }
    return 0;
}

extern "C" int spdp_map_align_h(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                                const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpSignalModelH* sigmodel,
                                const SpdpBlkFindParams* fprm, const SpdpRescoreParamsH* rp,
                                const uint8_t* codes, const int64_t* offs, int32_t n,
                                SpdpMapGene* genes, SpdpMapExon** exons, double* seconds)
{
    try { return map_align_h(ctx, ix, hix, genome, sc, sp, sigmodel, fprm, rp, codes, offs, n, genes, exons, seconds); }
    catch (const std::bad_alloc&) {
        if (ctx) ctx->err = "spdp_map_align_h: out of host memory (SPDP_MAP_CHUNK_MPOS sets the size of a chunk)";
        if (exons && *exons) { free(*exons); *exons = nullptr; }
        return -1;
    }
}
