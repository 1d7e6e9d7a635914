// spdp_rescore_api.cpp -- host side of spdp_skl_rng_s: packs the corner lists, launches
// spdp_rescore_s (spdp_rescore.hip) and hands the records back through the C ABI.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spdp.h"
#include "spdp_dev.h"
#include "spdp_internal.h"
#include "spdp_hostcpus.h"
#include <thread>
#include <atomic>
#include "spdp_gencode.h"

#define HIPCHK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

enum { R_POOL = 7 };
enum { RP_PROBS = 0, RP_SKL, RP_SOFF, RP_SCNT, RP_ROFF, RP_HDR, RP_REC };

// the rescoring walk over a batch; `out` (records of spdp_skl_rng_s) and `edits` (format != 0: spdp_skl_edits_s) may each be null
static int rescore_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpRescoreParams* rp,
                     const SpdpProblem* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out,
                     int format, SpdpEdits* edits)
{
    if (!ctx || !sc || !rp || !probs || n_probs < 0 || !aln || (!out && !edits)) return -1;
    if (rp->jneibr < 1 || rp->jneibr > 32) { ctx->err = "jneibr out of range (1 .. 32)"; return -1; }
    if (edits && (format < SPDP_FMT_CIGAR || format > SPDP_FMT_SAM)) { ctx->err = "edit records: unknown format"; return -1; }
    for (int i = 0; i < n_probs; ++i) {
        if (out) { memset(&out[i], 0, sizeof out[i]); out[i].score = SPDP_NEVSEL; }
        if (edits) memset(&edits[i], 0, sizeof edits[i]);
    }
    if (!n_probs) return 0;
    DevStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    if (!st.has_exact) { ctx->err = "rescoring needs SpdpScoring.intpen / t53 and SpdpProblem.dinc"; return -1; }
    // queries that have an alignment
    std::vector<int> idx;
    std::vector<DevProblem> descs;
    std::vector<SpdpSkl> skl;
    std::vector<int64_t> soff, roff;
    std::vector<int> scnt;
    int64_t rtot = 0;
    for (int i = 0; i < n_probs; ++i) {
        const int cnt = aln[i].n_skl - 1;                    // corners after the header record
        if (cnt < 2 || !aln[i].skl) continue;
        if (probs[i].b_right - probs[i].b_left >= sc->intpen_len) { ctx->err = "intpen table shorter than the window"; return -1; }
        DevProblem d;
        memset(&d, 0, sizeof d);
        d.a_left = probs[i].a_left; d.a_right = probs[i].a_right; d.b_left = probs[i].b_left; d.b_right = probs[i].b_right;
        d.flags = (probs[i].a_exgl ? 1 : 0) | (probs[i].a_exgr ? 2 : 0) | (probs[i].b_exgl ? 4 : 0) | (probs[i].b_exgr ? 8 : 0);
        d.a_off = st.a_off[i]; d.col_off = st.col_off[i];
        d.cip_off = st.cip_off.empty() ? -1 : st.cip_off[i];
        idx.push_back(i); descs.push_back(d);
        soff.push_back((int64_t) skl.size()); scnt.push_back(cnt);
        skl.insert(skl.end(), aln[i].skl + 1, aln[i].skl + 1 + cnt);
        roff.push_back(rtot);
        rtot += cnt / 2 + 3;                                 // an exon needs two corners; + last exon, end marker
    }
    const int nr = (int) idx.size();
    if (!nr) return 0;
    DevPool& pool = ctx->pool[R_POOL];
    void* d_probs = pool.get(RP_PROBS, nr * sizeof(DevProblem));
    void* d_skl = pool.get(RP_SKL, skl.size() * sizeof(SpdpSkl));
    void* d_soff = pool.get(RP_SOFF, nr * sizeof(int64_t));
    void* d_scnt = pool.get(RP_SCNT, nr * sizeof(int));
    void* d_roff = pool.get(RP_ROFF, nr * sizeof(int64_t));
    void* d_hdr = pool.get(RP_HDR, (size_t) nr * 8 * sizeof(int));
    void* d_rec = pool.get(RP_REC, (size_t) rtot * 21 * sizeof(int));
    if (!d_probs || !d_skl || !d_soff || !d_scnt || !d_roff || !d_hdr || !d_rec) { ctx->err = "out of device memory"; return -1; }
    HIPCHK(hipMemcpyAsync(d_probs, descs.data(), nr * sizeof(DevProblem), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_skl, skl.data(), skl.size() * sizeof(SpdpSkl), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_soff, soff.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_scnt, scnt.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_roff, roff.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    RescoreArgs A;
    memset(&A, 0, sizeof A);
    A.sc = (const DevScoring*) st.d_sc; A.probs = (const DevProblem*) d_probs; A.n_probs = nr;
    A.a_codes = (const uint8_t*) st.d_a; A.cols = (const int2*) st.d_cols; A.aux = (const uint8_t*) st.d_aux;
    A.intpen = (const int16_t*) st.d_intpen; A.intpen_len = sc->intpen_len;
    A.skl = (const int2*) d_skl; A.skl_off = (const int64_t*) d_soff; A.skl_cnt = (const int*) d_scnt;
    A.rec_off = (const int64_t*) d_roff; A.out_hdr = (int*) d_hdr; A.out_rec = (int*) d_rec;
    A.gop = sc->gop; A.gep = sc->gep; A.lgop = sc->lgop; A.lgep = sc->lgep;
    A.codonk1 = rp->codonk1; A.minl = rp->minl; A.jneibr = rp->jneibr; A.lsg = rp->lsg; A.ipen = sc->spj ? sc->ipen : 0;
    memcpy(A.t53, sc->t53, sizeof A.t53);
    if (!sc->spj && rp->lsg) { ctx->err = "splice-aware rescoring of a problem set uploaded without splice signals"; return -1; }
    // edit records: Cigar / Vulgar need at most 7 per corner, SAM up to two per aligned base on top of that
    std::vector<int64_t> ooff(nr + 1, 0);
    std::vector<int> alen(nr);
    void *d_ops = nullptr, *d_ooff = nullptr, *d_ocnt = nullptr, *d_alen = nullptr;
    struct Freer { void** p[4]; ~Freer() { for (void** q : p) if (*q) (void) hipFree(*q); } } freer{{&d_ops, &d_ooff, &d_ocnt, &d_alen}};
    if (edits) {
        for (int s = 0; s < nr; ++s) {
            alen[s] = probs[idx[s]].a_len;
            ooff[s + 1] = ooff[s] + 8ll * scnt[s] + 16 + (format == SPDP_FMT_SAM ? 2ll * alen[s] + 16 : 0);
        }
        HIPCHK(hipMalloc(&d_ops, (size_t) ooff[nr] * sizeof(int3)));
        HIPCHK(hipMalloc(&d_ooff, (nr + 1) * sizeof(int64_t)));
        HIPCHK(hipMalloc(&d_ocnt, nr * sizeof(int)));
        HIPCHK(hipMalloc(&d_alen, nr * sizeof(int)));
        HIPCHK(hipMemcpyAsync(d_ooff, ooff.data(), (nr + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_alen, alen.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        A.ops_format = format; A.ops = (int3*) d_ops; A.ops_off = (const int64_t*) d_ooff; A.ops_cnt = (int*) d_ocnt;
        A.a_len = (const int*) d_alen;
    }
    void* d_alen_all = nullptr;
    struct Freer1 { void** p; ~Freer1() { if (*p) (void) hipFree(*p); } } freer1{&d_alen_all};
    if (st.d_cip) {                                          // annotated intron positions (use_spb, src/fwd2s1.cc:615)
        std::vector<int> la(nr);
        for (int s = 0; s < nr; ++s) la[s] = probs[idx[s]].a_len;
        HIPCHK(hipMalloc(&d_alen_all, nr * sizeof(int)));
        HIPCHK(hipMemcpyAsync(d_alen_all, la.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        A.cip = (const int*) st.d_cip; A.a_len_all = (const int*) d_alen_all;
    }
    HIPCHK(spdp_launch_rescore(&A, ctx->stream));
    std::vector<int> hdr((size_t) nr * 8), rec((size_t) rtot * 21);
    HIPCHK(hipMemcpyAsync(hdr.data(), d_hdr, hdr.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(rec.data(), d_rec, rec.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < nr; ++s)
        if (hdr[(size_t) s * 8 + 7]) { ctx->err = "rescoring (protein): a read outside the region window held on the device (SPDP_RESCORE_WINDOW=0 holds whole regions)"; return -1; }
    for (int s = 0; s < nr && out; ++s) {
        SpdpRescored& o = out[idx[s]];
        const int* h = &hdr[(size_t) s * 8];
        o.score = h[0]; o.mch = h[1]; o.mmc = h[2]; o.gap = h[3]; o.unp = h[4]; o.val = h[5];
        o.n_exons = h[6];
        o.exons = (SpdpExon*) malloc(sizeof(SpdpExon) * (size_t) std::max(1, o.n_exons));
        memcpy(o.exons, &rec[(size_t) roff[s] * 21], sizeof(SpdpExon) * (size_t) o.n_exons);
    }
    if (edits) {
        std::vector<int3> ops((size_t) ooff[nr]);
        std::vector<int> ocnt(nr);
        HIPCHK(hipMemcpy(ops.data(), d_ops, ops.size() * sizeof(int3), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ocnt.data(), d_ocnt, nr * sizeof(int), hipMemcpyDeviceToHost));
        for (int s = 0; s < nr; ++s) {
            SpdpEdits& e = edits[idx[s]];
            if (ocnt[s] > ooff[s + 1] - ooff[s]) { ctx->err = "edit records: slot too small"; return -1; }
            e.n = ocnt[s];
            e.rec = (SpdpEdit*) malloc(sizeof(SpdpEdit) * (size_t) std::max(1, e.n));
            for (int k = 0; k < e.n; ++k) { const int3 o = ops[(size_t) ooff[s] + k]; e.rec[k].op = o.x; e.rec[k].alen = o.y; e.rec[k].blen = o.z; }
            if (format == SPDP_FMT_SAM) {
                // Samfmt's header fields for a forward-strand hit (b->inex.sens == 0; src/fwd2s1.cc:492-495, 678-687)
                const SpdpAlignment& al = aln[idx[s]];
                const SpdpProblem& p = probs[idx[s]];
                int first = 1;
                if (al.n_skl > 2 && al.skl[2].n == al.skl[1].n && p.b_exgl) first = 2;
                const int* h = &hdr[(size_t) s * 8];
                e.sam_flag = 0;
                e.sam_pos = al.skl[first].n;
                e.sam_left = al.skl[first].m;
                e.sam_right = al.skl[al.n_skl - 1].m;
                e.sam_mapq = 30 + (int) (100ll * (h[2] + h[4]) / std::max(1, p.a_len));
            }
        }
    }
    return 0;
}

int spdp_skl_rng_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpRescoreParams* rp,
                   const SpdpProblem* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out)
{
    if (!out) return -1;
    return rescore_s(ctx, sc, rp, probs, n_probs, aln, out, 0, nullptr);
}

int spdp_skl_edits_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpRescoreParams* rp,
                     const SpdpProblem* probs, int n_probs, const SpdpAlignment* aln, int format, SpdpEdits* out)
{
    if (!out) return -1;
    return rescore_s(ctx, sc, rp, probs, n_probs, aln, nullptr, format, out);
}

void spdp_free_edits(SpdpEdits* out, int n)
{
    for (int i = 0; i < n; ++i) { free(out[i].rec); out[i].rec = nullptr; out[i].n = 0; }
}

void spdp_free_rescored(SpdpRescored* out, int n)
{
    for (int i = 0; i < n; ++i) { free(out[i].exons); out[i].exons = nullptr; out[i].n_exons = 0; }
}

// ---- protein alignments: skl_rngH_ng ---------------------------------------------------------
enum { RH_MTX = 8, RH_PROBS, RH_A, RH_B, RH_SIG, RH_PHS, RH_DINC, RH_INTPEN };   // pool slots after the cDNA ones

static int rescore_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp,
                     const SpdpProblemH* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out,
                     int format, SpdpEdits* edits)
{
    if (!ctx || !sc || !rp || !probs || n_probs < 0 || !aln || (!out && !edits)) return -1;
    if (rp->jneibr < 1 || rp->jneibr > 32) { ctx->err = "jneibr out of range (1 .. 32)"; return -1; }
    if (edits && format != SPDP_FMT_CIGAR && format != SPDP_FMT_VULGAR) { ctx->err = "edit records (protein): Cigar or Vulgar"; return -1; }
    for (int i = 0; i < n_probs; ++i) {
        if (out) { memset(&out[i], 0, sizeof out[i]); out[i].score = SPDP_NEVSEL; }
        if (edits) memset(&edits[i], 0, sizeof edits[i]);
    }
    if (!n_probs) return 0;
    if (!sc->intpen || sc->intpen_len <= 0) { ctx->err = "rescoring needs SpdpScoringH.intpen / t53"; return -1; }
    if (sc->mtx_rows > 32 || sc->mtx_cols > 32) { ctx->err = "matrix larger than 32 x 32"; return -1; }
    std::vector<int> idx;
    std::vector<HRescoreProb> descs;
    std::vector<uint8_t> a_all, b_all, dinc;
    std::vector<short> sig;
    std::vector<int8_t> phs;
    std::vector<SpdpSkl> skl;
    std::vector<int64_t> soff, roff;
    std::vector<int> scnt;
    int64_t rtot = 0;
    // where every problem's pieces go, then the pieces themselves on the host's threads (a batch of a map + align call is tens of
    // thousands of loci of tens of kilobases: hundreds of millions of positions)
    int64_t a_tot = 0, b_tot = 0, col_tot = 0, skl_tot = 0;
    const char* we = getenv("SPDP_RESCORE_WINDOW");
    const bool windowed = !(we && atoi(we) == 0);
    const int margin = 64 + 3 * rp->jneibr;
    for (int i = 0; i < n_probs; ++i) {
        const SpdpProblemH& p = probs[i];
        const int cnt = aln[i].n_skl - 1;
        if (cnt < 2 || !aln[i].skl) continue;
        if (!p.dinc) { ctx->err = "rescoring needs SpdpProblemH.dinc"; return -1; }
        if (p.b_len + 2 > sc->intpen_len) { ctx->err = "intpen table shorter than the window"; return -1; }
        HRescoreProb d;
        memset(&d, 0, sizeof d);
        d.a_left = p.a_left; d.a_right = p.a_right; d.b_left = p.b_left; d.b_right = p.b_right;
        d.a_len = p.a_len; d.b_len = p.b_len;
        d.a_exgl = p.a_exgl; d.a_exgr = p.a_exgr; d.b_exgl = p.b_exgl; d.b_exgr = p.b_exgr;
        // the region positions the walk over the corners can read: what the corners span and a margin (junction neighbourhoods, the
        // codons around a split codon, the stop codon behind the last corner); SPDP_RESCORE_WINDOW=0: the whole region
        int y_lo = INT32_MAX, y_hi = INT32_MIN;
        for (int k = 1; k <= cnt; ++k) { y_lo = std::min(y_lo, aln[i].skl[k].n); y_hi = std::max(y_hi, aln[i].skl[k].n); }
        d.w_lo = windowed ? std::max(0, y_lo - margin) : 0;
        d.w_hi = windowed ? std::min(p.b_len + 3, y_hi + margin) : p.b_len + 3;
        if (d.w_hi < d.w_lo) d.w_hi = d.w_lo;
        d.a_off = a_tot; d.b_off = b_tot; d.col_off = col_tot;
        a_tot += p.a_len; b_tot += d.w_hi - d.w_lo; col_tot += d.w_hi - d.w_lo;
        idx.push_back(i); descs.push_back(d);
        soff.push_back(skl_tot); scnt.push_back(cnt);
        skl_tot += cnt;
        roff.push_back(rtot);
        rtot += cnt + 3;                                    // exons, frame-shift records, end marker
    }
    // the region pieces go through the context's pinned staging blocks when it has them (no zero fill, and the bus at its rate: a map +
    // align call rescoring 20 000 loci moves 2 GB here); pageable vectors otherwise
    a_all.resize((size_t) a_tot); skl.resize((size_t) skl_tot);
    short* sig_p = (short*) ctx->staging(0, std::max<size_t>((size_t) col_tot * 5 * sizeof(short), 64));
    uint8_t* misc_p = sig_p ? (uint8_t*) ctx->staging(1, std::max<size_t>((size_t) b_tot + (size_t) col_tot * 3, 64)) : nullptr;
    if (!sig_p || !misc_p) {
        b_all.resize((size_t) b_tot); sig.resize((size_t) col_tot * 5); phs.resize((size_t) col_tot * 2); dinc.resize((size_t) col_tot);
        sig_p = sig.data();
    }
    uint8_t* const b_p = misc_p ? misc_p : b_all.data();
    int8_t* const phs_p = misc_p ? (int8_t*) (misc_p + b_tot) : phs.data();
    uint8_t* const dinc_p = misc_p ? misc_p + b_tot + 2 * col_tot : dinc.data();
    {
        std::atomic<int> next{0};
        auto work = [&] {
            for (int s; (s = next++) < (int) idx.size(); ) {
                const SpdpProblemH& p = probs[idx[s]];
                const HRescoreProb& d = descs[s];
                memcpy(a_all.data() + d.a_off, p.a, (size_t) p.a_len);
                uint8_t* bb = b_p + d.b_off;
                short* sg = sig_p + 5 * d.col_off; int8_t* ph = phs_p + 2 * d.col_off; uint8_t* dc = dinc_p + d.col_off;
                for (int x = d.w_lo; x < d.w_hi; ++x) {
                    const int o = x - d.w_lo;
                    bb[o] = x <= p.b_len ? p.b[x] : 0;
                    sg[5 * o] = p.sig5[x]; sg[5 * o + 1] = p.sig3[x]; sg[5 * o + 2] = p.sigS[x]; sg[5 * o + 3] = p.sigT[x]; sg[5 * o + 4] = p.sigE[x];
                    ph[2 * o] = p.phs5[x]; ph[2 * o + 1] = p.phs3[x];
                    dc[o] = x <= p.b_len ? p.dinc[x] : 0;
                }
                memcpy(skl.data() + soff[s], aln[idx[s]].skl + 1, sizeof(SpdpSkl) * (size_t) scnt[s]);
            }
        };
        const int nt = std::max(1, std::min(spdp_host_cpus(), (int) idx.size()));
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (std::thread& t : th) t.join();
    }
    const int nr = (int) idx.size();
    if (!nr) return 0;
    std::vector<int> mtx(32 * 32, 0);
    for (int i = 0; i < sc->mtx_rows; ++i)
        for (int j = 0; j < sc->mtx_cols; ++j) mtx[i * 32 + j] = sc->mtx[i * sc->mtx_cols + j];
    DevPool& pool = ctx->pool[R_POOL];
    void* d_mtx = pool.get(RH_MTX, mtx.size() * sizeof(int));
    void* d_probs = pool.get(RH_PROBS, nr * sizeof(HRescoreProb));
    void* d_a = pool.get(RH_A, a_all.size() + 16);
    void* d_b = pool.get(RH_B, (size_t) b_tot + 16);
    void* d_sig = pool.get(RH_SIG, std::max<size_t>((size_t) col_tot * 5 * sizeof(short), 16));
    void* d_phs = pool.get(RH_PHS, std::max<size_t>((size_t) col_tot * 2, 16));
    void* d_dinc = pool.get(RH_DINC, std::max<size_t>((size_t) col_tot, 16));
    void* d_intpen = pool.get(RH_INTPEN, sizeof(int16_t) * sc->intpen_len);
    void* d_skl = pool.get(RP_SKL, skl.size() * sizeof(SpdpSkl));
    void* d_soff = pool.get(RP_SOFF, nr * sizeof(int64_t));
    void* d_scnt = pool.get(RP_SCNT, nr * sizeof(int));
    void* d_roff = pool.get(RP_ROFF, nr * sizeof(int64_t));
    void* d_hdr = pool.get(RP_HDR, (size_t) nr * 8 * sizeof(int));
    void* d_rec = pool.get(RP_REC, (size_t) rtot * 21 * sizeof(int));
    if (!d_mtx || !d_probs || !d_a || !d_b || !d_sig || !d_phs || !d_dinc || !d_intpen || !d_skl || !d_soff ||
        !d_scnt || !d_roff || !d_hdr || !d_rec) { ctx->err = "out of device memory"; return -1; }
#define UP(dst, vec) HIPCHK(hipMemcpyAsync(dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, ctx->stream))
    UP(d_mtx, mtx); UP(d_probs, descs); UP(d_a, a_all);
    HIPCHK(hipMemcpyAsync(d_b, b_p, (size_t) b_tot, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_sig, sig_p, (size_t) col_tot * 5 * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_phs, phs_p, (size_t) col_tot * 2, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_dinc, dinc_p, (size_t) col_tot, hipMemcpyHostToDevice, ctx->stream));
    UP(d_skl, skl); UP(d_soff, soff); UP(d_scnt, scnt); UP(d_roff, roff);
#undef UP
    HIPCHK(hipMemcpyAsync(d_intpen, sc->intpen, sizeof(int16_t) * sc->intpen_len, hipMemcpyHostToDevice, ctx->stream));
    HRescoreArgs A;
    memset(&A, 0, sizeof A);
    A.mtx = (const int*) d_mtx; A.n_probs = nr; A.probs = (const HRescoreProb*) d_probs;
    A.a_codes = (const uint8_t*) d_a; A.b_codes = (const uint8_t*) d_b; A.sig = (const short*) d_sig;
    A.phs = (const int8_t*) d_phs; A.dinc = (const uint8_t*) d_dinc;
    A.intpen = (const int16_t*) d_intpen; A.intpen_len = sc->intpen_len;
    A.skl = (const int2*) d_skl; A.skl_off = (const int64_t*) d_soff; A.skl_cnt = (const int*) d_scnt;
    A.rec_off = (const int64_t*) d_roff; A.out_hdr = (int*) d_hdr; A.out_rec = (int*) d_rec;
    A.gop = sc->gop; A.gep = sc->gep; A.lgop = sc->lgop; A.lgep = sc->lgep; A.codonk1 = sc->codonk1;
    A.gape1 = sc->gape1; A.gape2 = sc->gape2; A.extragop = sc->extragop; A.diffu = sc->diffu; A.k1 = sc->k1;
    A.minl = rp->minl; A.jneibr = rp->jneibr; A.lcl = rp->lcl; A.sup_tcodon = rp->sup_tcodon;
    memcpy(A.t53, sc->t53, sizeof A.t53);
    spdp_genetic_code_tables(A.mid, A.tron_of);
    std::vector<int64_t> ooff(nr + 1, 0);
    void *d_ops = nullptr, *d_ooff = nullptr, *d_ocnt = nullptr;
    struct Freer { void** p[3]; ~Freer() { for (void** q : p) if (*q) (void) hipFree(*q); } } freer{{&d_ops, &d_ooff, &d_ocnt}};
    if (edits) {
        for (int s = 0; s < nr; ++s) ooff[s + 1] = ooff[s] + 12ll * scnt[s] + 16;       // an accepted intron pushes up to 8 Vulgar records
        HIPCHK(hipMalloc(&d_ops, (size_t) ooff[nr] * sizeof(int3)));
        HIPCHK(hipMalloc(&d_ooff, (nr + 1) * sizeof(int64_t)));
        HIPCHK(hipMalloc(&d_ocnt, nr * sizeof(int)));
        HIPCHK(hipMemcpyAsync(d_ooff, ooff.data(), (nr + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        A.ops_format = format; A.ops = (int3*) d_ops; A.ops_off = (const int64_t*) d_ooff; A.ops_cnt = (int*) d_ocnt;
    }
    HIPCHK(spdh_launch_rescore(&A, ctx->stream));
    std::vector<int> hdr((size_t) nr * 8), rec((size_t) rtot * 21);
    HIPCHK(hipMemcpyAsync(hdr.data(), d_hdr, hdr.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(rec.data(), d_rec, rec.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < nr; ++s)
        if (hdr[(size_t) s * 8 + 7]) { ctx->err = "rescoring (protein): a read outside the region window held on the device (SPDP_RESCORE_WINDOW=0 holds whole regions)"; return -1; }
    for (int s = 0; s < nr && out; ++s) {
        SpdpRescored& o = out[idx[s]];
        const int* h = &hdr[(size_t) s * 8];
        o.score = h[0]; o.mch = h[1]; o.mmc = h[2]; o.gap = h[3]; o.unp = h[4]; o.val = h[5];
        o.n_exons = h[6];
        o.exons = (SpdpExon*) malloc(sizeof(SpdpExon) * (size_t) std::max(1, o.n_exons));
        memcpy(o.exons, &rec[(size_t) roff[s] * 21], sizeof(SpdpExon) * (size_t) o.n_exons);
    }
    if (edits) {
        std::vector<int3> ops((size_t) ooff[nr]);
        std::vector<int> ocnt(nr);
        HIPCHK(hipMemcpy(ops.data(), d_ops, ops.size() * sizeof(int3), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ocnt.data(), d_ocnt, nr * sizeof(int), hipMemcpyDeviceToHost));
        for (int s = 0; s < nr; ++s) {
            SpdpEdits& e = edits[idx[s]];
            if (ocnt[s] > ooff[s + 1] - ooff[s]) { ctx->err = "edit records: slot too small"; return -1; }
            e.n = ocnt[s];
            e.rec = (SpdpEdit*) malloc(sizeof(SpdpEdit) * (size_t) std::max(1, e.n));
            for (int k = 0; k < e.n; ++k) { const int3 o = ops[(size_t) ooff[s] + k]; e.rec[k].op = o.x; e.rec[k].alen = o.y; e.rec[k].blen = o.z; }
        }
    }
    return 0;
}

int spdp_skl_rng_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp,
                   const SpdpProblemH* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out)
{
    if (!out) return -1;
    return rescore_h(ctx, sc, rp, probs, n_probs, aln, out, 0, nullptr);
}

int spdp_skl_edits_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp,
                     const SpdpProblemH* probs, int n_probs, const SpdpAlignment* aln, int format, SpdpEdits* out)
{
    if (!out) return -1;
    return rescore_h(ctx, sc, rp, probs, n_probs, aln, nullptr, format, out);
}
