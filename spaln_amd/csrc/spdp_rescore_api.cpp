// spdp_rescore_api.cpp -- host side of spdp_skl_rng_s: packs the corner lists, launches
// spdp_rescore_s (spdp_rescore.hip) and hands the records back through the C ABI.

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spdp.h"
#include "spdp_dev.h"
#include "spdp_internal.h"

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

int spdp_skl_rng_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpRescoreParams* rp,
                   const SpdpProblem* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out)
{
    if (!ctx || !sc || !rp || !probs || n_probs < 0 || !aln || !out) return -1;
    if (rp->jneibr < 1 || rp->jneibr > 32) { ctx->err = "jneibr out of range (1 .. 32)"; return -1; }
    for (int i = 0; i < n_probs; ++i) { memset(&out[i], 0, sizeof out[i]); out[i].score = SPDP_NEVSEL; }
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
    HIPCHK(spdp_launch_rescore(&A, ctx->stream));
    std::vector<int> hdr((size_t) nr * 8), rec((size_t) rtot * 21);
    HIPCHK(hipMemcpyAsync(hdr.data(), d_hdr, hdr.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(rec.data(), d_rec, rec.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < nr; ++s) {
        SpdpRescored& o = out[idx[s]];
        const int* h = &hdr[(size_t) s * 8];
        o.score = h[0]; o.mch = h[1]; o.mmc = h[2]; o.gap = h[3]; o.unp = h[4]; o.val = h[5];
        o.n_exons = h[6];
        o.exons = (SpdpExon*) malloc(sizeof(SpdpExon) * (size_t) std::max(1, o.n_exons));
        memcpy(o.exons, &rec[(size_t) roff[s] * 21], sizeof(SpdpExon) * (size_t) o.n_exons);
    }
    return 0;
}

void spdp_free_rescored(SpdpRescored* out, int n)
{
    for (int i = 0; i < n; ++i) { free(out[i].exons); out[i].exons = nullptr; out[i].n_exons = 0; }
}
