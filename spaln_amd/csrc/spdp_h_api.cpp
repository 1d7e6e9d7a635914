// spdp_h_api.cpp -- host side of the aa x genome (Fwd2h1 `_wip`) path: the extern "C" entry points
// spdp_stripe31 / spdp_cells_h / spdp_wip_forward_h / spdp_homscore_h / spdp_align_h and the resident
// batch variant.  Packs problems into the HBM layout of spdp_h_dev.h, launches the kernels of
// spdp_h_kernels.hip and runs the reference's dispatch around them (Aln2h1::lspH_ng /
// trcbkalignH_ng / globalH_ng, src/fwd2h1.cc:1997-2041, 2140-2231, 3267-3286; stdskl3,
// src/gaps.cc:178-227).  No CPU compute path: the DP, its boundary set-up, the end-cell selection
// and the traceback walk all run on the device.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spdp.h"
#include "spdp_internal.h"
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"

#define HIPCHK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

enum { H_POOL = 5 };                         // ctx->pool[] slot group of this path
enum { HP_SC = 0, HP_A, HP_COLS, HP_AUX, HP_PROBS, HP_BND, HP_TB, HP_RES, HP_SKL, HP_NSKL, HP_PACK, HP_OFF };
static const int H_SKL_CAP = 1024;

// ---- geometry --------------------------------------------------------------------------
static void stripe31_rng(int a_left, int a_right, int b_left, int b_right, int sh, SpdpWindow* w)
{   // stripe31(), src/aln2.cc:178-198 (cmode 3)
    if (sh < 0) {
        int shorter = std::min(a_right - a_left, b_right - b_left);
        sh = -sh * shorter / 100;
    }
    sh *= 3;
    w->up = b_right - 3 * a_right;
    w->lw = b_left - 3 * a_left;
    if (w->up < w->lw) std::swap(w->up, w->lw);
    w->up += sh;
    w->lw -= sh;
    int q;
    if ((q = b_right - 3 * a_left) < w->up) w->up = q;
    if ((q = b_left - 3 * a_right) > w->lw) w->lw = q;
    w->width = w->up - w->lw + 7;
}

void spdp_stripe31(const SpdpProblemH* p, int sh, SpdpWindow* w)
{
    stripe31_rng(p->a_left, p->a_right, p->b_left, p->b_right, sh, w);
}

int64_t spdp_cells_h(const SpdpProblemH* p, const SpdpWindow* w)
{   // (aa, nt) cells of the band as the scalar loops bound n, src/fwd2h1.cc:322-330
    int64_t c = 0;
    for (int m = p->a_left + 1; m <= p->a_right; ++m) {
        const int n0 = std::max(3 * m + w->lw - 1, p->b_left);
        const int n9 = std::min(3 * m + w->up, p->b_right);
        if (n9 > n0) c += n9 - n0;
    }
    return c;
}

// ---- resident inputs + work buffers of one set of problems -------------------------------
struct SpdpBatchH {
    SpdpContext* ctx = nullptr;
    SpdpScoringH sc;
    int n = 0;
    std::vector<SpdpProblemH> probs;            // ranges / flags only are used after upload
    std::vector<DevProblemH> h_probs;
    std::vector<int> cls;                        // per problem: 0 run, 1 needs an engine not built, 2 bad input
    std::vector<int> run_idx;                    // dispatch slot -> caller index
    void *d_sc = nullptr, *d_a = nullptr, *d_cols = nullptr, *d_aux = nullptr, *d_probs = nullptr,
         *d_bnd = nullptr, *d_tb = nullptr, *d_res = nullptr, *d_skl = nullptr, *d_nskl = nullptr;
    int64_t cells = 0, tb_elems = 0;
    float sweep_ms = 0.f, walk_ms = 0.f;
    std::string err;
};

static int validate(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* p, int i)
{
    char buf[160];
    auto fail = [&](const char* what) {
        snprintf(buf, sizeof buf, "aa x genome problem %d: %s", i, what);
        ctx->err = buf;
        return -1;
    };
    if (sc->mtx_rows <= 0 || sc->mtx_rows > 31 || sc->mtx_cols <= 0 || sc->mtx_cols > 31) return fail("matrix larger than 31 x 31");
    if (sc->nquant < 1 || sc->nquant > SPDP_MAX_QUANT) return fail("bad nquant");
    if (!p->a || !p->b || !p->sig5 || !p->sig3 || !p->sigS || !p->sigT || !p->sigE || !p->phs5 || !p->phs3)
        return fail("null input array");
    if (p->a_left < 0 || p->a_right > p->a_len || p->a_left > p->a_right) return fail("bad query range");
    if (p->b_left < 0 || p->b_right > p->b_len || p->b_left > p->b_right) return fail("bad genomic range");
    if (p->b_left < p->exin_left || p->b_right > p->exin_right) return fail("active range outside the Exinon range");
    return 0;
}

// the reference's decision ladder up to the engine call (lspH_ng, src/fwd2h1.cc:2140-2175)
static int classify(const SpdpScoringH* sc, const SpdpProblemH* p, const SpdpWindow& w, bool ladder)
{
    const int m = p->a_right - p->a_left, n = p->b_right - p->b_left;
    if (!m || !n) return 2;                                  // empty range: GapPenalty paths, not built
    if (w.width < 0) return 2;
    if (m < 8) return 1;                                     // scalar forwardH_ng
    if (!ladder) return 0;
    if (w.up == w.lw) return 1;                              // diagonalH_ng
    if (std::abs(n - m) < 16 || m == 1 || n <= 3) return 0;
    const float cvol = float(m) * float(n + 3 * m);
    if (2.f * cvol < (float) sc->max_vmf_space) return 0;    // coef_B = sizeof(short)
    return 1;                                                // hirschbergH1_wip
}

static int batch_build(SpdpBatchH* bt, SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n,
                       bool ladder)
{
    bt->ctx = ctx; bt->sc = *sc; bt->n = n;
    bt->probs.assign(probs, probs + n);
    bt->cls.assign(n, 0);
    DevPool& pool = ctx->pool[H_POOL];
    // ---- scoring
    DevScoringH ds;
    memset(&ds, 0, sizeof ds);
    ds.gop = sc->gop; ds.gep = sc->gep; ds.lgep = sc->lgep; ds.codonk1 = sc->codonk1;
    ds.g1 = sc->gapw1; ds.g2 = sc->gapw2; ds.g3 = sc->gapw3;
    ds.spj = sc->spj; ds.llmt = sc->llmt; ds.nquant = sc->nquant; ds.local = sc->local;
    ds.term_codon = sc->term_codon;
    for (int j = 0; j < 8; ++j) { ds.qm_len[j] = sc->qm_len[j]; ds.qm_pen[j] = sc->qm_pen[j]; }
    for (int i = 0; i < sc->mtx_rows; ++i)
        for (int j = 0; j < sc->mtx_cols; ++j) ds.mtx[i * 32 + j] = sc->mtx[i * sc->mtx_cols + j];
    const int ipen = sc->spj ? sc->ipen : SPDH_NEV;
    // ---- problems
    std::vector<uint8_t> a_all;
    std::vector<int4> cols;
    std::vector<short4> aux;
    int64_t bnd_ent = 0, tb_el = 0;
    bt->h_probs.clear(); bt->run_idx.clear(); bt->cells = 0;
    // dispatch order: largest problems first (the hardware hands blocks out in index order, so the
    // long ones start early and the short ones fill the tail)
    std::vector<std::pair<int64_t, int>> todo;
    for (int i = 0; i < n; ++i) {
        const SpdpProblemH& p = probs[i];
        if (validate(ctx, sc, &p, i)) return -1;
        SpdpWindow w;
        stripe31_rng(p.a_left, p.a_right, p.b_left, p.b_right, sc->sh, &w);
        bt->cls[i] = classify(sc, &p, w, ladder);
        if (!bt->cls[i]) todo.emplace_back(-spdp_cells_h(&p, &w), i);
    }
    std::stable_sort(todo.begin(), todo.end());
    for (const auto& td : todo) {
        const int i = td.second;
        const SpdpProblemH& p = probs[i];
        SpdpWindow w;
        stripe31_rng(p.a_left, p.a_right, p.b_left, p.b_right, sc->sh, &w);
        DevProblemH d;
        memset(&d, 0, sizeof d);
        d.a_left = p.a_left; d.a_right = p.a_right; d.b_left = p.b_left; d.b_right = p.b_right;
        d.lw = w.lw; d.up = w.up; d.width = w.width; d.buf_size = w.width + 6 * SPDH_NELEM;
        d.a_exgl = p.a_exgl; d.a_exgr = p.a_exgr; d.b_exgl = p.b_exgl; d.b_exgr = p.b_exgr;
        d.m_width = p.a_right - p.a_left + 1;
        d.n_width = p.b_right - p.b_left + 1 + 3 * d.m_width;
        d.tb_size = (int64_t) d.m_width * d.n_width + 32;
        if (d.tb_size + 64 >= (int64_t) 1 << 31) { ctx->err = "traceback bitmap of one problem exceeds 2^31 cells"; return -1; }
        d.col_len = p.b_len + 3 + SPDH_COL_PAD;
        d.a_off = (int64_t) a_all.size();
        d.col_off = (int64_t) cols.size();
        d.bnd_off = bnd_ent;
        d.tb_off = tb_el;
        d.cells = spdp_cells_h(&p, &w);
        bnd_ent += d.buf_size + SPDH_BND_PAD;
        tb_el += (d.tb_size + 64 + 7) / 8 * 8;
        bt->cells += d.cells;
        a_all.insert(a_all.end(), p.a, p.a + p.a_len);
        // column records (layout: spdp_h_dev.h); positions beyond the inputs read as zero
        const int N = p.b_len + 3;
        auto good = [&](int x) { return p.exin_left - 1 <= x && x < p.exin_right; };
        auto s16at = [&](const int16_t* v, int x) -> int { return (x >= 0 && x < N) ? v[x] : 0; };
        const size_t c0 = cols.size();
        cols.resize(c0 + d.col_len, make_int4(0, 0, 0, 0));
        aux.resize(c0 + d.col_len, make_short4(0, 0, 0, 0));
        for (int x = 0; x < N; ++x) {
            const int cp = (x - 2 >= 0 && good(x - 2)) ? p.sigE[x - 2] : 0;
            const int tron = (x - 2 >= 0 && x - 2 <= p.b_len) ? p.b[x - 2] : 0;
            unsigned fl = 0;
            int s3_0 = SPDH_MIN_SSV, s3_1 = SPDH_MIN_SSV, s5_0 = SPDH_MIN_SSV, s5_1 = SPDH_MIN_SSV;
            // candidates of fwd2h1_wip_simd.h:214-222 / 262-270: phase -1 / 0 / +1, and +1 again when phs == 2
            const int ph3 = p.phs3[x], ph5 = p.phs5[x];
            if (ph3 > -2) {
                const int phase = (ph3 == 2) ? -1 : ph3;
                fl |= (unsigned) (phase + 2);
                s3_0 = s16at(p.sig3, x - phase);
                if (ph3 == 2) { fl |= 4u; s3_1 = s16at(p.sig3, x - 1); }
            }
            if (ph5 > -2) {
                const int phase = (ph5 == 2) ? -1 : ph5;
                fl |= (unsigned) (phase + 2) << 3;
                s5_0 = (int16_t) (s16at(p.sig5, x - phase) + ipen);
                if (ph5 == 2) { fl |= 32u; s5_1 = (int16_t) (s16at(p.sig5, x - 1) + ipen); }
            }
            int4 rec;
            rec.x = (int) ((unsigned) (uint16_t) (int16_t) cp | ((unsigned) (tron > 31 ? SPDH_ZCODE : tron) << 16) | (fl << 24));
            rec.y = (int) ((unsigned) (uint16_t) (int16_t) s3_0 | ((unsigned) (uint16_t) (int16_t) s3_1 << 16));
            rec.z = (int) ((unsigned) (uint16_t) (int16_t) s5_0 | ((unsigned) (uint16_t) (int16_t) s5_1 << 16));
            rec.w = 0;
            cols[c0 + x] = rec;
            aux[c0 + x] = make_short4(p.sigS[x], p.sigT[x], p.sigE[x], p.sig5[x]);
        }
        bt->h_probs.push_back(d);
        bt->run_idx.push_back(i);
    }
    const int nr = (int) bt->h_probs.size();
    bt->tb_elems = tb_el;
    if (!nr) return 0;
    // ---- upload
    bt->d_sc = pool.get(HP_SC, sizeof ds);
    bt->d_a = pool.get(HP_A, a_all.size() + 16);
    bt->d_cols = pool.get(HP_COLS, cols.size() * sizeof(int4));
    bt->d_aux = pool.get(HP_AUX, aux.size() * sizeof(short4));
    bt->d_probs = pool.get(HP_PROBS, nr * sizeof(DevProblemH));
    bt->d_bnd = pool.get(HP_BND, (size_t) bnd_ent * sizeof(int2));
    bt->d_tb = pool.get(HP_TB, (size_t) tb_el * sizeof(uint16_t));
    bt->d_res = pool.get(HP_RES, nr * sizeof(DevResultH));
    bt->d_skl = pool.get(HP_SKL, (size_t) nr * H_SKL_CAP * sizeof(int2));
    bt->d_nskl = pool.get(HP_NSKL, nr * sizeof(int));
    if (!bt->d_sc || !bt->d_a || !bt->d_cols || !bt->d_aux || !bt->d_probs || !bt->d_bnd || !bt->d_tb ||
        !bt->d_res || !bt->d_skl || !bt->d_nskl) {
        ctx->err = "device allocation failed (aa x genome batch; traceback bitmaps need " +
                   std::to_string((size_t) tb_el * 2 >> 20) + " MiB)";
        return -1;
    }
    HIPCHK(hipMemcpyAsync(bt->d_sc, &ds, sizeof ds, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(bt->d_a, a_all.data(), a_all.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(bt->d_cols, cols.data(), cols.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(bt->d_aux, aux.data(), aux.size() * sizeof(short4), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(bt->d_probs, bt->h_probs.data(), nr * sizeof(DevProblemH), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// one pass: sweep (+ walk); results to host
static int batch_run(SpdpBatchH* bt, bool walk, std::vector<DevResultH>& res, std::vector<int>& n_skl,
                     std::vector<SpdpSkl>& skl)
{
    SpdpContext* ctx = bt->ctx;
    const int nr = (int) bt->h_probs.size();
    res.clear(); n_skl.clear(); skl.clear();
    bt->sweep_ms = bt->walk_ms = 0.f;
    if (!nr) return 0;
    HSweepArgs A;
    A.sc = (const DevScoringH*) bt->d_sc; A.probs = (const DevProblemH*) bt->d_probs; A.n_probs = nr;
    A.a_codes = (const uint8_t*) bt->d_a; A.cols = (const int4*) bt->d_cols; A.aux = (const short4*) bt->d_aux;
    A.bnd = (int2*) bt->d_bnd; A.tb = (uint16_t*) bt->d_tb; A.res = (DevResultH*) bt->d_res;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    const int pen_cap = bt->sc.nquant > 1 ? bt->sc.qm_len[bt->sc.nquant - 2] + 1 : 0;
    HIPCHK(spdh_launch_sweep(&A, bt->sc.spj, pen_cap, bt->sc.local, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (walk) {
        HWalkArgs W;
        W.probs = A.probs; W.n_probs = nr; W.tb = A.tb; W.res = A.res;
        W.skl = (int2*) bt->d_skl; W.n_skl = (int*) bt->d_nskl; W.skl_cap = H_SKL_CAP;
        HIPCHK(spdh_launch_walk(&W, ctx->stream));
    }
    res.resize(nr);
    HIPCHK(hipMemcpyAsync(res.data(), bt->d_res, nr * sizeof(DevResultH), hipMemcpyDeviceToHost, ctx->stream));
    if (walk) {
        n_skl.resize(nr);
        HIPCHK(hipMemcpyAsync(n_skl.data(), bt->d_nskl, nr * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&bt->sweep_ms, ctx->ev0, ctx->ev1));
    if (walk) {
        // compact copy of the records actually written
        DevPool& pool = ctx->pool[H_POOL];
        std::vector<int64_t> off(nr + 1, 0);
        for (int i = 0; i < nr; ++i) {
            int c = n_skl[i];
            if (c == -3) c = 1;                              // the single start record
            if (c < 0) c = 0;
            off[i + 1] = off[i] + c;
        }
        skl.resize(off[nr]);
        if (off[nr]) {
            void* d_off = pool.get(HP_OFF, (nr + 1) * sizeof(int64_t));
            void* d_pack = pool.get(HP_PACK, off[nr] * sizeof(int2));
            std::vector<int> cnt(nr);
            for (int i = 0; i < nr; ++i) cnt[i] = (int) (off[i + 1] - off[i]);
            HIPCHK(hipMemcpyAsync(d_off, off.data(), (nr + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            // n_skl on the device still holds the status codes: hand the counts over instead
            HIPCHK(hipMemcpyAsync(bt->d_nskl, cnt.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(spdp_launch_pack((const int2*) bt->d_skl, H_SKL_CAP, (const int*) bt->d_nskl, (const int64_t*) d_off,
                                    (int2*) d_pack, nr, ctx->stream));
            HIPCHK(hipMemcpyAsync(skl.data(), d_pack, off[nr] * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
    }
    return 0;
}

// stdskl3, src/gaps.cc:178-227 (UNITE_INDEL_FS = 0): records in any order -> corner list start -> end
static void std_skl3(std::vector<SpdpSkl>& rec, std::vector<SpdpSkl>& out)
{
    out.clear();
    if (rec.size() < 2) { out = rec; return; }
    std::sort(rec.begin(), rec.end(), [](const SpdpSkl& x, const SpdpSkl& y) {
        return x.m != y.m ? x.m < y.m : x.n < y.n;
    });
    int pr = -2;
    const SpdpSkl* prv = &rec[0];
    for (size_t i = 1; i < rec.size(); ++i) {
        const SpdpSkl* org = &rec[i];
        const int dm = (org->m - prv->m) * 3;
        const int dn = org->n - prv->n;
        if (!dm && !dn) continue;
        if (dn < 0) continue;
        int dd = std::min(dm, dn);
        const int df = dn - dm;
        const int dr = df ? (df > 0 ? 1 : -1) : 0;
        if (dd && df) {
            if (pr) out.push_back(*prv);
            SpdpSkl b;
            b.n = prv->n + dd;
            if (df < 0 && df % 3) dd += 2;
            b.m = prv->m + dd / 3;
            out.push_back(b);
            if (df > 0 && df % 3) { b.n += df % 3; out.push_back(b); }
        } else if (dr != pr || !dm)
            out.push_back(*prv);
        pr = dr;
        prv = org;
    }
    out.push_back(*prv);
}

static SpdpSkl* dup_skl(const std::vector<SpdpSkl>& v)
{
    if (v.empty()) return nullptr;
    SpdpSkl* p = (SpdpSkl*) malloc(v.size() * sizeof(SpdpSkl));
    memcpy(p, v.data(), v.size() * sizeof(SpdpSkl));
    return p;
}

// level 0: raw engine output; level 1: alignH_ng (header + stdskl3)
static int batch_deliver(SpdpBatchH* bt, int level, SpdpAlignment* out)
{
    std::vector<DevResultH> res;
    std::vector<int> n_skl;
    std::vector<SpdpSkl> skl;
    if (batch_run(bt, true, res, n_skl, skl)) return -1;
    int rc = 0;
    for (int i = 0; i < bt->n; ++i) {
        if (out) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; }
        if (bt->cls[i]) rc = 1;
    }
    int64_t off = 0;
    std::vector<SpdpSkl> rec, stdv, full;
    for (size_t s = 0; s < bt->run_idx.size(); ++s) {
        const int i = bt->run_idx[s];
        const int st = n_skl[s];
        const int cnt = st == -3 ? 1 : (st < 0 ? 0 : st);
        if (st == -1) { bt->ctx->err = "traceback record buffer overflow"; return -1; }
        if (out) {
            out[i].score = res[s].score;
            if (st == -2) out[i].n_skl = -1;                 // the reference's fatal "Unexpected dir"
            else if (st == -3) out[i].n_skl = -2;            // its traceback starts outside the bitmap
            else if (level == 0) {
                rec.assign(skl.begin() + off, skl.begin() + off + cnt);
                out[i].n_skl = cnt;
                out[i].skl = dup_skl(rec);
            } else {
                rec.assign(skl.begin() + off, skl.begin() + off + cnt);
                if (cnt >= 2) {                              // globalH_ng: fewer than 2 records = no alignment
                    std_skl3(rec, stdv);
                    full.clear();
                    SpdpSkl hd; hd.m = 1; hd.n = (int) stdv.size();
                    full.push_back(hd);
                    full.insert(full.end(), stdv.begin(), stdv.end());
                    out[i].n_skl = (int) full.size();
                    out[i].skl = dup_skl(full);
                }
            }
        }
        off += cnt;
    }
    return rc;
}

// ---- C ABI ------------------------------------------------------------------------------
int spdp_wip_forward_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                       SpdpAlignment* out)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !out) return -1;
    SpdpBatchH bt;
    if (batch_build(&bt, ctx, sc, probs, n_probs, false)) return -1;
    return batch_deliver(&bt, 0, out);
}

int spdp_homscore_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                    int32_t* scores)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !scores) return -1;
    SpdpBatchH bt;
    if (batch_build(&bt, ctx, sc, probs, n_probs, false)) return -1;
    std::vector<DevResultH> res;
    std::vector<int> n_skl;
    std::vector<SpdpSkl> skl;
    if (batch_run(&bt, false, res, n_skl, skl)) return -1;
    int rc = 0;
    for (int i = 0; i < n_probs; ++i) { scores[i] = SPDP_NEVSEL; if (bt.cls[i]) rc = 1; }
    for (size_t s = 0; s < bt.run_idx.size(); ++s) scores[bt.run_idx[s]] = res[s].score;
    return rc;
}

int spdp_align_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                 SpdpAlignment* out)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !out) return -1;
    SpdpBatchH bt;
    if (batch_build(&bt, ctx, sc, probs, n_probs, true)) return -1;
    return batch_deliver(&bt, 1, out);
}

SpdpBatchH* spdp_batch_upload_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs)
{
    if (!ctx || !sc || !probs || n_probs < 0) return nullptr;
    SpdpBatchH* bt = new SpdpBatchH();
    if (batch_build(bt, ctx, sc, probs, n_probs, true)) { delete bt; return nullptr; }
    return bt;
}
void spdp_batch_free_h(SpdpBatchH* bt) { delete bt; }
int64_t spdp_batch_cells_h(const SpdpBatchH* bt) { return bt ? bt->cells : 0; }

int spdp_batch_align_h(SpdpBatchH* bt, SpdpAlignment* out, float* kernel_ms, int64_t* kernel_cells)
{
    if (!bt) return -1;
    const int rc = batch_deliver(bt, 1, out);
    if (kernel_ms) *kernel_ms = bt->sweep_ms;
    if (kernel_cells) *kernel_cells = bt->cells;
    return rc;
}
