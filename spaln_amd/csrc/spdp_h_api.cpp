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
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/spdp.h"
#include "spdp_internal.h"
#include "spdp_h_dev.h"
#include "spdp_h_internal.h"
#include "spdp_h_requests.h"

#define HIPCHK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

enum { H_POOL = 5, HU_POOL = 6 };           // ctx->pool[] slot groups: inputs + traceback runs / linear-space runs
enum { HP_SC = 0, HP_A, HP_COLS, HP_AUX, HP_PROBS, HP_BND, HP_TB, HP_RES, HP_SKL, HP_NSKL, HP_PACK, HP_OFF, HP_INTPEN, HP_PIPE };
#include "spdp_gencode.h"
int spdh_signals_run(SpdpContext* ctx, const SpdpSignalModelH* m, const std::vector<SigJobH>& jobs, SignalArgsH args, int pack);   // spdp_signals_api.cpp
enum { HU_PROBS = 0, HU_BND, HU_IMD, HU_RES, HU_CPOS, HU_RANGES, HU_SCORES, HU_PIPE };
static const int H_SKL_CAP = 4096;       // slot of one traceback record list; a list is at most ~4 records per query row (diagonal / gap corners and
                                           // two per intron), and the slot is min(this, rows + columns + 8): protein queries stay far below

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

// ---- resident inputs of a set of parent problems ------------------------------------------
struct HStore {
    SpdpContext* ctx = nullptr;
    SpdpScoringH sc;
    int n = 0;
    std::vector<SpdpProblemH> probs;            // ranges / flags / lengths are used after upload
    std::vector<int64_t> a_off, col_off;
    std::vector<int32_t> col_len;
    void *d_sc = nullptr, *d_a = nullptr, *d_cols = nullptr, *d_aux = nullptr, *d_intpen = nullptr;
    bool ipen_runs_ok = false;
    std::vector<std::vector<int16_t>> own_sigE; // device-made signals: the host ladder still reads sigE (diagonalH_ng)
    void* d_cip = nullptr;                      // owned (hipMalloc): conserved-intron bonuses, SpdpProblemH::cip
    std::vector<int32_t> cip_off;               // per problem: first entry of its row in d_cip, -1 = none
    ~HStore() { if (d_cip) (void) hipFree(d_cip); }
    bool scalar_ok = false;                     // inputs of the scalar engine present (intpen / t53, dinc)
    int upload(SpdpContext* c, const SpdpScoringH* sc, const SpdpProblemH* probs, int n);
};

// the context whose streams / pools / scratch a call on `st` uses: the one the inputs were uploaded through, or -- for
// a request batch another dispatcher thread of the seeded path runs beside it (spdh_run_requests) -- that thread's lane
static thread_local SpdpContext* t_lane = nullptr;
static inline SpdpContext* lane_of(const HStore& st) { return t_lane ? t_lane : st.ctx; }

// one engine call on (a sub-range of) a parent problem
struct HItem {
    int top;                                    // caller index the records belong to
    int a_left, a_right, b_left, b_right;
    int a_exgl, a_exgr, b_exgl, b_exgr;
    SpdpWindow w;
    int n_im = 0, imd_intvl = 0;
    bool recursive = false, first = false;      // first: this call's return value is gsi->scr
    bool exact = false;                         // traceback by the -A1 engine (forwardH1)
    int slot = -1;                              // result slot (the caller's problem, or the request)
    bool nospj = false;                         // trcbkalignH_ng(wdw, spj = false): the scalar engine runs without introns
    int cut_l = 0, cut_r = 0;                   // cut_r > cut_l: forwardH_ng jumps over genomic columns (cut_l, cut_r]
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
    const int n_arr = (p->sig5 ? 1 : 0) + (p->sig3 ? 1 : 0) + (p->sigS ? 1 : 0) + (p->sigT ? 1 : 0) + (p->sigE ? 1 : 0) +
                      (p->phs5 ? 1 : 0) + (p->phs3 ? 1 : 0);
    if (!p->a || !p->b || (n_arr != 7 && !(n_arr == 0 && sc->sigmodel)))
        return fail("null input array (all seven signal arrays, or none of them together with SpdpScoringH::sigmodel)");
    if (p->a_left < 0 || p->a_right > p->a_len || p->a_left > p->a_right) return fail("bad query range");
    if (p->b_left < 0 || p->b_right > p->b_len || p->b_left > p->b_right) return fail("bad genomic range");
    if (p->b_left < p->exin_left || p->b_right > p->exin_right) return fail("active range outside the Exinon range");
    return 0;
}

int HStore::upload(SpdpContext* c, const SpdpScoringH* scp, const SpdpProblemH* pr, int cnt)
{
    ctx = c; sc = *scp; n = cnt;
    probs.assign(pr, pr + cnt);
    DevPool& pool = ctx->pool[H_POOL];
    DevScoringH ds;
    memset(&ds, 0, sizeof ds);
    ds.gop = sc.gop; ds.gep = sc.gep; ds.lgep = sc.lgep; ds.codonk1 = sc.codonk1;
    ds.g1 = sc.gapw1; ds.g2 = sc.gapw2; ds.g3 = sc.gapw3;
    ds.spj = sc.spj; ds.llmt = sc.llmt; ds.nquant = sc.nquant; ds.local = sc.local;
    ds.term_codon = sc.term_codon;
    for (int j = 0; j < 8; ++j) { ds.qm_len[j] = sc.qm_len[j]; ds.qm_pen[j] = sc.qm_pen[j]; }
    for (int i = 0; i < sc.mtx_rows; ++i)
        for (int j = 0; j < sc.mtx_cols; ++j) ds.mtx[i * 32 + j] = sc.mtx[i * sc.mtx_cols + j];
    const int ipen = sc.spj ? sc.ipen : SPDH_NEV;
    a_off.resize(n); col_off.resize(n); col_len.resize(n);
    int n_dev = 0;
    for (int i = 0; i < n; ++i) {
        if (validate(ctx, &sc, &probs[i], i)) return -1;
        n_dev += probs[i].sig5 ? 0 : 1;
    }
    if (n_dev && n_dev != n) { ctx->err = "signal arrays missing for part of the batch"; return -1; }
    const bool dev_sig = n_dev > 0;
    int64_t a_tot = 0, c_tot = 0;
    for (int i = 0; i < n; ++i) {
        a_off[i] = a_tot; a_tot += (int64_t) probs[i].a_len + 1;
        col_off[i] = c_tot; col_len[i] = probs[i].b_len + 3 + SPDH_COL_PAD; c_tot += col_len[i];
    }
    std::vector<uint8_t> a_all((size_t) a_tot, 0);      // (the byte behind a query reads 0 in the reference process: row a_len, see bad_range)
    // column records, packed by the host's cores into pinned staging memory the context keeps (one thread and pageable
    // vectors took 0.18 ms per protein window -- longer than the reference takes to ALIGN the pair on its seeded path)
    // ... through a ring of four groups' worth of it: a group's slot is packed again once its copy has left (a map + align call of
    // 20 000 protein loci holds 5 * 10^8 positions -- 12 GB of pinned memory when every group had its own place, two seconds of a
    // context's first call to allocate)
    std::vector<int> grp_first, grp_of(n);
    int64_t slot_cap = 1;
    {
        int64_t acc = 0;
        for (int i = 0; i < n; ++i) {
            if (i == 0 || acc >= (4 << 20)) { grp_first.push_back(i); acc = 0; }
            grp_of[i] = (int) grp_first.size() - 1;
            acc += col_len[i];
            slot_cap = std::max(slot_cap, acc);
        }
    }
    const int RING_SLOTS = (getenv("SPDP_UPLOAD_RING") && atoi(getenv("SPDP_UPLOAD_RING")) == 0) ? std::max<int>(1, (int) grp_first.size()) : 4;     // (0: every group its own place, as before round 6)
    int4* cols = dev_sig ? nullptr : (int4*) ctx->staging(0, (size_t) slot_cap * RING_SLOTS * sizeof(int4));
    short4* aux = dev_sig ? nullptr : (short4*) ctx->staging(1, (size_t) slot_cap * RING_SLOTS * sizeof(short4));
    if (!dev_sig && (!cols || !aux)) { ctx->err = "out of pinned host memory"; return -1; }
    auto stage_of = [&](int i) -> int64_t {             // where problem i's records lie in the staging ring
        const int g = grp_of[i];
        return (int64_t) (g % RING_SLOTS) * slot_cap + (col_off[i] - col_off[grp_first[g]]);
    };
    auto pack_one = [&](int i) {
        const SpdpProblemH& p = probs[i];
        memcpy(a_all.data() + a_off[i], p.a, p.a_len);
        if (dev_sig) return;
        // column records (layout: spdp_h_dev.h); positions beyond the inputs read as zero
        const int N = p.b_len + 3;
        auto good = [&](int x) { return p.exin_left - 1 <= x && x < p.exin_right; };
        auto s16at = [&](const int16_t* v, int x) -> int { return (x >= 0 && x < N) ? v[x] : 0; };
        const size_t c0 = (size_t) stage_of(i);
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
            rec.w = (p.dinc && x <= p.b_len) ? (int) p.dinc[x] : 0;     // b_len + 1 entries
            cols[c0 + x] = rec;
            aux[c0 + x] = make_short4(p.sigS[x], p.sigT[x], p.sigE[x], p.sig5[x]);
        }
        for (int x = N; x < col_len[i]; ++x) { cols[c0 + x] = make_int4(0, 0, 0, 0); aux[c0 + x] = make_short4(0, 0, 0, 0); }
    };
    if (n && !dev_sig) {
        d_cols = pool.get(HP_COLS, (size_t) c_tot * sizeof(int4));
        d_aux = pool.get(HP_AUX, (size_t) c_tot * sizeof(short4));
        if (!d_cols || !d_aux) { ctx->err = "device allocation failed (aa x genome inputs)"; return -1; }
    }
    {
        int n_thr = spdp_host_cpus();
        if (const char* e = getenv("SPDP_UPLOAD_THREADS")) n_thr = atoi(e);
        n_thr = std::max(1, std::min(std::min(n_thr, 32), n));
        // the copy of a group of problems (~4 M positions) starts as soon as the group is packed, under the packing of the next ones;
        // a packer that reaches a group whose slot still holds an earlier group waits for that group's copy
        const int n_grp = (int) grp_first.size();
        std::vector<std::atomic<int>> grp_done(std::max(1, n_grp));
        for (auto& g : grp_done) g.store(0);
        std::atomic<int> next_prob{0};
        std::atomic<int> copied{0};                     // groups whose copy has left the staging (or: give up, INT_MAX)
        auto pack = [&]() {
            for (;;) {
                const int i = next_prob.fetch_add(1);
                if (i >= n) break;
                if (!dev_sig) while (copied.load(std::memory_order_acquire) < grp_of[i] - RING_SLOTS + 1) std::this_thread::yield();
                pack_one(i);
                grp_done[grp_of[i]].fetch_add(1, std::memory_order_release);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t) th.emplace_back(pack);
        hipError_t ce = hipSuccess;
        std::vector<hipEvent_t> left((size_t) RING_SLOTS, nullptr);
        for (int k = 0; k < RING_SLOTS && !dev_sig && ce == hipSuccess; ++k) ce = hipEventCreateWithFlags(&left[k], hipEventDisableTiming);
        int synced = 0;                                 // groups known to have left
        for (int g = 0; g < n_grp && !dev_sig && ce == hipSuccess; ++g) {
            const int first = grp_first[g], last = g + 1 < n_grp ? grp_first[g + 1] : n;
            while (grp_done[g].load(std::memory_order_acquire) < last - first) std::this_thread::yield();
            const int64_t c0 = col_off[first], c1 = last < n ? col_off[last] : c_tot;
            const int64_t s0 = (int64_t) (g % RING_SLOTS) * slot_cap;
            ce = hipMemcpyAsync((int4*) d_cols + c0, cols + s0, (size_t) (c1 - c0) * sizeof(int4), hipMemcpyHostToDevice, ctx->stream);
            if (ce == hipSuccess)
                ce = hipMemcpyAsync((short4*) d_aux + c0, aux + s0, (size_t) (c1 - c0) * sizeof(short4), hipMemcpyHostToDevice, ctx->stream);
            if (ce == hipSuccess) ce = hipEventRecord(left[g % RING_SLOTS], ctx->stream);
            // the packers may be two groups ahead of the copies: the group before this one has to have left before the next is issued
            if (ce == hipSuccess && g >= 1) { ce = hipEventSynchronize(left[(g - 1) % RING_SLOTS]); synced = g; copied.store(g, std::memory_order_release); }
        }
        (void) synced;
        copied.store(INT32_MAX, std::memory_order_release);                  // (on an error too: no packer may wait for ever)
        for (std::thread& t : th) t.join();
        if (!dev_sig) (void) hipStreamSynchronize(ctx->stream);              // the last copies still read the staging
        for (int k = 0; k < RING_SLOTS; ++k) if (left[k]) (void) hipEventDestroy(left[k]);
        HIPCHK(ce);
    }
    scalar_ok = sc.intpen && sc.intpen_len > 0;
    for (int i = 0; i < n && !dev_sig; ++i) if (!probs[i].dinc) scalar_ok = false;      // (device-made signals bring dinc along)
    // double affine gaps (PwdB::Noll = 3, -yl3): built for the -A0 engines (forwardH_ng / hirschbergH_ng, spdh_rowwave<., ., ., true>);
    // the `_wip` and -A1 families have no well-defined form of it in the reference
    if (sc.noll != 0 && sc.noll != 2 && !(sc.noll == 3 && sc.scalar_engines == 1)) {
        ctx->err = "double affine gaps (SpdpScoringH.noll = 3) are built for the -A0 engines (scalar_engines = 1); noll must be 2 or 3";
        return -1;
    }
    if (!n) return 0;
    if (scalar_ok) {
        // the table, and behind it its steps beyond the part the kernels keep in LDS (spdp_ipen_runs.h)
        d_intpen = pool.get(HP_INTPEN, ((size_t) sc.intpen_len + SPDP_IPR_WORDS) * sizeof(int16_t));
        if (!d_intpen) { ctx->err = "device allocation failed (intron penalty table)"; return -1; }
        HIPCHK(hipMemcpyAsync(d_intpen, sc.intpen, (size_t) sc.intpen_len * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
        std::vector<int16_t> runs(SPDP_IPR_WORDS);
        ipen_runs_ok = spdp_intpen_runs(sc.intpen, sc.intpen_len, runs.data());
        if (ipen_runs_ok)
            HIPCHK(hipMemcpy((int16_t*) d_intpen + sc.intpen_len, runs.data(), runs.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    cip_off.assign(n, -1);
    {
        std::vector<int32_t> hcip;
        for (int i = 0; i < n; ++i)
            if (probs[i].cip) {
                cip_off[i] = (int32_t) hcip.size();
                hcip.insert(hcip.end(), probs[i].cip, probs[i].cip + 3 * probs[i].a_len + 2);
            }
        if (!hcip.empty()) {
            HIPCHK(hipMalloc(&d_cip, hcip.size() * sizeof(int32_t)));
            HIPCHK(hipMemcpy(d_cip, hcip.data(), hcip.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    d_sc = pool.get(HP_SC, sizeof ds);
    d_a = pool.get(HP_A, a_all.size() + 16);
    if (dev_sig) {
        d_cols = pool.get(HP_COLS, (size_t) c_tot * sizeof(int4));
        d_aux = pool.get(HP_AUX, (size_t) c_tot * sizeof(short4));
    }
    if (!d_sc || !d_a || !d_cols || !d_aux) { ctx->err = "device allocation failed (aa x genome inputs)"; return -1; }
    HIPCHK(hipMemcpyAsync(d_sc, &ds, sizeof ds, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_a, a_all.data(), a_all.size(), hipMemcpyHostToDevice, ctx->stream));
    if (dev_sig) {                              // (made on the device below: only zeros for the padding now)
        HIPCHK(hipMemsetAsync(d_cols, 0, (size_t) c_tot * sizeof(int4), ctx->stream));
        HIPCHK(hipMemsetAsync(d_aux, 0, (size_t) c_tot * sizeof(short4), ctx->stream));
    }                                           // (else: copied group by group above)
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (dev_sig) {
        // only the tron codes crossed PCIe (1 B per position instead of 24): spdp_signals_h.hip writes the column records
        const size_t tot = (size_t) c_tot;
        std::vector<uint8_t> hb(tot + 16, 0);
        std::vector<SigJobH> jobs(n);
        for (int i = 0; i < n; ++i) {
            memcpy(hb.data() + col_off[i], probs[i].b, (size_t) probs[i].b_len + 1);
            SigJobH& J = jobs[i];
            J.b_off = J.out_off = J.col_off = col_off[i]; J.pad = 0;
            J.b_len = probs[i].b_len; J.left = probs[i].exin_left; J.right = probs[i].exin_right;
        }
        void* tmp[10] = {nullptr};
        const size_t bytes[10] = {tot + 16, 2 * tot, 2 * tot, 2 * tot, 2 * tot, 2 * tot, tot, tot, tot, tot};
        struct Freer { void** p; ~Freer() { for (int k = 0; k < 10; ++k) if (p[k]) (void) hipFree(p[k]); } } freer{tmp};
        for (int k = 0; k < 10; ++k) HIPCHK(hipMalloc(&tmp[k], std::max<size_t>(bytes[k], 16)));
        HIPCHK(hipMemcpy(tmp[0], hb.data(), tot + 16, hipMemcpyHostToDevice));
        SignalArgsH A;
        memset(&A, 0, sizeof A);
        A.codes = (const uint8_t*) tmp[0];
        A.sig5 = (int16_t*) tmp[1]; A.sig3 = (int16_t*) tmp[2]; A.sigS = (int16_t*) tmp[3]; A.sigT = (int16_t*) tmp[4];
        A.sigE = (int16_t*) tmp[5]; A.phs5 = (int8_t*) tmp[6]; A.phs3 = (int8_t*) tmp[7];
        A.cano = (uint8_t*) tmp[8]; A.dinc = (uint8_t*) tmp[9];
        A.cols = (int4*) d_cols; A.aux = (short4*) d_aux; A.ipen = ipen;
        if (spdh_signals_run(ctx, sc.sigmodel, jobs, A, 1)) return -1;
        // the host ladder scores pure diagonals itself (diagonalH_ng) and reads sigE for that: bring it back
        std::vector<int16_t> all(tot);
        HIPCHK(hipMemcpy(all.data(), tmp[5], 2 * tot, hipMemcpyDeviceToHost));
        own_sigE.resize(n);
        for (int i = 0; i < n; ++i) {
            own_sigE[i].assign(all.begin() + col_off[i], all.begin() + col_off[i] + probs[i].b_len + 3);
            probs[i].sigE = own_sigE[i].data();
        }
    }
    sc.sigmodel = nullptr;                      // the caller's model is not ours to keep
    return 0;
}

static HItem item_of(const SpdpProblemH& p, int top, int sh)
{
    HItem it;
    it.top = it.slot = top;
    it.a_left = p.a_left; it.a_right = p.a_right; it.b_left = p.b_left; it.b_right = p.b_right;
    it.a_exgl = p.a_exgl; it.a_exgr = p.a_exgr; it.b_exgl = p.b_exgl; it.b_exgr = p.b_exgr;
    stripe31_rng(p.a_left, p.a_right, p.b_left, p.b_right, sh, &it.w);
    return it;
}

static int64_t cells_of(const HItem& it)
{
    SpdpProblemH q;
    memset(&q, 0, sizeof q);
    q.a_left = it.a_left; q.a_right = it.a_right; q.b_left = it.b_left; q.b_right = it.b_right;
    return spdp_cells_h(&q, &it.w);
}

// descriptors shared by both engines
static void fill_desc(const HStore& st, const HItem& it, DevProblemH& d)
{
    memset(&d, 0, sizeof d);
    d.a_left = it.a_left; d.a_right = it.a_right; d.b_left = it.b_left; d.b_right = it.b_right;
    d.cut_l = it.cut_l; d.cut_len = std::max(0, it.cut_r - it.cut_l); d.nospj = it.nospj ? 1 : 0;
    d.lw = it.w.lw; d.up = it.w.up; d.width = it.w.width - d.cut_len; d.buf_size = it.w.width + 6 * SPDH_NELEM;
    d.a_exgl = it.a_exgl; d.a_exgr = it.a_exgr; d.b_exgl = it.b_exgl; d.b_exgr = it.b_exgr;
    d.m_width = it.a_right - it.a_left + 1;
    d.n_width = it.b_right - it.b_left + 1 + 3 * d.m_width;
    d.tb_size = (int64_t) d.m_width * d.n_width + 32;
    d.col_len = st.col_len[it.top];
    d.a_off = st.a_off[it.top];
    d.col_off = st.col_off[it.top];
    d.n_im = it.n_im; d.imd_intvl = it.imd_intvl;
    d.a_len = st.probs[it.top].a_len; d.b_len = st.probs[it.top].b_len;
    d.cip_off = st.cip_off.empty() ? -1 : st.cip_off[it.top];
    d.a_pad = st.probs[it.top].a_pad;
    d.cells = cells_of(it);
}

// ---- forwardH1_wip over a list of items -----------------------------------------------------
struct HFwdOut {
    std::vector<DevResultH> res;                // in item order
    std::vector<int> n_skl;                     // records; -2 "Unexpected dir", -3 start outside the bitmap
    std::vector<int64_t> off;
    std::vector<SpdpSkl> skl;
    float sweep_ms = 0.f;
    int64_t cells = 0, tb_elems = 0;
};

static int run_forward_group(HStore& st, const std::vector<HItem>& items, bool walk, HFwdOut& out)
{
    SpdpContext* ctx = lane_of(st);
    DevPool& pool = ctx->pool[H_POOL];
    const int nr = (int) items.size();
    out = HFwdOut();
    if (!nr) return 0;
    // dispatch order: largest problems first (the hardware hands blocks out in index order, so the
    // long ones start early and the short ones fill the tail)
    std::vector<std::pair<int64_t, int>> order(nr);
    std::vector<DevProblemH> descs(nr);
    for (int i = 0; i < nr; ++i) { fill_desc(st, items[i], descs[i]); order[i] = {-descs[i].cells, i}; }
    std::stable_sort(order.begin(), order.end());
    std::vector<DevProblemH> h_probs(nr);
    int64_t bnd_ent = 0, tb_el = 0;
    for (int s = 0; s < nr; ++s) {
        DevProblemH d = descs[order[s].second];
        if (d.tb_size + 64 >= (int64_t) 1 << 31) { ctx->err = "traceback bitmap of one problem exceeds 2^31 cells"; return -1; }
        d.bnd_off = bnd_ent;
        d.tb_off = tb_el;
        bnd_ent += d.buf_size + SPDH_BND_PAD;
        tb_el += (d.tb_size + 64 + 7) / 8 * 8;
        out.cells += d.cells;
        h_probs[s] = d;
    }
    out.tb_elems = tb_el;
    void* d_probs = pool.get(HP_PROBS, nr * sizeof(DevProblemH));
    void* d_bnd = pool.get(HP_BND, (size_t) bnd_ent * sizeof(int2));
    void* d_tb = pool.get(HP_TB, (size_t) tb_el * sizeof(uint16_t));
    void* d_res = pool.get(HP_RES, nr * sizeof(DevResultH));
    void* d_skl = pool.get(HP_SKL, (size_t) nr * H_SKL_CAP * sizeof(int2));
    void* d_nskl = pool.get(HP_NSKL, nr * sizeof(int));
    if (!d_probs || !d_bnd || !d_tb || !d_res || !d_skl || !d_nskl) {
        ctx->err = "device allocation failed (aa x genome traceback run: bitmaps need " +
                   std::to_string((size_t) tb_el * 2 >> 20) + " MiB)";
        return -1;
    }
    HIPCHK(hipMemcpyAsync(d_probs, h_probs.data(), nr * sizeof(DevProblemH), hipMemcpyHostToDevice, ctx->stream));
    HSweepArgs A;
    A.sc = (const DevScoringH*) st.d_sc; A.probs = (const DevProblemH*) d_probs; A.n_probs = nr;
    A.a_codes = (const uint8_t*) st.d_a; A.cols = (const int4*) st.d_cols; A.aux = (const short4*) st.d_aux;
    A.bnd = (int2*) d_bnd; A.tb = (uint16_t*) d_tb; A.res = (DevResultH*) d_res;
    const int pen_cap = st.sc.nquant > 1 ? st.sc.qm_len[st.sc.nquant - 2] + 1 : 0;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(spdh_launch_sweep(&A, st.sc.spj, pen_cap, st.sc.local, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (walk) {
        HWalkArgs W;
        W.probs = A.probs; W.n_probs = nr; W.tb = A.tb; W.res = A.res;
        W.skl = (int2*) d_skl; W.n_skl = (int*) d_nskl; W.skl_cap = H_SKL_CAP;
        HIPCHK(spdh_launch_walk(&W, ctx->stream));
    }
    std::vector<DevResultH> res(nr);
    std::vector<int> n_skl(nr, 0);
    HIPCHK(hipMemcpyAsync(res.data(), d_res, nr * sizeof(DevResultH), hipMemcpyDeviceToHost, ctx->stream));
    if (walk) HIPCHK(hipMemcpyAsync(n_skl.data(), d_nskl, nr * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&out.sweep_ms, ctx->ev0, ctx->ev1));
    std::vector<int64_t> off(nr + 1, 0);
    std::vector<SpdpSkl> skl;
    if (walk) {
        for (int s = 0; s < nr; ++s) {
            int c = n_skl[s];
            if (c == -1) { n_skl[s] = -5; c = 0; }           // record list beyond its slot: this query only (flag -4 at the ABI)
            if (c == -3) c = 1;                              // the single start record
            if (c < 0) c = 0;
            off[s + 1] = off[s] + c;
        }
        skl.resize(off[nr]);
        if (off[nr]) {
            void* d_off = pool.get(HP_OFF, (nr + 1) * sizeof(int64_t));
            void* d_pack = pool.get(HP_PACK, off[nr] * sizeof(int2));
            std::vector<int> cnt(nr);
            for (int s = 0; s < nr; ++s) cnt[s] = (int) (off[s + 1] - off[s]);
            HIPCHK(hipMemcpyAsync(d_off, off.data(), (nr + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(d_nskl, cnt.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(spdp_launch_pack((const int2*) d_skl, H_SKL_CAP, (const int*) d_nskl, (const int64_t*) d_off,
                                    (int2*) d_pack, nr, ctx->stream));
            HIPCHK(hipMemcpyAsync(skl.data(), d_pack, off[nr] * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
    }
    // back to item order
    out.res.resize(nr); out.n_skl.resize(nr); out.off.assign(nr + 1, 0);
    std::vector<int> slot_of(nr);
    for (int s = 0; s < nr; ++s) slot_of[order[s].second] = s;
    for (int i = 0; i < nr; ++i) {
        const int s = slot_of[i];
        out.res[i] = res[s]; out.n_skl[i] = n_skl[s];
        out.off[i + 1] = out.off[i] + (off[s + 1] - off[s]);
    }
    out.skl.resize(out.off[nr]);
    for (int i = 0; i < nr; ++i) {
        const int s = slot_of[i];
        std::copy(skl.begin() + off[s], skl.begin() + off[s + 1], out.skl.begin() + out.off[i]);
    }
    return 0;
}

// the traceback bitmaps (2 B per cell of the reference's skewed layout, ~9 MB for a 400 aa query on a 12 kb window) bound
// how many problems one launch can hold: a list whose bitmaps exceed the budget (SPDP_H_TB_GB, default 160 GiB of the
// 288) runs as consecutive groups, each a full sweep + walk; results come back in item order either way
static int run_forward(HStore& st, const std::vector<HItem>& items, bool walk, HFwdOut& out)
{
    double gb = 160.0;
    if (const char* e = getenv("SPDP_H_TB_GB")) gb = std::max(0.001, atof(e));
    const int64_t budget = (int64_t) (gb * 1073741824.0 / 2.0);             // elements
    const int nr = (int) items.size();
    std::vector<int> cut(1, 0);
    int64_t acc = 0;
    for (int i = 0; i < nr; ++i) {
        const int64_t mw = items[i].a_right - items[i].a_left + 1;
        const int64_t el = mw * ((int64_t) items[i].b_right - items[i].b_left + 1 + 3 * mw) + 104;
        if (acc && acc + el > budget) { cut.push_back(i); acc = 0; }
        acc += el;
    }
    cut.push_back(nr);
    if (cut.size() == 2) return run_forward_group(st, items, walk, out);
    out = HFwdOut();
    out.off.assign(1, 0);
    for (size_t g = 0; g + 1 < cut.size(); ++g) {
        std::vector<HItem> part(items.begin() + cut[g], items.begin() + cut[g + 1]);
        HFwdOut po;
        if (run_forward_group(st, part, walk, po)) return -1;
        out.res.insert(out.res.end(), po.res.begin(), po.res.end());
        out.n_skl.insert(out.n_skl.end(), po.n_skl.begin(), po.n_skl.end());
        const int64_t base = out.off.back();
        for (size_t i = 1; i < po.off.size(); ++i) out.off.push_back(base + po.off[i]);
        out.skl.insert(out.skl.end(), po.skl.begin(), po.skl.end());
        out.sweep_ms += po.sweep_ms; out.cells += po.cells; out.tb_elems = std::max(out.tb_elems, po.tb_elems);
    }
    return 0;
}

// ---- the -A0 wavefront kernels with the tiles of a problem as a pipeline of waves (spdh_rowwave<., true>) ----
// The work list (problem, tile) in dispatch order behind the sync words of the problems; on whenever a problem has
// two tiles or more (SPDP_A0_PIPE=0: one wave per problem).  A wave that waited in vain for the tile above it
// leaves a mark and the caller repeats the launch without the pipeline.
struct HPipe {
    std::vector<int> items;
    int max_tiles = 1, stride = 0;
    size_t words = 0;
    int* d = nullptr;
    bool on = false;
};
static int pipe_setup(SpdpContext* ctx, DevPool& pool, int slot, const std::vector<DevProblemH>& probs, bool udh, int max_im, HPipe& pp)
{
    pp = HPipe();
    const char* e = getenv("SPDP_A0_PIPE");
    for (size_t j = 0; j < probs.size(); ++j) {
        const DevProblemH& P = probs[j];
        const int r0 = P.a_left + (P.a_exgl ? 1 : 0);
        const int th = udh ? std::max(1, std::min(64, P.imd_intvl)) : 64;
        const int nt = std::max(1, (P.a_right - r0 + th) / th);              // as the kernel counts them
        pp.max_tiles = std::max(pp.max_tiles, nt);
        for (int t = 0; t < nt; ++t) { pp.items.push_back((int) j); pp.items.push_back(t); }
    }
    if ((e && atoi(e) == 0) || pp.max_tiles < 2) return 0;
    // (Rounds 2 / 3 switched the pipeline off once every SIMD had a problem of its own: the kernels then ran ONE wave per
    // SIMD -- 256 VGPRs and a handful of AGPRs -- and the pipeline only added memory-side traffic.  Held to 256 registers
    // (amdgpu_waves_per_eu(2, 2), spdp_h_rowwave.hip) two waves share a SIMD and the pipeline pays at every size measured:
    // 1000 x 400 aa 2.9 -> 5.1 GCUPS, 1827 problems 3.7 - 4.6 (one wave per problem) -> 5.7.)
    pp.stride = 2 + 9 * pp.max_tiles + 3 * max_im;
    pp.words = (probs.size() * (size_t) pp.stride + 2 + 1) & ~(size_t) 1;
    pp.d = (int*) pool.get(slot, sizeof(int) * (pp.words + pp.items.size()));
    if (!pp.d) { ctx->err = "device allocation failed (tile pipeline of the scalar aa x genome engines)"; return -1; }
    pp.on = true;
    return 0;
}
static int pipe_arm(SpdpContext* ctx, const HPipe& pp, int n_probs, HScalarArgs& A)
{
    A.pipe = nullptr;
    if (!pp.on) return 0;
    HIPCHK(hipMemsetAsync(pp.d, 0, sizeof(int) * pp.words, ctx->stream));
    HIPCHK(hipMemcpyAsync(pp.d + pp.words, pp.items.data(), sizeof(int) * pp.items.size(), hipMemcpyHostToDevice, ctx->stream));
    A.pipe = pp.d; A.pipe_stride = pp.stride; A.pipe_ticket = n_probs * pp.stride; A.max_tiles = pp.max_tiles;
    A.items = (const int2*) (pp.d + pp.words); A.n_items = (int) (pp.items.size() / 2);
    return 0;
}
// after the launch has finished: 1 = a wave gave up (repeat without the pipeline), 0 = fine
static int pipe_stalled(SpdpContext* ctx, const HPipe& pp, int n_probs)
{
    if (!pp.on) return 0;
    int mark[2] = {0, 0};
    HIPCHK(spdp_copy_sync(mark, pp.d + (size_t) n_probs * pp.stride, sizeof mark, hipMemcpyDeviceToHost, ctx->stream));
    return (mark[1] != 0 || getenv("SPDP_A0_PIPE_TEST_STALL")) ? 1 : 0;
}

// the -A1 engines (spdh_exact): a pipelined work item is (G problems, stripe of 16 rows), one wave each; G = 4 once the
// launch fills the chip that way, fewer before.  SPDP_HX_PIPE=0: the stripes of a problem one after the other in one
// 16-lane group; SPDP_HX_GROUPS=1|2|4 forces G.
static int pipe_setup_exact(SpdpContext* ctx, DevPool& pool, int slot, const std::vector<DevProblemH>& probs, int max_im, HPipe& pp, int& G, bool nopipe = false)
{
    pp = HPipe();
    const int n = (int) probs.size();
    int64_t stripes = 0;
    for (const DevProblemH& P : probs) {
        const int ns = std::max(1, (P.a_right - P.a_left + 15) / 16);
        pp.max_tiles = std::max(pp.max_tiles, ns);
        stripes += ns;
    }
    G = stripes >= 4 * 4096 ? 4 : (stripes >= 2 * 4096 ? 2 : 1);
    if (const char* e = getenv("SPDP_HX_GROUPS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) G = v; }
    const char* e = getenv("SPDP_HX_PIPE");
    if (nopipe || (e && atoi(e) == 0) || pp.max_tiles < 2) return 0;
    for (int q = 0; q * G < n; ++q) {
        int ns = 1;
        for (int j = q * G; j < std::min(n, (q + 1) * G); ++j) ns = std::max(ns, std::max(1, (probs[j].a_right - probs[j].a_left + 15) / 16));
        for (int t = 0; t < ns; ++t) { pp.items.push_back(q); pp.items.push_back(t); }
    }
    pp.stride = 2 + 7 * pp.max_tiles + 3 * max_im;
    pp.words = ((size_t) n * pp.stride + 2 + 1) & ~(size_t) 1;
    pp.d = (int*) pool.get(slot, sizeof(int) * (pp.words + pp.items.size()));
    if (!pp.d) { ctx->err = "device allocation failed (stripe pipeline of the -A1 aa x genome engines)"; return -1; }
    pp.on = true;
    return 0;
}

// ---- scalar forwardH_ng over a list of items (spdp_h_rowwave.hip) ------------------------------
// Vmf record budget of one forwardH_ng / forwardH1 call: a record is written where a diagonal run starts, twice per
// accepted intron and per first-row restart -- far fewer than cells on real inputs.  One per two cells (at least 64 per
// row) to start with; a problem that outgrows it reports -3 and is run again with eight times as much, up to the
// four per cell nothing can exceed (as the cDNA path does, DevRun / run_vmf).
static int64_t vmf_budget_h(const DevProblemH& d, int scale)
{
    const int64_t rows = d.a_right - d.a_left + 1;
    const int64_t full = 4 * d.cells + 3ll * (d.b_right - d.b_left + 8) + 64;
    if (scale == 1 && getenv("SPDP_VMF_TEST_TINY")) return 96;      // test hook: every problem outgrows its first budget
    return std::min<int64_t>(full, std::max<int64_t>(d.cells / 2, 64 * rows) * scale + 3ll * (d.b_right - d.b_left + 8) + 64);
}

static int run_scalar_group(HStore& st, const std::vector<HItem>& items, bool forward, HFwdOut& out, bool exact, int scale, bool cut = false, bool nopipe = false)
{
    SpdpContext* ctx = lane_of(st);
    DevPool& pool = ctx->pool[H_POOL];
    const int nr = (int) items.size();
    out = HFwdOut();
    if (!nr) return 0;
    if (!st.scalar_ok) { ctx->err = "the scalar engine needs SpdpScoringH.intpen / t53 and SpdpProblemH.dinc"; return -1; }
    std::vector<DevProblemH> h_probs(nr);
    int64_t work_int = 0, vmf_rec = 0;
    int skl_cap = 64;
    for (int i = 0; i < nr; ++i) {
        DevProblemH& d = h_probs[i];
        fill_desc(st, items[i], d);
        d.bnd_off = work_int;
        // scalar: two rows of {val, ptr, dir}; -A1 (forwardH1): six boundary rows by diagonal + the record counter
        // (Noll = 3: a third group of planes, F2)
        work_int += exact ? 6ll * d.buf_size + 8 : 3ll * ((st.sc.noll == 3 ? 3 : 2) * ((int64_t) d.width + 4));
        d.tb_off = vmf_rec;
        // Vmf records: one per cell that starts a diagonal run, two per accepted intron, the boundary
        // row; 4 per cell is far above what the recurrence can emit on real inputs (overflow is reported)
        // (+ what the waves of a pipelined problem may leave unused of the chunks of numbers they reserve)
        // (-A1: every lane of a stripe may leave up to two chunks of 16 numbers unused: 32 per row)
        const int64_t cap = forward ? vmf_budget_h(d, scale) + (int64_t) (exact ? 4 : 1) * SPDP_VMF_CHUNK * ((d.a_right - d.a_left) / 64 + 2) : 0;
        if (cap >= (int64_t) 1 << 31) { ctx->err = "scalar engine: problem too large for its record store"; return -1; }
        d.imd_off = cap;
        vmf_rec += cap;
        out.cells += d.cells;
        skl_cap = std::max(skl_cap, std::min(H_SKL_CAP, (d.a_right - d.a_left) + (d.b_right - d.b_left) + 8));
    }
    void* d_probs = pool.get(HP_PROBS, nr * sizeof(DevProblemH));
    void* d_work = pool.get(HP_BND, (size_t) work_int * sizeof(int));
    void* d_vmf = pool.get(HP_TB, (size_t) std::max<int64_t>(vmf_rec, 1) * sizeof(int3));
    void* d_res = pool.get(HP_RES, nr * sizeof(DevResultH));
    void* d_skl = pool.get(HP_SKL, (size_t) nr * skl_cap * sizeof(int2));
    void* d_nskl = pool.get(HP_NSKL, nr * sizeof(int));
    if (!d_probs || !d_work || !d_vmf || !d_res || !d_skl || !d_nskl) {
        ctx->err = "device allocation failed (scalar aa x genome run: records need " +
                   std::to_string((size_t) vmf_rec * 12 >> 20) + " MiB)";
        return -1;
    }
    HIPCHK(hipMemcpyAsync(d_probs, h_probs.data(), nr * sizeof(DevProblemH), hipMemcpyHostToDevice, ctx->stream));
    HScalarArgs A;
    memset(&A, 0, sizeof A);
    A.sc = (const DevScoringH*) st.d_sc; A.probs = (const DevProblemH*) d_probs; A.n_probs = nr;
    A.a_codes = (const uint8_t*) st.d_a; A.cols = (const int4*) st.d_cols; A.aux = (const short4*) st.d_aux;
    A.intpen = (const int16_t*) st.d_intpen; A.intpen_len = st.sc.intpen_len;
    A.ipen_runs = st.ipen_runs_ok ? (const int16_t*) st.d_intpen + st.sc.intpen_len : nullptr;
    A.minl = st.sc.minl ? st.sc.minl : st.sc.llmt;
    A.gape1 = st.sc.gape1; A.gape2 = st.sc.gape2; A.extragop = st.sc.extragop;
    A.noll = (!exact && st.sc.noll == 3) ? 3 : 2; A.lgop = st.sc.lgop;
    memcpy(A.t53, st.sc.t53, sizeof A.t53);
    spdp_genetic_code_tables(A.mid, A.tron_of);
    A.cip = (const int*) st.d_cip;
    A.work = (int*) d_work; A.vmf = (int3*) d_vmf; A.res = (DevResultH*) d_res;
    A.skl = (int2*) d_skl; A.n_skl = (int*) d_nskl; A.skl_cap = skl_cap;
    HPipe pp;
    if (exact) { if (pipe_setup_exact(ctx, pool, HP_PIPE, h_probs, 0, pp, A.item_probs, nopipe)) return -1; }
    else if (!cut && A.noll != 3 && pipe_setup(ctx, pool, HP_PIPE, h_probs, false, 0, pp)) return -1;     // (the cut-range variant and double affine gaps run one wave per problem)
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (pipe_arm(ctx, pp, nr, A)) return -1;
        HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
        if (exact) HIPCHK(spdh_launch_exact(0, &A, ctx->stream));
        else HIPCHK(spdh_launch_scalar(cut ? 2 : (forward ? 1 : 0), &A, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        if (!pp.on) break;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const int st_ = pipe_stalled(ctx, pp, nr);
        if (st_ < 0) return -1;
        if (!st_) break;
        pp.on = false;
    }
    out.res.resize(nr); out.n_skl.assign(nr, 0); out.off.assign(nr + 1, 0);
    HIPCHK(hipMemcpyAsync(out.res.data(), d_res, nr * sizeof(DevResultH), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int2> skl;
    if (forward) {
        skl.resize((size_t) nr * skl_cap);
        HIPCHK(hipMemcpyAsync(out.n_skl.data(), d_nskl, nr * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(skl.data(), d_skl, skl.size() * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&out.sweep_ms, ctx->ev0, ctx->ev1));
    if (getenv("SPDP_TRACE_RUNS"))
        fprintf(stderr, "[spdp run] aa x genome %s forward=%d n %d cells %.3g pipe %d items %zu  %.2f ms  %.2f GCUPS\n",
                exact ? "-A1" : "-A0", (int) forward, nr, (double) out.cells, (int) pp.on, pp.items.size() / 2, out.sweep_ms,
                out.cells / (out.sweep_ms * 1e6));
    // -A1, pipelined: a result whose walk met a link word the reference leaves from stripe to stripe (spdp_h_exact.hip:
    // HXPOISON) runs again, one 16-lane group per problem
    std::vector<int> redo;
    for (int i = 0; exact && pp.on && i < nr; ++i) if (out.res[i].pad[1]) redo.push_back(i);
    HFwdOut ro;
    if (!redo.empty()) {
        std::vector<HItem> part;
        for (int i : redo) part.push_back(items[i]);
        if (run_scalar_group(st, part, forward, ro, exact, scale, cut, true)) return -1;
        out.sweep_ms += ro.sweep_ms;                    // (the second run is part of what the call cost)
        for (size_t k = 0; k < redo.size(); ++k) { out.res[redo[k]] = ro.res[k]; out.n_skl[redo[k]] = ro.n_skl[k]; }
    }
    for (int i = 0; i < nr; ++i) {
        const int c = out.n_skl[i];
        if (c == -1) { ctx->err = "traceback record buffer overflow (scalar engine)"; return -1; }
        if (c == -3) { out.off[i + 1] = out.off[i]; continue; }             // outgrew its record budget: the caller runs it again, larger
        if (c == -4) { out.n_skl[i] = -3; out.off[i + 1] = out.off[i]; continue; }   // -A1, mode 3: record pointer beyond an int16 lane (undefined)
        out.off[i + 1] = out.off[i] + std::max(c, 0);
    }
    out.skl.resize(out.off[nr]);
    for (int i = 0; i < nr; ++i)
        for (int k = 0; k < out.n_skl[i]; ++k) {
            SpdpSkl s; s.m = skl[(size_t) i * skl_cap + k].x; s.n = skl[(size_t) i * skl_cap + k].y;
            out.skl[out.off[i] + k] = s;
        }
    for (size_t k = 0; k < redo.size(); ++k)
        for (int j = 0; j < ro.n_skl[k]; ++j) out.skl[out.off[redo[k]] + j] = ro.skl[ro.off[k] + j];
    return 0;
}

// the same over any number of items: launches whose record space stays below SPDP_VMF_GB (default 32) gigabytes, and
// another round with a larger budget for the problems that outgrew theirs
static int run_scalar(HStore& st, const std::vector<HItem>& items, bool forward, HFwdOut& out, bool exact = false, bool cut = false)
{
    SpdpContext* ctx = lane_of(st);
    const int nr = (int) items.size();
    out = HFwdOut();
    if (!nr) return 0;
    if (!forward) return run_scalar_group(st, items, false, out, exact, 1);
    size_t limit = (size_t) 32 << 30;
    if (const char* e = getenv("SPDP_VMF_GB")) limit = (size_t) std::max(1, atoi(e)) << 30;
    std::vector<DevResultH> res(nr);
    std::vector<int> n_skl(nr, 0);
    std::vector<std::vector<SpdpSkl>> lists(nr);
    std::vector<int> todo(nr);
    for (int i = 0; i < nr; ++i) todo[i] = i;
    // dispatch order: the problems with the most anti-diagonals first.  A wave sweeps one problem (or one tile of it) and the hardware
    // hands blocks out in index order: a long problem that starts late ends the launch late, and a block whose four problems differ in
    // length holds its LDS until the longest is done
    {
        std::vector<int64_t> steps(nr);
        for (int i = 0; i < nr; ++i) {
            const HItem& t = items[i];
            const int64_t rows = t.a_right - t.a_left;
            steps[i] = std::min<int64_t>((int64_t) t.b_right - t.b_left, (int64_t) t.w.up - t.w.lw + 3 * rows) + rows;
        }
        std::stable_sort(todo.begin(), todo.end(), [&](int x, int y) { return steps[x] > steps[y]; });
    }
    for (int scale = 1; !todo.empty(); scale *= 8) {
        if (scale > 32768) { ctx->err = "scalar engine: Vmf record store overflow"; return -1; }
        std::vector<int> again;
        // launches of about equal size (a last small one runs at a fraction of the rate of a full one)
        size_t total = 0;
        for (int i : todo) { DevProblemH d; fill_desc(st, items[i], d); total += (size_t) vmf_budget_h(d, scale) * sizeof(int3); }
        const size_t n_launch = std::max<size_t>(1, (total + limit - 1) / limit);
        const size_t even = std::min(limit, total / n_launch + total / n_launch / 16 + 1);
        for (size_t lo = 0; lo < todo.size(); ) {
            size_t hi = lo, sum = 0;
            std::vector<HItem> part;
            while (hi < todo.size()) {
                DevProblemH d;
                fill_desc(st, items[todo[hi]], d);
                const size_t bytes = (size_t) vmf_budget_h(d, scale) * sizeof(int3);
                if (hi > lo && sum + bytes > even) break;
                sum += bytes; part.push_back(items[todo[hi++]]);
            }
            HFwdOut po;
            if (run_scalar_group(st, part, true, po, exact, scale, cut)) return -1;
            out.sweep_ms += po.sweep_ms; out.cells += po.cells;
            for (size_t k = lo; k < hi; ++k) {
                const int i = todo[k], c = po.n_skl[k - lo];
                if (c == -3) { again.push_back(i); continue; }
                res[i] = po.res[k - lo]; n_skl[i] = c;
                if (c > 0) lists[i].assign(po.skl.begin() + po.off[k - lo], po.skl.begin() + po.off[k - lo] + c);
            }
            lo = hi;
        }
        todo.swap(again);
    }
    out.res.swap(res); out.n_skl.swap(n_skl);
    out.off.assign(nr + 1, 0);
    for (int i = 0; i < nr; ++i) out.off[i + 1] = out.off[i] + (int64_t) lists[i].size();
    out.skl.resize(out.off[nr]);
    for (int i = 0; i < nr; ++i) std::copy(lists[i].begin(), lists[i].end(), out.skl.begin() + out.off[i]);
    return 0;
}

// ---- scalar hirschbergH_ng over a list of items (spdp_h_rowwave.hip) ----------------------------
struct HUdhOut;
// engine: 0 hirschbergH_ng (scalar), 1 hirschbergH1 (-A1), 2 hirschbergH1_wip with local ends (-LS)
static int run_scalar_udh(HStore& st, const std::vector<HItem>& items, HUdhOut& out, std::vector<int>& flags, int engine = 0, bool nopipe = false);

// ---- hirschbergH1_wip over a list of items ------------------------------------------------------
struct HUdhOut {
    std::vector<int32_t> scores, cpos, ranges;  // in item order; cpos stride = stride ints per item
    int stride = 0;
    float sweep_ms = 0.f;
    int64_t cells = 0;
};

static int run_udh(HStore& st, const std::vector<HItem>& items, HUdhOut& out)
{
    SpdpContext* ctx = lane_of(st);
    DevPool& pool = ctx->pool[HU_POOL];
    const int nr = (int) items.size();
    out = HUdhOut();
    if (!nr) return 0;
    int max_im = 1;
    for (const HItem& it : items) max_im = std::max(max_im, it.n_im);
    out.stride = (max_im + 1) * 10;
    std::vector<DevProblemH> h_probs(nr);
    int64_t bnd_ent = 0, imd_int = 0;
    for (int i = 0; i < nr; ++i) {
        DevProblemH& d = h_probs[i];
        fill_desc(st, items[i], d);
        d.bnd_off = bnd_ent;
        d.imd_off = imd_int;
        bnd_ent += d.buf_size + SPDH_BND_PAD;
        imd_int += (int64_t) d.n_im * 5 * d.width;
        out.cells += d.cells;
    }
    void* d_probs = pool.get(HU_PROBS, nr * sizeof(DevProblemH));
    void* d_bnd = pool.get(HU_BND, (size_t) bnd_ent * sizeof(int4));
    void* d_imd = pool.get(HU_IMD, (size_t) std::max<int64_t>(imd_int, 1) * sizeof(int));
    void* d_res = pool.get(HU_RES, nr * sizeof(DevResultH));
    void* d_cpos = pool.get(HU_CPOS, (size_t) nr * out.stride * sizeof(int));
    void* d_ranges = pool.get(HU_RANGES, (size_t) nr * 4 * sizeof(int));
    void* d_scores = pool.get(HU_SCORES, (size_t) nr * sizeof(int));
    if (!d_probs || !d_bnd || !d_imd || !d_res || !d_cpos || !d_ranges || !d_scores) {
        ctx->err = "device allocation failed (aa x genome linear-space run)";
        return -1;
    }
    HIPCHK(hipMemcpyAsync(d_probs, h_probs.data(), nr * sizeof(DevProblemH), hipMemcpyHostToDevice, ctx->stream));
    HUdhArgs A;
    A.sc = (const DevScoringH*) st.d_sc; A.probs = (const DevProblemH*) d_probs; A.n_probs = nr;
    A.a_codes = (const uint8_t*) st.d_a; A.cols = (const int4*) st.d_cols; A.aux = (const short4*) st.d_aux;
    A.bnd = (int4*) d_bnd; A.imd = (int*) d_imd; A.res = (DevResultH*) d_res;
    const int pen_cap = st.sc.nquant > 1 ? st.sc.qm_len[st.sc.nquant - 2] + 1 : 0;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(spdh_launch_udh(&A, st.sc.spj, pen_cap, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HCposArgs Cc;
    Cc.probs = A.probs; Cc.n_probs = nr; Cc.imd = A.imd; Cc.res = A.res;
    Cc.cpos = (int*) d_cpos; Cc.ranges = (int*) d_ranges; Cc.scores = (int*) d_scores; Cc.cpos_stride = out.stride;
    HIPCHK(spdh_launch_cpos(&Cc, ctx->stream));
    out.scores.resize(nr); out.cpos.resize((size_t) nr * out.stride); out.ranges.resize((size_t) nr * 4);
    HIPCHK(hipMemcpyAsync(out.scores.data(), d_scores, nr * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out.cpos.data(), d_cpos, (size_t) nr * out.stride * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out.ranges.data(), d_ranges, (size_t) nr * 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&out.sweep_ms, ctx->ev0, ctx->ev1));
    return 0;
}

static int run_scalar_udh(HStore& st, const std::vector<HItem>& items, HUdhOut& out, std::vector<int>& flags, int engine, bool nopipe)
{
    const bool exact = engine != 0;
    SpdpContext* ctx = lane_of(st);
    DevPool& pool = ctx->pool[HU_POOL];
    const int nr = (int) items.size();
    out = HUdhOut();
    flags.assign(nr, 0);
    if (!nr) return 0;
    if (engine != 2 && !st.scalar_ok) { ctx->err = "the scalar engine needs SpdpScoringH.intpen / t53 and SpdpProblemH.dinc"; return -1; }
    int max_im = 1;
    for (const HItem& it : items) max_im = std::max(max_im, it.n_im);
    out.stride = (max_im + 1) * 10;
    std::vector<DevProblemH> h_probs(nr);
    int64_t work_int = 0, imd_int = 0;
    for (int i = 0; i < nr; ++i) {
        DevProblemH& d = h_probs[i];
        fill_desc(st, items[i], d);
        d.bnd_off = work_int;
        const int noll = (engine == 0 && st.sc.noll == 3) ? 3 : 2;        // (Noll = 3: F2 planes, a third plane of links per intermediate row)
        work_int += exact ? 6ll * d.buf_size + 8 : 6ll * (noll * ((int64_t) d.width + 4));
        d.imd_off = imd_int;
        imd_int += (int64_t) d.n_im * 4 * noll * d.width;
        out.cells += d.cells;
    }
    void* d_probs = pool.get(HU_PROBS, nr * sizeof(DevProblemH));
    void* d_work = pool.get(HU_BND, (size_t) work_int * sizeof(int));
    void* d_imd = pool.get(HU_IMD, (size_t) std::max<int64_t>(imd_int, 1) * sizeof(int));
    void* d_res = pool.get(HU_RES, nr * sizeof(DevResultH));
    void* d_cpos = pool.get(HU_CPOS, (size_t) nr * out.stride * sizeof(int));
    void* d_ranges = pool.get(HU_RANGES, (size_t) nr * 4 * sizeof(int));
    void* d_scores = pool.get(HU_SCORES, (size_t) nr * sizeof(int));
    if (!d_probs || !d_work || !d_imd || !d_res || !d_cpos || !d_ranges || !d_scores) {
        ctx->err = "device allocation failed (scalar aa x genome linear-space run)";
        return -1;
    }
    HIPCHK(hipMemcpyAsync(d_probs, h_probs.data(), nr * sizeof(DevProblemH), hipMemcpyHostToDevice, ctx->stream));
    HScalarArgs A;
    memset(&A, 0, sizeof A);
    A.sc = (const DevScoringH*) st.d_sc; A.probs = (const DevProblemH*) d_probs; A.n_probs = nr;
    A.a_codes = (const uint8_t*) st.d_a; A.cols = (const int4*) st.d_cols; A.aux = (const short4*) st.d_aux;
    A.intpen = (const int16_t*) st.d_intpen; A.intpen_len = st.sc.intpen_len;
    A.ipen_runs = st.ipen_runs_ok ? (const int16_t*) st.d_intpen + st.sc.intpen_len : nullptr;
    A.minl = st.sc.minl ? st.sc.minl : st.sc.llmt;
    A.gape1 = st.sc.gape1; A.gape2 = st.sc.gape2; A.extragop = st.sc.extragop;
    A.noll = (engine == 0 && st.sc.noll == 3) ? 3 : 2; A.lgop = st.sc.lgop;
    memcpy(A.t53, st.sc.t53, sizeof A.t53);
    spdp_genetic_code_tables(A.mid, A.tron_of);
    A.cip = (const int*) st.d_cip;
    A.work = (int*) d_work; A.res = (DevResultH*) d_res;
    A.imd = (int*) d_imd; A.cpos = (int*) d_cpos; A.ranges = (int*) d_ranges; A.scores = (int*) d_scores;
    A.cpos_stride = out.stride;
    HPipe pp;
    if (engine == 1) { if (pipe_setup_exact(ctx, pool, HU_PIPE, h_probs, max_im, pp, A.item_probs, nopipe)) return -1; }
    else if (engine == 0 && A.noll != 3 && pipe_setup(ctx, pool, HU_PIPE, h_probs, true, max_im, pp)) return -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (pipe_arm(ctx, pp, nr, A)) return -1;
        HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
        if (engine == 2) HIPCHK(spdh_launch_local_udh(&A, ctx->stream));
        else if (exact) HIPCHK(spdh_launch_exact(1, &A, ctx->stream));
        else HIPCHK(spdh_launch_scalar_udh(&A, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        if (!pp.on) break;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const int st_ = pipe_stalled(ctx, pp, nr);
        if (st_ < 0) return -1;
        if (!st_) break;
        pp.on = false;
    }
    out.scores.resize(nr); out.cpos.resize((size_t) nr * out.stride); out.ranges.resize((size_t) nr * 4);
    std::vector<DevResultH> res(nr);
    HIPCHK(hipMemcpyAsync(out.scores.data(), d_scores, nr * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out.cpos.data(), d_cpos, (size_t) nr * out.stride * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out.ranges.data(), d_ranges, (size_t) nr * 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(res.data(), d_res, nr * sizeof(DevResultH), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&out.sweep_ms, ctx->ev0, ctx->ev1));
    if (getenv("SPDP_TRACE_RUNS"))
        fprintf(stderr, "[spdp run] aa x genome linear space engine %d n %d cells %.3g pipe %d items %zu  %.2f ms  %.2f GCUPS\n",
                engine, nr, (double) out.cells, (int) pp.on, pp.items.size() / 2, out.sweep_ms, out.cells / (out.sweep_ms * 1e6));
    for (int i = 0; i < nr; ++i) flags[i] = exact ? 0 : res[i].pad[0];
    // -A1, pipelined: dead results run again in the one-group form (see run_scalar_group)
    std::vector<int> redo;
    for (int i = 0; engine == 1 && pp.on && i < nr; ++i) if (res[i].pad[1]) redo.push_back(i);
    if (getenv("SPDP_TRACE_RUNS") && engine == 1 && pp.on) {
        fprintf(stderr, "[spdp run] -A1 linear space: %zu of %d results run again without the pipeline\n", redo.size(), nr);
    }
    if (!redo.empty()) {
        std::vector<HItem> part;
        for (int i : redo) part.push_back(items[i]);
        HUdhOut ro;
        std::vector<int> rf;
        if (run_scalar_udh(st, part, ro, rf, engine, true)) return -1;
        out.sweep_ms += ro.sweep_ms;
        for (size_t k = 0; k < redo.size(); ++k) {
            const int i = redo[k];
            out.scores[i] = ro.scores[k];
            for (int c = 0; c < 4; ++c) out.ranges[(size_t) i * 4 + c] = ro.ranges[k * 4 + c];
            for (int c = 0; c < std::min(out.stride, ro.stride); ++c) out.cpos[(size_t) i * out.stride + c] = ro.cpos[k * ro.stride + c];
            flags[i] = rf[k];
        }
    }
    return 0;
}

// ---- the reference's dispatch (lspH_ng & co.) over a batch, in rounds ---------------------------
struct HTop {
    int cls = 0;                                // 0 ok, 1 needs an engine that is not built, 2 bad input
    int flag = 0;                               // -1 "Unexpected dir", -2 traceback start outside the bitmap
    int score = SPDP_NEVSEL;
    std::vector<SpdpSkl> rec;
};

struct HStats { float fwd_ms = 0.f, udh_ms = 0.f; int64_t fwd_cells = 0, udh_cells = 0; int rounds = 0; };

static const int END_ULK = SPDP_END_OF_ULK;

static void push_rec(HTop& t, int m, int n) { SpdpSkl s; s.m = m; s.n = n; t.rec.push_back(s); }

// trcbkalignH_ng's engine choice (src/fwd2h1.cc:1997-2014): returns false when the item cannot run here
// a sub-range the links produced that does not lie inside the sequences: the reference would run its
// engine on it anyway (out-of-bounds reads); here the query is reported as not computed
static bool bad_range(const HItem& it, const SpdpProblemH& p)
{
    // a_len + 1: the right end the local linear-space engines report for a path that ends on the last row
    // (src/fwd2h1_wip_simd.h:652-653); the reference goes on with it and reads the padding residue
    return it.a_left < 0 || it.b_left < 0 || it.a_right > p.a_len + 1 || it.b_right > p.b_len ||
           it.a_right < it.a_left || it.b_right < it.b_left || it.b_left < p.exin_left || it.b_right > p.exin_right;
}

static thread_local int a0_mode = 0;             // SpdpScoringH.scalar_engines of the running ladder: 1 -A0, 2 -A1

static bool queue_trcbk(const HItem& it, std::vector<HItem>& fwd, std::vector<HItem>& scl, bool scalar_ok, HTop& t)
{
    if (it.w.width < 0) return true;                         // NEVSEL, no records
    if (it.cut_r > it.cut_l) {                               // a cut range: always the scalar engine (:2004-2008)
        // what the kernel serves: both ends global (shortcutH_ng clears all four flags), every row starts before the
        // cut, room for the seven entries a row needs
        const int cl = it.cut_r - it.cut_l;
        if (!scalar_ok || it.a_exgl || it.a_exgr || it.b_exgl || it.b_exgr || it.w.width - cl < 7 ||
            it.cut_l < it.b_left || it.cut_r > it.b_right || 3 * it.a_right + it.w.lw - 1 > it.cut_l) { t.cls = 1; return false; }
        scl.push_back(it);
        return true;
    }
    if (a0_mode == 1 || it.a_right - it.a_left < 8) {        // -A0, or below 8 rows: scalar forwardH_ng
        if (!scalar_ok) { t.cls = 1; return false; }
        scl.push_back(it);
        return true;
    }
    if (a0_mode == 2) {                                      // -A1: forwardH1 (modes 3 / 5)
        if (!scalar_ok) { t.cls = 1; return false; }
        scl.push_back(it);
        scl.back().exact = true;
        return true;
    }
    fwd.push_back(it);
    return true;
}

// lspH_ng (src/fwd2h1.cc:2140-2180) up to the engine call
static void queue_lsp(const SpdpScoringH& sc, const SpdpProblemH& p, HItem it, std::vector<HItem>& fwd, std::vector<HItem>& scl,
                      bool scalar_ok, std::vector<HItem>& udh, HTop& t)
{
    const int m = it.a_right - it.a_left, n = it.b_right - it.b_left;
    if (!m && !n) { if (it.first) t.score = 0; return; }
    if (!m || !n) {                                          // one sequence only: the two end records and the gap's
        push_rec(t, it.a_left, it.b_left);                   // price (src/fwd2h1.cc:2148-2159; PwdB::GapPenalty /
        push_rec(t, it.a_right, it.b_right);                 // GapExtPen / UnpPenalty3 / GapExtPen3, src/aln.h:275-304)
        if (it.first) {
            if (m) t.score = (it.a_exgl || it.a_exgr) ? (m > sc.codonk1 ? sc.lgep : sc.gep)
                                                      : (m > sc.codonk1 ? sc.lgop + m * sc.lgep : sc.gop + m * sc.gep);
            else if (it.b_exgl || it.b_exgr) t.score = n > sc.codonk1 ? sc.lgep : sc.gep;
            else {
                const int d = n / 3;
                const int egop = n % 3 == 1 ? sc.gape1 : (n % 3 == 2 ? sc.gape2 : 0);
                t.score = n <= sc.codonk1 ? d * sc.gep + egop : d * sc.gep - sc.diffu * (d - sc.k1) + egop;
            }
        }
        return;
    }
    if (it.w.up == it.w.lw) {                                // diagonalH_ng (src/fwd2h1.cc:1963-1995): one O(m) walk,
        const bool LocalL = sc.local && it.a_exgl && it.b_exgl;  // host side in the reference's ladder as well
        const bool LocalR = sc.local && it.a_exgr && it.b_exgr;
        int scr = 0, maxh = SPDP_NEVSEL, mL = it.a_left, mR = it.a_right;
        for (int mm = it.a_left; mm < it.a_right; ++mm) {
            const int nn = it.b_left + 1 + 3 * (mm - it.a_left);
            if (mm >= p.a_len || nn > p.b_len) break;
            scr += sc.mtx[p.a[mm] * sc.mtx_cols + p.b[nn]] + p.sigE[nn];
            if (LocalL && scr < 0) { scr = 0; mL = mm + 1; }
            if (LocalR && scr > maxh) { maxh = scr; mR = mm + 1; }
        }
        push_rec(t, mL, 3 * (mL - it.a_left) + it.b_left);
        push_rec(t, mR, 3 * (mR - it.a_left) + it.b_left);
        if (it.first) t.score = LocalR ? maxh : scr;
        return;
    }
    if (std::abs(n - m) < 16 || m == 1 || n <= 3) { queue_trcbk(it, fwd, scl, scalar_ok, t); return; }
    const float coef_B = 2.f, coef_C = 12.f;                 // sizeof(short); (Noll + 1) * sizeof(int)
    float cvol = float(m) * (n + 3 * m);                     // rhombic, simd >= 2
    if (a0_mode) {                                           // hexagonal, simd < 2 (src/fwd2h1.cc:2166-2169)
        const float k = it.w.lw - it.b_left + 3 * it.a_right;
        const float q = it.b_right - 3 * it.a_left - it.w.up;
        cvol = float(m) * n - (k * k + q * q) / 6;
    }
    if (coef_B * cvol < sc.max_vmf_space) { queue_trcbk(it, fwd, scl, scalar_ok, t); return; }
    bool recursive = sc.recursive != 0;                      // algmode.alg & 4
    int n_imd = 1;
    it.imd_intvl = (m + 1) / 2;
    if (!recursive) {
        const double z = 2. * m * coef_B / coef_C;
        const int imd1 = int(pow(z, 1. / 3) + 0.5) - 1;
        const float spc = coef_C * n * imd1 + coef_B * cvol / (imd1 + 1) / (imd1 + 1);
        if (spc > sc.max_vmf_space) recursive = true;
        else {
            const int imd3 = m / 16;
            n_imd = sc.ubh ? sc.ubh : std::min(imd1, imd3);
            const int intvl = it.imd_intvl = (m + n_imd) / (n_imd + 1);
            if (intvl * n_imd == m) --n_imd;
            if (n_imd == 0) { queue_trcbk(it, fwd, scl, scalar_ok, t); return; }
        }
    }
    if (a0_mode && !scalar_ok) { t.cls = 1; return; }
    it.n_im = n_imd; it.recursive = recursive;
    udh.push_back(it);
}

// window of a slab: stripe31() under SIMD; under -A0 the diagonal bounds hirschbergH_ng recorded in the
// cpos row (src/fwd2h1.cc:2068-2073, 2108-2128)
static void slab_window(HItem& it, int sh, const int32_t* row)
{
    if (a0_mode != 1) { stripe31_rng(it.a_left, it.a_right, it.b_left, it.b_right, sh, &it.w); return; }
    it.w.lw = row[8]; it.w.up = row[9];
    it.w.width = it.w.up - it.w.lw + 7;
}

static int run_ladder(HStore& st, bool ladder, std::vector<HTop>& tops, HStats& hs, const SpdhRequest* reqs = nullptr, int n_reqs = 0)
{
    const SpdpScoringH& sc = st.sc;
    a0_mode = ladder ? sc.scalar_engines : 0;
    tops.assign(reqs ? n_reqs : st.n, HTop());
    std::vector<HItem> pending, fwd, udh, scl;
    // explicit requests (the seeded walk, spdp_seeded_h.cpp): a sub-range of a resident parent with its own flags and
    // window; kind 0 enters the ladder as lspH_ng(wdw), 1 / 3 go straight to trcbkalignH_ng's engine choice
    for (int k = 0; reqs && k < n_reqs; ++k) {
        const SpdhRequest& q = reqs[k];
        HItem it;
        it.top = q.parent; it.slot = k; it.first = true;
        it.a_left = q.al; it.a_right = q.ar; it.b_left = q.bl; it.b_right = q.br;
        it.a_exgl = q.exg[0]; it.a_exgr = q.exg[1]; it.b_exgl = q.exg[2]; it.b_exgr = q.exg[3];
        it.w = q.w;
        if (q.parent < 0 || q.parent >= st.n) { tops[k].cls = 2; continue; }
        if (q.kind == 0) { pending.push_back(it); continue; }
        it.nospj = q.kind == 3; it.cut_l = q.cut_l; it.cut_r = q.cut_r;
        if (bad_range(it, st.probs[it.top])) { tops[k].cls = 1; continue; }
        queue_trcbk(it, fwd, scl, st.scalar_ok, tops[k]);
    }
    for (int i = 0; !reqs && i < st.n; ++i) {
        HItem it = item_of(st.probs[i], i, sc.sh);
        it.first = true;
        if (ladder) pending.push_back(it);
        else {                                               // engine level: forwardH1_wip on the stripe31 band
            const int m = it.a_right - it.a_left, n = it.b_right - it.b_left;
            if (!m || !n || it.w.width < 0) tops[i].cls = 2;
            else if (m < 8) tops[i].cls = 1;
            else fwd.push_back(it);
        }
    }
    while (!pending.empty() || !fwd.empty() || !scl.empty()) {
        ++hs.rounds;
        udh.clear();
        for (const HItem& it : pending) {
            if (tops[it.slot].cls) continue;
            if (bad_range(it, st.probs[it.top])) { tops[it.slot].cls = 1; continue; }
            queue_lsp(sc, st.probs[it.top], it, fwd, scl, st.scalar_ok, udh, tops[it.slot]);
        }
        pending.clear();
        // ---- linear-space round: cpos rows -> slabs (mimd_postwork) or halves (rcsv_postwork)
        if (!udh.empty()) {
            HUdhOut uo;
            std::vector<int> uflags(udh.size(), 0);
            if (a0_mode ? run_scalar_udh(st, udh, uo, uflags, a0_mode == 2 ? 1 : 0)
                        : sc.local ? run_scalar_udh(st, udh, uo, uflags, 2) : run_udh(st, udh, uo)) return -1;
            hs.udh_ms += uo.sweep_ms; hs.udh_cells += uo.cells;
            for (size_t u = 0; u < udh.size(); ++u) {
                const HItem& it = udh[u];
                HTop& t = tops[it.slot];
                if (t.cls) continue;
                const int scr = uo.scores[u];
                if (uflags[u]) { t.cls = 1; continue; }      // undefined in the reference
                if (it.first) t.score = scr;
                if (scr <= SPDP_NEVSEL) continue;
                const int32_t* cpos = &uo.cpos[u * uo.stride];
                const int32_t* rg = &uo.ranges[u * 4];
                HItem cur = it;
                cur.first = false; cur.n_im = 0; cur.recursive = false;
                cur.a_left = rg[0]; cur.a_right = rg[1]; cur.b_left = rg[2]; cur.b_right = rg[3];
#define CP(i, c) cpos[(i) * 10 + (c)]
                if (CP(0, 0) == END_ULK) {                   // does not cross an intermediate row
                    push_rec(t, cur.a_left, cur.b_left);
                    push_rec(t, cur.a_right, cur.b_right);
                } else if (it.recursive) {                   // rcsv_postwork, src/fwd2h1.cc:2091-2138
                    cur.a_exgl = cur.a_exgr = cur.b_exgl = cur.b_exgr = 0;
                    int c = 1;
                    while (c < 9 && CP(0, c + 1) < END_ULK) { ++c; push_rec(t, CP(0, 0), CP(0, c)); }
                    HItem h1 = cur, h2 = cur;
                    h1.a_right = CP(0, 0); h1.b_right = CP(0, c);
                    slab_window(h1, sc.sh, &CP(0, 0));
                    h2.a_left = CP(0, 0); h2.b_exgl = CP(0, 1); h2.b_left = CP(0, 2);
                    slab_window(h2, sc.sh, &CP(1, 0));
                    pending.push_back(h1);
                    pending.push_back(h2);
                } else {                                     // mimd_postwork, src/fwd2h1.cc:2045-2089
                    const int aleft = cur.a_left, bleft = cur.b_left;
                    const int a_len = st.probs[it.top].a_len, b_len = st.probs[it.top].b_len;
                    cur.a_exgl = cur.a_exgr = cur.b_exgl = cur.b_exgr = 0;
                    int i = it.n_im;
                    bool bad = false;
                    while (--i >= 0 && CP(i, 0) == END_ULK) ;
                    for ( ; i >= 0 && CP(i, 0) != END_ULK; --i) {
                        int c = 0;
                        cur.a_left = CP(i, c);
                        cur.b_exgl = CP(i, ++c);
                        cur.b_left = CP(i, ++c);
                        if (cur.a_right > a_len || cur.b_right > b_len || cur.a_left < 0 || cur.b_left < 0) { bad = true; break; }
                        if (cur.b_left < 0 || cur.b_left > cur.b_right) break;
                        while (c < 9 && CP(i, c + 1) < END_ULK) { ++c; push_rec(t, cur.a_left, CP(i, c)); }
                        ++c;
                        slab_window(cur, sc.sh, &CP(i + 1, 0));
                        if (!queue_trcbk(cur, fwd, scl, st.scalar_ok, t)) break;
                        cur.a_right = cur.a_left;
                        cur.b_right = CP(i, c - 1);
                    }
                    if (!bad && !t.cls && ((i < 0 && CP(0, 0) != END_ULK) || CP(0, 2) != END_ULK)) {
                        cur.a_left = aleft; cur.b_left = bleft;
                        slab_window(cur, sc.sh, &CP(0, 0));
                        queue_trcbk(cur, fwd, scl, st.scalar_ok, t);
                    }
                }
#undef CP
            }
        }
        // ---- traceback round; sub-problems below 8 rows go through the scalar engine
        for (int pass = 0; pass < 4; ++pass) {                // `_wip` forward, scalar forwardH_ng, -A1 forwardH1, scalar with a cut range
            std::vector<HItem>& list = pass ? scl : fwd;
            if (list.empty()) continue;
            std::vector<HItem> run;
            for (const HItem& it : list) {
                const int kind = it.cut_r > it.cut_l ? 3 : (it.exact ? 2 : 1);
                if (pass && kind != pass) continue;
                if (tops[it.slot].cls) continue;
                if (bad_range(it, st.probs[it.top])) { tops[it.slot].cls = 1; continue; }
                run.push_back(it);
            }
            if (pass == 0 || pass == 3) list.clear();
            if (run.empty()) continue;
            HFwdOut fo;
            if (pass ? run_scalar(st, run, true, fo, pass == 2, pass == 3) : run_forward(st, run, true, fo)) return -1;
            hs.fwd_ms += fo.sweep_ms; hs.fwd_cells += fo.cells;            // (every engine's launches: the exact-model passes too)
            for (size_t f = 0; f < run.size(); ++f) {
                HTop& t = tops[run[f].slot];
                if (run[f].first) t.score = fo.res[f].score;
                const int stt = fo.n_skl[f];
                if (stt == -2) { if (!t.flag) t.flag = -1; }
                else if (stt == -3) { if (!t.flag) t.flag = -2; }
                else if (stt == -5) { if (!t.flag) t.flag = -4; }
                if (stt == -2 || stt == -5) continue;
                t.rec.insert(t.rec.end(), fo.skl.begin() + fo.off[f], fo.skl.begin() + fo.off[f + 1]);
            }
        }
    }
    return 0;
}

static SpdpSkl* dup_skl(const std::vector<SpdpSkl>& v)
{
    if (v.empty()) return nullptr;
    SpdpSkl* p = (SpdpSkl*) malloc(v.size() * sizeof(SpdpSkl));
    memcpy(p, v.data(), v.size() * sizeof(SpdpSkl));
    return p;
}

// level 0: raw engine output; level 1: alignH_ng (header + stdskl3); level 2: lspH_ng (the ladder, records as written)
static int deliver(HStore& st, int level, SpdpAlignment* out, HStats& hs)
{
    std::vector<HTop> tops;
    if (run_ladder(st, level >= 1, tops, hs)) return -1;
    int rc = 0;
    std::vector<SpdpSkl> stdv, full;
    for (int i = 0; i < st.n; ++i) {
        HTop& t = tops[i];
        if (t.cls) rc = 1;
        if (!out) continue;
        out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0;
        if (t.cls) continue;
        out[i].score = t.score;
        if (t.flag) { out[i].n_skl = t.flag; continue; }
        if (level == 0 || level == 2) {
            out[i].n_skl = (int) t.rec.size();
            out[i].skl = dup_skl(t.rec);
        } else if (t.rec.size() >= 2) {                      // globalH_ng: fewer than 2 records = no alignment
            stdv = corner_list<3>(t.rec);                   // stdskl3 with UNITE_INDEL_FS = 0
            full.clear();
            SpdpSkl hd; hd.m = 1; hd.n = (int) stdv.size();
            full.push_back(hd);
            full.insert(full.end(), stdv.begin(), stdv.end());
            out[i].n_skl = (int) full.size();
            out[i].skl = dup_skl(full);
        }
    }
    return rc;
}

// ---- requests against a resident store (spdp_h_requests.h) --------------------------------
HStore* spdh_store_open(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n)
{
    HStore* st = new HStore();
    if (st->upload(ctx, sc, probs, n)) { delete st; return nullptr; }
    return st;
}
void spdh_store_close(HStore* st) { delete st; }

// out[k]: what the call returns + the Mfile records it writes, as written; n_skl < 0: not served here (an engine that is
// not built, a range outside the sequences) or undefined in the reference (spdp_align_h's flags)
int spdh_run_requests(HStore* st, const SpdhRequest* reqs, int n, SpdpAlignment* out, SpdpContext* lane)
{
    struct LaneScope { LaneScope(SpdpContext* l) { t_lane = l; } ~LaneScope() { t_lane = nullptr; } } scope(lane);
    LaneCopies own_stream;
    std::vector<HTop> tops;
    HStats hs;
    for (int k = 0; k < n; ++k) { out[k].score = SPDP_NEVSEL; out[k].n_skl = 0; out[k].skl = nullptr; out[k].flags = 0; out[k].reserved = 0; }
    if (n <= 0) return 0;
    if (run_ladder(*st, true, tops, hs, reqs, n)) return -1;
    for (int k = 0; k < n; ++k) {
        const HTop& t = tops[k];
        if (t.cls) { out[k].n_skl = -1; continue; }
        out[k].score = t.score;
        if (t.flag) { out[k].n_skl = t.flag; continue; }
        out[k].n_skl = (int) t.rec.size();
        out[k].skl = dup_skl(t.rec);
    }
    return 0;
}

// ---- C ABI ------------------------------------------------------------------------------
int spdp_wip_forward_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                       SpdpAlignment* out)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !out) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    HStats hs;
    return deliver(st, 0, out, hs);
}

int spdp_wip_udh_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs, int n_im,
                   int32_t* scores, int32_t* cpos, int32_t* ranges)
{
    if (!ctx || !sc || !probs || n_probs < 0 || n_im < 1 || !scores || !cpos || !ranges) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<HItem> items;
    for (int i = 0; i < n_probs; ++i) {
        HItem it = item_of(probs[i], i, sc->sh);
        it.n_im = n_im;
        if (it.a_right - it.a_left < 16 * 1 || it.w.width < 0) { ctx->err = "hirschbergH1_wip: problem too small"; return -1; }
        items.push_back(it);
    }
    HUdhOut uo;
    std::vector<int> lfl;
    if (sc->local ? run_scalar_udh(st, items, uo, lfl, 2) : run_udh(st, items, uo)) return -1;   // -LS: spdh_local_udh
    for (int i = 0; i < n_probs; ++i) {
        scores[i] = uo.scores[i];
        memcpy(cpos + (size_t) i * (n_im + 1) * 10, &uo.cpos[(size_t) i * uo.stride], (size_t) (n_im + 1) * 10 * sizeof(int32_t));
        memcpy(ranges + 4 * i, &uo.ranges[4 * i], 4 * sizeof(int32_t));
    }
    return 0;
}

int spdp_homscore_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                    int32_t* scores)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !scores) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<HItem> items, sitems;
    std::vector<int> idx, sidx;
    int rc = 0;
    for (int i = 0; i < n_probs; ++i) {
        scores[i] = SPDP_NEVSEL;
        HItem it = item_of(probs[i], i, sc->sh);
        const int m = it.a_right - it.a_left, n = it.b_right - it.b_left;
        if (!n || !m || it.w.width < 0) { rc = 1; continue; }
        // -A1 above 7 rows: the reference runs forwardH1 without a Vmf here and stops with SIGSEGV (Sjsites::get
        // dereferences it, src/fwd2h1_simd.h:431-434 reached through hfesmc of a mode-1 object): not computed
        if (sc->scalar_engines == 2 && m >= 8) { rc = 1; continue; }
        if (sc->scalar_engines || m < 8) {                            // scalar forwardH_ng (src/fwd2h1.cc:3297)
            if (!st.scalar_ok) { rc = 1; continue; }
            sitems.push_back(it); sidx.push_back(i);
            continue;
        }
        items.push_back(it); idx.push_back(i);
    }
    HFwdOut fo;
    if (run_forward(st, items, false, fo)) return -1;
    for (size_t f = 0; f < items.size(); ++f) scores[idx[f]] = fo.res[f].score;
    if (run_scalar(st, sitems, false, fo)) return -1;
    for (size_t f = 0; f < sitems.size(); ++f) scores[sidx[f]] = fo.res[f].score;
    return rc;
}

int spdp_scalar_forward_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                          int traceback, SpdpAlignment* out)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !out) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<HItem> items;
    std::vector<int> idx;
    for (int i = 0; i < n_probs; ++i) {
        out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0;
        HItem it = item_of(probs[i], i, sc->sh);
        if (it.w.width < 0) continue;                                 // trcbkalignH_ng returns NEVSEL
        items.push_back(it); idx.push_back(i);
    }
    HFwdOut fo;
    const bool exact = sc->scalar_engines == 2;                       // forwardH1 (-A1): traceback form only
    if (exact && !traceback) { ctx->err = "forwardH1 (-A1) has no score-only form (the reference's stops)"; return -1; }
    if (run_scalar(st, items, traceback != 0, fo, exact)) return -1;
    for (size_t f = 0; f < items.size(); ++f) {
        SpdpAlignment& o = out[idx[f]];
        o.score = fo.res[f].score;
        if (!traceback) continue;
        if (fo.n_skl[f] < 0) { o.n_skl = fo.n_skl[f] == -5 ? -4 : fo.n_skl[f]; continue; }    // -3: undefined in the reference (mode 3 pointer lanes)
        std::vector<SpdpSkl> rec(fo.skl.begin() + fo.off[f], fo.skl.begin() + fo.off[f + 1]);
        o.n_skl = (int) rec.size();
        o.skl = dup_skl(rec);
    }
    return 0;
}

int spdp_scalar_udh_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                      int n_im, int imd_intvl, int32_t* scores, int32_t* cpos, int32_t* ranges, int32_t* flags)
{
    if (!ctx || !sc || !probs || n_probs < 0 || n_im < 1 || imd_intvl < 1 || !scores || !cpos || !ranges || !flags) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<HItem> items;
    for (int i = 0; i < n_probs; ++i) {
        HItem it = item_of(probs[i], i, sc->sh);
        it.n_im = n_im; it.imd_intvl = imd_intvl;
        if (it.w.width < 0 || it.a_left + (int64_t) n_im * imd_intvl > it.a_right + imd_intvl) {
            ctx->err = "hirschbergH_ng: intermediate rows beyond the query range"; return -1;
        }
        items.push_back(it);
    }
    HUdhOut uo;
    std::vector<int> fl;
    if (run_scalar_udh(st, items, uo, fl, sc->scalar_engines == 2 ? 1 : 0)) return -1;  // 2: hirschbergH1 (imd_intvl unused)
    for (int i = 0; i < n_probs; ++i) {
        scores[i] = uo.scores[i];
        flags[i] = fl[i];
        memcpy(cpos + (size_t) i * (n_im + 1) * 10, &uo.cpos[(size_t) i * uo.stride], (size_t) (n_im + 1) * 10 * sizeof(int32_t));
        memcpy(ranges + 4 * i, &uo.ranges[4 * i], 4 * sizeof(int32_t));
    }
    return 0;
}

int spdp_align_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                 SpdpAlignment* out)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !out) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    HStats hs;
    return deliver(st, 1, out, hs);
}

// Aln2h1::lspH_ng (src/fwd2h1.cc:2134-2230) for a caller that keeps the record file itself -- the seeded path's
// interpolateH (:3106-3120): the ladder below spdp_align_h, records as written (no header, no stdskl3)
int spdp_lsp_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs, SpdpAlignment* out)
{
    if (!ctx || !sc || !probs || n_probs < 0 || !out) return -1;
    HStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    HStats hs;
    return deliver(st, 2, out, hs);
}

struct SpdpBatchH {
    HStore st;
    HStats last;
    int64_t cells = 0;
};

SpdpBatchH* spdp_batch_upload_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs)
{
    if (!ctx || !sc || !probs || n_probs < 0) return nullptr;
    SpdpBatchH* bt = new SpdpBatchH();
    if (bt->st.upload(ctx, sc, probs, n_probs)) { delete bt; return nullptr; }
    for (int i = 0; i < n_probs; ++i) {
        SpdpWindow w;
        spdp_stripe31(&probs[i], sc->sh, &w);
        bt->cells += spdp_cells_h(&probs[i], &w);
    }
    return bt;
}
void spdp_batch_free_h(SpdpBatchH* bt) { delete bt; }
int64_t spdp_batch_cells_h(const SpdpBatchH* bt) { return bt ? bt->cells : 0; }

int spdp_batch_align_h(SpdpBatchH* bt, SpdpAlignment* out, float* kernel_ms, int64_t* kernel_cells)
{
    if (!bt) return -1;
    bt->last = HStats();
    const int rc = deliver(bt->st, 1, out, bt->last);
    if (kernel_ms) *kernel_ms = bt->last.fwd_ms + bt->last.udh_ms;
    if (kernel_cells) *kernel_cells = bt->last.fwd_cells + bt->last.udh_cells;
    return rc;
}
