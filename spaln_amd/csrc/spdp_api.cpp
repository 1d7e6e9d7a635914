// spdp_api.cpp -- host side of libspdp_hip.so: the extern "C" surface declared in
// include/spdp.h.  Packs problems into the HBM layout of spdp_dev.h and launches
// the sweeps of spdp_kernels.hip; the reference's dispatch logic around the
// engines (Aln2s1::lspS_ng & co., src/fwd2s1.cc:1667-1897) lives in
// spdp_host.cpp.  There is no CPU compute path: without a HIP device
// spdp_create() fails.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spdp.h"
#include "spdp_dev.h"
#include "spdp_internal.h"

// ---- small helpers --------------------------------------------------------------
static void stripe_rng(int a_left, int a_right, int b_left, int b_right, int sh, SpdpWindow* w)
{   // stripe(), src/aln2.cc:156-176 (cmode 3)
    if (sh < 0) {
        int shorter = std::min(a_right - a_left, b_right - b_left);
        sh = -sh * shorter / 100;
    }
    w->up = b_right - a_right;
    w->lw = b_left - a_left;
    if (w->up < w->lw) std::swap(w->up, w->lw);
    w->up += sh;
    w->lw -= sh;
    int q;
    if ((q = b_right - a_left) < w->up) w->up = q;
    if ((q = b_left - a_right) > w->lw) w->lw = q;
    w->width = w->up - w->lw + 3;
}

void spdp_stripe(const SpdpProblem* p, int sh, SpdpWindow* w)
{
    stripe_rng(p->a_left, p->a_right, p->b_left, p->b_right, sh, w);
}

int64_t spdp_cells_w(int a_left, int a_right, int b_left, int b_right, const SpdpWindow& w)
{   // cells visited by the reference loops, src/fwd2s1.cc:249-256:
    //   sum over m in (a_left, a_right] of max(0, min(m + up + 1, b_right) - max(m + lw, b_left)),
    // evaluated piecewise (both clamps switch once along m)
    auto seg = [&](int64_t m0, int64_t m1, bool hi_clamped, bool lo_clamped) -> int64_t {
        if (m1 < m0) return 0;                           // rows m0 .. m1 inclusive
        const int64_t cnt = m1 - m0 + 1, sm = (m0 + m1) * cnt / 2;
        // width(m) = (hi_clamped ? b_right : m + up + 1) - (lo_clamped ? b_left : m + lw)
        int64_t tot = 0;
        tot += hi_clamped ? (int64_t) b_right * cnt : sm + (int64_t) (w.up + 1) * cnt;
        tot -= lo_clamped ? (int64_t) b_left * cnt : sm + (int64_t) w.lw * cnt;
        return tot;
    };
    const int64_t mh = (int64_t) b_right - w.up - 1;     // m >= mh: upper end clamped to b_right
    const int64_t ml = (int64_t) b_left - w.lw;          // m <= ml: lower end clamped to b_left
    int64_t c = 0;
    int64_t cuts[4] = {a_left + 1, std::min<int64_t>(std::max<int64_t>(mh, a_left + 1), a_right + 1),
                       std::min<int64_t>(std::max<int64_t>(ml + 1, a_left + 1), a_right + 1), a_right + 1};
    std::sort(cuts, cuts + 4);
    for (int i = 0; i < 3; ++i) {
        const int64_t m0 = cuts[i], m1 = cuts[i + 1] - 1;
        if (m1 < m0) continue;
        const bool hi = m0 >= mh, lo = m1 <= ml;
        // within a segment both flags are constant; widths are linear, so a negative width can only
        // occur over a whole prefix/suffix -- fall back to the exact loop in that rare case
        const int64_t w0 = (hi ? b_right : m0 + w.up + 1) - (lo ? b_left : m0 + w.lw);
        const int64_t w1 = (hi ? b_right : m1 + w.up + 1) - (lo ? b_left : m1 + w.lw);
        if (w0 >= 0 && w1 >= 0) c += seg(m0, m1, hi, lo);
        else
            for (int64_t m = m0; m <= m1; ++m) {
                const int64_t n1 = std::max<int64_t>(m + w.lw, b_left), n9 = std::min<int64_t>(m + w.up + 1, b_right);
                if (n9 > n1) c += n9 - n1;
            }
    }
    return c;
}

int64_t spdp_cells(const SpdpProblem* p, const SpdpWindow* w)
{
    return spdp_cells_w(p->a_left, p->a_right, p->b_left, p->b_right, *w);
}

RunItem spdp_item_of(const SpdpProblem& p, int parent, int sh)
{
    RunItem it;
    it.parent = parent;
    it.a_left = p.a_left; it.a_right = p.a_right; it.b_left = p.b_left; it.b_right = p.b_right;
    it.a_exgl = p.a_exgl; it.a_exgr = p.a_exgr; it.b_exgl = p.b_exgl; it.b_exgr = p.b_exgr;
    stripe_rng(p.a_left, p.a_right, p.b_left, p.b_right, sh, &it.w);
    it.n_im = 0;
    return it;
}

// ---- context ----------------------------------------------------------------------
SpdpContext* spdp_create(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "spdp_create: no HIP device (this library has no CPU path)\n");
        return nullptr;
    }
    if (device < 0 || device >= ndev) return nullptr;
    SpdpContext* ctx = new SpdpContext();
    ctx->device = device;
    // the main stream carries the latency-bound launches (linear-space rounds, slab tracebacks): highest priority;
    // the side stream (a big forward sweep beside them) the lowest
    int prio_lo = 0, prio_hi = 0;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return nullptr; }
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&ctx->stream, hipStreamDefault, prio_hi) != hipSuccess) {
        delete ctx;
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->n_cu = prop.multiProcessorCount;
        ctx->name = std::string(prop.name) + " " + prop.gcnArchName;
    }
    if (ctx->n_cu <= 0) ctx->n_cu = 256;
    (void) hipEventCreate(&ctx->ev0);
    (void) hipEventCreate(&ctx->ev1);
    if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_lo) != hipSuccess) ctx->stream2 = nullptr;
    (void) hipEventCreate(&ctx->ev2);
    (void) hipEventCreate(&ctx->ev3);
    return ctx;
}

void* DevPool::get(int slot, size_t bytes)
{
    if (bytes < 256) bytes = 256;
    if (cap[slot] >= bytes) return ptr[slot];
    if (ptr[slot]) (void) hipFree(ptr[slot]);
    ptr[slot] = nullptr; cap[slot] = 0;
    size_t want = bytes + bytes / 8;                    // head-room: batches of one run vary a little
    if (hipMalloc(&ptr[slot], want) != hipSuccess) {
        if (hipMalloc(&ptr[slot], bytes) != hipSuccess) return nullptr;
        want = bytes;
    }
    cap[slot] = want;
    return ptr[slot];
}

void DevPool::release()
{
    for (int i = 0; i < N_SLOTS; ++i) { if (ptr[i]) (void) hipFree(ptr[i]); ptr[i] = nullptr; cap[i] = 0; }
}

thread_local bool t_lane_copies = false;         // spdp_internal.h, spdp_copy_sync

// lane i of a context: lane 0 is the context itself, the others are contexts of their own (streams, events, pools)
// on the same device, created on first use and owned by the parent
SpdpContext* spdp_lane(SpdpContext* ctx, int i)
{
    if (i <= 0) return ctx;
    while ((int) ctx->lanes.size() < i) {
        SpdpContext* l = spdp_create(ctx->device);
        if (!l) return nullptr;
        ctx->lanes.push_back(l);
    }
    return ctx->lanes[i - 1];
}

void spdp_destroy(SpdpContext* ctx)
{
    if (!ctx) return;
    for (SpdpContext* l : ctx->lanes) spdp_destroy(l);
    ctx->lanes.clear();
    (void) hipSetDevice(ctx->device);
    for (DevPool& p : ctx->pool) p.release();
    for (int k = 0; k < 3; ++k) if (ctx->stage_ptr[k]) (void) hipHostFree(ctx->stage_ptr[k]);
    (void) hipEventDestroy(ctx->ev0);
    (void) hipEventDestroy(ctx->ev1);
    (void) hipEventDestroy(ctx->ev2);
    (void) hipEventDestroy(ctx->ev3);
    if (ctx->stream2) (void) hipStreamDestroy(ctx->stream2);
    (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

void* SpdpContext::staging(int k, size_t bytes)
{
    if (bytes <= stage_cap[k]) return stage_ptr[k];
    if (stage_ptr[k]) (void) hipHostFree(stage_ptr[k]);
    stage_ptr[k] = nullptr; stage_cap[k] = 0;
    const size_t want = bytes + bytes / 8;
    if (hipHostMalloc(&stage_ptr[k], want, hipHostMallocDefault) != hipSuccess) { stage_ptr[k] = nullptr; return nullptr; }
    stage_cap[k] = want;
    return stage_ptr[k];
}

const char* spdp_last_error(const SpdpContext* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int spdp_device_name(const SpdpContext* ctx, char* buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return -1;
    snprintf(buf, buflen, "%s (%d CUs)", ctx->name.c_str(), ctx->n_cu);
    return 0;
}

// ---- resident inputs ----------------------------------------------------------------
static void to_dev_scoring(const SpdpScoring* sc, DevScoring* d)
{
    memset(d, 0, sizeof *d);
    d->mtx_dim = sc->mtx_dim;
    d->gop = sc->gop; d->gep = sc->gep;
    d->spj = sc->spj; d->llmt = sc->llmt;
    d->nquant = std::max(1, std::min(sc->nquant, SPDP_MAX_QUANT));
    d->local = sc->local ? 1 : 0;
    for (int j = 0; j < SPDP_MAX_QUANT; ++j) { d->qm_len[j] = sc->qm_len[j]; d->qm_pen[j] = sc->qm_pen[j]; }
    for (int i = 1; i < sc->mtx_dim && i < 32; ++i)         // row / column 0 stay zero
        for (int j = 1; j < sc->mtx_dim && j < 32; ++j)
            d->mtx[i * 32 + j] = sc->mtx[i * sc->mtx_dim + j];
}

void DevStore::release()
{
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    if (d_sc) (void) hipFree(d_sc);
    if (d_a) (void) hipFree(d_a);
    if (d_cols) (void) hipFree(d_cols);
    if (d_aux) (void) hipFree(d_aux);
    if (d_intpen) (void) hipFree(d_intpen);
    if (d_ipen_runs) (void) hipFree(d_ipen_runs);
    d_ipen_runs = nullptr;
    if (d_cip) (void) hipFree(d_cip);
    d_sc = d_a = d_cols = d_aux = d_intpen = d_cip = nullptr;
}

int DevStore::upload(SpdpContext* c, const SpdpScoring* scp, const SpdpProblem* probs, int n)
{
    ctx = c; sc = *scp; n_parents = n;
    (void) hipSetDevice(ctx->device);
    // double affine gaps (Noll = 3, -yl3): forwardS_ng / hirschbergS_ng / scorealoneS_ng (the -A0 engines, spdp_rowwave<., ., ., DAGP>,
    // spdp_rowwave_udh<., DAGP>); the -A1 / -A2 / -A3 engines and the seeded walk's cut range refuse it (DevRun::prepare)
    if (sc.noll != 2 && !(sc.noll == 3 && (sc.scalar_engines == 1 || sc.scalar_engines == 2))) {
        ctx->err = "double affine gaps (Noll = 3) are built for the -A0 and -A1 engines (SpdpScoring.scalar_engines = 1 / 2); Noll must be 2 or 3";
        return -1;
    }
    // GapPenalty(1) of the first column is BasicGOP + BasicGEP only while codonk1 >= 1 (src/aln.h:275-282); a caller that leaves
    // the field 0 would silently get scores that are not the reference's
    if (sc.noll == 3 && sc.codonk1 < 1) { ctx->err = "Noll = 3 needs SpdpScoring.codonk1 >= 1 (alprm2.k1 of the reference)"; return -1; }
    if (sc.mtx_dim < 1 || sc.mtx_dim > 32) { ctx->err = "mtx_dim out of range"; return -1; }
    a_off.resize(n); col_off.resize(n); a_len.resize(n); b_len.resize(n);
    int64_t a_tot = 0, col_tot = 0;
    for (int i = 0; i < n; ++i) {
        const SpdpProblem& p = probs[i];
        if (p.a_right > p.a_len || p.b_right > p.b_len || p.a_left < 0 || p.b_left < 0 ||
            p.a_right < p.a_left || p.b_right < p.b_left) { ctx->err = "bad problem ranges"; return -1; }
        a_len[i] = p.a_len; b_len[i] = p.b_len;
        a_off[i] = a_tot; a_tot += ((int64_t) p.a_len + 15) & ~15ll;
        col_off[i] = col_tot; col_tot += (int64_t) p.b_len + 1 + SPDP_COL_PAD;
    }
    std::vector<uint8_t> ha(std::max<int64_t>(a_tot, 16), 0);
    // signals: either every problem brings sig5 / sig3 (then also cano5 / cano3 / dinc for the exact engines), or none
    // does and SpdpScoring::sigmodel says how to compute them from the codes -- on the device (spdp_signals.hip)
    int n_sig = 0;
    for (int i = 0; i < n; ++i) n_sig += (probs[i].sig5 && probs[i].sig3) ? 1 : 0;
    const bool dev_sig = n > 0 && n_sig == 0 && sc.sigmodel;
    if (n_sig != n && !dev_sig) {
        ctx->err = n_sig ? "sig5 / sig3 missing for part of the batch" : "sig5 / sig3 missing and no SpdpScoring::sigmodel";
        return -1;
    }
    has_exact = sc.intpen && sc.intpen_len > 0;
    for (int i = 0; i < n && has_exact && !dev_sig; ++i)
        if (!probs[i].cano5 || !probs[i].cano3 || !probs[i].dinc) has_exact = false;
    int max_s5 = INT32_MIN, max_s3 = INT32_MIN;
    for (int i = 0; i < n; ++i) memcpy(ha.data() + a_off[i], probs[i].a, probs[i].a_len);
    HIPCHK(hipMalloc(&d_cols, 2 * std::max<int64_t>(col_tot, 1) * sizeof(int32_t)));
    if (has_exact) HIPCHK(hipMalloc(&d_aux, 2 * std::max<int64_t>(col_tot, 1)));
    if (dev_sig) {
        // codes only over PCIe (1 B per position instead of 8 + 2): base i of problem p at hb[col_off + i]
        std::vector<uint8_t> hb(std::max<int64_t>(col_tot, 16), 0);
        std::vector<SigJob> jobs(n);
        for (int i = 0; i < n; ++i) {
            memcpy(hb.data() + col_off[i], probs[i].b, probs[i].b_len);
            SigJob& J = jobs[i];
            J.b_off = J.col_off = col_off[i]; J.out_off = 0; J.pad = 0;
            J.b_len = probs[i].b_len; J.left = probs[i].b_left; J.right = probs[i].b_right;
            if (probs[i].exin_left || probs[i].exin_right) {      // the range the reference's Exinon covers, not the call's sub-range
                if (probs[i].exin_left < 0 || probs[i].exin_right > probs[i].b_len || probs[i].exin_left > probs[i].b_left ||
                    probs[i].exin_right < probs[i].b_right) { ctx->err = "exin_left / exin_right must enclose b_left / b_right"; return -1; }
                J.left = probs[i].exin_left; J.right = probs[i].exin_right;
            }
        }
        void* d_b = nullptr;
        HIPCHK(hipMalloc(&d_b, hb.size()));
        hipError_t e = hipMemcpyAsync(d_b, hb.data(), hb.size(), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_cols, 0, 2 * std::max<int64_t>(col_tot, 1) * sizeof(int32_t), ctx->stream);
        if (e == hipSuccess && has_exact) e = hipMemsetAsync(d_aux, 0, 2 * std::max<int64_t>(col_tot, 1), ctx->stream);
        int rc = -1;
        if (e == hipSuccess) {
            SignalArgs A;
            memset(&A, 0, sizeof A);
            A.codes = (const uint8_t*) d_b;
            A.cols = (int2*) d_cols; A.aux = has_exact ? (uchar2*) d_aux : nullptr;
            A.ipen = sc.ipen; A.spj = sc.spj;
            rc = spdp_signals_run(ctx, sc.sigmodel, jobs, A, &max_s5, &max_s3);
        } else {
            ctx->err = std::string("signal upload: ") + hipGetErrorString(e);
        }
        (void) hipFree(d_b);
        if (rc) return -1;
    } else {
        // column records (8 B per genomic position) and class bytes, packed by all host cores into pinned staging
        // memory the context keeps (a 10 k batch is ~1 GB: one thread and pageable memory took 430 ms, 40 % of a step)
        int n_thr = spdp_host_cpus();
        if (const char* e = getenv("SPDP_UPLOAD_THREADS")) n_thr = atoi(e);
        n_thr = std::max(1, std::min(std::min(n_thr, 32), n));
        std::vector<int> t_s5(n_thr, INT32_MIN), t_s3(n_thr, INT32_MIN);
        // the copy of a group of problems (~8 M positions) starts as soon as the group is packed, under the packing of the next ones;
        // the groups go through a ring of four slots of the staging memory (a slot is packed again once its copy has left): a map +
        // align call holds 5 * 10^8 positions, which was 5 GB of pinned memory when every group had a place of its own
        std::vector<int> grp_first, grp_of(n);
        int64_t slot_cap = 1;
        {
            int64_t acc = 0;
            for (int i = 0; i < n; ++i) {
                if (i == 0 || acc >= (8 << 20)) { grp_first.push_back(i); acc = 0; }
                grp_of[i] = (int) grp_first.size() - 1;
                acc += (int64_t) probs[i].b_len + 1 + SPDP_COL_PAD;
                slot_cap = std::max(slot_cap, acc);
            }
        }
        const int n_grp = (int) grp_first.size();
        const int ring = (getenv("SPDP_UPLOAD_RING") && atoi(getenv("SPDP_UPLOAD_RING")) == 0) ? std::max(1, n_grp) : 4;     // (0: a place per group)
        const size_t nc = 2 * (size_t) slot_cap * (size_t) ring;
        int32_t* hc = (int32_t*) ctx->staging(0, nc * sizeof(int32_t));
        uint8_t* hx = has_exact ? (uint8_t*) ctx->staging(1, nc) : nullptr;
        if (!hc || (has_exact && !hx)) { ctx->err = "out of pinned host memory"; return -1; }
        auto stage_of = [&](int i) -> int64_t { const int g = grp_of[i]; return (int64_t) (g % ring) * slot_cap + (col_off[i] - col_off[grp_first[g]]); };
        std::atomic<int> copied{0};                     // groups whose copy has left the staging
        std::vector<std::atomic<int>> grp_done(n_grp);
        for (auto& g : grp_done) g.store(0);
        std::atomic<int> next_prob{0};
        auto pack = [&](int t) {
            int m5 = INT32_MIN, m3 = INT32_MIN;
            for (;;) {
                const int i = next_prob.fetch_add(1);
                if (i >= n) break;
                const SpdpProblem& p = probs[i];
                while (copied.load(std::memory_order_acquire) < grp_of[i] - ring + 1) std::this_thread::yield();
                if (has_exact) {
                    uint8_t* x = hx + 2 * stage_of(i);
                    for (int nn = 0; nn <= p.b_len; ++nn, x += 2) {
                        x[0] = (p.cano5[nn] ? 1 : 0) | (p.cano3[nn] ? 2 : 0);
                        x[1] = p.dinc[nn];
                    }
                    memset(x, 0, 2 * SPDP_COL_PAD);
                }
                int32_t* cr = hc + 2 * stage_of(i);
                for (int nn = 0; nn <= p.b_len; ++nn, cr += 2) {
                    const uint16_t s5 = (uint16_t) (int16_t) (p.sig5[nn] + sc.ipen);
                    const uint16_t s3 = (uint16_t) p.sig3[nn];
                    cr[0] = sc.spj ? (int32_t) ((uint32_t) s5 | ((uint32_t) s3 << 16)) : 0;
                    m5 = std::max(m5, (int) (int16_t) s5); m3 = std::max(m3, (int) (int16_t) s3);
                    cr[1] = nn > 0 ? p.b[nn - 1] : 0;
                }
                memset(cr, 0, 2 * SPDP_COL_PAD * sizeof(int32_t));
                grp_done[grp_of[i]].fetch_add(1, std::memory_order_release);
            }
            t_s5[t] = m5; t_s3[t] = m3;
        };
        {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; ++t) th.emplace_back(pack, t);
            hipError_t ce = hipSuccess;
            std::vector<hipEvent_t> left((size_t) ring, nullptr);
            for (int k = 0; k < ring && ce == hipSuccess && n_grp > ring; ++k) ce = hipEventCreateWithFlags(&left[k], hipEventDisableTiming);
            for (int g = 0; g < n_grp && ce == hipSuccess; ++g) {
                const int first = grp_first[g], last = g + 1 < n_grp ? grp_first[g + 1] : n;
                while (grp_done[g].load(std::memory_order_acquire) < last - first) std::this_thread::yield();
                const int64_t c0 = col_off[first], c1 = last < n ? col_off[last] : col_tot;
                const int64_t s0 = (int64_t) (g % ring) * slot_cap;
                ce = hipMemcpyAsync((int32_t*) d_cols + 2 * c0, hc + 2 * s0, (size_t) (c1 - c0) * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
                if (ce == hipSuccess && has_exact)
                    ce = hipMemcpyAsync((uint8_t*) d_aux + 2 * c0, hx + 2 * s0, (size_t) (c1 - c0) * 2, hipMemcpyHostToDevice, ctx->stream);
                if (n_grp > ring) {                     // (a call of up to four groups never waits: every group has its slot)
                    if (ce == hipSuccess) ce = hipEventRecord(left[g % ring], ctx->stream);
                    if (ce == hipSuccess && g >= 1) { ce = hipEventSynchronize(left[(g - 1) % ring]); copied.store(g, std::memory_order_release); }
                }
            }
            copied.store(INT32_MAX, std::memory_order_release);               // (on an error too: no packer may wait for ever)
            for (std::thread& t : th) t.join();
            if (ce != hipSuccess) (void) hipStreamSynchronize(ctx->stream);   // copies already issued still read the staging
            for (hipEvent_t e : left) if (e) (void) hipEventDestroy(e);
            HIPCHK(ce);
        }
        for (int t = 0; t < n_thr; ++t) { max_s5 = std::max(max_s5, t_s5[t]); max_s3 = std::max(max_s3, t_s3[t]); }
        HIPCHK(hipStreamSynchronize(ctx->stream));          // the staging buffers are the context's: one upload at a time
    }
    sc.sigmodel = nullptr;                                   // the caller's model is not ours to keep
    // bounds for the fp32 sweeps (DevRun::build): best substitution score, best net gain of one intron
    fp_maxpos = 1;
    for (int i = 0; i < sc.mtx_dim * sc.mtx_dim; ++i) fp_maxpos = std::max(fp_maxpos, (int) sc.mtx[i]);
    {
        int max_pen = INT32_MIN;
        for (int j = 0; j < std::max(1, std::min(sc.nquant, SPDP_MAX_QUANT)); ++j) max_pen = std::max(max_pen, (int) sc.qm_pen[j]);
        fp_gain = (sc.spj && max_s5 > INT32_MIN) ? std::max(0, max_s5 + max_s3 + max_pen) : 0;
    }
    // conserved-intron bonuses of the queries that carry them (SpdpProblem::cip), one row of a_len + 1 ints each
    cip_off.assign(n, -1);
    {
        std::vector<int32_t> hcip;
        for (int i = 0; i < n; ++i)
            if (probs[i].cip) {
                cip_off[i] = (int32_t) hcip.size();
                hcip.insert(hcip.end(), probs[i].cip, probs[i].cip + probs[i].a_len + 1);
            }
        if (!hcip.empty()) {
            HIPCHK(hipMalloc(&d_cip, hcip.size() * sizeof(int32_t)));
            HIPCHK(hipMemcpy(d_cip, hcip.data(), hcip.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    DevScoring hsc;
    to_dev_scoring(&sc, &hsc);
    HIPCHK(hipMalloc(&d_sc, sizeof(DevScoring)));
    HIPCHK(hipMalloc(&d_a, ha.size()));
    HIPCHK(hipMemcpyAsync(d_sc, &hsc, sizeof hsc, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_a, ha.data(), ha.size(), hipMemcpyHostToDevice, ctx->stream));
    if (has_exact) {
        HIPCHK(hipMalloc(&d_intpen, sizeof(int16_t) * sc.intpen_len));
        HIPCHK(hipMemcpyAsync(d_intpen, sc.intpen, sizeof(int16_t) * sc.intpen_len, hipMemcpyHostToDevice, ctx->stream));
        std::vector<int16_t> runs(SPDP_IPR_WORDS);
        if (spdp_intpen_runs(sc.intpen, sc.intpen_len, runs.data())) {
            HIPCHK(hipMalloc(&d_ipen_runs, sizeof(int16_t) * runs.size()));
            HIPCHK(hipMemcpy(d_ipen_runs, runs.data(), sizeof(int16_t) * runs.size(), hipMemcpyHostToDevice));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- one sweep over a list of items ---------------------------------------------------
static int stripe_blocks(const DevProblem& P, int s, bool forward)
{
    const int ml = P.a_left + s * SPDP_NELEM;
    const int j9 = std::min(SPDP_NELEM, P.a_right - ml);
    const int n_start = std::max(P.b_left, P.lw + ml);
    const int n9 = std::min(P.b_right, P.up + (ml + j9) + 1) + j9;
    const int len = std::max(0, n9 + (forward ? 1 : 0) - n_start);
    return (len + 15) >> 4;
}

// buffers belong to ctx->pool[..]; a run abandoned in flight (an error elsewhere in the batch) is waited for so
// that its pool can be reused
void DevRun::release() { if (in_flight && ctx) { (void) hipStreamSynchronize(strm()); in_flight = false; } }

#define POOLGET(dst, slot, bytes)                                                        \
    do {                                                                                 \
        (dst) = ctx->pool[side ? 8 : (flav == 7 ? 3 : (flav >= 5 ? 4 : flav))].get((slot), (size_t) (bytes)); \
        if (!(dst)) { ctx->err = "out of device memory"; return -1; }                    \
    } while (0)

// Vmf record budget of one forwardS_ng / forwardS1 call.  A record is written where a diagonal run starts and twice per
// accepted intron: on real alignments a few per row.  The budget is one record per two band cells (the band, not the
// bounding rectangle), at least 64 per row; a call that outgrows it reports so (n_skl = -3) and is run again with
// vmf_scale x 8 by the ladder -- the old "two per cell of the rectangle" made batches of slabs run in dozens of groups.
bool spdp_intpen_runs(const int16_t* intpen, int len, int16_t* out)
{
    if (len <= SPDP_IPR_BASE) {                         // nothing beyond the LDS table: one run
        if (len <= 0) return false;
        memset(out, 0, sizeof(int16_t) * SPDP_IPR_WORDS);
        uint16_t* st = (uint16_t*) out;
        st[0] = SPDP_IPR_BASE; st[1] = 65535;
        out[SPDP_IPR_RUNS + 1] = intpen[len - 1];
        return true;
    }
    if (len > 65536) return false;
    uint16_t* start = (uint16_t*) out;
    int16_t* val = out + SPDP_IPR_RUNS + 1;
    uint8_t* span = (uint8_t*) (out + SPDP_IPR_RUNS + 1 + SPDP_IPR_RUNS);
    int nr = 0;
    start[0] = SPDP_IPR_BASE; val[0] = intpen[SPDP_IPR_BASE];
    for (int i = SPDP_IPR_BASE + 1; i < len; ++i)
        if (intpen[i] != intpen[i - 1]) {
            if (++nr >= SPDP_IPR_RUNS) return false;
            start[nr] = (uint16_t) i; val[nr] = intpen[i];
        }
    for (int j = nr + 1; j <= SPDP_IPR_RUNS; ++j) start[j] = 65535;
    for (int j = nr + 1; j < SPDP_IPR_RUNS; ++j) val[j] = val[nr];
    int j = 0;
    for (int b = 0; b < SPDP_IPR_SPANS; ++b) {
        const int lo = SPDP_IPR_BASE + 64 * b, hi = lo + 63;
        while (j < nr && (int) start[j + 1] <= lo) ++j;
        span[b] = (uint8_t) j;
        if (j + 2 <= nr && (int) start[j + 2] <= hi) return false;      // two steps inside one span
    }
    return true;
}

static int64_t vmf_capacity(const RunItem& it)
{
    const int64_t rows = it.a_right - it.a_left + 1, cols = it.b_right - it.b_left + 1;
    const int64_t band = rows * std::min<int64_t>(cols, (int64_t) it.w.width);
    const int64_t full = 2 * rows * cols + 64;
    return std::min(full, std::max(band / 2, 64 * rows) * (int64_t) it.vmf_scale + 64);
}

int DevRun::build(const DevStore* st, const std::vector<RunItem>& items, int flav)
{
    store = st; ctx = use_ctx ? use_ctx : st->ctx; flavour = flav; n = (int) items.size();
    cut = false;
    (void) hipSetDevice(ctx->device);
    if (flav >= 3 && !st->has_exact) {
        ctx->err = "scalar exact engine needs intpen / t53 in SpdpScoring and cano5 / cano3 / dinc per problem";
        return -1;
    }
    if (st->sc.noll == 3 && !(flav >= 3 && flav <= 7)) {
        // (-A1: scoreonlyS1 / forwardS1 since round 5; the reference's own hirschbergS1 is not usable under -yl3 -- it never lets
        //  the second vertical gap reach H across an intermediate row -- so there is no result to be identical with)
        ctx->err = flav == 8 ? "double affine gaps (Noll = 3) under -A1: the linear-space engine (hirschbergS1) is undefined in the reference; "
                               "raise SpdpScoring.max_vmf_space so that the traceback branch is taken"
                             : "double affine gaps (Noll = 3): only the -A0 and -A1 engines are built";
        return -1;
    }
    // hirschbergS1_wip with local ends (-LS): its own kernel (spdp_local_udh.hip), flavour 9
    if (flav == 2 && st->sc.local) flav = flavour = 9;
    h_probs.assign(n, DevProblem());
    int64_t bnd_tot = 0, tb_tot = 0, imd_tot = 0;
    total_cells = 0; max_n_im = 0; max_skl = 0;
    for (int i = 0; i < n; ++i) {
        const RunItem& it = items[i];
        DevProblem& P = h_probs[i];
        if (it.parent < 0 || it.parent >= st->n_parents || it.a_left < 0 || it.b_left < 0 ||
            it.a_right > st->a_len[it.parent] || it.b_right > st->b_len[it.parent] ||
            it.a_right < it.a_left || it.b_right < it.b_left || it.w.width < 3) {
            ctx->err = "bad item ranges"; return -1;
        }
        P.a_left = it.a_left; P.a_right = it.a_right; P.b_left = it.b_left; P.b_right = it.b_right;
        P.lw = it.w.lw; P.up = it.w.up; P.width = it.w.width;
        P.cut_l = P.cut_len = 0;
        if (it.cut_r > it.cut_l) {              // forwardS_ng over a cut range: its arrays are narrower by the cut (src/fwd2s1.cc:234)
            if (flav != 3 || it.a_exgl || it.w.width - (it.cut_r - it.cut_l) < 3 || it.cut_l < it.b_left || it.cut_r > it.b_right ||
                (i > 0 && !cut)) { ctx->err = "bad cut range"; return -1; }
            cut = true;
            P.cut_l = it.cut_l; P.cut_len = it.cut_r - it.cut_l;
            P.width = it.w.width - P.cut_len;
        } else if (cut) { ctx->err = "bad cut range"; return -1; }
        P.buf_size = it.w.width + 2 * SPDP_NELEM;
        P.flags = (it.a_exgl ? 1 : 0) | (it.a_exgr ? 2 : 0) | (it.b_exgl ? 4 : 0) | (it.b_exgr ? 8 : 0);
        P.n_im = (flav == 2 || flav == 5 || flav == 8 || flav == 9) ? it.n_im : 0;
        P.imd_intvl = it.imd_intvl;
        P.a_off = st->a_off[it.parent];
        P.col_off = st->col_off[it.parent];
        P.cip_off = st->cip_off.empty() ? -1 : st->cip_off[it.parent];
        P.bnd_off = bnd_tot; bnd_tot += (int64_t) P.buf_size + SPDP_BND_PAD;
        P.tb_off = tb_tot;
        if (flav >= 6) {                // -A1 engines: hv / fv (/ hb / hc / fc) by diagonal, buf_size ints each, a counter
            P.bnd_off = bnd_tot - ((int64_t) P.buf_size + SPDP_BND_PAD);
            bnd_tot = P.bnd_off + (flav >= 8 ? 6ll : (st->sc.noll == 3 ? 8ll : 5ll)) * P.buf_size + 8;   // udh forms: + the `ml` row of F; Noll = 3: + fv2, fc2 behind the counter
            if (flav >= 8) { P.imd_off = imd_tot; imd_tot += (int64_t) it.n_im * 4 * it.w.width; }
            if (flav == 7) {
                // (+ what the lanes may leave unused of the chunks of numbers they reserve, per stripe when pipelined)
                const int64_t cap = vmf_capacity(it)
                                    + (int64_t) 2 * SPDP_VMF_LANE_CHUNK * SPDP_NELEM * ((it.a_right - it.a_left) / SPDP_NELEM + 2);
                P.imd_off = cap;
                tb_tot += cap;
            }
        } else if (flav == 5) { // scalar UDH: Noll * width + 4 states of 5 ints; 4 * Noll link / bound rows per intermediate
            P.bnd_off = bnd_tot - ((int64_t) P.buf_size + SPDP_BND_PAD);
            bnd_tot = P.bnd_off + 5ll * (st->sc.noll * it.w.width + 4);                  // H and Noll - 1 vertical-gap states
            P.imd_off = imd_tot;
            imd_tot += (int64_t) P.n_im * 4 * st->sc.noll * it.w.width;
        } else if (flav >= 3) { // scalar: work = 4 * width ints + width dir bytes; Vmf records
            P.bnd_off = bnd_tot - ((int64_t) P.buf_size + SPDP_BND_PAD);
            bnd_tot = P.bnd_off + (st->sc.noll == 3 ? 7ll : 5ll) * it.w.width + 8;        // H, F (values, Vmf pointers) and the direction entries by diagonal (+ F2 with Noll = 3)
            // (+ what the waves of a pipelined problem may leave unused of the chunks of numbers they reserve)
            const int64_t cap = (flav == 3) ? vmf_capacity(it) + (int64_t) SPDP_VMF_CHUNK * ((it.a_right - it.a_left) / 64 + 2) : 0;
            P.imd_off = cap;
            tb_tot += cap;
        }
        if (flav == 1) {
            const int ns = (it.a_right - it.a_left + SPDP_NELEM - 1) / SPDP_NELEM;
            for (int s = 0; s < ns; ++s) tb_tot += 256ll * stripe_blocks(P, s, true);
        }
        if (flav < 3) { P.imd_off = imd_tot; imd_tot += (int64_t) P.n_im * 4 * it.w.width; }
        P.cells = spdp_cells_w(it.a_left, it.a_right, it.b_left, it.b_right, it.w);
        total_cells += P.cells;
        max_n_im = std::max(max_n_im, P.n_im);
        max_skl = std::max(max_skl, (it.a_right - it.a_left) + (it.b_right - it.b_left) + 8);
    }
    tb_bytes = tb_tot;
    // fp32 sweeps: every score must stay an exactly representable integer below 2^22 - 2^16 (the penalty table
    // pushes a candidate down by 2^22 to disable it).  Upper bound of a score: matches on every row, plus
    // whatever an intron can gain where the signals outweigh its penalties (never, with real parameters).
    {
        int64_t rows = 0, cols_span = 0;
        for (int i = 0; i < n; ++i) {
            rows = std::max<int64_t>(rows, items[i].a_right - items[i].a_left);
            cols_span = std::max<int64_t>(cols_span, items[i].b_right - items[i].b_left);
        }
        const char* e = getenv("SPDP_FP");
        const int64_t ub = (rows + 1) * st->fp_maxpos + (st->fp_gain > 0 ? (cols_span + 1) * st->fp_gain : 0) + 65536;
        fp_ok = (!e || atoi(e) != 0) && ub < (1ll << 22) - 65536;
    }
    const int bw = (flav == 2) ? 4 : (flav >= 3 ? 1 : 2);
    const int nn = std::max(n, 1);
    skl_cap = std::min(std::max(max_skl, 1), 1024);     // typical lists are short; DevRun::fetch_skl walks the rest again
    if (const char* e = getenv("SPDP_SKL_CAP")) skl_cap = std::max(4, std::min(skl_cap, atoi(e)));   // test hook: tiny slots
    POOLGET(d_probs, POOL_PROBS, sizeof(DevProblem) * nn);
    POOLGET(d_bnd, POOL_BND, sizeof(int32_t) * bw * std::max<int64_t>(bnd_tot, 1));
    POOLGET(d_res, POOL_RES, sizeof(DevResult) * nn);
    if (flav == 3 || flav == 7) {
        POOLGET(d_tb, POOL_TB, sizeof(int3) * std::max<int64_t>(tb_tot, 16));
        POOLGET(d_skl, POOL_SKL, sizeof(int2) * (int64_t) skl_cap * nn);
        POOLGET(d_nskl, POOL_NSKL, sizeof(int) * nn);
    }
    if (flav == 1) {
        POOLGET(d_tb, POOL_TB, std::max<int64_t>(tb_tot, 16));
        POOLGET(d_skl, POOL_SKL, sizeof(int2) * (int64_t) skl_cap * nn);
        POOLGET(d_nskl, POOL_NSKL, sizeof(int) * nn);
    }
    if (flav == 2 || flav == 5 || flav == 8 || flav == 9) {
        POOLGET(d_imd, POOL_IMD, sizeof(int32_t) * std::max<int64_t>(imd_tot, 1));
        POOLGET(d_cpos, POOL_CPOS, sizeof(int32_t) * 10 * (max_n_im + 1) * nn);
        POOLGET(d_ranges, POOL_RANGES, sizeof(int32_t) * 4 * nn);
        POOLGET(d_scores, POOL_SCORES, sizeof(int32_t) * 2 * nn);          // scores, then the left-edge marks of spdp_udh_cpos
    }
    // dispatch order = largest problems first (longest-processing-time rule): one wave owns one
    // problem, so the big ones must not start last.  order[j] = caller index of dispatch slot j.
    order.resize(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    // Problems that can be spread over a block's waves (>= 16 stripes) come first, see below.
    const bool may_multi = flav <= 2 && !st->sc.local;
    auto wide = [&](int x) {
        return may_multi && (h_probs[x].a_right - h_probs[x].a_left + SPDP_NELEM - 1) / SPDP_NELEM >= 16;
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        const bool wx = wide(x), wy = wide(y);
        return wx != wy ? wx : h_probs[x].cells > h_probs[y].cells;
    });
    std::vector<DevProblem> sorted(n);
    for (int j = 0; j < n; ++j) sorted[j] = h_probs[order[j]];
    h_probs.swap(sorted);
    // Big problems (>= 16 stripes = 4 passes) are spread over the 4 waves of a block (pipelined passes).
    // That pays in full launches too: the longest problem bounds the launch, and the passes of one
    // problem running a few blocks apart share their column records in L2.  (Measured on C2, 10000
    // queries: 540 -> 652 GCUPS end to end.)  SPDP_MULTI=0 restores one wave per problem.
    n_multi = 0; wpb = 4;
    if (may_multi) {
        const char* force = getenv("SPDP_MULTI");
        if (!force || atoi(force) != 0)
            for (int j = 0; j < n; ++j) {
                const int stripes = (h_probs[j].a_right - h_probs[j].a_left + SPDP_NELEM - 1) / SPDP_NELEM;
                if (stripes >= 16) n_multi = j + 1; else break;          // sorted: a prefix
            }
        // a launch of a few huge problems only (top levels of the recursion on a long cDNA): one
        // 16-wave block, i.e. a whole CU, per problem
        const char* w16 = getenv("SPDP_WPB16");
        if (w16) { if (atoi(w16) != 0) wpb = 16; }
        else if (n_multi == n && n <= 2 * ctx->n_cu) {
            const int smallest = (h_probs[n - 1].a_right - h_probs[n - 1].a_left + SPDP_NELEM - 1) / SPDP_NELEM;
            if (smallest >= 4 * 32) wpb = 16;                            // >= 32 passes each
            // a handful of mid-sized problems (the stragglers of an EST batch that need the linear-space
            // engine): the launch is latency-bound, more passes in flight shorten it
            else if (!beside && n <= ctx->n_cu / 2 && smallest > 4 * 4) wpb = 16;
        }
        // fewer huge problems than a quarter of the CUs (the top levels of the recursion on one long
        // cDNA): spread each over several CUs -- cross-CU pass pipelines, all blocks resident
        cross_g = 0;
        const char* cg = getenv("SPDP_CROSS");
        // (the traceback sweep too, where spdp_sweep_fp serves it: the tall slabs a long cDNA's recursion leaves, spdp_host.cpp)
        bool fwd_cross = false;
        if (flav == 1 && fp_ok && !(getenv("SPDP_FP_FWD") && atoi(getenv("SPDP_FP_FWD")) == 0)) {
            const int nq = std::max(1, std::min(st->sc.nquant, SPDP_MAX_QUANT));
            fwd_cross = spdp_sweep_fp_serves(st->sc.local ? 1 : 0, st->sc.spj ? 1 : 0, nq, nq > 1 ? st->sc.qm_len[nq - 2] + 1 : 0, st->sc.llmt) != 0;
        }
        if ((flav == 2 || fwd_cross) && wpb == 16 && n_multi == n && n > 0 && (!cg || atoi(cg) != 0)) {
            int max_passes = 0, min_passes = 1 << 30;
            for (int j = 0; j < n; ++j) {
                const int stripes = (h_probs[j].a_right - h_probs[j].a_left + SPDP_NELEM - 1) / SPDP_NELEM;
                max_passes = std::max(max_passes, (stripes + 3) / 4);
                min_passes = std::min(min_passes, (stripes + 3) / 4);
            }
            const int room = ctx->n_cu / n;                              // blocks per problem, one block per CU
            int g = std::min(room, (max_passes + 15) / 16);
            // with CUs to spare a wave gets a SIMD to itself: 4-wave blocks, one pass per wave
            const char* c4 = getenv("SPDP_CROSS_WPB");
            if (c4 ? atoi(c4) == 4 : (max_passes + 3) / 4 <= room) { wpb = 4; g = std::min(room, (max_passes + 3) / 4); }
            if (cg && atoi(cg) > 1) g = std::min(room, atoi(cg));
            if (g >= 2 && min_passes >= 32) cross_g = g;
            else wpb = 16;
        }
    }
    if (cross_g > 0) {
        const size_t words = (size_t) n * (cross_g * wpb + 2);
        POOLGET(d_gprog, POOL_GPROG, sizeof(int) * words);
        HIPCHK(hipMemsetAsync(d_gprog, 0, sizeof(int) * words, strm()));
    }
    // forwardS_ng / scorealoneS_ng: a problem's 64-row tiles as a pipeline of waves (SPDP_A0_PIPE=0: one wave each)
    pipe_on = false;
    if (flav >= 6 && flav <= 8 && n > 0) {
        // -A1 engines: work item = (four problems, 16-row stripe), spdp_exact<., true>
        const char* e = getenv("SPDP_A1_PIPE");
        int mt = 1;
        h_items.clear();
        for (int q = 0; 4 * q < n; ++q) {
            int ns = 1;
            for (int j = 4 * q; j < std::min(n, 4 * q + 4); ++j)
                ns = std::max(ns, (h_probs[j].a_right - h_probs[j].a_left + SPDP_NELEM - 1) / SPDP_NELEM);
            mt = std::max(mt, ns);
            for (int t = 0; t < ns; ++t) { h_items.push_back(q); h_items.push_back(t); }
        }
        if ((!e || atoi(e) != 0) && mt >= 2) {
            pipe_on = true; pipe_tiles = mt;
            pipe_stride = 2 + 7 * mt + max_n_im;
            pipe_words = ((size_t) n * pipe_stride + 2 + 1) & ~(size_t) 1;
            POOLGET(d_gprog, POOL_GPROG, sizeof(int) * (pipe_words + h_items.size()));
        }
    }
    if ((flav == 3 || flav == 4 || flav == 5) && n > 0) {
        const char* e = getenv("SPDP_A0_PIPE");
        int mt = 1;
        h_items.clear();
        for (int j = 0; j < n; ++j) {
            const DevProblem& P = h_probs[j];
            const int r0 = P.a_left + (P.flags & 1);
            const int th = flav == 5 ? std::max(1, std::min(64, P.imd_intvl)) : 64;
            const int nt = std::max(1, (P.a_right - r0 + th) / th);          // as the kernel counts them
            mt = std::max(mt, nt);
            for (int t = 0; t < nt; ++t) { h_items.push_back(j); h_items.push_back(t); }
        }
        if ((!e || atoi(e) != 0) && mt >= 2 && !cut) {
            pipe_on = true; pipe_tiles = mt;
            pipe_stride = flav == 5 ? 2 + 9 * mt + max_n_im : 2 + 5 * mt;
            pipe_words = ((size_t) n * pipe_stride + 2 + 1) & ~(size_t) 1;
            POOLGET(d_gprog, POOL_GPROG, sizeof(int) * (pipe_words + h_items.size()));
        }
    }
    if (n) HIPCHK(hipMemcpyAsync(d_probs, h_probs.data(), sizeof(DevProblem) * n, hipMemcpyHostToDevice, strm()));
    return 0;
}

int DevRun::launch()
{
    (void) hipSetDevice(ctx->device);
    if (n == 0) return 0;
    in_flight = true;
    if (flavour >= 3) {
        ScalarArgs S{};
        if (pipe_on) {
            int* w = (int*) d_gprog;
            HIPCHK(hipMemsetAsync(w, 0, sizeof(int) * pipe_words, strm()));
            HIPCHK(hipMemcpyAsync(w + pipe_words, h_items.data(), sizeof(int) * h_items.size(), hipMemcpyHostToDevice, strm()));
            S.pipe = w; S.pipe_stride = pipe_stride; S.pipe_ticket = n * S.pipe_stride; S.max_tiles = pipe_tiles;
            S.items = (const int2*) (w + pipe_words); S.n_items = (int) (h_items.size() / 2);
        }
        S.sc = (const DevScoring*) store->d_sc; S.probs = (const DevProblem*) d_probs; S.n_probs = n;
        S.a_codes = (const uint8_t*) store->d_a; S.cols = (const int2*) store->d_cols;
        S.aux = (const uint8_t*) store->d_aux; S.intpen = (const int16_t*) store->d_intpen;
        S.ipen_runs = (const int16_t*) store->d_ipen_runs;
        S.cip = (const int*) store->d_cip;
        S.intpen_len = store->sc.intpen_len; S.ipen = store->sc.ipen;
        memcpy(S.t53, store->sc.t53, sizeof S.t53);
        S.work = (int*) d_bnd; S.vmf = (int3*) d_tb; S.res = (DevResult*) d_res;
        S.skl = (int2*) d_skl; S.n_skl = (int*) d_nskl; S.skl_cap = skl_cap;
        S.imd = (int*) d_imd; S.cpos = (int*) d_cpos; S.ranges = (int*) d_ranges; S.scores = (int*) d_scores;
        S.cpos_stride = 10 * (max_n_im + 1);
        HIPCHK(hipEventRecord(evb(), strm()));
        S.minl = store->sc.minl ? store->sc.minl : store->sc.llmt;
        S.noll = store->sc.noll; S.lgop = store->sc.lgop; S.lgep = store->sc.lgep; S.codonk1 = store->sc.codonk1;
        if (flavour == 9) HIPCHK(spdp_launch_local_udh(&S, strm()));
        else if (flavour >= 6) HIPCHK(spdp_launch_exact(flavour - 6, &S, strm()));
        else if (flavour == 5) HIPCHK(spdp_launch_rowwave_udh(&S, strm()));
        else HIPCHK(spdp_launch_rowwave(flavour == 3 ? (cut ? 2 : 1) : 0, &S, strm()));
        HIPCHK(hipEventRecord(eve(), strm()));
        if (flavour >= 8) {                 // hirschbergS1's / the local hirschbergS1_wip's link walk
            CposArgs C{};
            C.probs = S.probs; C.n_probs = n; C.imd = (const int*) d_imd; C.res = (const DevResult*) d_res;
            C.cpos = (int*) d_cpos; C.ranges = (int*) d_ranges; C.scores = (int*) d_scores;
            C.cpos_stride = 10 * (max_n_im + 1); C.strict = flavour == 8; C.local = store->sc.local ? 1 : 0;
            C.edge = (int*) d_scores + n;
            C.pipe = (flavour == 8 && pipe_on) ? (const int*) d_gprog : nullptr;
            C.pipe_stride = pipe_stride; C.rlf_off = 2 + 7 * pipe_tiles;
            HIPCHK(spdp_launch_cpos(&C, strm()));
        }
        return 0;
    }
    SweepArgs A;
    A.sc = (const DevScoring*) store->d_sc; A.probs = (const DevProblem*) d_probs; A.n_probs = n;
    A.a_codes = (const uint8_t*) store->d_a; A.cols = (const int2*) store->d_cols; A.bnd = (int*) d_bnd;
    A.tb = (uint8_t*) d_tb; A.imd = (int*) d_imd; A.res = (DevResult*) d_res; A.n_multi = n_multi;
    A.cross_g = cross_g; A.gprog = (int*) d_gprog;
    int grid = cross_g > 0 ? n * cross_g : n_multi + (n - n_multi + wpb - 1) / wpb;
    if (cross_g > 0 && getenv("SPDP_CROSS_TEST_SHORT")) --grid;    // test hook: one block never arrives -> the fallback runs
    HIPCHK(hipEventRecord(evb(), strm()));
    const int nq = std::max(1, std::min(store->sc.nquant, SPDP_MAX_QUANT));
    const int pen_cap = nq > 1 ? store->sc.qm_len[nq - 2] + 1 : 0;
    // the fp32-issue form of the sweep (spdp_sweep_fp.hip) where it applies: non-local score-only / linear-space runs
    // whose scores stay inside the exact fp32 integer range; SPDP_FP=0 keeps everything on spdp_kernels.hip
    bool done = false;
    const bool fp_fwd = !(getenv("SPDP_FP_FWD") && atoi(getenv("SPDP_FP_FWD")) == 0);       // SPDP_FP_FWD=0: the traceback sweep stays on spdp_kernels.hip
    if (fp_ok && (flavour != 1 || fp_fwd) && !store->sc.local) {
        const hipError_t e = spdp_launch_sweep_fp(flavour, 0, store->sc.spj ? 1 : 0, nq, pen_cap, store->sc.llmt, &A, grid, wpb, strm());
        if (e == hipSuccess) done = true;
        else if (e != hipErrorNotSupported) HIPCHK(e);
    }
    if (!done) HIPCHK(spdp_launch_sweep(flavour, store->sc.local ? 1 : 0, nq, pen_cap, &A, grid, wpb, strm()));
    HIPCHK(hipEventRecord(eve(), strm()));
    if (flavour == 1) {
        WalkArgs W;
        W.probs = A.probs; W.n_probs = n; W.tb = (const uint8_t*) d_tb; W.res = (const DevResult*) d_res;
        W.skl = (int2*) d_skl; W.n_skl = (int*) d_nskl; W.skl_cap = skl_cap;
        { const char* sw = getenv("SPDP_WALK_SEQ"); W.seq = (sw && atoi(sw) != 0) ? 1 : 0; }
        HIPCHK(spdp_launch_walk(&W, strm()));
    }
    if (flavour == 2) {
        CposArgs C{};
        C.probs = A.probs; C.n_probs = n; C.imd = (const int*) d_imd; C.res = (const DevResult*) d_res;
        C.cpos = (int*) d_cpos; C.ranges = (int*) d_ranges; C.scores = (int*) d_scores;
        C.cpos_stride = 10 * (max_n_im + 1); C.strict = 0; C.local = 0;
        C.edge = (int*) d_scores + n;
        HIPCHK(spdp_launch_cpos(&C, strm()));
    }
    return 0;
}

extern "C" void spdp_rerun_stats(SpdpContext* ctx, int64_t* out, int reset)
{
    if (!ctx || !out) return;
    out[0] = ctx->rerun_stats[0]; out[1] = ctx->rerun_stats[1];
    if (reset) ctx->rerun_stats[0] = ctx->rerun_stats[1] = 0;
}

int DevRun::sync()
{
    HIPCHK(hipStreamSynchronize(strm()));
    in_flight = false;
    if (cross_g > 0) {
        // cross-CU pipelines need every block resident; on a shared GPU the start-up barrier can time out, the
        // blocks then leave a mark (second barrier word of their problem) and the launch is repeated without them
        const size_t per = (size_t) cross_g * wpb + 2;
        std::vector<int> words(per * n);
        HIPCHK(spdp_copy_sync(words.data(), d_gprog, sizeof(int) * words.size(), hipMemcpyDeviceToHost, strm()));
        bool gave_up = false;
        for (int j = 0; j < n; ++j) gave_up = gave_up || words[per * j + per - 1] != 0;
        if (gave_up) {
            ++ctx->rerun_stats[0];
            if (getenv("SPDP_TRACE_RUNS")) fprintf(stderr, "[spdp run] a cross-CU group was not resident%s: launch repeated without the groups\n", side ? " (side stream, beside another launch)" : "");
            cross_g = 0; wpb = 16;
            if (launch()) return -1;
            HIPCHK(hipStreamSynchronize(strm()));
        }
    }
    if (pipe_on) {
        // a wave that waited in vain for the tile above it leaves a mark: the launch is repeated with one wave per problem
        int mark[2] = {0, 0};
        HIPCHK(spdp_copy_sync(mark, (int*) d_gprog + (size_t) n * pipe_stride, sizeof mark, hipMemcpyDeviceToHost, strm()));
        if (mark[1] != 0 || getenv("SPDP_A0_PIPE_TEST_STALL")) {
            ++ctx->rerun_stats[1];
            pipe_on = false;
            if (launch()) return -1;
            HIPCHK(hipStreamSynchronize(strm()));
        }
    }
    kernel_ms = 0.f;
    if (n) HIPCHK(hipEventElapsedTime(&kernel_ms, evb(), eve()));
    if (n && getenv("SPDP_TRACE_RUNS")) {
        int mr = 0, mc = 0; int64_t mcell = 0;
        for (int j = 0; j < n; ++j) {
            mr = std::max(mr, h_probs[j].a_right - h_probs[j].a_left); mc = std::max(mc, h_probs[j].width);
            mcell = std::max<int64_t>(mcell, h_probs[j].cells);
        }
        fprintf(stderr, "[spdp run] flavour %d n %d cells %.3g (largest %.3g, rows <= %d, width <= %d) pipe %d items %zu  %.2f ms  %.1f GCUPS\n",
                flavour, n, (double) total_cells, (double) mcell, mr, mc, (int) pipe_on, h_items.size() / 2, kernel_ms,
                total_cells / (kernel_ms * 1e6));
    }
    return 0;
}

int DevRun::fetch_results(std::vector<DevResult>& out)
{
    out.resize(n);
    std::vector<DevResult> tmp(n);
    if (n) HIPCHK(spdp_copy_sync(tmp.data(), d_res, sizeof(DevResult) * n, hipMemcpyDeviceToHost, strm()));
    for (int j = 0; j < n; ++j) out[order[j]] = tmp[j];
    return 0;
}

int DevRun::fetch_skl(std::vector<int>& n_skl, std::vector<int64_t>& off, std::vector<SpdpSkl>& skl)
{
    const int flav = flavour;
    n_skl.assign(n, 0); off.assign(n + 1, 0);
    if (!n) { skl.clear(); return 0; }
    std::vector<int> cnt(n);                            // dispatch order
    HIPCHK(spdp_copy_sync(cnt.data(), d_nskl, sizeof(int) * n, hipMemcpyDeviceToHost, strm()));
    // Lists that did not fit their slot (skl_cap is sized for typical lists; an indel-rich slab can need more): the
    // walk is repeated for those few with slots of the longest list their problem can produce.  Only the `_wip`
    // walk (flavour 1) can be repeated on its own; the scalar / -A1 engines walk inside their sweep kernel and
    // report the overflow (-1) to the caller, who gives up that one query, not the batch.
    std::vector<int> over;
    for (int j = 0; j < n; ++j) if (cnt[j] == -1) over.push_back(j);
    std::vector<std::vector<SpdpSkl>> redo(over.size());
    if (!over.empty() && flav == 1) {
        const int m = (int) over.size();
        std::vector<DevResult> all_res(n), sub_res(m);
        std::vector<DevProblem> sub_probs(m);
        HIPCHK(spdp_copy_sync(all_res.data(), d_res, sizeof(DevResult) * n, hipMemcpyDeviceToHost, strm()));
        int cap2 = 1;
        for (int k = 0; k < m; ++k) {
            const DevProblem& P = h_probs[over[k]];
            sub_probs[k] = P; sub_res[k] = all_res[over[k]];
            cap2 = std::max(cap2, (P.a_right - P.a_left) + (P.b_right - P.b_left) + 8);
        }
        void *dp = nullptr, *dr = nullptr, *ds = nullptr, *dn = nullptr;
        HIPCHK(hipMalloc(&dp, sizeof(DevProblem) * m));
        HIPCHK(hipMalloc(&dr, sizeof(DevResult) * m));
        HIPCHK(hipMalloc(&ds, sizeof(int2) * (size_t) cap2 * m));
        HIPCHK(hipMalloc(&dn, sizeof(int) * m));
        HIPCHK(spdp_copy_sync(dp, sub_probs.data(), sizeof(DevProblem) * m, hipMemcpyHostToDevice, strm()));
        HIPCHK(spdp_copy_sync(dr, sub_res.data(), sizeof(DevResult) * m, hipMemcpyHostToDevice, strm()));
        WalkArgs W;
        W.probs = (const DevProblem*) dp; W.n_probs = m; W.tb = (const uint8_t*) d_tb; W.res = (const DevResult*) dr;
        W.skl = (int2*) ds; W.n_skl = (int*) dn; W.skl_cap = cap2; W.seq = 0;
        HIPCHK(spdp_launch_walk(&W, strm()));
        HIPCHK(hipStreamSynchronize(strm()));
        std::vector<int> c2(m);
        HIPCHK(spdp_copy_sync(c2.data(), dn, sizeof(int) * m, hipMemcpyDeviceToHost, strm()));
        for (int k = 0; k < m; ++k) {
            cnt[over[k]] = c2[k];
            if (c2[k] > 0) {
                redo[k].resize(c2[k]);
                HIPCHK(spdp_copy_sync(redo[k].data(), (const int2*) ds + (size_t) k * cap2, sizeof(SpdpSkl) * c2[k], hipMemcpyDeviceToHost, strm()));
            }
        }
        (void) hipFree(dp); (void) hipFree(dr); (void) hipFree(ds); (void) hipFree(dn);
    }
    std::vector<int64_t> doff(n + 1, 0);
    for (int j = 0; j < n; ++j) {
        if (cnt[j] > skl_cap && redo.empty()) { ctx->err = "traceback record list exceeds the per-problem slot"; return -1; }
        // (cnt = -3: the call outgrew its Vmf record budget; the ladder runs it again with a larger one)
        doff[j + 1] = doff[j] + std::max(cnt[j], 0);
        n_skl[order[j]] = cnt[j];
    }
    std::vector<SpdpSkl> packed(doff[n]);
    if (doff[n]) {
        void *d_off, *d_pack;
        POOLGET(d_off, POOL_SKLOFF, sizeof(int64_t) * (n + 1));
        POOLGET(d_pack, POOL_SKLPACK, sizeof(int2) * doff[n]);
        HIPCHK(hipMemcpyAsync(d_off, doff.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, strm()));
        HIPCHK(spdp_launch_pack((const int2*) d_skl, skl_cap, (const int*) d_nskl, (const int64_t*) d_off,
                                (int2*) d_pack, n, strm()));
        HIPCHK(hipMemcpyAsync(packed.data(), d_pack, sizeof(SpdpSkl) * doff[n], hipMemcpyDeviceToHost, strm()));
        HIPCHK(hipStreamSynchronize(strm()));
    }
    // back to the caller's order
    for (int i = 0; i < n; ++i) off[i + 1] = off[i] + std::max(n_skl[i], 0);
    skl.resize(off[n]);
    for (int j = 0; j < n; ++j)
        if (cnt[j] > 0) memcpy(skl.data() + off[order[j]], packed.data() + doff[j], sizeof(SpdpSkl) * cnt[j]);
    for (size_t k = 0; k < over.size(); ++k)            // the lists of the second walk (the pack kernel skipped them)
        if (!redo[k].empty()) memcpy(skl.data() + off[order[over[k]]], redo[k].data(), sizeof(SpdpSkl) * redo[k].size());
    return 0;
}

int DevRun::fetch_udh(std::vector<int32_t>& scores, std::vector<int32_t>& cpos, std::vector<int32_t>& ranges,
                      std::vector<int32_t>* edge)
{
    const size_t st = (size_t) 10 * (max_n_im + 1);
    scores.resize(n); ranges.resize((size_t) 4 * n); cpos.resize(st * n);
    if (edge) edge->assign(n, 0);
    if (n) {
        std::vector<int32_t> ts(n), tr((size_t) 4 * n), tc(st * n);
        HIPCHK(spdp_copy_sync(ts.data(), d_scores, sizeof(int32_t) * n, hipMemcpyDeviceToHost, strm()));
        if (edge && (flavour == 2 || flavour >= 8)) {
            std::vector<int32_t> te(n);
            HIPCHK(spdp_copy_sync(te.data(), (int32_t*) d_scores + n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, strm()));
            for (int j = 0; j < n; ++j) (*edge)[order[j]] = te[j];
        }
        HIPCHK(spdp_copy_sync(tr.data(), d_ranges, sizeof(int32_t) * 4 * n, hipMemcpyDeviceToHost, strm()));
        HIPCHK(spdp_copy_sync(tc.data(), d_cpos, sizeof(int32_t) * tc.size(), hipMemcpyDeviceToHost, strm()));
        for (int j = 0; j < n; ++j) {
            const int i = order[j];
            scores[i] = ts[j];
            memcpy(&ranges[(size_t) 4 * i], &tr[(size_t) 4 * j], sizeof(int32_t) * 4);
            memcpy(&cpos[st * i], &tc[st * j], sizeof(int32_t) * st);
        }
    }
    return 0;
}

// ---- engine-level entry points (SimdAln2s1 methods) -----------------------------------------
static std::vector<RunItem> items_of(const SpdpScoring* sc, const SpdpProblem* probs, int n)
{
    std::vector<RunItem> v(n);
    for (int i = 0; i < n; ++i) v[i] = spdp_item_of(probs[i], i, sc->sh);
    return v;
}

int spdp_wip_scoreonly(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                       int n_probs, int32_t* scores)
{
    if (!ctx) return -1;
    if (n_probs <= 0) return 0;
    DevStore st; DevRun run;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    if (run.build(&st, items_of(sc, probs, n_probs), 0) || run.launch() || run.sync()) return -1;
    std::vector<DevResult> r;
    if (run.fetch_results(r)) return -1;
    for (int i = 0; i < n_probs; ++i) scores[i] = r[i].score;
    return 0;
}

// HomScoreS_ng on an uploaded store (src/fwd2s1.cc:2696-2716): scoreonlyS1_wip on the stripe() band for
// simd > 1; scorealoneS_ng when simd == 0 or the query range has fewer than 4 rows.  Returns 1 when a
// problem needs the scalar engine and its inputs are missing (score NEVSEL).
static int homscore_on_store(SpdpContext* ctx, const DevStore& st, const SpdpProblem* probs, int n_probs, int32_t* scores)
{
    const SpdpScoring* sc = &st.sc;
    std::vector<RunItem> vec, sca, exa;
    std::vector<int> vi, si, ei;
    int rc = 0;
    for (int i = 0; i < n_probs; ++i) {
        scores[i] = SPDP_NEVSEL;
        RunItem it = spdp_item_of(probs[i], i, sc->sh);
        const int m = it.a_right - it.a_left;
        if (it.w.width < 3) { rc = 1; continue; }
        if (sc->scalar_engines == 1 || m < 4) {
            if (!st.has_exact) { rc = 1; continue; }
            sca.push_back(it); si.push_back(i);
        } else if (sc->scalar_engines == 2) {           // -A1: scoreonlyS1
            if (!st.has_exact) { rc = 1; continue; }
            exa.push_back(it); ei.push_back(i);
        } else { vec.push_back(it); vi.push_back(i); }
    }
    std::vector<DevResult> r;
    if (!vec.empty()) {
        DevRun run;
        if (run.build(&st, vec, 0) || run.launch() || run.sync() || run.fetch_results(r)) return -1;
        for (size_t k = 0; k < vec.size(); ++k) scores[vi[k]] = r[k].score;
    }
    if (!sca.empty()) {
        DevRun run;
        if (run.build(&st, sca, 4) || run.launch() || run.sync() || run.fetch_results(r)) return -1;
        for (size_t k = 0; k < sca.size(); ++k) scores[si[k]] = r[k].score;
    }
    if (!exa.empty()) {
        DevRun run;
        if (run.build(&st, exa, 6) || run.launch() || run.sync() || run.fetch_results(r)) return -1;
        for (size_t k = 0; k < exa.size(); ++k) scores[ei[k]] = r[k].score;
    }
    return rc;
}

int spdp_homscore_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                    int n_probs, int32_t* scores)
{
    if (!ctx || !sc || !probs || !scores) return -1;
    if (n_probs <= 0) return 0;
    DevStore st;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    return homscore_on_store(ctx, st, probs, n_probs, scores);
}

int spdp_wip_forward(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                     int n_probs, SpdpAlignment* out);

int spdp_wip_udh(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                 int n_im, int32_t* scores, int32_t* cpos, int32_t* ranges)
{
    if (!ctx) return -1;
    if (n_probs <= 0) return 0;
    DevStore st; DevRun run;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<RunItem> items = items_of(sc, probs, n_probs);
    for (auto& it : items) it.n_im = n_im;
    if (run.build(&st, items, 2) || run.launch() || run.sync()) return -1;
    std::vector<int32_t> s, c, r;
    if (run.fetch_udh(s, c, r)) return -1;
    memcpy(scores, s.data(), sizeof(int32_t) * n_probs);
    memcpy(ranges, r.data(), sizeof(int32_t) * 4 * n_probs);
    memcpy(cpos, c.data(), sizeof(int32_t) * c.size());
    return 0;
}

int spdp_scalar_udh(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                    int n_im, int imd_intvl, int32_t* scores, int32_t* cpos, int32_t* ranges, int32_t* flags)
{
    if (!ctx || !sc || !probs || !scores || !cpos || !ranges || !flags || n_im < 1 || imd_intvl < 1) return -1;
    if (n_probs <= 0) return 0;
    DevStore st; DevRun run;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<RunItem> items = items_of(sc, probs, n_probs);
    for (auto& it : items) {
        it.n_im = n_im; it.imd_intvl = imd_intvl;
        if (it.a_left + (int64_t) n_im * imd_intvl > it.a_right + imd_intvl) { ctx->err = "intermediate rows beyond the query range"; return -1; }
    }
    // test hook: the same call on hirschbergS1 (the -A1 linear-space engine, spdp_exact<2> + spdp_udh_cpos), which the ABI
    // otherwise reaches through alignS_ng only; imd_intvl is then the engine's own (a_right - a_left + n_im) / (n_im + 1)
    const int flav = getenv("SPDP_UDH_ENGINE_A1") ? 8 : 5;
    if (run.build(&st, items, flav) || run.launch() || run.sync()) return -1;
    std::vector<int32_t> s, c, r;
    std::vector<DevResult> res;
    if (run.fetch_udh(s, c, r) || run.fetch_results(res)) return -1;
    memcpy(scores, s.data(), sizeof(int32_t) * n_probs);
    memcpy(ranges, r.data(), sizeof(int32_t) * 4 * n_probs);
    memcpy(cpos, c.data(), sizeof(int32_t) * c.size());
    for (int i = 0; i < n_probs; ++i) flags[i] = res[i].pad[0];
    return 0;
}

static int forward_like(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                        SpdpAlignment* out, int flav)
{
    if (!ctx) return -1;
    for (int i = 0; i < n_probs; ++i) { out[i].score = SPDP_NEVSEL; out[i].n_skl = 0; out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; }
    if (n_probs <= 0) return 0;
    DevStore st; DevRun run;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    std::vector<RunItem> items = items_of(sc, probs, n_probs);
    for (RunItem& it : items) it.vmf_scale = 1 << 20;           // engine-level call: the full record budget at once
    if (run.build(&st, items, flav) || run.launch() || run.sync()) return -1;
    std::vector<DevResult> r;
    std::vector<int> nskl;
    std::vector<int64_t> off;
    std::vector<SpdpSkl> skl;
    if (run.fetch_results(r) || run.fetch_skl(nskl, off, skl)) return -1;
    for (int i = 0; i < n_probs; ++i) {
        out[i].score = r[i].score;
        if (nskl[i] < 0) { ctx->err = "traceback failed"; return -1; }
        out[i].n_skl = nskl[i];
        out[i].skl = (SpdpSkl*) malloc(sizeof(SpdpSkl) * std::max(1, nskl[i]));
        memcpy(out[i].skl, skl.data() + off[i], sizeof(SpdpSkl) * nskl[i]);
    }
    return 0;
}

int spdp_scalar_forward(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                        int n_probs, SpdpAlignment* out)
{   // Aln2s1::forwardS_ng through trcbkalignS_ng (scalar exact engine)
    return forward_like(ctx, sc, probs, n_probs, out, 3);
}

int spdp_scalar_scorealone(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                           int n_probs, int32_t* scores)
{   // Aln2s1::scorealoneS_ng = HomScoreS_ng under -A0
    if (!ctx) return -1;
    if (n_probs <= 0) return 0;
    DevStore st; DevRun run;
    if (st.upload(ctx, sc, probs, n_probs)) return -1;
    if (run.build(&st, items_of(sc, probs, n_probs), 4) || run.launch() || run.sync()) return -1;
    std::vector<DevResult> r;
    if (run.fetch_results(r)) return -1;
    for (int i = 0; i < n_probs; ++i) scores[i] = r[i].score;
    return 0;
}

int spdp_wip_forward(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                     int n_probs, SpdpAlignment* out)
{
    return forward_like(ctx, sc, probs, n_probs, out, 1);
}

void spdp_free_alignments(SpdpAlignment* out, int n)
{
    for (int i = 0; i < n; ++i) { free(out[i].skl); out[i].skl = nullptr; out[i].flags = 0; out[i].reserved = 0; out[i].n_skl = 0; }
}
