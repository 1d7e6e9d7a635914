// spdp_signals_api.cpp -- host side of the splice-signal precompute (spdp_signals.hip): the C-ABI entry
// spdp_splice_signals and the helper DevStore::upload uses when a batch arrives as plain codes.
#include "spdp_internal.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

struct DevBuf {                     // scoped device allocation
    void* p = nullptr;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t get(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T> T* as() const { return (T*) p; }
};

const char* check_model(const SpdpSignalModel* m)
{
    if (!m || !m->mtx5 || !m->mtx3) return "signal model: matrices missing";
    if (m->rows != 84) return "signal model: only second-order Markov matrices over 4 letters (84 rows) are implemented";
    const int c[2] = {m->cols5, m->cols3}, o[2] = {m->off5, m->off3};
    for (int i = 0; i < 2; ++i)
        if (c[i] < 1 || c[i] > 48 || o[i] < 0 || o[i] > 60 || c[i] - o[i] + 2 > 60) return "signal model: matrix shape out of range";
    return nullptr;
}

}  // namespace

// model -> device, then one launch per <= 65535 windows; outputs as SignalArgs says (device pointers)
int spdp_signals_run(SpdpContext* ctx, const SpdpSignalModel* m, const std::vector<SigJob>& jobs, SignalArgs args,
                     int* max_s5, int* max_s3)
{
    if (const char* e = check_model(m)) { ctx->err = e; return -1; }
    (void) hipSetDevice(ctx->device);
    SigModelDev hm;
    memset(&hm, 0, sizeof hm);
    hm.rows = m->rows; hm.cols5 = m->cols5; hm.off5 = m->off5; hm.cols3 = m->cols3; hm.off3 = m->off3;
    hm.any = m->any; hm.both_ori = m->both_ori ? 1 : 0;
    hm.fs = m->fs; hm.tonic5 = m->tonic5; hm.min5 = m->min5; hm.tonic3 = m->tonic3; hm.min3 = m->min3;
    memcpy(hm.tab5, m->tab5, sizeof hm.tab5); memcpy(hm.tab3, m->tab3, sizeof hm.tab3);
    const size_t n5 = (size_t) m->rows * m->cols5, n3 = (size_t) m->rows * m->cols3;
    DevBuf d_model, d_mtx, d_jobs, d_max;
    HIPCHK(d_model.get(sizeof hm));
    HIPCHK(d_mtx.get((n5 + n3) * sizeof(float)));
    HIPCHK(d_jobs.get(jobs.size() * sizeof(SigJob)));
    HIPCHK(d_max.get(2 * sizeof(int)));
    const int init[2] = {INT32_MIN, INT32_MIN};
    HIPCHK(hipMemcpyAsync(d_model.p, &hm, sizeof hm, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_mtx.p, m->mtx5, n5 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_mtx.as<float>() + n5, m->mtx3, n3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_jobs.p, jobs.data(), jobs.size() * sizeof(SigJob), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_max.p, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    args.model = d_model.as<SigModelDev>();
    args.mtx5 = d_mtx.as<float>(); args.mtx3 = d_mtx.as<float>() + n5;
    args.maxes = d_max.as<int>();
    for (size_t j0 = 0; j0 < jobs.size(); j0 += 65535) {
        const int nj = (int) std::min<size_t>(65535, jobs.size() - j0);
        int max_len = 0;
        for (int j = 0; j < nj; ++j) max_len = std::max(max_len, jobs[j0 + j].b_len);
        args.jobs = d_jobs.as<SigJob>() + j0;
        HIPCHK(spdp_launch_signals(&args, nj, max_len, (int) (n5 + n3), ctx->stream));
    }
    int got[2];
    HIPCHK(hipMemcpyAsync(got, d_max.p, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (max_s5) *max_s5 = got[0];
    if (max_s3) *max_s3 = got[1];
    return 0;
}

extern "C" int spdp_splice_signals(SpdpContext* ctx, const SpdpSignalModel* model, const uint8_t* b, int32_t b_len,
                                   int32_t left, int32_t right, int16_t* sig5, int16_t* sig3,
                                   uint8_t* cano5, uint8_t* cano3, uint8_t* dinc)
{
    if (!ctx) return -1;
    if (!b || b_len < 0 || left < 0 || right > b_len || right < left) { ctx->err = "spdp_splice_signals: bad window"; return -1; }
    (void) hipSetDevice(ctx->device);
    const size_t n1 = (size_t) b_len + 1;
    DevBuf d_b, d_s5, d_s3, d_c5, d_c3, d_dc;
    HIPCHK(d_b.get(b_len)); HIPCHK(d_s5.get(2 * n1)); HIPCHK(d_s3.get(2 * n1));
    HIPCHK(d_c5.get(n1)); HIPCHK(d_c3.get(n1)); HIPCHK(d_dc.get(n1));
    if (b_len) HIPCHK(hipMemcpyAsync(d_b.p, b, b_len, hipMemcpyHostToDevice, ctx->stream));
    SigJob J;
    memset(&J, 0, sizeof J);
    J.b_len = b_len; J.left = left; J.right = right;
    SignalArgs A;
    memset(&A, 0, sizeof A);
    A.codes = d_b.as<uint8_t>();
    A.sig5 = d_s5.as<int16_t>(); A.sig3 = d_s3.as<int16_t>();
    A.cano5 = d_c5.as<uint8_t>(); A.cano3 = d_c3.as<uint8_t>(); A.dinc = d_dc.as<uint8_t>();
    if (spdp_signals_run(ctx, model, std::vector<SigJob>(1, J), A, nullptr, nullptr)) return -1;
    if (sig5) HIPCHK(hipMemcpy(sig5, d_s5.p, 2 * n1, hipMemcpyDeviceToHost));
    if (sig3) HIPCHK(hipMemcpy(sig3, d_s3.p, 2 * n1, hipMemcpyDeviceToHost));
    if (cano5) HIPCHK(hipMemcpy(cano5, d_c5.p, n1, hipMemcpyDeviceToHost));
    if (cano3) HIPCHK(hipMemcpy(cano3, d_c3.p, n1, hipMemcpyDeviceToHost));
    if (dinc) HIPCHK(hipMemcpy(dinc, d_dc.p, n1, hipMemcpyDeviceToHost));
    return 0;
}

// ---- protein side (spdp_signals_h.hip) ---------------------------------------------------------------------------
#include "spdp_h_internal.h"

static const char* check_model_h(const SpdpSignalModelH* m)
{
    if (!m || !m->pm5.mtx || !m->pm3.mtx) return "signal model: splice-site matrices missing";
    const SpdpPatMat* pm[5] = {&m->pm5, &m->pm3, &m->pmI, &m->pmT, &m->pmB};
    for (int i = 0; i < 5; ++i) {
        const SpdpPatMat& q = *pm[i];
        if (i >= 2 && (!q.rows || !q.mtx)) continue;
        if ((q.order == 2 && q.rows != 84) || (q.order == 1 && q.rows != 20) || (q.order == 0 && q.rows != 4) || q.order < 0 || q.order > 2)
            return "signal model: matrix rows do not match its Markov order";
        if (q.cols < 1 || q.cols > 48 || q.offset < 0 || q.offset > 60 || q.cols - q.offset + 2 > 60) return "signal model: matrix shape out of range";
    }
    if (m->pot_ndata && (!m->pot || m->pot_ndata != 4096)) return "signal model: coding potential must be the 5th-order table (4096 x 3)";
    return nullptr;
}

// model -> device; jobs (host) -> device; the three kernels; `args` carries the device pointers of codes and outputs
int spdh_signals_run(SpdpContext* ctx, const SpdpSignalModelH* m, const std::vector<SigJobH>& jobs, SignalArgsH args, int pack)
{
    if (const char* e = check_model_h(m)) { ctx->err = e; return -1; }
    (void) hipSetDevice(ctx->device);
    SigModelHDev hm;
    memset(&hm, 0, sizeof hm);
    const SpdpPatMat* pm[5] = {&m->pm5, &m->pm3, &m->pmI, &m->pmT, &m->pmB};
    SigPatMatDev* dm[5] = {&hm.pm5, &hm.pm3, &hm.pmI, &hm.pmT, &hm.pmB};
    size_t tot = 0;
    for (int i = 0; i < 5; ++i) {
        const bool on = pm[i]->rows && pm[i]->mtx;
        if (on) { dm[i]->rows = pm[i]->rows; dm[i]->cols = pm[i]->cols; dm[i]->offset = pm[i]->offset; dm[i]->order = pm[i]->order;
                  dm[i]->tonic = pm[i]->tonic; dm[i]->min_elem = pm[i]->min_elem; tot += (size_t) pm[i]->rows * pm[i]->cols; }
    }
    hm.pot_ndata = m->pot ? m->pot_ndata : 0; hm.any = m->any; hm.dvsp = m->dvsp ? 1 : 0; hm.trm = m->trm; hm.trm2 = m->trm2;
    hm.fE = m->fE; hm.fT = m->fT; hm.fO = m->fO; hm.fS = m->fS; hm.fs = m->fs; hm.tonic5 = m->tonic5; hm.tonic3 = m->tonic3;
    hm.fB = m->fB; hm.thB = (float) (int16_t) (int) m->tonicB; hm.maxb3d = m->maxb3d;      // (STYPE thB, src/codepot.cc:536)
    memcpy(hm.tab5, m->tab5, sizeof hm.tab5); memcpy(hm.tab3, m->tab3, sizeof hm.tab3);
    std::vector<float> hmtx;
    hmtx.reserve(tot);
    for (int i = 0; i < 5; ++i) if (dm[i]->rows) hmtx.insert(hmtx.end(), pm[i]->mtx, pm[i]->mtx + (size_t) dm[i]->rows * dm[i]->cols);
    DevBuf d_model, d_mtx, d_pot, d_jobs, d_sb;
    args.sb = nullptr;
    if (hm.pmB.rows) {                                       // the branch-point scores of all positions, between the two kernels
        int64_t n_pos = 0;
        for (const SigJobH& j : jobs) n_pos = std::max<int64_t>(n_pos, j.out_off + j.b_len + 3);
        HIPCHK(d_sb.get((size_t) n_pos * sizeof(int32_t)));
        args.sb = d_sb.as<int32_t>();
    }
    HIPCHK(d_model.get(sizeof hm));
    HIPCHK(d_mtx.get(hmtx.size() * sizeof(float)));
    HIPCHK(d_pot.get(hm.pot_ndata ? 3 * (size_t) hm.pot_ndata * sizeof(float) : 16));
    HIPCHK(d_jobs.get(jobs.size() * sizeof(SigJobH)));
    HIPCHK(hipMemcpyAsync(d_model.p, &hm, sizeof hm, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_mtx.p, hmtx.data(), hmtx.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (hm.pot_ndata) HIPCHK(hipMemcpyAsync(d_pot.p, m->pot, 3 * (size_t) hm.pot_ndata * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_jobs.p, jobs.data(), jobs.size() * sizeof(SigJobH), hipMemcpyHostToDevice, ctx->stream));
    args.model = d_model.as<SigModelHDev>(); args.mtx = d_mtx.as<float>(); args.pot = d_pot.as<float>();
    for (size_t j0 = 0; j0 < jobs.size(); j0 += 65535) {
        const int nj = (int) std::min<size_t>(65535, jobs.size() - j0);
        int max_len = 0;
        for (int j = 0; j < nj; ++j) max_len = std::max(max_len, jobs[j0 + j].b_len);
        args.jobs = d_jobs.as<SigJobH>() + j0;
        HIPCHK(spdh_launch_signals(&args, nj, max_len, (int) hmtx.size(), pack, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int spdp_splice_signals_h(SpdpContext* ctx, const SpdpSignalModelH* model, const uint8_t* b, int32_t b_len,
                                     int32_t left, int32_t right, int16_t* sig5, int16_t* sig3, int16_t* sigS, int16_t* sigT,
                                     int16_t* sigE, int8_t* phs5, int8_t* phs3, uint8_t* dinc)
{
    if (!ctx) return -1;
    if (!b || b_len < 0 || left < 0 || right > b_len || right < left) { ctx->err = "spdp_splice_signals_h: bad window"; return -1; }
    (void) hipSetDevice(ctx->device);
    const size_t N = (size_t) b_len + 3;
    DevBuf d_b, d_s[5], d_p[2], d_c, d_d;
    HIPCHK(d_b.get(b_len + 1));
    for (DevBuf& x : d_s) HIPCHK(x.get(2 * N));
    for (DevBuf& x : d_p) HIPCHK(x.get(N));
    HIPCHK(d_c.get(N)); HIPCHK(d_d.get(N));
    HIPCHK(hipMemcpyAsync(d_b.p, b, b_len + 1, hipMemcpyHostToDevice, ctx->stream));
    SigJobH J;
    memset(&J, 0, sizeof J);
    J.b_len = b_len; J.left = left; J.right = right;
    SignalArgsH A;
    memset(&A, 0, sizeof A);
    A.codes = d_b.as<uint8_t>();
    A.sig5 = d_s[0].as<int16_t>(); A.sig3 = d_s[1].as<int16_t>(); A.sigS = d_s[2].as<int16_t>(); A.sigT = d_s[3].as<int16_t>();
    A.sigE = d_s[4].as<int16_t>(); A.phs5 = d_p[0].as<int8_t>(); A.phs3 = d_p[1].as<int8_t>();
    A.cano = d_c.as<uint8_t>(); A.dinc = d_d.as<uint8_t>();
    if (spdh_signals_run(ctx, model, std::vector<SigJobH>(1, J), A, 0)) return -1;
    int16_t* outs[5] = {sig5, sig3, sigS, sigT, sigE};
    for (int i = 0; i < 5; ++i) if (outs[i]) HIPCHK(hipMemcpy(outs[i], d_s[i].p, 2 * N, hipMemcpyDeviceToHost));
    if (phs5) HIPCHK(hipMemcpy(phs5, d_p[0].p, N, hipMemcpyDeviceToHost));
    if (phs3) HIPCHK(hipMemcpy(phs3, d_p[1].p, N, hipMemcpyDeviceToHost));
    if (dinc) HIPCHK(hipMemcpy(dinc, d_d.p, N, hipMemcpyDeviceToHost));
    return 0;
}
