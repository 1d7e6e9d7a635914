// spdp_internal.h -- declarations shared by spdp_kernels.hip, spdp_api.cpp and spdp_host.cpp
#ifndef SPDP_INTERNAL_H_
#define SPDP_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/spdp.h"
#include "spdp_dev.h"

struct SweepArgs {
    const DevScoring* sc;
    const DevProblem* probs;
    int               n_probs;
    const uint8_t*    a_codes;
    const int2*       cols;
    int*              bnd;        // int2 (score / forward) or int4 (udh) entries
    uint8_t*          tb;         // forward: traceback codes
    int*              imd;        // udh: hlnk0, hlnk1, vlnk0, vlnk1 per intermediate
    DevResult*        res;
    int*              queue;      // atomic problem counter
    volatile int*     dbg;        // optional host-pinned progress markers (debugging)
};

struct WalkArgs {
    const DevProblem* probs;
    int               n_probs;
    const uint8_t*    tb;
    const DevResult*  res;
    int2*             skl;        // per problem: skl_cap records
    int*              n_skl;      // per problem: number of records, -1 on overflow, -2 on bad code
    int               skl_cap;
};

struct CposArgs {
    const DevProblem* probs;
    int               n_probs;
    const int*        imd;
    const DevResult*  res;
    int*              cpos;       // per problem cpos_stride ints ((n_im_max + 1) Dim10 rows)
    int*              ranges;     // per problem 4 ints
    int*              scores;
    int               cpos_stride;
};

extern "C" hipError_t spdp_launch_sweep(int flavour, int local, const SweepArgs* args, int grid, hipStream_t s);
extern "C" hipError_t spdp_launch_walk(const WalkArgs* a, hipStream_t s);
extern "C" hipError_t spdp_launch_cpos(const CposArgs* a, hipStream_t s);

struct SpdpContext {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string name;
    std::string err;
};

// one packed, HBM-resident batch of problems for one sweep flavour (0 score, 1 forward, 2 udh)
struct DevBatch {
    SpdpContext* ctx = nullptr;
    int flavour = 0, n_probs = 0, local = 0;
    int max_n_im = 0, max_skl = 0;
    int64_t total_cells = 0, tb_bytes = 0;
    std::vector<DevProblem> h_probs;
    void *d_sc = nullptr, *d_probs = nullptr, *d_a = nullptr, *d_cols = nullptr, *d_bnd = nullptr,
         *d_tb = nullptr, *d_imd = nullptr, *d_res = nullptr, *d_queue = nullptr, *d_skl = nullptr,
         *d_nskl = nullptr, *d_cpos = nullptr, *d_ranges = nullptr, *d_scores = nullptr;
    DevBatch() = default;
    DevBatch(const DevBatch&) = delete;
    DevBatch& operator=(const DevBatch&) = delete;
    ~DevBatch() { release(); }
    int build(SpdpContext* c, const SpdpScoring* sc, const SpdpProblem* probs, int n,
              const SpdpWindow* wdws, const int* n_im, int flav);
    int run(float* kernel_ms);
    int fetch_results(std::vector<DevResult>& out);
    void release();
};

// engine entry points with explicit bands (used by the dispatch layer)
int spdp_wip_forward_w(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                       const SpdpWindow* wdws, int n_probs, SpdpAlignment* out);
int spdp_wip_udh_w(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs,
                   const SpdpWindow* wdws, int n_probs, const int* n_im, int cpos_rows,
                   int32_t* scores, int32_t* cpos, int32_t* ranges);
#endif
