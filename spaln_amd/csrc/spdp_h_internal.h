// spdp_h_internal.h -- kernel argument blocks of the aa x genome path (spdp_h_kernels.hip / spdp_h_api.cpp)
#ifndef SPDP_H_INTERNAL_H_
#define SPDP_H_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spdp_h_dev.h"
#include "spdp_ipen_runs.h"

struct HSweepArgs {
    const DevScoringH* sc;
    const DevProblemH* probs;
    int                n_probs;
    const uint8_t*     a_codes;
    const int4*        cols;
    const short4*      aux;
    int2*              bnd;
    uint16_t*          tb;
    DevResultH*        res;
};

struct HWalkArgs {
    const DevProblemH* probs;
    int                n_probs;
    const uint16_t*    tb;
    const DevResultH*  res;
    int2*              skl;       // per problem: skl_cap records
    int*               n_skl;     // records; -1 overflow, -2 "Unexpected dir", -3 start outside the bitmap
    int                skl_cap;
};

struct HUdhArgs {
    const DevScoringH* sc;
    const DevProblemH* probs;
    int                n_probs;
    const uint8_t*     a_codes;
    const int4*        cols;
    const short4*      aux;
    int4*              bnd;
    int*               imd;
    DevResultH*        res;
};

struct HCposArgs {
    const DevProblemH* probs;
    int                n_probs;
    int*               imd;
    const DevResultH*  res;
    int*               cpos;       // per problem cpos_stride ints ((n_im_max + 1) Dim10 rows)
    int*               ranges;     // per problem 4 ints
    int*               scores;
    int                cpos_stride;
};

#ifndef SPDP_VMF_CHUNK
#define SPDP_VMF_CHUNK 512          // Vmf record numbers a wave of a pipelined forward problem reserves at a time
#endif
// the -A0 engines forwardH_ng / hirschbergH_ng (spdp_h_rowwave.hip): one wave per problem, lane = row
struct HScalarArgs {
    const DevScoringH* sc;
    const DevProblemH* probs;      // bnd_off: into work (ints), tb_off: into vmf (records), imd_off: record capacity
    int                n_probs;
    const uint8_t*     a_codes;
    const int4*        cols;
    const short4*      aux;
    const int16_t*     intpen;     // IntronPenalty::Penalty(len)
    const int*         cip;        // Cip_score::cip_score(c) rows (3 a_len + 2 ints) of the queries that have one, or null
    const int16_t*     ipen_runs;  // SPDP_IPR_WORDS words (spdp_intpen_runs), or null: long introns read `intpen`
    int                intpen_len;
    int                minl;       // IntronPrm.minl
    int                gape1, gape2, extragop;
    int                noll, lgop; // PwdB::Noll (3: double affine gaps, spdh_rowwave<., ., ., true>), LongGOP
    int16_t            t53[256];
    uint8_t            mid[32];    // middle base a tron code pins (4: none)
    uint8_t            tron_of[64];
    int*               work;       // per problem 3 * (2 * width + 8) ints: two rows of {val, ptr, dir}
    int3*              vmf;        // Vmf records {m, n, prev}
    DevResultH*        res;
    int2*              skl;        // per problem skl_cap records
    int*               n_skl;      // records; -1 skl overflow, -3 Vmf capacity exceeded
    int                skl_cap;
    // hirschbergH_ng only
    int*               imd;        // per problem n_im * 8 * width ints at DevProblemH::imd_off
    int*               cpos;       // per problem cpos_stride ints
    int*               ranges;     // per problem 4 ints
    int*               scores;
    int                cpos_stride;
    // tiles of a problem as separate waves (spdh_rowwave<., true>): null = one wave per problem
    int*               pipe;       // per problem pipe_stride ints {records, overflow, prog[max_tiles], best[max_tiles][8], rlf[n_im][3]}; then {ticket, stalled}
    int                pipe_stride, pipe_ticket, max_tiles;
    const int2*        items;      // (problem, tile) in dispatch order
    int                n_items;
    int                item_probs; // spdh_exact: problems per wave (1, 2, 4); a pipelined item is (item_probs problems, stripe)
};

// protein-side signal precompute (spdp_signals_h.hip): Exinon::intron53_c / intron53_p for tron windows
struct SigPatMatDev { int32_t rows, cols, offset, order; float tonic, min_elem; };
struct SigModelHDev {
    SigPatMatDev pm5, pm3, pmI, pmT, pmB;
    int32_t pot_ndata, any, dvsp, trm, trm2, maxb3d;
    float   fE, fT, fO, fS, fs, tonic5, tonic3, fB, thB;
    int16_t tab5[16], tab3[16];
};
struct SigJobH {
    int64_t b_off;                 // first tron code of the window in `codes`
    int64_t out_off;               // its plain arrays (b_len + 3 entries each)
    int64_t col_off;               // its column records (HStore layout)
    int32_t b_len, left, right, pad;
};
struct SignalArgsH {
    const SigModelHDev* model;
    const float*        mtx;       // pm5, pm3, pmI, pmT, pmB matrices back to back
    int32_t*            sb;        // branch-point term on: per position (int16) (fB x score) of a site above the threshold,
                                   // INT32_MIN elsewhere (spdh_signals writes it, spdh_signal_phases folds it into sig3)
    const float*        pot;       // coding potential, 3 * pot_ndata floats
    const SigJobH*      jobs;
    const uint8_t*      codes;
    int16_t*            sig5;  int16_t* sig3;  int16_t* sigS;  int16_t* sigT;  int16_t* sigE;
    int8_t*             phs5;  int8_t*  phs3;
    uint8_t*            cano;      // cano5 | cano3 << 4 (levels)
    uint8_t*            dinc;      // dinc5 << 4 | dinc3
    int4*               cols;      // packed records (pack != 0)
    short4*             aux;
    int32_t             ipen;      // IntronPenalty::Penalty() or SPDH_NEV when splicing is off (as HStore::upload)
};
extern "C" hipError_t spdh_launch_signals(const SignalArgsH* a, int n_jobs, int max_len, int lds_floats, int pack, hipStream_t s);

extern "C" hipError_t spdh_launch_scalar(int forward, const HScalarArgs* a, hipStream_t s);
extern "C" hipError_t spdh_launch_scalar_udh(const HScalarArgs* a, hipStream_t s);
extern "C" hipError_t spdh_launch_exact(int udh, const HScalarArgs* a, hipStream_t s);   // -A1: forwardH1 / hirschbergH1
extern "C" hipError_t spdh_launch_local_udh(const HScalarArgs* a, hipStream_t s);        // hirschbergH1_wip, -LS
extern "C" hipError_t spdh_launch_udh(const HUdhArgs* a, int spj, int pen_cap, hipStream_t s);
extern "C" hipError_t spdh_launch_cpos(const HCposArgs* a, hipStream_t s);
extern "C" hipError_t spdh_launch_sweep(const HSweepArgs* a, int spj, int pen_cap, int local, hipStream_t s);
extern "C" hipError_t spdh_launch_walk(const HWalkArgs* a, hipStream_t s);
#endif
