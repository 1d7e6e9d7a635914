// spdp_internal.h -- declarations shared by spdp_kernels.hip, spdp_api.cpp and spdp_host.cpp
#ifndef SPDP_INTERNAL_H_
#define SPDP_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/spdp.h"
#include "spdp_dev.h"
#include "spdp_hostcpus.h"

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
    int               n_multi;    // the first n_multi problems get a whole 4-wave block each
    // cross-CU pass pipelines (CROSS kernels): every problem is spread over cross_g blocks
    int               cross_g;    // blocks per problem (0: off)
    int*              gprog;      // per problem: cross_g * WPB progress words, then 2 barrier words; zeroed per launch
};

#define HIPCHK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

struct WalkArgs {
    const DevProblem* probs;
    int               n_probs;
    const uint8_t*    tb;
    const DevResult*  res;
    int2*             skl;        // per problem: skl_cap records
    int*              n_skl;      // per problem: number of records, -1 on overflow, -2 on bad code
    int               skl_cap;
    int               seq;        // 1: one diagonal step per load (SPDP_WALK_SEQ=1, for A/B runs)
};

#define SPDP_RLST_INHERITED 0x7ffffff0   // pipelined linear-space engines: "rlst as the intermediate rows above left it" (an HLNK value)
struct CposArgs {
    const DevProblem* probs;
    int               n_probs;
    const int*        imd;
    const DevResult*  res;
    int*              cpos;       // per problem cpos_stride ints ((n_im_max + 1) Dim10 rows)
    int*              ranges;     // per problem 4 ints
    int*              scores;
    int               cpos_stride;
    int               strict;     // 1: the -A1 form of the walk (hirschbergS1: r > up, lw <= vlnk)
    int               local;      // 1: local ends (DevResult::ml is the left-end row of the path)
    int*              edge;       // per problem: 1 = the path crosses an intermediate row on or left of the first column, or is
                                  // empty: the one class where the reference's own linear-space result depends on what a previous
                                  // stripe left in its link lanes -- reported, not imitated; may be null
    const int*        pipe;       // pipelined spdp_exact<2>: the sync words of the launch (rlf[] resolves SPDP_RLST_INHERITED), or null
    int               pipe_stride, rlf_off;
};

extern "C" hipError_t spdp_launch_sweep(int flavour, int local, int nquant, int pen_cap, const SweepArgs* args,
                                        int grid, int wpb, hipStream_t s);
// spdp_sweep_fp.hip: the fp32-issue form of the score-only / linear-space sweeps; hipErrorNotSupported = use spdp_launch_sweep
extern "C" int spdp_sweep_fp_serves(int local, int spj, int nquant, int pen_cap, int llmt);
extern "C" hipError_t spdp_launch_sweep_fp(int flavour, int local, int spj, int nquant, int pen_cap, int llmt,
                                           const SweepArgs* args, int grid, int wpb, hipStream_t s);
extern "C" hipError_t spdp_launch_walk(const WalkArgs* a, hipStream_t s);
extern "C" hipError_t spdp_launch_cpos(const CposArgs* a, hipStream_t s);
// exact-intron-length (-A0) engines (spdp_rowwave.hip: one wave per problem, lane = row) and the -A1 engines
#include "spdp_ipen_runs.h"

#define SPDP_VMF_LANE_CHUNK 32      // Vmf record numbers a lane of forwardS1 (spdp_exact<1>) reserves at a time
#ifndef SPDP_VMF_CHUNK
#define SPDP_VMF_CHUNK 512          // Vmf record numbers a wave of a pipelined forwardS_ng problem reserves at a time
#endif
struct ScalarArgs {
    const DevScoring* sc;
    const DevProblem* probs;      // bnd_off = work offset (ints), tb_off = Vmf offset (records), imd_off = Vmf capacity
    int               n_probs;
    const uint8_t*    a_codes;
    const int2*       cols;
    const uint8_t*    aux;        // per position: {bit0 isDonor | bit1 isAccpt, dinc5 << 4 | dinc3}
    const int*        cip;        // Cip_score::cip_score(m) rows of the queries that have one (DevProblem::cip_off), or null
    const int16_t*    intpen;
    int               intpen_len;
    int               ipen;
    const int16_t*    ipen_runs;  // SPDP_IPR_WORDS words (spdp_intpen_runs), or null: long introns read `intpen`
    int               minl;       // IntronPrm.minl (-A1 engines)
    int16_t           t53[256];
    int*              work;
    int3*             vmf;
    DevResult*        res;
    int2*             skl;
    int*              n_skl;
    int               skl_cap;
    // scalar UDH (hirschbergS_ng) only
    int*              imd;        // per problem n_im * 8 * width ints at DevProblem::imd_off
    int*              cpos;       // per problem cpos_stride ints
    int*              ranges;     // per problem 4 ints
    int*              scores;
    int               cpos_stride;
    // tiles of a problem as separate waves (spdp_rowwave<., true>): null = one wave per problem
    int*              pipe;       // per problem pipe_stride ints {records, overflow, prog[max_tiles], best[max_tiles][4]}; then {ticket, stalled}
    int               pipe_stride, pipe_ticket, max_tiles;
    const int2*       items;      // (problem, tile) in dispatch order
    int               n_items;
    // double affine gaps (spdp_rowwave<., ., ., DAGP>): PwdB::Noll, LongGOP, LongGEP, codonk1
    int               noll, lgop, lgep, codonk1;
};
extern "C" hipError_t spdp_launch_rowwave(int forward, const ScalarArgs* a, hipStream_t s);    // spdp_rowwave.hip: -A0 forward / score-only
extern "C" hipError_t spdp_launch_rowwave_udh(const ScalarArgs* a, hipStream_t s);            // spdp_rowwave.hip: -A0 linear space
extern "C" hipError_t spdp_launch_local_udh(const ScalarArgs* a, hipStream_t s);          // spdp_local_udh.hip: hirschbergS1_wip, -LS
extern "C" hipError_t spdp_launch_exact(int mode, const ScalarArgs* a, hipStream_t s);   // spdp_exact.hip: 0 score, 1 forward, 2 udh
extern "C" hipError_t spdp_launch_pack(const int2* skl, int skl_cap, const int* n_skl, const int64_t* off,
                                       int2* packed, int n_probs, hipStream_t s);

// splice-signal precompute (spdp_signals.hip): Exinon::intron53_c / intron53_n per position of a genomic window
struct SigModelDev {
    int32_t rows, cols5, off5, cols3, off3, any, both_ori;
    float   fs, tonic5, min5, tonic3, min3;
    int16_t tab5[16], tab3[16];
};
struct SigJob {
    int64_t b_off;                // first base of the window in `codes`
    int64_t col_off;              // its column records (DevStore layout)
    int64_t out_off;              // its plain output arrays
    int32_t b_len, left, right, pad;
};
struct SignalArgs {
    const SigModelDev* model;
    const float*       mtx5;      // cols5 x rows
    const float*       mtx3;
    const SigJob*      jobs;
    const uint8_t*     codes;
    int16_t*           sig5;      // plain arrays, b_len + 1 entries per job at out_off (or null)
    int16_t*           sig3;
    uint8_t*           cano5;
    uint8_t*           cano3;
    uint8_t*           dinc;
    int2*              cols;      // column records / aux of the sweeps (or null)
    uchar2*            aux;
    int*               maxes;     // {max(sig5 + ipen), max sig3} over everything written (or null)
    int32_t            ipen, spj;
};
int spdp_signals_run(SpdpContext* ctx, const SpdpSignalModel* m, const std::vector<SigJob>& jobs, SignalArgs args,
                     int* max_s5, int* max_s3);          // spdp_signals_api.cpp
extern "C" hipError_t spdp_launch_signals(const SignalArgs* a, int n_jobs, int max_len, int lds_floats, hipStream_t s);

// skl_rngS_ng on the device (spdp_rescore.hip): one thread per query
struct RescoreArgs {
    const DevScoring* sc;
    const DevProblem* probs;
    int               n_probs;
    const uint8_t*    a_codes;
    const int2*       cols;
    const uint8_t*    aux;
    const int16_t*    intpen;
    int               intpen_len;
    const int2*       skl;        // corner lists without header, query i at skl_off[i], skl_cnt[i] corners
    const int64_t*    skl_off;
    const int*        skl_cnt;
    const int64_t*    rec_off;    // first exon record of query i (21 ints each)
    int*              out_hdr;    // per query 8 ints: h, mch, mmc, gap, unp, val, n_records, 0
    int*              out_rec;
    int               gop, gep, lgop, lgep, codonk1, minl, jneibr, lsg, ipen;
    int16_t           t53[256];
    // edit records for the Cigar / Vulgar / SAM writers (0: none)
    int               ops_format; // 1 Cigar, 2 Vulgar, 3 SAM
    int3*             ops;        // query i: records at ops_off[i] .. ops_off[i + 1]
    const int64_t*    ops_off;
    int*              ops_cnt;    // records the walk produced (may exceed the slot: then the slot is full and the rest lost)
    const int*        a_len;      // SAM: query lengths
    const int*        cip;        // Cip_score::cip_score(m) rows (DevProblem::cip_off), or null: the bonus of use_spb()
    const int*        a_len_all;  // query lengths (always set when cip is)
};
extern "C" hipError_t spdp_launch_rescore(const RescoreArgs* a, hipStream_t s);

// skl_rngH_ng on the device
struct HRescoreProb {
    int32_t a_left, a_right, b_left, b_right;
    int32_t a_len, b_len;
    int32_t a_exgl, a_exgr, b_exgl, b_exgr;
    int64_t a_off, b_off, col_off;
    int32_t w_lo, w_hi;               // the region positions the launch holds: [w_lo, w_hi) of the problem's b_len + 3 (the corners' span and a margin)
};
struct HRescoreArgs {
    const int*        mtx;        // stride 32: mtx[aa * 32 + tron]
    int               n_probs;
    const HRescoreProb* probs;
    const uint8_t*    a_codes;
    const uint8_t*    b_codes;    // tron codes, b_len + 1 per problem
    const short*      sig;        // per position 5 shorts: sig5, sig3, sigS, sigT, sigE
    const int8_t*     phs;        // per position 2: phs5, phs3
    const uint8_t*    dinc;
    const int16_t*    intpen;
    int               intpen_len;
    const int2*       skl;
    const int64_t*    skl_off;
    const int*        skl_cnt;
    const int64_t*    rec_off;
    int*              out_hdr;
    int*              out_rec;
    int               gop, gep, lgop, lgep, codonk1, gape1, gape2, extragop, diffu, k1;
    int               minl, jneibr, lcl, sup_tcodon;
    int16_t           t53[256];
    uint8_t           mid[32];    // tron code -> its codon's middle base (0..3), 4 = ambiguous
    uint8_t           tron_of[64];// codon (A C G T order) -> tron code
    // edit records for the Cigar / Vulgar writers (0: none; 1 Cigar, 2 Vulgar), as RescoreArgs
    int               ops_format;
    int3*             ops;
    const int64_t*    ops_off;
    int*              ops_cnt;
};
extern "C" hipError_t spdh_launch_rescore(const void* args, hipStream_t s);

// grow-only device allocations reused across launches (hipMalloc / hipFree of multi-GB
// work buffers per batch costs seconds)
struct DevPool {
    enum { N_SLOTS = 16 };
    void*  ptr[N_SLOTS] = {nullptr};
    size_t cap[N_SLOTS] = {0};
    void*  get(int slot, size_t bytes);
    void   release();
};
enum { POOL_PROBS = 0, POOL_BND, POOL_TB, POOL_IMD, POOL_RES, POOL_SKL, POOL_NSKL, POOL_CPOS,
       POOL_RANGES, POOL_SCORES, POOL_SKLPACK, POOL_SKLOFF, POOL_GPROG, POOL_FLAV_STRIDE = 0 };

// a blocking copy.  hipMemcpy goes through the null stream, which first waits for every blocking stream of the process.  For the
// request batches of the seeded path -- several dispatcher lanes at work, a 16-byte read-back of the short class would wait
// for the long class's multi-millisecond sweep -- the copy waits for its OWN stream only (t_lane_copies, set by
// spdp_run_requests / spdh_run_requests for the duration of the call).  Everywhere else the null-stream form stays: in the
// chunked ladder of spdp_align_s it keeps one chunk's slab sweeps from starting under the other chunk's linear-space sweep,
// which changes nothing end to end (290 ms a step either way) but is how the kernels of the headline step were profiled.
extern thread_local bool t_lane_copies;
struct LaneCopies { bool was; LaneCopies() : was(t_lane_copies) { t_lane_copies = true; } ~LaneCopies() { t_lane_copies = was; } };
static inline hipError_t spdp_copy_sync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s)
{
    if (!t_lane_copies) return hipMemcpy(dst, src, n, kind);
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}

struct SpdpContext {
    DevPool pool[9];                 // one pool per engine flavour (they coexist in a pipeline); [5], [6] = aa x genome path, [7] = rescoring,
                                     // [8] = the forward run on the side stream (coexists with a regular forward run)
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream2 = nullptr;   // side stream (non-blocking): a forward run beside the linear-space rounds
    hipEvent_t ev2 = nullptr, ev3 = nullptr;
    std::string name;
    std::string err;
    std::vector<SpdpContext*> lanes;  // further lanes of this context (spdp_lane): chunks of a batch run side by side
    int64_t seed_stats[12] = {0};      // spdp_seeded_stats
    std::vector<std::vector<SpdpPhaseMark>> seed_marks;    // spdp_seeded_phase_marks: per query of the last spdp_align_h_seeded call
    int64_t rerun_stats[2] = {0, 0};   // launches repeated because a cross-CU group / a tile pipeline gave up (spdp_rerun_stats)
    void*  stage_ptr[3] = {nullptr, nullptr, nullptr};   // pinned host staging (grow-only): [0], [1] DevStore::upload, [2] the regions and
    size_t stage_cap[3] = {0, 0, 0};                     // signal arrays of spdp_map_align_s
    void*  staging(int k, size_t bytes);
};
SpdpContext* spdp_lane(SpdpContext* ctx, int i);

// Resident inputs of a set of parent problems: residues, per-position column
// records and the scoring bundle.  Sub-problems (UDH slabs, engine calls on
// sub-ranges) are descriptors pointing into these arrays -- no re-upload.
struct DevStore {
    SpdpContext* ctx = nullptr;
    SpdpScoring sc;
    int n_parents = 0;
    std::vector<int64_t> a_off, col_off;
    std::vector<int32_t> a_len, b_len;
    void *d_sc = nullptr, *d_a = nullptr, *d_cols = nullptr, *d_aux = nullptr, *d_intpen = nullptr, *d_cip = nullptr;
    void* d_ipen_runs = nullptr;            // spdp_intpen_runs() of the table, when it has that shape
    std::vector<int32_t> cip_off;           // per parent: first entry of its cip row in d_cip, -1 = none
    bool has_exact = false;                 // exact-model inputs (cano / dinc / intpen) were supplied
    int fp_maxpos = 1, fp_gain = 0;         // largest substitution score; best net score of one intron (>= 0)
    DevStore() = default;
    DevStore(const DevStore&) = delete;
    DevStore& operator=(const DevStore&) = delete;
    ~DevStore() { release(); }
    int upload(SpdpContext* c, const SpdpScoring* sc, const SpdpProblem* probs, int n);
    void release();
};

// one DP call on (a sub-range of) a parent problem
struct RunItem {
    int parent;
    int a_left, a_right, b_left, b_right;
    uint8_t a_exgl, a_exgr, b_exgl, b_exgr;
    SpdpWindow w;
    int n_im;          // UDH only
    int imd_intvl = 0; // scalar UDH only: Aln2s1::imd_intvl as lspS_ng set it
    int vmf_scale = 1; // Vmf-backed forward runs: 1 = the usual record budget, larger on the retry after an overflow
    int cut_l = 0, cut_r = 0;   // scalar forward only: forwardS_ng's cut range (cut_r > cut_l), see spdp_rowwave<1, false, true>
};

// descriptors + work buffers of one engine flavour over a DevStore:
// 0 score, 1 forward, 2 udh (the _wip sweeps); 3 scalar exact forward, 4 scalar exact score,
// 5 scalar udh, 6 -A1 score-only (both share pool 4 with the scalar score run: they never coexist),
// 7 -A1 forward (shares pool 3 with the scalar forward run), 8 -A1 udh (pool 4)
struct DevRun {
    SpdpContext* ctx = nullptr;
    SpdpContext* use_ctx = nullptr;         // set before build: the lane to run on (default: the store's context)
    const DevStore* store = nullptr;
    int flavour = 0, n = 0;
    int n_multi = 0;                        // leading problems run as multi-wave pipelines
    int wpb = 4;                            // waves per block of the sweep launch (16: one huge problem per CU)
    int cross_g = 0;                        // > 0: every problem spread over this many 16-wave blocks (CUs)
    bool fp_ok = false;                     // scores stay inside the exact fp32 range: spdp_sweep_fp.hip may run it
    void* d_gprog = nullptr;                // progress / barrier words of the cross-CU pipelines
    // -A0 wavefront engines with the tiles of a problem as separate waves (spdp_rowwave<., true>): shares d_gprog
 bool cut = false;                       // flavour 3 with cut ranges on every item (set by build)
    bool pipe_on = false;
    int pipe_tiles = 0;                     // most tiles of one problem
    int pipe_stride = 0;                    // sync words per problem
    size_t pipe_words = 0;                  // sync words ahead of the item list
    std::vector<int> h_items;               // (problem, tile) pairs in dispatch order
    int max_n_im = 0, max_skl = 0;
    int64_t total_cells = 0, tb_bytes = 0;
    std::vector<DevProblem> h_probs;        // in dispatch order
    std::vector<int> order;                 // dispatch slot -> caller index
    void *d_probs = nullptr, *d_bnd = nullptr, *d_tb = nullptr, *d_imd = nullptr, *d_res = nullptr,
         *d_skl = nullptr, *d_nskl = nullptr, *d_cpos = nullptr,
         *d_ranges = nullptr, *d_scores = nullptr;         // all owned by ctx->pool[flavour]
    int skl_cap = 0;
    float kernel_ms = 0.f;
    bool side = false;                      // run on ctx->stream2 (set before build)
    bool in_flight = false;                 // launched, not yet waited for
    bool beside = false;                    // another kernel fills the GPU meanwhile: keep to 4-wave blocks (a 16-wave
                                            // block finds no CU with room while small blocks keep refilling them)
    hipStream_t strm() const { return side ? ctx->stream2 : ctx->stream; }
    hipEvent_t evb() const { return side ? ctx->ev2 : ctx->ev0; }
    hipEvent_t eve() const { return side ? ctx->ev3 : ctx->ev1; }
    DevRun() = default;
    DevRun(const DevRun&) = delete;
    DevRun& operator=(const DevRun&) = delete;
    ~DevRun() { release(); }
    int build(const DevStore* st, const std::vector<RunItem>& items, int flav);
    int launch();                       // async on ctx->stream: sweep (+ walk / cpos)
    int sync();                         // waits, fills kernel_ms
    int fetch_results(std::vector<DevResult>& out);
    // forward: records of problem i are skl[off[i] .. off[i] + n_skl[i])
    int fetch_skl(std::vector<int>& n_skl, std::vector<int64_t>& off, std::vector<SpdpSkl>& skl);
    int fetch_udh(std::vector<int32_t>& scores, std::vector<int32_t>& cpos, std::vector<int32_t>& ranges,
                  std::vector<int32_t>* edge = nullptr);     // edge: CposArgs::edge per problem (flavours 2, 8, 9), zeros otherwise
    void release();
};

RunItem spdp_item_of(const SpdpProblem& p, int parent, int sh);
// explicit engine calls on sub-ranges of store entries (spdp_host.cpp: spdp_run_requests); all arrays indexed by request
struct SpdpRequests {
    const int* parents;            // store entry
    const SpdpWindow* windows;
    const uint8_t* kinds;          // 0 lspS_ng, 1 trcbkalignS_ng, 2 trcbkalignS_ng with a cut range
    const int* cuts;               // 2 per request: cut_l, cut_r (kind 2)
};
int spdp_run_requests(SpdpContext* ctx, const DevStore* st, const SpdpProblem* probs, int n, const SpdpRequests* req,
                      SpdpAlignment* out);
void trim_skl_of(std::vector<SpdpSkl>& s, const SpdpProblem& p);     // trimskl, src/gaps.cc:254-273
int64_t spdp_cells_w(int a_left, int a_right, int b_left, int b_right, const SpdpWindow& w);
// corner list of path records (spdp_host.cpp): M_UNIT = 1 nucleotide rows, 3 protein rows
template <int M_UNIT> std::vector<SpdpSkl> corner_list(std::vector<SpdpSkl> pts);

#endif
