/* spdp.h -- C ABI of the MI355X spliced-alignment DP engine (libspdp_hip.so).
 *
 * Drop-in boundary for the fine-alignment path of ogotoh/spaln v3.0.7.  The
 * reference has no FFI layer; the surface a binding replaces is
 *   - the free functions declared in src/aln.h:348-357
 *       VTYPE HomScoreS_ng(const Seq* seqs[], const PwdB* pwd)      src/fwd2s1.cc:2696
 *       SKL*  alignS_ng(Seq* seqs[], const PwdB*, Gsinfo*, int ori) src/fwd2s1.cc:2746
 *   - the engine object they construct:  SimdAln2s1(seqs, pwd, wdw, spjcs, cip, mode, vmf)
 *       src/fwd2s1_simd.h:191, with methods
 *       scoreonlyS1_wip()            src/fwd2s1_wip_simd.h:42
 *       forwardS1_wip(Mfile*)        src/fwd2s1_wip_simd.h:233
 *       hirschbergS1_wip(cpos, n_im) src/fwd2s1_wip_simd.h:476
 *   - the dispatch between them: Aln2s1::lspS_ng / trcbkalignS_ng / mimd_postwork
 *       src/fwd2s1.cc:1801 / 1667 / 1714.
 * Every pointer below is plain host memory unless the name says "_dev".
 * Plain C types only; no C++/torch types cross this boundary.
 *
 * A "problem" is everything one DP call of the reference reads (SURVEY.md §8a
 * row a18): residue codes of both sequences, the active sub-ranges, the
 * per-genome-position splice signals, substitution matrix, gap and intron
 * parameters, band and end-gap flags.  Positions are absolute (the reference's
 * Seq::left/right convention): DP row m in (a_left, a_right] is residue
 * a[m-1]; DP column n in (b_left, b_right] is b[n-1]; sig5/sig3 are indexed by
 * n in [b_left, b_right].
 */
#ifndef SPDP_H_
#define SPDP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPDP_MAX_QUANT   8
#define SPDP_NEVSEL      (INT32_MIN / 16 * 7)   /* "no alignment", src/cmn.h:79 NEVSEL   */
#define SPDP_END_OF_ULK  (INT32_MAX - 2)        /* src/aln.h:49 end_of_ulk              */
#define SPDP_REF_NELEM   16                     /* int16 lanes of the AVX2 reference build */

/* scoring bundle shared by many problems: the PwdB / IntronPenalty / Simmtx
 * subset the _wip engines read (src/aln.h:235-308, src/codepot.h:223-257). */
typedef struct SpdpScoring {
    int32_t mtx_dim;                 /* Simmtx::dim (17 = NSIMD for DNA)                 */
    int32_t mtx[32 * 32];            /* row-major mtx[a * mtx_dim + b]                   */
    int32_t gop, gep;                /* PwdB::BasicGOP, BasicGEP (both <= 0)             */
    int32_t lgop, lgep;              /* PwdB::LongGOP, LongGEP                           */
    int32_t noll;                    /* PwdB::Noll: 2 affine, 3 double affine            */
    int32_t spj;                     /* b->inex.intr: splice-aware                       */
    int32_t llmt;                    /* IntronPrm.llmt: introns need hil > llmt          */
    int32_t ipen;                    /* IntronPenalty::Penalty() = GapWI, added to sig5  */
    int32_t nquant;                  /* IntronPrm.nquant (1 = flat, the -A3 model)       */
    int32_t qm_len[SPDP_MAX_QUANT];  /* IntronPenalty::qm[j].len                         */
    int32_t qm_pen[SPDP_MAX_QUANT];  /* IntronPenalty::qm[j].pen                         */
    int32_t local;                   /* algmode.lcl & 16                                 */
    int32_t sh;                      /* alprm.sh band shoulder used by stripe()          */
    int32_t max_vmf_space;           /* MaxVmfSpace (src/vmf.h:27), traceback/UDH switch */
    int32_t ubh;                     /* alprm.ubh: forced #intermediates, 0 = automatic  */
    int32_t ref_nelem;               /* stripe height of the reference build to reproduce */
    /* ---- exact intron-length model (scalar engines, -A0): optional, may be NULL / 0 ---- */
    const int16_t* intpen;           /* IntronPenalty::Penalty(len) for len in [0, intpen_len)  */
    int32_t intpen_len;              /*   (src/codepot.h:242-247); must cover the longest window */
    int16_t t53[256];                /* Exinon::sig53(m, n, IE53) - sig3[n], by 16*dinc5[m]+dinc3[n]
                                        (src/codepot.cc:411-415)                              */
    int32_t scalar_engines;          /* 0: the `_wip` engines (-A2 / -A3, the default of the reference);
                                        1: algmode.alg == 0 (-A0): spdp_align_s runs forwardS_ng /
                                        hirschbergS_ng throughout, spdp_homscore_s scorealoneS_ng;
                                        2: algmode.alg == 1 (-A1): spdp_homscore_s runs scoreonlyS1,
                                        spdp_align_s forwardS1 / hirschbergS1 */
    int32_t minl;                    /* IntronPrm.minl: shortest intron of the -A1 engines (0 = llmt) */
    int32_t recursive;               /* algmode.alg & 4 (-A4 .. -A7): lspS_ng always takes the recursive
                                        linear-space branch (one intermediate row, halves of halves)   */
    const struct SpdpSignalModel* sigmodel;   /* optional: with it, problems whose sig5 / sig3 are NULL get their
                                        signals (and cano5 / cano3 / dinc) computed on the device from b[] alone */
    int32_t codonk1;                 /* PwdB::codonk1 (src/aln2.cc:114), read when noll == 3 only: gaps longer than this
                                        are priced with lgop / lgep by GapPenalty / GapExtPen (src/aln.h:275-282)     */
} SpdpScoring;

/* the splice-site model behind SGPT2::sig5 / sig3 (Exinon::intron53_n, src/codepot.cc:479-520): the two
 * second-order Markov position weight matrices (PatMat, src/utilseq.h:64-88, as read from the species'
 * parameter tables), the per-dinucleotide terms sig53tab[0 / 1][class] and the scale fs = fS * alprm2.sss. */
typedef struct SpdpSignalModel {
    int32_t rows;                    /* PatMat::rows = 84 (4 + 16 + 64 terms per column), order 2, 4 letters */
    int32_t cols5, off5;             /* pattern5: PatMat::cols, PatMat::offset                               */
    int32_t cols3, off3;             /* pattern3                                                             */
    float   fs;
    float   tonic5, min5;            /* PatMat::tonic, PatMat::min_elem                                      */
    float   tonic3, min3;
    const float* mtx5;               /* PatMat::mtx, cols5 * rows floats                                     */
    const float* mtx3;
    int16_t tab5[16], tab3[16];      /* sig53tab[0][dinc5], sig53tab[1][dinc3]                               */
    int32_t any;                     /* algmode.any (canonical-site levels, codepot.cc:438-475)              */
    int32_t both_ori;                /* Exinon::both_ori (ori == 3 callers)                                  */
} SpdpSignalModel;

typedef struct SpdpProblem {
    const uint8_t* a;  int32_t a_len;      /* query codes, a[0 .. a_len)                 */
    const uint8_t* b;  int32_t b_len;      /* genomic codes                              */
    const int16_t* sig5;                   /* SGPT2::sig5 per position, index 0 .. b_len */
    const int16_t* sig3;                   /* SGPT2::sig3                                */
    int32_t a_left, a_right;               /* active ranges (Seq::left / right)          */
    int32_t b_left, b_right;
    uint8_t a_exgl, a_exgr, b_exgl, b_exgr;/* Seq::inex.exgl / exgr (free end gaps)      */
    /* ---- exact model only (NULL for the _wip engines): per position, index 0 .. b_len --------- */
    const uint8_t* cano5;                  /* Exinon::isDonor(n)  (src/codepot.h:106)            */
    const uint8_t* cano3;                  /* Exinon::isAccpt(n)                                 */
    const uint8_t* dinc;                   /* INT53::dinc5 << 4 | INT53::dinc3 (src/codepot.h:49) */
    /* ---- optional, the -A0 / -A1 engines only (the `_wip` engines do not read it, as in the reference) ---------- */
    const int32_t* cip;                    /* Cip_score::cip_score(m) for query row m = 0 .. a_len (src/gsinfo.h:128-140,
                                              src/fwd2s1.cc:254, src/fwd2s1_simd.cc:50): the bonus every intron accepted
                                              in row m earns when the query carries conserved intron positions; NULL = none */
    /* ---- optional, the seeded path only (spdp_align_s_seeded): per position, index 0 .. b_len ------------------- */
    const int8_t* phs5;                    /* SGPT2::phs5 / phs3 (src/codepot.h:27-32) as Exinon::intron53_n leaves them        */
    const int8_t* phs3;                    /*   (src/codepot.cc:504-518); NULL: derived from cano5 / cano3 by the same rule     */
    /* ---- optional, with SpdpScoring.sigmodel only ---------------------------------------------------------------- */
    int32_t exin_left, exin_right;         /* Seq::left / right when the reference built its Exinon (the whole gene window): the
                                              device computes the signals over [exin_left, exin_right), so that a problem on a
                                              sub-range (an lspS_ng call between two HSPs) reads what the reference's engines
                                              read there; both 0: the problem's own b_left / b_right */
} SpdpProblem;

typedef struct SpdpWindow { int32_t lw, up, width; } SpdpWindow;   /* WINDOW, src/cmn.h:133 */
typedef struct SpdpSkl { int32_t m, n; } SpdpSkl;                  /* SKL,    src/cmn.h:130 */

/* result of an alignment call: corner list in the reference's SKL format
 * (skl[0].n = number of corners, skl[0].m = flags, corners follow start->end) */
typedef struct SpdpAlignment {
    int32_t  score;            /* gsi->scr: raw engine score, SPDP_NEVSEL on failure      */
    int32_t  n_skl;            /* entries in skl[] including the header record, 0 = none */
    SpdpSkl* skl;              /* owned by the library until spdp_free_alignments        */
    int32_t  flags;            /* SPDP_ALN_* below                                       */
    int32_t  reserved;
} SpdpAlignment;
/* a linear-space (hirschbergS1[_wip]) call of this query found its path on the free left edge of its window -- a crossing of
 * an intermediate row on or left of the first genomic column, or an empty optimum.  There the reference records link
 * lanes it never initialised (src/fwd2s1_wip_simd.h:524: what the previous stripe left behind, heap contents in the first
 * stripe), so its own result is run-dependent; the device starts those lanes from zero and may differ.
 * Planted-gene inputs with the default semi-global ends never set it. */
#define SPDP_ALN_LEFT_EDGE 1

typedef struct SpdpContext SpdpContext;

/* ---- lifetime -------------------------------------------------------- */
/* Creates a context bound to HIP device `device` (fails, returning NULL, when
 * no HIP device is available -- there is no CPU fallback). */
SpdpContext* spdp_create(int device);
void         spdp_destroy(SpdpContext* ctx);
const char*  spdp_last_error(const SpdpContext* ctx);
int          spdp_device_name(const SpdpContext* ctx, char* buf, int buflen);

/* band of one problem: stripe(seqs, &wdw, sh), src/aln2.cc:156-176 */
void spdp_stripe(const SpdpProblem* p, int sh, SpdpWindow* wdw);
/* DP cells the reference loops visit for one call (SURVEY.md §8d, fwd2s1.cc:249-256) */
int64_t spdp_cells(const SpdpProblem* p, const SpdpWindow* wdw);

/* ---- engine level (SimdAln2s1 methods), batched ----------------------- */
/* scoreonlyS1_wip over each problem with its own stripe() band.  scores[i]
 * receives what the reference method returns. */
/* Exinon::intron53_c + intron53_n for one genomic window on the device: arrays of b_len + 1 entries indexed by
 * position, computed for the range [left, right) and zero outside it like the reference's (any output may be NULL).
 * The two cells the reference itself leaves to stale memory (sig5 / cano5 of right - 1, sig3 / cano3 of left) are
 * computed from class 0 here. */
int spdp_splice_signals(SpdpContext* ctx, const SpdpSignalModel* model, const uint8_t* b, int32_t b_len,
                        int32_t left, int32_t right, int16_t* sig5, int16_t* sig3,
                        uint8_t* cano5, uint8_t* cano3, uint8_t* dinc);

int spdp_wip_scoreonly(SpdpContext* ctx, const SpdpScoring* sc,
                       const SpdpProblem* probs, int n_probs, int32_t* scores);

/* forwardS1_wip: score + the raw corner records the reference writes to its
 * Mfile (end -> start order, before stdskl/trimskl).  out[i].skl holds the
 * records without header; out[i].n_skl their count. */
int spdp_wip_forward(SpdpContext* ctx, const SpdpScoring* sc,
                     const SpdpProblem* probs, int n_probs, SpdpAlignment* out);

/* hirschbergS1_wip with n_im intermediate rows: cpos[i] points at
 * (n_im + 1) * 10 ints (Dim10 rows, src/udh_intermediate.h:90); ranges[i*4..]
 * receives the written-back a_left, a_right, b_left, b_right.  With SpdpScoring.local the
 * local-ends form runs (spdp_local_udh.hip). */
int spdp_wip_udh(SpdpContext* ctx, const SpdpScoring* sc,
                 const SpdpProblem* probs, int n_probs, int n_im,
                 int32_t* scores, int32_t* cpos, int32_t* ranges);

/* scalar exact-intron-length engines (-A0; also what trcbkalignS_ng uses below 8 query rows):
 * Aln2s1::forwardS_ng via trcbkalignS_ng (src/fwd2s1.cc:217, 1667) and scorealoneS_ng (:1163).
 * Need SpdpScoring.intpen / t53 and SpdpProblem.cano5 / cano3 / dinc. */
int spdp_scalar_forward(SpdpContext* ctx, const SpdpScoring* sc,
                        const SpdpProblem* probs, int n_probs, SpdpAlignment* out);
int spdp_scalar_scorealone(SpdpContext* ctx, const SpdpScoring* sc,
                           const SpdpProblem* probs, int n_probs, int32_t* scores);
/* Aln2s1::hirschbergS_ng (src/fwd2s1.cc:762-1104), the scalar linear-space engine, with n_im
 * intermediate rows imd_intvl rows apart (Aln2s1::imd_intvl as lspS_ng sets it, :1839 / :1850): cpos,
 * ranges as for spdp_wip_udh; cpos[..][8], [9] carry the diagonal bounds of each slab (what
 * mimd_postwork / rcsv_postwork use as its window under -A0); entries the reference leaves
 * uninitialised read end_of_ulk.  flags[i] = -3: the reference reads or writes outside its arrays on
 * this input (undefined there), 0 otherwise. */
int spdp_scalar_udh(SpdpContext* ctx, const SpdpScoring* sc,
                    const SpdpProblem* probs, int n_probs, int n_im, int imd_intvl,
                    int32_t* scores, int32_t* cpos, int32_t* ranges, int32_t* flags);

/* ---- Aln2 surface, batched --------------------------------------------- */
/* HomScoreS_ng: stripe() then scoreonlyS1_wip (-A2 / -A3); scorealoneS_ng under -A0
 * (SpdpScoring.scalar_engines) and for query ranges below 4 rows (src/fwd2s1.cc:2705), which needs the
 * exact-model inputs -- without them such problems come back SPDP_NEVSEL and the call returns 1. */
int spdp_homscore_s(SpdpContext* ctx, const SpdpScoring* sc,
                    const SpdpProblem* probs, int n_probs, int32_t* scores);

/* alignS_ng with a fixed orientation (ori = 1) and seeding off (-Q0/-Q4):
 * stripe() -> lspS_ng decision ladder -> traceback or multi-intermediate UDH
 * + per-slab traceback -> stdskl -> trimskl. */
int spdp_align_s(SpdpContext* ctx, const SpdpScoring* sc,
                 const SpdpProblem* probs, int n_probs, SpdpAlignment* out);

/* alignS_ng(seqs, pwd, gsi, ori = 3) with seeding off (src/fwd2s1.cc:2746-2760): infer_orientation
 * (:2718-2730) = HomScoreS_ng on the query as given (fwd[i]) and on its reverse complement against the
 * opposite genomic strand (rev[i]: comrev(a) + antiseq(b), with that strand's own signals), the reverse
 * wins only with a strictly higher score; then one alignment.  orient[i] = 0 / 1 says which problem
 * out[i] refers to; an alignment of the flipped pair carries A_RevCom (0x10) in its header record, as
 * globalS_ng sets it (src/fwd2s1.cc:2691-2692). */
int spdp_align_s_ori3(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* fwd, const SpdpProblem* rev,
                      int n_probs, SpdpAlignment* out, int32_t* orient);
void spdp_free_alignments(SpdpAlignment* out, int n);

/* Aln2s1::lspS_ng (src/fwd2s1.cc:1817-1880) for a caller that keeps the record file itself -- the seeded path's
 * interpolateS (:2497-2514): the ladder below spdp_align_s, but out[i].skl holds the Mfile records as written (n_skl of
 * them, no header record, in no particular order: globalS_ng's stdskl sorts) and nothing is trimmed. */
int spdp_lsp_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, SpdpAlignment* out);

/* ---- batching of single-problem calls from many host threads ----------------------------------
 * The reference's seeded path walks one query per CPU thread and calls lspS_ng synchronously for the gaps between
 * HSPs; its thread pool runs many walks at once.  A collector owns `ctx` (do not use it elsewhere meanwhile) and one
 * scoring bundle; spdp_collector_align_s blocks the calling thread until the batch its problem joined has run:
 * a batch closes at max_batch problems or max_wait_us after its first arrival.  raw_records = 1: spdp_lsp_s results,
 * 0: spdp_align_s results.  Returns 0, 1 (this query came back without an alignment, see spdp_align_s) or -1. */
typedef struct SpdpCollector SpdpCollector;
SpdpCollector* spdp_collector_create(SpdpContext* ctx, const SpdpScoring* sc, int max_batch, int max_wait_us, int raw_records);
void spdp_collector_destroy(SpdpCollector* c);
int spdp_collector_align_s(SpdpCollector* c, const SpdpProblem* p, SpdpAlignment* out);
const char* spdp_collector_last_error(const SpdpCollector* c);
int spdp_collector_stats(SpdpCollector* c, int64_t* n_requests, int64_t* n_batches, int64_t* largest_batch);

/* ---- the seeded path: alignS_ng with algmode.qck = 1 .. 3 (-Q5 .. -Q7) ----------------------------
 * Aln2s1::globalS_ng -> seededS_ng -> interpolateS (src/fwd2s1.cc:2587-2694, 2405-2539): the HSPs of a query (b->jxt,
 * what the block search or geneorient() left there) are joined pairwise by closed-form rules -- abutting HSPs, an
 * indel-free junction inside an overlap (indelfreespjS), a plain gap (backforth), terminal exons found by exact search
 * (first_exon / last_exon), micro exons, the X-drop extensions of open ends -- and, where none applies, by the DP engines
 * (lspS_ng, trcbkalignS_ng with or without a cut range).  The walks of all queries of a call run side by side on host
 * threads; a walk that needs a DP result waits, and when every walk in flight waits (or has ended) their requests run as
 * ONE device batch on the resident inputs, after which the walks go on.  No DP cell is computed on the host but the
 * X-drop end extensions (sequential by construction: a row's column range follows from where the previous row dropped
 * off; a few thousand cells each).  Needs the exact-model inputs (SpdpScoring.intpen / t53, SpdpProblem.cano5 / cano3 /
 * dinc) whatever SpdpScoring.scalar_engines selects for the DP calls. */
typedef struct SpdpJuxt { int32_t jx, jy, jlen, nid, jscr; } SpdpJuxt;      /* JUXT, src/seq.h:174 */
typedef struct SpdpSeedParams {
    int32_t qck;                     /* algmode.qck: 1 .. 3, depth of the HSP recursion                                   */
    int32_t wl_width[4];             /* setwlprm(level)->width, level 0 .. 3 (src/wln.cc:50; level 3 is 0 in the reference) */
    int32_t elmt, minl;              /* IntronPrm.elmt (shortest exon), IntronPrm.minl                                    */
    int32_t vthr;                    /* PwdB::Vthr: X-drop of the end extensions, creep limit, end margin                 */
    int32_t desert;                  /* alprm2.desert: gaps longer than desert * (4 - level) query residues are not aligned */
    float   maxsp;                   /* alprm.maxsp: DP space limit in MiB / 32 (src/fwd2s1.cc:2501)                      */
    int32_t crs;                     /* algmode.crs                                                                       */
    float   smn4;                    /* getsmn(4): per-residue score of the short terminal stretches (src/simmtx.cc:563)  */
    float   w2;                      /* alprm2.w: weight of the match score in the micro / terminal exon searches         */
    int32_t gc_sig5;                 /* Exinon::gc_sig5 (src/codepot.cc:497)                                              */
    int32_t lcl;                     /* algmode.lcl (bit 16: local, bit 32: LocalC)                                       */
    int32_t codonk1;                 /* PwdB::codonk1 (GapPenalty, src/aln.h:275)                                         */
    int32_t any, both_ori;           /* algmode.any, Exinon::both_ori: Exinon::isCanon's site levels follow from them and
                                        the dinucleotide classes (src/codepot.cc:435-475, src/codepot.h:108-113)          */
    int32_t ip_maxl, ip_mode;        /* IntronPrm.maxl, IntronPrm.mode (protein walk: first_exon_wmm / last_exon_wmm)      */
    const struct SpdpWilipModel* wilip;   /* optional (round 5): with it -- and no SpdpHspSource -- the walks' HSP searches at the
                                        recursion levels are the library's own (spdp_hsp_host.h + spdp_hsp_chain.h); NULL: the
                                        caller's SpdpHspSource answers them                                                */
} SpdpSeedParams;
/* Wilip(seqs, pwd, level) (src/wln.cc:980) for the recursion levels above the one the caller's HSPs come from: the HSP
 * search stays with the caller (the reference's wln.cc in an integration).  units() is called from the walks' threads
 * (concurrently for different queries) with span = {a_left, a_right, b_left, b_right, a_exgl, a_exgr, b_exgl, b_exgr}: the active
 * sub-ranges AND the end flags the walk holds at that moment -- Wlp scores an HSP with an end bonus that depends on
 * a->inex.exgl / exgr (src/wln.cc:378-382, 438-443), so a binding sets both before it constructs Wilip; it returns 0 and
 * a flat record in *flat: n_units, then per unit {num, nid, tlen, llmt, ulmt, scr} followed by num + 1 JUXT records of
 * five ints each (the slot behind the last HSP included, as WLUNIT::jxt has it).  release() hands the record back.
 * Threads: every walk runs on a user-level fiber of the library's worker threads; a walk never changes threads, so
 * thread-local state of the callback (errno, allocator caches, thread_local objects) is safe -- but the fiber's stack is
 * 1 MB, and many walks share one thread: the callback must not hold a lock across calls or rely on per-thread state of
 * a single query. */
typedef struct SpdpHspSource {
    void* user;
    int  (*units)(void* user, int32_t query, int32_t level, const int32_t span[8], const int32_t** flat, int32_t* n_flat);
    void (*release)(void* user, int32_t query, const int32_t* flat);
} SpdpHspSource;
/* alignS_ng(seqs, pwd, gsi, ori = 1) with seeding on.  hsps[i] / n_hsps[i]: b->jxt / b->CdsNo of query i (hsps[i] holds
 * n_hsps[i] + 1 records, the last one a free slot; n_hsps[i] = 0: none, the first level searches itself);
 * lowest_level[i]: b->wllvl.  src may be NULL when qck = 1 and every query brings its HSPs.  Return value as
 * spdp_align_s; out[i].score = gsi->scr. */
int spdp_align_s_seeded(SpdpContext* ctx, const SpdpScoring* sc, const SpdpSeedParams* sp,
                        const SpdpProblem* probs, int n_probs, const SpdpJuxt* const* hsps, const int32_t* n_hsps,
                        const int32_t* lowest_level, const SpdpHspSource* src, SpdpAlignment* out);
/* alignS_ng(seqs, pwd, gsi, ori = 3) with seeding on (src/fwd2s1.cc:2762-2777): the walk on the pair as given (fwd[i]) and on
 * the reverse-complemented query against the other genomic strand (rev[i]: comrev(a) + antiseq(b), that strand's own signals),
 * whose HSP list is the given one turned around as Seq::revjxt does; the reverse result is taken only if it scores strictly
 * higher and then carries A_RevCom (0x10) in its header record; orient[i] = 0 / 1 says which.  The HSP source sees the
 * reverse walk of query i as query n_probs + i. */
int spdp_align_s_seeded_ori3(SpdpContext* ctx, const SpdpScoring* sc, const SpdpSeedParams* sp,
                             const SpdpProblem* fwd, const SpdpProblem* rev, int n_probs,
                             const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level,
                             const SpdpHspSource* src, SpdpAlignment* out, int32_t* orient);
/* counters of the last spdp_align_s_seeded call on this context: [0] device batches, [1] lspS_ng requests,
 * [2] trcbkalignS_ng requests, [3] of those with a cut range, [4] Wilip calls, [5] walks; microseconds: [6] upload of the
 * inputs, [7] the walks' host code (device idle), [8] device batches (walks asleep), [9] handing results back, [10] the call */
int spdp_seeded_stats(const SpdpContext* ctx, int64_t* out, int n);
/* launches this context had to repeat since the last reset: out[0] a cross-CU pass group that did not find all its
 * blocks resident (a cooperative launch beside another kernel), out[1] a tile pipeline of the -A0 / -A1 engines whose
 * predecessor never arrived.  Results are unaffected (the launch runs again without the pipeline); a non-zero count is
 * time lost. */
void spdp_rerun_stats(SpdpContext* ctx, int64_t* out, int reset);

/* stdskl (m_unit 1) / stdskl3 (m_unit 3), src/gaps.cc:140-227: corner list of n path records in any order;
 * out[] needs 2 n + 1 entries, returns the number written.  Host only (no device work). */
int spdp_corner_list(const SpdpSkl* recs, int n, int m_unit, SpdpSkl* out);

/* ---- rescoring: skl_rngS_ng (src/fwd2s1.cc:446) ---------------------------------------------- */
/* The score the CLI prints and the per-exon records come from a walk over the finished corner
 * list, not from the DP engines.  Needs the exact-model inputs (SpdpScoring.intpen / t53,
 * SpdpProblem.dinc).  Cigar / Vulgar / SAM strings are not produced. */
typedef struct SpdpExon {            /* EISCR, src/gsinfo.h:262-284; all scores raw ints */
    int32_t left, right;             /* genomic range of the exon                                  */
    int32_t rleft, rright;           /* query range                                                */
    int32_t mch, mmc, gap, unp;      /* matches, mismatches, gaps, unpaired residues in the exon   */
    int32_t mch5, mmc5, gap5, unp5;  /* ... within jneibr positions after its 5' end               */
    int32_t mch3, mmc3, gap3, unp3;  /* ... within jneibr positions before its 3' end              */
    int32_t phs;
    int32_t escr, iscr;              /* exon score, score of the intron that follows               */
    int32_t sig3, sig5;              /* acceptor signal at the exon start, donor signal at its end */
} SpdpExon;

typedef struct SpdpRescoreParams {
    int32_t codonk1;                 /* PwdB::codonk1: GapPenalty switches to LongGOP/LongGEP above it (src/aln.h:275) */
    int32_t minl;                    /* IntronPrm.minl: shorter deletions are never introns         */
    int32_t jneibr;                  /* alprm2.jneibr: junction neighbourhood (<= 32)               */
    int32_t lsg;                     /* algmode.lsg: splice-aware                                   */
} SpdpRescoreParams;

typedef struct SpdpRescored {
    int32_t   score;                 /* return value of skl_rngS_ng (gsi->scr of the CLI)           */
    int32_t   mch, mmc, gap, unp, val;   /* Gsinfo::fstat                                           */
    int32_t   n_exons;               /* records in exons[], including the reference's end marker    */
    SpdpExon* exons;                 /* Eijnc records; owned by the library until spdp_free_rescored */
} SpdpRescored;

/* aln[i] is an alignment as spdp_align_s returns it (header record + corners); problems without an
 * alignment (n_skl < 3) come back with n_exons = 0. */
int spdp_skl_rng_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpRescoreParams* rp,
                   const SpdpProblem* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out);
void spdp_free_rescored(SpdpRescored* out, int n);

/* The edit records skl_rngS_ng collects for the Cigar / Vulgar / SAM writers (Cigar::push / Vulgar::push / Samfmt,
 * src/gsinfo.h:286-375; pushed at src/fwd2s1.cc:492-689), as raw records in the order the reference writes them -- one
 * format per call, as algmode.nsa selects one there.  CIGAR and SAM: {op, len} in {op, alen}, blen = 0; VULGAR:
 * {op, alen, blen}.  The records are the reference's as they are: SAM carries the running '=' / 'X' stretch once per
 * aligned base, and a gap whose intron candidate lost can come out with a non-positive length (:509-538).  The SAM header
 * fields are those of a forward-strand hit (b->inex.sens == 0). */
#define SPDP_FMT_CIGAR  1
#define SPDP_FMT_VULGAR 2
#define SPDP_FMT_SAM    3
typedef struct SpdpEdit { int32_t op, alen, blen; } SpdpEdit;
typedef struct SpdpEdits {
    int32_t   n;
    SpdpEdit* rec;                   /* owned by the library until spdp_free_edits */
    int32_t   sam_flag, sam_pos, sam_mapq, sam_left, sam_right;   /* Samfmt::flag, pos, mapq, left, right (SAM only) */
} SpdpEdits;
int spdp_skl_edits_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpRescoreParams* rp,
                     const SpdpProblem* probs, int n_probs, const SpdpAlignment* aln, int format, SpdpEdits* out);
void spdp_free_edits(SpdpEdits* out, int n);

/* ---- device groups: every GPU of the node behind one handle -------------- */
/* The reference is one process with worker threads (spaln -t N, src/spaln.cc:1389-1468: a master hands
 * whole queries to the workers).  A group owns one context per listed HIP device (a device may be listed more
 * than once); the calls below shard the problem list by DP cells, one shard per member, run them
 * concurrently -- problems are independent, nothing is exchanged between devices -- and fill scores / out in
 * the caller's order.  Return values as for the single-device calls (the worst of the members). */
typedef struct SpdpGroup SpdpGroup;
SpdpGroup*  spdp_group_create(const int* devices, int n_devices);
void        spdp_group_destroy(SpdpGroup* g);
int         spdp_group_size(const SpdpGroup* g);
const char* spdp_group_last_error(const SpdpGroup* g);
/* member[i] = the group member that ran problem i in the last call (shards are balanced by DP cells -- spdp_cells /
 * spdp_cells_h per problem, longest first to the least loaded member --, not by count); returns the number of problems */
int spdp_group_last_shards(const SpdpGroup* g, int32_t* member, int n);
int spdp_group_homscore_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, int32_t* scores);
int spdp_group_align_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs, SpdpAlignment* out);
struct SpdpScoringH;
struct SpdpProblemH;
int spdp_group_homscore_h(SpdpGroup* g, const struct SpdpScoringH* sc, const struct SpdpProblemH* probs, int n_probs,
                          int32_t* scores);
int spdp_group_align_h(SpdpGroup* g, const struct SpdpScoringH* sc, const struct SpdpProblemH* probs, int n_probs,
                       SpdpAlignment* out);


/* ---- submit / wait ------------------------------------------------------ */
/* Asynchronous form of the batched calls: a worker thread runs the call and owns `ctx` until
 * spdp_wait() returns (one ticket in flight per context; inputs and `out` must stay valid until
 * then).  spdp_wait returns what the synchronous call would have returned and frees the ticket;
 * spdp_poll says whether it would return at once.  Use several contexts -- one per GPU, or more than
 * one per GPU -- to keep batches in flight from a single host thread. */
typedef struct SpdpTicket SpdpTicket;
struct SpdpScoringH;
struct SpdpProblemH;
SpdpTicket* spdp_submit_align_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                                SpdpAlignment* out);
SpdpTicket* spdp_submit_homscore_s(SpdpContext* ctx, const SpdpScoring* sc, const SpdpProblem* probs, int n_probs,
                                   int32_t* scores);
SpdpTicket* spdp_submit_align_h(SpdpContext* ctx, const struct SpdpScoringH* sc, const struct SpdpProblemH* probs,
                                int n_probs, SpdpAlignment* out);
SpdpTicket* spdp_submit_homscore_h(SpdpContext* ctx, const struct SpdpScoringH* sc, const struct SpdpProblemH* probs,
                                   int n_probs, int32_t* scores);
int spdp_poll(const SpdpTicket* t);
int spdp_wait(SpdpTicket* t);

/* ---- resident batches (benchmarking / pipelines) ------------------------ */
/* Uploads a batch once; the run calls below then work on HBM-resident inputs
 * (timed region excludes PCIe).  Returns NULL on failure. */
typedef struct SpdpBatch SpdpBatch;
SpdpBatch* spdp_batch_upload(SpdpContext* ctx, const SpdpScoring* sc,
                             const SpdpProblem* probs, int n_probs);
void       spdp_batch_free(SpdpBatch* bt);
int64_t    spdp_batch_cells(const SpdpBatch* bt);
/* one pass of HomScoreS_ng over the resident batch; scores copied out only if
 * scores != NULL.  kernel_ms (optional) = HIP-event time of the DP kernel on
 * the stream it was launched on. */
int spdp_batch_homscore(SpdpBatch* bt, int32_t* scores, float* kernel_ms);
int spdp_batch_align(SpdpBatch* bt, SpdpAlignment* out, float* kernel_ms, int64_t* kernel_cells);
/* per-kernel figures of the last spdp_batch_align call:
 *  [0] UDH sweep ms (HIP events)  [1] UDH cells  [2] UDH problems
 *  [3] forward sweep ms           [4] forward cells  [5] forward problems (direct + slabs)
 *  [6] UDH rounds                 [7] traceback bytes written */
#define SPDP_N_STATS 8
int spdp_batch_stats(const SpdpBatch* bt, double* out, int n);


/* ======================================================================== */
/* protein x genome (aa x 3-frame "tron" codes): the Fwd2h1 `_wip` path      */
/*   VTYPE HomScoreH_ng(const Seq* seqs[], const PwdB* pwd)   src/fwd2h1.cc:3288 */
/*   SKL*  alignH_ng(const Seq* seqs[], const PwdB*, Gsinfo*) src/fwd2h1.cc:3310 */
/*   SimdAln2h1(seqs, pwd, wdw, spjcs, cip, mode = 1)         src/fwd2h1_simd.h:198 */
/*     forwardH1_wip(Mfile*)                                  src/fwd2h1_wip_simd.h:50 */
/* DP row m in (a_left, a_right] is residue a[m-1]; DP column n is the       */
/* nucleotide position, the codon ending at n is b[n-2] (tron code, after    */
/* Seq::nuc2tron); diagonals are r = n - 3m.                                 */
/* ======================================================================== */

/* PwdB / IntronPenalty / Simmtx subset read by SimdAln2h1 (src/aln.h:235-308) */
typedef struct SpdpScoringH {
    int32_t mtx_rows, mtx_cols;      /* Simmtx::rows (aa, 23), Simmtx::dim (tron, 26)       */
    int32_t mtx[32 * 32];            /* row-major mtx[aa * mtx_cols + tron]                 */
    int32_t gop, gep;                /* PwdB::BasicGOP, BasicGEP (per codon)                */
    int32_t lgep, codonk1;           /* GapExtPen3(i) = i > codonk1 ? LongGEP : BasicGEP    */
    int32_t gapw1, gapw2, gapw3;     /* PwdB::GapW1 / GapW2 (frame shifts), GapW3 (codon gap open) */
    int32_t spj;                     /* b->inex.intr                                        */
    int32_t llmt, ipen;              /* IntronPrm.llmt, IntronPenalty::Penalty()            */
    int32_t nquant;
    int32_t qm_len[SPDP_MAX_QUANT];
    int32_t qm_pen[SPDP_MAX_QUANT];
    int32_t local;                   /* algmode.lcl & 16                                    */
    int32_t term_codon;              /* algmode.lcl & 2: termination codon as a right end (fwd2h1_simd.h:720) */
    int32_t sh;                      /* alprm.sh (codons), stripe31()                       */
    int32_t max_vmf_space;           /* MaxVmfSpace                                         */
    int32_t ubh;                     /* alprm.ubh                                           */
    int32_t ref_nelem;               /* 16                                                  */
    /* ---- rescoring (spdp_skl_rng_h) and the scalar engine (forwardH_ng) only; zero / NULL otherwise */
    int32_t lgop;                    /* PwdB::LongGOP                                       */
    int32_t gape1, gape2, extragop;  /* PwdB::GapE1, GapE2, ExtraGOP (frame-shift terms)    */
    int32_t diffu, k1;               /* PwdB::diffu, alprm.k1 (UnpPenalty3, src/aln.h:290)  */
    const int16_t* intpen;           /* IntronPenalty::Penalty(len), len in [0, intpen_len) */
    int32_t intpen_len;
    int16_t t53[256];                /* sig53(m, n, IE53) - sig3[n] by 16 * dinc5[m] + dinc3[n] */
    int32_t minl;                    /* IntronPrm.minl (scalar engine: shortest intron; 0 = llmt)   */
    int32_t scalar_engines;          /* 0: the `_wip` engines (-A2 / -A3); 1: algmode.alg == 0 (-A0):
                                        spdp_align_h / spdp_homscore_h run forwardH_ng / hirschbergH_ng;
                                        2: algmode.alg == 1 (-A1): spdp_align_h runs forwardH1 / hirschbergH1
                                        (src/fwd2h1_simd.h:820, 1100) with the -A0 ladder geometry; HomScoreH_ng
                                        above 7 rows stops with SIGSEGV in the reference under -A1 (forwardH1
                                        without a Vmf) and comes back here as "not computed" (return value 1) */
    int32_t recursive;               /* algmode.alg & 4: lspH_ng always takes the recursive branch    */
    int32_t noll;                    /* PwdB::Noll: 2 (also 0) affine gaps; 3 double affine gaps (-yl3): the -A0 engines
                                        (forwardH_ng / hirschbergH_ng, src/fwd2h1.cc:297, 1088) keep a second vertical and a
                                        second insertion state priced with GapW3L = lgop + lgep / lgep; other engines refuse */
    const struct SpdpSignalModelH* sigmodel;  /* optional: with it, problems whose signal arrays are all NULL get them (and
                                        dinc) computed on the device from the tron codes (spdp_signals_h.hip) */
} SpdpScoringH;

/* the model behind the SGPT6 arrays (Exinon::intron53_p, src/codepot.cc:524-611): position weight matrices of the splice
 * sites (order 2), the start-codon context (order <= 1) and the stop-codon context (order 2) -- PatMat, src/utilseq.h:64-88,
 * rows = 4 + 16 (+ 64) terms per column --, the 5th-order three-phase coding potential (ExinPot, utilseq.h:90-168:
 * 3 * ndata floats), the scale factors and the per-dinucleotide terms.  A model with a branch-point matrix
 * (EijPat::patternB) or an intron potential (alprm2.Z > 0) is not supported. */
typedef struct SpdpPatMat {
    int32_t rows, cols, offset, order;          /* rows = 0: matrix absent                                  */
    float   tonic, min_elem;
    const float* mtx;                           /* cols * rows                                              */
} SpdpPatMat;
typedef struct SpdpSignalModelH {
    SpdpPatMat pm5, pm3, pmI, pmT;              /* EijPat::pattern5, pattern3, patternI, patternT           */
    int32_t pot_ndata;  const float* pot;       /* PwdB::codepot: size(), begin() (0 / NULL: none)          */
    float   fE, fT, fO;                         /* alprm2.z * fact, alprm2.bti * fact, -alprm2.o * fact     */
    float   fS, fs;                             /* Exinon::fS, fS * alprm2.sss                              */
    float   tonic5, tonic3;                     /* EijPat::tonic5, tonic3                                   */
    int16_t tab5[16], tab3[16];                 /* sig53tab[0][dinc5], sig53tab[1][dinc3]                   */
    int32_t any;                                /* algmode.any                                              */
    int32_t dvsp;                               /* PwdB::DvsP != 3: start / stop contexts are scored        */
    int32_t trm, trm2;                          /* the two termination tron codes (TRM, TRM2)               */
    /* branch-point term of the acceptor signal (-yB; Exinon::intron53_p, src/codepot.cc:543, 586-597): an acceptor
     * earns fB x the score of the last branch site stronger than tonicB at most maxb3d + 1 positions upstream.
     * pmB.rows = 0 (a zeroed tail): off, as in the reference's default and all its species tables */
    SpdpPatMat pmB;                             /* EijPat::patternB                                         */
    float   fB, tonicB;                         /* bpprm.factor * fact, EijPat::tonicB                      */
    int32_t maxb3d;                             /* bpprm.maxb3d                                             */
} SpdpSignalModelH;
/* the SGPT6 arrays of one tron window on the device: b_len + 3 entries each, computed for [left, right) (any output may be
 * NULL); b holds b_len + 1 codes */
int spdp_splice_signals_h(SpdpContext* ctx, const SpdpSignalModelH* model, const uint8_t* b, int32_t b_len,
                          int32_t left, int32_t right, int16_t* sig5, int16_t* sig3, int16_t* sigS, int16_t* sigT,
                          int16_t* sigE, int8_t* phs5, int8_t* phs3, uint8_t* dinc);

typedef struct SpdpProblemH {
    const uint8_t* a;  int32_t a_len;      /* amino-acid codes a[0 .. a_len)                      */
    const uint8_t* b;  int32_t b_len;      /* tron codes b[0 .. b_len]: b_len + 1 readable entries
                                              (the engine reads the terminator, fwd2h1_wip_simd.h:188) */
    /* SGPT6 fields per genomic position (src/codepot.h:34-43), index 0 .. b_len + 2; the reference
     * holds them for [exin_left - 1, exin_right + 1] (Exinon ctor, src/codepot.cc:357-366) */
    const int16_t* sig5;  const int16_t* sig3;
    const int16_t* sigS;  const int16_t* sigT;  const int16_t* sigE;
    const int8_t*  phs5;  const int8_t*  phs3;
    int32_t exin_left, exin_right;         /* Seq::left / right when the Exinon was built: good(n)
                                              <=> exin_left - 1 <= n < exin_right (codepot.h:120-123) */
    int32_t a_left, a_right, b_left, b_right;
    uint8_t a_exgl, a_exgr, b_exgl, b_exgr;
    uint8_t a_pad;                         /* the code the reference's Seq holds behind the query, a->at(a->len)[0]: the
                                              -A0 / -A1 engines price a phase -1 acceptor of the last row with the profile
                                              of the "next" residue (qprof[1], src/fwd2h1.cc:368-370, 489).  Seq::exg_seq
                                              (src/seq.cc:926-938) leaves nil_code = 0 there when the query's end gaps are
                                              free or under the default tgapf, so 0 is the usual value */
    uint8_t reserved_[3];
    const uint8_t* dinc;                   /* rescoring only: INT53::dinc5 << 4 | dinc3 per position, or NULL */
    const int32_t* cip;                    /* optional, the -A0 / -A1 engines only: Cip_score::cip_score(c) for coding
                                              position c = 0 .. 3 a_len + 1 (an intron accepted in row m at phase phs
                                              earns cip[3 m - phs], src/fwd2h1.cc:352-354, 483; fwd2h1_simd.h:407);
                                              NULL = none */
} SpdpProblemH;

/* Aln2h1::lspH_ng (src/fwd2h1.cc:2134-2230) for a caller that keeps the record file itself -- the protein side of the
 * seeded path (interpolateH, src/fwd2h1.cc:3106-3120): the ladder below spdp_align_h, out[i].skl = the Mfile records as
 * written (n_skl of them, no header, any order: globalH_ng's stdskl3 sorts), flags as spdp_align_h. */
int spdp_lsp_h(SpdpContext* ctx, const struct SpdpScoringH* sc, const struct SpdpProblemH* probs, int n_probs, SpdpAlignment* out);
/* alignH_ng with seeding on (algmode.qck = 1 .. 3; Aln2h1::globalH_ng -> seededH_ng -> interpolateH, src/fwd2h1.cc:3267-3286,
 * 3177-3265, 3023-3131): arguments as spdp_align_s_seeded (jy of an HSP is a nucleotide position of the tron sequence).
 * The problems need all seven signal arrays and dinc on the host, sc->intpen / t53.  Return value and out[] as
 * spdp_align_h; a walk with a DP call on which the reference itself is undefined is reported as not served (return 1, no
 * alignment). */
/* ---- the HSP search itself (Wilip, src/wln.cc; SURVEY 8 row f4, second slice) ------------------------------------------------
 * What the reference's Wilip reads besides the two sequences: per recursion level the word parameters of setwlprm(level)
 * (src/wln.cc:53-128: reduced alphabet, tuple size, bit pattern, gains, cut-offs) and, shared, the HSP-search substitution
 * matrix getSimmtx(WlnPamNo), the end bonus, and a few switches.  With a model a seeded call needs no SpdpHspSource: the
 * library searches the sub-ranges itself (spdp_hsp_host.h), and spdp_wilip answers one request the way Wilip::Wilip does. */
typedef struct SpdpWilipLevel {
    int32_t elem, tpl, mask, width, gain, gain1, thr, xdrp, cutoff, vthr;     /* WLPRM, src/wln.h:35-49                    */
    int32_t bitpat_len;              /* 0: contiguous words of `width`; else the spaced pattern, bitpat[i] = 1 / 0      */
    uint8_t bitpat[32];
    uint8_t convtab[32];             /* WLPRM::ConvTab[code]: class of the reduced alphabet, >= elem: not part of a word */
} SpdpWilipLevel;
typedef struct SpdpWilipModel {
    SpdpWilipLevel level[3];
    int32_t mtx_rows, mtx_cols;      /* getSimmtx(WlnPamNo): query code x genomic code (nucleotide or tron)              */
    int32_t mtx[32 * 32];
    int32_t dvsp;                    /* PwdB::DvsP: 0 nucleotide query, 1 protein query x tron codes                      */
    int32_t end_bonus;               /* Wlprms::EndBonus = (VTYPE) AvTrc() / 2 (src/wln.cc:146)                           */
    int32_t crs, lsg, mlt;           /* algmode.crs, lsg, mlt                                                             */
    int32_t hard_minl, hard_maxl, minl, maxl, llmt;                           /* IntronPrm                                */
    int32_t avrsig;                  /* IntronPenalty::AvrSig: PenaltyPlus(n) = Penalty(n) + AvrSig (src/codepot.h:248)   */
    int32_t shortquery;              /* 50 (src/wln.h:33): a level -1 search on a shorter query scales its cut-offs       */
    int32_t min_hit;                 /* 3 (src/wln.cc:34)                                                                 */
    int32_t met, ser, ser2;          /* residue codes MET, SER, SER2 (src/cmn.h:117)                                      */
} SpdpWilipModel;

/* Wilip::Wilip(seqs, pwd, level) on one request (src/wln.cc:980): the HSPs of query range [a_left, a_right) against genomic
 * range [b_left, b_right), chained into units, best unit first -- the flat record SpdpHspSource::units hands over (n_units,
 * per unit {num, nid, tlen, llmt, ulmt, scr} + (num + 1) x {jx, jy, jlen, nid, jscr}; the nid of a unit's closing record is 0
 * where the reference leaves it unset).  Exactly one of p / ph is given (nucleotide / protein query; the protein form reads
 * sigS / sigE / sigT where present) with the scoring that belongs to it (gap penalties of the chaining, intpen).  level:
 * 0 .. 2 as the walks and geneorient() ask, -1 as FindHsp does (cut-offs scaled for queries below shortquery residues).
 * exg = {a->inex.exgl, a->inex.exgr} at the call.  *flat is malloc'ed (free() it); returns the number of ints, < 0 on error.
 * Host code: no device involved. */
int spdp_wilip(const struct SpdpWilipModel* model, const SpdpProblem* p, const SpdpScoring* sc,
               const struct SpdpProblemH* ph, const struct SpdpScoringH* sch,
               int32_t level, const int32_t span[4], const int32_t exg[2], int32_t** flat);

typedef struct SpdpPhaseMark {
    int32_t n;                       /* position of the tron sequence                                              */
    int8_t  side;                    /* 5: SGPT6::phs5, 3: SGPT6::phs3                                              */
    int8_t  value;                   /* the phase the walk wrote there                                              */
    int16_t reserved;
} SpdpPhaseMark;
/* A side effect of the reference's walk that its caller depends on: where the walk closes a gap between two HSPs with an
 * intron of its own choice (Aln2h1::indelfreespjH, src/fwd2h1.cc:2508-2517) it WRITES the junction's phase into the Exinon
 * (SGPT6::phs5 at the donor, phs3 at the acceptor), and skl_rngH_ng -- which spalign2 runs next on the same objects --
 * reads the phase of every junction from those fields (src/fwd2h1.cc:824-825).  The library does not write into the
 * caller's arrays; the marks of the last spdp_align_h_seeded call on this context, per query and in the order they were
 * made, are handed out here for the binding to apply (b->exin->score_p(n)->phs5 / phs3 = value) before it rescores.
 * *marks stays valid until the next seeded call on the context.  Returns the number of marks of query q (0: none). */
int spdp_seeded_phase_marks(const SpdpContext* ctx, int q, const SpdpPhaseMark** marks);
int spdp_align_h_seeded(SpdpContext* ctx, const struct SpdpScoringH* sc, const SpdpSeedParams* sp,
                        const struct SpdpProblemH* probs, int n_probs, const SpdpJuxt* const* hsps, const int32_t* n_hsps,
                        const int32_t* lowest_level, const SpdpHspSource* src, SpdpAlignment* out);
/* the protein entry of the collector: spdp_collector_create_h owns `ctx` and one SpdpScoringH (its intpen table is copied; a
 * signal model, if any, must outlive the collector); spdp_collector_align_h is spdp_collector_align_s for one SpdpProblemH
 * (raw_records = 1: spdp_lsp_h results, 0: spdp_align_h results).  Destroy, error and stats calls are shared. */
SpdpCollector* spdp_collector_create_h(SpdpContext* ctx, const struct SpdpScoringH* sc, int max_batch, int max_wait_us, int raw_records);
int spdp_collector_align_h(SpdpCollector* c, const struct SpdpProblemH* p, SpdpAlignment* out);

/* stripe31(seqs, &wdw, sh), src/aln2.cc:178-198 */
void spdp_stripe31(const SpdpProblemH* p, int sh, SpdpWindow* wdw);
/* (aa, nt) cells inside the band: rows m, columns max(b_left, lw + 3m) < n <= min(b_right, up + 3m) */
int64_t spdp_cells_h(const SpdpProblemH* p, const SpdpWindow* wdw);

/* forwardH1_wip(mfd): returned score (the reference returns its `nevsel` here unless a local
 * right end was tracked: fhlastH1 never sets maxh.val, fwd2h1_simd.h:692-785) + raw Mfile
 * records end -> start.  n_skl = -1 flags the reference's fatal "Unexpected dir"; n_skl = -2 says
 * its traceback would start outside its bitmap (a winning genomic end gap moves the end cell
 * beyond b_right, fwd2h1_simd.h:756-766, 780: an out-of-bounds read there, undefined result);
 * n_skl = -4: the record list of this query outgrew its slot (4096 records per traceback call): this query only comes
 * back without records, the rest of the batch is unaffected. */
int spdp_wip_forward_h(SpdpContext* ctx, const SpdpScoringH* sc,
                       const SpdpProblemH* probs, int n_probs, SpdpAlignment* out);
/* hirschbergH1_wip with n_im intermediate rows (src/fwd2h1_wip_simd.h:338): cpos[i] points at
 * (n_im + 1) * 10 ints, ranges[i*4..] receives the written-back a_left, a_right, b_left, b_right.
 * With SpdpScoringH.local the local-ends form runs (its own kernel; a_right may come back one row beyond the
 * query when the path ends on the last row, as in the reference, :652-653). */
int spdp_wip_udh_h(SpdpContext* ctx, const SpdpScoringH* sc,
                   const SpdpProblemH* probs, int n_probs, int n_im,
                   int32_t* scores, int32_t* cpos, int32_t* ranges);
/* HomScoreH_ng for -A2/-A3: stripe31() then forwardH1_wip() */
int spdp_homscore_h(SpdpContext* ctx, const SpdpScoringH* sc,
                    const SpdpProblemH* probs, int n_probs, int32_t* scores);
/* alignH_ng with seeding off (-Q0/-Q4): stripe31 -> lspH_ng decision ladder -> forwardH1_wip, or
 * hirschbergH1_wip + per-slab forwardH1_wip (mimd_postwork / rcsv_postwork) -> stdskl3.
 * Sub-problems below 8 query rows run the scalar forwardH_ng (needs the scalar engine's inputs, see
 * spdp_scalar_forward_h).  A window without width (up == lw) takes diagonalH_ng.  Return value 1: for some
 * problem the links lead outside the sequences (undefined in the reference) or the scalar inputs are
 * missing; those come back with n_skl = 0, score NEVSEL. */
int spdp_align_h(SpdpContext* ctx, const SpdpScoringH* sc,
                 const SpdpProblemH* probs, int n_probs, SpdpAlignment* out);

/* Aln2h1::forwardH_ng (src/fwd2h1.cc:294-617, with initH_ng / lastH_ng): the scalar -A0 engine -- int32,
 * exact intron-length penalty, top-4 donor list per row and codon phase -- on the stripe31() band.
 * traceback != 0: the Mfile records trcbkalignH_ng (src/fwd2h1.cc:1997-2041) writes, end -> start (Vmf
 * traceback + boundary fix-up); traceback == 0: score only, as HomScoreH_ng runs it under -A0 (:3297).
 * Needs SpdpScoringH.intpen / t53 / gape1 / gape2 / extragop / minl and SpdpProblemH.dinc.  One GPU
 * thread per problem: meant for the sub-problems below 8 query rows that the -A2 / -A3 dispatch hands
 * to this engine (spdp_align_h and spdp_homscore_h do that themselves), correct at any size.
 * With SpdpScoringH.scalar_engines = 2 this entry runs SimdAln2h1::forwardH1 instead (the -A1 engine, modes
 * 3 / 5, src/fwd2h1_simd.h:820-1096; 16 lanes per problem; traceback form only); n_skl = -3 marks a run
 * whose record pointers leave the int16 lane the reference keeps them in under mode 3 (undefined there). */
int spdp_scalar_forward_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                          int traceback, SpdpAlignment* out);

/* Aln2h1::hirschbergH_ng (src/fwd2h1.cc:1085-1520), the scalar linear-space engine, with n_im
 * intermediate rows imd_intvl rows apart (Aln2h1::imd_intvl as lspH_ng sets it): cpos, ranges as for
 * spdp_wip_udh_h; cpos[..][8], [9] carry the diagonal bounds of each slab (its window under -A0);
 * entries the reference leaves uninitialised read end_of_ulk.  flags[i] = -3: the reference indexes
 * outside its arrays on this input (undefined there), 0 otherwise.
 * With SpdpScoringH.scalar_engines = 2 this entry runs SimdAln2h1::hirschbergH1 (src/fwd2h1_simd.h:1100-1470,
 * non-local ends): cpos / ranges as for spdp_wip_udh_h, imd_intvl unused, flags 0. */
int spdp_scalar_udh_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n_probs,
                      int n_im, int imd_intvl, int32_t* scores, int32_t* cpos, int32_t* ranges, int32_t* flags);

/* skl_rngH_ng (src/fwd2h1.cc:635): the same for protein alignments (codon-split introns, frame
 * shifts, start / stop signals).  Needs SpdpScoringH.intpen / t53 / lgop ... and SpdpProblemH.dinc.
 * The standard genetic code is assumed for codons split by an intron, as the reference's static
 * spj_tron_tab does (src/codepot.h:130). */
typedef struct SpdpRescoreParamsH {
    int32_t minl;                    /* IntronPrm.minl                                              */
    int32_t jneibr;                  /* alprm2.jneibr (<= 32)                                       */
    int32_t lcl;                     /* algmode.lcl: which ends may take start / stop / splice signals */
    int32_t sup_tcodon;              /* OutPrm.supTcodon                                            */
} SpdpRescoreParamsH;
int spdp_skl_rng_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp,
                   const SpdpProblemH* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out);

/* resident batch (benchmarking / pipelines); one live batch of this kind per context */
/* ---- result records ----------------------------------------------------------------------------
 * What Gsinfo::ExonForm (src/sqpr.cc:820-996) makes of an alignment's EISCR records: the ExonRecord / GeneRecord
 * payload of the -O12 `.erd` / `.grd` files (same layout as src/seq.h:1212-1255, so a caller can fwrite them as they
 * are) and the text lines of -O4.  Host only.  `eij` = SpdpRescored::exons of spdp_skl_rng_s / _h (end marker
 * included or not). */
typedef struct SpdpSiteMap { int32_t site0, step; } SpdpSiteMap;    /* Seq::SiteNo(n) = site0 + n * step (step -1 on the reverse strand) */
typedef struct SpdpExonRecord {      /* ExonRecord */
    int32_t Elen, Nmmc, Nunp, Rleft, Rright, Gleft, Gright, Ilen, Bmmc, Bunp, miss, phase;
    float   Pmatch, Escore, Iscore, Sig3, Sig5;
    char    Iends[4];
} SpdpExonRecord;
typedef struct SpdpGeneRecord {      /* GeneRecord */
    int32_t  Cid, Gstart, Gend;
    uint32_t Nrecord, nexn;
    int32_t  Rid, Rlen, Rstart, Rend, mmc, unp, bmmc, bunp, ng;
    float    Gscore, Pmatch, Pcover;
    int16_t  Csense, Rsense;
} SpdpGeneRecord;
typedef struct SpdpExonFormIn {
    const SpdpExon* eij;  int32_t n_eij;
    int32_t scr;                     /* Gsinfo::scr (the alignment's score, raw) */
    const uint8_t* gene_codes;       /* residue codes of the genomic sequence (nucleotide or tron), whole sequence */
    int32_t gene_is_tron;            /* 1: tron genome (protein queries): intron ends print through `ncodon` */
    int32_t qry_is_protein;
    int32_t q_left, q_right, q_len, q_many, q_sens;      /* qry->left, right, len, many, inex.sens */
    SpdpSiteMap gmap, qmap;
    float   scale;                   /* alprm.scale * (gene->exin->fact, when set) */
    float   aln_scale;               /* alprm.scale */
    int32_t hsp_len;                 /* sum of gene->jxt[].jlen ("HC:" of the @ line), 0 without seeding */
    int32_t gene_id, qry_id;         /* GeneRecord::Cid (gene->did), Rid (running query index of the .qrd file) */
    int32_t first_exon_record;       /* GeneRecord::Nrecord: exon records written before this gene */
} SpdpExonFormIn;
/* returns the number of exon records written (<= cap), -1 on bad input */
int spdp_exon_form(const SpdpExonFormIn* in, SpdpExonRecord* exons, int cap, SpdpGeneRecord* gene);
/* the -O4 lines (header != 0: with the column header the reference prints once per run); returns the length written,
 * or -(needed + 1) when buf is too small */
int spdp_exon_form_text(const SpdpExonFormIn* in, const char* qname, const char* gname, int header, char* buf, int cap);

/* the -O12 record files sortgrcd reads (src/sqpr.cc:853-885, 960-985): <prefix>.grd (GeneRecord[]), <prefix>.erd
 * (ExonRecord[]), <prefix>.qrd (database name, then one query name per gene, NUL-terminated).  spdp_o12_write numbers the
 * records itself (GeneRecord::Nrecord, ::Rid). */
typedef struct SpdpO12 SpdpO12;
SpdpO12* spdp_o12_open(const char* prefix, const char* db_name);
int spdp_o12_write(SpdpO12* h, const SpdpExonRecord* exons, int n_exons, const SpdpGeneRecord* gene, const char* qname);
int spdp_o12_close(SpdpO12* h);

/* the protein-side edit records (skl_rngH_ng, src/fwd2h1.cc:663-667, 695-924): SPDP_FMT_CIGAR or SPDP_FMT_VULGAR (the
 * latter after Vulgar::postproc, as the reference keeps them; no SAM form exists there) */
int spdp_skl_edits_h(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp,
                     const SpdpProblemH* probs, int n_probs, const SpdpAlignment* aln, int format, SpdpEdits* out);

typedef struct SpdpBatchH SpdpBatchH;
SpdpBatchH* spdp_batch_upload_h(SpdpContext* ctx, const SpdpScoringH* sc,
                                const SpdpProblemH* probs, int n_probs);
void        spdp_batch_free_h(SpdpBatchH* bt);
int64_t     spdp_batch_cells_h(const SpdpBatchH* bt);
/* one pass of alignH_ng over the resident batch; out may be NULL.  kernel_ms = HIP-event time of the
 * DP sweep kernel on the stream it was launched on. */
int spdp_batch_align_h(SpdpBatchH* bt, SpdpAlignment* out, float* kernel_ms, int64_t* kernel_cells);

/* ---- block search: the vote of SrchBlk::findblock (first slice) -----------------------------------------
 * Replaces, for cDNA / EST queries against a genome index, what the reference computes between the TestOutput calls of
 * SrchBlk::findblock (src/blksrc.cc:2971-3087; Qwords :2819-2969, Bhit4 :2763-2817, Randbs :2047-2069) and the list of
 * candidate block pairs TestOutput builds for FindHsp (extract_to_work :2547-2603, TestOutput :2620-2672).  FindHsp itself
 * (Wilip on the candidate region, :2346) and the index file reader (:1697-1925) stay with the caller: an integration fills
 * SpdpBlkIndexDesc from the SrchBlk object its own ReadBlkInfo produced.  One query per device lane; the
 * index is resident in HBM.
 *
 * SpdpBlkIndexDesc = BlkWcPrm + ContBlk (src/blksrc.h:186-233) + the members / file statics of blksrc.cc the vote reads. */
typedef struct SpdpBlkIndexDesc {
    int32_t nalpha, tabsize, nshift, nbitpat;        /* wcp.Nalpha, TabSize, Nshift, Nbitpat */
    int32_t convts, n_chr, maxblk;                   /* pbwc->ConvTS, ChrNo, MaxBlk */
    int32_t kk, drna, maxmmc, nseg;                  /* SrchBlk::kk, DRNA, maxmmc, nseg */
    int32_t minsigpr, ncand, nascr;                  /* MinSigpr, Ncand, Nascr (src/blksrc.cc:52-69, 2220) */
    int32_t maxblock, extblock, extblockl, shortquery;   /* MaxBlock, ExtBlock, ExtBlockL (:2213-2216), shortquery (src/wln.h:34, set :2219) */
    int32_t hh_size, hh_step;                        /* geometry of Dhash<INT,int>(2 * MaxBlk, 0): size1, size2 (src/clib.h:257-267); */
    int32_t hb_size, hb_step, ha_size, ha_step;      /*   of the position hashes of the Ncand / Nascr queues; 0 = derive (hh_step etc. = 8) */
    int32_t gdb;                                     /* Randbs: log (genomic database) or sqrt transform beyond its table */
    int32_t blklen;                                  /* wcp.blklen: block b of a chromosome whose first block is z covers [(b - z) * blklen, ..) (SrchBlk::setgnmrng); not read by the vote */
    float   rbscoef, rbscons;                        /* Randbs::RbsCoef, RbsCons */
    double  bclw, bcup, bcce;                        /* Block2Chr */
    double  cfact;                                   /* app_c = Nbitpat ^ cfact (:2826) */
    const uint8_t*  convtab;                         /* ConvTab[convts]: residue code -> reduced alphabet */
    const uint16_t* nblk;                            /* pbwc->Nblk[tabsize] */
    const int16_t*  wscr;                            /* pbwc->wscr[tabsize] */
    const int32_t*  blkp;                            /* per word: offset of its posting list in blkb + 1, 0 = none (pbwc->blkp) */
    const uint32_t* blkb;  int64_t n_words;          /* pbwc->blkb[WordNo] */
    const int32_t*  rscrtab;                         /* Randbs::rscrtab[128] */
    const int32_t*  chr;                             /* per chromosome 0 .. n_chr: (ChrID.spos, first block = chrblk(c)) */
    const int32_t*  bitpat; int32_t n_bitpat;        /* per pattern k < kk: weight, width, wshift, exam[2 * weight] (Bitpat, src/bitpat.h:59-72) */
} SpdpBlkIndexDesc;
typedef struct SpdpBlkIndex SpdpBlkIndex;
SpdpBlkIndex* spdp_blk_index_create(SpdpContext* ctx, const SpdpBlkIndexDesc* desc);     /* uploads; NULL on error (spdp_last_error) */
void          spdp_blk_index_destroy(SpdpBlkIndex* ix);
/* The reference's own index file (<db>.bkn of `spaln -W -KD`, or <db>.bkp of `spaln -W -KP`: the amino-acid words of the
 * translated genome, for protein queries; format version 26) read on the host, with the search
 * parameters SrchBlk::initialize derives on opening one (src/blksrc.cc:1697-1858, 2179-2227).  What the file does not hold
 * comes from SpdpBlkSearchOpts (defaults = the reference's: spdp_blk_search_opts_default); ExtBlock follows from the species'
 * intron length distribution (max_intron_len(0.996), src/codepot.cc:648), which the caller supplies as max_intron_len or
 * directly as ext_block. */
typedef struct SpdpBlkSearchOpts {
    int32_t max_out;                 /* OutPrm.MaxOut (Ncand = max_out + 10) */
    int32_t max_mmc, min_sigpr, nascr;   /* MaxMmc (-Xm), MinSigpr (-Xn), Nascr (-Xd) */
    int32_t ext_block, max_intron_len;   /* ExtBlock given (-XE), or max_intron_len(ild_up_quantile) to derive it from */
    int32_t local;                   /* algmode.lcl & 16: no limit on recurrences */
    int32_t genomic_db;              /* Randbs transform: 1 log (a genomic database), 0 sqrt */
    float   rbs_fact, rbs_base;      /* RbsFact (-Xf), RbsBase (-Xe) */
    double  cfact;                   /* -Xc */
} SpdpBlkSearchOpts;
typedef struct SpdpBlkIndexHost SpdpBlkIndexHost;
void spdp_blk_search_opts_default(SpdpBlkSearchOpts* o);
SpdpBlkIndexHost* spdp_blk_index_read(const char* path, const SpdpBlkSearchOpts* opts, char* err, int err_cap);   /* NULL + message on error */
const SpdpBlkIndexDesc* spdp_blk_index_host_desc(const SpdpBlkIndexHost* h);      /* valid until spdp_blk_index_host_free */
void spdp_blk_index_host_free(SpdpBlkIndexHost* h);

/* n queries: codes of query i = codes[offs[i] .. offs[i + 1]), searched range [left[i], right[i]) (Seq::left / right);
 * stop_at[i] (NULL = 0 for all) = which TestOutput call of findblock to stop at -- the earlier ones are taken to have
 * answered "nothing found, go on" (a caller whose FindHsp rejects every pair of call k asks again with k + 1; most queries
 * end at call 0).  out: n records of out_cap ints each: [0] ints used, [1] TestOutput calls met, [2] flags
 * (SPDP_BLK_REACHED: the asked call was reached and the record follows; SPDP_BLK_CUT: out_cap too small; SPDP_BLK_TABLE:
 * one of the reference-sized hash tables ran full, where the reference would grow it), then sign[4] mmct[4] nhit[4] maxs[4]
 * testword[4] (Bhit4); per direction n, (block, score) x n: the significant blocks (prqueue_b, heap order); n_pairs and
 * nine ints per candidate pair, best first (BPAIR: bscr chr lb rb ub db zl zr rvs); n_runs and (block | direction << 28,
 * score) x n_runs: the run scores (Bhit4::bscr) FindHsp can look at -- within 4 x max(ExtBlockL, ExtBlock) blocks of a reported pair (its reach over three moves of an end), inside its
 * chromosome, on its strand -- unordered.  kernel_ms (may be NULL): HIP-event time. */
#define SPDP_BLK_REACHED 1
#define SPDP_BLK_CUT     2
#define SPDP_BLK_TABLE   4
#define SPDP_BLK_FORCED  8         /* the reached call is the one findblock makes behind its scan: TestOutput(1) */
int spdp_blk_vote(SpdpContext* ctx, const SpdpBlkIndex* ix, const uint8_t* codes, const int64_t* offs,
                  const int32_t* left, const int32_t* right, const int32_t* stop_at, int32_t n,
                  int32_t* out, int32_t out_cap, float* kernel_ms);
/* the same with queries and records resident on the device (d_* are device pointers; the bench's timed region) */
int spdp_blk_vote_resident(SpdpContext* ctx, const SpdpBlkIndex* ix, const uint8_t* d_codes, const int64_t* d_offs,
                           const int32_t* d_left, const int32_t* d_right, const int32_t* d_stop_at, int32_t n,
                           int32_t* d_out, int32_t out_cap, float* kernel_ms);

/* ---- block search, third slice (round 5): from the vote to candidate loci ------------------------------------------------
 * What SrchBlk::findblock returns to its caller: TestOutput's second half (src/blksrc.cc:2677-2692: the candidate block pairs
 * against the random expectation of their mismatch counts) and FindHsp (:2346-2545: the region of a pair cut from the genome,
 * the HSP search on it -- spdp_wilip, level -1 --, the units that hold against critjscr, the pair's ends moved towards what the
 * HSPs leave uncovered, the candidate loci with their overlap / order / pruning rules).  spdp_blk_find runs the vote on the
 * device for all queries, call after call (stop_at = 0, 1, ..: a query whose pairs all fail asks again, as findblock does),
 * and this part on the host's threads in between.  Nucleotide queries.  A locus is what the aligner is then given: the region
 * [base, base + len) of chromosome chr (reverse-complemented when rvs), its range [left, right) and the HSPs inside it --
 * exactly the window + SpdpJuxt list spdp_align_s_seeded takes. */
typedef struct SpdpBlkFindParams {
    int32_t vthr;                    /* alprm.scale * 2 * alprm.thr (src/blksrc.cc:2210)                                   */
    float   drop_rate;               /* 1 unless -Xr (:56, 115)                                                            */
    int32_t max_out, max_out2;       /* OutPrm.MaxOut, MaxOut2                                                             */
    int32_t min_agap;                /* SrchBlk::min_agap                                                                  */
    int32_t phase1t;                 /* Randbs::Phase1T = (int) (RbsBias * avr) (:2059)                                    */
    int32_t a_exgl, a_exgr;          /* query->inex.exgl / exgr while it is searched (the HSP search's end bonus reads them) */
} SpdpBlkFindParams;
typedef struct SpdpGenome {          /* residue codes of the chromosomes in host memory, in the index's order              */
    const uint8_t* codes; const int64_t* chr_off; int32_t n_chr;     /* chromosome c = codes[chr_off[c] .. chr_off[c + 1]) */
} SpdpGenome;
typedef struct SpdpLocus {
    int32_t query, chr, rvs;         /* whose, where: chromosome, strand                                                   */
    int32_t base, len;               /* region [base, base + len) of the chromosome's forward strand                       */
    int32_t left, right;             /* range inside the region as the aligner sees it (reverse-complemented when rvs)     */
    int32_t jscr, n_hsp;             /* Seq::jscr, CdsNo                                                                   */
    int64_t hsp_off;                 /* its n_hsp + 1 SpdpJuxt records (the last one the closing slot) start at hsps[hsp_off] */
} SpdpLocus;
/* out: *loci (n_loci of them, query by query, best locus of a query first) and *hsps, both malloc'ed (free() them); status[i]
 * (may be NULL): TestOutput calls the query took, negative when the search ended without a locus.  sc: the gap penalties and
 * the intron penalty table the HSP chaining prices with (SpdpScoring gop / gep / lgop / lgep / codonk1, intpen / intpen_len).
 * hix: the same index on the host (its chromosome and random-score tables; spdp_blk_index_host_desc or the caller's own).
 * Protein queries (model->dvsp = 1) run against the index of the translated genome (`spaln -W -KP`, <db>.bkp; hix->drna = 0):
 * codes are amino-acid codes, the genome stays nucleotide codes -- a candidate region is turned into tron codes for the HSP
 * search as Seq::nuc2tron does (src/seq.cc:774-798), a pair whose ends the HSPs moved is searched again on the grown region
 * (FindHsp's retry, src/blksrc.cc:2462-2466), and jy of an HSP is a nucleotide position of the region's tron sequence. */
int spdp_blk_find(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                  const struct SpdpWilipModel* model, const SpdpScoring* sc, const SpdpBlkFindParams* prm,
                  const uint8_t* codes, const int64_t* offs, const int32_t* left, const int32_t* right, int32_t n,
                  SpdpLocus** loci, int32_t* n_loci, SpdpJuxt** hsps, int32_t* status);

/* ---- block search, fourth slice (round 5): the index builder ------------------------------------------------------------
 * `spaln -W -KD genome.mfa` for the block index (<db>.bkn): MakeBlk::idxblk / m_idxblk + blkscrtab + findChrBbound
 * (src/blksrc.cc:1185-1241, 1459-1593, 944-997, 583-596) with Block::c2w (:448-464) and Bitpat_wq::word (src/bitpat.cc:188-212).
 * Every residue of the genome ends a word of every bit pattern; a word counts once per block (blklen residues + the margin of
 * the widest pattern) at the phases its run of unambiguous residues gives it; words by how many blocks hold them get a
 * score, the rare enough ones a posting list in block order.  On the device: the words of all positions, their (word, block)
 * keys and the word counts in one pass over the resident genome, a radix sort of the keys, the lists compacted; on the host:
 * the scores (the reference's libm logarithms), the cut-off, the file.  The reference builds its blocks in two ways, and
 * the tables differ: without -t a block's word state starts afresh behind the previous block's end (threaded = 0), with
 * -t >= 1 every block is scanned over its own blklen + margin residues (threaded = 1).  Nucleotide genomes; residues = the
 * library's codes, anything but A / C / G / T an ambiguous residue (letters the reference skips -- not IUPAC -- must not be in
 * the array).  Refused: a word that lies in more than 65 535 blocks (the reference's 16-bit counters wrap there). */
typedef struct SpdpBlkBuildParams {  /* BlkWcPrm as setupbitpat leaves it (src/blksrc.cc:680-737)                          */
    int32_t ktuple, nshift, blklen, maxgene, nbitpat, afact;
    uint32_t bitpat, bitpat2;        /* nbitpat = 1: the one pattern (2^ktuple - 1 = contiguous); 3 / 5: the spaced pairs     */
    int32_t threaded;
} SpdpBlkBuildParams;
/* what `spaln -W -KD [-XC<n>]` picks for a FASTA file of fasta_bytes bytes (k from its logarithm, blklen and MaxGene from its
 * square root; the spaced patterns of DefBitPat[k] when nbitpat > 1); nbitpat = 0 or 1: the contiguous k-mer.  0, or -1. */
int spdp_blk_build_params_default(int64_t fasta_bytes, int32_t nbitpat, SpdpBlkBuildParams* p);
/* the index of `genome`, with the search parameters derived as spdp_blk_index_read derives them from a file (opts may be
 * NULL).  seconds (may be NULL): [0] device passes (words, sort, lists), [1] host (scores, cut-off, tables), [2] the call. */
SpdpBlkIndexHost* spdp_blk_index_build(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParams* p,
                                       const SpdpBlkSearchOpts* opts, double* seconds);
/* The translated index, `spaln -W -KP genome.mfa` (<db>.bkp: what protein queries are searched in; spdp_blk_find with model->dvsp = 1,
 * spdp_map_align_h): MakeBlk::idxblk / m_idxblk with Block::c2w6 / c2w6_pp (src/blksrc.cc:466-532), blkscrtab(segn) (:879-942).  A
 * residue completes one codon on each strand; the codon's class in the reduced amino-acid alphabet (ReducWord's g2r, src/bitpat.cc:
 * 88-106) joins the word of its reading frame; words are taken every nshift codons counted from the start of the open reading frame,
 * reach their block minorf residues later, and are dropped when their frame closes before minorf nucleotides.  On the device: the
 * classes of both strands' codons per residue, the frames' last class-less codon as six prefix maxima, the reference's delay ring
 * followed per word (no state carried from residue to residue); then the nucleotide builder's sort and list passes.  On the host: the
 * scores with their composition term -- a running sum over the word table in the reference's order -- and the file.
 * b.ktuple = amino acids per word (3 .. 7), b.nbitpat = 1 and b.bitpat = 2^k - 1 (contiguous words: what -KP builds unless -XC is
 * given), b.threaded as above.  acomp: MakeBlk::prepacomp's per-class terms (src/blksrc.cc:844-877), functions of the reference's
 * substitution-matrix tables and of -Xp / -Xq only -- not of the genome: constants of a deployment.  The values of the reference's
 * defaults (twenty classes, PAM 20) are in spaln_amd/defaults.py (BLOCK_ACOMP_20), recorded from the compiled reference by
 * oracle/ref_build/idx_tap.cc; for other alphabets the caller records them the same way.  convtab: ReducWord's iConvTab over the tron
 * codes (3 .. 22 = A .. V, 23 the AGY serines, 24 Sec), written to the file as the search reads it.
 * ContBlk::MaxBlk: the reference leaves it unset on this path and its files carry 65535; so do these. */
typedef struct SpdpBlkBuildParamsP {
    SpdpBlkBuildParams b;
    int32_t nalpha, minorf;          /* wcp.Nalpha (-XA), MinOrf (-Xr, nucleotides; 30)                                      */
    double aaafact;                  /* -Xq (1)                                                                              */
    double acomp[20];
    int32_t convts;                  /* entries of convtab (27 for the tron alphabet)                                        */
    uint8_t convtab[32];
} SpdpBlkBuildParamsP;
/* what `spaln -W -KP` picks for a FASTA file of fasta_bytes bytes (k from 0.36 ln(bytes), 3 .. 6; twenty classes, MinOrf 30); p->acomp
 * is kept as the caller set it.  0, or -1. */
int spdp_blk_build_params_default_p(int64_t fasta_bytes, SpdpBlkBuildParamsP* p);
SpdpBlkIndexHost* spdp_blk_index_build_p(SpdpContext* ctx, const SpdpGenome* genome, const SpdpBlkBuildParamsP* p,
                                         const SpdpBlkSearchOpts* opts, double* seconds);
/* the index as the reference's file (format version 26; the five pointers of its header, which the reference writes as its
 * heap held them, as zeros; ConvTab entries the reference leaves unset as "ambiguous"): 0, or -1 */
int spdp_blk_index_write(const SpdpBlkIndexHost* h, const char* path);

/* ---- map and align (round 5): the aligner's caller for nucleotide queries, inside the library ------------------------------
 * What spaln's per-query driver does between the block search and the printer when the genome is searched (-Q4 .. -Q7, cDNA
 * queries; src/spaln.cc:880-1010 spalign2 / blkaln, genomicseq at :913): every candidate locus of a query becomes a problem
 * -- its region cut from the genome (reverse-complemented when the locus is on the other strand), the region's splice
 * signals (Exinon::intron53, here spdp_signals.hip on the device for all loci of a chunk in one launch), the locus' HSPs --,
 * the seeded aligner runs on all of them (spdp_align_s_seeded, the recursion levels searched by sp->wilip), skl_rngS_ng
 * rescoring follows (spdp_skl_rng_s), and of a query's loci the one with the highest Gsinfo::fstat.val stays (the first one
 * on ties).  Its exons come back in the coordinates the reference prints (-O4: query positions 1-based inclusive, chromosome
 * positions 1-based on the forward strand, left > right on the reverse strand: Seq::SiteNo, src/seq.h).
 * One call = spdp_blk_find + one batched signal launch + one seeded call + one rescoring call per chunk of loci.
 * ori = a->inex.ori as spaln_job sets it (src/spaln.cc:1153-1156): 1 = the query as given (-S1), 3 = both orientations (the
 * default for a cDNA without a poly-A tail: alignS_ng(.., 3) -- the walk on the pair as given and on the reverse-complemented
 * query against the other strand of the locus, both with Exinon::both_ori signals; the reverse one stays only if it scores
 * strictly higher). */
typedef struct SpdpMapExon { int32_t q_left, q_right, g_left, g_right; } SpdpMapExon;
typedef struct SpdpMapGene {
    int32_t chr, rvs;                /* -1 / 0 when the query has no alignment; rvs: the strand the gene lies on              */
    int32_t q_rev;                   /* ori = 3 only: the reverse-complemented query gave the alignment (its exons then carry
                                        descending query positions, as the reference prints them)                            */
    int32_t score, val;              /* skl_rngS_ng's return value, Gsinfo::fstat.val of the locus that stayed             */
    int32_t n_loci;                  /* candidate loci of the query that were aligned                                      */
    int32_t n_exons;
    int64_t exon_off;                /* its exons: (*exons)[exon_off .. exon_off + n_exons)                                */
} SpdpMapGene;
/* genes: n records of the caller; *exons: malloc'ed (free() it); seconds (may be NULL): [0] block search, [1] regions +
 * signals, [2] seeded alignment, [3] rescoring + selection.  sp->wilip must be set (the HSP searches are the library's). */
int spdp_map_align_s(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                     const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpSignalModel* sigmodel,
                     const SpdpBlkFindParams* fprm, const SpdpRescoreParams* rp,
                     const uint8_t* codes, const int64_t* offs, int32_t n, int32_t ori,
                     SpdpMapGene* genes, SpdpMapExon** exons, double* seconds);

/* The same for protein queries against the translated index (`spaln -W -KP`, <db>.bkp): spdp_blk_find (model->dvsp = 1) -> per locus
 * the region as tron codes and its SGPT6 signals (one launch of the signal kernels per chunk) -> spdp_align_h_seeded with the
 * library's own HSP searches -> the junction phases of the walks written back -> spdp_skl_rng_h -> the locus with the highest
 * fstat.val: what blkaln / genomicseq / spalign2 do for one query (src/spaln.cc:846-1010, 1137-1152), for a batch.  codes:
 * amino-acid codes; exons: query positions in residues, genomic positions as -O4 prints them.  sp->wilip: the protein model. */
int spdp_map_align_h(SpdpContext* ctx, const SpdpBlkIndex* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                     const struct SpdpScoringH* sc, const SpdpSeedParams* sp, const struct SpdpSignalModelH* sigmodel,
                     const SpdpBlkFindParams* fprm, const struct SpdpRescoreParamsH* rp,
                     const uint8_t* codes, const int64_t* offs, int32_t n,
                     SpdpMapGene* genes, SpdpMapExon** exons, double* seconds);

/* ---- device groups, continued ------------------------------------------------------------------------------------------ */
/* the same sharding for the calls of the seeded path, rescoring and the block vote (rounds 3 / 4).  The HSP source of a
 * seeded call is asked with the CALLER's query numbers, from the worker threads of every member.  spdp_group_blk_vote takes
 * one index per member, created on that member's context (spdp_group_context(g, r)). */
int spdp_group_align_s_seeded(SpdpGroup* g, const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpProblem* probs, int n_probs,
                              const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level,
                              const SpdpHspSource* src, SpdpAlignment* out);
int spdp_group_align_h_seeded(SpdpGroup* g, const SpdpScoringH* sc, const SpdpSeedParams* sp, const SpdpProblemH* probs,
                              int n_probs, const SpdpJuxt* const* hsps, const int32_t* n_hsps, const int32_t* lowest_level,
                              const SpdpHspSource* src, SpdpAlignment* out);
int spdp_group_skl_rng_s(SpdpGroup* g, const SpdpScoring* sc, const SpdpRescoreParams* rp, const SpdpProblem* probs, int n_probs,
                         const SpdpAlignment* aln, SpdpRescored* out);
int spdp_group_skl_rng_h(SpdpGroup* g, const SpdpScoringH* sc, const SpdpRescoreParamsH* rp,
                         const SpdpProblemH* probs, int n_probs, const SpdpAlignment* aln, SpdpRescored* out);
SpdpContext* spdp_group_context(SpdpGroup* g, int member);
int spdp_group_blk_vote(SpdpGroup* g, const SpdpBlkIndex* const* ix, const uint8_t* codes, const int64_t* offs,
                        const int32_t* left, const int32_t* right, const int32_t* stop_at, int32_t n, int32_t* out, int32_t out_cap);
/* spdp_map_align_s with the queries sharded over the members by length (each member: its own index, block search, signals,
 * walks and rescoring; nothing is exchanged between members); genes / exons in the caller's query order */
int spdp_group_map_align_s(SpdpGroup* g, const SpdpBlkIndex* const* ix, const SpdpBlkIndexDesc* hix, const SpdpGenome* genome,
                           const SpdpScoring* sc, const SpdpSeedParams* sp, const SpdpSignalModel* sigmodel,
                           const SpdpBlkFindParams* fprm, const SpdpRescoreParams* rp,
                           const uint8_t* codes, const int64_t* offs, int32_t n, int32_t ori,
                           SpdpMapGene* genes, SpdpMapExon** exons);

#ifdef __cplusplus
}
#endif
#endif /* SPDP_H_ */
