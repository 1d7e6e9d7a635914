// spdp_h_requests.h -- explicit DP requests against the resident inputs of a batch of aa x genome problems: what the
// protein seeded walk (spdp_seeded_h.cpp) asks of the ladder in spdp_h_api.cpp
#ifndef SPDP_H_REQUESTS_H_
#define SPDP_H_REQUESTS_H_
#include "../../include/spdp.h"

struct HStore;
struct SpdhRequest {
    int parent;                 // index of the problem whose sequences / signals the request reads
    int al, ar, bl, br;         // Seq::left / right of both sequences
    uint8_t exg[4];             // a_exgl, a_exgr, b_exgl, b_exgr
    SpdpWindow w;
    int kind;                   // 0 lspH_ng(wdw), 1 trcbkalignH_ng(wdw, true, mc), 3 trcbkalignH_ng(wdw, false)
    int cut_l, cut_r;           // mc (kind 1): the sweep jumps over genomic columns (cut_l, cut_r]; cut_r <= cut_l: none
};
HStore* spdh_store_open(SpdpContext* ctx, const SpdpScoringH* sc, const SpdpProblemH* probs, int n);
void spdh_store_close(HStore* st);
// lane: the context whose streams and scratch the batch uses (null: the store's own); one batch per context at a time
int spdh_run_requests(HStore* st, const SpdhRequest* reqs, int n, SpdpAlignment* out, SpdpContext* lane = nullptr);
#endif
