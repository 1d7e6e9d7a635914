// spdp_host.cpp -- the reference's dispatch around the DP engines, host side:
// Aln2s1::lspS_ng / trcbkalignS_ng / mimd_postwork (src/fwd2s1.cc:1667-1897),
// globalS_ng + stdskl / trimskl (src/fwd2s1.cc:2674-2694, src/gaps.cc:140-273).
#include "spdp_internal.h"

int spdp_align_s(SpdpContext* ctx, const SpdpScoring*, const SpdpProblem*, int, SpdpAlignment*)
{
    if (ctx) ctx->err = "spdp_align_s: not implemented yet";
    return -1;
}
int spdp_batch_align(SpdpBatch*, SpdpAlignment*, float*, int64_t*) { return -1; }
