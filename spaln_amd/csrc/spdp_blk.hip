// spdp_blk.hip -- the block search's vote on the device: one query per lane (spdp_blk_core.h has the routine and says why).
//
// A launch fills the chip with persistent lanes (waves x 64), each bound to a private slab of HBM; lane g takes queries
// g, g + lanes, ...  The traffic is what the reference's loop does per word -- one posting list (sequential 4-byte reads),
// two or three score slots and a handful of hash / heap slots per listed block, all in the lane's slab -- i.e. random
// 4 .. 8-byte accesses: the bound is HBM transactions in flight, not bandwidth in bytes; the lanes of a wave diverge freely
// (every loop is data dependent), which costs issue slots, not correctness.  No LDS, no cross-lane traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define SPDP_HD __device__ __forceinline__
#include "spdp_blk_core.h"
#include "spdp_blk_internal.h"

__global__ void __launch_bounds__(64) spdp_blk_vote_kernel(BlkArgs A)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= A.n_lanes) return;
    const BlkDev& ix = A.ix;
    BlkWork w;
    blk_work_bind(w, ix, A.slabs + (size_t) g * A.slab_ints, A.touched_cap);
    BlkPair* bpair = (BlkPair*) (A.scratch + (size_t) g * A.scratch_ints);
    uint32_t* sw = (uint32_t*) (bpair + ix.ncand + 2);
    for (int q = g; q < A.n; q += A.n_lanes) {
        const int64_t o = A.offs[q];
        const int len = (int) (A.offs[q + 1] - o);
        BlkVote v;
        int calls = 0;
        const int reached = blk_vote_run(ix, w, v, A.codes + o, len, A.left[q], A.right[q], A.stop_at ? A.stop_at[q] : 0, &calls);
        blk_emit_and_reset(ix, w, v, reached, calls, bpair, sw, A.out + (size_t) q * A.out_cap, A.out_cap);
    }
}

extern "C" hipError_t spdp_blk_launch(const BlkArgs* a, hipStream_t s)
{
    BlkArgs A = *a;
    hipLaunchKernelGGL(spdp_blk_vote_kernel, dim3((A.n_lanes + 63) / 64), dim3(64), 0, s, A);
    return hipGetLastError();
}
