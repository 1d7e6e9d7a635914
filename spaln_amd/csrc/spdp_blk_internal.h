// spdp_blk_internal.h -- launch arguments of the block vote (spdp_blk.hip <-> spdp_blk_api.cpp)
#ifndef SPDP_BLK_INTERNAL_H_
#define SPDP_BLK_INTERNAL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

struct BlkArgs {
    BlkDev ix;
    const uint8_t* codes; const int64_t* offs; const int32_t* left; const int32_t* right; const int32_t* stop_at;
    int32_t* out; int out_cap, n;
    int32_t* slabs; size_t slab_ints;          // n_lanes private slabs, zero between queries
    int32_t* scratch; size_t scratch_ints;     // per lane: the pair list and the two sort buffers
    int n_lanes, touched_cap;
};
extern "C" hipError_t spdp_blk_launch(const BlkArgs* a, hipStream_t s);
#endif
