// spdp_hsp_dev.h -- the HSP search on the device (spdp_hsp.hip) and its host side (spdp_blk_api.cpp): tasks and launch arguments
#ifndef SPDP_HSP_DEV_H_
#define SPDP_HSP_DEV_H_
#include <stdint.h>
#include <stddef.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#endif
#include "../../include/spdp.h"

enum { HSP_TOO_LONG = 1, HSP_TOO_MANY = 2 };           // a task the device form does not hold: the host's form serves it

struct HspTask {
    int64_t a_off;                      // the query's codes in HspArgs::codes
    int64_t g_off;                      // first residue of the region in HspArgs::genome (forward strand)
    int32_t a_len, a_left, a_right;     // the query and its active range
    int32_t b_len, rvs;                 // the region: b_len residues, read as the other strand when rvs
    int32_t a_exgl, a_exgr;             // free query ends (end bonuses)
    int32_t pad;
};

struct HspArgs {
    const uint8_t* codes; const uint8_t* genome;
    const HspTask* tasks; int n_tasks;
    SpdpWilipLevel level;               // the level's parameters (level 0 of the model: FindHsp's level -1 uses them, scaled for short queries)
    const int32_t* mtx; int mtx_rows, mtx_cols;
    const uint8_t* tron_of;             // 64 codons -> tron codes (protein queries)
    int bbt, shortquery, end_bonus, crs, ser, ser2;
    int hash_slots, hit_cap, seg_cap;   // LDS budget of a wave: query words (a power of two), shared words, seed segments
    int32_t* out; int out_cap;          // per task out_cap records of 8 ints {jx, jy, jlen, nid, jscr, diagonal, first word, 0}
    int32_t* counts;                    // per task {records, flags}
};
#ifdef __HIPCC__
extern "C" hipError_t spdp_hsp_launch(const HspArgs* a, int n_waves, hipStream_t s);
#endif
extern "C" uint32_t spdp_hsp_lds_bytes(const HspArgs* a);
#endif
