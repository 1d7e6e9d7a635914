// spdp_blk_dev.h -- what the block search's device code and its host side share: the index as the kernels see it and the
// launch arguments of the vote (spdp_blk_vote.hip <-> spdp_blk_api.cpp).  Data only.
#ifndef SPDP_BLK_DEV_H_
#define SPDP_BLK_DEV_H_
#include <stdint.h>
#include <stddef.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#endif

#define SPDP_BLK_MAX_SHIFT 32
#define SPDP_BLK_PRE_PHASES 12          // phases of a direction whose posting lists are fetched ahead (Nshift beyond: fetched one by one)
#define SPDP_BLK_RES_CAP 2048            // query residues a wave keeps in LDS (longer queries are read where they lie)
#define SPDP_BLK_HASH_LEVELS 4          // the run hash may grow three times (x ~8) before a query is reported as SPDP_BLK_TABLE

struct BlkDev {                         // the index and the search parameters (pointers into HBM on the device side)
    int32_t nalpha, tabsize, nshift, nbitpat, convts, n_chr, kk, drna, maxmmc, nseg, minsigpr, ncand, nascr;
    int32_t maxblock, extblock, extblockl, shortquery, hh_size1, hh_size2, hb_size1, hb_size2, ha_size1, ha_size2, gdb;
    int32_t hh_sizes[SPDP_BLK_HASH_LEVELS];     // hh_size1 and what the reference's table becomes when it grows: the next prime >= twice the size
    int32_t hb_sizes[SPDP_BLK_HASH_LEVELS], ha_sizes[SPDP_BLK_HASH_LEVELS];     // the same for the position tables of the two kinds of best-of lists
    int32_t maxlist;                            // longest posting list (ContBlk::MaxBlk)
    float rbscoef, rbscons;
    double bclw, bcup, bcce, app_c;
    const uint8_t* convtab;
    const uint16_t* nblk;
    const int16_t* wscr;
    const int32_t* blkp;
    const uint32_t* blkb;
    const int32_t* rscrtab;
    const int32_t* chr;
    const int32_t* bitpat;
    int32_t pat_off[3];                 // where pattern k starts in bitpat: weight, width, wshift, exam[2 * weight]
};

// the smallest prime >= n (the table sizes of the reference's containers: src/supprime.cc:375)
inline uint32_t blk_next_prime(uint32_t n)
{
    if (n <= 3) return n;
    n |= 1;
    for ( ; ; n += 2) {
        uint32_t x = 3;
        while (x * x <= n && n % x) x += 2;
        if (x * x > n) return n;
    }
}
inline void blk_fill_hash_levels(BlkDev& ix)
{
    ix.hh_sizes[0] = ix.hh_size1; ix.hb_sizes[0] = ix.hb_size1; ix.ha_sizes[0] = ix.ha_size1;
    for (int l = 1; l < SPDP_BLK_HASH_LEVELS; ++l) {
        ix.hh_sizes[l] = (int32_t) blk_next_prime(2u * (uint32_t) ix.hh_sizes[l - 1]);
        ix.hb_sizes[l] = (int32_t) blk_next_prime(2u * (uint32_t) ix.hb_sizes[l - 1]);
        ix.ha_sizes[l] = (int32_t) blk_next_prime(2u * (uint32_t) ix.ha_sizes[l - 1]);
    }
}

// One wave per query.  A wave's working set: LDS (scan positions, the eight bounded queues with their position tables, the run
// hash when it fits) and a slab of HBM (score records of every block and direction, the larger run-hash levels, staging).
struct BlkVoteArgs {
    const BlkDev* ix;                           // in device memory (a kernel argument whose address is taken would be copied to every lane's stack)
    const uint8_t* codes; const int64_t* offs; const int32_t* left; const int32_t* right; const int32_t* stop_at;
    int32_t* out; int out_cap, n;
    uint8_t* slabs; size_t slab_bytes;          // one per wave of the launch; zero when allocated, kept between launches
    uint32_t* next;                             // the launch's query counter (zero at launch)
    int n_waves;
    int hh_in_lds;                              // level 0 of the run hash lives in LDS
    uint32_t lds_bytes;
    int res_cap;
};
#ifdef __HIPCC__
extern "C" hipError_t spdp_blk_vote_launch(const BlkVoteArgs* a, hipStream_t s);
#endif
extern "C" size_t spdp_blk_vote_slab_bytes(const BlkDev* ix, int hh_in_lds);
extern "C" uint32_t spdp_blk_vote_lds_bytes(const BlkDev* ix, int hh_in_lds);
#endif
