// spdp_blk_build.h -- between the host side of the index builder (spdp_blk_index_io.cpp) and its device passes
// (spdp_blk_build.hip)
#ifndef SPDP_BLK_BUILD_H
#define SPDP_BLK_BUILD_H
#include <stdint.h>
#include <vector>

struct SpdpContext;
struct BlkBuildArgs {
    const uint8_t* codes; const int64_t* chr_off; const int32_t* chr_first;     // chr_first[c]: number of chromosome c's first block (1-based)
    int n_chr;
    int64_t G;                                  // residues in all
    int nbit, nshift, blklen, margin, threaded, weight;
    int width[5], spaced[5];
    int exam[5][16];                            // the offsets a pattern examines inside its window (the mirrored set for a reverse pattern)
    const int64_t* tile_carry;                  // the last ambiguous residue before a tile (-1: none)
    uint32_t* tcount;
    unsigned long long* keys; unsigned long long* n_keys; unsigned long long cap;
};
// the translated index (`spaln -W -KP`): amino-acid words of the six reading frames
struct BlkBuildArgsP {
    const uint8_t* codes; const int64_t* chr_off; const int32_t* chr_first;
    int n_chr;
    int64_t G;
    int K, nalpha, nshift, blklen, margin, minorf, threaded;
    uint32_t tabsize;
    uint8_t codon_class[64];                    // codon 16 b1 + 4 b2 + b3 (A C G T = 0 1 2 3) -> class; >= nalpha: none
    const int64_t* tile_carry;                  // six per tile: the last residue (same residue class mod 3, same strand) whose codon has no class
    uint32_t* tcount;
    unsigned long long* keys; unsigned long long* n_keys; unsigned long long cap;
};
struct BlkBuildDev;
int spdp_blkidx_words_p(SpdpContext* ctx, const uint8_t* codes, const int64_t* chr_off, const int32_t* chr_first, int n_chr,
                        BlkBuildArgsP A, int key_bits, std::vector<uint32_t>& tcount, std::vector<uint32_t>& cnt, BlkBuildDev** out);
int spdp_blkidx_words(SpdpContext* ctx, const uint8_t* codes, const int64_t* chr_off, const int32_t* chr_first, int n_chr,
                      BlkBuildArgs A, uint32_t tabsize, int key_bits, std::vector<uint32_t>& tcount, std::vector<uint32_t>& cnt,
                      BlkBuildDev** out);
int spdp_blkidx_lists(SpdpContext* ctx, BlkBuildDev* d, const int32_t* blkp, int64_t word_no, uint32_t* blkb);
void spdp_blkidx_free(BlkBuildDev* d);
#endif
