// spdp_region.h -- a candidate region of the block search as residue codes in host memory: cut from the chromosome, turned to the other
// strand, translated.  The device search reads regions where they lie in the resident genome (spdp_hsp.hip does the same three
// things on the fly); the host form of the search -- the few tasks the device hands over, the tests' checker -- needs them as an array.
#ifndef SPDP_REGION_H_
#define SPDP_REGION_H_
#include <stdint.h>
#include <string.h>
#include <vector>
#include "spdp_gencode.h"

namespace spdp_region {

inline uint8_t other_strand(uint8_t c)  // A 2, C 3, G 5, T 9; the ambiguity codes stay (src/seq.cc: comrev on the 4-bit codes)
{
    switch (c) { case 2: return 9; case 9: return 2; case 3: return 5; case 5: return 3; default: return c; }
}

// Seq::nuc2tron (src/seq.cc:774-798, src/utilseq.cc:204-225): position p becomes the codon (p - 1, p, p + 1) in the tron alphabet; the
// pads of the sequence stand for the residues before the first and behind the last one
inline void to_tron(uint8_t* s, int len)
{
    static const uint8_t plain[17] = {15, 15, 0, 1, 4, 2, 5, 6, 10, 3, 7, 8, 10, 9, 12, 13, 14};     // A C G T -> 0 .. 3, the rest >= 4 (ncredctab)
    static const uint8_t first_of[17] = {0, 0, 0, 1, 2, 2, 0, 2, 0, 3, 3, 3, 1, 1, 2, 3, 0};         // an ambiguous third base by its first element (ncelements)
    static const uint8_t likely[4] = {14, 3, 10, 13};                                                  // first base unknown: LYS, ALA, GLY, LEU by the middle one
    struct Table { uint8_t tron_of[64]; Table() { uint8_t mid[32]; spdp_genetic_code_tables(mid, tron_of); } };
    static const Table t;
    int before = 0;
    for (int p = 0; p < len; ++p) {
        const int c0 = before, c1 = s[p] > 16 ? 16 : s[p], c2 = p + 1 < len ? (s[p + 1] > 16 ? 16 : s[p + 1]) : 0;
        before = c1;
        s[p] = c1 <= 1 ? 1 : plain[c1] >= 4 ? 2 : plain[c0] >= 4 ? likely[plain[c1]] : t.tron_of[16 * plain[c0] + 4 * plain[c1] + first_of[c2]];
    }
}

// into dst[0 .. len]: the region's residues and a closing 0
inline void materialize_into(const uint8_t* genome, const int64_t* chr_off, int chr, int base, int len, bool rvs, bool tron, uint8_t* dst)
{
    const uint8_t* src = genome + chr_off[chr] + base;
    if (!rvs) memcpy(dst, src, (size_t) len);
    else for (int i = 0; i < len; ++i) dst[i] = other_strand(src[len - 1 - i]);
    dst[len] = 0;
    if (tron) to_tron(dst, len);
}
inline void materialize(const uint8_t* genome, const int64_t* chr_off, int chr, int base, int len, bool rvs, bool tron, std::vector<uint8_t>& out)
{
    out.resize((size_t) len + 1);
    materialize_into(genome, chr_off, chr, base, len, rvs, tron, out.data());
}

}   // namespace spdp_region
#endif
