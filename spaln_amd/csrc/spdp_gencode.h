// spdp_gencode.h -- the standard genetic code in the reference's tron alphabet (shared by the rescoring walk, the protein
// engines' launchers and the seeded walk)
#ifndef SPDP_GENCODE_H_
#define SPDP_GENCODE_H_
#include <stdint.h>
#include <string.h>

// the standard genetic code in the reference's tron alphabet (A = 3 ... V = 22, AGY serines 23, TGA 24,
// TAA / TAG 25), as its static spj_tron_tab / tnredctab assume (src/codepot.h:130, src/seq.cc:41)
inline void spdp_genetic_code_tables(uint8_t mid[32], uint8_t tron_of[64])
{
    static const char* aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";   // TCAG order
    static const char* order = "ARNDCQEGHILKMFPSTWYV";
    const int tcag2acgt[4] = {3, 1, 0, 2};
    memset(mid, 4, 32);
    for (int c = 0; c < 64; ++c) {
        const int b1 = tcag2acgt[c >> 4], b2 = tcag2acgt[(c >> 2) & 3], b3 = tcag2acgt[c & 3];
        const char aa = aas[c];
        int code;
        if (aa == '*') code = (b1 == 3 && b2 == 2 && b3 == 0) ? 24 : 25;        // TGA : TAA / TAG
        else if (aa == 'S' && b1 == 0) code = 23;                                // AGY
        else code = 3 + (int) (strchr(order, aa) - order);
        tron_of[16 * b1 + 4 * b2 + b3] = (uint8_t) code;
        mid[code] = (uint8_t) b2;
    }
}

#endif
