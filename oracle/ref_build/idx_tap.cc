// idx_tap.cc -- TEST INFRASTRUCTURE ONLY (build container): MakeBlk::WriteBlkInfo under its own name as a call of the reference's
// routine (renamed to ref_write_blk_info in a copy of its object file, oracle/ref_build/Makefile) that first prints what
// MakeBlk::prepacomp (src/blksrc.cc:844-877) left in the object -- the per-class composition terms of the translated index's word
// scores, which `spaln -W -KP` derives from its substitution-matrix tables:
//     [idx_tap] acomp <Nalpha> <deltaa> <acomp[0]> ... <acomp[Nalpha - 1]>        (hex floats: exact)
// tests/golden/make_blk_goldens.py turns the line into the table spaln_amd/defaults.py holds.  Compiled with -fno-access-control.
#include <cstdio>
#include "aln.h"
#include "utilseq.h"
#include "blksrc.h"

extern "C" void ref_write_blk_info(MakeBlk* self);

void MakeBlk::WriteBlkInfo()
{
	if (acomp) {
	    fprintf(stderr, "[idx_tap] acomp %u %a", Nalpha, deltaa);
	    for (INT q = 0; q < Nalpha; ++q) fprintf(stderr, " %a", acomp[q]);
	    fputc('\n', stderr);
	}
	ref_write_blk_info(this);
}
