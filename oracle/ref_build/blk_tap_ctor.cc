// blk_tap_ctor.cc -- TEST INFRASTRUCTURE ONLY (see blk_tap.cc): the three constructors of the reference's SrchBlk under
// their own names, each a call of the reference's constructor (renamed by objcopy in the object of blk_tap.cc) that also
// tells the recorder where the object lives.
#include "aln.h"
#include "utilseq.h"
#include "blksrc.h"

extern SrchBlk*	g_blk_tap_this;
extern "C" void ref_sbk_ctor_file(SrchBlk* self, Seq** sqs, const char* fn, bool gdb);
extern "C" void ref_sbk_ctor_mb(SrchBlk* self, Seq** sqs, MakeBlk* mb, bool gdb);
extern "C" void ref_sbk_ctor_copy(SrchBlk* self, SrchBlk* sbk, DbsDt* df);

SrchBlk::SrchBlk(Seq* sqs[], const char* fn, bool gdb) { ref_sbk_ctor_file(this, sqs, fn, gdb); g_blk_tap_this = this; }
SrchBlk::SrchBlk(Seq* sqs[], MakeBlk* mb, bool gdb) { ref_sbk_ctor_mb(this, sqs, mb, gdb); g_blk_tap_this = this; }
SrchBlk::SrchBlk(SrchBlk* sbk, DbsDt* df) { ref_sbk_ctor_copy(this, sbk, df); }
