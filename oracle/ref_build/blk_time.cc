// blk_time.cc -- TEST INFRASTRUCTURE ONLY (build container; bench.py's cpu_baseline of the block-search leg): SrchBlk::findblock under
// its own name as a timed call of the reference's routine (renamed to ref_findblock in a copy of its object file, oracle/ref_build/
// Makefile).  At exit the program prints how many calls its worker threads made and the thread-seconds they spent inside:
//     [blk_time] findblock: <calls> calls, <seconds> thread-seconds
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "aln.h"
#include "utilseq.h"
#include "blksrc.h"

extern "C" int ref_findblock(SrchBlk* self, Seq** sqs);
static std::atomic<long long> g_ns{0}, g_calls{0};
namespace { struct AtExit { ~AtExit() { fprintf(stderr, "[blk_time] findblock: %lld calls, %.6f thread-seconds\n", (long long) g_calls, g_ns * 1e-9); } } g_at_exit; }

int SrchBlk::findblock(Seq** sqs)
{
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = ref_findblock(this, sqs);
	g_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
	++g_calls;
	return rc;
}
