// spaln_gpu_main.cc -- the reference's command line program (src/spaln.cc, compiled from where it lies, not copied) and the
// binding of spaln_gpu_shim.cc as ONE translation unit: the shim's batching boundary does, after the worker threads have
// joined, what the reference's blkaln does with an alignment (rescoring, filter, order, output), and that needs statics of
// src/spaln.cc (outputs, skl2exrng, thread_num).  MasterWorker and all_in_func end a run of the workers with
// closeGeneRecord() (src/spaln.cc:1201, 1454): that one call is pointed at the shim's spaln_gpu_after_workers(), which
// aligns everything the workers recorded, writes the output and then calls the real closeGeneRecord() (src/sqpr.cc:997).
// This is the whole of the change to the reference's driver; INTEGRATION.md shows the equivalent patch.
// (the standard headers the shim needs come first: src/iolib.h overloads fclose for gzFile, after which <wchar.h> no longer parses)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
void spaln_gpu_after_workers();
#define closeGeneRecord spaln_gpu_after_workers
#include "spaln.cc"
#undef closeGeneRecord
extern void closeGeneRecord();
#include "spaln_gpu_shim.cc"
