// spdp_seeded_rv.h -- where the walks of a seeded call (spdp_seeded.cpp, spdp_seeded_h.cpp) meet the thread that runs
// the device: a walk parks a DP request and sleeps; when every walk in flight sleeps, all parked requests run as one batch
#ifndef SPDP_SEEDED_RV_H_
#define SPDP_SEEDED_RV_H_
#include <condition_variable>
#include <mutex>
#include <vector>
#include "spdp_seeded_walk.h"

namespace spdp_seed {

struct Parked {
    int query = 0, kind = 0;
    Span s{}; SpdpWindow w{}; int cut[2] = {0, 0};
    int score = SPDP_NEVSEL;
    std::vector<SpdpSkl> rec;
    bool done = false, failed = false;
    int flags = 0;                              // SpdpAlignment::flags of the request
};

struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv_walk, cv_main;
    std::vector<Parked*> parked;
    int running = 0;                            // walker threads that are neither parked nor finished
};

}   // namespace spdp_seed
#endif
