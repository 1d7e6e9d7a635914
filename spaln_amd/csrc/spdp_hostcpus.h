// spdp_hostcpus.h -- how many host threads are worth starting
#ifndef SPDP_HOSTCPUS_H_
#define SPDP_HOSTCPUS_H_
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host
// may be granted 16 CPUs worth of time: more runnable threads than that only contend)
static inline int spdp_host_cpus()
{
    int n = (int) std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : 1 << 20, (int) CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0}; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
            n = std::min(n, std::max(1, (int) ((atol(q) + period / 2) / period)));
        fclose(f);
    }
    return std::max(1, n);
}

#endif
