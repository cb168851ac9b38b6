// hb_threads.h - how many host threads the OpenMP parts of the library should use.
// omp_get_num_procs() reports the machine's hardware threads (256 on the MI355X boxes) even when a container's cgroup
// lets the process run on far fewer CPUs at a time (16 there): 256 threads time-slicing 16 CPUs cost the store writer
// and the column reader more than half their throughput.  HB_HOST_THREADS overrides.
#pragma once
#include <omp.h>

#include <cstdio>
#include <cstdlib>

namespace hb {
inline int host_threads()
{
    static const int n = [] {
        // the machine's processors, NOT omp_get_max_threads(): that is a mutable setting another OpenMP user of the process
        // (the CPU oracle's omp_set_num_threads in bench.py) may have left at 1 or 2
        int hw = omp_get_num_procs();
        if (hw < 1) hw = 1;
        if (const char *e = std::getenv("HB_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v >= 1) return v;
        }
        long long quota = -1, period = -1;
        if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = std::atoll(q);
            std::fclose(f);
        } else {
            if (std::FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
                std::fclose(g);
            }
            if (std::FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g, "%lld", &period) != 1) period = -1;
                std::fclose(g);
            }
        }
        if (quota > 0 && period > 0) {
            const long long cpus = (quota + period - 1) / period;
            if (cpus >= 1 && cpus < hw) return (int)cpus;
        }
        return hw;
    }();
    return n;
}
} // namespace hb
