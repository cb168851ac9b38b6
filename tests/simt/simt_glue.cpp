// tests/simt/simt_glue.cpp - TEST INFRASTRUCTURE (tests/simt/simt.h): what the interpreted build of the library needs besides
// the fake runtime - the storage behind the one `extern __shared__` array of the device sources (dynamic LDS of
// hb_experiments.hip.h: at most 2048 counters) and a marker the loader can ask for.
#include <hip/hip_runtime.h>

namespace hbk {
thread_local uint4 s_tile[2048 * 4];
}
extern "C" int hb_simt_interpreter(void) { return 1; }
extern "C" void hb_simt_stats(unsigned long long out[5])
{
    const simt::Stats &s = simt::stats();
    out[0] = s.launches;
    out[1] = s.blocks;
    out[2] = s.collectives;
    out[3] = s.partial_groups;
    out[4] = s.reads_of_inactive_lanes;
}
