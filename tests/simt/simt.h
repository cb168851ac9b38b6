// tests/simt/simt.h - TEST INFRASTRUCTURE, never part of the product.
//
// A small SIMT interpreter: it runs the library's DEVICE SOURCES (stract_amd/csrc/*.hip*, unchanged) on the host, so that the
// logic of the kernels - tiling, quad/wave collectives, LDS hand-overs, bitmaps, the lazy double buffer, the epilogues - can be
// compared with the oracle by `pytest -m "not gpu"` in a container without a GPU, before GPU minutes are spent on a change.
// It says NOTHING about performance, memory ordering between workgroups or anything else the hardware decides; the parity
// tests proper stay the `-m gpu` tests on the MI355X.  Nothing under stract_amd/ refers to it, the shipped library is never
// built from it, and stract_amd/_lib.py loads a library that reports this build only when a test asks for it by path.
//
// Execution model: the workgroups of a launch run one after the other; the lanes of a workgroup are fibers (own stacks, a
// dozen lines of x86-64 context switch).  A lane runs until it reaches a cross-lane operation - ballot, shuffle, DPP quad
// permutation, a wavefront fence / barrier (the kernels hand LDS data over inside a wave: on the machine the lanes run in
// lock step, here the fence is where the other lanes catch up), __syncthreads - and waits there.  When every live lane of a
// wave waits, the lanes at the LOWEST code address are served together, as ONE operation with exactly those lanes active
// ("min-PC" re-convergence: lanes that left a loop early wait at the code behind it until the others arrive, lanes inside
// the loop keep being served), and continue.  Atomics are plain operations (one OS thread per launch).
#pragma once
#include <cstdint>
#include <functional>

namespace simt {
struct Idx {
    unsigned x = 0, y = 0, z = 0;
};
extern thread_local Idx threadIdx_, blockIdx_, blockDim_, gridDim_;

enum Kind { K_BALLOT = 1, K_SHFL, K_SHFL_UP, K_SHFL_DOWN, K_SHFL_XOR, K_DPP, K_WAVE_SYNC, K_BLOCK_SYNC };
// one cross-lane operation of the calling lane; returns its result (see simt_core.cpp)
uint64_t collective(int kind, uint64_t value, uint64_t arg) __attribute__((noinline));
void launch(unsigned grid, unsigned block, const std::function<void()> &body);
// statistics of the last launches (tests print them on a failure)
struct Stats {
    uint64_t launches = 0, blocks = 0, collectives = 0, partial_groups = 0, reads_of_inactive_lanes = 0;
};
Stats &stats();
} // namespace simt
