// hb_experiments.hip.h - kernels of measured-and-rejected experiments, kept OFF the default path and out of the product
// header so that their measurements stay reproducible (hb_options.tune selects them; DESIGN.md "tried and rejected").
#pragma once
#include "hb_kernels.hip.h"

namespace hbk {

// ---- experiment (north-star "LDS-staged counter tiles"; off by default, hb_options.tune[7]) -----------
// Dense pull over the level-1 hub chunks with the `tile` hottest counters (device rows [0, tile)) staged in
// LDS once per workgroup: gathers of those sources are served from LDS instead of L2.  Same results as
// pass_kernel<false,false,false,false,4>; measured against it in profiles/r02*_lds_tile*.txt.
__global__ __launch_bounds__(256) void hub_lds_tile_kernel(const PassParams p, uint32_t tile)
{
    extern __shared__ uint4 s_tile[]; // tile * 4 uint4
    for (uint32_t i = threadIdx.x; i < tile * 4; i += 256) s_tile[i] = p.rd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 2, q = lane & 3;
    uint64_t row_lo = p.row_lo, row_hi = p.row_hi, tile0 = blockIdx.x, tstride = gridDim.x;
    if (p.xcd_map) {
        const int x = blockIdx.x & 7;
        row_lo = p.xcd_lo[x];
        row_hi = p.xcd_hi[x];
        tile0 = blockIdx.x >> 3;
        tstride = gridDim.x >> 3;
    }
    const uint64_t ntiles = (row_hi - row_lo + 63) >> 6;
    for (uint64_t t = tile0; t < ntiles; t += tstride) {
        const uint64_t row = row_lo + (t << 6) + ((uint64_t)wave << 4) + (uint64_t)g;
        if (row >= row_hi) continue;
        const uint64_t beg = p.row_ptr[row], end = p.row_ptr[row + 1];
        Acc acc;
        acc_zero(acc);
        if (beg < end) {
            const uint32_t first = p.src[beg];
            for (uint64_t e = beg; e < end; e += 16) {
                uint32_t idx[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint64_t ee = e + 4 * u + q;
                    idx[u] = (ee < end) ? p.src[ee] : first;
                }
                uint4 r[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t sv[4] = {quad_bcast<0>(idx[u]), quad_bcast<1>(idx[u]), quad_bcast<2>(idx[u]), quad_bcast<3>(idx[u])};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (sv[j] < tile) r[u][j] = s_tile[sv[j] * 4 + q];
                        else r[u][j] = p.rd[(uint64_t)sv[j] * 4 + q];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) acc_merge(acc, r[u][j]);
                }
            }
        }
        p.part[(row - p.n_pad) * 4 + q] = acc_value(acc); // dense: the partial is overwritten unread (see pass_kernel)
    }
}

} // namespace hbk
