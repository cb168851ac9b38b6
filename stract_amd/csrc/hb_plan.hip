// hb_plan.hip - the device work-layout planner: the same plan as build_plan() in hb_host.cpp, built in HBM
// from a CSR that is already on the device, with rocPRIM sorts / scans / selections and a few small kernels.
//
// Why: on the 100 M-host / 2 B-edge graph the host planner took 18-25 s (and 65 s at 5 B edges) for 0.27 s of
// HyperBall passes; a drop-in replacement of HarmonicCentrality::calculate (harmonic.rs:292) is judged on
// wall time.  What the plan is (device order, hub-row chunk trees, slice cuts, XCD groups) and why is described
// in hb_host.cpp / DESIGN.md; this file only restates HOW each step is computed in parallel:
//   device order      stable radix sort of (owner, ~out-degree) carrying the sid
//   rows + relabel    one 64-bit radix sort of (device row, hotness rank of the source) over all edges
//   chunk cutting     one wave per hub row walks its sorted list a chunk per step (ballot finds the cut)
//   XCD groups        stable sort by (slice, longer first), class prefix sums against eight thresholds per
//                     class (hb_internal.h XcdQuota, computed on the host from ten device-side sums), stable
//                     sort by group
//   upper levels      per hub row: scans give the new row ids / list offsets, a wave copies the id lists
//   assembly          scans of the row lengths, quad-per-row copies
// The result must equal the host planner's output entry for entry (tests/test_gpu.py::test_device_plan_equals_host_plan).
#include "hb_guard_alloc.h" // FIRST: no-op unless built with -DHB_GUARD_ALLOC=<mode> (debug allocators: guard pages / poison / red zones)
#include "hb_pool.h"        // then: every hipMalloc / hipFree below goes through the caching device allocator (shipped build)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "hb_internal.h"

namespace {

using hb::kNone;
using hb::kRowAlign;

// -DHB_DEBUG_BOUNDS (`make bounds`): index checks inside planner kernels; the first failing source line is kept in a device
// word and reported by gpu_build_plan (no printf, no trap - see hb_kernels.hip.h)
#ifdef HB_DEBUG_BOUNDS
__device__ unsigned int g_plan_dbg_line = 0;
#define PL_DBG_ASSERT(cond)                                              \
    do {                                                                 \
        if (!(cond)) atomicCAS(&g_plan_dbg_line, 0u, (unsigned int)__LINE__); \
    } while (0)
#else
#define PL_DBG_ASSERT(cond) ((void)0)
#endif

#define PL_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) return std::string(#call) + ": " + hipGetErrorString(e_); \
    } while (0)

// Work memory of the planner.  Every temporary is carved out of ONE device slab by a small first-fit heap:
// hipMalloc / hipFree of multi-GB buffers are cheap in isolation, but the planner's churn of them (tens of GB
// allocated and freed between kernels) cost 3.5 s of host time at 2 B edges against 0.4 s of kernels
// (profiles/r02j_C4_load_kernel_stats.csv) - freed memory seems to be handed back lazily, on later calls.
// Buffers that outlive the planner (the plan itself) are separate hipMallocs (alloc_out).
struct DevMem {
    struct Block {
        size_t off, size;
        bool free;
    };
    char *slab = nullptr;
    size_t slab_bytes = 0;
    std::vector<Block> blocks;   // address-ordered, covers the slab
    std::vector<void *> extra;   // overflow: plain hipMallocs
    std::vector<void *> outputs; // alloc_out results not yet handed over
    ~DevMem()
    {
        for (void *p : extra)
            if (p) (void)hipFree(p);
        for (void *p : outputs)
            if (p) (void)hipFree(p);
        if (slab) (void)hipFree(slab);
    }
    hipError_t init(size_t bytes)
    {
        // Round 4: no slab.  Every temporary is an allocation of its own - from the caching device allocator (hb_pool.h) in
        // the shipped build, where the ingest's freed sort buffers come back as the planner's, and from the debug allocator in
        // the debug builds, where an overrun of one planner buffer must hit the guard / red zone behind it.  (The slab heap
        // below is what rounds 2-3 used against the runtime's allocation cost; it is kept for HB_PLAN_SLAB=1 A/B runs.)
        if (!std::getenv("HB_PLAN_SLAB")) {
            (void)bytes;
            return hipSuccess;
        }
        bytes = (bytes + 4095) & ~(size_t)4095;
        hipError_t e = hipMalloc((void **)&slab, bytes);
        if (e != hipSuccess) { // no slab: every alloc() falls back to hipMalloc
            (void)hipGetLastError();
            slab = nullptr;
            return hipSuccess;
        }
        slab_bytes = bytes;
        blocks.push_back({0, bytes, true});
        return hipSuccess;
    }
    template <typename T>
    hipError_t alloc(T **out, size_t count)
    {
#if defined(HB_GUARD_ALLOC) || defined(HB_EXACT_ALLOC)
        const size_t need = std::max<size_t>(count * sizeof(T), 16); // exact size: the guard sits right behind the last element
#else
        const size_t need = (std::max<size_t>(count * sizeof(T), 256) + 255) & ~(size_t)255;
#endif
        for (size_t i = 0; i < blocks.size(); i++) {
            if (!blocks[i].free || blocks[i].size < need) continue;
            if (blocks[i].size > need) {
                const Block rest{blocks[i].off + need, blocks[i].size - need, true};
                blocks[i].size = need;
                blocks.insert(blocks.begin() + (long)i + 1, rest);
            }
            blocks[i].free = false;
            *out = (T *)(slab + blocks[i].off);
            return hipSuccess;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) extra.push_back(p);
        *out = (T *)p;
        return e;
    }
    void release(void *p)
    {
        if (!p) return;
        if (slab && (char *)p >= slab && (char *)p < slab + slab_bytes) {
            const size_t off = (size_t)((char *)p - slab);
            for (size_t i = 0; i < blocks.size(); i++) {
                if (blocks[i].off != off) continue;
                blocks[i].free = true;
                if (i + 1 < blocks.size() && blocks[i + 1].free) {
                    blocks[i].size += blocks[i + 1].size;
                    blocks.erase(blocks.begin() + (long)i + 1);
                }
                if (i > 0 && blocks[i - 1].free) {
                    blocks[i - 1].size += blocks[i].size;
                    blocks.erase(blocks.begin() + (long)i);
                }
                return;
            }
            return;
        }
        for (auto &q : extra)
            if (q == p) {
                (void)hipFree(p);
                q = nullptr;
            }
    }
    template <typename T>
    hipError_t alloc_out(T **out, size_t count) // outlives the planner
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 256));
        if (e == hipSuccess) outputs.push_back(p);
        *out = (T *)p;
        return e;
    }
    void *disown(void *p) // the caller keeps an alloc_out buffer
    {
        for (auto &q : outputs)
            if (q == p) q = nullptr;
        return p;
    }
};

// one thread per item; a dispatch is limited to 2^32 - 1 work-items, so callers with more items than that must use
// grid-stride kernels and grid_strided()
unsigned grid_for(uint64_t count, unsigned per_block = 256) { return (unsigned)std::max<uint64_t>(1, (count + per_block - 1) / per_block); }
unsigned grid_strided(uint64_t count) { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((count + 255) / 256, 1u << 22)); }
int bits_for(uint64_t v) // number of bits needed to represent values < v
{
    int b = 1;
    while (b < 64 && (1ull << b) < v) b++;
    return b;
}

struct BandParams {
    uint32_t band_w;      // 0 = no banding
    uint32_t warm_slices; // kWarmSlices of hb_host.cpp
};
__device__ __forceinline__ uint32_t band_of(const BandParams bp, uint32_t idx)
{
    if (!bp.band_w || idx < bp.band_w) return 0;
    const uint64_t j = (uint64_t)idx / bp.band_w;
    if (j <= bp.warm_slices) return (uint32_t)j;
    const uint64_t q = j / (bp.warm_slices + 1u); // >= 1
    return bp.warm_slices + 1u + (uint32_t)(63 - __clzll((long long)q));
}

// ---- device order --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void order_keys_kernel(const uint32_t *outdeg, uint64_t n, uint64_t world, uint64_t *key, uint32_t *val)
{
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    key[s] = ((s % world) << 32) | (uint64_t)(0xFFFFFFFFu - outdeg[s]);
    val[s] = (uint32_t)s;
}
// sids of owner o occupy sorted positions [first(o), first(o + 1)); first(o) = number of sids with s % world < o
__device__ __forceinline__ uint64_t owner_first(uint64_t n, uint64_t world, uint64_t o)
{
    uint64_t f = 0;
    for (uint64_t j = 0; j < o; j++) f += (n > j) ? (n - j + world - 1) / world : 0;
    return f;
}
__global__ __launch_bounds__(256) void place_sorted_kernel(const uint64_t *key, const uint32_t *val, uint64_t n, uint64_t world, uint64_t slice,
                                                           uint32_t *order, uint32_t *dev_of)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t owner = key[i] >> 32;
    const uint64_t pos = owner * slice + (i - owner_first(n, world, owner));
    order[pos] = val[i];
    dev_of[val[i]] = (uint32_t)pos;
}
__global__ __launch_bounds__(256) void place_identity_kernel(uint64_t n, uint64_t world, uint64_t slice, uint32_t *order, uint32_t *dev_of)
{
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const uint64_t pos = (s % world) * slice + s / world;
    order[pos] = (uint32_t)s;
    dev_of[s] = (uint32_t)pos;
}

// ---- rows in device order, sources as hotness ranks ------------------------------------------------
__global__ __launch_bounds__(256) void row_start_kernel(const uint64_t *row_ptr, uint64_t n, uint32_t *rowid)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const uint64_t b = row_ptr[v];
    if (row_ptr[v + 1] > b) rowid[b] = (uint32_t)v;
}
// (grid-stride: one dispatch holds at most 2^32 - 1 work-items, graphs here have up to 5.3 G edges)
__global__ __launch_bounds__(256) void edge_keys_kernel(const uint32_t *rowid, const uint32_t *src, uint64_t m, const uint32_t *dev_of,
                                                        uint64_t slice, uint64_t world, uint64_t *key)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (uint64_t)gridDim.x * 256) {
        const uint64_t d = dev_of[rowid[e]];
        const uint64_t idx = dev_of[src[e]];
        const uint64_t hr = world == 1 ? idx : (idx % slice) * world + idx / slice; // hotness rank of a device position
        key[e] = (d << 32) | hr;
    }
}
__global__ __launch_bounds__(256) void low_half_kernel(const uint64_t *key, uint64_t m, uint32_t *out)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (uint64_t)gridDim.x * 256) out[e] = (uint32_t)key[e];
}
// per device row: in-degree, split flag, out-degree in device order
__global__ __launch_bounds__(256) void row_info_kernel(const uint32_t *order, uint64_t n_pad, const uint64_t *row_ptr, const uint32_t *outdeg,
                                                       uint32_t direct_max, uint64_t *deg, uint8_t *is_split, uint32_t *outdeg_dev,
                                                       unsigned long long *sums /* [0] direct edges, [1] rows with in-edges, [2] sum of out-degrees */)
{
    const uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long direct = 0, with_in = 0, od = 0;
    if (d < n_pad) {
        const uint32_t s = order[d];
        uint64_t g = 0;
        uint32_t o = 0;
        if (s != kNone) {
            g = row_ptr[s + 1] - row_ptr[s];
            o = outdeg[s];
        }
        deg[d] = g;
        const bool split = g > direct_max;
        is_split[d] = split ? 1 : 0;
        outdeg_dev[d] = o;
        if (!split) direct = g;
        with_in = g > 0;
        od = o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        direct += __shfl_down(direct, off);
        with_in += __shfl_down(with_in, off);
        od += __shfl_down(od, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (direct) atomicAdd(&sums[0], direct);
        if (with_in) atomicAdd(&sums[1], with_in);
        if (od) atomicAdd(&sums[2], od);
    }
}

// ---- chunk cutting: one wave per hub row (hb_host.cpp cut_row) ------------------------------------------
// FILL = false: count[h] = chunks of hub row h;  FILL = true: chunk k = first[h] + i gets (begin, length, key)
template <bool FILL>
__global__ __launch_bounds__(256) void cut_rows_kernel(const uint32_t *hub_rows, uint64_t H, const uint64_t *rp, const uint32_t *rs,
                                                       uint32_t chunk, uint32_t minc, BandParams bp, uint64_t *count, const uint64_t *first,
                                                       uint64_t *cbeg, uint32_t *clen, uint32_t *ckey)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t h = wave; h < H; h += nwaves) {
        const uint64_t d = hub_rows[h];
        const uint64_t b = rp[d], e = rp[d + 1];
        uint64_t i = b, cnt = 0;
        const uint64_t w0 = FILL ? first[h] : 0;
        while (i < e) {
            const uint32_t b0 = band_of(bp, rs[i]);
            const uint64_t lim = std::min<uint64_t>(e, i + chunk);
            // j = first position in [i + minc, lim) whose band differs from b0, else lim
            uint64_t j = lim;
            for (uint64_t base = i + minc; base < lim; base += 64) {
                const uint64_t jj = base + lane;
                const bool cut = jj < lim && band_of(bp, rs[jj]) != b0;
                const uint64_t bal = __ballot(cut);
                if (bal) {
                    j = base + (uint64_t)(__ffsll((long long)bal) - 1);
                    break;
                }
            }
            if (e - j < minc && e - i <= chunk) j = e; // do not leave a tiny remainder behind
            if (FILL && lane == 0) {
                cbeg[w0 + cnt] = i;
                clen[w0 + cnt] = (uint32_t)(j - i);
                ckey[w0 + cnt] = b0;
            }
            cnt++;
            i = j;
        }
        if (!FILL && lane == 0) count[h] = cnt;
    }
}

// ---- XCD groups ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort1_keys_kernel(const uint32_t *clen, const uint32_t *ckey, uint64_t C, uint32_t *key, uint32_t *val)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= C) return;
    key[k] = (ckey[k] << 13) | (4096u - clen[k]); // slice ascending, longer chunks first (chunk <= 4096)
    val[k] = (uint32_t)k;
}
// sums[x] (x < 8) warm load of group x, sums[8] hot load, sums[9] cold load
__global__ __launch_bounds__(256) void quota_sums_kernel(const uint32_t *clen, const uint32_t *ckey, uint64_t C, uint32_t warm_slices,
                                                         unsigned long long *sums)
{
    __shared__ unsigned long long s_acc[10];
    if (threadIdx.x < 10) s_acc[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < C; k += (uint64_t)gridDim.x * 256) {
        const uint32_t key = ckey[k];
        const unsigned long long load = hb::XcdQuota::load_of(clen[k]);
        const int slot = (key >= 1 && key <= warm_slices) ? (int)(key & 7u) : (key == 0 ? 8 : 9);
        atomicAdd(&s_acc[slot], load);
    }
    __syncthreads();
    if (threadIdx.x < 10 && s_acc[threadIdx.x]) atomicAdd(&sums[threadIdx.x], s_acc[threadIdx.x]);
}
// in sort-1 order: the load a flexible chunk adds to its class prefix (warm chunks add nothing)
__global__ __launch_bounds__(256) void flex_load_kernel(const uint32_t *corder, const uint32_t *clen, const uint32_t *ckey, uint64_t C,
                                                        uint32_t warm_slices, uint64_t *load)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= C) return;
    const uint32_t k = corder[r];
    const uint32_t key = ckey[k];
    load[r] = (key >= 1 && key <= warm_slices) ? 0ull : hb::XcdQuota::load_of(clen[k]);
}
struct QuotaBounds {
    uint64_t bound[2][8];
    uint64_t stripe[2];
    uint64_t hot_total;
};
__global__ __launch_bounds__(256) void assign_groups_kernel(const uint32_t *corder, const uint32_t *ckey, const uint64_t *prefix, uint64_t C,
                                                            uint32_t warm_slices, QuotaBounds qb, uint32_t *grp_sorted, uint8_t *grp_of_chunk,
                                                            unsigned int *group_count)
{
    __shared__ unsigned int s_cnt[8];
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < C) {
        const uint32_t k = corder[r];
        const uint32_t key = ckey[k];
        uint32_t gsel;
        if (key >= 1 && key <= warm_slices) {
            gsel = key & 7u;
        } else {
            const int cls = key == 0 ? 0 : 1;
            // all hot chunks (slice 0) precede every cold one in sort-1 order; warm chunks add no load
            const uint64_t pre = cls == 0 ? prefix[r] : prefix[r] - qb.hot_total;
            const uint64_t scaled = (pre % qb.stripe[cls]) * hb::XcdQuota::kStripes; // XcdQuota::group_of
            gsel = 0;
#pragma unroll
            for (int x = 1; x < 8; x++)
                if (qb.bound[cls][x] <= scaled) gsel = (uint32_t)x;
        }
        grp_sorted[r] = gsel;
        grp_of_chunk[k] = (uint8_t)gsel;
        atomicAdd(&s_cnt[gsel], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 8 && s_cnt[threadIdx.x]) atomicAdd(&group_count[threadIdx.x], s_cnt[threadIdx.x]);
}
struct GroupLayout {
    uint64_t row_first[8];  // first virtual row id of group x
    uint64_t sort_first[8]; // first position of group x in the final (group-sorted) chunk order
};
// final order position j -> level-1 row; vid_of[chunk], row_chunk[row - first_vid], vlen[row - first_vid]
__global__ __launch_bounds__(256) void level1_rows_kernel(const uint32_t *forder, const uint8_t *grp_of_chunk, const uint32_t *clen, uint64_t C,
                                                          GroupLayout gl, uint64_t first_vid, uint32_t *vid_of, uint32_t *row_chunk,
                                                          uint32_t *vlen)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= C) return;
    const uint32_t k = forder[j];
    const int x = grp_of_chunk[k];
    const uint64_t row = gl.row_first[x] + (j - gl.sort_first[x]);
    vid_of[k] = (uint32_t)row;
    row_chunk[row - first_vid] = k;
    vlen[row - first_vid] = clen[k];
}
// quad per level-1 row: its sources = the chunk's hotness ranks mapped back to device positions
__global__ __launch_bounds__(256) void level1_fill_kernel(const uint32_t *row_chunk, uint64_t rows, const uint64_t *vrow_ptr, const uint64_t *cbeg,
                                                          const uint32_t *clen, const uint32_t *rs, uint64_t slice, uint64_t world, uint32_t *vsrc,
                                                          uint64_t m, uint64_t C, uint64_t vsrc_cap)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = t >> 2;
    const int q = (int)(t & 3);
    if (r >= rows) return;
    const uint32_t k = row_chunk[r];
    if (k == kNone) return;
    PL_DBG_ASSERT(k < C);
    const uint64_t o = vrow_ptr[r], b = cbeg[k];
    const uint32_t len = clen[k];
    PL_DBG_ASSERT(b + len <= m);
    PL_DBG_ASSERT(o + len <= vsrc_cap);
    (void)m;
    (void)C;
    (void)vsrc_cap;
    for (uint32_t i = q; i < len; i += 4) {
        const uint64_t hr = rs[b + i];
        vsrc[o + i] = (uint32_t)(world == 1 ? hr : (hr % world) * slice + hr / world);
    }
}

// ---- upper levels ---------------------------------------------------------------------------------------
// per hub row: parts = rows of the next level it needs (0 = its list already fits), newcnt = length of its list after
__global__ __launch_bounds__(256) void level_parts_kernel(const uint64_t *lptr, uint64_t H, uint32_t chunk, uint64_t *parts, uint64_t *newcnt,
                                                          uint64_t *moved)
{
    const uint64_t h = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const uint64_t cnt = lptr[h + 1] - lptr[h];
    const uint64_t p = cnt > chunk ? (cnt + chunk - 1) / chunk : 0;
    parts[h] = p;
    newcnt[h] = p ? p : cnt;
    moved[h] = p ? cnt : 0;
}
// one wave per hub row: rows that need grouping emit `parts` new virtual rows over consecutive slices of their id
// list (ids copied to the level's source region) and list the new rows instead; others keep their list
__global__ __launch_bounds__(256) void level_emit_kernel(const uint64_t *lptr, const uint32_t *lids, uint64_t H, const uint64_t *parts,
                                                         const uint64_t *vidoff, const uint64_t *noff, const uint64_t *srcoff, uint64_t level_first,
                                                         uint64_t first_vid, uint32_t *vlen, uint32_t *vsrc_level, uint32_t *nids)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t h = wave; h < H; h += nwaves) {
        const uint64_t b = lptr[h], e = lptr[h + 1], cnt = e - b;
        const uint64_t p = parts[h];
        if (!p) {
            for (uint64_t i = lane; i < cnt; i += 64) nids[noff[h] + i] = lids[b + i];
            continue;
        }
        const uint64_t per = (cnt + p - 1) / p;
        for (uint64_t i = lane; i < cnt; i += 64) vsrc_level[srcoff[h] + i] = lids[b + i];
        for (uint64_t k = lane; k < p; k += 64) {
            const uint64_t pb = k * per, pe = std::min<uint64_t>(cnt, pb + per);
            const uint64_t row = level_first + vidoff[h] + k;
            vlen[row - first_vid] = (uint32_t)(pe - pb);
            nids[noff[h] + k] = (uint32_t)row;
        }
    }
}

// ---- assembly ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hub_index_kernel(const uint32_t *hub_rows, uint64_t H, uint32_t *hub_index)
{
    const uint64_t h = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (h < H) hub_index[hub_rows[h]] = (uint32_t)h;
}
__global__ __launch_bounds__(256) void real_len_kernel(const uint64_t *deg, const uint8_t *is_split, const uint32_t *hub_index, const uint64_t *lptr,
                                                       uint64_t n_pad, uint64_t *len)
{
    const uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_pad) return;
    if (is_split[d]) {
        const uint32_t h = hub_index[d];
        len[d] = lptr[h + 1] - lptr[h];
    } else {
        len[d] = deg[d];
    }
}
// quad per node row
__global__ __launch_bounds__(256) void real_fill_kernel(const uint64_t *row_ptr, const uint8_t *is_split, const uint32_t *hub_index, const uint64_t *lptr,
                                                        const uint32_t *lids, const uint64_t *rp, const uint32_t *rs, uint64_t n_pad, uint64_t slice,
                                                        uint64_t world, uint32_t *src)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t d = t >> 2;
    const int q = (int)(t & 3);
    if (d >= n_pad) return;
    const uint64_t o = row_ptr[d], len = row_ptr[d + 1] - o;
    if (is_split[d]) {
        const uint64_t b = lptr[hub_index[d]];
        for (uint64_t i = q; i < len; i += 4) src[o + i] = lids[b + i];
    } else {
        const uint64_t b = rp[d];
        for (uint64_t i = q; i < len; i += 4) {
            const uint64_t hr = rs[b + i];
            src[o + i] = (uint32_t)(world == 1 ? hr : (hr % world) * slice + hr / world);
        }
    }
}
__global__ __launch_bounds__(256) void virtual_row_ptr_kernel(const uint64_t *vrow_ptr, uint64_t nv, uint64_t real_total, uint64_t *row_ptr_tail)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k <= nv) row_ptr_tail[k] = real_total + vrow_ptr[k]; // row_ptr[n_pad + k]
}

struct Widen32 {
    __device__ uint64_t operator()(uint32_t v) const { return v; }
};

} // namespace

namespace hb {

// offsets[i] = sum of counts[0 .. i) for i = 0 .. count (count + 1 entries); the total must equal `expect`
std::string device_offsets(void *stream_v, uint32_t *d_counts /* count + 1 entries, the last one is overwritten with 0 */, uint64_t count,
                           uint64_t *d_offsets, uint64_t expect)
{
    hipStream_t stream = (hipStream_t)stream_v;
    // the scan runs over count + 1 inputs so that the last output is the total: d_counts must have room for one more
    auto wide = rocprim::make_transform_iterator(d_counts, Widen32());
    size_t bytes = 0;
    void *tmp = nullptr;
    PL_HIP(rocprim::exclusive_scan(nullptr, bytes, wide, d_offsets, (uint64_t)0, (size_t)count, rocprim::plus<uint64_t>(), stream));
    PL_HIP(hipMalloc(&tmp, std::max<size_t>(bytes, 256)));
    hipError_t e = rocprim::exclusive_scan(tmp, bytes, wide, d_offsets, (uint64_t)0, (size_t)count, rocprim::plus<uint64_t>(), stream);
    uint64_t last_off = 0;
    uint32_t last_cnt = 0;
    if (e == hipSuccess && count) e = hipMemcpyAsync(&last_off, d_offsets + (count - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(&last_cnt, d_counts + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return std::string("device_offsets: ") + hipGetErrorString(e);
    const uint64_t total = last_off + last_cnt;
    if (total != expect) return "transposed plan graph: entry count mismatch";
    e = hipMemcpyAsync(d_offsets + count, &total, sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return std::string("device_offsets: ") + hipGetErrorString(e);
    return "";
}

// d_offsets[i] = sum of d_counts[0 .. i), i = 0 .. count (no check of the total)
std::string device_prefix(void *stream_v, const uint32_t *d_counts, uint64_t count, uint64_t *d_offsets)
{
    hipStream_t stream = (hipStream_t)stream_v;
    // count + 1 outputs: the scan runs over count + 1 inputs whose last one is never read as a value that matters
    auto wide = rocprim::make_transform_iterator(d_counts, Widen32());
    size_t bytes = 0;
    void *tmp = nullptr;
    if (!count) {
        PL_HIP(hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), stream));
        return "";
    }
    PL_HIP(rocprim::inclusive_scan(nullptr, bytes, wide, d_offsets + 1, (size_t)count, rocprim::plus<uint64_t>(), stream));
    PL_HIP(hipMalloc(&tmp, std::max<size_t>(bytes, 256)));
    hipError_t e = hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), stream);
    if (e == hipSuccess) e = rocprim::inclusive_scan(tmp, bytes, wide, d_offsets + 1, (size_t)count, rocprim::plus<uint64_t>(), stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return std::string("device_prefix: ") + hipGetErrorString(e);
    return "";
}

// ---- the transposed work-row graph by SORTING [r6] ----------------------------------------------------------------------------------
// out_rows / out_ptr (who reads each node or virtual row: the sweep passes' expansion) used to be built by scattering: one atomic
// cursor bump and one 4-byte store at a random place per entry - 126 bytes of HBM traffic per entry at C4 (266 GB for 2.1 G entries,
// L2 hit rate 8.7 %: profiles/r05c_C4_pmc.json), plus a counting pass of atomics before it.  Here: one 64-bit key (source << 32 | row)
// per entry, written row by row (so equal sources keep ascending row order), ONE stable radix sort on the source bits with two
// ping-pong buffers, the low halves are the reader lists and the run ends - max-scanned - the offsets.  Streaming traffic only:
// 8 B written + 4 passes x 16 B + 12 B read back per entry.  Needs 16 bytes per entry of work memory; when the device cannot give
// that (C5 on a full device) the caller falls back to the scatter form (same lists up to the order inside a list, which no kernel
// depends on).
static __global__ __launch_bounds__(256) void transpose_keys_kernel(const uint64_t *row_ptr, const uint32_t *src, uint64_t rows, uint64_t *keys)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nq = (uint64_t)gridDim.x * 64; // quads in the grid
    const int q = threadIdx.x & 3;
    for (uint64_t row = t >> 2; row < rows; row += nq) {
        const uint64_t b = row_ptr[row], e = row_ptr[row + 1];
        for (uint64_t k = b + q; k < e; k += 4) keys[k] = ((uint64_t)src[k] << 32) | row;
    }
}
static __global__ __launch_bounds__(256) void transpose_emit_kernel(const uint64_t *keys, uint64_t entries, uint32_t *out_rows, uint64_t *ends /* out_ptr + 1 */)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < entries; e += (uint64_t)gridDim.x * 256) {
        const uint64_t k = keys[e];
        out_rows[e] = (uint32_t)k;
        const uint64_t s = k >> 32;
        if (e + 1 == entries || (keys[e + 1] >> 32) != s) ends[s] = e + 1; // the last reader of s: its list ends here
    }
}
std::string gpu_transpose_rows(void *stream_v, const uint64_t *d_row_ptr, const uint32_t *d_src, uint64_t rows_total, uint64_t entries, uint64_t *d_out_ptr,
                               uint32_t *d_out_rows)
{
    hipStream_t stream = (hipStream_t)stream_v;
    PL_HIP(hipMemsetAsync(d_out_ptr, 0, (rows_total + 1) * sizeof(uint64_t), stream));
    if (!entries || !rows_total) {
        PL_HIP(hipStreamSynchronize(stream));
        return "";
    }
    struct Work {
        uint64_t *a = nullptr, *b = nullptr;
        void *tmp = nullptr;
        ~Work()
        {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            if (tmp) (void)hipFree(tmp);
        }
    } w;
    if (hipMalloc((void **)&w.a, entries * sizeof(uint64_t)) != hipSuccess || hipMalloc((void **)&w.b, entries * sizeof(uint64_t)) != hipSuccess) {
        (void)hipGetLastError();
        return "out of memory for the sort-based transposition";
    }
    hipLaunchKernelGGL(transpose_keys_kernel, dim3((unsigned)std::min<uint64_t>((rows_total * 4 + 255) / 256, 1u << 16)), dim3(256), 0, stream, d_row_ptr, d_src, rows_total, w.a);
    PL_HIP(hipGetLastError());
    const unsigned end_bit = 32 + (unsigned)bits_for(std::max<uint64_t>(rows_total, 2));
    rocprim::double_buffer<uint64_t> keys(w.a, w.b);
    size_t bytes = 0;
    PL_HIP(rocprim::radix_sort_keys(nullptr, bytes, keys, (size_t)entries, 32u, end_bit, stream));
    if (hipMalloc(&w.tmp, std::max<size_t>(bytes, 256)) != hipSuccess) {
        (void)hipGetLastError();
        return "out of memory for the sort-based transposition";
    }
    PL_HIP(rocprim::radix_sort_keys(w.tmp, bytes, keys, (size_t)entries, 32u, end_bit, stream));
    hipLaunchKernelGGL(transpose_emit_kernel, dim3(grid_strided(entries)), dim3(256), 0, stream, (const uint64_t *)keys.current(), entries, d_out_rows, d_out_ptr + 1);
    PL_HIP(hipGetLastError());
    // out_ptr[s + 1] = end of the last non-empty list at or below s: an inclusive maximum scan of the ends (a source nobody reads keeps 0)
    (void)hipFree(w.tmp);
    w.tmp = nullptr;
    bytes = 0;
    PL_HIP(rocprim::inclusive_scan(nullptr, bytes, d_out_ptr + 1, d_out_ptr + 1, (size_t)rows_total, rocprim::maximum<uint64_t>(), stream));
    if (hipMalloc(&w.tmp, std::max<size_t>(bytes, 256)) != hipSuccess) {
        (void)hipGetLastError();
        return "out of memory for the sort-based transposition";
    }
    PL_HIP(rocprim::inclusive_scan(w.tmp, bytes, d_out_ptr + 1, d_out_ptr + 1, (size_t)rows_total, rocprim::maximum<uint64_t>(), stream));
    PL_HIP(hipStreamSynchronize(stream));
    return "";
}

// ---- destination partition: keep the in-edges of the owned rows only (the device form of hb_host.cpp keep_owned_rows) -----
static __global__ __launch_bounds__(256) void owned_counts_kernel(const uint64_t *row_ptr, uint64_t n, uint64_t world, uint64_t rank, uint32_t *cnt)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < n) cnt[v] = (v % world == rank) ? (uint32_t)(row_ptr[v + 1] - row_ptr[v]) : 0u;
}
// one thread per kept edge: its row by binary search in the new row pointers, its source from the old list
static __global__ __launch_bounds__(256) void owned_copy_kernel(const uint64_t *old_ptr, const uint32_t *old_src, const uint64_t *new_ptr, uint64_t n, uint64_t m_new,
                                                          uint32_t *new_src)
{
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m_new; j += (uint64_t)gridDim.x * 256) {
        uint64_t lo = 0, hi = n; // last row with new_ptr[row] <= j
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (new_ptr[mid] <= j) lo = mid;
            else hi = mid;
        }
        new_src[j] = old_src[old_ptr[lo] + (j - new_ptr[lo])];
    }
}
std::string gpu_keep_owned_rows(void *stream_v, DeviceCsr *csr, uint64_t n, uint64_t world, uint64_t rank)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (world <= 1 || n == 0 || !csr->d_row_ptr) return "";
    uint32_t *d_cnt = nullptr;
    uint64_t *d_new_ptr = nullptr;
    uint32_t *d_new_src = nullptr;
    auto bail = [&](const std::string &m) {
        for (void *q : {(void *)d_cnt, (void *)d_new_ptr, (void *)d_new_src})
            if (q) (void)hipFree(q);
        return m;
    };
    if (hipMalloc((void **)&d_cnt, (n + 1) * sizeof(uint32_t)) != hipSuccess || hipMalloc((void **)&d_new_ptr, (n + 1) * sizeof(uint64_t)) != hipSuccess)
        return bail("gpu_keep_owned_rows: out of device memory");
    hipLaunchKernelGGL(owned_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint64_t *)csr->d_row_ptr, n, world, rank, d_cnt);
    if (hipGetLastError() != hipSuccess) return bail("gpu_keep_owned_rows: launch failed");
    std::string e = device_prefix(stream_v, d_cnt, n, d_new_ptr);
    if (!e.empty()) return bail(e);
    uint64_t m_new = 0;
    if (hipMemcpyAsync(&m_new, d_new_ptr + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return bail("gpu_keep_owned_rows: read-back failed");
    if (m_new == csr->m) return bail(""); // the caller handed over the owned rows only: nothing to drop
    if (hipMalloc((void **)&d_new_src, std::max<uint64_t>(m_new, 1) * sizeof(uint32_t)) != hipSuccess) return bail("gpu_keep_owned_rows: out of device memory");
    if (m_new) {
        const unsigned blocks = (unsigned)std::min<uint64_t>((m_new + 255) / 256, 65536);
        hipLaunchKernelGGL(owned_copy_kernel, dim3(blocks), dim3(256), 0, stream, (const uint64_t *)csr->d_row_ptr, (const uint32_t *)csr->d_src,
                           (const uint64_t *)d_new_ptr, n, m_new, d_new_src);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return bail("gpu_keep_owned_rows: copy failed");
    }
    (void)hipFree(csr->d_row_ptr);
    (void)hipFree(csr->d_src);
    (void)hipFree(d_cnt);
    csr->d_row_ptr = d_new_ptr;
    csr->d_src = d_new_src;
    csr->m = m_new;
    return "";
}

std::string gpu_build_plan(void *stream_v, uint64_t n, const uint64_t *d_row_ptr_in, const uint32_t *d_src_in, const uint32_t *d_outdeg_sid,
                           bool reorder, const PlanTune &tune_in, Plan *p, DevicePlan *out)
{
    hipStream_t stream = (hipStream_t)stream_v;
    PlanTune tune = tune_in;
    uint32_t chunk = tune.chunk ? tune.chunk : kDefaultChunk;
    if (chunk < 4) chunk = 4;
    if (chunk > 4096) chunk = 4096;
    if (tune.direct_max == 0 || tune.direct_max > chunk) tune.direct_max = chunk;
    if (tune.minc == 0) tune.minc = 8;
    const uint32_t minc = std::max<uint32_t>(1, std::min(tune.minc, chunk));
    const uint64_t world = tune.world > 1 ? tune.world : 1;
    const uint64_t slice = ((n + world - 1) / world + kRowAlign - 1) / kRowAlign * kRowAlign;
    const uint64_t n_pad = slice * world;
    *out = DevicePlan{};
    p->n = n;
    p->slice = slice;
    p->n_pad = n_pad;
    p->chunk = chunk;
    p->nv = 0;
    p->order.clear();
    p->dev_of.clear();
    p->level_begin.clear();
    decltype(p->row_ptr)().swap(p->row_ptr);
    decltype(p->src)().swap(p->src);
    uint64_t m = 0;
    if (n) PL_HIP(hipMemcpyAsync(&m, d_row_ptr_in + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    PL_HIP(hipStreamSynchronize(stream));
    p->m_eff = m;
    const bool timing = std::getenv("HB_PLAN_TIMING") != nullptr;
    double tmark = now_ms();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(stream);
        const double t = now_ms();
        std::fprintf(stderr, "[gpu plan] %-28s %8.1f ms\n", what, t - tmark);
        tmark = t;
    };

    DevMem mem;
    // peak of the live temporaries: the edge sort (hotness ranks 4m + row ids 4m + two 8m key buffers + its scratch);
    // everything later (chunk / level arrays, the level-1 lists) fits into what the sort leaves behind
    PL_HIP(mem.init(24 * m + m / 2 + 96 * n_pad + (512ull << 20)));
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    auto need_tmp = [&](size_t bytes) -> hipError_t {
        if (bytes <= tmp_bytes) return hipSuccess;
        if (tmp) mem.release(tmp);
        tmp_bytes = bytes + (bytes >> 3) + 256;
        char *q = nullptr;
        hipError_t e = mem.alloc(&q, tmp_bytes);
        tmp = q;
        return e;
    };

    // ---- device order ------------------------------------------------------------------------------
    uint32_t *d_order = nullptr, *d_dev_of = nullptr, *d_outdeg_dev = nullptr;
    PL_HIP(mem.alloc_out(&d_order, n_pad));
    PL_HIP(mem.alloc_out(&d_dev_of, n));
    PL_HIP(mem.alloc_out(&d_outdeg_dev, n_pad));
    PL_HIP(hipMemsetAsync(d_order, 0xFF, std::max<uint64_t>(n_pad, 1) * sizeof(uint32_t), stream));
    if (n) {
        if (reorder) {
            uint64_t *k_in = nullptr, *k_out = nullptr;
            uint32_t *v_in = nullptr, *v_out = nullptr;
            PL_HIP(mem.alloc(&k_in, n));
            PL_HIP(mem.alloc(&k_out, n));
            PL_HIP(mem.alloc(&v_in, n));
            PL_HIP(mem.alloc(&v_out, n));
            hipLaunchKernelGGL(order_keys_kernel, dim3(grid_for(n)), dim3(256), 0, stream, d_outdeg_sid, n, world, k_in, v_in);
            PL_HIP(hipGetLastError());
            size_t bytes = 0;
            const unsigned end_bit = 32 + (unsigned)bits_for(world);
            PL_HIP(rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, end_bit, stream));
            PL_HIP(need_tmp(bytes));
            PL_HIP(rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, end_bit, stream));
            hipLaunchKernelGGL(place_sorted_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const uint64_t *)k_out, (const uint32_t *)v_out, n,
                               world, slice, d_order, d_dev_of);
            PL_HIP(hipGetLastError());
            PL_HIP(hipStreamSynchronize(stream));
            mem.release(k_in);
            mem.release(k_out);
            mem.release(v_in);
            mem.release(v_out);
        } else {
            hipLaunchKernelGGL(place_identity_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, world, slice, d_order, d_dev_of);
            PL_HIP(hipGetLastError());
        }
    }
    lap("device order");

    // ---- rows in device order; sources relabelled to hotness ranks, ascending ------------------------
    uint32_t *d_rs = nullptr;
    PL_HIP(mem.alloc(&d_rs, m));
    if (m) {
        uint32_t *d_rowid = nullptr;
        uint64_t *k_in = nullptr, *k_out = nullptr;
        PL_HIP(mem.alloc(&d_rowid, m));
        PL_HIP(hipMemsetAsync(d_rowid, 0, m * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(row_start_kernel, dim3(grid_for(n)), dim3(256), 0, stream, d_row_ptr_in, n, d_rowid);
        PL_HIP(hipGetLastError());
        size_t bytes = 0;
        PL_HIP(rocprim::inclusive_scan(nullptr, bytes, d_rowid, d_rowid, (size_t)m, rocprim::maximum<uint32_t>(), stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::inclusive_scan(tmp, bytes, d_rowid, d_rowid, (size_t)m, rocprim::maximum<uint32_t>(), stream));
        lap("  rows: row id per edge");
        PL_HIP(mem.alloc(&k_in, m));
        hipLaunchKernelGGL(edge_keys_kernel, dim3(grid_strided(m)), dim3(256), 0, stream, (const uint32_t *)d_rowid, d_src_in, m,
                           (const uint32_t *)d_dev_of, slice, world, k_in);
        PL_HIP(hipGetLastError());
        PL_HIP(hipStreamSynchronize(stream));
        mem.release(d_rowid);
        lap("  rows: keys");
        PL_HIP(mem.alloc(&k_out, m));
        const unsigned end_bit = 32 + (unsigned)bits_for(std::max<uint64_t>(n_pad, 2));
        bytes = 0;
        rocprim::double_buffer<uint64_t> keys(k_in, k_out); // ping-pong between the two buffers: no third copy as scratch
        PL_HIP(rocprim::radix_sort_keys(nullptr, bytes, keys, (size_t)m, 0u, end_bit, stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::radix_sort_keys(tmp, bytes, keys, (size_t)m, 0u, end_bit, stream));
        lap("  rows: 64-bit radix sort");
        hipLaunchKernelGGL(low_half_kernel, dim3(grid_strided(m)), dim3(256), 0, stream, (const uint64_t *)keys.current(), m, d_rs);
        PL_HIP(hipGetLastError());
        PL_HIP(hipStreamSynchronize(stream));
        mem.release(k_in);
        mem.release(k_out);
        if (tmp_bytes > (64u << 20)) { // the sort scratch is as big as the keys
            mem.release(tmp);
            tmp = nullptr;
            tmp_bytes = 0;
        }
    }
    uint64_t *d_deg = nullptr, *d_rp = nullptr;
    uint8_t *d_is_split = nullptr;
    unsigned long long *d_sums = nullptr; // small accumulators, reused
    PL_HIP(mem.alloc(&d_deg, n_pad + 1));
    PL_HIP(mem.alloc(&d_rp, n_pad + 1));
    PL_HIP(mem.alloc(&d_is_split, n_pad + 1));
    PL_HIP(mem.alloc(&d_sums, 32));
    PL_HIP(hipMemsetAsync(d_sums, 0, 32 * sizeof(unsigned long long), stream));
    PL_HIP(hipMemsetAsync(d_deg, 0, (n_pad + 1) * sizeof(uint64_t), stream));
    if (n_pad) {
        hipLaunchKernelGGL(row_info_kernel, dim3(grid_for(n_pad)), dim3(256), 0, stream, (const uint32_t *)d_order, n_pad, d_row_ptr_in,
                           d_outdeg_sid, tune.direct_max, d_deg, d_is_split, d_outdeg_dev, d_sums);
        PL_HIP(hipGetLastError());
    }
    {
        size_t bytes = 0;
        PL_HIP(rocprim::exclusive_scan(nullptr, bytes, d_deg, d_rp, (uint64_t)0, (size_t)(n_pad + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::exclusive_scan(tmp, bytes, d_deg, d_rp, (uint64_t)0, (size_t)(n_pad + 1), rocprim::plus<uint64_t>(), stream));
    }
    unsigned long long h_sums[32];
    PL_HIP(hipMemcpyAsync(h_sums, d_sums, sizeof(h_sums), hipMemcpyDeviceToHost, stream));
    PL_HIP(hipStreamSynchronize(stream));
    p->direct_edges = h_sums[0];
    p->rows_with_in_edges = h_sums[1];
    out->m_global = h_sums[2];
    lap("relabel + sort rows");

    // ---- hub rows and their level-1 chunks ------------------------------------------------------------
    const uint32_t warm_slices = tune.xcd_map ? 64u : 0u;
    const BandParams bp{tune.band_w, warm_slices};
    uint32_t *d_hub_rows = nullptr;
    uint64_t *d_cnt = nullptr;
    PL_HIP(mem.alloc(&d_hub_rows, n_pad + 1));
    PL_HIP(mem.alloc(&d_cnt, 2));
    uint64_t H = 0;
    if (n_pad) {
        auto iota = rocprim::make_counting_iterator<uint32_t>(0);
        size_t bytes = 0;
        PL_HIP(rocprim::select(nullptr, bytes, iota, d_is_split, d_hub_rows, d_cnt, (size_t)n_pad, stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::select(tmp, bytes, iota, d_is_split, d_hub_rows, d_cnt, (size_t)n_pad, stream));
        PL_HIP(hipMemcpyAsync(&H, d_cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        PL_HIP(hipStreamSynchronize(stream));
    }
    uint64_t *d_hub_cnt = nullptr, *d_hub_first = nullptr;
    PL_HIP(mem.alloc(&d_hub_cnt, H + 1));
    PL_HIP(mem.alloc(&d_hub_first, H + 1));
    PL_HIP(hipMemsetAsync(d_hub_cnt, 0, (H + 1) * sizeof(uint64_t), stream));
    const unsigned cut_blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((H + 3) / 4, 4096));
    uint64_t C = 0;
    if (H) {
        hipLaunchKernelGGL(cut_rows_kernel<false>, dim3(cut_blocks), dim3(256), 0, stream, (const uint32_t *)d_hub_rows, H, (const uint64_t *)d_rp,
                           (const uint32_t *)d_rs, chunk, minc, bp, d_hub_cnt, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                           (uint32_t *)nullptr);
        PL_HIP(hipGetLastError());
    }
    {
        size_t bytes = 0;
        PL_HIP(rocprim::exclusive_scan(nullptr, bytes, d_hub_cnt, d_hub_first, (uint64_t)0, (size_t)(H + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::exclusive_scan(tmp, bytes, d_hub_cnt, d_hub_first, (uint64_t)0, (size_t)(H + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(hipMemcpyAsync(&C, d_hub_first + H, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        PL_HIP(hipStreamSynchronize(stream));
    }
    if (n_pad + C + C / 8 + 64ull * kRowAlign >= (uint64_t)kNone) return "row id space exhausted (n + virtual rows >= 2^32 - 1)";
    uint64_t *d_cbeg = nullptr;
    uint32_t *d_clen = nullptr, *d_ckey = nullptr;
    PL_HIP(mem.alloc(&d_cbeg, C + 1));
    PL_HIP(mem.alloc(&d_clen, C + 1));
    PL_HIP(mem.alloc(&d_ckey, C + 1));
    if (H) {
        hipLaunchKernelGGL(cut_rows_kernel<true>, dim3(cut_blocks), dim3(256), 0, stream, (const uint32_t *)d_hub_rows, H, (const uint64_t *)d_rp,
                           (const uint32_t *)d_rs, chunk, minc, bp, (uint64_t *)nullptr, (const uint64_t *)d_hub_first, d_cbeg, d_clen, d_ckey);
        PL_HIP(hipGetLastError());
    }
    lap("cut chunks");

    // ---- order of the level-1 rows: slice ascending / longer first, then XCD groups ---------------------
    const int groups = tune.xcd_map ? 8 : 1;
    uint32_t *d_forder = nullptr; // final order position -> chunk
    uint8_t *d_grp_of = nullptr;
    PL_HIP(mem.alloc(&d_forder, C + 1));
    PL_HIP(mem.alloc(&d_grp_of, C + 1));
    PL_HIP(hipMemsetAsync(d_grp_of, 0, C + 1, stream));
    unsigned int h_group_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    h_group_count[0] = (unsigned int)C;
    if (C) {
        uint32_t *k1 = nullptr, *k1s = nullptr, *v1 = nullptr, *corder = nullptr;
        PL_HIP(mem.alloc(&k1, C));
        PL_HIP(mem.alloc(&k1s, C));
        PL_HIP(mem.alloc(&v1, C));
        PL_HIP(mem.alloc(&corder, C));
        hipLaunchKernelGGL(sort1_keys_kernel, dim3(grid_for(C)), dim3(256), 0, stream, (const uint32_t *)d_clen, (const uint32_t *)d_ckey, C, k1, v1);
        PL_HIP(hipGetLastError());
        size_t bytes = 0;
        PL_HIP(rocprim::radix_sort_pairs(nullptr, bytes, k1, k1s, v1, corder, (size_t)C, 0u, 32u, stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::radix_sort_pairs(tmp, bytes, k1, k1s, v1, corder, (size_t)C, 0u, 32u, stream));
        lap("  chunks: sort by slice/len");
        if (groups > 1) {
            PL_HIP(hipMemsetAsync(d_sums, 0, 32 * sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(quota_sums_kernel, dim3((unsigned)std::min<uint64_t>(grid_for(C), 2048)), dim3(256), 0, stream, (const uint32_t *)d_clen,
                               (const uint32_t *)d_ckey, C, warm_slices, d_sums);
            PL_HIP(hipGetLastError());
            PL_HIP(hipMemcpyAsync(h_sums, d_sums, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            PL_HIP(hipStreamSynchronize(stream));
            XcdQuota quota;
            for (int x = 0; x < 8; x++) quota.warm[x] = h_sums[x];
            quota.flex[0] = h_sums[8];
            quota.flex[1] = h_sums[9];
            quota.finish();
            lap("  chunks: quota sums");
            QuotaBounds qb;
            for (int c = 0; c < 2; c++)
                for (int x = 0; x < 8; x++) qb.bound[c][x] = quota.bound[c][x];
            qb.hot_total = quota.flex[0];
            qb.stripe[0] = quota.stripe(0);
            qb.stripe[1] = quota.stripe(1);
            uint64_t *d_load = nullptr, *d_prefix = nullptr;
            uint32_t *d_grp_sorted = nullptr, *d_grp_sorted2 = nullptr;
            unsigned int *d_gcount = nullptr;
            PL_HIP(mem.alloc(&d_load, C));
            PL_HIP(mem.alloc(&d_prefix, C));
            PL_HIP(mem.alloc(&d_grp_sorted, C));
            PL_HIP(mem.alloc(&d_grp_sorted2, C));
            PL_HIP(mem.alloc(&d_gcount, 8));
            PL_HIP(hipMemsetAsync(d_gcount, 0, 8 * sizeof(unsigned int), stream));
            hipLaunchKernelGGL(flex_load_kernel, dim3(grid_for(C)), dim3(256), 0, stream, (const uint32_t *)corder, (const uint32_t *)d_clen,
                               (const uint32_t *)d_ckey, C, warm_slices, d_load);
            PL_HIP(hipGetLastError());
            bytes = 0;
            PL_HIP(rocprim::exclusive_scan(nullptr, bytes, d_load, d_prefix, (uint64_t)0, (size_t)C, rocprim::plus<uint64_t>(), stream));
            PL_HIP(need_tmp(bytes));
            PL_HIP(rocprim::exclusive_scan(tmp, bytes, d_load, d_prefix, (uint64_t)0, (size_t)C, rocprim::plus<uint64_t>(), stream));
            hipLaunchKernelGGL(assign_groups_kernel, dim3(grid_for(C)), dim3(256), 0, stream, (const uint32_t *)corder, (const uint32_t *)d_ckey,
                               (const uint64_t *)d_prefix, C, warm_slices, qb, d_grp_sorted, d_grp_of, d_gcount);
            PL_HIP(hipGetLastError());
            lap("  chunks: prefix + groups");
            bytes = 0;
            PL_HIP(rocprim::radix_sort_pairs(nullptr, bytes, d_grp_sorted, d_grp_sorted2, corder, d_forder, (size_t)C, 0u, 3u, stream));
            PL_HIP(need_tmp(bytes));
            PL_HIP(rocprim::radix_sort_pairs(tmp, bytes, d_grp_sorted, d_grp_sorted2, corder, d_forder, (size_t)C, 0u, 3u, stream));
            PL_HIP(hipMemcpyAsync(h_group_count, d_gcount, sizeof(h_group_count), hipMemcpyDeviceToHost, stream));
            PL_HIP(hipStreamSynchronize(stream));
            mem.release(d_load);
            mem.release(d_prefix);
            mem.release(d_grp_sorted);
            mem.release(d_grp_sorted2);
            mem.release(d_gcount);
        } else {
            PL_HIP(hipMemcpyAsync(d_forder, corder, C * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            PL_HIP(hipStreamSynchronize(stream));
        }
        mem.release(k1);
        mem.release(k1s);
        mem.release(v1);
        mem.release(corder);
    }
    lap("sort chunks");

    // ---- level-1 rows: ids, lengths, lists ------------------------------------------------------------
    const uint64_t first_vid = n_pad;
    GroupLayout gl{};
    uint64_t next_vid = first_vid;
    p->level_begin.push_back(next_vid);
    {
        uint64_t sorted_pos = 0;
        for (int x = 0; x < 8; x++) {
            p->xcd_begin[x] = next_vid;
            gl.row_first[x] = next_vid;
            gl.sort_first[x] = sorted_pos;
            if (x < groups) {
                next_vid += h_group_count[x];
                sorted_pos += h_group_count[x];
                if (groups > 1) next_vid = (next_vid - first_vid + kRowAlign - 1) / kRowAlign * kRowAlign + first_vid;
            }
        }
        p->xcd_begin[8] = next_vid;
        p->xcd_groups = groups;
    }
    const uint64_t rows_l1 = next_vid - first_vid;
    // capacity of the virtual-row arrays: every upper level has at most 1/chunk of the rows below it (+ padding)
    const uint64_t vrow_cap = rows_l1 + rows_l1 / 2 + 64ull * kRowAlign;
    uint32_t *d_vid_of = nullptr, *d_row_chunk = nullptr, *d_vlen = nullptr;
    PL_HIP(mem.alloc(&d_vid_of, C + 1));
    PL_HIP(mem.alloc(&d_row_chunk, rows_l1 + 1));
    PL_HIP(mem.alloc(&d_vlen, vrow_cap + 1));
    PL_HIP(hipMemsetAsync(d_row_chunk, 0xFF, (rows_l1 + 1) * sizeof(uint32_t), stream));
    PL_HIP(hipMemsetAsync(d_vlen, 0, (vrow_cap + 1) * sizeof(uint32_t), stream));
    if (C) {
        hipLaunchKernelGGL(level1_rows_kernel, dim3(grid_for(C)), dim3(256), 0, stream, (const uint32_t *)d_forder, (const uint8_t *)d_grp_of,
                           (const uint32_t *)d_clen, C, gl, first_vid, d_vid_of, d_row_chunk, d_vlen);
        PL_HIP(hipGetLastError());
    }
    uint64_t *d_vrow_ptr = nullptr;
    PL_HIP(mem.alloc(&d_vrow_ptr, vrow_cap + 2));
    uint64_t l1_edges = 0;
    {
        auto wide = rocprim::make_transform_iterator(d_vlen, Widen32());
        size_t bytes = 0;
        PL_HIP(rocprim::exclusive_scan(nullptr, bytes, wide, d_vrow_ptr, (uint64_t)0, (size_t)(rows_l1 + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::exclusive_scan(tmp, bytes, wide, d_vrow_ptr, (uint64_t)0, (size_t)(rows_l1 + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(hipMemcpyAsync(&l1_edges, d_vrow_ptr + rows_l1, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        PL_HIP(hipStreamSynchronize(stream));
    }
    p->level1_edges = l1_edges;
    p->level1_rows = C;
    const uint64_t vsrc_cap = l1_edges + rows_l1 + rows_l1 / 8 + 64ull * kRowAlign;
    uint32_t *d_vsrc = nullptr;
    PL_HIP(mem.alloc(&d_vsrc, vsrc_cap + 1));
    if (rows_l1) {
        hipLaunchKernelGGL(level1_fill_kernel, dim3(grid_for(rows_l1 * 4)), dim3(256), 0, stream, (const uint32_t *)d_row_chunk, rows_l1,
                           (const uint64_t *)d_vrow_ptr, (const uint64_t *)d_cbeg, (const uint32_t *)d_clen, (const uint32_t *)d_rs, slice, world, d_vsrc, m, C,
                           vsrc_cap);
        PL_HIP(hipGetLastError());
        PL_HIP(hipStreamSynchronize(stream));
    }
    mem.release(d_row_chunk);
    mem.release(d_cbeg);
    mem.release(d_ckey);
    mem.release(d_forder);
    mem.release(d_grp_of);
    lap("emit level-1 lists");

    // ---- upper levels: while some hub row still lists more than `chunk` virtual rows, group them -------------
    uint64_t *d_lptr = d_hub_first; // H + 1
    uint32_t *d_lids = d_vid_of;    // C entries: per hub row the ids it currently reads, creation order
    uint64_t lids_len = C;
    uint64_t vsrc_len = l1_edges;
    uint64_t *d_parts = nullptr, *d_newcnt = nullptr, *d_moved = nullptr, *d_vidoff = nullptr, *d_noff = nullptr, *d_srcoff = nullptr;
    PL_HIP(mem.alloc(&d_parts, H + 1));
    PL_HIP(mem.alloc(&d_newcnt, H + 1));
    PL_HIP(mem.alloc(&d_moved, H + 1));
    PL_HIP(mem.alloc(&d_vidoff, H + 1));
    PL_HIP(mem.alloc(&d_noff, H + 1));
    PL_HIP(mem.alloc(&d_srcoff, H + 1));
    while (true) {
        next_vid = (next_vid - first_vid + kRowAlign - 1) / kRowAlign * kRowAlign + first_vid; // pad the level to whole tiles
        p->level_begin.push_back(next_vid);
        if (!H) break;
        PL_HIP(hipMemsetAsync(d_parts + H, 0, sizeof(uint64_t), stream));
        PL_HIP(hipMemsetAsync(d_newcnt + H, 0, sizeof(uint64_t), stream));
        PL_HIP(hipMemsetAsync(d_moved + H, 0, sizeof(uint64_t), stream));
        hipLaunchKernelGGL(level_parts_kernel, dim3(grid_for(H)), dim3(256), 0, stream, (const uint64_t *)d_lptr, H, chunk, d_parts, d_newcnt, d_moved);
        PL_HIP(hipGetLastError());
        uint64_t tot[3] = {0, 0, 0};
        uint64_t *ins[3] = {d_parts, d_newcnt, d_moved}, *outs[3] = {d_vidoff, d_noff, d_srcoff};
        for (int k = 0; k < 3; k++) {
            size_t bytes = 0;
            PL_HIP(rocprim::exclusive_scan(nullptr, bytes, ins[k], outs[k], (uint64_t)0, (size_t)(H + 1), rocprim::plus<uint64_t>(), stream));
            PL_HIP(need_tmp(bytes));
            PL_HIP(rocprim::exclusive_scan(tmp, bytes, ins[k], outs[k], (uint64_t)0, (size_t)(H + 1), rocprim::plus<uint64_t>(), stream));
            PL_HIP(hipMemcpyAsync(&tot[k], outs[k] + H, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        }
        PL_HIP(hipStreamSynchronize(stream));
        if (tot[0] == 0) break; // every list fits
        if (next_vid + tot[0] >= (uint64_t)kNone || next_vid + tot[0] - first_vid > vrow_cap || vsrc_len + tot[2] > vsrc_cap)
            return "row id space exhausted (n + virtual rows >= 2^32 - 1)";
        uint32_t *d_nids = nullptr;
        PL_HIP(mem.alloc(&d_nids, tot[1] + 1));
        const unsigned eb = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((H + 3) / 4, 4096));
        hipLaunchKernelGGL(level_emit_kernel, dim3(eb), dim3(256), 0, stream, (const uint64_t *)d_lptr, (const uint32_t *)d_lids, H,
                           (const uint64_t *)d_parts, (const uint64_t *)d_vidoff, (const uint64_t *)d_noff, (const uint64_t *)d_srcoff, next_vid,
                           first_vid, d_vlen, d_vsrc + vsrc_len, d_nids);
        PL_HIP(hipGetLastError());
        PL_HIP(hipStreamSynchronize(stream));
        // lptr <- noff (the scan already has H + 1 entries), lids <- nids
        std::swap(d_lptr, d_noff);
        if (d_lids != d_vid_of) mem.release(d_lids);
        else mem.release(d_vid_of);
        d_lids = d_nids;
        lids_len = tot[1];
        next_vid += tot[0];
        vsrc_len += tot[2];
    }
    if (p->level_begin.size() >= 2 && p->level_begin[p->level_begin.size() - 1] == p->level_begin[p->level_begin.size() - 2])
        p->level_begin.pop_back(); // no trailing empty level
    p->nv = next_vid - first_vid;
    if (next_vid >= (uint64_t)kNone) return "row id space exhausted (n + virtual rows >= 2^32 - 1)";
    (void)lids_len;
    lap("upper levels");

    // ---- assemble: node rows [0, n_pad), then the virtual rows ------------------------------------------------
    const uint64_t nv = p->nv, rows_total = n_pad + nv;
    uint32_t *d_hub_index = nullptr;
    uint64_t *d_len = nullptr, *d_row_ptr = nullptr;
    PL_HIP(mem.alloc(&d_hub_index, n_pad + 1));
    PL_HIP(mem.alloc(&d_len, n_pad + 1));
    PL_HIP(mem.alloc_out(&d_row_ptr, rows_total + 2));
    PL_HIP(hipMemsetAsync(d_len, 0, (n_pad + 1) * sizeof(uint64_t), stream));
    if (H) {
        hipLaunchKernelGGL(hub_index_kernel, dim3(grid_for(H)), dim3(256), 0, stream, (const uint32_t *)d_hub_rows, H, d_hub_index);
        PL_HIP(hipGetLastError());
    }
    uint64_t real_total = 0;
    if (n_pad) {
        hipLaunchKernelGGL(real_len_kernel, dim3(grid_for(n_pad)), dim3(256), 0, stream, (const uint64_t *)d_deg, (const uint8_t *)d_is_split,
                           (const uint32_t *)d_hub_index, (const uint64_t *)d_lptr, n_pad, d_len);
        PL_HIP(hipGetLastError());
    }
    {
        size_t bytes = 0;
        PL_HIP(rocprim::exclusive_scan(nullptr, bytes, d_len, d_row_ptr, (uint64_t)0, (size_t)(n_pad + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::exclusive_scan(tmp, bytes, d_len, d_row_ptr, (uint64_t)0, (size_t)(n_pad + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(hipMemcpyAsync(&real_total, d_row_ptr + n_pad, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        // virtual rows: exclusive scan of all their lengths (padding rows are empty, lists lie in row order)
        auto wide = rocprim::make_transform_iterator(d_vlen, Widen32());
        bytes = 0;
        PL_HIP(rocprim::exclusive_scan(nullptr, bytes, wide, d_vrow_ptr, (uint64_t)0, (size_t)(nv + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(need_tmp(bytes));
        PL_HIP(rocprim::exclusive_scan(tmp, bytes, wide, d_vrow_ptr, (uint64_t)0, (size_t)(nv + 1), rocprim::plus<uint64_t>(), stream));
        PL_HIP(hipStreamSynchronize(stream));
    }
    hipLaunchKernelGGL(virtual_row_ptr_kernel, dim3(grid_for(nv + 1)), dim3(256), 0, stream, (const uint64_t *)d_vrow_ptr, nv, real_total,
                       d_row_ptr + n_pad);
    PL_HIP(hipGetLastError());
    const uint64_t src_len = real_total + vsrc_len;
    uint32_t *d_src = nullptr;
    PL_HIP(mem.alloc_out(&d_src, src_len + 4));
    if (n_pad) {
        hipLaunchKernelGGL(real_fill_kernel, dim3(grid_for(n_pad * 4)), dim3(256), 0, stream, (const uint64_t *)d_row_ptr, (const uint8_t *)d_is_split,
                           (const uint32_t *)d_hub_index, (const uint64_t *)d_lptr, (const uint32_t *)d_lids, (const uint64_t *)d_rp,
                           (const uint32_t *)d_rs, n_pad, slice, world, d_src);
        PL_HIP(hipGetLastError());
    }
    if (vsrc_len) PL_HIP(hipMemcpyAsync(d_src + real_total, d_vsrc, vsrc_len * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    PL_HIP(hipStreamSynchronize(stream));
    lap("assemble");

    out->d_row_ptr = (uint64_t *)mem.disown(d_row_ptr);
    out->d_src = (uint32_t *)mem.disown(d_src);
    out->src_len = src_len;
    out->d_order = (uint32_t *)mem.disown(d_order);
    out->d_dev_of = (uint32_t *)mem.disown(d_dev_of);
    out->d_outdeg_dev = (uint32_t *)mem.disown(d_outdeg_dev);
#ifdef HB_DEBUG_BOUNDS
    {
        unsigned int line = 0;
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(&line, HIP_SYMBOL(g_plan_dbg_line), sizeof(line)) == hipSuccess && line)
            return "HB_DEBUG_BOUNDS: planner index check failed at hb_plan.hip line " + std::to_string(line);
    }
#endif
    return "";
}

} // namespace hb
