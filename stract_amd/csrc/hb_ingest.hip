// hb_ingest.hip - the reference's node-set / edge-set semantics on the GPU.
//
// Same contract as the host path (hb_host.cpp: ingest_edges), bit-identical output:
//   node set   = every from / to id of every record, flagged ones included (or the caller's list)
//                (crates/core/src/webgraph/store.rs:338-357), ascending numeric u128 order
//   edge set   = FIRST record of each (from,to) pair in stream order (itertools::unique_by,
//                store.rs:313), THEN dropped when rel_flags & SKIPPED_REL != 0 (harmonic.rs:36-49,131)
//   output     = CSR by destination over sids (rank of the id), sources ascending inside a row
//
// Pipeline (one HIP stream; rocPRIM device primitives for the sorts / scans / selections):
//   records --H2D in slabs--> unpack (from, to as u128 keys; "flagged" byte)
//   node set: radix sort of the 2m keys + unique                      (128-bit keys)
//   endpoints -> sids: binary search in the sorted id array
//   (to_sid, from_sid) 64-bit keys + stream position: STABLE radix sort -> the first record of
//   every pair is the head of its run -> flag filter on the head -> select -> CSR
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "hb_internal.h"

namespace {

using u128 = rocprim::uint128_t;

#define IG_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) return std::string(#call) + ": " + hipGetErrorString(e_); \
    } while (0)

struct DevMem {
    std::vector<void *> ptrs;
    ~DevMem()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T **out, size_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 256));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
    void release(void *p)
    {
        for (auto &q : ptrs)
            if (q == p) {
                (void)hipFree(p);
                q = nullptr;
            }
    }
};

__device__ __forceinline__ u128 make_key(const hb_u128 &v) { return ((u128)v.hi << 64) | (u128)v.lo; }

// one thread per record of the slab: keys[2i] = from, keys[2i+1] = to, bad[i] = flagged
__global__ __launch_bounds__(256) void unpack_kernel(const hb_edge *slab, uint64_t count, uint64_t base, u128 *keys, uint8_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const hb_edge e = slab[i];
    keys[2 * (base + i)] = make_key(e.from);
    keys[2 * (base + i) + 1] = make_key(e.to);
    bad[base + i] = (e.rel_flags & HB_SKIPPED_REL_MASK) ? 1 : 0;
}

__global__ __launch_bounds__(256) void ids_to_keys_kernel(const hb_u128 *ids, uint64_t n, u128 *keys)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = make_key(ids[i]);
}

__global__ __launch_bounds__(256) void keys_to_ids_kernel(const u128 *keys, uint64_t n, hb_u128 *ids)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        hb_u128 v;
        v.lo = (uint64_t)keys[i];
        v.hi = (uint64_t)(keys[i] >> 64);
        ids[i] = v;
    }
}

__device__ __forceinline__ uint32_t find_sid(const u128 *ids, uint64_t n, u128 key)
{
    uint64_t lo = 0, hi = n; // first index with ids[idx] >= key
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ids[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && ids[lo] == key) ? (uint32_t)lo : 0xFFFFFFFFu;
}

// one thread per record: 64-bit pair key (to_sid, from_sid), ~0 when an endpoint is unknown
// (harmonic.rs:135: such records are ignored), and its stream position
__global__ __launch_bounds__(256) void pair_keys_kernel(const u128 *endpoints, uint64_t m, const u128 *ids, uint64_t n,
                                                        uint64_t *pair, uint64_t *pos)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint32_t f = find_sid(ids, n, endpoints[2 * i]);
    const uint32_t t = find_sid(ids, n, endpoints[2 * i + 1]);
    pair[i] = (f == 0xFFFFFFFFu || t == 0xFFFFFFFFu) ? ~0ull : (((uint64_t)t << 32) | (uint64_t)f);
    pos[i] = i;
}

// after the stable sort: head[i] = first record of its pair; keep[i] = head, known endpoints, not flagged
__global__ __launch_bounds__(256) void heads_kernel(const uint64_t *pair, const uint64_t *pos, const uint8_t *bad, uint64_t m,
                                                    uint8_t *keep, unsigned long long *counts, uint32_t *row_count)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool head = false, kept = false;
    if (i < m) {
        const uint64_t k = pair[i];
        head = (k != ~0ull) && (i == 0 || pair[i - 1] != k);
        kept = head && !bad[pos[i]];
        keep[i] = kept ? 1 : 0;
        if (kept) atomicAdd(&row_count[k >> 32], 1u);
    }
    const unsigned long long nh = __popcll(__ballot(head));
    if ((threadIdx.x & 63) == 0 && nh) atomicAdd(&counts[blockIdx.x & 63], nh); // striped; summed on the host
}

__global__ __launch_bounds__(256) void widen_kernel(const uint32_t *in, uint64_t n, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// position of every kept result in the order (Reverse(total_cmp(centrality)), NodeID ascending)
// - the rank that store_harmonic writes to the "harmonic_rank" store
// (crates/core/src/webgraph/centrality/mod.rs:92-103, SortableFloat = f64::total_cmp, lib.rs:259-263).
// vals: one f64 per node in ascending-NodeID order, negative = absent (hb_finish).
__global__ __launch_bounds__(256) void rank_keys_kernel(const double *vals, uint64_t n, uint64_t *key, uint8_t *keep)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = vals[i];
    keep[i] = v >= 0.0 ? 1 : 0;
    uint64_t b = (uint64_t)__double_as_longlong(v);
    b ^= (b >> 63) ? ~0ull : 0x8000000000000000ull; // total_cmp order as unsigned order
    key[i] = ~b;                                    // Reverse(...)
}
__global__ __launch_bounds__(256) void rank_scatter_kernel(const uint64_t *sorted_idx, uint64_t k, uint64_t *rank)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < k) rank[sorted_idx[i]] = i;
}

struct LowHalf {
    __device__ uint32_t operator()(uint64_t k) const { return (uint32_t)k; }
};

unsigned grid_for(uint64_t count) { return (unsigned)((count + 255) / 256); }

} // namespace

namespace hb {

// Records -> device: endpoint keys (2 per record, stream order) and "flagged" bytes, written at record offset `base`
// of buffers that hold at least base + m records.  Slab-wise H2D through two staging buffers.
std::string gpu_ingest_unpack(void *stream_v, const hb_edge *edges, uint64_t m, uint64_t base, void *d_end_v, uint8_t *d_bad)
{
    hipStream_t stream = (hipStream_t)stream_v;
    u128 *d_end = (u128 *)d_end_v;
    if (!m) return "";
    DevMem mem;
    const uint64_t slab = 1ull << 22; // 4 Mi records = 160 MiB per slab, two slabs in flight
    hb_edge *d_slab[2] = {nullptr, nullptr};
    IG_HIP(mem.alloc(&d_slab[0], std::min<uint64_t>(slab, m)));
    IG_HIP(mem.alloc(&d_slab[1], std::min<uint64_t>(slab, m)));
    struct Events { // destroyed on every exit path
        hipEvent_t e[2] = {nullptr, nullptr};
        ~Events()
        {
            for (hipEvent_t x : e)
                if (x) (void)hipEventDestroy(x);
        }
    } evs;
    hipEvent_t *done = evs.e;
    IG_HIP(hipEventCreateWithFlags(&done[0], hipEventDisableTiming));
    IG_HIP(hipEventCreateWithFlags(&done[1], hipEventDisableTiming));
    int b = 0;
    for (uint64_t off = 0; off < m; off += slab, b ^= 1) {
        const uint64_t cnt = std::min(slab, m - off);
        IG_HIP(hipEventSynchronize(done[b])); // the kernel that last read this slab buffer has finished
        IG_HIP(hipMemcpyAsync(d_slab[b], edges + off, cnt * sizeof(hb_edge), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(cnt)), dim3(256), 0, stream, (const hb_edge *)d_slab[b], cnt, base + off, d_end, d_bad);
        IG_HIP(hipGetLastError());
        IG_HIP(hipEventRecord(done[b], stream));
    }
    IG_HIP(hipStreamSynchronize(stream));
    return "";
}

std::string gpu_ingest_edges(void *stream_v, const hb_u128 *node_ids, uint64_t n_in, const hb_edge *edges, uint64_t m,
                             DenseGraph *out, DeviceCsr *keep)
{
    if (keep) *keep = DeviceCsr{};
    if (m && !edges) return "edges == NULL with m > 0";
    void *d_end = nullptr;
    uint8_t *d_bad = nullptr;
    if (hipMalloc(&d_end, std::max<uint64_t>(2 * m * 16, 256)) != hipSuccess) return "hipMalloc(endpoint keys): out of memory";
    if (hipMalloc((void **)&d_bad, std::max<uint64_t>(m, 256)) != hipSuccess) {
        (void)hipFree(d_end);
        return "hipMalloc(flag bytes): out of memory";
    }
    std::string e = gpu_ingest_unpack(stream_v, edges, m, 0, d_end, d_bad);
    if (!e.empty()) {
        (void)hipFree(d_end);
        (void)hipFree(d_bad);
        return e;
    }
    return gpu_ingest_reduce(stream_v, node_ids, n_in, d_end, d_bad, m, out, keep);
}

// The reduction proper, from records that are already on the device (gpu_ingest_unpack); takes ownership of
// d_end_v / d_bad (hipMalloc'ed) and frees them as soon as they are no longer needed.
std::string gpu_ingest_reduce(void *stream_v, const hb_u128 *node_ids, uint64_t n_in, void *d_end_v, uint8_t *d_bad, uint64_t m,
                              DenseGraph *out, DeviceCsr *keep)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (keep) *keep = DeviceCsr{};
    out->ids.clear();
    out->row_ptr.clear();
    out->src.clear();
    out->m_input = m;
    out->m_unique = 0;
    DevMem mem;
    mem.ptrs.push_back(d_end_v);
    mem.ptrs.push_back(d_bad);
    // one-thread-per-record kernels below: a dispatch holds at most 2^32 - 1 work-items
    if (m >= 0xFFFFFF00ull) return "too many records for the device ingest (2^32 limit): use HB_FLAG_HOST_INGEST";
    u128 *d_end = (u128 *)d_end_v; // 2m endpoint keys in stream order
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    auto need_tmp = [&](size_t bytes) -> hipError_t {
        if (bytes <= tmp_bytes) return hipSuccess;
        if (tmp) mem.release(tmp);
        tmp_bytes = bytes + (bytes >> 3);
        char *p = nullptr;
        hipError_t e = mem.alloc(&p, tmp_bytes);
        tmp = p;
        return e;
    };

    // ---- node set: sorted unique u128 keys
    const uint64_t cand = (node_ids && n_in) ? n_in : 2 * m;
    u128 *d_keys = nullptr, *d_sorted = nullptr, *d_ids = nullptr;
    uint64_t n = 0;
    if (cand) {
        IG_HIP(mem.alloc(&d_keys, cand));
        IG_HIP(mem.alloc(&d_sorted, cand));
        if (node_ids && n_in) {
            hb_u128 *d_raw = nullptr;
            IG_HIP(mem.alloc(&d_raw, n_in));
            IG_HIP(hipMemcpyAsync(d_raw, node_ids, n_in * sizeof(hb_u128), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(ids_to_keys_kernel, dim3(grid_for(n_in)), dim3(256), 0, stream, (const hb_u128 *)d_raw, n_in, d_keys);
            IG_HIP(hipGetLastError());
            IG_HIP(hipStreamSynchronize(stream));
            mem.release(d_raw);
        } else {
            IG_HIP(hipMemcpyAsync(d_keys, d_end, cand * sizeof(u128), hipMemcpyDeviceToDevice, stream));
        }
        size_t bytes = 0;
        IG_HIP(rocprim::radix_sort_keys(nullptr, bytes, d_keys, d_sorted, (size_t)cand, 0, 128, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::radix_sort_keys(tmp, bytes, d_keys, d_sorted, (size_t)cand, 0, 128, stream));
        // unique -> d_keys (reused as the output), count on the device
        uint64_t *d_n = nullptr;
        IG_HIP(mem.alloc(&d_n, 1));
        bytes = 0;
        IG_HIP(rocprim::unique(nullptr, bytes, d_sorted, d_keys, d_n, (size_t)cand, rocprim::equal_to<u128>(), stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::unique(tmp, bytes, d_sorted, d_keys, d_n, (size_t)cand, rocprim::equal_to<u128>(), stream));
        IG_HIP(hipMemcpyAsync(&n, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        mem.release(d_sorted);
        d_ids = d_keys;
    }
    if (n >= 0xFFFFFFFFull - (1u << 20)) return "too many nodes (n must be < 2^32 - 2^20)";
    try {
        out->ids.resize(n);
        out->row_ptr.assign(n + 1, 0);
    } catch (const std::bad_alloc &) {
        return "out of host memory for the node set";
    }
    if (n) {
        hb_u128 *d_out_ids = nullptr;
        IG_HIP(mem.alloc(&d_out_ids, n));
        hipLaunchKernelGGL(keys_to_ids_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const u128 *)d_ids, n, d_out_ids);
        IG_HIP(hipGetLastError());
        IG_HIP(hipMemcpyAsync(out->ids.data(), d_out_ids, n * sizeof(hb_u128), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        mem.release(d_out_ids);
    }
    if (n == 0 || m == 0) return "";

    // ---- pair keys, stable sort by (to, from): the first record of every pair heads its run
    uint64_t *d_pair = nullptr, *d_pos = nullptr, *d_pair_s = nullptr, *d_pos_s = nullptr;
    IG_HIP(mem.alloc(&d_pair, m));
    IG_HIP(mem.alloc(&d_pos, m));
    hipLaunchKernelGGL(pair_keys_kernel, dim3(grid_for(m)), dim3(256), 0, stream, (const u128 *)d_end, m, (const u128 *)d_ids, n, d_pair, d_pos);
    IG_HIP(hipGetLastError());
    IG_HIP(hipStreamSynchronize(stream));
    mem.release(d_end);
    mem.release(d_ids);
    IG_HIP(mem.alloc(&d_pair_s, m));
    IG_HIP(mem.alloc(&d_pos_s, m));
    {
        size_t bytes = 0;
        IG_HIP(rocprim::radix_sort_pairs(nullptr, bytes, d_pair, d_pair_s, d_pos, d_pos_s, (size_t)m, 0, 64, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::radix_sort_pairs(tmp, bytes, d_pair, d_pair_s, d_pos, d_pos_s, (size_t)m, 0, 64, stream));
    }
    IG_HIP(hipStreamSynchronize(stream));
    mem.release(d_pair);
    mem.release(d_pos);

    // ---- heads, flag filter, row counts
    uint8_t *d_keep = nullptr;
    unsigned long long *d_counts = nullptr;
    uint32_t *d_row_count = nullptr;
    IG_HIP(mem.alloc(&d_keep, m));
    IG_HIP(mem.alloc(&d_counts, 64));
    IG_HIP(mem.alloc(&d_row_count, n + 1));
    IG_HIP(hipMemsetAsync(d_counts, 0, 64 * sizeof(unsigned long long), stream));
    IG_HIP(hipMemsetAsync(d_row_count, 0, (n + 1) * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(m)), dim3(256), 0, stream, (const uint64_t *)d_pair_s, (const uint64_t *)d_pos_s,
                       (const uint8_t *)d_bad, m, d_keep, d_counts, d_row_count);
    IG_HIP(hipGetLastError());
    unsigned long long h_counts[64];
    IG_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost, stream));

    // ---- sources of the kept records, in (to, from) order
    uint32_t *d_src = nullptr;
    uint64_t *d_meff = nullptr;
    IG_HIP(mem.alloc(&d_src, m));
    IG_HIP(mem.alloc(&d_meff, 1));
    {
        auto low = rocprim::make_transform_iterator(d_pair_s, LowHalf());
        size_t bytes = 0;
        IG_HIP(rocprim::select(nullptr, bytes, low, d_keep, d_src, d_meff, (size_t)m, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::select(tmp, bytes, low, d_keep, d_src, d_meff, (size_t)m, stream));
    }
    uint64_t m_eff = 0;
    IG_HIP(hipMemcpyAsync(&m_eff, d_meff, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));

    // ---- row pointers: exclusive scan of the per-destination counts
    uint64_t *d_cnt64 = nullptr, *d_row_ptr = nullptr;
    IG_HIP(mem.alloc(&d_cnt64, n + 1));
    IG_HIP(mem.alloc(&d_row_ptr, n + 1));
    hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n + 1)), dim3(256), 0, stream, (const uint32_t *)d_row_count, n + 1, d_cnt64);
    IG_HIP(hipGetLastError());
    {
        size_t bytes = 0;
        IG_HIP(rocprim::exclusive_scan(nullptr, bytes, d_cnt64, d_row_ptr, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::exclusive_scan(tmp, bytes, d_cnt64, d_row_ptr, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), stream));
    }
    IG_HIP(hipStreamSynchronize(stream));
    for (int s = 0; s < 64; s++) out->m_unique += h_counts[s];
    const bool to_host = !keep || m_eff <= kKeepHostGraph;
    if (to_host) {
        try {
            out->src.resize(m_eff);
        } catch (const std::bad_alloc &) {
            return "out of host memory for the edge set";
        }
        IG_HIP(hipMemcpyAsync(out->row_ptr.data(), d_row_ptr, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        if (m_eff) IG_HIP(hipMemcpyAsync(out->src.data(), d_src, m_eff * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        if (out->row_ptr[n] != m_eff) return "gpu ingest: row pointer / edge count mismatch";
    } else {
        std::vector<uint64_t>().swap(out->row_ptr); // the reduced graph exists on the device only
    }
    if (keep) { // hand the device CSR to the caller (the device planner continues from it)
        uint64_t last = 0;
        IG_HIP(hipMemcpyAsync(&last, d_row_ptr + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        if (last != m_eff) return "gpu ingest: row pointer / edge count mismatch";
        for (auto &q : mem.ptrs)
            if (q == (void *)d_row_ptr || q == (void *)d_src) q = nullptr; // not freed by ~DevMem
        keep->d_row_ptr = d_row_ptr;
        keep->d_src = d_src;
        keep->m = m_eff;
    }
    return "";
}

// ranks[j] for the j-th kept result (ascending NodeID): its position in the store_harmonic order.
// order_out (optional): the result indices of the first `top` positions of that order (top_nodes, centrality/mod.rs:33-52).
std::string gpu_rank_results(void *stream_v, const double *d_vals, uint64_t n, uint64_t expect, uint64_t *ranks_out, uint64_t *order_out,
                             uint64_t top)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (n == 0 || expect == 0) return "";
    DevMem mem;
    uint64_t *d_key = nullptr, *d_kkey = nullptr, *d_kkey_s = nullptr, *d_idx_s = nullptr, *d_rank = nullptr, *d_cnt = nullptr;
    uint8_t *d_keep = nullptr;
    IG_HIP(mem.alloc(&d_key, n));
    IG_HIP(mem.alloc(&d_keep, n));
    IG_HIP(mem.alloc(&d_kkey, expect));
    IG_HIP(mem.alloc(&d_kkey_s, expect));
    IG_HIP(mem.alloc(&d_idx_s, expect));
    IG_HIP(mem.alloc(&d_rank, expect));
    IG_HIP(mem.alloc(&d_cnt, 1));
    hipLaunchKernelGGL(rank_keys_kernel, dim3(grid_for(n)), dim3(256), 0, stream, d_vals, n, d_key, d_keep);
    IG_HIP(hipGetLastError());
    void *tmp = nullptr;
    size_t bytes = 0, cap = 0;
    auto need = [&](size_t b) -> hipError_t {
        if (b <= cap) return hipSuccess;
        char *p = nullptr;
        hipError_t e = mem.alloc(&p, b);
        tmp = p;
        cap = b;
        return e;
    };
    // kept keys in ascending-NodeID order; their result indices are 0..k-1 in that order
    IG_HIP(rocprim::select(nullptr, bytes, d_key, d_keep, d_kkey, d_cnt, (size_t)n, stream));
    IG_HIP(need(bytes));
    IG_HIP(rocprim::select(tmp, bytes, d_key, d_keep, d_kkey, d_cnt, (size_t)n, stream));
    uint64_t k = 0;
    IG_HIP(hipMemcpyAsync(&k, d_cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    IG_HIP(hipStreamSynchronize(stream));
    if (k != expect) return "gpu rank: kept-result count mismatch";
    auto iota = rocprim::make_counting_iterator<uint64_t>(0);
    bytes = 0;
    // stable: equal centralities keep ascending NodeID order
    IG_HIP(rocprim::radix_sort_pairs(nullptr, bytes, d_kkey, d_kkey_s, iota, d_idx_s, (size_t)k, 0, 64, stream));
    IG_HIP(need(bytes));
    IG_HIP(rocprim::radix_sort_pairs(tmp, bytes, d_kkey, d_kkey_s, iota, d_idx_s, (size_t)k, 0, 64, stream));
    if (ranks_out) {
        hipLaunchKernelGGL(rank_scatter_kernel, dim3(grid_for(k)), dim3(256), 0, stream, (const uint64_t *)d_idx_s, k, d_rank);
        IG_HIP(hipGetLastError());
        IG_HIP(hipMemcpyAsync(ranks_out, d_rank, k * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    }
    if (order_out && top) IG_HIP(hipMemcpyAsync(order_out, d_idx_s, std::min(top, k) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    IG_HIP(hipStreamSynchronize(stream));
    return "";
}

} // namespace hb
