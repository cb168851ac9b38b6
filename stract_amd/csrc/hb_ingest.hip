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
//   records --H2D in slabs--> unpack (from, to as u128 keys; "flagged" byte), kept in chunks of <= 64 Mi records
//   node set: per chunk radix sort + unique of its endpoint keys, merged into the running sorted set (128-bit keys)
//   endpoints -> sids: binary search in the sorted id array
//   (to_sid, from_sid) 64-bit keys + stream position: STABLE radix sort -> the first record of
//   every pair is the head of its run -> flag filter on the head -> select -> CSR
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "hb_internal.h"
#include "hb_guard_alloc.h" // no-op unless built with -DHB_GUARD_ALLOC (debug: unmapped guard range behind every buffer)

namespace {

using u128 = rocprim::uint128_t;

#define IG_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) return std::string(#call) + ": " + hipGetErrorString(e_); \
    } while (0)

struct DevMem {
    std::vector<void *> ptrs;
    std::vector<size_t> sizes;
    size_t cur = 0, peak = 0; // bytes held through this object (high-water mark: hb_stats.ingest_peak_bytes)
    ~DevMem()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    void note(size_t bytes) // memory held elsewhere while this object lives (the record chunks)
    {
        cur += bytes;
        peak = std::max(peak, cur);
    }
    void unnote(size_t bytes) { cur -= std::min(cur, bytes); }
    template <typename T>
    hipError_t alloc(T **out, size_t count)
    {
        void *p = nullptr;
        const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) {
            ptrs.push_back(p);
            sizes.push_back(bytes);
            note(bytes);
        }
        *out = (T *)p;
        return e;
    }
    void release(void *p)
    {
        for (size_t i = 0; i < ptrs.size(); i++)
            if (ptrs[i] == p && p) {
                (void)hipFree(p);
                ptrs[i] = nullptr;
                unnote(sizes[i]);
            }
    }
    void disown(void *p) // the caller keeps it
    {
        for (auto &q : ptrs)
            if (q == p) q = nullptr;
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

// one thread per record of a chunk: 64-bit pair key (to_sid, from_sid), ~0 when an endpoint is unknown
// (harmonic.rs:135: such records are ignored), and its stream position (base + i < 2^32)
__global__ __launch_bounds__(256) void pair_keys_kernel(const u128 *endpoints, uint64_t count, uint64_t base, const u128 *ids, uint64_t n,
                                                        uint64_t *pair, uint32_t *pos)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t f = find_sid(ids, n, endpoints[2 * i]);
    const uint32_t t = find_sid(ids, n, endpoints[2 * i + 1]);
    pair[base + i] = (f == 0xFFFFFFFFu || t == 0xFFFFFFFFu) ? ~0ull : (((uint64_t)t << 32) | (uint64_t)f);
    pos[base + i] = (uint32_t)(base + i);
}

// after the stable sort: head[i] = first record of its pair; keep[i] = head, known endpoints, not flagged
__global__ __launch_bounds__(256) void heads_kernel(const uint64_t *pair, const uint32_t *pos, const uint8_t *bad, uint64_t m,
                                                    uint8_t *keep, unsigned long long *counts)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool head = false, kept = false;
    if (i < m) {
        const uint64_t k = pair[i];
        head = (k != ~0ull) && (i == 0 || pair[i - 1] != k);
        kept = head && !bad[pos[i]];
        keep[i] = kept ? 1 : 0;
    }
    const unsigned long long nh = __popcll(__ballot(head));
    if ((threadIdx.x & 63) == 0 && nh) atomicAdd(&counts[blockIdx.x & 63], nh); // striped; summed on the host
}

// the kept pair keys, ascending (to, from): src[i] = from; row_ptr[r] = first position whose `to` is >= r.  A thread that
// starts a new destination fills the row pointers of the gap behind it (no atomics: per-row counters under a sorted
// sweep are same-address atomic chains - a hub with a million in-edges serialises a million of them)
__global__ __launch_bounds__(256) void csr_from_keys_kernel(const uint64_t *sel, uint64_t m_eff, uint64_t n, uint32_t *src, uint64_t *row_ptr)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > m_eff) return;
    if (i == m_eff) { // rows behind the last destination (all rows when nothing was kept)
        const uint64_t last = m_eff ? (sel[m_eff - 1] >> 32) + 1 : 0;
        for (uint64_t r = last; r <= n; r++) row_ptr[r] = m_eff;
        return;
    }
    const uint64_t k = sel[i];
    src[i] = (uint32_t)k;
    const uint64_t t = k >> 32, tp = i ? (sel[i - 1] >> 32) + 1 : 0; // rows tp .. t start here
    for (uint64_t r = tp; r <= t; r++) row_ptr[r] = i;
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
struct ByteToU64 {
    __device__ uint64_t operator()(uint8_t b) const { return (uint64_t)b; }
};

unsigned grid_for(uint64_t count) { return (unsigned)((count + 255) / 256); }

} // namespace

namespace hb {

void IngestStream::free_all()
{
    for (IngestChunk &c : chunks) {
        if (c.d_end) (void)hipFree(c.d_end);
        if (c.d_bad) (void)hipFree(c.d_bad);
    }
    chunks.clear();
    for (void *&p : d_slab) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    count = bytes = 0;
}

// Records -> device, behind what is already there: endpoint keys (2 per record, stream order) and "flagged" bytes, in
// chunks of at most chunk_records records (no reallocation / copy when the stream grows, nothing over-allocated: the
// whole stream costs 33 bytes per record).  Slab-wise H2D through two staging buffers.  On failure ("... out of
// memory") the stream holds exactly the records of the earlier batches.
std::string gpu_ingest_append(void *stream_v, IngestStream *st, const hb_edge *edges, uint64_t m)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (!m) return "";
    if (st->max_records && st->count + m > st->max_records) return "too many records for the device ingest (2^32 limit): use HB_FLAG_HOST_INGEST";
    const uint64_t slab = 1ull << 22; // 4 Mi records = 160 MiB per slab, two slabs in flight
    for (int k = 0; k < 2; k++)
        if (!st->d_slab[k]) {
            st->slab_cap = std::max(st->slab_cap, std::min<uint64_t>(slab, m));
            if (hipMalloc(&st->d_slab[k], st->slab_cap * sizeof(hb_edge)) != hipSuccess) {
                (void)hipGetLastError();
                st->d_slab[k] = nullptr;
                return "hipMalloc(record staging): out of memory";
            }
        }
    if (std::min<uint64_t>(slab, m) > st->slab_cap) { // an earlier, smaller batch sized the staging buffers
        for (void *&p : st->d_slab) {
            (void)hipFree(p);
            p = nullptr;
        }
        st->slab_cap = 0;
        return gpu_ingest_append(stream_v, st, edges, m);
    }
    struct Events { // destroyed on every exit path
        hipEvent_t e[2] = {nullptr, nullptr};
        ~Events()
        {
            for (hipEvent_t x : e)
                if (x) (void)hipEventDestroy(x);
        }
    } evs;
    IG_HIP(hipEventCreateWithFlags(&evs.e[0], hipEventDisableTiming));
    IG_HIP(hipEventCreateWithFlags(&evs.e[1], hipEventDisableTiming));
    // state to restore if the device runs out of memory in the middle of this batch
    const size_t chunks_before = st->chunks.size();
    const uint64_t last_count_before = chunks_before ? st->chunks.back().count : 0, count_before = st->count, bytes_before = st->bytes;
    auto undo = [&]() {
        (void)hipStreamSynchronize(stream);
        while (st->chunks.size() > chunks_before) {
            (void)hipFree(st->chunks.back().d_end);
            (void)hipFree(st->chunks.back().d_bad);
            st->chunks.pop_back();
        }
        if (chunks_before) st->chunks.back().count = last_count_before;
        st->count = count_before;
        st->bytes = bytes_before;
        (void)hipGetLastError();
    };
    const uint64_t chunk_records = st->chunk_records ? st->chunk_records : (1ull << 26);
    int b = 0;
    for (uint64_t off = 0; off < m; b ^= 1) {
        if (st->chunks.empty() || st->chunks.back().count == st->chunks.back().cap) {
            IngestChunk c;
            uint64_t cap = 1ull << 20;
            while (cap < m - off && cap < chunk_records) cap <<= 1;
            c.cap = std::min(cap, chunk_records);
            const uint64_t add = c.cap * 33;
            if ((st->max_bytes && st->bytes + add > st->max_bytes) || hipMalloc(&c.d_end, c.cap * 32) != hipSuccess) {
                undo();
                return "hipMalloc(endpoint keys): out of memory";
            }
            if (hipMalloc((void **)&c.d_bad, c.cap) != hipSuccess) {
                (void)hipFree(c.d_end);
                undo();
                return "hipMalloc(flag bytes): out of memory";
            }
            st->bytes += add;
            st->chunks.push_back(c);
        }
        IngestChunk &c = st->chunks.back();
        const uint64_t cnt = std::min(std::min(slab, m - off), c.cap - c.count);
        IG_HIP(hipEventSynchronize(evs.e[b])); // the kernel that last read this slab buffer has finished
        IG_HIP(hipMemcpyAsync(st->d_slab[b], edges + off, cnt * sizeof(hb_edge), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(cnt)), dim3(256), 0, stream, (const hb_edge *)st->d_slab[b], cnt, c.count, (u128 *)c.d_end, c.d_bad);
        IG_HIP(hipGetLastError());
        IG_HIP(hipEventRecord(evs.e[b], stream));
        c.count += cnt;
        st->count += cnt;
        off += cnt;
    }
    IG_HIP(hipStreamSynchronize(stream));
    return "";
}

std::string gpu_ingest_edges(void *stream_v, const hb_u128 *node_ids, uint64_t n_in, const hb_edge *edges, uint64_t m,
                             DenseGraph *out, DeviceCsr *keep, uint64_t *peak_bytes)
{
    if (keep) *keep = DeviceCsr{};
    if (m && !edges) return "edges == NULL with m > 0";
    IngestStream st;
    st.max_records = 0xFFFFFF00ull;
    std::string e = gpu_ingest_append(stream_v, &st, edges, m);
    if (!e.empty()) {
        st.free_all();
        return e;
    }
    return gpu_ingest_reduce(stream_v, node_ids, n_in, &st, out, keep, peak_bytes);
}

// sorted unique keys of `count` device keys (in place in d_keys; d_alt = second buffer of the same size); *n_out = their number
static std::string sort_unique_keys(hipStream_t stream, DevMem &mem, u128 *d_keys, u128 *d_alt, uint64_t count, void *&tmp, size_t &tmp_bytes,
                                    uint64_t *d_n, uint64_t *n_out)
{
    auto need_tmp = [&](size_t bytes) -> hipError_t {
        if (bytes <= tmp_bytes) return hipSuccess;
        if (tmp) mem.release(tmp);
        tmp_bytes = bytes + (bytes >> 3);
        char *p = nullptr;
        hipError_t e = mem.alloc(&p, tmp_bytes);
        tmp = p;
        return e;
    };
    rocprim::double_buffer<u128> db(d_keys, d_alt);
    size_t bytes = 0;
    IG_HIP(rocprim::radix_sort_keys(nullptr, bytes, db, (size_t)count, 0, 128, stream));
    IG_HIP(need_tmp(bytes));
    IG_HIP(rocprim::radix_sort_keys(tmp, bytes, db, (size_t)count, 0, 128, stream));
    u128 *sorted = db.current(), *other = db.alternate();
    bytes = 0;
    IG_HIP(rocprim::unique(nullptr, bytes, sorted, other, d_n, (size_t)count, rocprim::equal_to<u128>(), stream));
    IG_HIP(need_tmp(bytes));
    IG_HIP(rocprim::unique(tmp, bytes, sorted, other, d_n, (size_t)count, rocprim::equal_to<u128>(), stream));
    IG_HIP(hipMemcpyAsync(n_out, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    IG_HIP(hipStreamSynchronize(stream));
    if (other != d_keys) IG_HIP(hipMemcpyAsync(d_keys, other, *n_out * sizeof(u128), hipMemcpyDeviceToDevice, stream));
    return "";
}

// The reduction proper, from records that are already on the device (gpu_ingest_append); consumes the stream: every
// chunk is freed as soon as its records are turned into 12-byte (pair key, position) entries.
// Device memory: 33 B per record while the stream is held, + 12 B per record of pair keys / positions, + the node set
// (16 B per node, built chunk by chunk: a chunk's 2 x count endpoint keys are sorted, made unique and merged into the
// running sorted set - never a sort over all 2m endpoints); then, the chunks gone, 24 B per record for the stable
// (to, from) sort.
std::string gpu_ingest_reduce(void *stream_v, const hb_u128 *node_ids, uint64_t n_in, IngestStream *st, DenseGraph *out, DeviceCsr *keep,
                              uint64_t *peak_bytes)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (keep) *keep = DeviceCsr{};
    if (peak_bytes) *peak_bytes = 0;
    struct FreeStream {
        IngestStream *s;
        ~FreeStream() { s->free_all(); }
    } free_stream{st};
    const uint64_t m = st->count;
    out->ids.clear();
    out->row_ptr.clear();
    out->src.clear();
    out->m_input = m;
    out->m_unique = 0;
    DevMem mem;
    struct Peak {
        DevMem &mem;
        uint64_t *out;
        ~Peak()
        {
            if (out) *out = mem.peak;
        }
    } peak_guard{mem, peak_bytes};
    mem.note(st->bytes);
    for (void *&p : st->d_slab) { // the staging buffers are no longer needed
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    // one-thread-per-record kernels below: a dispatch holds at most 2^32 - 1 work-items, positions are 32-bit
    if (m >= 0xFFFFFF00ull) return "too many records for the device ingest (2^32 limit): use HB_FLAG_HOST_INGEST";
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
    uint64_t *d_n = nullptr;
    IG_HIP(mem.alloc(&d_n, 1));
    // HB_TRACE_INGEST=1: phase times on stderr
    const bool trace = std::getenv("HB_TRACE_INGEST") != nullptr;
    double t_lap = now_ms();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(stream);
        const double t = now_ms();
        std::fprintf(stderr, "[hb ingest] %-28s %9.1f ms   (device bytes held %.2f GB, peak %.2f GB)\n", what, t - t_lap, mem.cur / 1e9, mem.peak / 1e9);
        t_lap = t;
    };

    // ---- node set: sorted unique u128 keys in d_ids
    u128 *d_ids = nullptr;
    uint64_t n = 0;
    if (node_ids && n_in) {
        u128 *d_alt = nullptr;
        hb_u128 *d_raw = nullptr;
        IG_HIP(mem.alloc(&d_ids, n_in));
        IG_HIP(mem.alloc(&d_alt, n_in));
        IG_HIP(mem.alloc(&d_raw, n_in));
        IG_HIP(hipMemcpyAsync(d_raw, node_ids, n_in * sizeof(hb_u128), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(ids_to_keys_kernel, dim3(grid_for(n_in)), dim3(256), 0, stream, (const hb_u128 *)d_raw, n_in, d_ids);
        IG_HIP(hipGetLastError());
        IG_HIP(hipStreamSynchronize(stream));
        mem.release(d_raw);
        std::string e = sort_unique_keys(stream, mem, d_ids, d_alt, n_in, tmp, tmp_bytes, d_n, &n);
        if (!e.empty()) return e;
        IG_HIP(hipStreamSynchronize(stream));
        mem.release(d_alt);
    } else {
        // work buffers are allocated once (multi-GB hipMalloc / hipFree per chunk is what would dominate here): two key
        // buffers for the largest chunk, two set buffers that grow by doubling
        uint64_t max_cand = 0;
        for (const IngestChunk &c : st->chunks) max_cand = std::max(max_cand, 2 * c.count);
        u128 *d_keys = nullptr, *d_alt = nullptr, *d_set[2] = {nullptr, nullptr};
        uint64_t set_cap = 0;
        if (max_cand) {
            IG_HIP(mem.alloc(&d_keys, max_cand));
            IG_HIP(mem.alloc(&d_alt, max_cand));
        }
        for (const IngestChunk &c : st->chunks) {
            if (!c.count) continue;
            const uint64_t cand = 2 * c.count;
            IG_HIP(hipMemcpyAsync(d_keys, c.d_end, cand * sizeof(u128), hipMemcpyDeviceToDevice, stream));
            uint64_t cu = 0;
            std::string e = sort_unique_keys(stream, mem, d_keys, d_alt, cand, tmp, tmp_bytes, d_n, &cu);
            if (!e.empty()) return e;
            IG_HIP(hipStreamSynchronize(stream));
            if (n + cu > set_cap) { // grow both set buffers; the running set moves to the new d_set[0]
                const uint64_t cap = std::max<uint64_t>(n + cu, 2 * set_cap);
                u128 *a0 = nullptr, *a1 = nullptr;
                IG_HIP(mem.alloc(&a0, cap));
                if (n) IG_HIP(hipMemcpyAsync(a0, d_set[0], n * sizeof(u128), hipMemcpyDeviceToDevice, stream));
                IG_HIP(hipStreamSynchronize(stream));
                if (d_set[0]) mem.release(d_set[0]);
                if (d_set[1]) mem.release(d_set[1]);
                IG_HIP(mem.alloc(&a1, cap));
                d_set[0] = a0;
                d_set[1] = a1;
                set_cap = cap;
            }
            if (n == 0) { // the first chunk's set is the running set
                IG_HIP(hipMemcpyAsync(d_set[0], d_keys, cu * sizeof(u128), hipMemcpyDeviceToDevice, stream));
                n = cu;
                continue;
            }
            // running set U chunk set: merge the two sorted unique lists, drop the keys present in both
            size_t bytes = 0;
            IG_HIP(rocprim::merge(nullptr, bytes, d_set[0], d_keys, d_set[1], (size_t)n, (size_t)cu, rocprim::less<u128>(), stream));
            IG_HIP(need_tmp(bytes));
            IG_HIP(rocprim::merge(tmp, bytes, d_set[0], d_keys, d_set[1], (size_t)n, (size_t)cu, rocprim::less<u128>(), stream));
            bytes = 0;
            IG_HIP(rocprim::unique(nullptr, bytes, d_set[1], d_set[0], d_n, (size_t)(n + cu), rocprim::equal_to<u128>(), stream));
            IG_HIP(need_tmp(bytes));
            IG_HIP(rocprim::unique(tmp, bytes, d_set[1], d_set[0], d_n, (size_t)(n + cu), rocprim::equal_to<u128>(), stream));
            IG_HIP(hipMemcpyAsync(&n, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            IG_HIP(hipStreamSynchronize(stream));
        }
        if (d_keys) mem.release(d_keys);
        if (d_alt) mem.release(d_alt);
        if (d_set[1]) mem.release(d_set[1]);
        d_ids = d_set[0];
    }
    lap("node set");
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

    // ---- pair keys + stream positions, chunk by chunk; a chunk is freed once its records are mapped
    uint64_t *d_pair = nullptr, *d_pair_alt = nullptr;
    uint32_t *d_pos = nullptr, *d_pos_alt = nullptr;
    uint8_t *d_bad = nullptr;
    IG_HIP(mem.alloc(&d_pair, m));
    IG_HIP(mem.alloc(&d_pos, m));
    IG_HIP(mem.alloc(&d_bad, m));
    {
        uint64_t base = 0;
        for (IngestChunk &c : st->chunks) {
            if (c.count) {
                hipLaunchKernelGGL(pair_keys_kernel, dim3(grid_for(c.count)), dim3(256), 0, stream, (const u128 *)c.d_end, c.count, base,
                                   (const u128 *)d_ids, n, d_pair, d_pos);
                IG_HIP(hipGetLastError());
                IG_HIP(hipMemcpyAsync(d_bad + base, c.d_bad, c.count, hipMemcpyDeviceToDevice, stream));
                IG_HIP(hipStreamSynchronize(stream));
            }
            base += c.count;
            (void)hipFree(c.d_end);
            (void)hipFree(c.d_bad);
            c.d_end = nullptr;
            c.d_bad = nullptr;
            mem.unnote(c.cap * 33);
        }
        st->chunks.clear();
        st->bytes = 0;
    }
    mem.release(d_ids);
    lap("pair keys, chunks freed");

    // ---- stable sort by (to, from): the first record of every pair heads its run
    IG_HIP(mem.alloc(&d_pair_alt, m));
    IG_HIP(mem.alloc(&d_pos_alt, m));
    uint64_t *d_pair_s = nullptr;
    uint32_t *d_pos_s = nullptr;
    {
        rocprim::double_buffer<uint64_t> kb(d_pair, d_pair_alt);
        rocprim::double_buffer<uint32_t> vb(d_pos, d_pos_alt);
        size_t bytes = 0;
        IG_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kb, vb, (size_t)m, 0, 64, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::radix_sort_pairs(tmp, bytes, kb, vb, (size_t)m, 0, 64, stream));
        IG_HIP(hipStreamSynchronize(stream));
        d_pair_s = kb.current();
        d_pos_s = vb.current();
        mem.release(kb.alternate());
        mem.release(vb.alternate());
    }
    lap("stable sort of the pairs");

    // ---- heads, flag filter
    uint8_t *d_keep = nullptr;
    unsigned long long *d_counts = nullptr;
    IG_HIP(mem.alloc(&d_keep, m));
    IG_HIP(mem.alloc(&d_counts, 64));
    IG_HIP(hipMemsetAsync(d_counts, 0, 64 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(m)), dim3(256), 0, stream, (const uint64_t *)d_pair_s, (const uint32_t *)d_pos_s,
                       (const uint8_t *)d_bad, m, d_keep, d_counts);
    IG_HIP(hipGetLastError());
    unsigned long long h_counts[64];
    IG_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost, stream));
    IG_HIP(hipStreamSynchronize(stream));
    mem.release(d_pos_s);
    mem.release(d_bad);
    lap("heads");

    // ---- the kept (to, from) keys, still ascending -> sources + row pointers
    uint32_t *d_src = nullptr;
    uint64_t *d_row_ptr = nullptr, *d_sel = nullptr;
    uint64_t m_eff = 0;
    {
        // count first, so that the arrays are allocated at their final size (the source array outlives this function)
        size_t bytes = 0;
        IG_HIP(rocprim::reduce(nullptr, bytes, rocprim::make_transform_iterator(d_keep, ByteToU64()), d_n, (uint64_t)0, (size_t)m, rocprim::plus<uint64_t>(), stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::reduce(tmp, bytes, rocprim::make_transform_iterator(d_keep, ByteToU64()), d_n, (uint64_t)0, (size_t)m, rocprim::plus<uint64_t>(), stream));
        IG_HIP(hipMemcpyAsync(&m_eff, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        IG_HIP(mem.alloc(&d_sel, std::max<uint64_t>(m_eff, 1)));
        bytes = 0;
        IG_HIP(rocprim::select(nullptr, bytes, d_pair_s, d_keep, d_sel, d_n, (size_t)m, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::select(tmp, bytes, d_pair_s, d_keep, d_sel, d_n, (size_t)m, stream));
        uint64_t m_sel = 0;
        IG_HIP(hipMemcpyAsync(&m_sel, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        if (m_sel != m_eff) return "gpu ingest: kept-record count mismatch";
    }
    mem.release(d_pair_s);
    mem.release(d_keep);
    IG_HIP(mem.alloc(&d_src, std::max<uint64_t>(m_eff, 1)));
    IG_HIP(mem.alloc(&d_row_ptr, n + 1));
    hipLaunchKernelGGL(csr_from_keys_kernel, dim3(grid_for(m_eff + 1)), dim3(256), 0, stream, (const uint64_t *)d_sel, m_eff, n, d_src, d_row_ptr);
    IG_HIP(hipGetLastError());
    IG_HIP(hipStreamSynchronize(stream));
    mem.release(d_sel);
    lap("csr");
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
        mem.disown(d_row_ptr); // not freed by ~DevMem
        mem.disown(d_src);
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
