// hb_ingest.hip - the reference's node-set / edge-set semantics on the GPU.
//
// Same contract as the host path (hb_host.cpp: ingest_edges), bit-identical output:
//   node set   = every from / to id of every record, flagged ones included (or the caller's list)
//                (crates/core/src/webgraph/store.rs:338-357), ascending numeric u128 order
//   edge set   = FIRST record of each (from,to) pair in stream order (itertools::unique_by,
//                store.rs:313), THEN dropped when rel_flags & SKIPPED_REL != 0 (harmonic.rs:36-49,131)
//   output     = CSR by destination over sids (rank of the id), sources ascending inside a row
//
// Round 4 rebuild (the round-3 form held 2 x 16-byte endpoint keys per record, found every endpoint's sid by two binary
// searches per record and carried 32-bit stream positions: 47 B per record at its peak, < 2^32 records, 4.3 s of
// reduction at 2.5 G records):
//   append    every endpoint is looked up in / inserted into a device hash table (open addressing, 128-bit keys) AS THE
//             RECORDS ARRIVE and becomes a 32-bit provisional id ("pid", the order of first arrival); a record is stored
//             as (from pid, to pid) + its flag byte = 9 B.  The table work hides behind the host link.
//   finalize  the n table keys are sorted once (128-bit radix sort carrying the pid) -> ids, sid of every pid;
//             records -> 64-bit keys (to sid, from sid, flag bit) by two 4-byte gathers per record;
//             ONE stable radix sort over the (to, from) bits only: equal pairs stay in stream order, so the first record
//             of a pair heads its run and the flag travels in bit 0 - no stream positions, no 2^32 limit;
//             heads -> flag filter -> compaction (rocPRIM select over a computed flag iterator, in <= 2^30-record
//             segments) -> sources + row pointers (run ends + max-scan).
//   memory    9 B per record while the stream is held, 16-17 B per record during the sort, + 20 B per table slot
//             (load factor <= 1/2) and 40 B per node while the ids are sorted.
#include "hb_guard_alloc.h" // FIRST: no-op unless built with -DHB_GUARD_ALLOC=<mode> (debug allocators: guard pages / poison / red zones)
#include "hb_pool.h"        // then: every hipMalloc / hipFree below goes through the caching device allocator (shipped build)
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "hb_internal.h"
#include "hb_table.hip.h"

namespace {

using hbt::u128;
using hbt::Table;
using hbt::table_get;
using hbt::kEmpty;
using hbt::kBusy;
using hbt::kMaxPids;

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
    void note(size_t bytes) // memory held elsewhere while this object lives (the record chunks, the endpoint table)
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

// grid-stride everywhere: a dispatch holds < 2^32 work-items and the stream may hold more records than that
constexpr unsigned kGridCap = 1u << 16;
unsigned grid_for(uint64_t count) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>((count + 255) / 256, 1), kGridCap); }
#define HB_GRID_STRIDE(i, count) for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (count); i += (uint64_t)gridDim.x * 256)

// ---- the endpoint table: hb_table.hip.h (shared with the AMPC counter shard, hb_ampc.hip) -------------------------------------
__global__ __launch_bounds__(256) void table_clear_kernel(uint32_t *pids, uint64_t slots)
{
    HB_GRID_STRIDE(i, slots) pids[i] = kEmpty;
}

// one thread per record of the slab: both endpoints -> pids, (from pid | to pid << 32) and the "flagged" byte
__global__ __launch_bounds__(256) void insert_kernel(const hb_edge *slab, uint64_t count, uint64_t base, Table t, uint64_t *pair, uint8_t *bad)
{
    HB_GRID_STRIDE(i, count)
    {
        const hb_edge e = slab[i];
        const uint32_t f = table_get(t, make_key(e.from), kEmpty);
        const uint32_t to = table_get(t, make_key(e.to), kEmpty);
        pair[base + i] = (uint64_t)f | ((uint64_t)to << 32);
        bad[base + i] = (e.rel_flags & HB_SKIPPED_REL_MASK) ? 1 : 0;
    }
}

// grow: every published entry of the old table goes to the new one with its pid
__global__ __launch_bounds__(256) void rehash_kernel(const u128 *old_keys, const uint32_t *old_pids, uint64_t old_slots, Table t)
{
    HB_GRID_STRIDE(i, old_slots)
    {
        const uint32_t p = old_pids[i];
        if (p < kBusy) (void)table_get(t, old_keys[i], p);
    }
}

// key_of_pid[pid] = key, for every published entry
__global__ __launch_bounds__(256) void table_export_kernel(const u128 *keys, const uint32_t *pids, uint64_t slots, u128 *key_of_pid, uint64_t npid)
{
    HB_GRID_STRIDE(i, slots)
    {
        const uint32_t p = pids[i];
        if (p < kBusy && p < npid) key_of_pid[p] = keys[i];
    }
}

__global__ __launch_bounds__(256) void ids_to_keys_kernel(const hb_u128 *ids, uint64_t n, u128 *keys)
{
    HB_GRID_STRIDE(i, n) keys[i] = make_key(ids[i]);
}

__global__ __launch_bounds__(256) void keys_to_ids_kernel(const u128 *keys, uint64_t n, hb_u128 *ids, uint64_t *lo_out)
{
    HB_GRID_STRIDE(i, n)
    {
        hb_u128 v;
        v.lo = (uint64_t)keys[i];
        v.hi = (uint64_t)(keys[i] >> 64);
        ids[i] = v;
        if (lo_out) lo_out[i] = v.lo;
    }
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *v, uint64_t n)
{
    HB_GRID_STRIDE(i, n) v[i] = (uint32_t)i;
}

// after the sort of (key, pid): the pid at sorted position i has sid i
__global__ __launch_bounds__(256) void sid_scatter_kernel(const uint32_t *pid_sorted, uint64_t n, uint32_t *sid_of_pid)
{
    HB_GRID_STRIDE(i, n) sid_of_pid[pid_sorted[i]] = (uint32_t)i;
}

// caller-supplied node list: sid of a pid = position of its key in the sorted list, kEmpty if it is not a node
// (harmonic.rs:135: records with such an endpoint are ignored)
__global__ __launch_bounds__(256) void sid_search_kernel(const u128 *key_of_pid, uint64_t npid, const u128 *ids, uint64_t n, uint32_t *sid_of_pid)
{
    HB_GRID_STRIDE(i, npid)
    {
        const u128 key = key_of_pid[i];
        uint64_t lo = 0, hi = n; // first index with ids[idx] >= key
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ids[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        sid_of_pid[i] = (lo < n && ids[lo] == key) ? (uint32_t)lo : kEmpty;
    }
}

// (from pid, to pid), flag -> sort key: to sid << (nb + 1) | from sid << 1 | flag; ~0 when an endpoint is no node.
// nb bits hold every sid AND leave the all-ones value unused (nb = bits of n), so ~0 collides with no pair.
__global__ __launch_bounds__(256) void pair_keys_kernel(const uint64_t *pair, const uint8_t *bad, uint64_t count, const uint32_t *sid_of_pid, uint32_t nb,
                                                        uint64_t *out)
{
    HB_GRID_STRIDE(i, count)
    {
        const uint64_t pr = pair[i];
        const uint32_t f = sid_of_pid[(uint32_t)pr], t = sid_of_pid[(uint32_t)(pr >> 32)];
        out[i] = (f == kEmpty || t == kEmpty) ? ~0ull : (((uint64_t)t << (nb + 1)) | ((uint64_t)f << 1) | (uint64_t)(bad[i] & 1));
    }
}

// flag iterators over the sorted keys (computed, never stored): a HEAD is the first record of its (to, from) pair - the
// sort is stable and ignores bit 0, so that is the pair's first record in stream order (store.rs:313); it is KEPT if that
// record is not flagged (harmonic.rs:131)
struct HeadKept {
    const uint64_t *keys; // whole sorted array
    __device__ uint8_t operator()(uint64_t i) const
    {
        const uint64_t k = keys[i];
        if (k == ~0ull || (k & 1)) return 0;
        return (i == 0 || ((keys[i - 1] ^ k) >> 1) != 0) ? 1 : 0;
    }
};
struct HeadAny {
    const uint64_t *keys;
    __device__ uint64_t operator()(uint64_t i) const
    {
        const uint64_t k = keys[i];
        if (k == ~0ull) return 0;
        return (i == 0 || ((keys[i - 1] ^ k) >> 1) != 0) ? 1 : 0;
    }
};

// kept keys, ascending (to, from): src[i] = from; the thread at the END of a destination's run writes the run's end to
// row_end[to + 1] (all other entries stay 0); an inclusive max-scan of row_end then IS the row-pointer array - rows
// without edges inherit the end of the last row before them, and no thread ever walks a gap
__global__ __launch_bounds__(256) void csr_from_keys_kernel(const uint64_t *sel, uint64_t m_eff, uint32_t nb, uint32_t *src, uint64_t *row_end)
{
    const uint64_t mask = (1ull << nb) - 1;
    HB_GRID_STRIDE(i, m_eff)
    {
        const uint64_t k = sel[i];
        src[i] = (uint32_t)((k >> 1) & mask);
        const uint64_t t = k >> (nb + 1);
        if (i + 1 == m_eff || (sel[i + 1] >> (nb + 1)) != t) row_end[t + 1] = i + 1;
    }
}

struct MaxU64 {
    __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
};

// spill: (from pid, to pid), flag -> hb_edge records again (rel_flags collapses to "skipped or not")
__global__ __launch_bounds__(256) void unpack_records_kernel(const uint64_t *pair, const uint8_t *bad, uint64_t count, const u128 *key_of_pid, hb_edge *out)
{
    HB_GRID_STRIDE(i, count)
    {
        const uint64_t pr = pair[i];
        const u128 f = key_of_pid[(uint32_t)pr], t = key_of_pid[(uint32_t)(pr >> 32)];
        hb_edge e;
        e.from.lo = (uint64_t)f;
        e.from.hi = (uint64_t)(f >> 64);
        e.to.lo = (uint64_t)t;
        e.to.hi = (uint64_t)(t >> 64);
        e.rel_flags = bad[i] ? HB_SKIPPED_REL_MASK : 0;
        out[i] = e;
    }
}

// position of every kept result in the order (Reverse(total_cmp(centrality)), NodeID ascending)
// - the rank that store_harmonic writes to the "harmonic_rank" store
// (crates/core/src/webgraph/centrality/mod.rs:92-103, SortableFloat = f64::total_cmp, lib.rs:259-263).
// vals: one f64 per node in ascending-NodeID order, negative = absent (hb_finish).
__global__ __launch_bounds__(256) void rank_keys_kernel(const double *vals, uint64_t n, uint64_t *key, uint8_t *keep)
{
    HB_GRID_STRIDE(i, n)
    {
        const double v = vals[i];
        keep[i] = v >= 0.0 ? 1 : 0;
        uint64_t b = (uint64_t)__double_as_longlong(v);
        b ^= (b >> 63) ? ~0ull : 0x8000000000000000ull; // total_cmp order as unsigned order
        key[i] = ~b;                                    // Reverse(...)
    }
}
__global__ __launch_bounds__(256) void rank_scatter_kernel(const uint64_t *sorted_idx, uint64_t k, uint64_t *rank)
{
    HB_GRID_STRIDE(i, k) rank[sorted_idx[i]] = i;
}


} // namespace

namespace hb {

void IngestStream::free_all()
{
    for (IngestChunk &c : chunks) {
        if (c.d_pair) (void)hipFree(c.d_pair);
        if (c.d_bad) (void)hipFree(c.d_bad);
    }
    chunks.clear();
    for (void *&p : d_slab) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    if (kstream) {
        (void)hipStreamSynchronize((hipStream_t)kstream);
        (void)hipStreamDestroy((hipStream_t)kstream);
        kstream = nullptr;
    }
    if (d_tab_keys) (void)hipFree(d_tab_keys);
    if (d_tab_pids) (void)hipFree(d_tab_pids);
    if (d_counter) (void)hipFree(d_counter);
    if (h_counter) (void)hipHostFree(h_counter);
    d_tab_keys = nullptr;
    d_tab_pids = nullptr;
    d_counter = nullptr;
    h_counter = nullptr;
    tab_slots = 0;
    npid_known = unsynced = 0;
    count = bytes = 0;
}

namespace {

Table table_of(const IngestStream *st)
{
    return Table{(u128 *)st->d_tab_keys, st->d_tab_pids, st->tab_slots - 1, (unsigned long long *)st->d_counter};
}

// exact number of pids handed out (synchronises the stream)
std::string read_npid(hipStream_t stream, IngestStream *st)
{
    IG_HIP(hipMemcpyAsync(st->h_counter, st->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    IG_HIP(hipStreamSynchronize(stream));
    st->npid_known = *st->h_counter;
    st->unsynced = 0;
    return "";
}

// The table must be able to take `incoming` more records (2 endpoints each) at a load factor <= 1/2.  The number of keys
// is only known exactly after a synchronisation, so the test runs on an upper bound (keys at the last read-back + 2 per
// record launched since) and synchronises only when that bound says the table might be too small.
// "... out of memory" when the (larger) table cannot be allocated or the debug byte limit forbids it.
template <class Poll, class WaitOldest>
std::string table_reserve(hipStream_t stream, IngestStream *st, uint64_t incoming, Poll &&poll, WaitOldest &&wait_oldest)
{
    if (!st->d_counter) {
        IG_HIP(hipMalloc((void **)&st->d_counter, 256));
        IG_HIP(hipHostMalloc((void **)&st->h_counter, 4 * sizeof(unsigned long long))); // [0] blocking read-back, [1 + b] snapshot behind slab buffer b
        IG_HIP(hipMemsetAsync(st->d_counter, 0, 256, stream));
    }
    auto fits = [&](uint64_t slots) { return slots && (st->npid_known + 2 * st->unsynced + 2 * incoming) * 2 <= slots; };
    if (fits(st->tab_slots)) return ""; // the common case costs no runtime call at all
    if (st->tab_slots) {
        const double t0 = now_ms();
        poll(); // snapshots behind the slabs that have completed (non-blocking)
        st->trace_polls++;
        st->trace_ms_poll += now_ms() - t0;
        if (fits(st->tab_slots)) return "";
        // wait for the OLDER of the two slabs in flight only (its kernel is usually done, the newer one keeps the device busy),
        // take its snapshot; draining the whole pipe (read_npid) is the last resort
        wait_oldest();
        poll();
        if (fits(st->tab_slots)) return "";
        const double t1 = now_ms();
        const std::string e = read_npid(stream, st);
        st->trace_reads++;
        st->trace_ms_read += now_ms() - t1;
        if (!e.empty()) return e;
        if (fits(st->tab_slots)) return "";
    }
    if (st->npid_known + 2 * incoming >= kMaxPids) return "too many nodes for the device ingest (2^31 endpoint ids)";
    uint64_t slots = 1ull << 12;
    while (slots < 4 * (st->npid_known + 2 * st->unsynced + 2 * incoming)) slots <<= 1; // load factor <= 1/4 even if every record in flight brings two new ids
    const uint64_t add = slots * 20;
    if (st->max_bytes && st->bytes + add > st->max_bytes) return "hipMalloc(endpoint table): out of memory";
    void *nk = nullptr;
    uint32_t *np = nullptr;
    if (hipMalloc(&nk, slots * 16) != hipSuccess) {
        (void)hipGetLastError();
        return "hipMalloc(endpoint table): out of memory";
    }
    if (hipMalloc((void **)&np, slots * 4) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(nk);
        return "hipMalloc(endpoint table): out of memory";
    }
    hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for(slots)), dim3(256), 0, stream, np, slots);
    IG_HIP(hipGetLastError());
    const uint64_t old_slots = st->tab_slots;
    void *ok = st->d_tab_keys;
    uint32_t *op = st->d_tab_pids;
    st->d_tab_keys = nk;
    st->d_tab_pids = np;
    st->tab_slots = slots;
    st->bytes += add;
    st->peak_bytes = std::max(st->peak_bytes, st->bytes);
    if (old_slots) {
        hipLaunchKernelGGL(rehash_kernel, dim3(grid_for(old_slots)), dim3(256), 0, stream, (const u128 *)ok, (const uint32_t *)op, old_slots, table_of(st));
        IG_HIP(hipGetLastError());
        IG_HIP(hipStreamSynchronize(stream));
        (void)hipFree(ok);
        (void)hipFree(op);
        st->bytes -= old_slots * 20;
    }
    return "";
}

} // namespace

// Records -> device, behind what is already there: every endpoint goes through the endpoint table (a 32-bit provisional
// id per distinct NodeID), the record is kept as (from pid, to pid) + its "flagged" byte - 9 bytes - in chunks of at most
// chunk_records records (no reallocation / copy when the stream grows).  Slab-wise H2D through two staging buffers: the
// table kernel of slab k runs while slab k + 1 crosses the link.  On failure ("... out of memory", "too many ...") the
// stream holds exactly the records of the earlier batches (the table may know a few endpoints more: harmless, a failed
// stream is spilled to the host and the table dropped).
std::string gpu_ingest_append(void *stream_v, IngestStream *st, const hb_edge *edges, uint64_t m)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (!m) return "";
    if (!st->count && st->chunks.empty()) HB_POOL_RESET_PEAK(); // a new stream: its high-water mark starts here
    if (st->max_records && st->count + m >= st->max_records) return "too many records for the device ingest: use HB_FLAG_HOST_INGEST";
    // 1 Mi records = 40 MiB per slab, two slabs in flight: large enough for the link's full rate (55 GB/s from 16 MiB up,
    // profiles/r04a_h2d_probe.txt), small enough that "every record in flight may bring two new ids" stays a small reserve
    const uint64_t slab = 1ull << 20;
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
    // two streams: the copies run on the caller's stream, the table kernels on the stream of the ingest; slab k is being
    // reduced while slab k + 1 crosses the link (events: copied[b] = slab buffer b filled, consumed[b] = its kernel done)
    if (!st->kstream) {
        hipStream_t ks = nullptr;
        IG_HIP(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
        st->kstream = (void *)ks;
    }
    hipStream_t kstream = (hipStream_t)st->kstream;
    struct Events { // destroyed on every exit path
        hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
        ~Events()
        {
            for (hipEvent_t x : copied)
                if (x) (void)hipEventDestroy(x);
            for (hipEvent_t x : consumed)
                if (x) (void)hipEventDestroy(x);
        }
    } evs;
    for (int k = 0; k < 2; k++) {
        IG_HIP(hipEventCreateWithFlags(&evs.copied[k], hipEventDisableTiming));
        IG_HIP(hipEventCreateWithFlags(&evs.consumed[k], hipEventDisableTiming));
    }
    IG_HIP(hipStreamSynchronize(stream)); // whatever the caller's stream still does must not race the kernels of the other one
    // Key-count snapshots without blocking: behind every table kernel the counter is copied to a pinned word; once that
    // slab's `consumed` event has fired the word is exact for the moment after that slab, and only the records launched
    // since then are unknown (table_reserve's upper bound) - a synchronisation per slab would serialise copy and kernel.
    uint64_t launched = 0, snap_launched[2] = {0, 0};
    bool snap_pending[2] = {false, false};
    auto poll_snapshots = [&]() {
        for (int k = 0; k < 2; k++)
            if (snap_pending[k] && hipEventQuery(evs.consumed[k]) == hipSuccess) {
                snap_pending[k] = false;
                const uint64_t v = st->h_counter[1 + k];
                if (v >= st->npid_known && launched - snap_launched[k] <= st->unsynced) {
                    st->npid_known = v;
                    st->unsynced = launched - snap_launched[k];
                }
            }
    };
    // state to restore if the device runs out of memory in the middle of this batch
    const size_t chunks_before = st->chunks.size();
    const uint64_t last_count_before = chunks_before ? st->chunks.back().count : 0, count_before = st->count;
    auto undo = [&]() {
        (void)hipStreamSynchronize(stream);
        (void)hipStreamSynchronize(kstream);
        while (st->chunks.size() > chunks_before) {
            (void)hipFree(st->chunks.back().d_pair);
            (void)hipFree(st->chunks.back().d_bad);
            st->bytes -= st->chunks.back().cap * 9;
            st->chunks.pop_back();
        }
        if (chunks_before) st->chunks.back().count = last_count_before;
        st->count = count_before;
        (void)hipGetLastError();
    };
    const uint64_t chunk_records = st->chunk_records ? st->chunk_records : (1ull << 27);
    // HB_TRACE_INGEST=1: where the host time of this call goes
    const bool trace = std::getenv("HB_TRACE_INGEST") != nullptr;
    const double t_call = now_ms();
    double ms_alloc = 0, ms_reserve = 0, ms_issue = 0;
    int b = 0;
    for (uint64_t off = 0; off < m; b ^= 1) {
        const double t_a = now_ms();
        if (st->chunks.empty() || st->chunks.back().count == st->chunks.back().cap) {
            IngestChunk c;
            // capacity: at least this batch, and as much as the stream already holds (doubling), up to chunk_records - few
            // large blocks, which the caching allocator can hand to later large requests, instead of one block per batch
            uint64_t cap = 1ull << 20;
            while (cap < std::max(m - off, st->count) && cap < chunk_records) cap <<= 1;
            c.cap = std::min(cap, chunk_records);
            const uint64_t add = c.cap * 9;
            if ((st->max_bytes && st->bytes + add > st->max_bytes) || hipMalloc((void **)&c.d_pair, c.cap * 8) != hipSuccess) {
                undo();
                return "hipMalloc(record chunk): out of memory";
            }
            if (hipMalloc((void **)&c.d_bad, c.cap) != hipSuccess) {
                (void)hipFree(c.d_pair);
                undo();
                return "hipMalloc(flag bytes): out of memory";
            }
            st->bytes += add;
            st->peak_bytes = std::max(st->peak_bytes, st->bytes);
            st->chunks.push_back(c);
        }
        IngestChunk &c = st->chunks.back();
        const uint64_t cnt = std::min(std::min(slab, m - off), c.cap - c.count);
        const double t_b = now_ms();
        {
            auto wait_oldest = [&]() {
                if (snap_pending[b]) (void)hipEventSynchronize(evs.consumed[b]); // buffer b is the one about to be reused: its slab is the older one
            };
            const std::string e = table_reserve(kstream, st, cnt, poll_snapshots, wait_oldest); // (all table work lives on the ingest's stream)
            if (!e.empty()) {
                undo();
                return e;
            }
        }
        const double t_c = now_ms();
        IG_HIP(hipStreamWaitEvent(stream, evs.consumed[b], 0)); // the kernel that last read this slab buffer has finished
        IG_HIP(hipMemcpyAsync(st->d_slab[b], edges + off, cnt * sizeof(hb_edge), hipMemcpyHostToDevice, stream));
        IG_HIP(hipEventRecord(evs.copied[b], stream));
        IG_HIP(hipStreamWaitEvent(kstream, evs.copied[b], 0));
        hipLaunchKernelGGL(insert_kernel, dim3(grid_for(cnt)), dim3(256), 0, kstream, (const hb_edge *)st->d_slab[b], cnt, c.count, table_of(st), c.d_pair,
                           c.d_bad);
        IG_HIP(hipGetLastError());
        IG_HIP(hipMemcpyAsync(&st->h_counter[1 + b], st->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, kstream));
        IG_HIP(hipEventRecord(evs.consumed[b], kstream));
        launched += cnt;
        snap_launched[b] = launched;
        snap_pending[b] = true;
        st->unsynced += cnt;
        c.count += cnt;
        st->count += cnt;
        off += cnt;
        ms_alloc += t_b - t_a;
        ms_reserve += t_c - t_b;
        ms_issue += now_ms() - t_c;
    }
    const double t_s = now_ms();
    IG_HIP(hipStreamSynchronize(stream));
    IG_HIP(hipStreamSynchronize(kstream));
    poll_snapshots(); // everything has completed: the newest snapshot is exact and nothing is unknown any more
    if (trace)
        std::fprintf(stderr, "[hb ingest] append of %llu records: %.1f ms (chunk allocation %.1f, table reserve %.1f, issue %.1f, final wait %.1f); "
                             "%llu ids in %llu slots; so far %llu polls %.1f ms, %llu blocking read-backs %.1f ms\n", (unsigned long long)m, now_ms() - t_call,
                     ms_alloc, ms_reserve, ms_issue, now_ms() - t_s, (unsigned long long)st->npid_known, (unsigned long long)st->tab_slots,
                     (unsigned long long)st->trace_polls, st->trace_ms_poll, (unsigned long long)st->trace_reads, st->trace_ms_read);
    return "";
}

// the records held on the device -> hb_edge records on the host, appended to *out (the stream moves to the host path when
// the device cannot hold it); consumes the stream
std::string gpu_ingest_spill(void *stream_v, IngestStream *st, std::vector<hb_edge> *out)
{
    hipStream_t stream = (hipStream_t)stream_v;
    struct FreeStream {
        IngestStream *s;
        ~FreeStream() { s->free_all(); }
    } free_stream{st};
    if (!st->count) return "";
    std::string e = read_npid(stream, st);
    if (!e.empty()) return e;
    const uint64_t npid = st->npid_known;
    DevMem mem;
    u128 *d_key_of_pid = nullptr;
    IG_HIP(mem.alloc(&d_key_of_pid, std::max<uint64_t>(npid, 1)));
    hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(st->tab_slots)), dim3(256), 0, stream, (const u128 *)st->d_tab_keys, (const uint32_t *)st->d_tab_pids,
                       st->tab_slots, d_key_of_pid, npid);
    IG_HIP(hipGetLastError());
    const size_t at = out->size();
    out->resize(at + st->count);
    uint64_t base = 0, piece = 0;
    for (const IngestChunk &c : st->chunks) piece = std::max(piece, std::min<uint64_t>(c.count, 1ull << 22));
    hb_edge *d_rec = nullptr;
    IG_HIP(mem.alloc(&d_rec, std::max<uint64_t>(piece, 1)));
    for (const IngestChunk &c : st->chunks) {
        for (uint64_t o = 0; o < c.count; o += piece) {
            const uint64_t k = std::min(piece, c.count - o);
            hipLaunchKernelGGL(unpack_records_kernel, dim3(grid_for(k)), dim3(256), 0, stream, (const uint64_t *)c.d_pair + o, (const uint8_t *)c.d_bad + o, k,
                               (const u128 *)d_key_of_pid, d_rec);
            IG_HIP(hipGetLastError());
            IG_HIP(hipMemcpyAsync(out->data() + at + base + o, d_rec, k * sizeof(hb_edge), hipMemcpyDeviceToHost, stream));
            IG_HIP(hipStreamSynchronize(stream));
        }
        base += c.count;
    }
    return "";
}

std::string gpu_ingest_edges(void *stream_v, const hb_u128 *node_ids, uint64_t n_in, const hb_edge *edges, uint64_t m,
                             DenseGraph *out, DeviceCsr *keep, uint64_t *peak_bytes)
{
    if (keep) *keep = DeviceCsr{};
    if (m && !edges) return "edges == NULL with m > 0";
    IngestStream st;
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

// The reduction proper, from records that are already on the device (gpu_ingest_append); consumes the stream.
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
        IngestStream *st;
        uint64_t *out;
        ~Peak()
        {
            if (out) *out = std::max<uint64_t>(mem.peak, st->peak_bytes); // live bytes (what the caching allocator held: hb_stats.pool_peak_bytes)
        }
    } peak_guard{mem, st, peak_bytes};
    for (void *&p : st->d_slab) { // the staging buffers are no longer needed
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    mem.note(st->bytes);
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
        std::fprintf(stderr, "[hb ingest] %-34s %9.1f ms   (device bytes held %.2f GB, peak %.2f GB)\n", what, t - t_lap, mem.cur / 1e9, mem.peak / 1e9);
        t_lap = t;
    };

    // ---- the table's keys by pid; the table itself is no longer needed
    uint64_t npid = 0;
    u128 *d_key_of_pid = nullptr;
    if (st->tab_slots) {
        const std::string e = read_npid(stream, st);
        if (!e.empty()) return e;
        npid = st->npid_known;
        if (npid >= kMaxPids) return "too many nodes for the device ingest (2^31 endpoint ids)";
        IG_HIP(mem.alloc(&d_key_of_pid, std::max<uint64_t>(npid, 1)));
        hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(st->tab_slots)), dim3(256), 0, stream, (const u128 *)st->d_tab_keys,
                           (const uint32_t *)st->d_tab_pids, st->tab_slots, d_key_of_pid, npid);
        IG_HIP(hipGetLastError());
        IG_HIP(hipStreamSynchronize(stream));
        (void)hipFree(st->d_tab_keys);
        (void)hipFree(st->d_tab_pids);
        st->d_tab_keys = nullptr;
        st->d_tab_pids = nullptr;
        mem.unnote(st->tab_slots * 20);
        st->bytes -= st->tab_slots * 20;
        st->tab_slots = 0;
    }
    lap("endpoint keys by pid");

    // ---- node set: sorted unique u128 keys in d_ids, and the sid of every pid
    u128 *d_ids = nullptr;
    uint32_t *d_sid_of_pid = nullptr;
    uint64_t n = 0;
    IG_HIP(mem.alloc(&d_sid_of_pid, std::max<uint64_t>(npid, 1)));
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
        if (npid) {
            hipLaunchKernelGGL(sid_search_kernel, dim3(grid_for(npid)), dim3(256), 0, stream, (const u128 *)d_key_of_pid, npid, (const u128 *)d_ids, n,
                               d_sid_of_pid);
            IG_HIP(hipGetLastError());
        }
    } else if (npid) {
        // table keys are distinct: one 128-bit sort carrying the pid gives the ids in order and every pid's rank
        u128 *d_alt = nullptr;
        uint32_t *d_pid = nullptr, *d_pid_alt = nullptr;
        IG_HIP(mem.alloc(&d_alt, npid));
        IG_HIP(mem.alloc(&d_pid, npid));
        IG_HIP(mem.alloc(&d_pid_alt, npid));
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(npid)), dim3(256), 0, stream, d_pid, npid);
        IG_HIP(hipGetLastError());
        rocprim::double_buffer<u128> kb(d_key_of_pid, d_alt);
        rocprim::double_buffer<uint32_t> vb(d_pid, d_pid_alt);
        size_t bytes = 0;
        IG_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kb, vb, (size_t)npid, 0, 128, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::radix_sort_pairs(tmp, bytes, kb, vb, (size_t)npid, 0, 128, stream));
        hipLaunchKernelGGL(sid_scatter_kernel, dim3(grid_for(npid)), dim3(256), 0, stream, (const uint32_t *)vb.current(), npid, d_sid_of_pid);
        IG_HIP(hipGetLastError());
        IG_HIP(hipStreamSynchronize(stream));
        d_ids = kb.current();
        mem.release(kb.alternate());
        mem.release(d_pid);
        mem.release(d_pid_alt);
        d_key_of_pid = nullptr; // either it is d_ids now or it was the alternate buffer just released
        n = npid;
    }
    if (d_key_of_pid) {
        IG_HIP(hipStreamSynchronize(stream));
        mem.release(d_key_of_pid);
    }
    lap("node set, sid of every pid");
    if (n >= (1ull << 30)) return "too many nodes (n must be < 2^30)";
    const bool small_graph = !keep || m <= kKeepHostGraph; // (m_eff <= m: the host copy of the CSR is only made for small graphs)
    try {
        out->ids.resize(n);
        if (small_graph || n == 0 || m == 0) out->row_ptr.assign(n + 1, 0);
    } catch (const std::bad_alloc &) {
        return "out of host memory for the node set";
    }
    uint64_t *d_id_lo = nullptr; // (keep) handed to the caller with the CSR
    if (n) {
        hb_u128 *d_out_ids = nullptr;
        IG_HIP(mem.alloc(&d_out_ids, n));
        if (keep) IG_HIP(mem.alloc(&d_id_lo, n));
        hipLaunchKernelGGL(keys_to_ids_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const u128 *)d_ids, n, d_out_ids, d_id_lo);
        IG_HIP(hipGetLastError());
        IG_HIP(hipMemcpyAsync(out->ids.data(), d_out_ids, n * sizeof(hb_u128), hipMemcpyDeviceToHost, stream));
        IG_HIP(hipStreamSynchronize(stream));
        mem.release(d_out_ids);
    }
    if (d_ids) mem.release(d_ids);
    lap("ids to the host");
    if (n == 0 || m == 0) return "";

    // ---- sort keys (to sid, from sid, flag), chunk by chunk into one array; a chunk is freed once its records are mapped
    uint32_t nb = 1;
    while ((1ull << nb) <= n) nb++; // bits of n: every sid < n fits, and the all-ones field never is a sid
    uint64_t *d_key = nullptr, *d_key_alt = nullptr;
    IG_HIP(mem.alloc(&d_key, m));
    {
        uint64_t base = 0;
        for (IngestChunk &c : st->chunks) {
            if (c.count) {
                hipLaunchKernelGGL(pair_keys_kernel, dim3(grid_for(c.count)), dim3(256), 0, stream, (const uint64_t *)c.d_pair, (const uint8_t *)c.d_bad,
                                   c.count, (const uint32_t *)d_sid_of_pid, nb, d_key + base);
                IG_HIP(hipGetLastError());
                IG_HIP(hipStreamSynchronize(stream));
            }
            base += c.count;
            (void)hipFree(c.d_pair);
            (void)hipFree(c.d_bad);
            c.d_pair = nullptr;
            c.d_bad = nullptr;
            mem.unnote(c.cap * 9);
        }
        st->chunks.clear();
        st->bytes = 0;
    }
    mem.release(d_sid_of_pid);
    lap("pair keys, chunks freed");

    // ---- stable sort by (to, from) - bit 0, the flag, is NOT a sort bit: the first record of every pair heads its run
    IG_HIP(mem.alloc(&d_key_alt, m));
    uint64_t *d_sorted = nullptr, *d_other = nullptr;
    {
        rocprim::double_buffer<uint64_t> kb(d_key, d_key_alt);
        size_t bytes = 0;
        IG_HIP(rocprim::radix_sort_keys(nullptr, bytes, kb, (size_t)m, 1u, 1u + 2u * nb, stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::radix_sort_keys(tmp, bytes, kb, (size_t)m, 1u, 1u + 2u * nb, stream));
        IG_HIP(hipStreamSynchronize(stream));
        d_sorted = kb.current();
        d_other = kb.alternate();
    }
    lap("stable sort of the pairs");

    // ---- heads, flag filter, compaction into the other sort buffer; segments of <= 2^30 records (one rocPRIM call never
    // sees more items than fit 32-bit offsets)
    uint64_t m_eff = 0;
    {
        const uint64_t seg = 1ull << 30;
        for (uint64_t off = 0; off < m; off += seg) {
            const uint64_t cnt = std::min(seg, m - off);
            auto idx = rocprim::make_counting_iterator<uint64_t>(off);
            auto kept = rocprim::make_transform_iterator(idx, HeadKept{d_sorted});
            auto any = rocprim::make_transform_iterator(idx, HeadAny{d_sorted});
            size_t bytes = 0;
            IG_HIP(rocprim::reduce(nullptr, bytes, any, d_n, (uint64_t)0, (size_t)cnt, rocprim::plus<uint64_t>(), stream));
            IG_HIP(need_tmp(bytes));
            IG_HIP(rocprim::reduce(tmp, bytes, any, d_n, (uint64_t)0, (size_t)cnt, rocprim::plus<uint64_t>(), stream));
            uint64_t heads = 0, sel = 0;
            IG_HIP(hipMemcpyAsync(&heads, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            IG_HIP(hipStreamSynchronize(stream)); // heads read back before d_n is reused, the reduction done before tmp may move
            bytes = 0;
            IG_HIP(rocprim::select(nullptr, bytes, d_sorted + off, kept, d_other + m_eff, d_n, (size_t)cnt, stream));
            IG_HIP(need_tmp(bytes));
            IG_HIP(rocprim::select(tmp, bytes, d_sorted + off, kept, d_other + m_eff, d_n, (size_t)cnt, stream));
            IG_HIP(hipMemcpyAsync(&sel, d_n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            IG_HIP(hipStreamSynchronize(stream));
            out->m_unique += heads;
            m_eff += sel;
        }
    }
    mem.release(d_sorted);
    lap("heads, filter, compaction");

    // ---- the kept (to, from) keys, still ascending -> sources + row pointers
    uint32_t *d_src = nullptr;
    uint64_t *d_row_end = nullptr, *d_row_ptr = nullptr;
    IG_HIP(mem.alloc(&d_src, std::max<uint64_t>(m_eff, 1)));
    IG_HIP(mem.alloc(&d_row_end, n + 1));
    IG_HIP(mem.alloc(&d_row_ptr, n + 1));
    IG_HIP(hipMemsetAsync(d_row_end, 0, (n + 1) * sizeof(uint64_t), stream));
    if (m_eff) {
        hipLaunchKernelGGL(csr_from_keys_kernel, dim3(grid_for(m_eff)), dim3(256), 0, stream, (const uint64_t *)d_other, m_eff, nb, d_src, d_row_end);
        IG_HIP(hipGetLastError());
    }
    {
        size_t bytes = 0;
        IG_HIP(rocprim::inclusive_scan(nullptr, bytes, d_row_end, d_row_ptr, (size_t)(n + 1), MaxU64(), stream));
        IG_HIP(need_tmp(bytes));
        IG_HIP(rocprim::inclusive_scan(tmp, bytes, d_row_end, d_row_ptr, (size_t)(n + 1), MaxU64(), stream));
    }
    IG_HIP(hipStreamSynchronize(stream));
    mem.release(d_other);
    mem.release(d_row_end);
    lap("csr");
    const bool to_host = !keep || m_eff <= kKeepHostGraph;
    if (to_host) {
        try {
            out->row_ptr.assign(n + 1, 0);
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
        if (d_id_lo) {
            mem.disown(d_id_lo);
            keep->d_id_lo = d_id_lo;
        }
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


// ---- key order of the centrality stores on the device (hb_store_harmonic_results) -----------------------------------------------
namespace {
// the bincode varint encoding of a u128 (serialized.rs:86-92 -> bincode standard(): < 251 one byte; 251 + u16; 252 + u32; 253 + u64;
// 254 + u128, little endian) as hb_store.cpp holds it: 17 key bytes, zero padded, in two big-endian words + the last byte
__global__ __launch_bounds__(256) void store_keys_kernel(const hb_u128 *ids, uint64_t count, u128 *key, uint64_t *k2_index)
{
    HB_GRID_STRIDE(i, count)
    {
        const uint64_t lo = ids[i].lo, hi = ids[i].hi;
        uint8_t b[17];
#pragma unroll
        for (int k = 0; k < 17; k++) b[k] = 0;
        int n = 0;
        if (!hi && lo < 251) {
            b[0] = (uint8_t)lo;
        } else {
            if (!hi && lo < (1ull << 16)) b[0] = 251, n = 2;
            else if (!hi && lo < (1ull << 32)) b[0] = 252, n = 4;
            else if (!hi) b[0] = 253, n = 8;
            else b[0] = 254, n = 16;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < n) b[1 + k] = (uint8_t)((k < 8 ? lo >> (8 * k) : hi >> (8 * (k - 8))) & 0xFFu);
        }
        uint64_t k0 = 0, k1 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) k0 = (k0 << 8) | b[k];
#pragma unroll
        for (int k = 0; k < 8; k++) k1 = (k1 << 8) | b[8 + k];
        key[i] = ((u128)k0 << 64) | (u128)k1;
        k2_index[i] = ((uint64_t)b[16] << 56) | i;
    }
}
__global__ __launch_bounds__(256) void store_keys_pack_kernel(const u128 *key, const uint64_t *k2_index, uint64_t count, uint64_t *out3)
{
    HB_GRID_STRIDE(i, count)
    {
        const u128 k = key[i];
        out3[3 * i] = (uint64_t)(k >> 64);
        out3[3 * i + 1] = (uint64_t)k;
        out3[3 * i + 2] = k2_index[i];
    }
}
} // namespace

std::string gpu_store_keys(void *stream_v, const hb_u128 *ids, uint64_t count, StoreKey *sorted_out)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (!count) return "";
    if (count >= (1ull << 56)) return "too many entries";
    DevMem mem;
    hb_u128 *d_ids = nullptr;
    u128 *d_key = nullptr, *d_key2 = nullptr;
    uint64_t *d_w = nullptr, *d_w2 = nullptr, *d_out = nullptr;
    IG_HIP(mem.alloc(&d_ids, count));
    IG_HIP(mem.alloc(&d_key, count));
    IG_HIP(mem.alloc(&d_key2, count));
    IG_HIP(mem.alloc(&d_w, count));
    IG_HIP(mem.alloc(&d_w2, count));
    IG_HIP(hipMemcpyAsync(d_ids, ids, count * sizeof(hb_u128), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(store_keys_kernel, dim3(grid_for(count)), dim3(256), 0, stream, (const hb_u128 *)d_ids, count, d_key, d_w);
    IG_HIP(hipGetLastError());
    // LSD: first the 17th key byte (the top byte of the index word; ties keep the ascending index), then the first 16 bytes
    size_t b1 = 0, b2 = 0;
    IG_HIP(rocprim::radix_sort_pairs(nullptr, b1, (const uint64_t *)d_w, d_w2, (const u128 *)d_key, d_key2, (size_t)count, 56, 64, stream));
    IG_HIP(rocprim::radix_sort_pairs(nullptr, b2, (const u128 *)d_key2, d_key, (const uint64_t *)d_w2, d_w, (size_t)count, 0, 128, stream));
    char *tmp = nullptr;
    IG_HIP(mem.alloc(&tmp, std::max(b1, b2)));
    IG_HIP(rocprim::radix_sort_pairs(tmp, b1, (const uint64_t *)d_w, d_w2, (const u128 *)d_key, d_key2, (size_t)count, 56, 64, stream));
    IG_HIP(rocprim::radix_sort_pairs(tmp, b2, (const u128 *)d_key2, d_key, (const uint64_t *)d_w2, d_w, (size_t)count, 0, 128, stream));
    mem.release(d_ids);
    IG_HIP(mem.alloc(&d_out, 3 * count));
    hipLaunchKernelGGL(store_keys_pack_kernel, dim3(grid_for(count)), dim3(256), 0, stream, (const u128 *)d_key, (const uint64_t *)d_w, count, d_out);
    IG_HIP(hipGetLastError());
    IG_HIP(hipMemcpyAsync(sorted_out, d_out, count * sizeof(StoreKey), hipMemcpyDeviceToHost, stream));
    IG_HIP(hipStreamSynchronize(stream));
    return "";
}

} // namespace hb
