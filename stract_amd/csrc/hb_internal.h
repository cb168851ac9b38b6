// hb_internal.h - shared declarations of the host-side planner and the HIP side.
#ifndef HB_INTERNAL_H
#define HB_INTERNAL_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/hyperball.h"

namespace hb {

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the multi-GB index
// arrays of the planner are filled by parallel loops right after being sized (first touch by the
// threads that fill them instead of a serial zero fill)
template <class T>
struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <class U>
    NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t) { ::operator delete(p); }
    template <class U, class... A>
    void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(static_cast<A &&>(a)...);
    }
    template <class U>
    bool operator==(const NoInitAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
template <class T>
using uvec = std::vector<T, NoInitAlloc<T>>;

constexpr uint32_t kNone = 0xFFFFFFFFu;   // "no source" sentinel in the device src stream
constexpr uint32_t kRowAlign = 64;        // row-id alignment of level boundaries (rows per block tile)
constexpr uint32_t kDefaultChunk = 64;    // max sources per work row

static inline bool u128_less(const hb_u128 &a, const hb_u128 &b)
{
    return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo;
}
static inline bool u128_eq(const hb_u128 &a, const hb_u128 &b) { return a.hi == b.hi && a.lo == b.lo; }

// The reduced graph in ascending-NodeID ("sid") indexing: what the reference's
// host_nodes()/host_edges() semantics leave.
struct DenseGraph {
    uvec<hb_u128> ids;             // n, strictly ascending (uvec: sized, then filled by a copy from the device or from the caller - no zero fill of 1.6 GB at C4)
    std::vector<uint64_t> row_ptr; // n + 1, in-edges of sid v
    std::vector<uint32_t> src;     // m_eff, sids
    uint64_t m_input = 0, m_unique = 0;
};

// Device work layout produced by the planner.
struct Plan {
    uint64_t n = 0;          // real nodes
    uint64_t n_pad = 0;      // world * slice: virtual row ids start here
    uint64_t slice = 0;      // rows per owner slice (multiple of kRowAlign); == n_pad when world == 1
    uint64_t nv = 0;         // virtual rows (incl. padding rows)
    uint64_t m_eff = 0;      // real edges
    uint32_t chunk = kDefaultChunk;
    std::vector<uint32_t> order;      // device index -> sid (n_pad entries, kNone = padding row)
    std::vector<uint32_t> dev_of;     // sid -> device index
    uvec<uint64_t> row_ptr;           // (n_pad + nv) + 1 offsets into src
    uvec<uint32_t> src;               // device indices (real < n_pad <= virtual ids)
    std::vector<uint64_t> level_begin; // virtual level l = rows [level_begin[l], level_begin[l+1])
    uint64_t xcd_begin[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // level-1 rows of XCD group x = [xcd_begin[x], xcd_begin[x+1])
    int xcd_groups = 1;
    // layout statistics (hb_stats)
    uint64_t level1_edges = 0;       // real edges gathered by level-1 chunk rows
    uint64_t level1_rows = 0;        // level-1 chunk rows (without padding rows)
    uint64_t direct_edges = 0;       // real edges gathered directly by node rows
    uint64_t rows_with_in_edges = 0; // nodes with in-degree > 0
};

// XCD-group quotas of the flexible (hot / cold) level-1 chunks; shared by the host planner (hb_host.cpp) and the
// device planner (hb_plan.hip), which must produce identical plans.  Integer arithmetic only.
struct XcdQuota {
    uint64_t warm[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // load of the pinned warm-slice chunks per group
    uint64_t flex[2] = {0, 0};                   // total load of the hot (0) / cold (1) class
    uint64_t bound[2][8];                        // class prefix load at which group x starts
    static constexpr uint64_t load_of(uint32_t len) { return (uint64_t)len + 4; } // +4: per-row overhead in gather units
    void add(uint32_t key, uint32_t len, uint32_t warm_slices)
    {
        if (key >= 1 && key <= warm_slices) warm[key & 7u] += load_of(len);
        else flex[key == 0 ? 0 : 1] += load_of(len);
    }
    // Two stages, like dealing the chunks coldest first to the least loaded group: the COLD chunks (every gather a
    // miss, as slow as the warm ones) level warm + cold; the HOT chunks (L2 hits, cheap) then level the total.
    void finish()
    {
        uint64_t load[8];
        for (int x = 0; x < 8; x++) load[x] = warm[x];
        for (int c = 1; c >= 0; c--) { // cold (1) first, then hot (0)
            uint64_t total = flex[c];
            for (int x = 0; x < 8; x++) total += load[x];
            const uint64_t target = (total + 7) / 8;
            uint64_t q[8], sq = 0;
            for (int x = 0; x < 8; x++) {
                q[x] = target > load[x] ? target - load[x] : 0;
                sq += q[x];
            }
            if (sq == 0) {
                for (int x = 0; x < 8; x++) q[x] = 1;
                sq = 8;
            }
            uint64_t cum = 0;
            for (int x = 0; x < 8; x++) {
                bound[c][x] = (uint64_t)(((unsigned __int128)cum * flex[c]) / sq);
                cum += q[x];
            }
            for (int x = 0; x < 8; x++) { // what the class adds to every group (for the next stage)
                const uint64_t hi = x < 7 ? bound[c][x + 1] : flex[c];
                load[x] += hi - bound[c][x];
            }
        }
    }
    // The class (in slice / longer-first order) is cut into kStripes stripes of equal load and EVERY stripe is
    // shared out by the quotas, so that each group gets long and short chunks alike (contiguous ranges gave one
    // group all the short chunks, whose per-row overhead is higher than the load model says).
    static constexpr uint64_t kStripes = 4096;
    uint64_t stripe(int cls) const { return flex[cls] / kStripes ? flex[cls] / kStripes : 1; }
    // group of the flexible chunk whose class-prefix load (exclusive) is `prefix`
    int group_of(int cls, uint64_t prefix) const
    {
        const uint64_t scaled = (prefix % stripe(cls)) * kStripes;
        int gsel = 0;
        for (int x = 1; x < 8; x++)
            if (bound[cls][x] <= scaled) gsel = x;
        return gsel;
    }
};

// Planner knobs (hb_options.chunk / tune[3..5]).
struct PlanTune {
    uint32_t chunk = kDefaultChunk; // max sources per work row
    uint32_t band_w = 1u << 16;     // hottest band of the source index space, in counters (0 = no banding)
    uint32_t minc = 8;              // a band cut needs at least this many sources in the chunk
    uint32_t direct_max = 0;        // rows with at most this many sources are not split (0 = chunk)
    bool xcd_map = true;            // XCD-affine groups of level-1 chunks (HB_FLAG_NO_XCD_MAP clears it)
    uint32_t world = 1;             // destination partition: rows are laid out as `world` equal slices,
                                    // slice g = the nodes with sid % world == g
};

// --- hb_host.cpp ---------------------------------------------------------------------
// Reference ingest semantics: node set (store.rs:338-357), first-occurrence dedup
// (store.rs:313), then rel-flag filter (harmonic.rs:131).  Returns "" or an error text.
std::string ingest_edges(const hb_u128 *node_ids, uint64_t n, const hb_edge *edges, uint64_t m,
                         DenseGraph *out);
void keep_owned_rows(DenseGraph *g, uint64_t world, uint64_t rank);
// HB_FLAG_REFERENCE_TAIL: page-level records -> keys (source device row << 32 | target device row) of the records
// between two host nodes that pass the rel filter; all keys -> CSR by source
struct TailIndex; // id -> sid hash index over the (sorted, caller-owned) id array, built once per graph
TailIndex *tail_index_build(const hb_u128 *ids, uint64_t n);
void tail_index_free(TailIndex *t);
// a page-level document whose from_id is a host node id, kept until its segment ends (hb_host.cpp)
struct TailDoc {
    uint32_t from_sid;
    uint32_t pass; // rel_flags & SKIPPED_REL == 0
    hb_u128 to;
};
std::string tail_collect(const TailIndex *tix, uint64_t n, const hb_edge *recs, uint64_t count, std::vector<TailDoc> *open);
std::string tail_close_segment(const TailIndex *tix, const hb_u128 *ids, const uint32_t *dev_of, std::vector<TailDoc> *open,
                               std::vector<uint64_t> *keys);
std::string build_tail_csr(std::vector<uint64_t> *keys, uint64_t n_pad, std::vector<uint64_t> *ptr, std::vector<uint32_t> *to);
// --- hb_ingest.hip: the same reduction on the GPU (stream = hipStream_t); identical output
// keep != NULL: the CSR stays on the device (returned in *keep, owned by the caller) and out->row_ptr / out->src
// are only filled for small graphs (m_eff <= kKeepHostGraph, for hb_debug_copy_graph)
std::string gpu_ingest_edges(void *stream, const hb_u128 *node_ids, uint64_t n, const hb_edge *edges, uint64_t m,
                             DenseGraph *out, struct DeviceCsr *keep = nullptr, uint64_t *peak_bytes = nullptr);
constexpr uint64_t kKeepHostGraph = 1ull << 26;
// the two halves of gpu_ingest_edges, for streamed input (hb_append_edges): every endpoint of a batch goes through a device
// hash table (NodeID -> 32-bit provisional id) as the batch arrives, a record is kept as (from pid, to pid) + 1 flag byte
// = 9 bytes, in chunks; one reduction at hb_finalize
struct IngestChunk {
    uint64_t *d_pair = nullptr; // cap x (from pid | to pid << 32), stream order
    uint8_t *d_bad = nullptr;   // cap "rel_flags & SKIPPED_REL" bytes
    uint64_t count = 0, cap = 0;
};
struct IngestStream {
    std::vector<IngestChunk> chunks;
    uint64_t count = 0;         // records held
    uint64_t bytes = 0;         // device bytes of the chunks + the endpoint table
    uint64_t peak_bytes = 0;    // their high-water mark
    void *d_slab[2] = {nullptr, nullptr}; // H2D staging
    uint64_t slab_cap = 0;
    void *kstream = nullptr;    // hipStream_t of the table kernels (the copies run on the caller's stream)
    // endpoint table (hb_ingest.hip): open addressing, tab_slots (power of two) x { 16-byte key, 4-byte pid / state }
    void *d_tab_keys = nullptr;
    uint32_t *d_tab_pids = nullptr;
    uint64_t tab_slots = 0;
    void *d_counter = nullptr;               // pids handed out (device)
    unsigned long long *h_counter = nullptr; // pinned read-back word
    uint64_t npid_known = 0;                 // ... at the last read-back
    uint64_t unsynced = 0;                   // records launched since then (each can add two keys)
    uint64_t trace_polls = 0, trace_reads = 0; // HB_TRACE_INGEST: slow-path visits of the table-size check
    double trace_ms_poll = 0, trace_ms_read = 0;
    // limits (0 = none / default); the test hooks of hb_debug_set_ingest_limits lower them to reach the refusal and
    // spill paths at small sizes
    uint64_t max_records = 0;   // refuse to hold this many records or more ("too many records")
    uint64_t max_bytes = 0;     // treat chunk / table memory beyond this as a failed allocation ("out of memory")
    uint64_t chunk_records = 0; // records per chunk (default 2^27)
    void free_all();
};
std::string gpu_ingest_append(void *stream, IngestStream *st, const hb_edge *edges, uint64_t m);
// the records held on the device back as hb_edge records (appended to *out; rel_flags collapses to skipped-or-not); consumes *st
std::string gpu_ingest_spill(void *stream, IngestStream *st, std::vector<hb_edge> *out);
// consumes *st (freed on every path); *peak_bytes = high-water mark of the device memory the ingest held
std::string gpu_ingest_reduce(void *stream, const hb_u128 *node_ids, uint64_t n, IngestStream *st, DenseGraph *out, struct DeviceCsr *keep,
                              uint64_t *peak_bytes = nullptr);
std::string check_dense(const hb_u128 *sorted_ids, uint64_t n, const uint64_t *row_ptr,
                        const uint32_t *src, uint64_t m);
// out_degree[sid] over the local edges.
void count_out_degree(const uint64_t *row_ptr, const uint32_t *src, uint64_t n, std::vector<uint32_t> *deg);
// ranks of the kept results in store_harmonic's order (hb_ingest.hip); d_vals = per-node f64, < 0 = absent
std::string gpu_rank_results(void *stream, const double *d_vals, uint64_t n, uint64_t expect, uint64_t *ranks_out, uint64_t *order_out = nullptr,
                             uint64_t top = 0);
// Builds the device layout.  global_out_degree: per sid (already summed over ranks).
std::string build_plan(uint64_t n, const uint64_t *row_ptr, const uint32_t *src,
                       const std::vector<uint32_t> &global_out_degree, bool reorder, const PlanTune &tune,
                       Plan *plan);
// Tables for the estimator's linear-counting branch (hyperloglog.rs:4472-4476,
// :4505-4515): lc[v] = trunc(64 ln(64/v)) when that is <= 40, else 0xFF (v = 0..64).  Built with the
// host libm; returns false if some value sits too close to an integer/threshold to be
// libm-independent.
bool build_lc_table(uint8_t lc[68]);

// --- hb_plan.hip: the same plan built on the device ----------------------------------------------------------
// A CSR by destination that lives in device memory (sid indexing), e.g. the output of the GPU ingest.
struct DeviceCsr {
    uint64_t *d_row_ptr = nullptr; // n + 1
    uint32_t *d_src = nullptr;     // m
    uint64_t m = 0;
    uint64_t *d_id_lo = nullptr;   // n, optional: the low 64 bits of every NodeID in ascending-id order (the device ingest has them: the
                                   // state stage hashes them on the device instead of collecting 8 of every 16 host bytes and uploading them)
};
// Device buffers of a finished plan (hipMalloc'ed; the caller owns them).
struct DevicePlan {
    uint64_t *d_row_ptr = nullptr; // n_pad + nv + 1
    uint32_t *d_src = nullptr;     // src_len (+ 4 slack)
    uint64_t src_len = 0;
    uint32_t *d_order = nullptr;      // n_pad: device row -> sid, kNone = padding row
    uint32_t *d_dev_of = nullptr;     // n: sid -> device row
    uint32_t *d_outdeg_dev = nullptr; // n_pad: (global) out-degree per device row
    uint64_t m_global = 0;            // sum of the out-degrees
};
// Fills the host-side meta data of `plan` (sizes, level_begin, xcd_begin, statistics; its big vectors stay empty) and
// `out`.  d_outdeg_sid: global out-degree per sid.  Must produce exactly build_plan()'s layout.
// d_offsets[i] = sum of d_counts[0 .. i), i = 0 .. count; fails unless the total equals `expect`
std::string device_offsets(void *stream, uint32_t *d_counts, uint64_t count, uint64_t *d_offsets, uint64_t expect);
// d_offsets[0 .. count]: exclusive prefix sums of d_counts (count + 1 outputs)
std::string device_prefix(void *stream, const uint32_t *d_counts, uint64_t count, uint64_t *d_offsets);
// the transposed work-row graph (out_ptr: rows_total + 1 offsets, out_rows: `entries` reader rows) by one stable radix sort of
// (source, row) keys; "" or an error text - "out of memory ..." = the caller may fall back to the scatter kernels
std::string gpu_transpose_rows(void *stream, const uint64_t *d_row_ptr, const uint32_t *d_src, uint64_t rows_total, uint64_t entries, uint64_t *d_out_ptr,
                               uint32_t *d_out_rows);
// destination partition: drop the in-edges of the rows other ranks own from a device CSR (the device form of keep_owned_rows)
std::string gpu_keep_owned_rows(void *stream, DeviceCsr *csr, uint64_t n, uint64_t world, uint64_t rank);
std::string gpu_build_plan(void *stream, uint64_t n, const uint64_t *d_row_ptr, const uint32_t *d_src, const uint32_t *d_outdeg_sid,
                           bool reorder, const PlanTune &tune, Plan *plan, DevicePlan *out);

double now_ms();
// [r6] the two host-parallel loops of the result path, on the OpenMP team the process already has (hb_host.cpp): no thread is CREATED
// inside hb_finish / hb_result_copy any more.  Creating one needs the process' address-space lock (its stack is mmap'ed), and
// hb_load_webgraph's background unmapping of a 100 GB store holds that lock for seconds: the first hb_finish behind a real webgraph load
// took 419 ms at C4 against 4.7 ms steady, hb_result_copy 2.1 s (profiles/r06c_bench_default.err).
//   out[idx[k]] = val[k] for k < n; every idx occurs at most once
void host_scatter_f64(double *out, const uint32_t *idx, const double *val, uint64_t n);
// lo[s] = ids[s].lo (the half of a NodeID HyperLogLog::add_u128 hashes), on the OpenMP team
void host_gather_id_lo(const hb_u128 *ids, uint64_t n, uint64_t *lo);
//   the (id, value) pairs with src[sid] >= 0.0 in ascending sid order, at most cap of them; ids / vals may be NULL
// in_bits != NULL [r6]: src is the COMPACT image - one entry per sid whose bit is set, in sid order (hb_aux.hip.h "the compact result image")
void host_compact_results(const double *src, const hb_u128 *idsrc, uint64_t n, hb_u128 *ids, double *vals, uint64_t cap, const uint64_t *in_bits = nullptr);

// ---- store emission (hb_store.cpp; the key order may come from the device: hb_ingest.hip gpu_store_keys) ---------------------
// one entry of the sort: the 17 key bytes as two big-endian words + the last byte, so that integer order = byte order of
// the encodings (the first byte fixes the length: zero padding never decides an order); index into the caller's arrays
struct StoreKey {
    uint64_t k0, k1;
    uint64_t k2_index; // key byte 16 << 56 | index (< 2^56)
    bool operator<(const StoreKey &o) const
    {
        if (k0 != o.k0) return k0 < o.k0;
        if (k1 != o.k1) return k1 < o.k1;
        return k2_index < o.k2_index;
    }
    bool same_key(const StoreKey &o) const { return k0 == o.k0 && k1 == o.k1 && (k2_index >> 56) == (o.k2_index >> 56); }
    uint64_t index() const { return k2_index & ((1ull << 56) - 1); }
    static int len_of_first(uint8_t b0) { return b0 < 251 ? 1 : b0 == 251 ? 3 : b0 == 252 ? 5 : b0 == 253 ? 9 : 17; }
    int key_len() const { return len_of_first((uint8_t)(k0 >> 56)); }
    void key_bytes(uint8_t out[17]) const
    {
        for (int i = 0; i < 8; i++) out[i] = (uint8_t)(k0 >> (56 - 8 * i));
        for (int i = 0; i < 8; i++) out[8 + i] = (uint8_t)(k1 >> (56 - 8 * i));
        out[16] = (uint8_t)(k2_index >> 56);
    }
};

// std::vector whose resize / sized constructor leaves trivially-constructible elements UNINITIALISED: the arrays of the store path are
// filled at once by a copy from the device or by all host cores (value-initialising 4.4 GB on one thread first cost C4 two seconds)
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value)
    {
        ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using RawVec = std::vector<T, DefaultInitAllocator<T>>;
using StoreKeyVec = RawVec<StoreKey>;
// keys of `count` ids in ascending key-byte order (bincode varint encodings, serialized.rs:86-92), computed on the device: ids go
// up once, (key words, index) come back sorted - a 136-bit LSD radix sort (stable: one pass on the 17th byte, then 128 bits)
std::string gpu_store_keys(void *stream, const hb_u128 *ids, uint64_t count, StoreKey *sorted_out);
// store_harmonic (centrality/mod.rs:72-114) from keys that are already in order; *sorted is consumed
int store_harmonic_presorted(const char *output, StoreKeyVec *sorted, const double *centralities, const uint64_t *ranks, char *err, size_t err_len);
} // namespace hb
#endif
