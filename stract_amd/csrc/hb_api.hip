// hb_api.hip - C ABI (include/hyperball.h) + pass driver of the HyperBall library.
//
// Replaces HarmonicCentrality::calculate / calculate_centrality
// (crates/core/src/webgraph/centrality/harmonic.rs:215-287,292) behind a C boundary.
// Host orchestration only: every arithmetic step of the path runs in the gfx950 kernels
// of hb_kernels.hip.h.  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "hb_internal.h"
#include "hb_kernels.hip.h"
#include "hb_experiments.hip.h"
#include "hll64_tables.inc"

using namespace hb;

namespace {
thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};
} // namespace

struct hb_ctx {
    hb_options opt{};
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    std::string err;
    std::string arch;

    // host-side graph / plan
    std::vector<hb_edge> pending;  // hb_append_edges, host-ingest mode only
    // hb_append_edges, default: records are unpacked on the device as they arrive (2 x 16-byte endpoint keys + 1
    // flag byte each); nothing is buffered on the host
    IngestStream app;
    uint64_t lim_records = 0xFFFFFF00ull, lim_bytes = 0, lim_chunk = 0; // hb_debug_set_ingest_limits
    DenseGraph g;                  // ids kept; row_ptr/src kept only for hb_debug_copy_graph
    Plan plan;
    bool loaded = false, begun = false, finished = false;

    // device
    std::vector<DevBuf> allocs;
    uint64_t *d_row_ptr = nullptr;
    uint32_t *d_src = nullptr;
    uint16_t *d_src_jp = nullptr; // parallel to d_src: the sources' initial register (pass 0 streams it, hb_kernels.hip.h)
    uint4 *d_regs[2] = {nullptr, nullptr};
    uint4 *d_part = nullptr;
    uint32_t *d_bits[2] = {nullptr, nullptr};
    uint32_t *d_kdirty = nullptr;
    double *d_ksum = nullptr, *d_kerr = nullptr;
    uint64_t *d_size = nullptr;
    uint64_t *d_idlow = nullptr;
    uint32_t *d_dev_of = nullptr;
    uint32_t *d_sid_of = nullptr; // device row -> sid, kNone for padding rows
    uint32_t *d_outdeg = nullptr; // device row -> (global) out-degree
    uint64_t m_global = 0;        // edges of the whole graph (all ranks)
    uint64_t last_active = 0;     // out-degree sum of the nodes changed in the previous pass
    unsigned long long *d_counters = nullptr; // (max_passes + 1) * kCounterWords, striped (hb_kernels.hip.h)
    double *d_raw = nullptr, *d_bias = nullptr;
    uint8_t *d_lc = nullptr;
    // sparse (data-driven) tail passes: transposed work-row graph + worklists
    uint64_t *d_out_ptr = nullptr;
    uint32_t *d_out_rows = nullptr;
    uint32_t *d_touch = nullptr;
    uint32_t *d_seeds = nullptr, *d_heavy = nullptr;
    unsigned int *d_sparse_counts = nullptr;
    bool sparse_ok = false;
    uint64_t plan_entries = 0; // entries of all work rows' source lists
    unsigned long long *h_counters = nullptr; // pinned, kCounterWords words; [0..3] hold the stripe sums after a pass
    uint64_t bits_words = 0;
    uint64_t ksum_len = 0; // entries allocated for ksum (world * slice in RCCL mode)
    uint64_t slice_rows = 0;
    // changed-only exchange (HB_FLAG_CHANGED_ONLY): packed changed counters, popcounts / prefix of the bitmap words
    uint4 *d_pack = nullptr;
    uint32_t *d_wpop = nullptr;
    uint64_t *d_wprefix = nullptr;
    // edge partition + HB_FLAG_CHANGED_ONLY: rows the local merge changed, the ranks' bitmaps gathered, their union
    uint32_t *d_lbits = nullptr, *d_lbits_all = nullptr, *d_ubits = nullptr;
    uint64_t co_rows = 0;     // rows in the union of this pass
    bool ubits_valid = false; // d_ubits holds this pass' union (set by the exchange, consumed by the epilogue)
    std::vector<uint64_t> ex_off; // world + 1: first packed position of every rank's slice
    uint64_t wire_bytes = 0;      // counter bytes this rank received over the run (changed-only accounting)
    // reference-tail mode (HB_FLAG_REFERENCE_TAIL): the reference's changed-node machinery as written
    uint64_t *d_tail_ptr = nullptr; // page-level records by source device row (hb_load_tail_edges), n_pad + 1
    uint32_t *d_tail_to = nullptr;
    uint64_t tail_count = 0;
    std::vector<uint64_t> tail_keys; // records of the closed segments, mapped (hb_host.cpp tail_close_segment)
    std::vector<TailDoc> tail_open;  // documents of the segment being appended (hb_tail_segment_end closes it)
    bool tail_dirty = false;         // tail_keys differ from what d_tail_* hold: rebuilt by hb_begin
    TailIndex *tail_index = nullptr; // id -> sid index for the batches of tail records (built at the first batch)
    uint32_t *d_bloom = nullptr;    // new_changed_nodes of the last pass (U64BloomFilter), bloom_bits bits
    uint64_t bloom_bits = 0;
    unsigned long long *d_bloom_ones = nullptr; // [0] count_ones, [1] (low word) length of d_list
    uint32_t *d_list = nullptr;     // exact_changed_nodes as device rows, <= ref_threshold entries
    uint64_t ref_threshold = 0;     // exact_counting_threshold (harmonic.rs:228)
    bool exact_counting = false;    // harmonic.rs:231,277-279
    bool exact_valid = false;       // the previous pass filled exact_changed_nodes (ran with Some(..) or was a tail pass)
    bool stale = false;             // a tail pass has run: host-level edges may have been skipped, the bloom filter's
                                    // false positives are no longer results-inert

    // loop state
    uint64_t t = 0;
    int cur = 0; // d_regs[cur] = "old"
    bool has_changes = false;
    bool pending_local = false; // between hb_step_local and hb_step_finish
    uint64_t last_changed = 0;
    uint32_t max_passes = 4096;
    std::vector<hb_pass_stats> pstats;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // [5] = after the level-1 hub launch
    // edge partition with a communicator: the node rows are merged, all-reduced and finished in kOverlap row ranges - range
    // k's ncclAllReduce runs on comm_stream while range k + 1 is still being merged, its epilogue while k + 1 is reduced
    static constexpr int kOverlap = 4;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ov_merged[kOverlap] = {nullptr, nullptr, nullptr, nullptr}, ov_reduced[kOverlap] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t ov_lo[kOverlap + 1] = {0, 0, 0, 0, 0};
    int ov_ranges = 0; // ranges of the pending pass (0 = one launch over all rows, no overlap)
    uint32_t cur_mode = 0;

    hb_stats stats{};
    // results: one f64 per node in ascending-NodeID order, -1.0 = absent (centrality <= 0)
    double *d_out = nullptr;
    double *h_out = nullptr; // pinned, n entries
    uint64_t h_out_len = 0;
    uint64_t res_count = 0;
};

namespace {

int fail(hb_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define HB_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(c, HB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

#define HB_NCCL(call)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail(c, HB_ERR_RCCL, std::string(#call) + ": " + ncclGetErrorString(r_));      \
    } while (0)

template <typename T>
int dev_alloc(hb_ctx *c, T **out, size_t count)
{
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return fail(c, HB_ERR_NOMEM, "hipMalloc(" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    c->allocs.push_back({p, bytes});
    c->stats.device_bytes += bytes;
    *out = (T *)p;
    return HB_OK;
}

void free_graph_buffers(hb_ctx *c)
{
    for (auto &b : c->allocs) (void)hipFree(b.p);
    c->allocs.clear();
    c->stats.device_bytes = 0;
    c->d_row_ptr = nullptr;
    c->d_src = nullptr;
    c->d_src_jp = nullptr;
    c->d_regs[0] = c->d_regs[1] = nullptr;
    c->d_part = nullptr;
    c->d_bits[0] = c->d_bits[1] = nullptr;
    c->d_kdirty = nullptr;
    c->d_ksum = c->d_kerr = nullptr;
    c->d_size = nullptr;
    c->d_idlow = nullptr;
    c->d_dev_of = nullptr;
    c->d_sid_of = nullptr;
    c->d_outdeg = nullptr;
    c->d_counters = nullptr;
    c->d_raw = c->d_bias = nullptr;
    c->d_lc = nullptr;
    c->d_out = nullptr;
    c->d_pack = nullptr;
    c->d_wpop = nullptr;
    c->d_wprefix = nullptr;
    c->d_lbits = c->d_lbits_all = c->d_ubits = nullptr;
    c->ubits_valid = false;
    c->d_out_ptr = nullptr;
    c->d_out_rows = nullptr;
    c->d_touch = nullptr;
    c->d_seeds = c->d_heavy = nullptr;
    c->d_sparse_counts = nullptr;
    c->sparse_ok = false;
    c->d_tail_ptr = nullptr;
    c->d_tail_to = nullptr;
    c->tail_count = 0;
    std::vector<uint64_t>().swap(c->tail_keys);
    std::vector<TailDoc>().swap(c->tail_open);
    c->tail_dirty = false;
    tail_index_free(c->tail_index);
    c->tail_index = nullptr;
    c->d_bloom = nullptr;
    c->d_bloom_ones = nullptr;
    c->d_list = nullptr;
    if (c->h_out) (void)hipHostFree(c->h_out);
    c->h_out = nullptr;
    c->h_out_len = 0;
}

// hb_options.chunk / tune[3..5] -> planner knobs
PlanTune plan_tune(uint32_t chunk, const uint32_t *tune)
{
    PlanTune t;
    t.chunk = chunk ? chunk : kDefaultChunk;
    if (tune) {
        if (tune[3] == 1) t.band_w = 0;                     // banding off
        else if (tune[3] >= 4 && tune[3] < 31) t.band_w = 1u << tune[3];
        if (tune[4]) t.minc = tune[4];
        if (tune[5]) t.direct_max = tune[5];
    }
    return t;
}

bool multi_rank(const hb_ctx *c) { return c->opt.world_size > 1; }
// destination partition: this rank owns the rows (nodes) with sid % world == rank and holds all their
// in-edges; one all-gather of the owned counter slices per pass
// (also with a 1-rank communicator, HB_FLAG_RCCL_SELF: the grouped all-gathers run for real on one GPU)
bool dest_mode(const hb_ctx *c)
{
    return (c->opt.flags & HB_FLAG_DEST_PARTITION) && (multi_rank(c) || (c->opt.flags & HB_FLAG_RCCL_SELF));
}
// edge partition: every rank holds some in-edges of every row; one all-reduce(max) of all counters per pass
bool edge_partitioned(const hb_ctx *c) { return multi_rank(c) && !dest_mode(c); }
bool ref_tail(const hb_ctx *c) { return (c->opt.flags & HB_FLAG_REFERENCE_TAIL) != 0; }
bool unfused(const hb_ctx *c)
{
    return edge_partitioned(c) || (c->comm && !dest_mode(c)) || (c->opt.flags & HB_FLAG_UNFUSED) || ref_tail(c);
}

// Transposed work-row graph (who reads each node / virtual row), touch bitmap and seed lists for the
// sweep-mode passes; built on the device from the uploaded plan, prefix sum on the host.
int build_sparse_support(hb_ctx *c)
{
    const Plan &p = c->plan;
    c->sparse_ok = false;
    if (unfused(c) || multi_rank(c) || (c->opt.flags & HB_FLAG_NO_SPARSE) || p.n == 0) return HB_OK;
    const uint64_t rows_total = p.n_pad + p.nv;
    const uint64_t entries = c->plan_entries;
    int rc;
    uint32_t *d_count = nullptr;
    if ((rc = dev_alloc(c, &c->d_out_ptr, rows_total + 1))) return rc;
    if ((rc = dev_alloc(c, &c->d_out_rows, entries))) return rc;
    if ((rc = dev_alloc(c, &c->d_touch, c->bits_words + 64))) return rc; // + 64: a wave reads 64 words at a time
    if ((rc = dev_alloc(c, &c->d_seeds, p.n_pad))) return rc;
    if ((rc = dev_alloc(c, &c->d_heavy, p.n_pad))) return rc;
    if ((rc = dev_alloc(c, &c->d_sparse_counts, 64))) return rc;
    if ((rc = dev_alloc(c, &d_count, rows_total))) return rc; // stays allocated (small next to out_rows)
    HB_HIP(hipMemsetAsync(c->d_touch, 0, (c->bits_words + 64) * sizeof(uint32_t), c->stream));
    HB_HIP(hipMemsetAsync(d_count, 0, rows_total * sizeof(uint32_t), c->stream));
    const unsigned blocks = (unsigned)std::min<uint64_t>((rows_total * 4 + 255) / 256, (uint64_t)c->num_cu * 16);
    hipLaunchKernelGGL(hbk::transpose_count_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint64_t *)c->d_row_ptr,
                       (const uint32_t *)c->d_src, rows_total, d_count);
    HB_HIP(hipGetLastError());
    {   // out_ptr = exclusive prefix sums of the reader counts (rows_total + 1 entries), on the device
        std::string e = device_offsets((void *)c->stream, d_count, rows_total, c->d_out_ptr, entries);
        if (!e.empty()) return fail(c, HB_ERR_HIP, e);
    }
    HB_HIP(hipMemsetAsync(d_count, 0, rows_total * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(hbk::transpose_fill_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint64_t *)c->d_row_ptr,
                       (const uint32_t *)c->d_src, rows_total, (const uint64_t *)c->d_out_ptr, d_count, c->d_out_rows);
    HB_HIP(hipGetLastError());
    HB_HIP(hipStreamSynchronize(c->stream));
    c->sparse_ok = true;
    return HB_OK;
}

// ---- plan + upload (common tail of every load entry point) -------------------------------
// The reduced graph arrives either on the host (c->g.row_ptr / c->g.src) or already on the device (csr, e.g. from
// the GPU ingest).  Default: everything from here on happens on the device (hb_plan.hip); HB_FLAG_HOST_PLAN and the
// destination partition use the host planner (hb_host.cpp), which produces the same layout.
bool device_plan(const hb_ctx *c) { return !(c->opt.flags & HB_FLAG_HOST_PLAN); }

// destination partition: this rank keeps the in-edges of the rows it owns - in the host copy of the reduced graph (if one
// is kept) and in the device CSR (if the graph lives there)
int keep_owned(hb_ctx *c, DeviceCsr *csr)
{
    if (!dest_mode(c)) return HB_OK;
    const uint64_t world = (uint64_t)std::max(c->opt.world_size, 1), rank = (uint64_t)c->opt.rank, n = c->g.ids.size();
    if (c->g.row_ptr.size() == n + 1 && n) keep_owned_rows(&c->g, world, rank);
    if (csr && csr->d_row_ptr) {
        const std::string e = gpu_keep_owned_rows((void *)c->stream, csr, n, world, rank);
        if (!e.empty()) return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : HB_ERR_HIP, e);
    }
    return HB_OK;
}

__global__ __launch_bounds__(256) void idlow_kernel(const uint64_t *lo_by_sid, const uint32_t *order, uint64_t n_pad, uint64_t *idlow)
{
    const uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_pad) return;
    const uint32_t s = order[d];
    idlow[d] = s == kNone ? 0ull : lo_by_sid[s];
}

void adopt(hb_ctx *c, void *p, size_t bytes) // a buffer allocated elsewhere becomes the context's
{
    c->allocs.push_back({p, bytes});
    c->stats.device_bytes += bytes;
}

int plan_and_upload(hb_ctx *c, DeviceCsr *csr_in, uint64_t m_eff)
{
    const uint64_t n = c->g.ids.size();
    c->loaded = false;
    c->begun = c->finished = false;
    free_graph_buffers(c);
    c->stats.n = n;
    c->stats.m_input = c->g.m_input;
    c->stats.m_unique = c->g.m_unique;
    c->stats.m_eff = m_eff;
    double t0 = now_ms();
    const bool on_device = device_plan(c);
    // the input CSR on the device (uploaded here if it is not there yet); freed when the plan exists
    struct InputCsr {
        DeviceCsr d;
        ~InputCsr()
        {
            if (d.d_row_ptr) (void)hipFree(d.d_row_ptr);
            if (d.d_src) (void)hipFree(d.d_src);
        }
    } in;
    if (csr_in) {
        in.d = *csr_in;
        *csr_in = DeviceCsr{};
    }
    if (!on_device && n && c->g.row_ptr.size() != n + 1) {
        // host planner, but the reduced graph only exists on the device: bring it back
        try {
            c->g.row_ptr.resize(n + 1);
            c->g.src.resize(m_eff);
        } catch (const std::bad_alloc &) {
            return fail(c, HB_ERR_NOMEM, "out of host memory for the reduced graph");
        }
        HB_HIP(hipMemcpyAsync(c->g.row_ptr.data(), in.d.d_row_ptr, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        if (m_eff) HB_HIP(hipMemcpyAsync(c->g.src.data(), in.d.d_src, m_eff * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
    }
    if (n && !in.d.d_src) {
        HB_HIP(hipMalloc((void **)&in.d.d_src, std::max<uint64_t>(m_eff, 1) * sizeof(uint32_t)));
        if (m_eff) HB_HIP(hipMemcpyAsync(in.d.d_src, c->g.src.data(), m_eff * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (on_device) {
            HB_HIP(hipMalloc((void **)&in.d.d_row_ptr, (n + 1) * sizeof(uint64_t)));
            HB_HIP(hipMemcpyAsync(in.d.d_row_ptr, c->g.row_ptr.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        }
        in.d.m = m_eff;
    }
    // global out-degree (the device order must be identical on every rank): histogram on the device
    bool reorder = !(c->opt.flags & HB_FLAG_NO_REORDER);
    if (multi_rank(c) && !c->comm) reorder = false; // logical ranks without a communicator
    struct Tmp {
        uint32_t *d_deg = nullptr;
        uint64_t *d_lo = nullptr;
        ~Tmp()
        {
            if (d_deg) (void)hipFree(d_deg);
            if (d_lo) (void)hipFree(d_lo);
        }
    } tmp;
    if (n) {
        HB_HIP(hipMalloc((void **)&tmp.d_deg, n * sizeof(uint32_t)));
        HB_HIP(hipMemsetAsync(tmp.d_deg, 0, n * sizeof(uint32_t), c->stream));
        if (m_eff) {
            const unsigned blocks = (unsigned)std::min<uint64_t>((m_eff + 255) / 256, (uint64_t)c->num_cu * 16);
            hipLaunchKernelGGL(hbk::histogram_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint32_t *)in.d.d_src, m_eff, tmp.d_deg);
            HB_HIP(hipGetLastError());
        }
        if (c->comm) HB_NCCL(ncclAllReduce(tmp.d_deg, tmp.d_deg, n, ncclUint32, ncclSum, c->comm, c->stream));
    }
    PlanTune pt = plan_tune(c->opt.chunk, c->opt.tune);
    pt.xcd_map = !(c->opt.flags & HB_FLAG_NO_XCD_MAP);
    if (dest_mode(c)) pt.world = (uint32_t)std::max(c->opt.world_size, 1);
    // The two counter arrays are what the passes gather from at random: allocate them BEFORE the planner churns
    // through tens of GB of work memory, while the device heap can still back them with large contiguous
    // fragments (allocated after it, the same kernels ran 1-2 % slower: more TLB misses on the gathers).
    {
        const uint64_t w = pt.world > 1 ? pt.world : 1;
        const uint64_t n_pad_pre = (((n + w - 1) / w + kRowAlign - 1) / kRowAlign * kRowAlign) * w;
        int rc0;
        if ((rc0 = dev_alloc(c, &c->d_regs[0], n_pad_pre * 4))) return rc0;
        if ((rc0 = dev_alloc(c, &c->d_regs[1], n_pad_pre * 4))) return rc0;
    }
    std::vector<uint32_t> outdeg; // host planner only
    DevicePlan dp;
    if (on_device) {
        std::string perr = gpu_build_plan((void *)c->stream, n, in.d.d_row_ptr, in.d.d_src, tmp.d_deg, reorder, pt, &c->plan, &dp);
        if (!perr.empty()) {
            for (void *q : {(void *)dp.d_row_ptr, (void *)dp.d_src, (void *)dp.d_order, (void *)dp.d_dev_of, (void *)dp.d_outdeg_dev})
                if (q) (void)hipFree(q);
            (void)hipGetLastError();
            return fail(c, perr.find("memory") != std::string::npos ? HB_ERR_NOMEM : (perr.find("exhausted") != std::string::npos ? HB_ERR_LIMIT : HB_ERR_HIP), perr);
        }
    } else {
        outdeg.assign(n, 0);
        if (n) {
            HB_HIP(hipMemcpyAsync(outdeg.data(), tmp.d_deg, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
        }
        static const uint64_t zero = 0;
        std::string perr = build_plan(n, n ? c->g.row_ptr.data() : &zero, c->g.src.data(), outdeg, reorder, pt, &c->plan);
        if (!perr.empty()) return fail(c, perr.find("memory") != std::string::npos ? HB_ERR_NOMEM : HB_ERR_LIMIT, perr);
    }
    // the input CSR is no longer needed on the device
    if (in.d.d_row_ptr) (void)hipFree(in.d.d_row_ptr);
    if (in.d.d_src) (void)hipFree(in.d.d_src);
    in.d = DeviceCsr{};
    c->stats.ms_plan = now_ms() - t0;
    const Plan &p = c->plan;
    const uint64_t rows_total = p.n_pad + p.nv;
    const uint64_t src_len = on_device ? dp.src_len : p.src.size();
    c->plan_entries = src_len;
    c->stats.work_rows = rows_total;
    c->stats.virtual_rows = p.nv;
    c->stats.levels = p.level_begin.size() > 1 ? p.level_begin.size() - 1 : 0;
    c->stats.level1_edges = p.level1_edges;
    c->stats.level1_rows = p.level1_rows;
    c->stats.direct_edges = p.direct_edges;
    c->stats.rows_with_in_edges = p.rows_with_in_edges;

    // ---- device memory
    t0 = now_ms();
    c->bits_words = (rows_total + 31) / 32 + 2;
    int rc;
    if (on_device) {
        c->d_row_ptr = dp.d_row_ptr;
        c->d_src = dp.d_src;
        c->d_sid_of = dp.d_order;
        c->d_dev_of = dp.d_dev_of;
        c->d_outdeg = dp.d_outdeg_dev;
        adopt(c, dp.d_row_ptr, (rows_total + 2) * sizeof(uint64_t));
        adopt(c, dp.d_src, (src_len + 4) * sizeof(uint32_t));
        adopt(c, dp.d_order, std::max<uint64_t>(p.n_pad, 64) * sizeof(uint32_t));
        adopt(c, dp.d_dev_of, std::max<uint64_t>(n, 64) * sizeof(uint32_t));
        adopt(c, dp.d_outdeg_dev, std::max<uint64_t>(p.n_pad, 64) * sizeof(uint32_t));
        c->m_global = dp.m_global;
    } else {
        if ((rc = dev_alloc(c, &c->d_row_ptr, rows_total + 1))) return rc;
        if ((rc = dev_alloc(c, &c->d_src, src_len + 4))) return rc;
        if ((rc = dev_alloc(c, &c->d_dev_of, n))) return rc;
        if ((rc = dev_alloc(c, &c->d_sid_of, p.n_pad))) return rc;
        if ((rc = dev_alloc(c, &c->d_outdeg, p.n_pad))) return rc;
    }
    if ((rc = dev_alloc(c, &c->d_part, p.nv * 4))) return rc;
    if ((rc = dev_alloc(c, &c->d_bits[0], c->bits_words))) return rc;
    if ((rc = dev_alloc(c, &c->d_bits[1], c->bits_words))) return rc;
    if ((rc = dev_alloc(c, &c->d_kdirty, p.n_pad / 32 + 2))) return rc;
    // Kahan ownership: one contiguous slice of rows per rank (multiple of 64 rows)
    const uint64_t world = c->comm ? (uint64_t)c->opt.world_size : 1;
    c->slice_rows = dest_mode(c) ? p.slice : ((p.n_pad + world - 1) / world + 63) / 64 * 64;
    c->ksum_len = std::max<uint64_t>(c->slice_rows * (dest_mode(c) ? (uint64_t)std::max(c->opt.world_size, 1) : world), p.n_pad);
    if ((rc = dev_alloc(c, &c->d_ksum, c->ksum_len))) return rc;
    if ((rc = dev_alloc(c, &c->d_kerr, p.n_pad))) return rc;
    if ((rc = dev_alloc(c, &c->d_size, p.n_pad))) return rc;
    if ((rc = dev_alloc(c, &c->d_idlow, p.n_pad))) return rc;
    if ((rc = dev_alloc(c, &c->d_counters, ((size_t)c->max_passes + 1) * hbk::kCounterWords))) return rc;
    if ((rc = dev_alloc(c, &c->d_raw, HLL64_TABLE_LEN))) return rc;
    if ((rc = dev_alloc(c, &c->d_bias, HLL64_TABLE_LEN))) return rc;
    if ((rc = dev_alloc(c, &c->d_lc, 68))) return rc;
    if ((rc = dev_alloc(c, &c->d_out, n))) return rc;
    if (n) {
        if (hipHostMalloc((void **)&c->h_out, n * sizeof(double)) != hipSuccess)
            return fail(c, HB_ERR_NOMEM, "hipHostMalloc(result buffer) failed");
        c->h_out_len = n;
    }

    uint8_t lc[68];
    if (!build_lc_table(lc)) return fail(c, HB_ERR_INVALID, "host libm log() too close to a rounding boundary for the linear-counting table");
    HB_HIP(hipMemcpyAsync(c->d_raw, HLL64_RAW_ESTIMATE, sizeof(HLL64_RAW_ESTIMATE), hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipMemcpyAsync(c->d_bias, HLL64_BIAS, sizeof(HLL64_BIAS), hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipMemcpyAsync(c->d_lc, lc, 68, hipMemcpyHostToDevice, c->stream));
    if (!on_device) {
        c->stats.virtual_edges = p.row_ptr.empty() ? 0 : p.row_ptr[rows_total] - p.row_ptr[p.n_pad];
        HB_HIP(hipMemcpyAsync(c->d_row_ptr, p.row_ptr.data(), (rows_total + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        if (!p.src.empty())
            HB_HIP(hipMemcpyAsync(c->d_src, p.src.data(), p.src.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (p.n_pad)
            HB_HIP(hipMemcpyAsync(c->d_sid_of, p.order.data(), p.n_pad * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        std::vector<uint32_t> outdeg_dev(p.n_pad, 0);
        c->m_global = 0;
        for (uint64_t d = 0; d < p.n_pad; d++)
            if (p.order[d] != kNone) {
                outdeg_dev[d] = outdeg[p.order[d]];
                c->m_global += outdeg[p.order[d]];
            }
        if (p.n_pad)
            HB_HIP(hipMemcpyAsync(c->d_outdeg, outdeg_dev.data(), p.n_pad * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (n) HB_HIP(hipMemcpyAsync(c->d_dev_of, p.dev_of.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream)); // outdeg_dev goes out of scope
    } else {
        uint64_t ends[2] = {0, 0};
        HB_HIP(hipMemcpyAsync(&ends[0], c->d_row_ptr + p.n_pad, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipMemcpyAsync(&ends[1], c->d_row_ptr + rows_total, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        c->stats.virtual_edges = ends[1] - ends[0];
    }
    // low 64 bits of every NodeID in device order (HyperLogLog::add_u128 hashes only those, hyperloglog.rs:4398-4400)
    if (n) {
        std::vector<uint64_t> lo(n);
        for (uint64_t s = 0; s < n; s++) lo[s] = c->g.ids[s].lo;
        HB_HIP(hipMalloc((void **)&tmp.d_lo, n * sizeof(uint64_t)));
        HB_HIP(hipMemcpyAsync(tmp.d_lo, lo.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(idlow_kernel, dim3((unsigned)((p.n_pad + 255) / 256)), dim3(256), 0, c->stream, (const uint64_t *)tmp.d_lo,
                           (const uint32_t *)c->d_sid_of, p.n_pad, c->d_idlow);
        HB_HIP(hipGetLastError());
        HB_HIP(hipStreamSynchronize(c->stream));
    }
    if (src_len && !(c->opt.flags & HB_FLAG_NO_INIT_PASS)) {
        if ((rc = dev_alloc(c, &c->d_src_jp, src_len + 4))) return rc;
        const unsigned blocks = (unsigned)std::min<uint64_t>((src_len + 255) / 256, (uint64_t)c->num_cu * 16);
        hipLaunchKernelGGL(hbk::src_jp_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint32_t *)c->d_src, src_len, (const uint64_t *)c->d_idlow,
                           (const uint32_t *)c->d_sid_of, p.n_pad, c->d_src_jp);
        HB_HIP(hipGetLastError());
        HB_HIP(hipStreamSynchronize(c->stream));
    }
    if ((rc = build_sparse_support(c))) return rc;
    // the plan's big host arrays are no longer needed
    decltype(c->plan.row_ptr)().swap(c->plan.row_ptr);
    decltype(c->plan.src)().swap(c->plan.src);
    c->stats.ms_h2d = now_ms() - t0;
    c->loaded = true;
    return HB_OK;
}

// sid -> device row on the host (debug exports): downloaded on first use when the plan was built on the device
int need_host_dev_of(hb_ctx *c)
{
    const uint64_t n = c->plan.n;
    if (c->plan.dev_of.size() == n) return HB_OK;
    c->plan.dev_of.resize(n);
    if (n) {
        HB_HIP(hipMemcpyAsync(c->plan.dev_of.data(), c->d_dev_of, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
    }
    return HB_OK;
}

// ---- kernel dispatch ----------------------------------------------------------------------
// dense pass kernel: template instance from the run-time choices
template <bool REAL, bool FUSED>
void launch_dense(hb_ctx *c, const hbk::PassParams &pp, bool stats, int unroll, dim3 grid, bool init, bool epi4)
{
    hipStream_t s = c->stream;
    if (init && !stats) { // pass 0: the sources' initial registers stream in with the edge list (hb_kernels.hip.h)
        if constexpr (REAL && FUSED) {
            if (epi4) {
                if (unroll == 2) hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 2, true, true>), grid, dim3(256), 0, s, pp);
                else hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 4, true, true>), grid, dim3(256), 0, s, pp);
                return;
            }
        }
        if (unroll == 2) hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 2, true>), grid, dim3(256), 0, s, pp);
        else hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 4, true>), grid, dim3(256), 0, s, pp);
        return;
    }
    if constexpr (REAL && FUSED) {
        if (epi4 && !stats) { // once-per-row estimator / Kahan epilogue (default for the fused node rows)
            if (unroll == 1) hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 1, false, true>), grid, dim3(256), 0, s, pp);
            else if (unroll == 2) hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 2, false, true>), grid, dim3(256), 0, s, pp);
            else hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, false, 4, false, true>), grid, dim3(256), 0, s, pp);
            return;
        }
    }
#define HB_LAUNCH(ST, UN) hipLaunchKernelGGL((hbk::pass_kernel<REAL, FUSED, ST, UN>), grid, dim3(256), 0, s, pp)
    if (stats) {
        if (unroll == 1) HB_LAUNCH(true, 1);
        else if (unroll == 2) HB_LAUNCH(true, 2);
        else HB_LAUNCH(true, 4);
    } else {
        if (unroll == 1) HB_LAUNCH(false, 1);
        else if (unroll == 2) HB_LAUNCH(false, 2);
        else HB_LAUNCH(false, 4);
    }
#undef HB_LAUNCH
}

void launch_pass(hb_ctx *c, const hbk::PassParams &pp, bool real, bool frontier, bool fused)
{
    const bool stats = (c->opt.flags & HB_FLAG_PASS_STATS) != 0;
    int unroll = (int)(c->opt.tune[1] & 0xFFu);
    const bool epi4 = !(c->opt.tune[1] & 0x100u); // tune[1] bit 8: the per-tile epilogue (measurement switch)
    // default: 16 gathers in flight per quad for the hub chunks (pure gather loops); 8 for the node rows,
    // whose fused estimator/Kahan epilogue needs the registers (unroll 4 drops them to 3 waves/SIMD)
    if (unroll != 1 && unroll != 2 && unroll != 4) unroll = real ? 2 : 4;
    const uint64_t ntiles = (pp.row_hi - pp.row_lo + 63) / 64;
    if (ntiles == 0) return;
    // workgroups per CU: low byte of tune[0] = node rows, second byte = hub chunks (0 = default).  Measured
    // (profiles/r02s_sweep_bpc_*): the hub-chunk gather loop is fastest with only 2 workgroups (8 waves) per CU - each
    // quad already keeps 16 gathers in flight, more waves only add contention (dense pass -12 % at C3, -16 % at C4);
    // the bitmap pass has dependent bit tests in front of the gathers and wants 4.  Node rows: many small
    // workgroups, the hardware scheduler levels the uneven tiles (1.13 -> 1.05 ms at C3).
    uint32_t bpc = real ? (c->opt.tune[0] & 0xFFu) : ((c->opt.tune[0] >> 8) & 0xFFu);
    if (!bpc) bpc = real ? (frontier ? 32u : 64u) : (frontier ? 4u : 2u);
    uint64_t blocks = std::min<uint64_t>(ntiles, (uint64_t)c->num_cu * bpc);
    if (pp.xcd_map) blocks = std::max<uint64_t>((blocks + 7) / 8 * 8, 8); // 8 queues, equal shares of the grid
    const bool init = c->t == 0 && !frontier && pp.src_jp != nullptr && !(c->opt.flags & HB_FLAG_NO_INIT_PASS) && unroll != 1;
    if (init && !real && !((c->opt.tune[0] >> 8) & 0xFFu)) {
        // pass 0 streams its sources: no L2 window to protect, the scratch-counter updates want every wave the CU can hold
        blocks = std::min<uint64_t>(ntiles, (uint64_t)c->num_cu * 8);
        if (pp.xcd_map) blocks = std::max<uint64_t>((blocks + 7) / 8 * 8, 8);
    }
    dim3 grid((unsigned)blocks);
    if (frontier) {
        // the bitmap pass: all indices / all bit words / needed gathers of a row as three batched round trips
        hipStream_t st = c->stream;
#define HB_FRONT(R, F) \
    do { \
        if (stats) hipLaunchKernelGGL((hbk::frontier_kernel<R, F, true, (R ? 4 : 16)>), grid, dim3(256), 0, st, pp); \
        else hipLaunchKernelGGL((hbk::frontier_kernel<R, F, false, (R ? 4 : 16)>), grid, dim3(256), 0, st, pp); \
    } while (0)
        if (real && fused) HB_FRONT(true, true);
        else if (real) HB_FRONT(true, false);
        else HB_FRONT(false, false);
#undef HB_FRONT
        return;
    }
    if (real) {
        if (fused) launch_dense<true, true>(c, pp, stats, unroll, grid, init, epi4);
        else launch_dense<true, false>(c, pp, stats, unroll, grid, init, false);
    } else {
        launch_dense<false, false>(c, pp, stats, unroll, grid, init, false);
    }
}

hbk::PassParams make_params(hb_ctx *c)
{
    const Plan &p = c->plan;
    hbk::PassParams pp{};
    pp.row_ptr = c->d_row_ptr;
    pp.src = c->d_src;
    pp.src_jp = c->d_src_jp;
    pp.rd = c->d_regs[c->cur];
    pp.wr = c->d_regs[c->cur ^ 1];
    pp.part = c->d_part;
    pp.bits_rd = c->d_bits[c->cur];
    pp.bits_wr = c->d_bits[c->cur ^ 1];
    pp.kdirty = c->d_kdirty;
    pp.ksum = c->d_ksum;
    pp.kerr = c->d_kerr;
    pp.size = c->d_size;
    pp.counters = c->d_counters + (size_t)hbk::kCounterWords * c->t;
    pp.outdeg = c->d_outdeg;
    pp.raw = c->d_raw;
    pp.bias = c->d_bias;
    pp.lc = c->d_lc;
    pp.n = p.n;
    pp.n_pad = p.n_pad;
    if (c->comm || dest_mode(c)) {
        pp.slice_lo = (uint64_t)c->opt.rank * c->slice_rows;
        pp.slice_hi = std::min<uint64_t>(pp.slice_lo + c->slice_rows, p.n_pad);
    } else {
        pp.slice_lo = 0;
        pp.slice_hi = p.n_pad;
    }
    pp.t_plus_1 = (double)(c->t + 1);
    return pp;
}

bool changed_only(const hb_ctx *c) { return dest_mode(c) && (c->opt.flags & HB_FLAG_CHANGED_ONLY); }
// plain edge partition on a communicator: pipeline merge / all-reduce / epilogue over row ranges (tune[1] bit 12 = off)
bool edge_overlap(const hb_ctx *c)
{
    return c->comm && !dest_mode(c) && !ref_tail(c) && !(c->opt.flags & HB_FLAG_CHANGED_ONLY) && !(c->opt.tune[1] & 0x1000u) && c->comm_stream;
}
// edge partition (all-reduce) with HB_FLAG_CHANGED_ONLY: only the rows some rank's local merge changed are exchanged
bool edge_changed_only(const hb_ctx *c)
{
    return (c->opt.flags & HB_FLAG_CHANGED_ONLY) && !dest_mode(c) && !ref_tail(c) && (multi_rank(c) || c->comm);
}

int edge_co_alloc(hb_ctx *c)
{
    if (c->d_lbits) return HB_OK;
    const Plan &p = c->plan;
    const uint64_t words = p.n_pad / 32;
    const uint64_t world = (uint64_t)std::max(c->opt.world_size, 1);
    int rc;
    if ((rc = dev_alloc(c, &c->d_lbits, words + 2))) return rc;
    if ((rc = dev_alloc(c, &c->d_lbits_all, world * words + 2))) return rc;
    if ((rc = dev_alloc(c, &c->d_ubits, words + 2))) return rc;
    if (!c->d_pack) {
        if ((rc = dev_alloc(c, &c->d_pack, p.n_pad * 4))) return rc;
        if ((rc = dev_alloc(c, &c->d_wpop, words + 1))) return rc;
        if ((rc = dev_alloc(c, &c->d_wprefix, words + 2))) return rc;
    }
    HB_HIP(hipMemsetAsync(c->d_lbits, 0, (words + 2) * 4, c->stream));
    return HB_OK;
}

// d_ubits holds the union: positions of its rows, their number (one read-back), this rank's rows packed
int edge_co_pack(hb_ctx *c)
{
    const Plan &p = c->plan;
    const uint64_t words = p.n_pad / 32;
    hbk::PassParams pp = make_params(c);
    if (words) {
        const unsigned blocks = (unsigned)std::min<uint64_t>((words + 255) / 256, (uint64_t)c->num_cu * 8);
        hipLaunchKernelGGL(hbk::popcount_words_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint32_t *)c->d_ubits, words, c->d_wpop);
        HB_HIP(hipGetLastError());
    }
    const std::string e = device_prefix((void *)c->stream, c->d_wpop, words, c->d_wprefix);
    if (!e.empty()) return fail(c, HB_ERR_HIP, e);
    c->co_rows = 0;
    HB_HIP(hipMemcpyAsync(&c->co_rows, c->d_wprefix + words, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    if (c->co_rows) {
        hipLaunchKernelGGL(hbk::pack_changed_kernel, dim3((unsigned)((p.n_pad * 4 + 255) / 256)), dim3(256), 0, c->stream, (const uint4 *)pp.wr,
                           (const uint32_t *)c->d_ubits, (const uint64_t *)c->d_wprefix, (uint64_t)0, p.n_pad, c->d_pack);
        HB_HIP(hipGetLastError());
    }
    return HB_OK;
}

int edge_co_unpack(hb_ctx *c)
{
    const Plan &p = c->plan;
    hbk::PassParams pp = make_params(c);
    if (c->co_rows) {
        hipLaunchKernelGGL(hbk::unpack_rows_kernel, dim3((unsigned)((p.n_pad * 4 + 255) / 256)), dim3(256), 0, c->stream, pp.wr, (const uint32_t *)c->d_ubits,
                           (const uint64_t *)c->d_wprefix, p.n_pad, (const uint4 *)c->d_pack);
        HB_HIP(hipGetLastError());
    }
    c->ubits_valid = true;
    return HB_OK;
}

// changed-only exchange, step 1 (the changed bits of ALL slices are in bits_wr): prefix sums over the bitmap words,
// the packed position where every rank's run starts, and this rank's changed rows packed at their place
int exchange_pack(hb_ctx *c)
{
    const Plan &p = c->plan;
    const uint64_t words = p.n_pad / 32, S = c->slice_rows;
    const uint64_t world = (uint64_t)std::max(c->opt.world_size, 1), r = (uint64_t)c->opt.rank;
    int rc;
    if (!c->d_pack) {
        if ((rc = dev_alloc(c, &c->d_pack, p.n_pad * 4))) return rc;
        if ((rc = dev_alloc(c, &c->d_wpop, words + 1))) return rc;
        if ((rc = dev_alloc(c, &c->d_wprefix, words + 2))) return rc;
    }
    hbk::PassParams pp = make_params(c);
    if (words) {
        const unsigned blocks = (unsigned)std::min<uint64_t>((words + 255) / 256, (uint64_t)c->num_cu * 8);
        hipLaunchKernelGGL(hbk::popcount_words_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint32_t *)pp.bits_wr, words, c->d_wpop);
        HB_HIP(hipGetLastError());
    }
    const std::string e = device_prefix((void *)c->stream, c->d_wpop, words, c->d_wprefix);
    if (!e.empty()) return fail(c, HB_ERR_HIP, e);
    c->ex_off.assign(world + 1, 0);
    for (uint64_t k = 0; k <= world; k++)
        HB_HIP(hipMemcpyAsync(&c->ex_off[k], c->d_wprefix + std::min<uint64_t>(k * S, p.n_pad) / 32, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    const uint64_t lo = r * S, hi = std::min<uint64_t>(lo + S, p.n_pad);
    if (hi > lo) {
        hipLaunchKernelGGL(hbk::pack_changed_kernel, dim3((unsigned)(((hi - lo) * 4 + 255) / 256)), dim3(256), 0, c->stream, (const uint4 *)pp.wr,
                           (const uint32_t *)pp.bits_wr, (const uint64_t *)c->d_wprefix, lo, hi, c->d_pack);
        HB_HIP(hipGetLastError());
    }
    c->wire_bytes += (c->ex_off[world] - (c->ex_off[r + 1] - c->ex_off[r])) * 64 + (p.n_pad - (hi - lo)) / 8;
    return HB_OK;
}

// step 3 (the packed runs of all ranks are in d_pack): scatter the foreign ones
int exchange_unpack(hb_ctx *c)
{
    const Plan &p = c->plan;
    const uint64_t S = c->slice_rows, r = (uint64_t)c->opt.rank;
    hbk::PassParams pp = make_params(c);
    const uint64_t lo = r * S, hi = std::min<uint64_t>(lo + S, p.n_pad);
    const uint64_t ranges[2][2] = {{0, lo}, {hi, p.n_pad}};
    for (auto &rg : ranges) {
        if (rg[1] <= rg[0]) continue;
        hipLaunchKernelGGL(hbk::unpack_changed_kernel, dim3((unsigned)(((rg[1] - rg[0]) * 4 + 255) / 256)), dim3(256), 0, c->stream, pp.wr, pp.rd,
                           (const uint32_t *)pp.bits_wr, pp.bits_rd, (const uint64_t *)c->d_wprefix, rg[0], rg[1], (const uint4 *)c->d_pack);
        HB_HIP(hipGetLastError());
    }
    return HB_OK;
}

// ---- reference-tail mode ---------------------------------------------------------------------------------
// bloom/src/lib.rs:38-41
uint64_t bloom_num_bits(uint64_t estimated_items, double fp)
{
    const double ln2 = std::log(2.0);
    return (uint64_t)std::ceil((double)estimated_items * std::log(fp) / (-8.0 * (ln2 * ln2)));
}
// bloom/src/lib.rs:108-123: the logarithm is cast to i64 BEFORE the multiplication; a negative product -> 0
uint64_t bloom_estimate_card(uint64_t num_bits, uint64_t num_ones)
{
    if (num_ones == 0 || num_bits == 0) return 0;
    if (num_ones == num_bits) return ~0ull;
    const int64_t l = (int64_t)std::log(1.0 - (double)num_ones / (double)num_bits);
    const int64_t v = -(int64_t)num_bits * l;
    return v < 0 ? 0 : (uint64_t)v;
}

// update_changed_counters (harmonic.rs:75-114): counters.new starts as the clone of counters.old (Counters::step),
// the changed nodes push their OLD counter along the page-level records; everything else (changed bits, sizes,
// Kahan) is the unfused epilogue of step_finish, like after any other pass
int tail_pass(hb_ctx *c, const hbk::PassParams &pp)
{
    const Plan &p = c->plan;
    c->cur_mode = 3;
    c->stale = true;
    unsigned int *d_len = (unsigned int *)(c->d_bloom_ones + 1);
    HB_HIP(hipMemcpyAsync(pp.wr, pp.rd, p.n_pad * 64, hipMemcpyDeviceToDevice, c->stream));
    HB_HIP(hipMemsetAsync(d_len, 0, sizeof(unsigned int), c->stream));
    const unsigned blocks = (unsigned)std::min<uint64_t>(std::max<uint64_t>(p.n_pad / 256, 1), (uint64_t)c->num_cu * 8);
    hipLaunchKernelGGL(hbk::changed_list_kernel, dim3(blocks), dim3(256), 0, c->stream, pp.bits_rd, p.n_pad, c->d_list, d_len,
                       (uint32_t)(c->ref_threshold + 1));
    HB_HIP(hipEventRecord(c->ev[5], c->stream));
    HB_HIP(hipEventRecord(c->ev[1], c->stream));
    const unsigned tblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((c->last_changed + 3) / 4, (uint64_t)c->num_cu * 8));
    hipLaunchKernelGGL(hbk::tail_merge_kernel, dim3(tblocks), dim3(256), 0, c->stream, (const uint32_t *)c->d_list, (const unsigned int *)d_len,
                       (const uint64_t *)c->d_tail_ptr, (const uint32_t *)c->d_tail_to, (const uint32_t *)pp.rd, (uint32_t *)pp.wr);
    HB_HIP(hipEventRecord(c->ev[2], c->stream));
    HB_HIP(hipGetLastError());
    c->pending_local = true;
    return HB_OK;
}

// after a pass: new_changed_nodes as the reference builds it, and the exact-counting switch (harmonic.rs:273-279)
int reference_changed_state(hb_ctx *c, const uint32_t *bits_changed, uint64_t changed)
{
    const Plan &p = c->plan;
    const bool tracked = c->cur_mode == 3 || c->exact_counting; // this pass ran with Some(&mut exact_changed_nodes)
    if (!c->exact_counting || c->stale) {
        const uint64_t words = (c->bloom_bits + 31) / 32;
        HB_HIP(hipMemsetAsync(c->d_bloom, 0, (words + 1) * 4, c->stream));
        HB_HIP(hipMemsetAsync(c->d_bloom_ones, 0, sizeof(unsigned long long), c->stream));
        unsigned long long ones = 0;
        if (changed && c->bloom_bits) {
            const unsigned blocks = (unsigned)std::min<uint64_t>(std::max<uint64_t>(p.n_pad / 256, 1), (uint64_t)c->num_cu * 8);
            hipLaunchKernelGGL(hbk::bloom_insert_kernel, dim3(blocks), dim3(256), 0, c->stream, bits_changed, (const uint64_t *)c->d_idlow, p.n_pad,
                               c->bloom_bits, c->d_bloom);
            if (!c->exact_counting) {
                const unsigned cblocks = (unsigned)std::min<uint64_t>(std::max<uint64_t>(words / 256, 1), (uint64_t)c->num_cu * 8);
                hipLaunchKernelGGL(hbk::bloom_count_kernel, dim3(cblocks), dim3(256), 0, c->stream, (const uint32_t *)c->d_bloom, words, c->d_bloom_ones);
                HB_HIP(hipMemcpyAsync(&ones, c->d_bloom_ones, sizeof(ones), hipMemcpyDeviceToHost, c->stream));
                HB_HIP(hipStreamSynchronize(c->stream));
            }
            HB_HIP(hipGetLastError());
        }
        if (!c->exact_counting && bloom_estimate_card(c->bloom_bits, ones) <= c->ref_threshold) c->exact_counting = true;
    }
    c->exact_valid = tracked;
    return HB_OK;
}

// tail_keys -> d_tail_ptr / d_tail_to (hb_begin, when records were given since the last upload)
int upload_tail_index(hb_ctx *c)
{
    const Plan &p = c->plan;
    std::vector<uint64_t> ptr;
    std::vector<uint32_t> to;
    const std::string e = build_tail_csr(&c->tail_keys, p.n_pad, &ptr, &to);
    if (!e.empty()) return fail(c, HB_ERR_NOMEM, e);
    for (void *old : {(void *)c->d_tail_ptr, (void *)c->d_tail_to}) {
        if (!old) continue;
        for (size_t i = 0; i < c->allocs.size(); i++)
            if (c->allocs[i].p == old) {
                c->stats.device_bytes -= c->allocs[i].bytes;
                (void)hipFree(old);
                c->allocs.erase(c->allocs.begin() + (long)i);
                break;
            }
    }
    c->d_tail_ptr = nullptr;
    c->d_tail_to = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &c->d_tail_ptr, p.n_pad + 1))) return rc;
    if ((rc = dev_alloc(c, &c->d_tail_to, to.size() + 1))) return rc;
    HB_HIP(hipMemcpyAsync(c->d_tail_ptr, ptr.data(), (p.n_pad + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    if (!to.empty()) HB_HIP(hipMemcpyAsync(c->d_tail_to, to.data(), to.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    c->tail_count = to.size();
    c->tail_dirty = false;
    return HB_OK;
}

int step_local(hb_ctx *c)
{
    if (!c->begun || c->finished) return fail(c, HB_ERR_INVALID, "hb_step*: call hb_begin first");
    if (c->pending_local) return fail(c, HB_ERR_INVALID, "hb_step_local called twice");
    if (c->t >= c->max_passes) return fail(c, HB_ERR_LIMIT, "max_passes exceeded");
    const Plan &p = c->plan;
    // mode: dense while most nodes still change (the frontier test would only cost), frontier
    // (bitmap) after, sparse (worklists over the transposed graph) for the convergence tail
    // A_t = edges whose source changed in the previous pass (= out-degree sum of those nodes, counted by
    // the previous pass).  dense: every source is gathered (no test); frontier: every index is read and
    // bit-tested, only active sources are gathered (pays while A_t < ~half of the edges); sparse: only the
    // work rows that read a changed node are visited at all.
    const uint32_t thr = c->opt.tune[2] ? c->opt.tune[2] : 50; // frontier when A_t < thr % of the edges
    bool frontier = !(c->opt.flags & HB_FLAG_NO_FRONTIER) && c->t > 0 &&
                    (c->last_active * 100ull < (uint64_t)thr * c->m_global || thr > 100);
    // sweep mode when A_t * div < edges: measured crossover with the bitmap pass at A_t = 10-12 % of the edges
    // (profiles/r02c_sweep_*: 7.4 % -> 1.25 ms vs 2.15 ms, 16 % -> 4.1 ms vs 2.1 ms on the C3-sized graphs)
    const uint64_t sparse_div = c->opt.tune[6] ? c->opt.tune[6] : 10;
    bool sparse = frontier && c->sparse_ok && (c->last_active * sparse_div < c->m_global || c->opt.tune[6] == 1);
    c->cur_mode = sparse ? 2 : (frontier ? 1 : 0);
    const bool fused = !unfused(c);
    hbk::PassParams pp = make_params(c);
    HB_HIP(hipEventRecord(c->ev[0], c->stream));
    if (ref_tail(c)) {
        // harmonic.rs:244-246: `!exact_changed_nodes.is_empty() && exact_changed_nodes.len() <= threshold`
        if (c->exact_valid && c->last_changed != 0 && c->last_changed <= c->ref_threshold) return tail_pass(c, pp);
        if (c->stale && c->t > 0) {
            // update_all_counters after a tail pass: sources pass `changed_nodes.contains_u128` (harmonic.rs:133) - the
            // bloom filter of the previous pass' changed nodes INCLUDING its false positives (they may hold updates a
            // tail pass did not deliver); never a dense pass (it would deliver all of them)
            frontier = true;
            sparse = false; // (the sweep support is never built in this mode anyway: it runs unfused)
            c->cur_mode = 1;
            hipLaunchKernelGGL(hbk::bloom_frontier_kernel, dim3((unsigned)(p.n_pad / 256 + 1)), dim3(256), 0, c->stream, (const uint32_t *)c->d_bloom,
                               (const uint64_t *)c->d_idlow, (const uint32_t *)c->d_sid_of, p.n_pad, c->bloom_bits, c->d_bits[c->cur]);
            HB_HIP(hipGetLastError());
        }
    }
    if (sparse) {
        // sweep mode: changed nodes -> touch bits of their readers; then the levels, then the node rows
        hbk::SweepParams sp{};
        sp.p = pp;
        sp.out_ptr = c->d_out_ptr;
        sp.out_rows = c->d_out_rows;
        sp.touch = c->d_touch;
        sp.seeds = c->d_seeds;
        sp.heavy = c->d_heavy;
        const uint64_t real_words = p.n_pad / 32;
        // the seed / heavy counters live in two slots used by alternate passes: this pass' first kernel zeroes the other one
        // (no memset launch per pass; hb_begin clears both)
        sp.counts = c->d_sparse_counts + 2 * (c->t & 1);
        sp.counts_next = c->d_sparse_counts + 2 * ((c->t & 1) ^ 1);
        // no bitmap is cleared here: the sweep kernels rewrite every word of this pass' changed bits (node rows in
        // bits_wr, virtual rows in the upper part of bits_rd) and keep the touch bitmap all-zero between passes
        const unsigned sblocks = (unsigned)std::min<uint64_t>(std::max<uint64_t>(real_words / 256, 1), (uint64_t)c->num_cu * 4);
        const unsigned wblocks = (unsigned)c->num_cu * 4;
        if (c->last_changed <= 4096 && !(c->opt.tune[1] & 0x800u)) {
            // convergence tail: one launch instead of collect + expand + heavy (tune[1] bit 11 = the general path, measurement switch)
            hipLaunchKernelGGL(hbk::sweep_seed_small_kernel, dim3(sblocks), dim3(256), 0, c->stream, sp);
        } else {
            hipLaunchKernelGGL(hbk::sweep_collect_kernel, dim3(sblocks), dim3(256), 0, c->stream, sp);
            hipLaunchKernelGGL(hbk::sweep_expand_kernel, dim3(wblocks), dim3(256), 0, c->stream, sp);
            hipLaunchKernelGGL(hbk::sweep_expand_heavy_kernel, dim3(wblocks), dim3(256), 0, c->stream, sp);
        }
        HB_HIP(hipEventRecord(c->ev[5], c->stream)); // sweep passes: ms_level1 = seed collection + expansion
        auto sweep_blocks = [&](uint64_t rows) { // a wave-iteration covers 16 groups of 128 rows
            const uint64_t waves = (rows + 2047) / 2048;
            return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((waves + 3) / 4, (uint64_t)c->num_cu * 4));
        };
        for (size_t l = 0; l + 1 < p.level_begin.size(); l++) {
            sp.p.row_lo = p.level_begin[l];
            sp.p.row_hi = p.level_begin[l + 1];
            if (sp.p.row_hi > sp.p.row_lo)
                hipLaunchKernelGGL(hbk::sweep_rows_kernel<false>, dim3(sweep_blocks(sp.p.row_hi - sp.p.row_lo)), dim3(256), 0, c->stream, sp);
        }
        HB_HIP(hipEventRecord(c->ev[1], c->stream));
        sp.p.row_lo = 0;
        sp.p.row_hi = p.n_pad;
        if (p.n_pad) hipLaunchKernelGGL(hbk::sweep_rows_kernel<true>, dim3(sweep_blocks(p.n_pad)), dim3(256), 0, c->stream, sp);
        HB_HIP(hipEventRecord(c->ev[2], c->stream));
    } else {
        for (size_t l = 0; l + 1 < p.level_begin.size(); l++) {
            pp.row_lo = p.level_begin[l];
            pp.row_hi = p.level_begin[l + 1];
            pp.xcd_map = (l == 0 && p.xcd_groups == 8) ? 1 : 0;
            for (int x = 0; x < 8; x++) {
                pp.xcd_lo[x] = p.xcd_begin[x];
                pp.xcd_hi[x] = p.xcd_begin[x + 1];
            }

            const uint32_t lds_tile = std::min<uint32_t>(c->opt.tune[7], 2048u); // experiment, see hub_lds_tile_kernel
            if (l == 0 && !frontier && lds_tile && !(c->opt.flags & HB_FLAG_PASS_STATS) && !multi_rank(c)) {
                const uint64_t ntiles = (pp.row_hi - pp.row_lo + 63) / 64;
                const size_t lds = (size_t)lds_tile * 64;
                const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(8, (160 * 1024) / (lds + 1024)));
                uint64_t blocks = std::min<uint64_t>(ntiles, (uint64_t)c->num_cu * per_cu);
                if (pp.xcd_map) blocks = std::max<uint64_t>((blocks + 7) / 8 * 8, 8);
                if (lds > 48 * 1024)
                    HB_HIP(hipFuncSetAttribute((const void *)hbk::hub_lds_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (ntiles) hipLaunchKernelGGL(hbk::hub_lds_tile_kernel, dim3((unsigned)blocks), dim3(256), lds, c->stream, pp, lds_tile);
            } else {
                launch_pass(c, pp, false, frontier, false);
            }
            if (l == 0) HB_HIP(hipEventRecord(c->ev[5], c->stream));
        }
        pp.xcd_map = 0;
        HB_HIP(hipEventRecord(c->ev[1], c->stream));
        pp.row_lo = dest_mode(c) ? pp.slice_lo : 0; // destination partition: only the owned rows
        pp.row_hi = dest_mode(c) ? pp.slice_hi : p.n_pad;
        if (edge_changed_only(c)) {
            int rc = edge_co_alloc(c);
            if (rc) return rc;
            pp.lbits = c->d_lbits; // which rows the local merge changed
            c->ubits_valid = false;
        }
        c->ov_ranges = 0;
        if (edge_overlap(c) && p.n_pad >= 64ull * hb_ctx::kOverlap) {
            // the node rows in kOverlap ranges: range k is all-reduced (step_finish, comm_stream) while k + 1 is merged here
            const uint64_t tiles = p.n_pad / 64;
            for (int k = 0; k <= hb_ctx::kOverlap; k++) c->ov_lo[k] = tiles * (uint64_t)k / hb_ctx::kOverlap * 64;
            c->ov_ranges = hb_ctx::kOverlap;
            for (int k = 0; k < hb_ctx::kOverlap; k++) {
                pp.row_lo = c->ov_lo[k];
                pp.row_hi = c->ov_lo[k + 1];
                launch_pass(c, pp, true, frontier, fused);
                HB_HIP(hipEventRecord(c->ov_merged[k], c->stream));
            }
        } else {
            launch_pass(c, pp, true, frontier, fused);
        }
        HB_HIP(hipEventRecord(c->ev[2], c->stream));
    }
    HB_HIP(hipGetLastError());
    c->pending_local = true;
    return HB_OK;
}

int step_finish(hb_ctx *c, int *has_changes)
{
    if (!c->pending_local) return fail(c, HB_ERR_INVALID, "hb_step_finish without hb_step_local");
    const Plan &p = c->plan;
    float ms_coll = 0.f;
    if (unfused(c)) {
        hbk::PassParams pp = make_params(c);
        pp.row_lo = 0;
        pp.row_hi = p.n_pad;
        if (c->comm && edge_changed_only(c)) {
            // union of the ranks' locally-changed rows (all-gather of the bitmaps + OR), then an all-reduce(max) over those
            // rows only, packed in the same order everywhere: -20 % of the bytes in the dense passes of the R-MAT configs,
            // ~ -100 % in the tail
            const uint64_t words = p.n_pad / 32;
            const int world = std::max(c->opt.world_size, 1);
            HB_NCCL(ncclAllGather(c->d_lbits, c->d_lbits_all, words, ncclUint32, c->comm, c->stream));
            HB_HIP(hipMemcpyAsync(c->d_ubits, c->d_lbits_all, words * 4, hipMemcpyDeviceToDevice, c->stream));
            for (int k = 1; k < world && words; k++) {
                hipLaunchKernelGGL(hbk::or_words_kernel, dim3((unsigned)std::min<uint64_t>((words + 255) / 256, 2048)), dim3(256), 0, c->stream, c->d_ubits,
                                   (const uint32_t *)(c->d_lbits_all + (uint64_t)k * words), words);
            }
            HB_HIP(hipGetLastError());
            int rc = edge_co_pack(c);
            if (rc) return rc;
            if (c->co_rows) HB_NCCL(ncclAllReduce(c->d_pack, c->d_pack, c->co_rows * 64, ncclUint8, ncclMax, c->comm, c->stream));
            if ((rc = edge_co_unpack(c))) return rc;
            c->wire_bytes += (uint64_t)(world - 1) * words * 4 + (world > 1 ? 2 * (uint64_t)(world - 1) * c->co_rows * 64 / (uint64_t)world : 0);
        } else if (c->comm && c->ov_ranges) {
            // pipelined: all-reduce of range k on comm_stream as soon as it is merged; its epilogue on the main stream as
            // soon as it is reduced (the epilogue launches below wait for ov_reduced[k])
            for (int k = 0; k < c->ov_ranges; k++) {
                const uint64_t lo = c->ov_lo[k], hi = c->ov_lo[k + 1];
                HB_HIP(hipStreamWaitEvent(c->comm_stream, c->ov_merged[k], 0));
                if (hi > lo) HB_NCCL(ncclAllReduce(pp.wr + lo * 4, pp.wr + lo * 4, (hi - lo) * 64, ncclUint8, ncclMax, c->comm, c->comm_stream));
                HB_HIP(hipEventRecord(c->ov_reduced[k], c->comm_stream));
            }
            c->wire_bytes += c->opt.world_size > 1 ? 2 * (uint64_t)(c->opt.world_size - 1) * p.n_pad * 64 / (uint64_t)c->opt.world_size : 0;
        } else if (c->comm) {
            HB_NCCL(ncclAllReduce(pp.wr, pp.wr, p.n_pad * 64, ncclUint8, ncclMax, c->comm, c->stream));
            c->wire_bytes += c->opt.world_size > 1 ? 2 * (uint64_t)(c->opt.world_size - 1) * p.n_pad * 64 / (uint64_t)c->opt.world_size : 0;
        }
        if (edge_changed_only(c) && c->ubits_valid) pp.ubits = c->d_ubits; // the epilogue visits the exchanged rows only
        c->ubits_valid = false;
        const int ranges = (c->comm && c->ov_ranges && !edge_changed_only(c)) ? c->ov_ranges : 1;
        for (int k = 0; k < ranges; k++) {
            if (ranges > 1) {
                pp.row_lo = c->ov_lo[k];
                pp.row_hi = c->ov_lo[k + 1];
                HB_HIP(hipStreamWaitEvent(c->stream, c->ov_reduced[k], 0));
            }
            if (k == ranges - 1) HB_HIP(hipEventRecord(c->ev[3], c->stream)); // (collective time: up to the last range reduced)
            const uint64_t ntiles = (pp.row_hi - pp.row_lo) / 64;
            if (ntiles) {
                uint32_t bpc = (c->opt.tune[0] & 0xFFu) ? (c->opt.tune[0] & 0xFFu) : 8;
                uint64_t blocks = std::min<uint64_t>(ntiles, (uint64_t)c->num_cu * bpc);
                hipLaunchKernelGGL(hbk::epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, pp);
            }
        }
        c->ov_ranges = 0;
        HB_HIP(hipGetLastError());
    }
    if (dest_mode(c) && c->comm) {
        // every rank produced the final counters, changed bits and changed count of ITS rows (fused
        // kernel): all-gather the slices in place; sum the counts
        hbk::PassParams pp = make_params(c);
        const uint64_t S = c->slice_rows, r = (uint64_t)c->opt.rank;
        if (changed_only(c)) {
            // bits and counts first, then only the counters that changed: one broadcast per rank of its packed run
            HB_NCCL(ncclGroupStart());
            HB_NCCL(ncclAllGather(pp.bits_wr + r * (S / 32), pp.bits_wr, S / 32, ncclUint32, c->comm, c->stream));
            HB_NCCL(ncclAllReduce(pp.counters, pp.counters, hbk::kCounterWords, ncclUint64, ncclSum, c->comm, c->stream));
            HB_NCCL(ncclGroupEnd());
            int rc = exchange_pack(c);
            if (rc) return rc;
            const int world = std::max(c->opt.world_size, 1);
            HB_NCCL(ncclGroupStart());
            for (int k = 0; k < world; k++) {
                const uint64_t cnt = c->ex_off[k + 1] - c->ex_off[k];
                if (cnt) HB_NCCL(ncclBroadcast(c->d_pack + c->ex_off[k] * 4, c->d_pack + c->ex_off[k] * 4, cnt * 64, ncclUint8, k, c->comm, c->stream));
            }
            HB_NCCL(ncclGroupEnd());
            if ((rc = exchange_unpack(c))) return rc;
        } else {
            HB_NCCL(ncclGroupStart());
            HB_NCCL(ncclAllGather(pp.wr + r * S * 4, pp.wr, S * 64, ncclUint8, c->comm, c->stream));
            HB_NCCL(ncclAllGather(pp.bits_wr + r * (S / 32), pp.bits_wr, S / 32, ncclUint32, c->comm, c->stream));
            HB_NCCL(ncclAllReduce(pp.counters, pp.counters, hbk::kCounterWords, ncclUint64, ncclSum, c->comm, c->stream));
            HB_NCCL(ncclGroupEnd());
            c->wire_bytes += (p.n_pad - std::min<uint64_t>(S, p.n_pad)) * 64 + (p.n_pad - std::min<uint64_t>(S, p.n_pad)) / 8;
        }
        HB_HIP(hipEventRecord(c->ev[3], c->stream));
    }
    hipEvent_t ev_end = c->ev[2];
    if (dest_mode(c) && c->comm) ev_end = c->ev[3];
    if (unfused(c)) {
        // events: [0] start, [1] after virtual levels, [2] after local merge, [3] after collective,
        // [4] after the epilogue
        HB_HIP(hipEventRecord(c->ev[4], c->stream));
        ev_end = c->ev[4];
    }
    HB_HIP(hipMemcpyAsync(c->h_counters, c->d_counters + (size_t)hbk::kCounterWords * c->t,
                          hbk::kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    for (int s = 1; s < hbk::kStripes; s++)
        for (int k = 0; k < 4; k++) c->h_counters[k] += c->h_counters[4 * s + k];
    if (unfused(c) || (dest_mode(c) && c->comm)) HB_HIP(hipEventElapsedTime(&ms_coll, c->ev[2], c->ev[3]));
    hb_pass_stats ps{};
    ps.pass = c->t;
    ps.changed = c->h_counters[0];
    ps.active_edges = (c->opt.flags & HB_FLAG_PASS_STATS) ? c->h_counters[1] : c->last_active; // A_t
    ps.touched = c->h_counters[2];
    ps.mode = c->cur_mode;
    float ms_all = 0.f, ms_main = 0.f;
    HB_HIP(hipEventElapsedTime(&ms_all, c->ev[0], ev_end));
    HB_HIP(hipEventElapsedTime(&ms_main, c->ev[1], c->ev[2]));
    ps.ms_gpu = ms_all;
    ps.ms_main = ms_main;
    ps.ms_collective = c->comm ? ms_coll : 0.f;
    if ((c->cur_mode < 2 && p.level_begin.size() > 1) || c->cur_mode == 2) {
        float ms_l1 = 0.f;
        HB_HIP(hipEventElapsedTime(&ms_l1, c->ev[0], c->ev[5]));
        ps.ms_level1 = ms_l1;
    }
    c->pstats.push_back(ps);
    if (ref_tail(c)) {
        int rc = reference_changed_state(c, (const uint32_t *)c->d_bits[c->cur ^ 1], ps.changed);
        if (rc) return rc;
    }
    // counters.step(); changed_nodes = new_changed_nodes; t += 1 (harmonic.rs:273-275)
    c->last_changed = ps.changed;
    c->last_active = c->h_counters[3];
    c->has_changes = ps.changed != 0;
    c->cur ^= 1;
    c->t += 1;
    c->pending_local = false;
    if (has_changes) *has_changes = c->has_changes ? 1 : 0;
    return HB_OK;
}

// The C ABI never unwinds (include/hyperball.h): every entry point that can allocate runs under this guard.
template <class F>
int guarded(hb_ctx *c, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        try { return fail(c, HB_ERR_NOMEM, "out of host memory"); } catch (...) { return HB_ERR_NOMEM; }
    } catch (const std::exception &e) {
        try { return fail(c, HB_ERR_INVALID, std::string("C++ exception: ") + e.what()); } catch (...) { return HB_ERR_INVALID; }
    } catch (...) {
        try { return fail(c, HB_ERR_INVALID, "unknown C++ exception"); } catch (...) { return HB_ERR_INVALID; }
    }
}

// records appended so far live on the device as (from key, to key) pairs + flag bytes: turn them back into hb_edge
// records on the host (rel_flags collapses to "skipped or not", all the reduction needs) - only when the device
// ran out of memory in the middle of a stream
int spill_appended_to_host(hb_ctx *c)
{
    const uint64_t k = c->app.count;
    if (k) {
        std::vector<hb_u128> keys;
        std::vector<uint8_t> bad;
        c->pending.resize(k);
        uint64_t base = 0;
        for (const IngestChunk &ch : c->app.chunks) {
            if (!ch.count) continue;
            keys.resize(2 * ch.count);
            bad.resize(ch.count);
            HB_HIP(hipMemcpyAsync(keys.data(), ch.d_end, ch.count * 32, hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipMemcpyAsync(bad.data(), ch.d_bad, ch.count, hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            for (uint64_t i = 0; i < ch.count; i++) {
                c->pending[base + i].from = keys[2 * i];
                c->pending[base + i].to = keys[2 * i + 1];
                c->pending[base + i].rel_flags = bad[i] ? HB_SKIPPED_REL_MASK : 0;
            }
            base += ch.count;
        }
    }
    c->app.free_all();
    return HB_OK;
}

int set_device(hb_ctx *c)
{
    HB_HIP(hipSetDevice(c->device));
    return HB_OK;
}

} // namespace

// =============================================================================================
extern "C" {

int hb_abi_version(void) { return HB_ABI_VERSION; }

const char *hb_last_error(const hb_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hb_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (count) *count = n;
    return HB_OK;
}

int hb_rccl_unique_id(uint8_t out[128])
{
    return guarded(nullptr, [&]() -> int {
        hb_ctx *c = nullptr;
        if (!out) return fail(c, HB_ERR_INVALID, "out == NULL");
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
        ncclUniqueId id;
        HB_NCCL(ncclGetUniqueId(&id));
        std::memcpy(out, &id, 128);
        return HB_OK;
    });
}

int hb_create(const hb_options *opt, hb_ctx **out)
{
    return guarded(nullptr, [&]() -> int {
        hb_ctx *c = nullptr; // errors before the ctx exists go to the thread-local slot
        if (!out) return fail(c, HB_ERR_INVALID, "out == NULL");
        *out = nullptr;
        hb_options o{};
        if (opt) {
            size_t sz = opt->struct_size ? std::min<size_t>(opt->struct_size, sizeof(hb_options)) : sizeof(hb_options);
            std::memcpy(&o, opt, sz);
        } else {
            o.device = -1;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(c, HB_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
        int dev = o.device;
        if (dev < 0) {
            if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        }
        if (dev >= ndev) return fail(c, HB_ERR_INVALID, "device ordinal out of range");
        hipDeviceProp_t prop;
        HB_HIP(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(c, HB_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        if (o.world_size > 1 && (o.rank < 0 || o.rank >= o.world_size)) return fail(c, HB_ERR_INVALID, "rank out of range");
        if ((o.flags & HB_FLAG_REFERENCE_TAIL) && (o.world_size > 1 || (o.flags & (HB_FLAG_RCCL_SELF | HB_FLAG_DEST_PARTITION))))
            return fail(c, HB_ERR_INVALID, "HB_FLAG_REFERENCE_TAIL is a single-rank mode (no partition / RCCL flags)");
        hb_ctx *ctx = new (std::nothrow) hb_ctx();
        if (!ctx) return fail(c, HB_ERR_NOMEM, "out of host memory");
        ctx->opt = o;
        ctx->device = dev;
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ctx->arch = prop.gcnArchName;
        ctx->max_passes = o.max_passes ? o.max_passes : 4096;
        c = ctx;
        auto bail = [&](int code) {
            std::string m = ctx->err;
            hb_destroy(ctx);
            g_create_error = m;
            return code;
        };
        if (hipSetDevice(dev) != hipSuccess) { ctx->err = "hipSetDevice failed"; return bail(HB_ERR_HIP); }
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(HB_ERR_HIP); }
        for (int i = 0; i < 6; i++)
            if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(HB_ERR_HIP); }
        if (hipHostMalloc((void **)&ctx->h_counters, hbk::kCounterWords * sizeof(unsigned long long)) != hipSuccess) { ctx->err = "hipHostMalloc failed"; return bail(HB_ERR_NOMEM); }
        if ((o.world_size > 1 && !(o.flags & HB_FLAG_NO_RCCL)) || (o.flags & HB_FLAG_RCCL_SELF)) {
            if (hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(HB_ERR_HIP); }
            for (int i = 0; i < hb_ctx::kOverlap; i++)
                if (hipEventCreateWithFlags(&ctx->ov_merged[i], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ctx->ov_reduced[i], hipEventDisableTiming) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(HB_ERR_HIP); }
        }
        if (o.world_size > 1 && !(o.flags & HB_FLAG_NO_RCCL)) {
            ncclUniqueId id;
            std::memcpy(&id, o.rccl_id, 128);
            ncclResult_t r = ncclCommInitRank(&ctx->comm, o.world_size, id, o.rank);
            if (r != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); ctx->comm = nullptr; return bail(HB_ERR_RCCL); }
        } else if (o.world_size == 1 && (o.flags & HB_FLAG_RCCL_SELF)) {
            ncclUniqueId id;
            std::memcpy(&id, o.rccl_id, 128);
            ncclResult_t r = ncclCommInitRank(&ctx->comm, 1, id, 0);
            if (r != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); ctx->comm = nullptr; return bail(HB_ERR_RCCL); }
        }
        *out = ctx;
        return HB_OK;
    });
}

void hb_destroy(hb_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    ctx->app.free_all();
    free_graph_buffers(ctx);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    for (int i = 0; i < 6; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < hb_ctx::kOverlap; i++) {
        if (ctx->ov_merged[i]) (void)hipEventDestroy(ctx->ov_merged[i]);
        if (ctx->ov_reduced[i]) (void)hipEventDestroy(ctx->ov_reduced[i]);
    }
    if (ctx->comm_stream) {
        (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamDestroy(ctx->comm_stream);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int hb_device_name(const hb_ctx *ctx, char *name, uint64_t cap)
{
    if (!ctx || !name || !cap) return HB_ERR_INVALID;
    std::snprintf(name, (size_t)cap, "%s", ctx->arch.c_str());
    return HB_OK;
}

int hb_device_synchronize(hb_ctx *c)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        HB_HIP(hipStreamSynchronize(c->stream));
        return HB_OK;
    });
}

// ---- input --------------------------------------------------------------------------------
int hb_load_edges(hb_ctx *c, const hb_u128 *node_ids, uint64_t n, const hb_edge *edges, uint64_t m)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        c->stats = hb_stats{};
        double t0 = now_ms();
        // node/edge-set reduction: on the GPU (hb_ingest.hip) unless the host path is forced; identical output
        // The device pipeline keeps ~100 B per record resident at its peak (endpoint keys, sort double buffers):
        // when that cannot fit, or an allocation fails anyway, the host path produces the same graph.
        bool on_host = (c->opt.flags & HB_FLAG_HOST_INGEST) != 0 || m >= c->lim_records;
        if (!on_host) {
            size_t free_b = 0, total_b = 0;
            const double need = 50.0 * (double)m + 48.0 * (double)((node_ids && n) ? n : 0) + 512e6;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (double)free_b) on_host = true;
        }
        DeviceCsr csr;
        const bool keep_on_device = !on_host && device_plan(c);
        uint64_t peak = 0;
        std::string e = on_host ? ingest_edges(node_ids, n, edges, m, &c->g)
                                : gpu_ingest_edges((void *)c->stream, node_ids, n, edges, m, &c->g, keep_on_device ? &csr : nullptr, &peak);
        if (!on_host && !e.empty() && (e.find("out of memory") != std::string::npos || e.find("OutOfMemory") != std::string::npos)) {
            (void)hipGetLastError(); // clear the sticky allocation error
            peak = 0;
            e = ingest_edges(node_ids, n, edges, m, &c->g);
        }
        if (!e.empty())
            return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : (e.find("hip") != std::string::npos ? HB_ERR_HIP : HB_ERR_LIMIT), e);
        if ((rc = keep_owned(c, &csr))) {
            if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
            if (csr.d_src) (void)hipFree(csr.d_src);
            return rc;
        }
        const uint64_t nn = c->g.ids.size();
        const uint64_t m_eff = csr.d_row_ptr ? csr.m : (nn && c->g.row_ptr.size() == nn + 1 ? c->g.row_ptr[nn] : 0);
        c->stats.ms_ingest = now_ms() - t0;
        double ing = c->stats.ms_ingest;
        rc = plan_and_upload(c, csr.d_row_ptr ? &csr : nullptr, m_eff);
        if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr); // only if plan_and_upload bailed out before taking them
        if (csr.d_src) (void)hipFree(csr.d_src);
        c->stats.ms_ingest = ing;
        c->stats.ingest_peak_bytes = peak;
        return rc;
    });
}

int hb_append_edges(hb_ctx *c, const hb_edge *edges, uint64_t m)
{
    return guarded(c, [&]() -> int {
        if (!c || (m && !edges)) return c ? fail(c, HB_ERR_INVALID, "edges == NULL") : HB_ERR_INVALID;
        if (!m) return HB_OK;
        int rc = set_device(c);
        if (rc) return rc;
        const bool on_host = (c->opt.flags & HB_FLAG_HOST_INGEST) != 0 || !c->pending.empty();
        if (!on_host) {
            // unpack this batch behind what is already on the device (chunked: no reallocation, 33 bytes per record)
            c->app.max_records = c->lim_records;
            c->app.max_bytes = c->lim_bytes;
            c->app.chunk_records = c->lim_chunk;
            const std::string err = gpu_ingest_append((void *)c->stream, &c->app, edges, m);
            if (err.empty()) return HB_OK;
            if (err.find("out of memory") == std::string::npos && err.find("too many records") == std::string::npos) return fail(c, HB_ERR_HIP, err);
            // the device cannot hold the stream (memory, or the 2^32-record limit of the device reduction): bring back
            // what is there and continue on the host
            rc = spill_appended_to_host(c);
            if (rc) return rc;
        }
        c->pending.insert(c->pending.end(), edges, edges + m);
        return HB_OK;
    });
}

int hb_finalize(hb_ctx *c, const hb_u128 *node_ids, uint64_t n)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        if (c->app.chunks.empty()) { // host mode (or nothing appended)
            const uint64_t keep_lim = c->lim_records;
            if (!c->pending.empty()) c->lim_records = 0; // the stream was spilled: it stays on the host path
            rc = hb_load_edges(c, node_ids, n, c->pending.data(), c->pending.size());
            c->lim_records = keep_lim;
            std::vector<hb_edge>().swap(c->pending);
            return rc;
        }
        c->stats = hb_stats{};
        const double t0 = now_ms();
        DeviceCsr csr;
        const bool keep_on_device = device_plan(c);
        uint64_t peak = 0;
        const std::string e = gpu_ingest_reduce((void *)c->stream, node_ids, n, &c->app, &c->g, keep_on_device ? &csr : nullptr, &peak);
        if (!e.empty())
            return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : (e.find("hip") != std::string::npos ? HB_ERR_HIP : HB_ERR_LIMIT), e);
        if ((rc = keep_owned(c, &csr))) {
            if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
            if (csr.d_src) (void)hipFree(csr.d_src);
            return rc;
        }
        const uint64_t nn = c->g.ids.size();
        const uint64_t m_eff = csr.d_row_ptr ? csr.m : (nn && c->g.row_ptr.size() == nn + 1 ? c->g.row_ptr[nn] : 0);
        const double ing = now_ms() - t0;
        rc = plan_and_upload(c, csr.d_row_ptr ? &csr : nullptr, m_eff);
        if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
        if (csr.d_src) (void)hipFree(csr.d_src);
        c->stats.ms_ingest = ing;
        c->stats.ingest_peak_bytes = peak;
        return rc;
    });
}

int hb_debug_set_ingest_limits(hb_ctx *c, uint64_t max_records, uint64_t max_device_bytes, uint64_t chunk_records)
{
    if (!c) return HB_ERR_INVALID;
    c->lim_records = max_records ? std::min<uint64_t>(max_records, 0xFFFFFF00ull) : 0xFFFFFF00ull;
    c->lim_bytes = max_device_bytes;
    c->lim_chunk = chunk_records;
    return HB_OK;
}

int hb_append_tail_edges(hb_ctx *c, const hb_edge *records, uint64_t count)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!ref_tail(c)) return fail(c, HB_ERR_INVALID, "tail records: create the context with HB_FLAG_REFERENCE_TAIL");
        if (!c->loaded) return fail(c, HB_ERR_INVALID, "tail records: load the graph first");
        if (count && !records) return fail(c, HB_ERR_INVALID, "records == NULL with count > 0");
        int rc = set_device(c);
        if (rc) return rc;
        if ((rc = need_host_dev_of(c))) return rc;
        if (!c->tail_index && count) c->tail_index = tail_index_build(c->g.ids.data(), c->g.ids.size());
        if (!c->tail_index && count) return fail(c, HB_ERR_NOMEM, "out of host memory indexing the node ids");
        const std::string e = tail_collect(c->tail_index, c->g.ids.size(), records, count, &c->tail_open);
        if (!e.empty()) return fail(c, HB_ERR_NOMEM, e);
        c->tail_dirty = true;
        return HB_OK;
    });
}

int hb_tail_segment_end(hb_ctx *c)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!ref_tail(c)) return fail(c, HB_ERR_INVALID, "tail records: create the context with HB_FLAG_REFERENCE_TAIL");
        if (c->tail_open.empty()) return HB_OK;
        int rc = need_host_dev_of(c);
        if (rc) return rc;
        const std::string e = tail_close_segment(c->tail_index, c->g.ids.data(), c->plan.dev_of.data(), &c->tail_open, &c->tail_keys);
        if (!e.empty()) return fail(c, HB_ERR_NOMEM, e);
        c->tail_dirty = true;
        return HB_OK;
    });
}

int hb_load_tail_edges(hb_ctx *c, const hb_edge *records, uint64_t count)
{
    if (c && ref_tail(c) && c->loaded) {
        c->tail_keys.clear();
        c->tail_open.clear();
        c->tail_dirty = true;
    }
    return hb_append_tail_edges(c, records, count);
}

int hb_load_dense(hb_ctx *c, const hb_u128 *sorted_ids, uint64_t n, const uint64_t *row_ptr, const uint32_t *src,
                  uint64_t m_eff)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        c->stats = hb_stats{};
        double t0 = now_ms();
        std::string e = check_dense(sorted_ids, n, row_ptr, src, m_eff);
        if (!e.empty()) return fail(c, e.find("too many") != std::string::npos ? HB_ERR_LIMIT : HB_ERR_INVALID, e);
        const bool on_device = device_plan(c);
        const bool keep_host = !on_device || m_eff <= kKeepHostGraph; // host copies: host planner, hb_debug_copy_graph
        try {
            c->g.ids.assign(sorted_ids, sorted_ids + n);
            if (keep_host) {
                c->g.row_ptr.assign(row_ptr, row_ptr + n + 1);
                if (n == 0) c->g.row_ptr.assign(1, 0);
                c->g.src.assign(src, src + m_eff);
            } else {
                std::vector<uint64_t>().swap(c->g.row_ptr);
                std::vector<uint32_t>().swap(c->g.src);
            }
        } catch (const std::bad_alloc &) {
            return fail(c, HB_ERR_NOMEM, "out of host memory copying the graph");
        }
        c->g.m_input = m_eff;
        c->g.m_unique = m_eff;
        DeviceCsr csr;
        if (on_device && n) { // straight from the caller's arrays to the device: no host copy of a multi-GB CSR
            HB_HIP(hipMalloc((void **)&csr.d_row_ptr, (n + 1) * sizeof(uint64_t)));
            hipError_t he = hipMalloc((void **)&csr.d_src, std::max<uint64_t>(m_eff, 1) * sizeof(uint32_t));
            if (he == hipSuccess) he = hipMemcpyAsync(csr.d_row_ptr, row_ptr, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess && m_eff) he = hipMemcpyAsync(csr.d_src, src, m_eff * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) {
                (void)hipFree(csr.d_row_ptr);
                if (csr.d_src) (void)hipFree(csr.d_src);
                return fail(c, he == hipErrorOutOfMemory ? HB_ERR_NOMEM : HB_ERR_HIP, std::string("uploading the graph: ") + hipGetErrorString(he));
            }
            csr.m = m_eff;
        }
        if ((rc = keep_owned(c, &csr))) {
            if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
            if (csr.d_src) (void)hipFree(csr.d_src);
            return rc;
        }
        const uint64_t m_local = csr.d_row_ptr ? csr.m : ((dest_mode(c) && n) ? c->g.row_ptr[n] : m_eff);
        double ing = now_ms() - t0;
        rc = plan_and_upload(c, csr.d_row_ptr ? &csr : nullptr, m_local);
        if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
        if (csr.d_src) (void)hipFree(csr.d_src);
        c->stats.ms_ingest = ing;
        return rc;
    });
}

// ---- compute ------------------------------------------------------------------------------
int hb_begin(hb_ctx *c)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!c->loaded) return fail(c, HB_ERR_INVALID, "hb_begin: no graph loaded");
        int rc = set_device(c);
        if (rc) return rc;
        const Plan &p = c->plan;
        if (p.n_pad * 4 >= 0xFFFFFFFFull) return fail(c, HB_ERR_LIMIT, "more than 2^30 nodes: init_kernel is one quad per node in one dispatch");
        {
            hipError_t stale = hipGetLastError(); // an unchecked failure of an earlier call on this thread
            if (stale != hipSuccess) return fail(c, HB_ERR_HIP, std::string("stale HIP error before hb_begin: ") + hipGetErrorString(stale));
        }
        // d_part needs no clearing: pass 0 is always dense, and a dense pass overwrites every partial without
        // reading it (hb_kernels.hip.h)
        HB_HIP(hipMemsetAsync(c->d_bits[0], 0, c->bits_words * 4, c->stream));
        HB_HIP(hipMemsetAsync(c->d_bits[1], 0, c->bits_words * 4, c->stream));
        HB_HIP(hipMemsetAsync(c->d_counters, 0, ((size_t)c->max_passes + 1) * hbk::kCounterWords * sizeof(unsigned long long), c->stream));
        if (c->d_sparse_counts) HB_HIP(hipMemsetAsync(c->d_sparse_counts, 0, 64 * sizeof(unsigned int), c->stream));
        if (c->ksum_len > p.n_pad) // slice padding beyond the rows init_kernel writes (all-reduce mode)
            HB_HIP(hipMemsetAsync(c->d_ksum + p.n_pad, 0, (c->ksum_len - p.n_pad) * sizeof(double), c->stream));
        if (p.n_pad) {
            unsigned blocks = (unsigned)((p.n_pad * 4 + 255) / 256);
            hipLaunchKernelGGL(hbk::init_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_idlow, (const uint32_t *)c->d_sid_of, p.n_pad,
                               c->d_regs[0], c->d_regs[1], c->d_ksum, c->d_kerr, c->d_size, c->d_bits[0], c->d_kdirty,
                               c->d_raw, c->d_bias, c->d_lc);
            HB_HIP(hipGetLastError());
        }
        if (ref_tail(c)) {
            // harmonic.rs:221,228: U64BloomFilter::new(num_nodes, 0.05); threshold = sqrt(num_nodes).max(0).round()
            c->bloom_bits = bloom_num_bits(p.n, 0.05);
            c->ref_threshold = (uint64_t)std::round(std::max(std::sqrt((double)p.n), 0.0));
            if (!c->d_bloom) {
                if ((rc = dev_alloc(c, &c->d_bloom, (c->bloom_bits + 31) / 32 + 2))) return rc;
                if ((rc = dev_alloc(c, &c->d_bloom_ones, 2))) return rc;
                if ((rc = dev_alloc(c, &c->d_list, c->ref_threshold + 2))) return rc;
            }
            // the segment still open ends here; no records given: the forward-links query finds nothing (an empty index)
            if (!c->tail_open.empty() && (rc = hb_tail_segment_end(c))) return rc;
            if ((c->tail_dirty || !c->d_tail_ptr) && (rc = upload_tail_index(c))) return rc;
            c->exact_counting = c->exact_valid = c->stale = false;
        }
        HB_HIP(hipStreamSynchronize(c->stream));
        c->t = 0;
        c->cur = 0;
        c->wire_bytes = 0;
        c->has_changes = true; // harmonic.rs:232
        c->last_changed = p.n;
        c->last_active = c->m_global;
        c->pending_local = false;
        c->pstats.clear();
        c->begun = true;
        c->finished = false;
        c->res_count = 0;
        return HB_OK;
    });
}

int hb_step_local(hb_ctx *c)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        return step_local(c);
    });
}

int hb_step_finish(hb_ctx *c, int *has_changes)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        return step_finish(c, has_changes);
    });
}

int hb_step(hb_ctx *c, int *has_changes)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        if ((rc = step_local(c))) return rc;
        return step_finish(c, has_changes);
    });
}

int hb_finish(hb_ctx *c)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!c->begun) return fail(c, HB_ERR_INVALID, "hb_finish: call hb_begin first");
        int rc = set_device(c);
        if (rc) return rc;
        const Plan &p = c->plan;
        double t0 = now_ms();
        if (c->comm && p.n_pad) {
            // every rank ends with all Kahan sums: in-place all-gather of the owned slices
            HB_NCCL(ncclAllGather(c->d_ksum + (uint64_t)c->opt.rank * c->slice_rows, c->d_ksum, c->slice_rows, ncclDouble,
                                  c->comm, c->stream));
        }
        // normalize_centralities (harmonic.rs:178-195) on the device, in ascending-NodeID order;
        // norm_factor = (num_nodes - 1) as f64 (:229)
        const double norm = (double)(p.n ? p.n - 1 : 0);
        unsigned long long *cnt = c->d_counters + (size_t)c->max_passes * hbk::kCounterWords; // the spare slot
        HB_HIP(hipMemsetAsync(cnt, 0, hbk::kCounterWords * sizeof(unsigned long long), c->stream));
        if (p.n) {
            unsigned blocks = (unsigned)std::min<uint64_t>((p.n + 255) / 256, (uint64_t)c->num_cu * 8);
            hipLaunchKernelGGL(hbk::finish_kernel, dim3(blocks), dim3(256), 0, c->stream, (const double *)c->d_ksum,
                               (const uint32_t *)c->d_dev_of, p.n, norm, c->d_out, cnt);
            HB_HIP(hipGetLastError());
            HB_HIP(hipMemcpyAsync(c->h_out, c->d_out, p.n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        HB_HIP(hipMemcpyAsync(c->h_counters, cnt, hbk::kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        c->res_count = 0;
        for (int s = 0; s < hbk::kStripes; s++) c->res_count += c->h_counters[4 * s];
        c->stats.ms_d2h = now_ms() - t0;
        c->stats.results = c->res_count;
        c->stats.passes = c->t;
        double g = 0, coll = 0;
        for (auto &ps : c->pstats) { g += ps.ms_gpu; coll += ps.ms_collective; }
        c->stats.ms_loop_gpu = g;
        c->stats.ms_collective = coll;
        c->stats.wire_bytes = c->wire_bytes;
        c->finished = true;
        return HB_OK;
    });
}

int hb_run(hb_ctx *c, hb_stats *stats)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = hb_begin(c);
        if (rc) return rc;
        double t0 = now_ms();
        int has = 1;
        // harmonic.rs:237-240: loop { if !has_changes { break } ... }
        while (has) {
            if ((rc = hb_step(c, &has))) return rc;
        }
        c->stats.ms_loop = now_ms() - t0;
        if ((rc = hb_finish(c))) return rc;
        if (stats) *stats = c->stats;
        return HB_OK;
    });
}

int hb_get_stats(const hb_ctx *c, hb_stats *out)
{
    if (!c || !out) return HB_ERR_INVALID;
    *out = c->stats;
    out->passes = c->t;
    return HB_OK;
}

int hb_get_pass_stats(const hb_ctx *c, uint64_t t, hb_pass_stats *out)
{
    if (!c || !out || t >= c->pstats.size()) return HB_ERR_INVALID;
    *out = c->pstats[t];
    return HB_OK;
}

// ---- results ------------------------------------------------------------------------------
int hb_result_count(hb_ctx *c, uint64_t *count)
{
    return guarded(c, [&]() -> int {
        if (!c || !count) return HB_ERR_INVALID;
        if (!c->finished) return fail(c, HB_ERR_INVALID, "no results: hb_run / hb_finish not called");
        *count = c->res_count;
        return HB_OK;
    });
}

int hb_result_copy(hb_ctx *c, hb_u128 *ids, double *vals, uint64_t cap)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!c->finished) return fail(c, HB_ERR_INVALID, "no results: hb_run / hb_finish not called");
        // compaction of the per-node array (absent = negative) into the caller's buffers
        const uint64_t n = c->plan.n;
        uint64_t k = 0;
        for (uint64_t sid = 0; sid < n && k < cap; sid++) {
            const double v = c->h_out[sid];
            if (v < 0.0) continue;
            if (ids) ids[k] = c->g.ids[sid];
            if (vals) vals[k] = v;
            k++;
        }
        return HB_OK;
    });
}

int hb_result_ranks(hb_ctx *c, uint64_t *ranks, uint64_t cap)
{
    return guarded(c, [&]() -> int {
        if (!c || (cap && !ranks)) return HB_ERR_INVALID;
        if (!c->finished) return fail(c, HB_ERR_INVALID, "no results: hb_run / hb_finish not called");
        if (cap < c->res_count) return fail(c, HB_ERR_INVALID, "hb_result_ranks: cap < hb_result_count");
        int rc = set_device(c);
        if (rc) return rc;
        std::string e = gpu_rank_results((void *)c->stream, c->d_out, c->plan.n, c->res_count, ranks);
        if (!e.empty()) return fail(c, HB_ERR_HIP, e);
        return HB_OK;
    });
}

int hb_result_top(hb_ctx *c, uint64_t k, hb_u128 *ids, double *vals, uint64_t *written)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!c->finished) return fail(c, HB_ERR_INVALID, "no results: hb_run / hb_finish not called");
        int rc = set_device(c);
        if (rc) return rc;
        const uint64_t top = std::min<uint64_t>(k, c->res_count);
        if (written) *written = top;
        if (!top || (!ids && !vals)) return HB_OK;
        std::vector<uint64_t> order(top);
        std::string e = gpu_rank_results((void *)c->stream, c->d_out, c->plan.n, c->res_count, nullptr, order.data(), top);
        if (!e.empty()) return fail(c, HB_ERR_HIP, e);
        // result index (ascending NodeID among the kept ones) -> sid
        std::vector<uint32_t> kept;
        kept.reserve(c->res_count);
        for (uint64_t sid = 0; sid < c->plan.n; sid++)
            if (c->h_out[sid] >= 0.0) kept.push_back((uint32_t)sid);
        if (kept.size() != c->res_count) return fail(c, HB_ERR_INVALID, "hb_result_top: result buffer changed since hb_finish");
        for (uint64_t i = 0; i < top; i++) {
            const uint32_t sid = kept[order[i]];
            if (ids) ids[i] = c->g.ids[sid];
            if (vals) vals[i] = c->h_out[sid];
        }
        return HB_OK;
    });
}

// ---- debug exports ------------------------------------------------------------------------
int hb_debug_copy_registers(hb_ctx *c, uint8_t *out)
{
    return guarded(c, [&]() -> int {
        if (!c || !out) return HB_ERR_INVALID;
        if (!c->begun) return fail(c, HB_ERR_INVALID, "call hb_begin first");
        int rc = set_device(c);
        if (rc) return rc;
        const uint64_t n = c->plan.n;
        if (!n) return HB_OK;
        uint4 *tmp = nullptr;
        HB_HIP(hipMalloc((void **)&tmp, n * 64));
        unsigned blocks = (unsigned)((n * 4 + 255) / 256);
        hipLaunchKernelGGL(hbk::gather_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint4 *)c->d_regs[c->cur],
                           (const uint32_t *)c->d_dev_of, n, tmp);
        hipError_t e = hipMemcpyAsync(out, tmp, n * 64, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(c, HB_ERR_HIP, hipGetErrorString(e));
        return HB_OK;
    });
}

int hb_debug_copy_kahan(hb_ctx *c, double *sum, double *err)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!c->begun) return fail(c, HB_ERR_INVALID, "call hb_begin first");
        int rc = set_device(c);
        if (rc) return rc;
        if ((rc = need_host_dev_of(c))) return rc;
        const Plan &p = c->plan;
        std::vector<double> tmp(p.n_pad ? p.n_pad : 1);
        for (int k = 0; k < 2; k++) {
            double *dst = k ? err : sum;
            if (!dst || !p.n_pad) continue;
            HB_HIP(hipMemcpyAsync(tmp.data(), k ? c->d_kerr : c->d_ksum, p.n_pad * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            for (uint64_t sid = 0; sid < p.n; sid++) dst[sid] = tmp[p.dev_of[sid]];
        }
        return HB_OK;
    });
}

int hb_debug_copy_sizes(hb_ctx *c, uint64_t *out)
{
    return guarded(c, [&]() -> int {
        if (!c || !out) return HB_ERR_INVALID;
        if (!c->begun) return fail(c, HB_ERR_INVALID, "call hb_begin first");
        int rc = set_device(c);
        if (rc) return rc;
        if ((rc = need_host_dev_of(c))) return rc;
        const Plan &p = c->plan;
        if (!p.n_pad) return HB_OK;
        std::vector<uint64_t> tmp(p.n_pad);
        HB_HIP(hipMemcpyAsync(tmp.data(), c->d_size, p.n_pad * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        for (uint64_t sid = 0; sid < p.n; sid++) out[sid] = tmp[p.dev_of[sid]];
        return HB_OK;
    });
}

int hb_debug_hll_size(hb_ctx *c, const uint8_t *regs, uint64_t count, uint64_t *out)
{
    return guarded(c, [&]() -> int {
        if (!c || (count && (!regs || !out))) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        if (!count) return HB_OK;
        // tables may not be on the device yet (no graph loaded): stage private copies
        double *d_raw = nullptr, *d_bias = nullptr;
        uint8_t *d_lc = nullptr, *d_regs = nullptr;
        uint64_t *d_out = nullptr;
        uint8_t lc[68];
        if (!build_lc_table(lc)) return fail(c, HB_ERR_INVALID, "linear-counting table not robust on this libm");
        const uint64_t rows_pad = (count + 15) & ~15ull;
        hipError_t e = hipMalloc((void **)&d_raw, sizeof(HLL64_RAW_ESTIMATE));
        if (e == hipSuccess) e = hipMalloc((void **)&d_bias, sizeof(HLL64_BIAS));
        if (e == hipSuccess) e = hipMalloc((void **)&d_lc, 256);
        if (e == hipSuccess) e = hipMalloc((void **)&d_regs, rows_pad * 64);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, rows_pad * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d_raw, HLL64_RAW_ESTIMATE, sizeof(HLL64_RAW_ESTIMATE), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_bias, HLL64_BIAS, sizeof(HLL64_BIAS), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_lc, lc, 68, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_regs, regs, count * 64, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            unsigned blocks = (unsigned)((rows_pad * 4 + 255) / 256);
            hipLaunchKernelGGL(hbk::hll_size_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint4 *)d_regs, count, d_out,
                               (const double *)d_raw, (const double *)d_bias, (const uint8_t *)d_lc);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, count * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_raw); (void)hipFree(d_bias); (void)hipFree(d_lc); (void)hipFree(d_regs); (void)hipFree(d_out);
        if (e != hipSuccess) return fail(c, HB_ERR_HIP, hipGetErrorString(e));
        return HB_OK;
    });
}

int hb_debug_state_hash(hb_ctx *c, uint64_t out[2])
{
    return guarded(c, [&]() -> int {
        if (!c || !out) return HB_ERR_INVALID;
        if (!c->begun) return fail(c, HB_ERR_INVALID, "call hb_begin first");
        int rc = set_device(c);
        if (rc) return rc;
        out[0] = out[1] = 0;
        const uint64_t n = c->plan.n;
        if (!n) return HB_OK;
        unsigned long long *cnt = c->d_counters + (size_t)c->max_passes * hbk::kCounterWords; // the spare slot
        HB_HIP(hipMemsetAsync(cnt, 0, hbk::kCounterWords * sizeof(unsigned long long), c->stream));
        const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)c->num_cu * 8);
        hipLaunchKernelGGL(hbk::state_hash_kernel, dim3(blocks), dim3(256), 0, c->stream, (const uint4 *)c->d_regs[c->cur],
                           (const double *)c->d_ksum, (const double *)c->d_kerr, (const uint32_t *)c->d_dev_of, n, cnt);
        HB_HIP(hipGetLastError());
        std::vector<unsigned long long> h(hbk::kCounterWords);
        HB_HIP(hipMemcpyAsync(h.data(), cnt, hbk::kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        for (int s = 0; s < hbk::kStripes; s++) {
            out[0] += h[4 * s];
            out[1] += h[4 * s + 1];
        }
        if (multi_rank(c)) out[1] = 0; // a rank holds the Kahan state of its own rows only
        return HB_OK;
    });
}

int hb_debug_copy_graph(hb_ctx *c, hb_u128 *ids, uint64_t *row_ptr, uint32_t *src)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (!c->loaded) return fail(c, HB_ERR_INVALID, "no graph loaded");
        const uint64_t n = c->g.ids.size();
        if ((row_ptr || src) && c->g.row_ptr.size() != n + 1)
            return fail(c, HB_ERR_LIMIT, "the reduced graph is kept on the host only up to 2^26 edges (or with HB_FLAG_HOST_PLAN)");
        if (ids && n) std::memcpy(ids, c->g.ids.data(), n * sizeof(hb_u128));
        if (row_ptr) std::memcpy(row_ptr, c->g.row_ptr.data(), (n + 1) * sizeof(uint64_t));
        if (src && !c->g.src.empty()) std::memcpy(src, c->g.src.data(), c->g.src.size() * sizeof(uint32_t));
        return HB_OK;
    });
}

int hb_debug_copy_plan(hb_ctx *c, uint64_t sizes[4], uint32_t *order, uint64_t *plan_row_ptr, uint32_t *plan_src, uint64_t *level_begin)
{
    return guarded(c, [&]() -> int {
        if (!c || !sizes) return HB_ERR_INVALID;
        if (!c->loaded) return fail(c, HB_ERR_INVALID, "no graph loaded");
        int rc = set_device(c);
        if (rc) return rc;
        const Plan &p = c->plan;
        const uint64_t rows_total = p.n_pad + p.nv;
        sizes[0] = p.n_pad;
        sizes[1] = p.nv;
        sizes[2] = c->plan_entries;
        sizes[3] = p.level_begin.size() ? p.level_begin.size() - 1 : 0;
        if (order && p.n_pad) HB_HIP(hipMemcpyAsync(order, c->d_sid_of, p.n_pad * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        if (plan_row_ptr) HB_HIP(hipMemcpyAsync(plan_row_ptr, c->d_row_ptr, (rows_total + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        if (plan_src && c->plan_entries)
            HB_HIP(hipMemcpyAsync(plan_src, c->d_src, c->plan_entries * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        if (level_begin) std::memcpy(level_begin, p.level_begin.data(), p.level_begin.size() * sizeof(uint64_t));
        return HB_OK;
    });
}

int hb_host_ingest(const hb_u128 *node_ids, uint64_t n, const hb_edge *edges, uint64_t m, uint64_t *n_out,
                   uint64_t *m_unique, uint64_t *m_eff, hb_u128 *ids, uint64_t *row_ptr, uint32_t *src)
{
    return guarded(nullptr, [&]() -> int {
        hb_ctx *c = nullptr;
        DenseGraph g;
        std::string e = ingest_edges(node_ids, n, edges, m, &g);
        if (!e.empty()) return fail(c, HB_ERR_INVALID, e);
        const uint64_t nn = g.ids.size();
        if (n_out) *n_out = nn;
        if (m_unique) *m_unique = g.m_unique;
        if (m_eff) *m_eff = g.src.size();
        if (ids && nn) std::memcpy(ids, g.ids.data(), nn * sizeof(hb_u128));
        if (row_ptr) std::memcpy(row_ptr, g.row_ptr.data(), (nn + 1) * sizeof(uint64_t));
        if (src && !g.src.empty()) std::memcpy(src, g.src.data(), g.src.size() * sizeof(uint32_t));
        return HB_OK;
    });
}

int hb_host_plan(uint64_t n, const uint64_t *row_ptr, const uint32_t *src, uint32_t flags, uint32_t chunk,
                 const uint32_t *tune, uint64_t sizes[4], uint32_t *order, uint64_t *plan_row_ptr, uint32_t *plan_src,
                 uint64_t *level_begin)
{
    return guarded(nullptr, [&]() -> int {
        hb_ctx *c = nullptr;
        if (!sizes || (n && !row_ptr)) return fail(c, HB_ERR_INVALID, "NULL argument");
        std::vector<uint32_t> outdeg;
        const bool reorder = !(flags & HB_FLAG_NO_REORDER);
        static const uint64_t zero = 0;
        if (n == 0) row_ptr = &zero;
        if (reorder) count_out_degree(row_ptr, src, n, &outdeg);
        Plan p;
        PlanTune pt = plan_tune(chunk, tune);
        if (tune && tune[7] > 1) pt.world = tune[7]; // destination-partition layout (test hook)
        pt.xcd_map = !(flags & HB_FLAG_NO_XCD_MAP);
        std::string e = build_plan(n, row_ptr, src, outdeg, reorder, pt, &p);
        if (!e.empty()) return fail(c, HB_ERR_LIMIT, e);
        sizes[0] = p.n_pad;
        sizes[1] = p.nv;
        sizes[2] = p.src.size();
        sizes[3] = p.level_begin.size() ? p.level_begin.size() - 1 : 0;
        if (order && p.n_pad) std::memcpy(order, p.order.data(), p.n_pad * sizeof(uint32_t));
        if (plan_row_ptr) std::memcpy(plan_row_ptr, p.row_ptr.data(), p.row_ptr.size() * sizeof(uint64_t));
        if (plan_src && !p.src.empty()) std::memcpy(plan_src, p.src.data(), p.src.size() * sizeof(uint32_t));
        if (level_begin) std::memcpy(level_begin, p.level_begin.data(), p.level_begin.size() * sizeof(uint64_t));
        return HB_OK;
    });
}

int hb_debug_tail_index(uint64_t n, const hb_u128 *sorted_ids, const uint32_t *dev_of, uint64_t n_pad, const hb_edge *records,
                        uint64_t count, uint64_t *ptr_out, uint32_t *to_out, uint64_t to_cap, uint64_t *to_len)
{
    return guarded(nullptr, [&]() -> int {
        hb_ctx *c = nullptr;
        if ((n && (!sorted_ids || !dev_of)) || (count && !records) || !ptr_out || !to_len) return fail(c, HB_ERR_INVALID, "NULL argument");
        TailIndex *tix = n ? tail_index_build(sorted_ids, n) : nullptr;
        std::vector<uint64_t> keys, ptr;
        std::vector<uint32_t> to;
        std::vector<TailDoc> open; // the records are one segment in doc order
        std::string e = tail_collect(tix, n, records, count, &open);
        if (e.empty()) e = tail_close_segment(tix, sorted_ids, dev_of, &open, &keys);
        tail_index_free(tix);
        if (e.empty()) e = build_tail_csr(&keys, n_pad, &ptr, &to);
        if (!e.empty()) return fail(c, HB_ERR_NOMEM, e);
        std::memcpy(ptr_out, ptr.data(), (n_pad + 1) * sizeof(uint64_t));
        *to_len = to.size();
        if (to_out && !to.empty()) std::memcpy(to_out, to.data(), std::min<uint64_t>(to.size(), to_cap) * sizeof(uint32_t));
        return HB_OK;
    });
}

int hb_debug_exchange(hb_ctx **ctxs, int count, int phase)
{
    return guarded(nullptr, [&]() -> int {
        if (!ctxs || count < 1 || !ctxs[0]) return HB_ERR_INVALID;
        hb_ctx *c = ctxs[0];
        for (int i = 0; i < count; i++) {
            hb_ctx *o = ctxs[i];
            if (!o) return fail(c, HB_ERR_INVALID, "NULL context");
            if (o->plan.n_pad != c->plan.n_pad || o->device != c->device || dest_mode(o) != dest_mode(c))
                return fail(c, HB_ERR_INVALID, "contexts differ in size, device or partition mode");
            if (phase == 0 && !o->pending_local)
                return fail(c, HB_ERR_INVALID, "every context must be between hb_step_local and hb_step_finish");
            if (dest_mode(c) && (o->opt.rank != i || o->opt.world_size != count))
                return fail(c, HB_ERR_INVALID, "destination partition: ctxs[i] must be rank i of `count`");
        }
        int rc = set_device(c);
        if (rc) return rc;
        for (int i = 0; i < count; i++) HB_HIP(hipStreamSynchronize(ctxs[i]->stream));
        const Plan &p = c->plan;
        const uint64_t S = c->slice_rows;
        if (phase == 1) {
            // what hb_finish's ncclAllGather of the Kahan-sum slices does (edge partition without a
            // communicator: every logical rank already holds all sums)
            if (!dest_mode(c)) return HB_OK;
            for (int i = 0; i < count; i++)
                for (int j = 0; j < count; j++)
                    if (i != j && S)
                        HB_HIP(hipMemcpyAsync(ctxs[i]->d_ksum + (uint64_t)j * S, ctxs[j]->d_ksum + (uint64_t)j * S, S * sizeof(double),
                                              hipMemcpyDeviceToDevice, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            return HB_OK;
        }
        if (!dest_mode(c) && edge_changed_only(c)) {
            // the changed-only all-reduce between logical ranks: union of the locally-changed bitmaps, every context packs
            // its rows of the union, the packed buffers are max-folded (the all-reduce), every context unpacks
            const uint64_t words = p.n_pad / 32;
            for (int i = 0; i < count; i++)
                if (!ctxs[i]->d_lbits || !edge_changed_only(ctxs[i])) return fail(c, HB_ERR_INVALID, "changed-only exchange: every context needs HB_FLAG_CHANGED_ONLY");
            HB_HIP(hipMemcpyAsync(c->d_ubits, c->d_lbits, words * 4, hipMemcpyDeviceToDevice, c->stream));
            for (int i = 1; i < count && words; i++)
                hipLaunchKernelGGL(hbk::or_words_kernel, dim3((unsigned)std::min<uint64_t>((words + 255) / 256, 2048)), dim3(256), 0, c->stream, c->d_ubits,
                                   (const uint32_t *)ctxs[i]->d_lbits, words);
            HB_HIP(hipGetLastError());
            for (int i = 1; i < count; i++) HB_HIP(hipMemcpyAsync(ctxs[i]->d_ubits, c->d_ubits, words * 4, hipMemcpyDeviceToDevice, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            for (int i = 0; i < count; i++) {
                int rc2 = edge_co_pack(ctxs[i]);
                if (rc2) return rc2;
                HB_HIP(hipStreamSynchronize(ctxs[i]->stream));
            }
            const uint64_t count4 = c->co_rows * 4;
            for (int i = 1; i < count && count4; i++) {
                hipLaunchKernelGGL(hbk::merge_max_kernel, dim3(2048), dim3(256), 0, c->stream, c->d_pack, (const uint4 *)ctxs[i]->d_pack, count4);
                HB_HIP(hipGetLastError());
            }
            for (int i = 1; i < count && count4; i++) HB_HIP(hipMemcpyAsync(ctxs[i]->d_pack, c->d_pack, count4 * 16, hipMemcpyDeviceToDevice, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            for (int i = 0; i < count; i++) {
                int rc2 = edge_co_unpack(ctxs[i]);
                if (rc2) return rc2;
                ctxs[i]->wire_bytes += (uint64_t)(count - 1) * words * 4 + 2 * (uint64_t)(count - 1) * ctxs[i]->co_rows * 64 / (uint64_t)count;
                HB_HIP(hipStreamSynchronize(ctxs[i]->stream));
            }
        } else if (!dest_mode(c)) {
            // all-reduce(max) of the pending counters: fold everything into ctxs[0], then copy out
            const uint64_t count4 = p.n_pad * 4;
            for (int i = 1; i < count && count4; i++) {
                hipLaunchKernelGGL(hbk::merge_max_kernel, dim3(2048), dim3(256), 0, c->stream, c->d_regs[c->cur ^ 1],
                                   (const uint4 *)ctxs[i]->d_regs[ctxs[i]->cur ^ 1], count4);
                HB_HIP(hipGetLastError());
            }
            for (int i = 1; i < count && count4; i++)
                HB_HIP(hipMemcpyAsync(ctxs[i]->d_regs[ctxs[i]->cur ^ 1], c->d_regs[c->cur ^ 1], count4 * 16, hipMemcpyDeviceToDevice, c->stream));
        } else {
            // all-gather of the owned slices (counters, changed bits) + sum of the changed counts
            const size_t W = hbk::kCounterWords;
            std::vector<unsigned long long> total(W, 0), cnt((size_t)count * W, 0);
            for (int i = 0; i < count; i++) {
                HB_HIP(hipMemcpyAsync(&cnt[(size_t)i * W], ctxs[i]->d_counters + W * ctxs[i]->t, W * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            }
            HB_HIP(hipStreamSynchronize(c->stream));
            for (int i = 0; i < count; i++)
                for (size_t k = 0; k < W; k++) total[k] += cnt[(size_t)i * W + k];
            const bool packed = changed_only(c);
            for (int i = 0; i < count; i++) {
                hb_ctx *d = ctxs[i];
                for (int j = 0; j < count; j++) {
                    if (i == j || !S) continue;
                    hb_ctx *o = ctxs[j];
                    if (!packed)
                        HB_HIP(hipMemcpyAsync(d->d_regs[d->cur ^ 1] + (uint64_t)j * S * 4, o->d_regs[o->cur ^ 1] + (uint64_t)j * S * 4, S * 64,
                                              hipMemcpyDeviceToDevice, c->stream));
                    HB_HIP(hipMemcpyAsync(d->d_bits[d->cur ^ 1] + (uint64_t)j * (S / 32), o->d_bits[o->cur ^ 1] + (uint64_t)j * (S / 32), S / 8,
                                          hipMemcpyDeviceToDevice, c->stream));
                }
                HB_HIP(hipMemcpyAsync(d->d_counters + W * d->t, total.data(), W * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
            }
            if (packed) {
                // the changed-only protocol with copies in place of the broadcasts: pack everywhere, move the runs, unpack
                HB_HIP(hipStreamSynchronize(c->stream));
                for (int i = 0; i < count; i++) {
                    int rc2 = exchange_pack(ctxs[i]);
                    if (rc2) return rc2;
                    HB_HIP(hipStreamSynchronize(ctxs[i]->stream));
                }
                for (int i = 0; i < count; i++)
                    for (int j = 0; j < count; j++) {
                        if (i == j) continue;
                        const uint64_t off = ctxs[j]->ex_off[j], cnt = ctxs[j]->ex_off[j + 1] - off;
                        if (cnt) HB_HIP(hipMemcpyAsync(ctxs[i]->d_pack + off * 4, ctxs[j]->d_pack + off * 4, cnt * 64, hipMemcpyDeviceToDevice, c->stream));
                    }
                HB_HIP(hipStreamSynchronize(c->stream));
                for (int i = 0; i < count; i++) {
                    int rc2 = exchange_unpack(ctxs[i]);
                    if (rc2) return rc2;
                    HB_HIP(hipStreamSynchronize(ctxs[i]->stream));
                }
            }
        }
        HB_HIP(hipStreamSynchronize(c->stream));
        return HB_OK;
    });
}

} // extern "C"
