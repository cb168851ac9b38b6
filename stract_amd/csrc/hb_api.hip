// hb_api.hip - C ABI (include/hyperball.h) + pass driver of the HyperBall library.
//
// Replaces HarmonicCentrality::calculate / calculate_centrality
// (crates/core/src/webgraph/centrality/harmonic.rs:215-287,292) behind a C boundary.
// Host orchestration only: every arithmetic step of the path runs in the gfx950 kernels
// of hb_kernels.hip.h.  There is no CPU fallback.
#include "hb_guard_alloc.h" // FIRST: no-op unless built with -DHB_GUARD_ALLOC=<mode> (debug allocators: guard pages / poison / red zones)
#include "hb_pool.h"        // then: every hipMalloc / hipFree below goes through the caching device allocator (shipped build)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hb_internal.h"
#include "hb_experiments.h" // HB_XBITS: the switches of the experiments build (none in the product library)
#include "hb_kernels.hip.h"
#ifdef HB_EXPERIMENTS
#include "hb_experiments.hip.h"
#endif
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
    hb_collectives coll{}; // hb_set_collectives: the per-pass exchanges go through these instead of RCCL (HB_FLAG_NO_RCCL contexts)
    std::string err;
    std::string arch;

    // host-side graph / plan
    std::vector<hb_edge> pending;  // hb_append_edges, host-ingest mode only
    // hb_append_edges, default: records are unpacked on the device as they arrive (2 x 16-byte endpoint keys + 1
    // flag byte each); nothing is buffered on the host
    IngestStream app;
    uint64_t lim_records = 0, lim_bytes = 0, lim_chunk = 0; // hb_debug_set_ingest_limits (0 = no limit / default)
    DenseGraph g;                  // ids kept; row_ptr/src kept only for hb_debug_copy_graph
    Plan plan;
    bool loaded = false, begun = false, finished = false;

    // device
    std::vector<DevBuf> allocs;
    uint64_t *d_row_ptr = nullptr;
    uint32_t *d_src = nullptr;
    uint16_t *d_src_jp = nullptr; // parallel to d_src: the sources' initial register (pass 0 streams it, hb_kernels.hip.h)
    bool level0_all_real = false;    // no row of the first hub-chunk level reads virtual rows (checked at load): pass 0 may run init_level1_kernel there
    uint32_t *d_virt_rows = nullptr; // one bit per work row: its sources are virtual rows (pass 0, PassParams::virt_rows)
    uint16_t *d_self_jp = nullptr; // per node row: its OWN initial register in the same format (the lean pass 0 reads it instead of the counters)
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
    void *h_block = nullptr;                  // ONE page-locked block, allocated by hb_create, that h_counters / h_slot / h_tl_count / h_rank_cnt are
                                              // carved from: no hipHostMalloc ever happens inside a run (the first hb_run of a process is the ONLY one
                                              // a drop-in user makes, entrypoint/centrality.rs:49)
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
    unsigned long long *d_rank_cnt = nullptr; // world words: every rank's changed rows of this pass (summed over the ranks)
    unsigned long long *h_rank_cnt = nullptr; // pinned
    // destination partition + changed-only: the pass' only host round trip sits in the MIDDLE of the pass (counters + run lengths,
    // before the broadcasts); what follows it is left in flight, so the pass' timing events are read later (resolve_pass_times)
    struct EvSet {
        hipEvent_t e[6];
    };
    std::vector<EvSet> ev_pool;         // deferred timing (destination partition + changed-only on a communicator): one set per pass of a run
    EvSet ev_ring[4]{};                 // hb_run's tail pipeline (two passes in flight): pass t uses set t & 3; created by hb_create
    std::vector<uint64_t> pending_times; // passes whose ms_* fields still have to be read from their events
    // hb_run's tail pipeline: pass q + 1 is queued (guarded on the device by pass q's changed count) before pass q's counters are
    // read, so the convergence tail runs without a host round trip between passes
    const unsigned long long *spec_guard = nullptr; // set while step_local queues a guarded pass
    bool pipelined = false;                         // step_local: take this pass' events from ev_pool
    uint64_t pipelined_passes = 0;                  // passes of this run queued ahead of their predecessor's read-back
    // the far tail as one workgroup (hb_tail.hip.h): work lists, their counts / status words, and whether the lists describe the
    // bitmaps as they are now (any pass run by other kernels invalidates them: the next entry collects them again)
    uint32_t *d_tl_changed[2] = {nullptr, nullptr}, *d_tl_vchanged[2] = {nullptr, nullptr}, *d_tl_dirty[2] = {nullptr, nullptr};
    uint32_t *d_tl_work = nullptr, *d_tl_count = nullptr;
    uint32_t *h_tl_count = nullptr; // pinned, kTcWords
    bool tl_valid = false;
    bool tl_declined = false;       // the kernel found the next pass too large for its lists: do not ask again before an ordinary pass has run
    uint64_t tail_kernel_passes = 0;
    hipEvent_t tl_ev[2] = {nullptr, nullptr};
    unsigned long long *h_slot = nullptr;           // pinned, 2 x kCounterWords: the counters of the two passes in flight
    hipEvent_t slot_done[2] = {nullptr, nullptr};   // pass q's counters have arrived in h_slot[q & 1]
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
    bool lean_init = false; // hb_begin left the initial counters / Kahan words / sizes to pass 0 (PassParams::rd_init); cleared by pass 0, or by
                            // ensure_initial_state() when something wants to look at the state before pass 0
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
    // [r6] the image's layout: out_len live entries.  Identity (out_len = n, index = sid, d_cid_of = d_sid_of) or, when the results travel in
    // stages (rs.on: single rank), COMPACT: one entry per node with in-edges, ascending NodeID (hb_aux.hip.h "the compact result image");
    // h_in_bits = those nodes as a bitmap over the sids (empty = identity)
    uint64_t out_len = 0;
    uint32_t *d_cid_of = nullptr; // n_pad: device row -> index into out[] (kNone = not in the image)
    std::vector<uint64_t> h_in_bits;
    uint64_t res_count = 0;
    // results that travel while the passes still run (results_stage below; hb_aux.hip.h results_sync_kernel)
    struct ResultSync {
        bool on = false;          // this graph ships its results in stages (single rank, large enough or forced by tune[1] bit 15)
        bool valid = false;       // d_out / d_sent (and h_out, once `copied` has fired) hold one consistent snapshot of the sums
        uint64_t changed_since = 0; // counters changed since that snapshot (= sums that moved, give or take a `+= 0.0` flush)
        uint32_t stages = 0;      // snapshots shipped during the current run
        double *d_sent = nullptr; // device order: the sum out[] was last built from
        uint32_t *d_sid = nullptr; // the final list: (sid, value) of what moved after the last snapshot
        double *d_val = nullptr;
        unsigned long long *d_count = nullptr;
        unsigned long long *d_kept = nullptr; // kCounterWords scratch words of the snapshots' kernel (side stream)
        uint32_t *h_sid = nullptr; // pinned
        double *h_val = nullptr;
        unsigned long long *h_count = nullptr;
        uint64_t cap = 0;
        hipStream_t stream = nullptr;                 // the snapshots' downloads run here, beside the passes
        hipEvent_t ready = nullptr, copied = nullptr; // out[] built (main stream) / downloaded (side stream)
    } rs;
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
    c->d_self_jp = nullptr;
    c->d_virt_rows = nullptr;
    c->level0_all_real = false;
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
    c->d_cid_of = nullptr;
    c->out_len = 0;
    std::vector<uint64_t>().swap(c->h_in_bits);
    c->d_pack = nullptr;
    c->d_wpop = nullptr;
    c->d_wprefix = nullptr;
    c->d_lbits = c->d_lbits_all = c->d_ubits = nullptr;
    c->d_rank_cnt = nullptr;
    c->ubits_valid = false;
    c->d_out_ptr = nullptr;
    c->d_out_rows = nullptr;
    c->d_touch = nullptr;
    c->d_seeds = c->d_heavy = nullptr;
    c->d_sparse_counts = nullptr;
    c->sparse_ok = false;
    c->d_tl_changed[0] = c->d_tl_changed[1] = c->d_tl_vchanged[0] = c->d_tl_vchanged[1] = c->d_tl_dirty[0] = c->d_tl_dirty[1] = nullptr;
    c->d_tl_work = c->d_tl_count = nullptr;
    c->tl_valid = false;
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
    if (c->rs.stream) (void)hipStreamSynchronize(c->rs.stream);
    for (void *q : {(void *)c->rs.h_sid, (void *)c->rs.h_val, (void *)c->rs.h_count})
        if (q) (void)hipHostFree(q);
    c->rs.h_sid = nullptr;
    c->rs.h_val = nullptr;
    c->rs.h_count = nullptr;
    c->rs.d_sent = nullptr;
    c->rs.d_sid = nullptr;
    c->rs.d_val = nullptr;
    c->rs.d_count = nullptr;
    c->rs.d_kept = nullptr;
    c->rs.cap = 0;
    c->rs.on = c->rs.valid = false;
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
// the context can exchange with the other ranks by itself: an RCCL communicator, or the caller's collectives
bool linked(const hb_ctx *c) { return c->comm != nullptr || c->coll.all_reduce != nullptr; }

#define HB_COLL(call)                 \
    do {                              \
        const int rc_coll_ = (call);  \
        if (rc_coll_) return rc_coll_; \
    } while (0)
int coll_dtype(ncclDataType_t dt) { return dt == ncclUint8 ? HB_COLL_U8 : dt == ncclUint32 ? HB_COLL_U32 : dt == ncclUint64 ? HB_COLL_U64 : HB_COLL_F64; }
uint64_t coll_size(ncclDataType_t dt) { return dt == ncclUint8 ? 1 : dt == ncclUint32 ? 4 : 8; }
// in place, like every exchange of the pass driver
int coll_all_reduce(hb_ctx *c, void *buf, uint64_t count, ncclDataType_t dt, ncclRedOp_t op, hipStream_t s)
{
    if (c->comm) {
        HB_NCCL(ncclAllReduce(buf, buf, count, dt, op, c->comm, s));
        return HB_OK;
    }
    if (c->coll.all_reduce(c->coll.user, buf, count, coll_dtype(dt), op == ncclMax ? HB_COLL_MAX : HB_COLL_SUM, (void *)s))
        return fail(c, HB_ERR_RCCL, "the caller's all_reduce (hb_set_collectives) failed");
    return HB_OK;
}
// recv holds world x count elements, send = this rank's count elements (may lie inside recv at its place)
int coll_all_gather(hb_ctx *c, const void *send, void *recv, uint64_t count, ncclDataType_t dt, hipStream_t s)
{
    if (c->comm) {
        HB_NCCL(ncclAllGather(send, recv, count, dt, c->comm, s));
        return HB_OK;
    }
    if (c->coll.all_gather(c->coll.user, send, recv, count * coll_size(dt), (void *)s))
        return fail(c, HB_ERR_RCCL, "the caller's all_gather (hb_set_collectives) failed");
    return HB_OK;
}
int coll_broadcast(hb_ctx *c, void *buf, uint64_t count, ncclDataType_t dt, int root, hipStream_t s)
{
    if (c->comm) {
        HB_NCCL(ncclBroadcast(buf, buf, count, dt, root, c->comm, s));
        return HB_OK;
    }
    if (c->coll.broadcast(c->coll.user, buf, count * coll_size(dt), root, (void *)s))
        return fail(c, HB_ERR_RCCL, "the caller's broadcast (hb_set_collectives) failed");
    return HB_OK;
}
int coll_group_start(hb_ctx *c)
{
    if (c->comm) HB_NCCL(ncclGroupStart());
    return HB_OK;
}
int coll_group_end(hb_ctx *c)
{
    if (c->comm) HB_NCCL(ncclGroupEnd());
    return HB_OK;
}
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
    return edge_partitioned(c) || (linked(c) && !dest_mode(c)) || (c->opt.flags & HB_FLAG_UNFUSED) || ref_tail(c);
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
#ifdef HB_EXPERIMENTS
    for (int k = 0; k < 2; k++) { // the tail kernel's lists (hb_tail.hip.h): 2.7 MB in all
        if ((rc = dev_alloc(c, &c->d_tl_changed[k], hbk::kTailCap))) return rc;
        if ((rc = dev_alloc(c, &c->d_tl_vchanged[k], hbk::kTailCap))) return rc;
        if ((rc = dev_alloc(c, &c->d_tl_dirty[k], hbk::kTailCap))) return rc;
    }
    if ((rc = dev_alloc(c, &c->d_tl_work, (size_t)(hbk::kTailLevels + 1) * hbk::kTailCap))) return rc;
    if ((rc = dev_alloc(c, &c->d_tl_count, hbk::kTcWords))) return rc;
#endif
    HB_HIP(hipMemsetAsync(c->d_touch, 0, (c->bits_words + 64) * sizeof(uint32_t), c->stream));
    {   // [r6] by sorting (hb_plan.hip gpu_transpose_rows: streaming traffic only); the scatter form below only when the device cannot
        // lend the sort 16 bytes per entry, or on request (experiments build, tune[1] bit 25: A/B runs and the parity variant that
        // keeps the fallback exact)
        const bool scatter = (HB_XBITS(c->opt.tune[1]) & 0x2000000u) != 0;
        if (!scatter) {
            const std::string e = gpu_transpose_rows((void *)c->stream, c->d_row_ptr, c->d_src, rows_total, entries, c->d_out_ptr, c->d_out_rows);
            if (e.empty()) {
                c->sparse_ok = true;
                return HB_OK;
            }
            if (e.find("out of memory") == std::string::npos) return fail(c, HB_ERR_HIP, e);
            (void)hipGetLastError();
        }
    }
    if ((rc = dev_alloc(c, &d_count, rows_total))) return rc; // stays allocated (small next to out_rows)
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

#include "hb_api_load.inc"

#include "hb_api_pass.inc"

// The C ABI never unwinds (include/hyperball.h): every entry point that can allocate runs under this guard.
template <class F>
int guarded(hb_ctx *c, F &&f)
{
    try {
        const int rc = f();
        HB_GUARD_CHECK("exit of a C-ABI entry point"); // red-zone debug build only (hb_guard_alloc.h)
#ifdef HB_DEBUG_BOUNDS
        if (c) { // a device-side index check failed somewhere in this call (hb_kernels.hip.h HB_DBG_ASSERT)
            unsigned int line = 0;
            (void)hipDeviceSynchronize();
            if (hipMemcpyFromSymbol(&line, HIP_SYMBOL(hbk::g_dbg_line), sizeof(line)) == hipSuccess && line)
                return fail(c, HB_ERR_INVALID, "HB_DEBUG_BOUNDS: device-side index check failed at kernel source line " + std::to_string(line));
        }
#endif
        return rc;
    } catch (const std::bad_alloc &) {
        try { return fail(c, HB_ERR_NOMEM, "out of host memory"); } catch (...) { return HB_ERR_NOMEM; }
    } catch (const std::exception &e) {
        try { return fail(c, HB_ERR_INVALID, std::string("C++ exception: ") + e.what()); } catch (...) { return HB_ERR_INVALID; }
    } catch (...) {
        try { return fail(c, HB_ERR_INVALID, "unknown C++ exception"); } catch (...) { return HB_ERR_INVALID; }
    }
}

// the device cannot hold the stream (memory, or a debug limit): the records appended so far come back as hb_edge records
// (hb_ingest.hip gpu_ingest_spill) and the stream continues on the host path
int spill_appended_to_host(hb_ctx *c)
{
    const std::string e = gpu_ingest_spill((void *)c->stream, &c->app, &c->pending);
    if (!e.empty()) return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : HB_ERR_HIP, e);
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
#ifndef HB_EXPERIMENTS
        if ((o.tune[1] & ~0xFFu) || o.tune[7])
            return fail(c, HB_ERR_INVALID, "hb_options.tune[1] bits above the low byte and tune[7] are switches of the experiments build (libhyperball_exp.so, "
                                           "-DHB_EXPERIMENTS: stract_amd/csrc/hb_experiments.h); the product library has none of them");
#endif
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
        {   // every small page-locked word the pass driver will ever read back, in one block, NOW: a hipHostMalloc inside the first run
            // cost that run 13 ms at C3 and ~200 ms at C4 under a loaded page cache (profiles/r05j_e2e_C4_first_run_and_store_phases.txt)
            const size_t world = (size_t)std::max(o.world_size, 1);
            const size_t words = (size_t)hbk::kCounterWords * 3 + 32 + world + 1; // h_counters | h_slot (2 sets) | h_tl_count (kTcWords x u32) | h_rank_cnt
            if (hipHostMalloc(&ctx->h_block, words * sizeof(unsigned long long)) != hipSuccess) { ctx->h_block = nullptr; ctx->err = "hipHostMalloc failed"; return bail(HB_ERR_NOMEM); }
            std::memset(ctx->h_block, 0, words * sizeof(unsigned long long));
            unsigned long long *w = (unsigned long long *)ctx->h_block;
            ctx->h_counters = w;
            ctx->h_slot = w + hbk::kCounterWords;
            ctx->h_tl_count = (uint32_t *)(w + 3 * (size_t)hbk::kCounterWords);
            ctx->h_rank_cnt = w + 3 * (size_t)hbk::kCounterWords + 32;
#ifdef HB_EXPERIMENTS
            static_assert(hbk::kTcWords * sizeof(uint32_t) <= 32 * sizeof(unsigned long long), "h_tl_count slot too small");
#endif
        }
        for (auto &e : ctx->slot_done)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(HB_ERR_HIP); }
        for (auto &e : ctx->tl_ev)
            if (hipEventCreate(&e) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(HB_ERR_HIP); }
        for (auto &es : ctx->ev_ring)
            for (auto &e : es.e)
                if (hipEventCreate(&e) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(HB_ERR_HIP); }
        if (hipStreamCreateWithFlags(&ctx->rs.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->rs.ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->rs.copied, hipEventDisableTiming) != hipSuccess) { ctx->err = "hipStreamCreate / hipEventCreate failed"; return bail(HB_ERR_HIP); }
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
    struct Trim { // after everything below has gone back to the caching allocator: free blocks return to the runtime
        ~Trim() { HB_POOL_TRIM(); }
    } trim_at_exit;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    ctx->app.free_all();
    free_graph_buffers(ctx);
    if (ctx->h_block) (void)hipHostFree(ctx->h_block); // h_counters, h_slot, h_tl_count, h_rank_cnt
    for (auto &es : ctx->ev_ring)
        for (hipEvent_t e : es.e)
            if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tl_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->slot_done)
        if (e) (void)hipEventDestroy(e);
    for (auto &es : ctx->ev_pool)
        for (hipEvent_t e : es.e)
            if (e) (void)hipEventDestroy(e);
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
    if (ctx->rs.ready) (void)hipEventDestroy(ctx->rs.ready);
    if (ctx->rs.copied) (void)hipEventDestroy(ctx->rs.copied);
    if (ctx->rs.stream) (void)hipStreamDestroy(ctx->rs.stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int hb_release_cached_memory(uint64_t *released)
{
    try {
        const size_t before = HB_POOL_RESERVED();
        HB_POOL_TRIM();
        if (released) *released = (uint64_t)(before - std::min<size_t>(before, HB_POOL_RESERVED()));
        return HB_OK;
    } catch (...) {
        return HB_ERR_NOMEM;
    }
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
        HB_POOL_RESET_PEAK(); // hb_stats.pool_peak_bytes counts from the beginning of this load
        double t0 = now_ms();
        // node/edge-set reduction: on the GPU (hb_ingest.hip) unless the host path is forced; identical output
        // The device pipeline keeps ~18 B per record resident at its peak (9 B held, 16 B during the sort):
        // when that cannot fit, or an allocation fails anyway, the host path produces the same graph.
        bool on_host = (c->opt.flags & HB_FLAG_HOST_INGEST) != 0 || (c->lim_records && m >= c->lim_records);
        if (!on_host) {
            size_t free_b = 0, total_b = 0;
            // held: 9 B per record + the endpoint table; sort: 16 B per record; ~60 B per node while the ids are sorted
            // + the endpoint table: 20 B per slot at a load factor of 1/4 .. 1/2 = up to 160 B per distinct endpoint;
            // their number is not known before the records are read - the caller's node list if there is one, else at most
            // two per record and (what every graph this library is meant for satisfies) no more than a tenth of the records
            const double endpoints = (node_ids && n) ? (double)n : std::min(2.0 * (double)m, std::max(0.1 * (double)m, 1e6));
            const double need = 18.0 * (double)m + 48.0 * (double)((node_ids && n) ? n : 0) + 160.0 * endpoints + 1024e6;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (double)free_b + (double)HB_POOL_CACHED_FREE()) on_host = true;
        }
        DeviceCsr csr;
        const bool keep_on_device = !on_host && device_plan(c);
        uint64_t peak = 0;
        std::string e = on_host ? ingest_edges(node_ids, n, edges, m, &c->g)
                                : gpu_ingest_edges((void *)c->stream, node_ids, n, edges, m, &c->g, keep_on_device ? &csr : nullptr, &peak);
        // (the id space of the device table exhausted is a limit of the DEVICE ingest only: endpoints outside a caller-supplied
        // node list are legal and ignored by the reference, store.rs:338-357 - the host path takes over, like for memory)
        if (!on_host && !e.empty() && (e.find("out of memory") != std::string::npos || e.find("OutOfMemory") != std::string::npos ||
                                       e.find("too many nodes for the device ingest") != std::string::npos)) {
            (void)hipGetLastError(); // clear the sticky allocation error
            peak = 0;
            e = ingest_edges(node_ids, n, edges, m, &c->g);
        }
        if (!e.empty())
            return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : (e.find("hip") != std::string::npos ? HB_ERR_HIP : HB_ERR_LIMIT), e);
        if ((rc = keep_owned(c, &csr))) {
            if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
            if (csr.d_src) (void)hipFree(csr.d_src);
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
            return rc;
        }
        const uint64_t nn = c->g.ids.size();
        const uint64_t m_eff = csr.d_row_ptr ? csr.m : (nn && c->g.row_ptr.size() == nn + 1 ? c->g.row_ptr[nn] : 0);
        c->stats.ms_ingest = now_ms() - t0;
        double ing = c->stats.ms_ingest;
        rc = plan_and_upload(c, csr.d_row_ptr ? &csr : nullptr, m_eff);
        if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr); // only if plan_and_upload bailed out before taking them
        if (csr.d_src) (void)hipFree(csr.d_src);
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
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
            if (err.find("out of memory") == std::string::npos && err.find("too many records") == std::string::npos &&
                err.find("too many nodes for the device ingest") == std::string::npos)
                return fail(c, HB_ERR_HIP, err);
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
            if (!c->pending.empty()) c->lim_records = 1; // the stream was spilled: it stays on the host path (m >= 1)
            rc = hb_load_edges(c, node_ids, n, c->pending.data(), c->pending.size());
            c->lim_records = keep_lim;
            std::vector<hb_edge>().swap(c->pending);
            return rc;
        }
        c->stats = hb_stats{};
        const double t0 = now_ms();
        const bool trace = std::getenv("HB_TRACE_INGEST") != nullptr;
        DeviceCsr csr;
        const bool keep_on_device = device_plan(c);
        uint64_t peak = 0;
        const std::string e = gpu_ingest_reduce((void *)c->stream, node_ids, n, &c->app, &c->g, keep_on_device ? &csr : nullptr, &peak);
        if (trace) std::fprintf(stderr, "[hb finalize] gpu_ingest_reduce returned after %.1f ms\n", now_ms() - t0);
        if (!e.empty())
            return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : (e.find("hip") != std::string::npos ? HB_ERR_HIP : HB_ERR_LIMIT), e);
        if ((rc = keep_owned(c, &csr))) {
            if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
            if (csr.d_src) (void)hipFree(csr.d_src);
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
            return rc;
        }
        const uint64_t nn = c->g.ids.size();
        const uint64_t m_eff = csr.d_row_ptr ? csr.m : (nn && c->g.row_ptr.size() == nn + 1 ? c->g.row_ptr[nn] : 0);
        const double ing = now_ms() - t0;
        rc = plan_and_upload(c, csr.d_row_ptr ? &csr : nullptr, m_eff);
        if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
        if (csr.d_src) (void)hipFree(csr.d_src);
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
        c->stats.ms_ingest = ing;
        c->stats.ingest_peak_bytes = peak;
        if (trace) std::fprintf(stderr, "[hb finalize] done after %.1f ms (plan %.1f ms, state %.1f ms)\n", now_ms() - t0, c->stats.ms_plan, c->stats.ms_h2d);
        return rc;
    });
}

int hb_discard_appended(hb_ctx *c)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        HB_HIP(hipStreamSynchronize(c->stream));
        c->app.free_all();
        std::vector<hb_edge>().swap(c->pending);
        return HB_OK;
    });
}

int hb_set_collectives(hb_ctx *c, const hb_collectives *ops)
{
    return guarded(c, [&]() -> int {
        if (!c) return HB_ERR_INVALID;
        if (c->comm) return fail(c, HB_ERR_INVALID, "hb_set_collectives: the context already has an RCCL communicator (create it with HB_FLAG_NO_RCCL)");
        if (c->loaded) return fail(c, HB_ERR_INVALID, "hb_set_collectives: set the collectives before the graph is loaded");
        if (!ops) {
            c->coll = hb_collectives{};
            return HB_OK;
        }
        if (!ops->all_reduce || !ops->all_gather || !ops->broadcast) return fail(c, HB_ERR_INVALID, "hb_set_collectives: all three functions are required");
        int rc = set_device(c);
        if (rc) return rc;
        if (!c->comm_stream) { // the merge / all-reduce / epilogue pipeline of the edge partition runs the exchange on its own stream
            HB_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
            for (int i = 0; i < hb_ctx::kOverlap; i++) {
                HB_HIP(hipEventCreateWithFlags(&c->ov_merged[i], hipEventDisableTiming));
                HB_HIP(hipEventCreateWithFlags(&c->ov_reduced[i], hipEventDisableTiming));
            }
        }
        c->coll = *ops;
        return HB_OK;
    });
}

int hb_debug_staged_copy(void *dst, const void *src, uint64_t bytes, int to_device, void *stream)
{
    hipError_t e = hipSuccess;
    if (bytes) e = hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? HB_OK : HB_ERR_HIP;
}

int hb_pinned_alloc(uint64_t bytes, void **out)
{
    if (!out) return HB_ERR_INVALID;
    *out = nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return HB_ERR_NOMEM;
    }
    *out = p;
    return HB_OK;
}

void hb_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int hb_debug_h2d_rate(hb_ctx *c, const void *host, uint64_t bytes, int reps, double *gb_per_s)
{
    return guarded(c, [&]() -> int {
        if (!c || !host || !bytes || reps < 1 || !gb_per_s) return HB_ERR_INVALID;
        int rc = set_device(c);
        if (rc) return rc;
        void *d = nullptr;
        HB_HIP(hipMalloc(&d, bytes));
        hipError_t e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream); // warm-up (first touch of the mapping)
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        const double t0 = now_ms();
        for (int r = 0; r < reps && e == hipSuccess; r++) e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        const double ms = now_ms() - t0;
        (void)hipFree(d);
        if (e != hipSuccess) return fail(c, HB_ERR_HIP, std::string("hb_debug_h2d_rate: ") + hipGetErrorString(e));
        *gb_per_s = (double)bytes * reps / (ms * 1e-3) / 1e9;
        return HB_OK;
    });
}

int hb_debug_set_ingest_limits(hb_ctx *c, uint64_t max_records, uint64_t max_device_bytes, uint64_t chunk_records)
{
    if (!c) return HB_ERR_INVALID;
    c->lim_records = max_records;
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
        HB_POOL_RESET_PEAK();
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
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
                return fail(c, he == hipErrorOutOfMemory ? HB_ERR_NOMEM : HB_ERR_HIP, std::string("uploading the graph: ") + hipGetErrorString(he));
            }
            csr.m = m_eff;
        }
        if ((rc = keep_owned(c, &csr))) {
            if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
            if (csr.d_src) (void)hipFree(csr.d_src);
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
            return rc;
        }
        const uint64_t m_local = csr.d_row_ptr ? csr.m : ((dest_mode(c) && n) ? c->g.row_ptr[n] : m_eff);
        double ing = now_ms() - t0;
        rc = plan_and_upload(c, csr.d_row_ptr ? &csr : nullptr, m_local);
        if (csr.d_row_ptr) (void)hipFree(csr.d_row_ptr);
        if (csr.d_src) (void)hipFree(csr.d_src);
            if (csr.d_id_lo) (void)hipFree(csr.d_id_lo);
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
        c->cur = 0;
        c->t = 0;
        c->lean_init = false;
        if (p.n_pad) {
            if (lean_pass0(c)) {
                // [r6] pass 0 produces the initial state itself (hb_kernels.hip.h PassParams::rd_init): hb_begin writes two bitmaps instead of
                // 88 bytes per node (C4: 8.7 GB, 3.3 ms of every run)
                hipLaunchKernelGGL(hbk::init_lean_kernel, dim3((unsigned)((p.n_pad + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t *)c->d_sid_of, p.n_pad,
                                   c->d_bits[0], c->d_kdirty);
                HB_HIP(hipGetLastError());
                c->lean_init = true;
            } else if ((rc = launch_full_init(c))) {
                return rc;
            }
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
        if (c->rs.stream) HB_HIP(hipStreamSynchronize(c->rs.stream)); // (a download of an abandoned run)
        c->rs.valid = false;
        c->rs.changed_since = 0;
        c->rs.stages = 0;
        c->t = 0;
        c->cur = 0;
        c->wire_bytes = 0;
        c->has_changes = true; // harmonic.rs:232
        c->last_changed = p.n;
        c->last_active = c->m_global;
        c->pending_local = false;
        c->pstats.clear();
        c->pending_times.clear();
        c->pipelined_passes = 0;
        c->tail_kernel_passes = 0;
        c->tl_valid = c->tl_declined = false;
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
        if (tail_kernel_ready(c)) { // the far tail: one single-workgroup launch runs the pass from work lists (hb_tail.hip.h)
            uint32_t ran = 0;
            if ((rc = tail_kernel_run(c, 1, &ran))) return rc;
            if (ran) {
                if (has_changes) *has_changes = c->has_changes ? 1 : 0;
                return HB_OK;
            }
        }
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
        if ((rc = ensure_initial_state(c))) return rc; // (hb_finish right behind hb_begin)
        if (linked(c) && p.n_pad) {
            // every rank ends with all Kahan sums: in-place all-gather of the owned slices
            HB_COLL(coll_all_gather(c, c->d_ksum + (uint64_t)c->opt.rank * c->slice_rows, c->d_ksum, c->slice_rows, ncclDouble, c->stream));
        }
        // normalize_centralities (harmonic.rs:178-195) on the device, in ascending-NodeID order;
        // norm_factor = (num_nodes - 1) as f64 (:229)
        const double norm = (double)(p.n ? p.n - 1 : 0);
        unsigned long long *cnt = c->d_counters + (size_t)c->max_passes * hbk::kCounterWords; // the spare slot
        HB_HIP(hipMemsetAsync(cnt, 0, hbk::kCounterWords * sizeof(unsigned long long), c->stream));
        bool shipped = false;
        if (p.n && c->rs.on && c->rs.valid) {
            // the bulk is already on the host (results_stage): what moved since follows as a (sid, value) list
            auto &rs = c->rs;
            HB_HIP(hipMemsetAsync(rs.d_count, 0, sizeof(unsigned long long), c->stream));
            // [r6] the list kernel does NOT wait for the last snapshot's download any more (C4: that copy - 794 MB, 14 ms - outlasts the four
            // sweep passes behind it by ~3 ms, and the kernel's 1.1 ms then came on top).  It rewrites out[sid] only for the entries it also
            // puts on the list, and the host applies the list AFTER the download has landed (hipEventSynchronize below): whichever of the
            // two values of such an entry the copy carried, the list's is the one that stays.  If the list overflows, out[] is shipped
            // whole behind the download instead.
            const unsigned blocks = (unsigned)std::min<uint64_t>((p.n_pad + 2047) / 2048, (uint64_t)c->num_cu * 8);
            hipLaunchKernelGGL(hbk::results_sync_kernel, dim3(blocks), dim3(256), 0, c->stream, (const double *)c->d_ksum, rs.d_sent,
                               (const uint32_t *)c->d_cid_of, p.n_pad, norm, 0, c->d_out, rs.d_sid, rs.d_val, (unsigned long long)rs.cap, rs.d_count, cnt);
            HB_HIP(hipGetLastError());
            HB_HIP(hipMemcpyAsync(rs.h_count, rs.d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipMemcpyAsync(c->h_counters, cnt, hbk::kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            const uint64_t moved = *rs.h_count;
            if (moved <= rs.cap) {
                if (moved) {
                    HB_HIP(hipMemcpyAsync(rs.h_sid, rs.d_sid, moved * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                    HB_HIP(hipMemcpyAsync(rs.h_val, rs.d_val, moved * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                    HB_HIP(hipStreamSynchronize(c->stream));
                }
                HB_HIP(hipEventSynchronize(rs.copied)); // the snapshot's download is over before the list is applied on top of it
                host_scatter_f64(c->h_out, rs.h_sid, rs.h_val, moved); // every sid occurs once: the shares are independent (hb_host.cpp, OpenMP team)
            } else { // more moved than the list holds: out[] on the device is complete anyway, ship it whole
                HB_HIP(hipStreamWaitEvent(c->stream, rs.copied, 0)); // (behind the snapshot's download: both write h_out)
                HB_HIP(hipMemcpyAsync(c->h_out, c->d_out, c->out_len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HB_HIP(hipStreamSynchronize(c->stream));
            }
            shipped = true;
        }
        if (p.n && !shipped) {
            if (c->rs.on) {
                // no snapshot was taken (the loop ended before the policy asked for one): the compact image whole, from the same kernel
                // (all rows, no list; sent[] is rewritten - nobody reads it before the next run's first snapshot does the same)
                const unsigned blocks = (unsigned)std::min<uint64_t>((p.n_pad + 2047) / 2048, (uint64_t)c->num_cu * 8);
                hipLaunchKernelGGL(hbk::results_sync_kernel, dim3(blocks), dim3(256), 0, c->stream, (const double *)c->d_ksum, c->rs.d_sent,
                                   (const uint32_t *)c->d_cid_of, p.n_pad, norm, 1, c->d_out, (uint32_t *)nullptr, (double *)nullptr, 0ull, c->rs.d_count, cnt);
            } else {
                unsigned blocks = (unsigned)std::min<uint64_t>((p.n + 255) / 256, (uint64_t)c->num_cu * 8);
                hipLaunchKernelGGL(hbk::finish_kernel, dim3(blocks), dim3(256), 0, c->stream, (const double *)c->d_ksum,
                                   (const uint32_t *)c->d_dev_of, p.n, norm, c->d_out, cnt);
            }
            HB_HIP(hipGetLastError());
            HB_HIP(hipMemcpyAsync(c->h_out, c->d_out, c->out_len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        if (!shipped) {
            HB_HIP(hipMemcpyAsync(c->h_counters, cnt, hbk::kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
        }
        if (std::getenv("HB_TRACE_RESULTS"))
            std::fprintf(stderr, "[hb results] hb_finish: entered at %.3f ms, done at %.3f ms (host clock); %u snapshots, list %llu\n", t0, now_ms(), c->rs.stages,
                         shipped ? (unsigned long long)*c->rs.h_count : 0ull);
        c->stats.pipelined_passes = c->pipelined_passes;
        c->stats.tail_kernel_passes = c->tail_kernel_passes;
        c->stats.result_stages = c->rs.stages;
        c->stats.result_list = shipped ? *c->rs.h_count : 0;
        c->rs.valid = false;
        c->res_count = 0;
        for (int s = 0; s < hbk::kStripes; s++) c->res_count += c->h_counters[4 * s];
        c->stats.ms_d2h = now_ms() - t0;
        c->stats.results = c->res_count;
        c->stats.passes = c->t;
        if ((rc = resolve_pass_times(c))) return rc;
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
        const bool trace = std::getenv("HB_TRACE_RESULTS") != nullptr;
        // harmonic.rs:237-240: loop { if !has_changes { break } ... }
        while (has) {
            if (tail_kernel_ready(c)) { // passes from work lists, several per launch, until the loop ends or the changed set outgrows the lists
                uint32_t ran = 0;
                if ((rc = tail_kernel_run(c, 64, &ran))) return rc;
                has = c->has_changes ? 1 : 0;
                if (ran) continue;
            }
            if (tail_pipeline_ready(c)) {
                if ((rc = tail_pipeline(c, &has))) return rc; // runs passes until the loop ends or the changed set grows again
                continue;
            }
            if ((rc = hb_step(c, &has))) return rc;
            if (trace) std::fprintf(stderr, "[hb results] pass %llu returned at %.3f ms (host clock)\n", (unsigned long long)c->t - 1, now_ms());
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

int hb_get_pass_stats(const hb_ctx *cc, uint64_t t, hb_pass_stats *out)
{
    if (!cc || !out || t >= cc->pstats.size()) return HB_ERR_INVALID;
    if (!cc->pending_times.empty()) {
        // timing events left in flight by a pass whose host round trip sits in its middle (destination partition + changed-only on a
        // communicator): read now, on the context's device and under the guard like every other call that talks to the runtime
        // (hb_finish resolves them too, so this only happens between hb_step calls)
        hb_ctx *c = const_cast<hb_ctx *>(cc);
        const int rc = guarded(c, [&]() -> int {
            const int rc_dev = set_device(c);
            return rc_dev ? rc_dev : resolve_pass_times(c);
        });
        if (rc) return rc;
    }
    *out = cc->pstats[t];
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
        // compaction of the per-node array (absent = negative) into the caller's buffers, on the host cores the process may use
        // (C4: 99 M nodes -> 79 M results = 1.9 GB written; one thread took 1.3 s of the 15 s chain store -> load -> run -> store)
        host_compact_results(c->h_out, c->g.ids.data(), c->plan.n, ids, vals, cap, c->h_in_bits.empty() ? nullptr : c->h_in_bits.data());
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
        std::string e = gpu_rank_results((void *)c->stream, c->d_out, c->out_len, c->res_count, ranks); // (the compact image keeps NodeID order: same ranks)
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
        std::string e = gpu_rank_results((void *)c->stream, c->d_out, c->out_len, c->res_count, nullptr, order.data(), top);
        if (!e.empty()) return fail(c, HB_ERR_HIP, e);
        // result index (ascending NodeID among the kept ones) -> (sid, index into the image); compact image: the k-th set bit of h_in_bits
        std::vector<uint32_t> kept, at;
        kept.reserve(c->res_count);
        const bool compact = !c->h_in_bits.empty();
        if (compact) at.reserve(c->res_count);
        uint64_t cid = 0;
        for (uint64_t sid = 0; sid < c->plan.n; sid++) {
            if (compact && !((c->h_in_bits[sid >> 6] >> (sid & 63u)) & 1ull)) continue;
            const uint64_t pos = compact ? cid++ : sid;
            if (c->h_out[pos] >= 0.0) {
                kept.push_back((uint32_t)sid);
                if (compact) at.push_back((uint32_t)pos);
            }
        }
        if (kept.size() != c->res_count) return fail(c, HB_ERR_INVALID, "hb_result_top: result buffer changed since hb_finish");
        for (uint64_t i = 0; i < top; i++) {
            const uint32_t sid = kept[order[i]];
            if (ids) ids[i] = c->g.ids[sid];
            if (vals) vals[i] = c->h_out[compact ? at[order[i]] : sid];
        }
        return HB_OK;
    });
}

// store_harmonic (centrality/mod.rs:72-114) straight from the context that computed the results [ABI 5]
int hb_store_harmonic_results(hb_ctx *c, const char *output, char *err, uint64_t err_len)
{
    if (err && err_len) err[0] = 0;
    const int rc = guarded(c, [&]() -> int {
        if (!c || !output || !*output) return c ? fail(c, HB_ERR_INVALID, "hb_store_harmonic_results: output is empty") : HB_ERR_INVALID;
        if (!c->finished) return fail(c, HB_ERR_INVALID, "no results: hb_run / hb_finish not called");
        int rc2 = set_device(c);
        if (rc2) return rc2;
        const uint64_t k = c->res_count;
        // the (NodeID, f64) list and the ranks, as hb_result_copy / hb_result_ranks return them
        RawVec<hb_u128> ids(k); // (uninitialised: hb_result_copy fills them on all host cores, the ranks come off the device)
        RawVec<double> vals(k);
        RawVec<uint64_t> ranks(k);
        const bool trace = std::getenv("HB_TRACE_STORE") != nullptr;
        double t_lap = now_ms();
        auto lap = [&](const char *what) {
            if (trace) std::fprintf(stderr, "[hb store] %-28s %8.3f s\n", what, (now_ms() - t_lap) * 1e-3);
            t_lap = now_ms();
        };
        if ((rc2 = hb_result_copy(c, ids.data(), vals.data(), k))) return rc2;
        lap("result list (host cores)");
        if ((rc2 = hb_result_ranks(c, ranks.data(), k))) return rc2;
        lap("ranks (device sort)");
        // the key order of both databases: one radix sort on the device instead of a comparison sort of 24-byte records on the host
        StoreKeyVec sorted(k);
        const std::string e = gpu_store_keys((void *)c->stream, ids.data(), k, sorted.data());
        if (!e.empty()) return fail(c, e.find("memory") != std::string::npos ? HB_ERR_NOMEM : HB_ERR_HIP, "hb_store_harmonic_results: " + e);
        lap("key order (device sort)");
        RawVec<hb_u128>().swap(ids);
        char msg[512] = {0};
        const int rc3 = store_harmonic_presorted(output, &sorted, vals.data(), ranks.data(), msg, sizeof(msg));
        if (rc3) return fail(c, rc3, msg);
        return HB_OK;
    });
    if (rc && err && err_len && c) std::snprintf(err, (size_t)err_len, "%s", c->err.c_str());
    return rc;
}

#include "hb_api_debug.inc"

} // extern "C"
