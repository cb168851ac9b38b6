/*
 * hyperball.h - C ABI of the MI355X-native HyperBall harmonic-centrality library.
 *
 * Drop-in boundary.  The reference call site is
 *
 *     crates/core/src/entrypoint/centrality.rs:46-53
 *         let graph = WebgraphBuilder::new(path, 0).open();
 *         let hc    = HarmonicCentrality::calculate(&graph);          // harmonic.rs:292
 *         store_harmonic(hc.iter().map(|(n, c)| (*n, c)), out);       // centrality/mod.rs:72
 *
 * A Rust shim keeps `pub struct HarmonicCentrality(BTreeMap<NodeID,f64>)`
 * (harmonic.rs:289-311) and replaces the body of `calculate` by:
 *     hb_create -> hb_load_edges(graph.host_nodes(), graph.host_edges()) -> hb_run
 *     -> hb_result_count / hb_result_copy -> BTreeMap -> hb_destroy.
 * Everything upstream (Webgraph loader) and downstream (speedy_kv centrality store
 * writer) is untouched.
 *
 * All functions are extern "C", never unwind, keep no global state.  One host thread
 * per context; calls on one context are not re-entrant.  Return value: 0 = HB_OK,
 * negative = error (message via hb_last_error).  The library needs a gfx950 GPU; there
 * is NO CPU fallback - every entry point that computes fails with HB_ERR_NO_DEVICE
 * when no device is present.
 */
#ifndef HYPERBALL_H
#define HYPERBALL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HB_ABI_VERSION 6 /* 6 [r6]: hb_options.tune[1] above its low byte and tune[7] are refused by the product library (experiments build
                            only); layouts unchanged since 5 */

/* ---- error codes -------------------------------------------------------------- */
#define HB_OK 0
#define HB_ERR_INVALID   (-1) /* bad argument / call order                         */
#define HB_ERR_NO_DEVICE (-2) /* no HIP device, or not gfx950                      */
#define HB_ERR_HIP       (-3) /* a HIP runtime call failed                         */
#define HB_ERR_NOMEM     (-4) /* host or device allocation failed                  */
#define HB_ERR_RCCL      (-5) /* an RCCL call failed                               */
#define HB_ERR_LIMIT     (-6) /* n >= 2^30 nodes, or max_passes exceeded            */
#define HB_ERR_IO        (-7) /* a file could not be created / written (hb_store.h) */

/* ---- plain data ---------------------------------------------------------------- */

/* Rust `u128` <-> two little-endian u64 halves.  NodeID(u128): webgraph/node.rs:37.
 * Ordering everywhere is the numeric u128 order (hi first, then lo). */
typedef struct hb_u128 {
    uint64_t lo;
    uint64_t hi;
} hb_u128;

/* SmallEdge { from: NodeID, to: NodeID, rel_flags: RelFlags(u64) }, webgraph/edge.rs:31-35;
 * 40 bytes, #[repr(C)] on the Rust side. */
typedef struct hb_edge {
    hb_u128 from;
    hb_u128 to;
    uint64_t rel_flags;
} hb_edge;

/* Edges with rel_flags & HB_SKIPPED_REL_MASK != 0 are dropped (after first-occurrence
 * de-duplication): SKIPPED_REL, harmonic.rs:36-49, bit values
 * webpage/html/links.rs:114-141. */
#define HB_SKIPPED_REL_MASK 0x6FED00ull

/* ---- options ------------------------------------------------------------------- */
#define HB_FLAG_NO_FRONTIER   0x01u /* never skip unchanged sources (debug; same results)          */
#define HB_FLAG_NO_REORDER    0x02u /* keep ascending-NodeID order as the device order            */
#define HB_FLAG_UNFUSED       0x04u /* run merge and estimator/Kahan as two kernels even on 1 GPU  */
#define HB_FLAG_PASS_STATS    0x08u /* also count active edges / touched rows per pass (A_t, V_t)  */
#define HB_FLAG_NO_XCD_MAP    0x20u /* plain blockIdx -> tile mapping, no XCD-affine chunk groups  */
#define HB_FLAG_NO_RCCL       0x40u /* world_size > 1 bookkeeping without a communicator: the caller
                                       performs the exchange (hb_debug_exchange; tests)       */
#define HB_FLAG_NO_SPARSE     0x100u /* never use the data-driven sweep passes (debug; same results)       */
#define HB_FLAG_DEST_PARTITION 0x200u /* world_size > 1: destination partition instead of edge partition -
                                        rank r owns the nodes whose rank in ascending NodeID order is
                                        r mod world_size and must be given ALL in-edges of those nodes
                                        (records for other nodes are ignored); one ncclAllGather of the
                                        owned counter slices per pass instead of an all-reduce          */
#define HB_FLAG_HOST_INGEST   0x400u /* hb_load_edges: reduce the records on the host (hb_host.cpp) instead of
                                        on the GPU (hb_ingest.hip); same result                      */
#define HB_FLAG_HOST_PLAN     0x800u /* build the device work layout on the host (hb_host.cpp) instead of on the GPU
                                        (hb_plan.hip); same layout, also for the destination partition                     */
#define HB_FLAG_CHANGED_ONLY  0x1000u /* with HB_FLAG_DEST_PARTITION: after the changed bits (all-gather) only the counters that
                                         changed in the pass travel (one ncclBroadcast of its packed run per rank) instead of
                                         the all-gather of whole slices: ~18 % fewer bytes in the dense passes of the R-MAT
                                         configs, ~100 % fewer in the tail; same results.
                                         With the edge partition: the ranks' "my local merge changed this row" bitmaps are
                                         all-gathered and OR-ed, and the ncclAllReduce(max, u8) runs over the rows of that union
                                         only (packed in the same order on every rank) instead of all n counters */
#define HB_FLAG_REFERENCE_TAIL 0x2000u /* reproduce the reference's changed-node machinery AS WRITTEN instead of its host-level
                                          meaning: the bloom filter of changed nodes with its false
                                          positives (harmonic.rs:221-225,133; bloom/src/lib.rs:85-123), the exact-counting
                                          switch (:277-279) and, once 0 < |changed| <= round(sqrt(n)), update_changed_counters
                                          (:75-114), which follows the PAGE-level records a ForwardlinksQuery on the host id
                                          returns (query/forwardlink.rs:95-101,153-173) - give them with hb_load_tail_edges.
                                          On a graph whose pages are hosts this equals the default; on a real crawl it is
                                          what `stract centrality harmonic` prints.  Single rank only; passes run unfused. */
#define HB_FLAG_NO_INIT_PASS  0x4000u /* pass 0 like every other dense pass (64-byte gathers) instead of streaming the sources'
                                         single initial register with the edge list (2 bytes per edge, written at load time);
                                         same results, measurement switch */
#define HB_FLAG_RCCL_SELF     0x80u /* world_size == 1 but still create a 1-rank communicator and run
                                       the collectives (exercises the RCCL call path on one GPU)   */

typedef struct hb_options {
    uint32_t struct_size;   /* = sizeof(hb_options); 0 is accepted as "this version"       */
    int32_t  device;        /* HIP device ordinal; < 0 = current device                     */
    uint32_t flags;         /* HB_FLAG_*                                                   */
    uint32_t chunk;         /* max sources per work row (hub rows are split); 0 = default  */
    uint32_t max_passes;    /* safety bound on passes; 0 = 4096                            */
    int32_t  rank;          /* edge-partition mode: this process' rank ...                 */
    int32_t  world_size;    /* ... of world_size (<= 1: single GPU, no collective)         */
    uint8_t  rccl_id[128];  /* ncclUniqueId from hb_rccl_unique_id() of rank 0             */
    uint32_t tune[8];       /* tuning knobs, 0 = default (every setting gives the same results; these only move time):
                             *  [0] workgroups per CU of the pass launches: low byte = node rows (dense 64, bitmap 32),
                             *      second byte = hub chunks (dense 2, bitmap 5)
                             *  [1] low byte: gather unroll 1|2|4 (hub chunks 4, node rows 2).  The bits above it are switches of the
                             *      experiments build only (stract_amd/csrc/hb_experiments.h); hb_create refuses them
                             *  [2] bitmap passes when A_t < tune[2] % of the edges (50; > 100 = always)
                             *  [3] log2 of the hotness slice width in counters (16 = 4 MiB; 1 = no slices)
                             *  [4] min sources of a chunk at a slice cut (8)
                             *  [5] largest row that is not split into chunks (chunk)
                             *  [6] sweep passes (touched rows only) when A_t * tune[6] < edges (10; 1 = whenever a bitmap pass would run)
                             *  [7] reserved, must be 0 (experiments build: see hb_experiments.h)                 */
} hb_options;

typedef struct hb_ctx hb_ctx;

/* ---- statistics ------------------------------------------------------------------ */
typedef struct hb_stats {
    uint64_t n;             /* |V| (harmonic.rs:58-72)                                     */
    uint64_t m_input;       /* edge records received                                       */
    uint64_t m_unique;      /* unique (from,to) pairs (store.rs:313)                        */
    uint64_t m_eff;         /* ... surviving the rel-flag filter (harmonic.rs:131)          */
    uint64_t passes;        /* T, including the final no-change pass (harmonic.rs:237-240)  */
    uint64_t results;       /* nodes with centrality > 0                                    */
    double   ms_ingest;     /* host: dedup / remap / CSR build                              */
    double   ms_plan;       /* host: device ordering, hub-row splitting                     */
    double   ms_h2d;        /* graph upload + state initialisation                          */
    double   ms_loop;       /* wall time of the iteration loop (the timed quantity)         */
    double   ms_loop_gpu;   /* sum of per-pass GPU time (HIP events on the ctx stream)      */
    double   ms_collective; /* part of ms_loop_gpu spent in RCCL collectives                */
    double   ms_d2h;        /* result download + id mapping                                 */
    uint64_t work_rows;     /* real + virtual (hub-chunk) rows                              */
    uint64_t virtual_rows;
    uint64_t device_bytes;  /* device memory held by the context                            */
    uint64_t virtual_edges; /* entries of the virtual rows' source lists, all levels        */
    uint64_t levels;        /* virtual levels = hub-kernel launches per pass                */
    uint64_t level1_edges;  /* REAL edges gathered by the level-1 hub-chunk launch (the dominant kernel) */
    uint64_t level1_rows;   /* hub-chunk rows of level 1 (padding rows excluded)            */
    uint64_t direct_edges;  /* REAL edges gathered directly by the node-row launch          */
    uint64_t rows_with_in_edges; /* nodes with >= 1 in-edge: V_t of a dense pass t >= 1      */
    uint64_t wire_bytes;    /* counter + changed-bit bytes this rank received in the collectives of the last run
                               (what HB_FLAG_CHANGED_ONLY reduces), both decompositions               */
    uint64_t ingest_peak_bytes; /* high-water mark of the device memory the GPU ingest NEEDED: live record chunks,
                                   endpoint table and work arrays (0 = host ingest / hb_load_dense)   */
    uint64_t pool_peak_bytes;   /* [ABI 4] high-water mark of what the library's caching device allocator (hb_pool.h) HELD
                                   from the runtime during the load: the live bytes plus freed extents it kept for reuse
                                   (kept only while the device has room: above half of the device memory - or
                                   HB_POOL_LIMIT_BYTES - free blocks go back to the runtime before a new one is taken) */
    uint64_t result_stages;     /* [ABI 5] snapshots of the per-node sums that went to the host WHILE the passes of the last run
                                   were still running (second stream; single rank, n >= 2^20): hb_finish then ships only ... */
    uint64_t result_list;       /* [ABI 5] ... this many (node, value) entries that still moved afterwards (0 stages: the whole
                                   n x 8-byte image is downloaded by hb_finish, as before)                                      */
    uint64_t pipelined_passes;  /* [ABI 5] passes of the last hb_run that were queued BEFORE the host had read the previous pass'
                                   counters (convergence tail, one rank: a pass that changed <= 4096 nodes in sweep mode is
                                   followed by passes guarded on the device; the pass behind the loop's last one does nothing) */
    uint64_t tail_kernel_passes; /* [ABI 5] always 0 in the product library (experiments build: passes one single-workgroup launch ran
                                   from work lists; measured no faster than the launches it replaces, so not shipped)          */
} hb_stats;

typedef struct hb_pass_stats {
    uint64_t pass;          /* t                                                            */
    uint64_t changed;       /* nodes whose counter changed in pass t                        */
    uint64_t active_edges;  /* A_t = edges whose source changed in pass t-1 (out-degree sum of those
                               nodes; with HB_FLAG_PASS_STATS counted edge by edge in frontier passes) */
    uint64_t touched;       /* frontier / sparse passes: node rows with >= 1 gathered source (<= V_t: a split
                               row counts only when one of its partials changed); dense passes: 0   */
    uint32_t mode;          /* 0 = dense (no frontier test), 1 = frontier bitmap, 2 = sweep (touch bitmap of the rows
                               that read a changed node; only those rows run), 3 = reference tail
                               (HB_FLAG_REFERENCE_TAIL: update_changed_counters over the page-level records)  
                               (4 = experiments build only: the pass ran inside the single-workgroup tail kernel)                 */
    float    ms_gpu;        /* GPU time of the pass (all its launches + collective)         */
    float    ms_main;       /* GPU time of the dominant launch (real rows)                  */
    float    ms_collective;
    float    ms_level1;     /* GPU time of the level-1 hub-chunk launch (dense / frontier passes); sweep passes: of the
                               seed collection + expansion launches (then ms_main = node rows, the rest = virtual levels) */
    uint32_t reserved;
} hb_pass_stats;

/* ---- lifecycle --------------------------------------------------------------------- */
int  hb_abi_version(void);
/* opt may be NULL (defaults, current device). */
int  hb_create(const hb_options *opt, hb_ctx **out);
void hb_destroy(hb_ctx *ctx);
/* Message of the last failing call on ctx (or of hb_create when ctx == NULL). */
const char *hb_last_error(const hb_ctx *ctx);
/* [ABI 5] The library allocates device memory through a caching allocator (freed extents are kept for the next load: hipMalloc /
 * hipFree cost ~60 ms per GB cycled on these boxes); other allocators in the process (RCCL, torch, rocPRIM users) cannot reclaim
 * what it holds on their own out-of-memory.  This returns every cached block of the CURRENT device that no context uses to the
 * runtime (hb_destroy does the same); *released = bytes given back (may be NULL). */
int hb_release_cached_memory(uint64_t *released);

/* ---- graph input --------------------------------------------------------------------- */
/* Replaces the per-pass `graph.host_nodes()` / `graph.host_edges()` streaming
 * (webgraph/mod.rs:157,192 -> store.rs:297-357) by one hand-over.
 *   node_ids: graph.host_nodes() in any order, duplicates allowed; NULL/n = 0 -> the node
 *             set is derived from the endpoints of ALL edge records, flagged ones included
 *             (store.rs:338-357).
 *   edges:    graph.host_edges() order matters: the FIRST record of each (from,to) pair
 *             wins (itertools::unique_by, store.rs:313), THEN records with
 *             rel_flags & HB_SKIPPED_REL_MASK are dropped (harmonic.rs:131).  Records whose
 *             endpoint is not in the node set are ignored (harmonic.rs:135).
 * Buffers are copied; the caller may free them on return.  In edge-partition mode every
 * rank passes the full node set and its own subset of the records; all records of one
 * (from,to) pair must go to the same rank, in stream order. */
int hb_load_edges(hb_ctx *ctx, const hb_u128 *node_ids, uint64_t n, const hb_edge *edges,
                  uint64_t m);
/* Streamed variant for callers that cannot (or need not) hold all records: append any number of batches in stream
 * order, then finalize (node_ids as above).  Every batch is uploaded at once and reduced as it arrives: each endpoint
 * is looked up in / added to a device hash table that maps NodeIDs to 32-bit provisional ids, and the record stays
 * resident as (from id, to id) + 1 flag byte = 9 bytes, in chunks (no reallocation as the stream grows); nothing is
 * buffered on the host, so the caller's peak memory is one batch.  There is no record-count limit (round 3: 2^32 - 256;
 * stream positions are no longer stored - the stable sort keeps the stream order of equal pairs).  Device memory: 9 B
 * per record + 20 B per table slot (load factor <= 1/2) while the stream is held, 16-17 B per record + ~40 B per node at
 * hb_finalize (hb_stats.ingest_peak_bytes reports the high-water mark).  With HB_FLAG_HOST_INGEST, or if the device
 * runs out of memory mid-stream, the records are buffered on the host instead (same result).  Batches in pinned host
 * memory (hipHostMalloc / hipHostRegister) cross the link asynchronously at its full rate. */
int hb_append_edges(hb_ctx *ctx, const hb_edge *edges, uint64_t m);
int hb_finalize(hb_ctx *ctx, const hb_u128 *node_ids, uint64_t n);
/* Drops the batches appended since the last hb_finalize (device chunks, endpoint table, host buffer): the stream starts
 * empty again.  For callers that find out mid-stream that their source is damaged (hb_load_webgraph on a CRC mismatch). */
int hb_discard_appended(hb_ctx *ctx);

/* HB_FLAG_REFERENCE_TAIL only; after the graph is loaded, before hb_begin / hb_run.  records = the store's page-level
 * (from_id, to_id, rel_flags) documents, SEGMENT BY SEGMENT IN DOC ORDER (sort_score ascending, store.rs:67-72) - the
 * order decides what `graph.search(ForwardlinksQuery::new(h).with_limit(Unlimited))` returns (harmonic.rs:82-87): the
 * query runs one LinksScorer per segment over the documents whose from_id is h, which skips self links and every
 * document whose to_id equals the to_id of the document it yielded LAST (adjacent de-duplication, plus a skip-list
 * shortcut over whole 128-document blocks; query/raw/links.rs:115-232) BEFORE harmonic.rs:87 looks at the flags of
 * what is left.  So of two neighbouring documents (h -> x) with different flags only the first one's flags count.
 * The library replays exactly that per segment and host, then applies the rel filter (:87) and the two counter
 * lookups (:91-92).  Any superset of the relevant documents may be passed (documents whose from_id is no host node
 * are dropped at once); hb_tail_segment_end() marks the end of a segment, hb_begin closes the last one.
 * hb_load_tail_edges replaces the records of earlier calls (one segment, or the first part of one); count == 0 =
 * "the query finds nothing" (also the state before the first call). */
int hb_load_tail_edges(hb_ctx *ctx, const hb_edge *records, uint64_t count);
/* The same in batches (a crawl's page-level documents do not fit one array): a batch continues the current segment;
 * 24 bytes per document whose from_id is a host node stay on the host until the segment ends, 8 bytes per surviving
 * record after.  The index is built and uploaded by the next hb_begin / hb_run. */
int hb_append_tail_edges(hb_ctx *ctx, const hb_edge *records, uint64_t count);
/* The documents appended since the last call (or since hb_load_tail_edges) were one whole segment. */
int hb_tail_segment_end(hb_ctx *ctx);

/* Pre-reduced input (bench / large synthetic graphs): sorted_ids strictly ascending;
 * in-edges of node v (the v-th smallest id) are src[row_ptr[v] .. row_ptr[v+1]), already
 * unique and flag-filtered.  In edge-partition mode: full id list, local edge subset. */
int hb_load_dense(hb_ctx *ctx, const hb_u128 *sorted_ids, uint64_t n, const uint64_t *row_ptr,
                  const uint32_t *src, uint64_t m_eff);

/* ---- compute ----------------------------------------------------------------------------- */
/* Replaces calculate_centrality (harmonic.rs:215-287): runs passes until one changes
 * nothing, then normalises.  Blocking.  stats may be NULL. */
int hb_run(hb_ctx *ctx, hb_stats *stats);

/* Step-wise form of the same loop (parity tests, multi-rank drivers):
 *   hb_begin        initialize (harmonic.rs:53-73,219-235): counters, Kahan sums, t = 0
 *   hb_step         one loop body (harmonic.rs:237-280); *has_changes as in the reference
 *   hb_finish       normalize_centralities (harmonic.rs:178-195, :282)
 * hb_run == hb_begin; while (has_changes) hb_step; hb_finish. */
int hb_begin(hb_ctx *ctx);
int hb_step(hb_ctx *ctx, int *has_changes);
int hb_finish(hb_ctx *ctx);

int hb_get_stats(const hb_ctx *ctx, hb_stats *out);
/* Stats of pass t (0 <= t < passes). */
int hb_get_pass_stats(const hb_ctx *ctx, uint64_t t, hb_pass_stats *out);

/* ---- results: HarmonicCentrality::iter / len (harmonic.rs:296-311) ---------------------- */
/* Number of nodes with centrality > 0 (absent key <=> centrality <= 0). */
int hb_result_count(hb_ctx *ctx, uint64_t *count);
/* Ascending NodeID, only centrality > 0, value = sum / (n-1) (non-finite -> 0.0).
 * Writes min(count, cap) entries; ids or vals may be NULL. */
int hb_result_copy(hb_ctx *ctx, hb_u128 *ids, double *vals, uint64_t cap);

/* The second half of store_harmonic (centrality/mod.rs:92-103): ranks[j] = position of the j-th result
 * (hb_result_copy order) when all results are sorted by (Reverse(f64::total_cmp(centrality)), NodeID
 * ascending) - the value written to the "harmonic_rank" store.  Computed on the GPU (stable radix sort).
 * cap must be >= hb_result_count. */
int hb_result_ranks(hb_ctx *ctx, uint64_t *ranks, uint64_t cap);

/* top_nodes(&store, TopNodes::Top(k)) (centrality/mod.rs:33-52), which build_harmonic calls with k = 1_000_000 for
 * harmonic.csv (entrypoint/centrality.rs:55-68): the min(k, count) results with the largest centrality, descending
 * (f64::total_cmp); ties in ascending NodeID = the first k of the harmonic_rank order.  (The reference's sorted_k is an
 * unstable sort keyed by the centrality alone: the order of ties, and which of them make the cut, are unspecified
 * there.)  ids or vals may be NULL; *written = number of entries. */
int hb_result_top(hb_ctx *ctx, uint64_t k, hb_u128 *ids, double *vals, uint64_t *written);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) -------------------------------------- */
/* Rank 0 calls this and distributes the 128 bytes (e.g. torch.distributed broadcast);
 * every rank puts them in hb_options.rccl_id. */
int hb_rccl_unique_id(uint8_t out[128]);

/* The exchanges of a pass through the CALLER's collectives instead of RCCL [r4]: a context created with world_size > 1 and
 * HB_FLAG_NO_RCCL behaves exactly like one with a communicator (same kernels, same row slices, same event ordering of the
 * merge / all-reduce / epilogue pipeline, same changed-only packing) once these three functions are set - every
 * ncclAllReduce / ncclAllGather / ncclBroadcast of the pass driver goes through them.  Purpose: (1) the multi-process
 * protocol can run as real library code on ONE device, N processes exchanging through host-staged torch.distributed
 * (gloo) tensors - tests/test_gpu.py, since RCCL refuses two ranks on one device; (2) an integrator whose ranks are
 * already connected by another fabric (MPI, the reference's own DHT) needs no RCCL bootstrap.
 * All pointers are DEVICE pointers; `stream` is the hipStream_t the call is ordered on (work queued on it before the call
 * must be visible to the exchange, work queued after it must see the result; a blocking implementation simply
 * synchronises the stream first).  Return 0, or non-zero to fail the pass with HB_ERR_RCCL.  Set before loading the graph. */
#define HB_COLL_U8 0
#define HB_COLL_U32 1
#define HB_COLL_U64 2
#define HB_COLL_F64 3
#define HB_COLL_MAX 0
#define HB_COLL_SUM 1
typedef struct hb_collectives {
    void *user;
    /* in place over `count` elements of `dtype` on every rank */
    int (*all_reduce)(void *user, void *buf, uint64_t count, int dtype, int op, void *stream);
    /* recv = world x bytes_per_rank bytes, rank r's part at r x bytes_per_rank; send may be that very part of recv */
    int (*all_gather)(void *user, const void *send, void *recv, uint64_t bytes_per_rank, void *stream);
    /* `bytes` bytes at buf from rank `root` to every rank */
    int (*broadcast)(void *user, void *buf, uint64_t bytes, int root, void *stream);
} hb_collectives;
int hb_set_collectives(hb_ctx *ctx, const hb_collectives *ops);
/* helper for host-staged implementations of the above: copy on `stream`, then synchronise it (to_device: host -> device) */
int hb_debug_staged_copy(void *dst, const void *src, uint64_t bytes, int to_device, void *stream);

/* ---- pinned batch buffers ------------------------------------------------------------------- */
/* Page-locked host memory for the record batches of hb_append_edges / hb_load_edges (hipHostMalloc): a batch that lies
 * in pinned memory crosses the host link asynchronously at its full rate (57 GB/s measured on the MI355X boxes,
 * profiles/r04a_h2d_probe.txt) and overlaps with the reduction of the batch before it; from pageable memory the
 * runtime has to stage it.  A caller that owns its buffer can pin it in place with hipHostRegister instead.
 * hb_pinned_free(NULL) is a no-op. */
int hb_pinned_alloc(uint64_t bytes, void **out);
void hb_pinned_free(void *p);

/* ---- device helpers for harnesses -------------------------------------------------------- */
int hb_device_count(int *count);
/* Fills name (NUL-terminated, <= cap) with the gcnArchName of the ctx device. */
int hb_device_name(const hb_ctx *ctx, char *name, uint64_t cap);
int hb_device_synchronize(hb_ctx *ctx);

/* ---- test / bench-only exports (NOT part of the drop-in contract) ------------------------- */
/* Current counters, one 64-byte register block per node, in ascending-NodeID order
 * (HyperLogLog::registers, hyperloglog.rs:4544).  Valid after hb_begin / any hb_step. */
int hb_debug_copy_registers(hb_ctx *ctx, uint8_t *out /* n*64 */);
/* KahanSum state per node, ascending-NodeID order (kahan_sum.rs:30-33). */
int hb_debug_copy_kahan(hb_ctx *ctx, double *sum, double *err);
/* Cached HyperLogLog::size() of the current counter per node. */
int hb_debug_copy_sizes(hb_ctx *ctx, uint64_t *out);
/* Runs the device estimator (HyperLogLog::size, hyperloglog.rs:4484-4516) on `count`
 * arbitrary 64-byte register blocks. */
int hb_debug_hll_size(hb_ctx *ctx, const uint8_t *regs, uint64_t count, uint64_t *out);
/* Order-independent 64-bit checksums of the current state: out[0] over all counters, out[1] over the
 * Kahan (sum, err) bit patterns; node v (v-th smallest NodeID) contributes a mix of (v, its words), summed
 * mod 2^64 (definition: oracle/hb_oracle.c hbo_dense_state_hash computes the same function).  For per-pass
 * parity at sizes where n*64 bytes are too many to ship.  Multi-rank contexts: out[1] = 0 (a rank holds the
 * Kahan state of its own rows only); out[0] covers all counters (every rank holds them after the collective). */
int hb_debug_state_hash(hb_ctx *ctx, uint64_t out[2]);
/* Reduced graph as the library sees it after ingest (ascending-NodeID indexing):
 * any pointer may be NULL; row_ptr has n+1 entries, src has m_eff.  row_ptr / src are kept on the host only for
 * graphs of up to 2^26 edges, or with HB_FLAG_HOST_PLAN (HB_ERR_LIMIT otherwise). */
int hb_debug_copy_graph(hb_ctx *ctx, hb_u128 *ids, uint64_t *row_ptr, uint32_t *src);
/* The device work layout of the loaded graph as it lies in HBM (same conventions as hb_host_plan: two-call
 * pattern, sizes = {n_pad, nv, plan_src_len, levels}; order[n_pad], plan_row_ptr[n_pad+nv+1], plan_src[plan_src_len],
 * level_begin[levels+1]).  The device planner must reproduce the host planner's layout entry for entry. */
int hb_debug_copy_plan(hb_ctx *ctx, uint64_t sizes[4], uint32_t *order, uint64_t *plan_row_ptr, uint32_t *plan_src,
                       uint64_t *level_begin);
/* Emulates the collective of one pass between `count` logical ranks that live on ONE device
 * (contexts created with HB_FLAG_NO_RCCL; RCCL refuses two ranks on one device).
 *   phase 0, between hb_step_local and hb_step_finish of every context:
 *     edge partition:        all-reduce(max) of the pending counters
 *     destination partition: all-gather of the owned counter / changed-bit slices, sum of the
 *                            changed counts (ctxs[i] must be rank i)
 *   phase 1, before hb_finish: all-gather of the Kahan-sum slices. */
int hb_debug_exchange(hb_ctx **ctxs, int count, int phase);
/* Two halves of hb_step: local pull into the pending counters; [collective, or hb_debug_exchange];
 * then (edge partition / HB_FLAG_UNFUSED) estimator/Kahan/changed detection, bookkeeping. */
int hb_step_local(hb_ctx *ctx);
int hb_step_finish(hb_ctx *ctx, int *has_changes);

/* Bench hook: the rate (GB/s) at which `bytes` bytes at `host` reach the device through hipMemcpyAsync on the library's stream,
 * mean of `reps` copies into a scratch buffer - what hb_append_edges can get from this buffer in this process. */
int hb_debug_h2d_rate(hb_ctx *ctx, const void *host, uint64_t bytes, int reps, double *gb_per_s);
/* Test hook: lowers the limits of the device ingest so that its refusal (>= max_records -> host ingest), its
 * out-of-memory spill (chunk memory beyond max_device_bytes counts as a failed allocation) and its multi-chunk
 * paths (chunk_records per chunk) can be reached with small inputs.  0 = default for each. */
int hb_debug_set_ingest_limits(hb_ctx *ctx, uint64_t max_records, uint64_t max_device_bytes, uint64_t chunk_records);

/* ---- host-only test exports (no device needed) -------------------------------------------- */
/* The reference ingest semantics alone (node set, first-occurrence de-duplication, flag
 * filter; store.rs:297-357, harmonic.rs:131).  Two-call pattern: with ids/row_ptr/src NULL
 * only the counts are returned; then row_ptr needs n+1 and src m_eff entries. */
int hb_host_ingest(const hb_u128 *node_ids, uint64_t n, const hb_edge *edges, uint64_t m,
                   uint64_t *n_out, uint64_t *m_unique, uint64_t *m_eff, hb_u128 *ids,
                   uint64_t *row_ptr, uint32_t *src);
/* The device work layout the planner would build for a reduced graph (device order +
 * hub-row splitting), so its invariants can be checked on the host.  flags: HB_FLAG_NO_REORDER, HB_FLAG_NO_XCD_MAP.
 * Two-call pattern: sizes[0..3] = {n_pad, nv, plan_src_len, levels}; then
 * order[n_pad] (0xFFFFFFFF = padding row), plan_row_ptr[n_pad+nv+1], plan_src[plan_src_len],
 * level_begin[levels+1].  tune[7] > 1 lays the rows out as tune[7] owner slices (destination partition). */
int hb_host_plan(uint64_t n, const uint64_t *row_ptr, const uint32_t *src, uint32_t flags,
                 uint32_t chunk, const uint32_t *tune /* hb_options.tune or NULL */, uint64_t sizes[4],
                 uint32_t *order, uint64_t *plan_row_ptr, uint32_t *plan_src, uint64_t *level_begin);
/* Host only (no GPU): the index hb_load_tail_edges / hb_append_tail_edges build from page-level records (taken as ONE
 * segment in doc order) - CSR by SOURCE device row (dev_of[sid]) over the records the per-host LinksScorer yields that
 * pass the rel filter and whose two ids are nodes (harmonic.rs:87,91-92), duplicates dropped.  ptr_out: n_pad + 1 offsets; to_out: target device rows (first to_cap), *to_len = their number. */
int hb_debug_tail_index(uint64_t n, const hb_u128 *sorted_ids, const uint32_t *dev_of, uint64_t n_pad, const hb_edge *records,
                        uint64_t count, uint64_t *ptr_out, uint32_t *to_out, uint64_t to_cap, uint64_t *to_len);

#ifdef __cplusplus
}
#endif
#endif /* HYPERBALL_H */
