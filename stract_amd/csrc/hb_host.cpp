// hb_host.cpp - host side of the HyperBall library: ingest (reference node/edge-set
// semantics) and the device work-layout planner.  No GPU code here.
//
// Reference semantics restated (nothing is copied from it):
//   node set   = every from_host_id / to_host_id of every record, flagged ones included
//                (crates/core/src/webgraph/store.rs:338-357)
//   edge set   = first record of each (from,to) pair in stream order
//                (itertools::unique_by, store.rs:313), THEN dropped when
//                rel_flags & SKIPPED_REL != 0 (harmonic.rs:36-49,131)
//   node order = numeric u128 order (BTreeMap<NodeID,_>, node.rs:33-37)
#include "hb_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <numeric>

#include <omp.h> // (the query functions are linked in every build; a compiler that ignores the pragmas runs the loops serially)

#include "hb_threads.h"
#ifdef _OPENMP
#include <parallel/algorithm>
#define HB_SORT(b, e) __gnu_parallel::sort((b), (e))
#define HB_SORT_CMP(b, e, c) __gnu_parallel::sort((b), (e), (c))
#define HB_STABLE_SORT_CMP(b, e, c) __gnu_parallel::stable_sort((b), (e), (c))
#else
#define HB_SORT(b, e) std::sort((b), (e))
#define HB_SORT_CMP(b, e, c) std::sort((b), (e), (c))
#define HB_STABLE_SORT_CMP(b, e, c) std::stable_sort((b), (e), (c))
#endif

namespace hb {

// Caps the OpenMP team for the duration of a host stage by the amount of work: on a many-core box small
// inputs are far slower with every hardware thread (fork/join, idle spinning) than with a few.
// CPUs this process may really use: min(OpenMP default, affinity mask, cgroup v2/v1 CPU quota).  Containers
// often expose every hardware thread of the host while the quota is a fraction of it; an OpenMP team larger
// than the quota is slower than a smaller one (measured on the GPU box: 64 of 256 threads is fastest).
static int usable_cpus()
{
    static int cached = 0;
    if (cached) return cached;
    int n = 1;
#ifdef _OPENMP
    n = omp_get_max_threads();
#endif
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && a < n) n = a;
    }
    auto quota_from = [](const char *path, bool v2) -> double {
        FILE *f = std::fopen(path, "r");
        if (!f) return 0.0;
        char buf[128] = {0};
        const size_t got = std::fread(buf, 1, sizeof(buf) - 1, f);
        std::fclose(f);
        if (!got) return 0.0;
        if (v2) { // "max 100000" or "<quota> <period>"
            double q = 0, p = 0;
            if (std::sscanf(buf, "%lf %lf", &q, &p) == 2 && q > 0 && p > 0) return q / p;
            return 0.0;
        }
        return std::atof(buf);
    };
    double q = quota_from("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0.0) {
        const double quota = quota_from("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
        const double period = quota_from("/sys/fs/cgroup/cpu/cpu.cfs_period_us", false);
        if (quota > 0 && period > 0) q = quota / period;
    }
    if (q > 0.0) {
        const int c = (int)(q + 0.999);
        if (c > 0 && c < n) n = c;
    }
    if (n > 64) n = 64; // the host stages are memory-bound sorts/scatters: no gain beyond this
    cached = n < 1 ? 1 : n;
    return cached;
}

struct ThreadScope {
#ifdef _OPENMP
    int old;
    explicit ThreadScope(uint64_t work)
    {
        old = omp_get_max_threads();
        const uint64_t cap = work / 65536 + 1;
        const int cpus = usable_cpus();
        omp_set_num_threads((uint64_t)cpus > cap ? (int)cap : cpus);
    }
    ~ThreadScope() { omp_set_num_threads(old); }
#else
    explicit ThreadScope(uint64_t) {}
#endif
};

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

namespace {
struct IdLess {
    bool operator()(const hb_u128 &a, const hb_u128 &b) const { return u128_less(a, b); }
};

// id -> sid lookup (open addressing over the sorted id array)
struct IdIndex {
    const hb_u128 *ids = nullptr;
    std::vector<uint32_t> slot; // sid + 1, 0 = empty
    uint64_t mask = 0;
    void build(const hb_u128 *a, uint64_t n)
    {
        ids = a;
        uint64_t cap = 16;
        while (cap < 2 * n + 2) cap <<= 1;
        slot.assign(cap, 0);
        mask = cap - 1;
        for (uint64_t i = 0; i < n; i++) {
            uint64_t h = mix64(a[i].lo ^ mix64(a[i].hi)) & mask;
            while (slot[h]) h = (h + 1) & mask;
            slot[h] = (uint32_t)i + 1;
        }
    }
    inline int64_t find(const hb_u128 &k) const
    {
        uint64_t h = mix64(k.lo ^ mix64(k.hi)) & mask;
        while (slot[h]) {
            uint32_t s = slot[h] - 1;
            if (u128_eq(ids[s], k)) return (int64_t)s;
            h = (h + 1) & mask;
        }
        return -1;
    }
};

struct KeyPos {
    uint64_t key; // (to_sid << 32) | from_sid
    uint64_t pos; // stream position
    bool operator<(const KeyPos &o) const { return key != o.key ? key < o.key : pos < o.pos; }
};
} // namespace

std::string ingest_edges(const hb_u128 *node_ids, uint64_t n_in, const hb_edge *edges, uint64_t m,
                         DenseGraph *out)
{
    out->ids.clear();
    out->row_ptr.clear();
    out->src.clear();
    out->m_input = m;
    out->m_unique = 0;
    if (m && !edges) return "edges == NULL with m > 0";
    ThreadScope threads(m + n_in);
    // ---- node set
    auto &ids = out->ids;
    try {
        if (node_ids && n_in) {
            ids.assign(node_ids, node_ids + n_in);
        } else {
            ids.resize(2 * m);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)m; i++) {
                ids[2 * i] = edges[i].from;
                ids[2 * i + 1] = edges[i].to;
            }
        }
        HB_SORT_CMP(ids.begin(), ids.end(), IdLess());
        ids.erase(std::unique(ids.begin(), ids.end(), u128_eq), ids.end());
    } catch (const std::bad_alloc &) {
        return "out of host memory building the node set";
    }
    const uint64_t n = ids.size();
    if (n >= 0xFFFFFFFFull - (1u << 20)) return "too many nodes (n must be < 2^32 - 2^20)";
    out->row_ptr.assign(n + 1, 0);
    if (n == 0 || m == 0) return "";
    // ---- map endpoints, order records by (to, from, stream position)
    std::vector<KeyPos> recs;
    try {
        IdIndex index;
        index.build(ids.data(), n);
        recs.resize(m);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)m; i++) {
            int64_t f = index.find(edges[i].from), t = index.find(edges[i].to);
            // harmonic.rs:135: `if let (Some, Some)` - records with an unknown endpoint are ignored
            recs[i].key = (f < 0 || t < 0) ? ~0ull : (((uint64_t)t << 32) | (uint64_t)f);
            recs[i].pos = (uint64_t)i;
        }
        HB_SORT(recs.begin(), recs.end());
    } catch (const std::bad_alloc &) {
        return "out of host memory sorting edge records";
    }
    // ---- first occurrence wins, then the flag filter
    std::vector<uint64_t> &row_ptr = out->row_ptr;
    std::vector<uint32_t> &src = out->src;
    uint64_t m_unique = 0;
    src.reserve(m);
    for (uint64_t i = 0; i < m;) {
        uint64_t key = recs[i].key;
        if (key == ~0ull) break; // unknown-endpoint records sort last
        uint64_t j = i + 1;
        while (j < m && recs[j].key == key) j++;
        m_unique++;
        if ((edges[recs[i].pos].rel_flags & HB_SKIPPED_REL_MASK) == 0) {
            src.push_back((uint32_t)key);
            row_ptr[(key >> 32) + 1]++;
        }
        i = j;
    }
    for (uint64_t v = 0; v < n; v++) row_ptr[v + 1] += row_ptr[v];
    out->m_unique = m_unique;
    return "";
}

// HB_FLAG_REFERENCE_TAIL: the page-level records update_changed_counters follows (harmonic.rs:82-92) -> keys
// (source device row << 32 | target device row): what ForwardlinksQuery::new(host id) YIELDS per segment (its
// LinksScorer de-duplicates neighbouring documents on to_id BEFORE anything looks at the flags), then the rel
// filter (:87) and the two counter lookups (:91-92); only records between two host nodes stay (8 bytes each).
struct TailIndex {
    IdIndex index;
};
TailIndex *tail_index_build(const hb_u128 *ids, uint64_t n)
{
    TailIndex *t = new (std::nothrow) TailIndex();
    if (!t) return nullptr;
    try {
        t->index.build(ids, n); // ids must stay alive and unchanged (hb_ctx::g.ids) for the lifetime of the index
    } catch (const std::bad_alloc &) {
        delete t;
        return nullptr;
    }
    return t;
}
void tail_index_free(TailIndex *t) { delete t; }

// One batch of page-level documents (doc order inside the current segment): the documents whose from_id is a host
// node id are the posting lists ForwardlinksQuery::new(host id) walks (query/forwardlink.rs:95-101); they are kept
// (from sid, to id, "passes the rel filter") until the segment ends.  Everything else can never be returned.
std::string tail_collect(const TailIndex *tix, uint64_t n, const hb_edge *recs, uint64_t count, std::vector<TailDoc> *open)
{
    if (n == 0 || count == 0) return "";
    if (!tix) return "out of host memory indexing the node ids";
    ThreadScope threads(count);
    try {
        const IdIndex &index = tix->index;
        std::vector<int64_t> from(count);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)count; i++) from[i] = index.find(recs[i].from);
        for (uint64_t i = 0; i < count; i++) {
            if (from[i] < 0) continue;
            TailDoc d;
            d.from_sid = (uint32_t)from[i];
            d.pass = (recs[i].rel_flags & HB_SKIPPED_REL_MASK) == 0 ? 1u : 0u;
            d.to = recs[i].to;
            open->push_back(d);
        }
    } catch (const std::bad_alloc &) {
        return "out of host memory collecting the tail records";
    }
    return "";
}

// LinksScorer over one posting list (query/raw/links.rs:115-232; one scorer per segment and term): `to` = the
// dedup column (ToId) of the list's documents in doc order, `self` = the queried node.  emit[i] = the scorer
// yields document i.  What it does, as written: self links are skipped; a document is skipped when its to_id
// equals the to_id of the LAST YIELDED document (adjacent de-duplication only - the store keeps documents sorted
// by sort_score, so equal targets are normally neighbours); and after every advance whole 128-document blocks
// are jumped over while the LAST document of the block has that same to_id (skip-list shortcut, :203-213),
// whatever lies in between.  The final partial block has no skip entry (tantivy postings/skip.rs:122-126,
// 276-282: last_doc_in_block = TERMINATED), so it is never jumped.
static void links_scorer_walk(const hb_u128 *to, uint64_t len, const hb_u128 &self, uint8_t *emit)
{
    constexpr uint64_t kBlock = 128; // COMPRESSION_BLOCK_SIZE
    const uint64_t full = len / kBlock * kBlock; // documents in full blocks
    uint64_t pos = 0;
    while (pos < len && u128_eq(to[pos], self)) pos++; // LinksScorer::new, :143-165
    if (pos >= len) return;
    hb_u128 last = to[pos];
    while (pos < len) {
        emit[pos] = 1;
        pos++; // postings.advance()
        while (pos < full && u128_eq(to[pos / kBlock * kBlock + kBlock - 1], last)) pos = pos / kBlock * kBlock + kBlock;
        while (pos < len && (u128_eq(to[pos], last) || u128_eq(to[pos], self))) pos++;
        if (pos < len) last = to[pos];
    }
}

// End of a segment: walk every host's posting list like LinksScorer does, then apply what harmonic.rs does to the
// query's result: the rel filter on the YIELDED document (:87) and the two counter lookups (:91-92).  Keys
// (source device row << 32 | target device row) are appended to *keys; *open is emptied.
std::string tail_close_segment(const TailIndex *tix, const hb_u128 *ids, const uint32_t *dev_of, std::vector<TailDoc> *open,
                               std::vector<uint64_t> *keys)
{
    if (open->empty()) return "";
    if (!tix) return "out of host memory indexing the node ids";
    try {
        const IdIndex &index = tix->index;
        std::stable_sort(open->begin(), open->end(), [](const TailDoc &a, const TailDoc &b) { return a.from_sid < b.from_sid; });
        std::vector<hb_u128> to;
        std::vector<uint8_t> emit;
        for (size_t i = 0; i < open->size();) {
            size_t j = i;
            while (j < open->size() && (*open)[j].from_sid == (*open)[i].from_sid) j++;
            const uint32_t f = (*open)[i].from_sid;
            to.resize(j - i);
            for (size_t k = i; k < j; k++) to[k - i] = (*open)[k].to;
            emit.assign(j - i, 0);
            links_scorer_walk(to.data(), j - i, ids[f], emit.data());
            for (size_t k = i; k < j; k++) {
                if (!emit[k - i] || !(*open)[k].pass) continue;
                const int64_t t = index.find((*open)[k].to);
                if (t >= 0) keys->push_back(((uint64_t)dev_of[f] << 32) | (uint64_t)dev_of[t]);
            }
            i = j;
        }
        std::vector<TailDoc>().swap(*open);
    } catch (const std::bad_alloc &) {
        return "out of host memory mapping the tail records";
    }
    return "";
}
// all keys -> CSR by source device row (duplicates dropped: max is idempotent)
std::string build_tail_csr(std::vector<uint64_t> *keys, uint64_t n_pad, std::vector<uint64_t> *ptr, std::vector<uint32_t> *to)
{
    ThreadScope threads(keys->size());
    try {
        ptr->assign(n_pad + 1, 0);
        HB_SORT(keys->begin(), keys->end());
        keys->erase(std::unique(keys->begin(), keys->end()), keys->end());
        to->resize(keys->size());
        for (uint64_t i = 0; i < keys->size(); i++) {
            (*to)[i] = (uint32_t)(*keys)[i];
            (*ptr)[((*keys)[i] >> 32) + 1]++;
        }
        for (uint64_t v = 0; v < n_pad; v++) (*ptr)[v + 1] += (*ptr)[v];
    } catch (const std::bad_alloc &) {
        return "out of host memory building the tail index";
    }
    return "";
}

// destination partition: rank `rank` keeps the in-edges of the rows with sid % world == rank only
void keep_owned_rows(DenseGraph *g, uint64_t world, uint64_t rank)
{
    const uint64_t n = g->ids.size();
    if (world <= 1 || n == 0) return;
    uint64_t w = 0;
    std::vector<uint64_t> rp(n + 1, 0);
    for (uint64_t v = 0; v < n; v++) {
        rp[v] = w;
        if (v % world != rank) continue;
        const uint64_t b = g->row_ptr[v], e = g->row_ptr[v + 1];
        if (w != b) std::memmove(g->src.data() + w, g->src.data() + b, (e - b) * sizeof(uint32_t));
        w += e - b;
    }
    rp[n] = w;
    g->src.resize(w);
    g->row_ptr.swap(rp);
}

std::string check_dense(const hb_u128 *ids, uint64_t n, const uint64_t *row_ptr, const uint32_t *src,
                        uint64_t m)
{
    if (n >= 0xFFFFFFFFull - (1u << 20)) return "too many nodes (n must be < 2^32 - 2^20)";
    if (n && (!ids || !row_ptr)) return "NULL ids/row_ptr";
    if (m && !src) return "NULL src";
    if (n == 0) return m ? "edges without nodes" : "";
    if (row_ptr[0] != 0 || row_ptr[n] != m) return "row_ptr[0] != 0 or row_ptr[n] != m_eff";
    ThreadScope threads(m + n);
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t v = 0; v < (int64_t)n; v++) {
        if (row_ptr[v + 1] < row_ptr[v]) bad |= 1;
        if (v > 0 && !u128_less(ids[v - 1], ids[v])) bad |= 2;
    }
    if (bad & 1) return "row_ptr not monotone";
    if (bad & 2) return "sorted_ids not strictly ascending";
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t e = 0; e < (int64_t)m; e++)
        if (src[e] >= n) bad |= 4;
    if (bad & 4) return "src index out of range";
    return "";
}

void count_out_degree(const uint64_t *row_ptr, const uint32_t *src, uint64_t n, std::vector<uint32_t> *deg)
{
    deg->assign(n, 0);
    const uint64_t m = n ? row_ptr[n] : 0;
    ThreadScope threads(m);
    uint32_t *d = deg->data();
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < (int64_t)m; e++) {
#pragma omp atomic
        d[src[e]]++;
    }
}

// ---------------------------------------------------------------------------------------
// Planner.
//
// Device order: nodes sorted by (global) out-degree, descending - the counters that are
// gathered most often become one contiguous, cache-resident prefix of the register array
// (per-XCD L2 4 MiB = 64 Ki counters, Infinity Cache 256 MiB = 4 Mi counters).
//
// Hub splitting: a row with more than `chunk` sources is replaced by a tree of "virtual
// rows" (partial maxima over <= chunk sources each); every level is a separate launch,
// so the pass stays a synchronous (Jacobi) pull with one writer per row.
// ---------------------------------------------------------------------------------------
std::string build_plan(uint64_t n, const uint64_t *row_ptr, const uint32_t *src,
                       const std::vector<uint32_t> &out_degree, bool reorder, const PlanTune &tune_in, Plan *p)
{
    PlanTune tune = tune_in;
    uint32_t chunk = tune.chunk ? tune.chunk : kDefaultChunk;
    if (chunk < 4) chunk = 4;
    if (chunk > 4096) chunk = 4096;
    if (tune.direct_max == 0 || tune.direct_max > chunk) tune.direct_max = chunk;
    if (tune.minc == 0) tune.minc = 8;
    const uint64_t world = tune.world > 1 ? tune.world : 1;
    ThreadScope threads((n ? row_ptr[n] : 0) + 4 * n);
    const bool timing = std::getenv("HB_PLAN_TIMING") != nullptr;
    double tmark = now_ms();
    auto lap = [&](const char *what) {
        if (!timing) return;
        double t = now_ms();
        std::fprintf(stderr, "[plan] %-28s %8.1f ms\n", what, t - tmark);
        tmark = t;
    };
    p->n = n;
    // one contiguous slice of rows per owner (destination partition: owner(sid) = sid % world),
    // every slice padded to the same multiple of kRowAlign so that slices can be all-gathered
    const uint64_t slice = ((n + world - 1) / world + kRowAlign - 1) / kRowAlign * kRowAlign;
    p->slice = slice;
    p->n_pad = slice * world;
    p->chunk = chunk;
    p->m_eff = n ? row_ptr[n] : 0;
    p->order.assign(p->n_pad, kNone);
    p->dev_of.resize(n);
    p->level_begin.clear();
    const uint64_t n_pad = p->n_pad;
    // hotness rank of a device position: position j of every slice is equally hot
    auto hot_rank = [&](uint32_t idx) -> uint32_t { return (uint32_t)(((uint64_t)idx % slice) * world + (uint64_t)idx / slice); };
    auto from_hot = [&](uint32_t hr) -> uint32_t { return (uint32_t)(((uint64_t)hr % world) * slice + (uint64_t)hr / world); };
    try {
        // ---- device order: inside each owner's slice by descending (global) out-degree
        if (reorder && n) {
            std::vector<std::pair<uint64_t, uint32_t>> kv(n);
#pragma omp parallel for schedule(static)
            for (int64_t s = 0; s < (int64_t)n; s++)
                kv[s] = {(((uint64_t)s % world) << 32) | (uint64_t)(0xFFFFFFFFu - out_degree[s]), (uint32_t)s};
            HB_SORT(kv.begin(), kv.end());
            uint64_t pos = 0, cur_owner = 0;
            for (uint64_t i = 0; i < n; i++) {
                const uint64_t owner = kv[i].first >> 32;
                if (owner != cur_owner) {
                    cur_owner = owner;
                    pos = owner * slice;
                }
                p->order[pos++] = kv[i].second;
            }
        } else {
            std::vector<uint64_t> fill(world, 0);
            for (uint64_t s = 0; s < n; s++) {
                const uint64_t owner = s % world;
                p->order[owner * slice + fill[owner]++] = (uint32_t)s;
            }
        }
#pragma omp parallel for schedule(static)
        for (int64_t d = 0; d < (int64_t)n_pad; d++)
            if (p->order[d] != kNone) p->dev_of[p->order[d]] = (uint32_t)d;

        lap("device order");
        // ---- rows in device order; sources relabelled to hotness ranks and sorted (hottest first)
        std::vector<uint64_t> rp(n_pad + 1, 0);
        for (uint64_t d = 0; d < n_pad; d++) {
            uint32_t s = p->order[d];
            rp[d + 1] = rp[d] + (s == kNone ? 0 : (row_ptr[s + 1] - row_ptr[s]));
        }
        uvec<uint32_t> rs(p->m_eff);
        // small dynamic chunks: in device order the biggest hubs sit next to each other at the front
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t d = 0; d < (int64_t)n_pad; d++) {
            uint32_t s = p->order[d];
            if (s == kNone) continue;
            uint64_t b = row_ptr[s], e = row_ptr[s + 1], o = rp[d];
            for (uint64_t k = b; k < e; k++) rs[o + (k - b)] = hot_rank(p->dev_of[src[k]]);
            std::sort(rs.begin() + o, rs.begin() + o + (e - b));
        }
        lap("relabel + sort rows");
        // from here on `rs` holds hotness ranks; they are mapped back to device positions when the
        // lists are emitted (from_hot), so band_of() below works on ranks

        // ---- hub splitting, level by level
        // Level 1: a hub row's (ascending = hottest-first) source list is cut into chunks of
        // <= chunk sources; a cut is also made where the list crosses a band boundary of the
        // source index space (band 0 = the `band_w` hottest counters = what one XCD's L2 holds,
        // then bands doubling in width), provided the chunk already has >= minc sources.  The
        // chunks of ALL rows are then ordered by band, so that workgroups running at the same
        // time gather from the same few MiB of the counter array (L2 hits instead of fabric
        // requests; measured ceilings: tools/gather_bench.hip).
        std::vector<uint64_t> vrow_ptr; // offsets of virtual rows' lists in vsrc
        uvec<uint32_t> vsrc;
        vrow_ptr.push_back(0);
        std::vector<uint8_t> is_split(n_pad, 0);
        uint64_t next_vid = p->n_pad;
        p->level_begin.push_back(next_vid);
        // key of a source = its slice of the hotness order: slice 0 = the band_w hottest counters (every
        // XCD's L2 holds them), slices 1..kWarmSlices = the following band_w-wide ranges, each gathered by
        // ONE XCD only (8 L2s cache 8 different slices instead of 8 copies of the same lines); beyond that
        // the cold tail in doubling bands (no reuse to protect, ordered only for tidiness).
        const uint32_t kWarmSlices = tune.xcd_map ? 64u : 0u;
        auto band_of = [&](uint32_t idx) -> uint32_t {
            if (!tune.band_w || idx < tune.band_w) return 0;
            const uint64_t j = (uint64_t)idx / tune.band_w;
            if (j <= kWarmSlices) return (uint32_t)j;
            const uint64_t q = j / (kWarmSlices + 1u); // >= 1
            return kWarmSlices + 1u + (uint32_t)(63 - __builtin_clzll(q));
        };
        struct Chunk { uint64_t beg; uint32_t len; uint32_t key; };
        std::vector<Chunk> chunks;
        std::vector<uint32_t> hub_rows;          // split rows, ascending
        std::vector<uint64_t> hub_first;         // first chunk (creation order) of each split row, +1 sentinel
        const uint32_t minc = std::max<uint32_t>(1, std::min(tune.minc, chunk));
        for (uint64_t d = 0; d < n_pad; d++) {
            if (rp[d + 1] - rp[d] <= tune.direct_max) continue;
            is_split[d] = 1;
            hub_rows.push_back((uint32_t)d);
        }
        // the greedy cut of one row; emit(begin, length, key) per chunk.  Rows are independent: count in
        // parallel, prefix, then fill in parallel.
        auto cut_row = [&](uint64_t d, auto &&emit) {
            const uint64_t b = rp[d], e = rp[d + 1];
            uint64_t i = b;
            while (i < e) {
                const uint32_t b0 = band_of(rs[i]);
                uint64_t j = i + 1;
                while (j < e && j - i < chunk) {
                    if (j - i >= minc && band_of(rs[j]) != b0) break;
                    j++;
                }
                // do not leave a tiny remainder behind: absorb it when it fits
                if (e - j < minc && e - i <= chunk) j = e;
                emit(i, (uint32_t)(j - i), b0);
                i = j;
            }
        };
        hub_first.assign(hub_rows.size() + 1, 0);
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t h = 0; h < (int64_t)hub_rows.size(); h++) {
            uint64_t cnt = 0;
            cut_row(hub_rows[h], [&](uint64_t, uint32_t, uint32_t) { cnt++; });
            hub_first[h + 1] = cnt;
        }
        for (size_t h = 0; h < hub_rows.size(); h++) hub_first[h + 1] += hub_first[h];
        chunks.resize(hub_first[hub_rows.size()]);
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t h = 0; h < (int64_t)hub_rows.size(); h++) {
            uint64_t w = hub_first[h];
            cut_row(hub_rows[h], [&](uint64_t beg, uint32_t len, uint32_t key) { chunks[w++] = {beg, len, key}; });
        }
        lap("cut chunks");
        // XCD groups: a warm-slice chunk belongs to XCD (slice mod 8); hot (slice 0) and cold chunks fill the
        // groups up to equal work.  Inside a group: slice ascending, longer chunks first (wave-uniform
        // trip counts).  Without xcd_map there is one group.
        const int groups = tune.xcd_map ? 8 : 1;
        std::vector<uint8_t> grp(chunks.size(), 0);
        std::vector<uint32_t> corder(chunks.size());
        std::iota(corder.begin(), corder.end(), 0u);
        HB_STABLE_SORT_CMP(corder.begin(), corder.end(), [&](uint32_t a, uint32_t b2) {
            if (chunks[a].key != chunks[b2].key) return chunks[a].key < chunks[b2].key;
            return chunks[a].len > chunks[b2].len;
        });
        if (groups > 1) {
            // warm-slice chunks are pinned (slice mod 8); the flexible ones - hot (slice 0: every L2 holds it) and
            // cold (no reuse to protect) - level the groups: group x gets the share Q_x / sum(Q) of EACH flexible
            // class, Q_x = what it lacks to the mean load, as one contiguous range of the class in (slice, longer
            // first) order.  Prefix sums and eight thresholds per class: the same rule runs on the device
            // (hb_plan.hip) with scans instead of this loop.
            XcdQuota quota;
            for (size_t k = 0; k < chunks.size(); k++) quota.add(chunks[k].key, chunks[k].len, kWarmSlices);
            quota.finish();
            uint64_t prefix[2] = {0, 0}; // running load of the hot / cold class in corder order
            for (size_t r = 0; r < corder.size(); r++) {
                const uint32_t k = corder[r];
                const uint32_t key = chunks[k].key;
                if (key >= 1 && key <= kWarmSlices) {
                    grp[k] = (uint8_t)(key & 7u);
                } else {
                    const int cls = key == 0 ? 0 : 1;
                    grp[k] = (uint8_t)quota.group_of(cls, prefix[cls]);
                    prefix[cls] += chunks[k].len + 4; // +4: per-row overhead in gather units
                }
            }
            HB_STABLE_SORT_CMP(corder.begin(), corder.end(), [&](uint32_t a, uint32_t b2) { return grp[a] < grp[b2]; });
        }
        lap("sort chunks");
        std::vector<uint32_t> vid_of(chunks.size());
        {
            // row ids: group after group, every group padded to whole 64-row tiles; then a parallel fill
            std::vector<uint64_t> vrp;
            vrp.reserve(chunks.size() + 8 * kRowAlign + 1);
            std::vector<int64_t> row_chunk; // chunk of each emitted row, -1 = padding row
            row_chunk.reserve(chunks.size() + 8 * kRowAlign);
            uint64_t off = 0;
            size_t k = 0;
            for (int x = 0; x < groups; x++) {
                p->xcd_begin[x] = next_vid + row_chunk.size();
                while (k < corder.size() && grp[corder[k]] == x) {
                    vrp.push_back(off);
                    off += chunks[corder[k]].len;
                    vid_of[corder[k]] = (uint32_t)(next_vid + row_chunk.size());
                    row_chunk.push_back((int64_t)corder[k]);
                    k++;
                }
                while (groups > 1 && row_chunk.size() % kRowAlign) {
                    vrp.push_back(off);
                    row_chunk.push_back(-1);
                }
            }
            for (int x = groups; x <= 8; x++) p->xcd_begin[x] = next_vid + row_chunk.size();
            p->xcd_groups = groups;
            vrp.push_back(off);
            vrow_ptr.swap(vrp);
            // capacity for the upper levels too (each holds at most 1/chunk of the rows below it, plus
            // padding), so that appending them never reallocates - and copies - the level-1 lists
            vsrc.reserve(off + row_chunk.size() + row_chunk.size() / 8 + 64ull * kRowAlign);
            vsrc.resize(off);
            p->level1_edges = off;
            p->level1_rows = chunks.size();
#pragma omp parallel for schedule(static, 4096)
            for (int64_t r = 0; r < (int64_t)row_chunk.size(); r++) {
                if (row_chunk[r] < 0) continue;
                const Chunk &c = chunks[row_chunk[r]];
                uint32_t *dst = vsrc.data() + vrow_ptr[r];
                for (uint64_t i = 0; i < c.len; i++) dst[i] = from_hot(rs[c.beg + i]);
            }
            next_vid += row_chunk.size();
        }
        lap("emit level-1 lists");
        // per split row: the list of virtual ids it currently reads (flat, CSR-like)
        std::vector<uint64_t> lptr(hub_rows.size() + 1, 0);
        std::vector<uint32_t> lids(chunks.size());
        for (size_t h = 0; h < hub_rows.size(); h++) {
            lptr[h] = hub_first[h];
            for (uint64_t k = hub_first[h]; k < hub_first[h + 1]; k++) lids[k] = vid_of[k];
        }
        lptr[hub_rows.size()] = chunks.size();
        auto pad_level = [&]() {
            while ((next_vid - p->n_pad) % kRowAlign) {
                vrow_ptr.push_back(vsrc.size());
                next_vid++;
            }
        };
        // upper levels: while some row still reads more than `chunk` virtual rows, group them
        while (true) {
            pad_level();
            p->level_begin.push_back(next_vid);
            bool any = false;
            for (size_t h = 0; h < hub_rows.size() && !any; h++) any = (lptr[h + 1] - lptr[h]) > chunk;
            if (!any) break;
            std::vector<uint64_t> nptr(hub_rows.size() + 1, 0);
            std::vector<uint32_t> nids;
            nids.reserve(lids.size());
            for (size_t h = 0; h < hub_rows.size(); h++) {
                nptr[h] = nids.size();
                const uint64_t cnt = lptr[h + 1] - lptr[h];
                if (cnt <= chunk) {
                    nids.insert(nids.end(), lids.begin() + lptr[h], lids.begin() + lptr[h + 1]);
                    continue;
                }
                const uint64_t parts = (cnt + chunk - 1) / chunk;
                const uint64_t per = (cnt + parts - 1) / parts;
                for (uint64_t k = 0; k < parts; k++) {
                    const uint64_t b = lptr[h] + k * per, e = std::min<uint64_t>(lptr[h + 1], b + per);
                    vsrc.insert(vsrc.end(), lids.begin() + b, lids.begin() + e);
                    vrow_ptr.push_back(vsrc.size());
                    nids.push_back((uint32_t)next_vid++);
                }
            }
            nptr[hub_rows.size()] = nids.size();
            lptr.swap(nptr);
            lids.swap(nids);
        }
        if (p->level_begin.size() >= 2 && p->level_begin[p->level_begin.size() - 1] ==
                                              p->level_begin[p->level_begin.size() - 2])
            p->level_begin.pop_back(); // no trailing empty level
        p->nv = next_vid - p->n_pad;
        if (next_vid >= (uint64_t)kNone) return "row id space exhausted (n + virtual rows >= 2^32 - 1)";
        // hub index of a split row (for the assembly below)
        std::vector<uint32_t> hub_index(n_pad, 0);
        for (size_t h = 0; h < hub_rows.size(); h++) hub_index[hub_rows[h]] = (uint32_t)h;

        lap("upper levels");
        // ---- assemble: real rows [0, n_pad), then virtual rows
        const uint64_t rows_total = p->n_pad + p->nv;
        p->row_ptr.resize(rows_total + 1);
        uint64_t total = 0, direct = 0, with_in = 0;
        for (uint64_t d = 0; d < n_pad; d++) {
            p->row_ptr[d] = total;
            total += is_split[d] ? (lptr[hub_index[d] + 1] - lptr[hub_index[d]]) : (rp[d + 1] - rp[d]);
            if (!is_split[d]) direct += rp[d + 1] - rp[d];
            with_in += rp[d + 1] > rp[d];
        }
        p->direct_edges = direct;
        p->rows_with_in_edges = with_in;
        p->row_ptr[n_pad] = total;
        const uint64_t real_total = total;
        for (uint64_t k = 0; k < p->nv; k++) p->row_ptr[p->n_pad + k + 1] = real_total + vrow_ptr[k + 1];
        p->src.resize(real_total + vsrc.size());
#pragma omp parallel for schedule(dynamic, 4096)
        for (int64_t d = 0; d < (int64_t)n_pad; d++) {
            uint64_t o = p->row_ptr[d];
            if (is_split[d]) {
                const uint64_t b = lptr[hub_index[d]], e = lptr[hub_index[d] + 1];
                std::memcpy(p->src.data() + o, lids.data() + b, (e - b) * sizeof(uint32_t));
            } else {
                for (uint64_t k = rp[d]; k < rp[d + 1]; k++) p->src[o + (k - rp[d])] = from_hot(rs[k]);
            }
        }
        lap("assemble");
        {
            const int64_t blocks = (int64_t)((vsrc.size() + (1u << 20) - 1) >> 20);
#pragma omp parallel for schedule(static)
            for (int64_t b = 0; b < blocks; b++) {
                const uint64_t o = (uint64_t)b << 20, cnt = std::min<uint64_t>(1u << 20, vsrc.size() - o);
                std::memcpy(p->src.data() + real_total + o, vsrc.data() + o, cnt * sizeof(uint32_t));
            }
        }
    } catch (const std::bad_alloc &) {
        return "out of host memory in the planner";
    }
    return "";
}

bool build_lc_table(uint8_t lc[68])
{
    // linear_counting(v) = m * (m / v).ln() with m = 64 (hyperloglog.rs:4472-4476); used
    // only when v != 0 and the value is <= threshold(6) = 40 (:4505-4515), and then only
    // its truncation `as usize` matters.
    bool robust = true;
    lc[0] = 0xFF;
    lc[65] = lc[66] = lc[67] = 0xFF;
    lc[64] = 0; // 64 ln(1) = 0 exactly in every libm (an all-zero counter; padding rows only)
    for (int v = 1; v < 64; v++) {
        double h = 64.0 * std::log(64.0 / (double)v);
        if (std::fabs(h - 40.0) < 1e-6) robust = false;
        if (h <= 40.0) {
            double fl = std::floor(h);
            if (h - fl < 1e-6 || (fl + 1.0) - h < 1e-6) robust = false;
            lc[v] = (uint8_t)fl;
        } else {
            lc[v] = 0xFF;
        }
    }
    return robust;
}

// ---- host-parallel loops of the result path (hb_internal.h) ------------------------------------------------------------------------
void host_scatter_f64(double *out, const uint32_t *idx, const double *val, uint64_t n)
{
    if (n < (1u << 16)) {
        for (uint64_t k = 0; k < n; k++) out[idx[k]] = val[k];
        return;
    }
#pragma omp parallel for num_threads(std::min(host_threads(), 8)) schedule(static)
    for (int64_t k = 0; k < (int64_t)n; k++) out[idx[k]] = val[k];
}

void host_gather_id_lo(const hb_u128 *ids, uint64_t n, uint64_t *lo)
{
#pragma omp parallel for num_threads(n >= (1u << 18) ? std::min(host_threads(), 16) : 1) schedule(static)
    for (int64_t s = 0; s < (int64_t)n; s++) lo[s] = ids[s].lo;
}

void host_compact_results(const double *src, const hb_u128 *idsrc, uint64_t n, hb_u128 *ids, double *vals, uint64_t cap, const uint64_t *in_bits)
{
    const int want = n >= (1u << 18) ? std::min(host_threads(), 16) : 1;
    std::vector<uint64_t> first((size_t)want + 1, 0), cfirst((size_t)want + 1, 0);
    int team = 1;
#pragma omp parallel num_threads(want)
    {
        // (the team may be smaller than asked for: the shares are cut by the team that exists)
        const int nt = omp_get_num_threads(), t = omp_get_thread_num();
#pragma omp single
        team = nt;
        // shares are cut at multiples of 64 sids, so that a share owns whole words of the bitmap
        const uint64_t nw = (n + 63) / 64;
        const uint64_t lo = std::min(n, nw * (uint64_t)t / (uint64_t)nt * 64), hi = std::min(n, nw * (uint64_t)(t + 1) / (uint64_t)nt * 64);
        if (in_bits) { // where this share's entries begin in the compact image
            uint64_t bits = 0;
            for (uint64_t w = lo / 64; w < (hi + 63) / 64; w++) bits += (uint64_t)__builtin_popcountll(in_bits[w]);
            cfirst[(size_t)t + 1] = bits;
#pragma omp barrier
#pragma omp single
            for (int k = 0; k < nt; k++) cfirst[(size_t)k + 1] += cfirst[(size_t)k];
        }
        uint64_t kept = 0;
        if (in_bits) {
            for (uint64_t c = cfirst[(size_t)t]; c < cfirst[(size_t)t + 1]; c++) kept += src[c] >= 0.0;
        } else {
            for (uint64_t sid = lo; sid < hi; sid++) kept += src[sid] >= 0.0;
        }
        first[(size_t)t + 1] = kept;
#pragma omp barrier
#pragma omp single
        for (int k = 0; k < nt; k++) first[(size_t)k + 1] += first[(size_t)k];
        uint64_t at = first[(size_t)t];
        if (in_bits) {
            uint64_t c = cfirst[(size_t)t];
            for (uint64_t w = lo / 64; w < (hi + 63) / 64 && at < cap; w++) {
                uint64_t m = in_bits[w];
                while (m && at < cap) {
                    const uint64_t sid = (w << 6) + (uint64_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const double v = src[c++];
                    if (v < 0.0) continue;
                    if (ids) ids[at] = idsrc[sid];
                    if (vals) vals[at] = v;
                    at++;
                }
            }
        } else {
            for (uint64_t sid = lo; sid < hi && at < cap; sid++) {
                const double v = src[sid];
                if (v < 0.0) continue;
                if (ids) ids[at] = idsrc[sid];
                if (vals) vals[at] = v;
                at++;
            }
        }
    }
    (void)team;
}

} // namespace hb
