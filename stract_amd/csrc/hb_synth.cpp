// hb_synth.cpp - synthetic webgraph generator (bench / test INPUT support, host only).
//
// Not part of the drop-in boundary and not on the hot path: it manufactures the
// power-law host graphs BASELINE.json's configs name (R-MAT, Graph500 parameters
// a,b,c,d = 0.57,0.19,0.19,0.05) in the two forms the library
// ingests: (1) the reduced dense form hb_load_dense() takes (ascending NodeIDs +
// CSR by destination), (2) raw SmallEdge records (reference: crates/core/src/webgraph/
// edge.rs:31-35) for hb_load_edges(), optionally "salted" with skipped rel-flags,
// duplicates with different flags, and self-loops so the ingest semantics
// (store.rs:297-357, harmonic.rs:131) are exercised.
//
// Determinism: every raw edge k is a pure function of (seed, k) (counter-based
// splitmix64), so any thread count gives the same graph.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>
#define HBS_SORT(b, e) __gnu_parallel::sort((b), (e))
#else
#define HBS_SORT(b, e) std::sort((b), (e))
#endif

extern "C" {
typedef struct { uint64_t lo, hi; } hbs_u128;
typedef struct { hbs_u128 from, to; uint64_t rel_flags; } hbs_edge; // == hb_edge (include/hyperball.h)
}

static inline bool operator<(const hbs_u128 &a, const hbs_u128 &b)
{
    return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo;
}

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// NodeID of synthetic vertex i: looks like a 128-bit hash, id order != index order
// (SURVEY.md §8(d)).
static inline hbs_u128 node_id_of(uint64_t i)
{
    hbs_u128 r;
    r.hi = splitmix64(2 * i + 1);
    r.lo = splitmix64(2 * i + 2);
    return r;
}

struct hbs_graph {
    int scale = 0;
    uint64_t seed = 0;
    uint64_t raw_drawn = 0;
    std::vector<uint64_t> edges;   // unique non-self edges, key = (dst_dense << 32) | src_dense, sorted
    std::vector<hbs_u128> ids;     // ascending NodeID of dense node r
    std::vector<uint32_t> vertex;  // synthetic vertex index of dense node r
    std::vector<uint64_t> row_ptr; // n + 1
    std::vector<uint32_t> src;     // m
};

// One R-MAT edge: `scale` quadrant draws, 16 random bits per level.
// Thresholds: a = 0.57, a+b = 0.76, a+b+c = 0.95 of 65536.
static inline void rmat_edge(int scale, uint64_t seed, uint64_t k, uint32_t *from, uint32_t *to)
{
    uint64_t f = 0, t = 0;
    uint64_t ctr = seed ^ (k * 0xD1342543DE82EF95ull);
    uint64_t bits = 0;
    for (int l = 0; l < scale; l++) {
        if ((l & 3) == 0) bits = splitmix64(ctr + (uint64_t)(l >> 2));
        uint32_t r = (uint32_t)(bits & 0xFFFF);
        bits >>= 16;
        uint32_t fb, tb;
        if (r < 37356u) { fb = 0; tb = 0; }        // a
        else if (r < 49807u) { fb = 0; tb = 1; }   // b
        else if (r < 62259u) { fb = 1; tb = 0; }   // c
        else { fb = 1; tb = 1; }                   // d
        f = (f << 1) | fb;
        t = (t << 1) | tb;
    }
    *from = (uint32_t)f;
    *to = (uint32_t)t;
}

extern "C" {

void hbs_free(hbs_graph *g) { delete g; }

// Draw raw R-MAT edges until at least m_target unique non-self edges exist, then keep
// exactly the m_target smallest (in (to,from) vertex-key order is NOT stream order, so
// instead we keep all uniques of the drawn prefix; the measured m is reported).
// Long-tail variant (bench input for the convergence tail): the R-MAT core plus a levelled DAG hanging off it.
// tail_permille * n_core / 1000 extra hosts are laid out in levels of geometrically shrinking size
// (level d has ratio_permille/1000 of the hosts of level d-1); every host of level d gets `fanin` in-links from
// random hosts of level d-1 (level 0 = random core hosts).  Information only flows core -> tail, so the core
// converges as before and the tail keeps changing for one more pass per level: T grows by the tail depth
// (tens of passes, like real host graphs) while the changed set shrinks geometrically - first bitmap/frontier
// passes (A_t a few % of the edges), then many worklist passes.  tail_permille = 0: plain R-MAT.
static hbs_graph *build_graph(int scale, uint64_t m_target, uint64_t seed, int threads, uint32_t tail_permille,
                              uint32_t ratio_permille, uint32_t fanin);

hbs_graph *hbs_rmat(int scale, uint64_t m_target, uint64_t seed, int threads)
{
    return build_graph(scale, m_target, seed, threads, 0, 0, 0);
}

hbs_graph *hbs_rmat_tail(int scale, uint64_t m_target, uint64_t seed, int threads, uint32_t tail_permille,
                         uint32_t ratio_permille, uint32_t fanin)
{
    if (scale > 30 || ratio_permille >= 1000 || (tail_permille && !fanin)) return nullptr;
    return build_graph(scale, m_target, seed, threads, tail_permille, ratio_permille, fanin);
}

} // extern "C"

static hbs_graph *build_graph(int scale, uint64_t m_target, uint64_t seed, int threads, uint32_t tail_permille,
                              uint32_t ratio_permille, uint32_t fanin)
{
    if (scale < 1 || scale > 31) return nullptr;
#ifdef _OPENMP
    {   // small graphs: a few threads beat every hardware thread of a many-core box
        // all hardware threads unless told otherwise: NOT omp_get_max_threads(), which another library of the process
        // (the oracle's omp_set_num_threads) may have lowered - the random-access phases scale with threads
        int nt = threads > 0 ? threads : omp_get_num_procs();
        const uint64_t cap = m_target / 65536 + 1;
        if ((uint64_t)nt > cap) nt = (int)cap;
        omp_set_num_threads(nt < 1 ? 1 : nt);
    }
#else
    (void)threads;
#endif
    hbs_graph *g = new hbs_graph();
    g->scale = scale;
    g->seed = seed;
    std::vector<uint64_t> keys; // (to << 32) | from, vertex ids
    uint64_t drawn = 0;
    uint64_t want = m_target + m_target / 16 + 64;
    for (int round = 0; round < 64; round++) {
        size_t base = keys.size();
        keys.resize(base + want);
        uint64_t *kp = keys.data() + base;
        const uint64_t d0 = drawn;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)want; i++) {
            uint32_t f, t;
            rmat_edge(scale, seed, d0 + (uint64_t)i, &f, &t);
            kp[i] = (f == t) ? ~0ull : (((uint64_t)t << 32) | f); // self-loops dropped
        }
        drawn += want;
        HBS_SORT(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        if (!keys.empty() && keys.back() == ~0ull) keys.pop_back();
        if (keys.size() >= m_target) break;
        uint64_t missing = m_target - keys.size();
        want = missing + missing / 4 + 64;
    }
    g->raw_drawn = drawn;
    uint64_t m = keys.size();
    // touched vertices
    uint64_t space = 1ull << scale;
    std::vector<uint8_t> touched(space, 0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        touched[(uint32_t)keys[i]] = 1;
        touched[(uint32_t)(keys[i] >> 32)] = 1;
    }
    if (tail_permille) {
        std::vector<uint32_t> core;
        for (uint64_t v = 0; v < space; v++)
            if (touched[v]) core.push_back((uint32_t)v);
        const uint64_t n_core = core.size();
        const uint64_t n_tail = n_core * tail_permille / 1000;
        if (n_core && n_tail && space + n_tail < 0xFFFFFFF0ull) {
            // level sizes: L_1 = n_tail * (1 - r), L_d = L_{d-1} * r
            std::vector<uint64_t> level_begin; // tail vertex index (0-based) where each level starts
            uint64_t sz = std::max<uint64_t>(1, n_tail * (1000 - ratio_permille) / 1000), used = 0;
            while (sz && used < n_tail) {
                level_begin.push_back(used);
                used += std::min(sz, n_tail - used);
                sz = sz * ratio_permille / 1000;
            }
            level_begin.push_back(used);
            const uint64_t tail_nodes = used;
            const size_t base = keys.size();
            keys.resize(base + tail_nodes * fanin);
            for (size_t d = 0; d + 1 < level_begin.size(); d++) {
                const uint64_t lb = level_begin[d], le = level_begin[d + 1];
                const uint64_t pb = d ? level_begin[d - 1] : 0, pcount = d ? lb - pb : n_core;
#pragma omp parallel for schedule(static)
                for (int64_t j = (int64_t)lb; j < (int64_t)le; j++) {
                    for (uint32_t e = 0; e < fanin; e++) {
                        const uint64_t h = splitmix64(seed ^ 0x7A11ull ^ ((uint64_t)j * 64 + e) * 0x9E3779B97F4A7C15ull);
                        const uint64_t pick = h % pcount;
                        const uint32_t from = d ? (uint32_t)(space + pb + pick) : core[pick];
                        keys[base + (uint64_t)j * fanin + e] = ((uint64_t)(uint32_t)(space + (uint64_t)j) << 32) | from;
                    }
                }
            }
            HBS_SORT(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            m = keys.size();
            space += tail_nodes;
            touched.assign(space, 0);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)m; i++) {
                touched[(uint32_t)keys[i]] = 1;
                touched[(uint32_t)(keys[i] >> 32)] = 1;
            }
        }
    }
    struct IdV {
        hbs_u128 id;
        uint32_t v;
        bool operator<(const IdV &o) const { return id < o.id; }
    };
    std::vector<IdV> nodes;
    for (uint64_t v = 0; v < space; v++)
        if (touched[v]) nodes.push_back({node_id_of(v), (uint32_t)v});
    HBS_SORT(nodes.begin(), nodes.end()) ;
    const uint64_t n = nodes.size();
    g->ids.resize(n);
    g->vertex.resize(n);
    std::vector<uint32_t> dense_of(space, 0xFFFFFFFFu);
    for (uint64_t r = 0; r < n; r++) {
        g->ids[r] = nodes[r].id;
        g->vertex[r] = nodes[r].v;
        dense_of[nodes[r].v] = (uint32_t)r;
    }
    std::vector<IdV>().swap(nodes);
    // remap to dense indices and sort by (dst, src)
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        uint32_t f = dense_of[(uint32_t)keys[i]], t = dense_of[(uint32_t)(keys[i] >> 32)];
        keys[i] = ((uint64_t)t << 32) | f;
    }
    HBS_SORT(keys.begin(), keys.end());
    g->row_ptr.assign(n + 1, 0);
    g->src.resize(m);
    for (uint64_t i = 0; i < m; i++) g->row_ptr[(keys[i] >> 32) + 1]++;
    for (uint64_t v = 0; v < n; v++) g->row_ptr[v + 1] += g->row_ptr[v];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)m; i++) g->src[i] = (uint32_t)keys[i];
    g->edges.swap(keys);
    return g;
}

extern "C" {

uint64_t hbs_num_nodes(const hbs_graph *g) { return g->ids.size(); }
uint64_t hbs_num_edges(const hbs_graph *g) { return g->src.size(); }
uint64_t hbs_raw_drawn(const hbs_graph *g) { return g->raw_drawn; }
const hbs_u128 *hbs_ids(const hbs_graph *g) { return g->ids.data(); }
const uint64_t *hbs_row_ptr(const hbs_graph *g) { return g->row_ptr.data(); }
const uint32_t *hbs_src(const hbs_graph *g) { return g->src.data(); }

// Raw SmallEdge export of the same graph in a seeded pseudo-random stream order.
// salt = 0: exactly the m unique edges, rel_flags = 0.
// salt = 1 (parity runs): additionally
//   * ~10% extra edges (not in the clean graph) whose first occurrence carries one
//     SKIPPED_REL bit, half of them followed later by a clean copy that must stay lost
//     (dedup precedes the flag filter, store.rs:313 then harmonic.rs:131);
//   * ~1% duplicates of clean edges, with a skipped flag, placed AFTER the clean copy
//     (must be ignored);
//   * ~0.1% self-loops (kept, no-ops);
//   * clean edges get harmless flag bits (bits outside 0x6FED00) now and then.
// The reduced graph (after the reference's semantics) is NOT the clean graph: its node set
// is larger (flagged-only edges introduce endpoints that appear in no surviving edge) and
// a few extra edges survive; parity on salted input is checked against the oracle's
// structure-faithful path run on the same records.  Returns the number of records written, or
// the required capacity if out == NULL.
uint64_t hbs_export_edges(const hbs_graph *g, hbs_edge *out, uint64_t cap, int salt, uint64_t salt_seed)
{
    const uint64_t m = g->edges.size();
    const uint64_t n = g->ids.size();
    const uint64_t mask = 0x6FED00ull;
    static const int skipped_bits[12] = {8, 10, 11, 13, 14, 15, 16, 17, 18, 19, 21, 22};
    static const int harmless_bits[11] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 20};
    struct Rec { uint64_t order; uint32_t from, to; uint64_t flags; uint8_t synthetic_from, synthetic_to; };
    std::vector<Rec> recs;
    recs.reserve(m + (salt ? m / 8 + 16 : 0));
    for (uint64_t i = 0; i < m; i++) {
        uint64_t h = splitmix64(salt_seed ^ (i * 0x9E3779B97F4A7C15ull));
        Rec r;
        r.order = (h >> 2) | 1;      // odd, < 2^62
        r.from = (uint32_t)g->edges[i];
        r.to = (uint32_t)(g->edges[i] >> 32);
        r.flags = 0;
        r.synthetic_from = r.synthetic_to = 0;
        if (salt) {
            uint64_t h2 = splitmix64(h);
            if ((h2 & 7) == 0) r.flags = 1ull << harmless_bits[(h2 >> 8) % 11];
            if ((h2 >> 16) % 100 == 0) { // later flagged duplicate: ignored
                Rec d = r;
                d.order = r.order + ((splitmix64(h2) >> 3) % (0x3FFFFFFFFFFFFFFFull - r.order)) + 1;
                d.flags = 1ull << skipped_bits[(h2 >> 32) % 12];
                recs.push_back(d);
            }
        }
        recs.push_back(r);
    }
    if (salt && n >= 2) {
        // a set of the clean edges for membership tests
        const std::vector<uint64_t> &E = g->edges; // sorted
        uint64_t extra = m / 10 + 4;
        for (uint64_t j = 0; j < extra; j++) {
            uint64_t h = splitmix64(salt_seed + 0xABCDEF + j * 0x2545F4914F6CDD1Dull);
            Rec r;
            r.synthetic_from = r.synthetic_to = 0;
            r.from = (uint32_t)(h % n);
            r.to = (uint32_t)((h >> 32) % n);
            if (j % 97 == 0) r.to = r.from; // self-loop (flag 0: kept, no-op)
            else if (j % 89 == 0) {          // endpoint that exists only through a flagged edge
                r.synthetic_to = 1;
                r.to = (uint32_t)j;
            }
            uint64_t key = ((uint64_t)r.to << 32) | r.from;
            bool self = (r.to == r.from) && !r.synthetic_to;
            if (!self && !r.synthetic_to && std::binary_search(E.begin(), E.end(), key)) continue;
            r.order = (splitmix64(h) >> 2) | 1;
            r.flags = self ? 0 : (1ull << skipped_bits[(h >> 20) % 12]) | ((h & 1) ? 1ull << harmless_bits[(h >> 8) % 11] : 0);
            recs.push_back(r);
            if (!self && (j & 1)) { // clean copy AFTER the flagged first occurrence: stays lost
                Rec c = r;
                c.flags = 0;
                c.order = r.order + ((splitmix64(h + 7) >> 3) % (0x3FFFFFFFFFFFFFFFull - r.order)) + 1;
                recs.push_back(c);
            }
        }
    }
    if (!out) return recs.size();
    if (recs.size() > cap) return ~0ull;
    std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) {
        if (a.order != b.order) return a.order < b.order;
        if (a.to != b.to) return a.to < b.to;
        if (a.from != b.from) return a.from < b.from;
        return a.flags < b.flags;
    });
    for (uint64_t i = 0; i < recs.size(); i++) {
        const Rec &r = recs[i];
        out[i].from = r.synthetic_from ? node_id_of((1ull << 40) + r.from) : g->ids[r.from];
        out[i].to = r.synthetic_to ? node_id_of((1ull << 40) + r.to) : g->ids[r.to];
        out[i].rel_flags = r.flags;
    }
    (void)mask;
    return recs.size();
}

} // extern "C"

// ---- streamed export (for graphs whose 40-byte records do not fit host memory at once) --------------------------
// Record p of the stream is a pure function of (graph, salt, p), so any slab [first, first + count) can be produced on
// its own, in parallel, and fed to hb_append_edges batch by batch.
//   salt = 0: the m clean edges in a pseudo-random order (affine permutation of the edge index), rel_flags = 0.
//   salt = 2 ("reduces to the clean graph"): the same, with
//     * harmless flag bits (outside SKIPPED_REL) on some clean records;
//     * phase A: after every 15 clean records one EXTRA record: an edge between two existing hosts that is NOT in the
//       clean graph, carrying a SKIPPED_REL bit - its first occurrence is flagged, so the pair is lost for good
//       (store.rs:313 de-duplicates first, harmonic.rs:131 filters after);
//     * phase B (after all clean records): for extra j a CLEAN copy of the same pair (stays lost), and a FLAGGED
//       duplicate of some clean edge (ignored: the clean copy came first).
//     The reference semantics therefore reduce the stream to exactly the clean graph (same node set, same CSR): the
//     oracle's dense run over hbs_row_ptr / hbs_src is the expected result.  tests/test_host.py checks this claim on
//     small graphs against the structure-faithful oracle.
struct hbs_stream_plan {
    uint64_t m, n, full_blocks, len_a, len, mul, add;
};

static uint64_t gcd64(uint64_t a, uint64_t b)
{
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

static hbs_stream_plan stream_plan(const hbs_graph *g, int salt)
{
    hbs_stream_plan p;
    p.m = g->edges.size();
    p.n = g->ids.size();
    p.full_blocks = (salt == 2 && p.n >= 4) ? p.m / 15 : 0;
    p.len_a = p.m + p.full_blocks;
    p.len = p.len_a + 2 * p.full_blocks;
    // affine permutation c -> (c * mul + add) mod m of the clean-edge index
    p.mul = 1;
    p.add = 0;
    if (p.m > 2) {
        uint64_t a = (uint64_t)((long double)p.m * 0.6180339887498949L) | 1;
        while (a >= p.m || gcd64(a, p.m) != 1) a = a >= p.m ? 1 : a + 2;
        p.mul = a;
        p.add = splitmix64(g->seed ^ 0x57EA) % p.m;
    }
    return p;
}

static inline uint64_t stream_perm(const hbs_stream_plan &p, uint64_t c)
{
    return p.m ? (uint64_t)(((unsigned __int128)c * p.mul + p.add) % p.m) : 0;
}

// extra j: (from, to) dense indices of a pair that is not a clean edge and not a self loop; false if none was found
// (then the extra degenerates to a flagged self loop, which the filter drops)
static inline bool stream_extra(const hbs_graph *g, const hbs_stream_plan &p, uint64_t j, uint32_t *from, uint32_t *to)
{
    for (int attempt = 0; attempt < 8; attempt++) {
        const uint64_t h = splitmix64(g->seed + 0xE17A + j * 0x2545F4914F6CDD1Dull + (uint64_t)attempt * 0x9E3779B97F4A7C15ull);
        const uint32_t f = (uint32_t)(h % p.n), t = (uint32_t)((h >> 32) % p.n);
        if (f == t) continue;
        // (f -> t) a clean edge?  the sources of row t are ascending: a short search inside the row
        const uint32_t *rb = g->src.data() + g->row_ptr[t], *re = g->src.data() + g->row_ptr[t + 1];
        if (std::binary_search(rb, re, f)) continue;
        *from = f;
        *to = t;
        return true;
    }
    *from = *to = (uint32_t)(splitmix64(j) % p.n);
    return false;
}

extern "C" {

uint64_t hbs_stream_len(const hbs_graph *g, int salt) { return stream_plan(g, salt).len; }
// pairs that only occur flagged-first (they count in m_unique, not in m_eff): for the expected hb_stats
uint64_t hbs_stream_lost_pairs(const hbs_graph *g, int salt)
{
    const hbs_stream_plan p = stream_plan(g, salt);
    // distinct (from, to) pairs among the extras (two extras may draw the same pair; degenerate extras are flagged self
    // loops (v, v)): each is one unique pair of the stream that never becomes an edge
    std::vector<uint64_t> pairs(p.full_blocks);
#pragma omp parallel for schedule(static) num_threads(omp_get_num_procs())
    for (int64_t j = 0; j < (int64_t)p.full_blocks; j++) {
        uint32_t f, t;
        stream_extra(g, p, (uint64_t)j, &f, &t);
        pairs[j] = ((uint64_t)t << 32) | f;
    }
    HBS_SORT(pairs.begin(), pairs.end());
    return (uint64_t)(std::unique(pairs.begin(), pairs.end()) - pairs.begin());
}

uint64_t hbs_stream_fill(const hbs_graph *g, int salt, uint64_t first, uint64_t count, hbs_edge *out)
{
    const hbs_stream_plan p = stream_plan(g, salt);
    if (first > p.len) return 0;
    if (count > p.len - first) count = p.len - first;
    static const int skipped_bits[12] = {8, 10, 11, 13, 14, 15, 16, 17, 18, 19, 21, 22};
    static const int harmless_bits[11] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 20};
#pragma omp parallel for schedule(static) num_threads(omp_get_num_procs())
    for (int64_t i = 0; i < (int64_t)count; i++) {
        const uint64_t pos = first + (uint64_t)i;
        uint32_t f, t;
        uint64_t flags = 0;
        if (pos < p.len_a) {
            const uint64_t b = pos / 16, r = pos % 16;
            if (b < p.full_blocks && r == 15) { // extra b: flagged first occurrence of a pair outside the clean graph
                stream_extra(g, p, b, &f, &t);
                const uint64_t h = splitmix64(b ^ 0xF1A6);
                flags = (1ull << skipped_bits[h % 12]) | ((h & 0x100) ? 1ull << harmless_bits[(h >> 16) % 11] : 0);
            } else {
                const uint64_t c = b < p.full_blocks ? 15 * b + r : pos - p.full_blocks;
                const uint64_t key = g->edges[stream_perm(p, c)];
                f = (uint32_t)key;
                t = (uint32_t)(key >> 32);
                if (salt == 2) {
                    const uint64_t h = splitmix64(c ^ 0xC1EA);
                    if ((h & 7) == 0) flags = 1ull << harmless_bits[(h >> 8) % 11];
                }
            }
        } else {
            const uint64_t q = pos - p.len_a, j = q / 2;
            if (q & 1) { // clean copy of extra j: stays lost (degenerate extras repeat their flagged self loop)
                const bool real = stream_extra(g, p, j, &f, &t);
                flags = real ? 0 : (1ull << skipped_bits[j % 12]);
            } else {     // flagged duplicate of a clean edge: ignored, the clean record came first
                const uint64_t key = g->edges[stream_perm(p, (j * 15 + 7) % p.m)];
                f = (uint32_t)key;
                t = (uint32_t)(key >> 32);
                flags = 1ull << skipped_bits[splitmix64(j ^ 0xD0B1) % 12];
            }
        }
        out[i].from = g->ids[f];
        out[i].to = g->ids[t];
        out[i].rel_flags = flags;
    }
    return count;
}

} // extern "C"
