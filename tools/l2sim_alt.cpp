// l2sim_alt.cpp - what-if companion of l2sim.cpp: a small stand-alone planner (one chunk level, no trees) whose cutting and
// XCD-grouping policy can be changed in a few lines, replayed through the same L2 model.  Policy 0 restates the shipped
// layout (slice 0 = flexible "hot" chunks on every XCD, slices 1..64 pinned slice mod 8, cold flexible) so that the model
// can be compared with l2sim.cpp on the real plan; the other policies are candidates for the next planner.
//
// build:  g++ -O2 -fopenmp -std=c++17 -Iinclude -Istract_amd/csrc tools/l2sim_alt.cpp stract_amd/csrc/hb_host.cpp \
//             stract_amd/csrc/hb_synth.cpp -o tools/l2sim_alt.bin
// usage:  tools/l2sim_alt.bin <scale> <m_target> <policy> [hot_log2=16] [slice_log2=16] [minc=8] [hub blocks/CU=2]
//   policy 0  shipped layout
//   policy 1  hot range [0, 2^hot_log2) cut into 8 sub-ranges of equal out-degree mass, sub-range x pinned to XCD x
//             (no L2 holds another L2's hot lines); warm slices and cold as in policy 0
//   policy 2  policy 1 + warm slices cut by equal mass as well (8 per "ring", ring r slice x pinned to XCD x)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "hb_internal.h"

extern "C" {
struct hbs_graph;
hbs_graph *hbs_rmat(int scale, uint64_t m_target, uint64_t seed, int threads);
void hbs_free(hbs_graph *g);
uint64_t hbs_num_nodes(const hbs_graph *g);
uint64_t hbs_num_edges(const hbs_graph *g);
const uint64_t *hbs_row_ptr(const hbs_graph *g);
const uint32_t *hbs_src(const hbs_graph *g);
}

using namespace hb;

namespace {
struct Cache {
    static constexpr int kWays = 16;
    uint32_t sets;
    std::vector<uint64_t> tag;
    std::vector<uint32_t> stamp;
    uint32_t clock = 0;
    explicit Cache(uint64_t bytes) : sets((uint32_t)(bytes / 128 / kWays)), tag((size_t)sets * kWays, 0), stamp((size_t)sets * kWays, 0) {}
    bool access(uint64_t line)
    {
        const uint64_t h = line * 0x9E3779B97F4A7C15ull;
        const uint32_t s = (uint32_t)((h >> 32) % sets);
        uint64_t *t = &tag[(size_t)s * kWays];
        uint32_t *st = &stamp[(size_t)s * kWays];
        clock++;
        int victim = 0;
        for (int w = 0; w < kWays; w++) {
            if (t[w] == line + 1) {
                st[w] = clock;
                return true;
            }
            if (st[w] < st[victim]) victim = w;
        }
        t[victim] = line + 1;
        st[victim] = clock;
        return false;
    }
};
struct Chunk {
    uint64_t beg;
    uint32_t len, key, row;
};
struct Tally {
    uint64_t hit[4] = {0, 0, 0, 0}, miss[4] = {0, 0, 0, 0}; // 0 gathers of counters, 1 partial reads, 2 streams, 3 writes
};
} // namespace

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <scale> <m_target> <policy> [hot_log2] [slice_log2] [minc] [hub blocks/CU]\n", argv[0]);
        return 2;
    }
    const int scale = std::atoi(argv[1]);
    const uint64_t m_target = std::strtoull(argv[2], nullptr, 10);
    const int policy = std::atoi(argv[3]);
    const uint32_t H = 1u << (argc > 4 ? std::atoi(argv[4]) : 16), W = 1u << (argc > 5 ? std::atoi(argv[5]) : 16);
    const uint32_t minc = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 8, chunk = 64, direct_max = 64, kWarm = 64;
    const int bpc = argc > 7 ? std::atoi(argv[7]) : 2;
    hbs_graph *g = hbs_rmat(scale, m_target, 0x5712AC7ull, 0);
    if (!g) return 1;
    const uint64_t n = hbs_num_nodes(g), m = hbs_num_edges(g);
    const uint64_t *grp = hbs_row_ptr(g);
    const uint32_t *gsrc = hbs_src(g);
    std::vector<uint32_t> outdeg;
    count_out_degree(grp, gsrc, n, &outdeg);
    // device order = hotness order: descending out-degree, ties by sid
    std::vector<uint32_t> order(n), rank_of(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return outdeg[a] > outdeg[b]; });
    for (uint64_t d = 0; d < n; d++) rank_of[order[d]] = (uint32_t)d;
    std::vector<uint64_t> rp(n + 1, 0);
    for (uint64_t d = 0; d < n; d++) rp[d + 1] = rp[d] + (grp[order[d] + 1] - grp[order[d]]);
    std::vector<uint32_t> rs(m);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t d = 0; d < (int64_t)n; d++) {
        const uint32_t s = order[d];
        uint64_t o = rp[d];
        for (uint64_t k = grp[s]; k < grp[s + 1]; k++) rs[o++] = rank_of[gsrc[k]];
        std::sort(rs.begin() + rp[d], rs.begin() + rp[d + 1]);
    }
    // equal-mass boundaries of the hot range / of the warm rings
    std::vector<uint64_t> cum(n + 1, 0);
    for (uint64_t d = 0; d < n; d++) cum[d + 1] = cum[d] + outdeg[order[d]];
    std::vector<uint32_t> hot_cut(9, 0); // sub-range x = ranks [hot_cut[x], hot_cut[x+1])
    for (int x = 0; x <= 8; x++) {
        const uint64_t target = cum[std::min<uint64_t>(H, n)] * x / 8;
        hot_cut[x] = (uint32_t)(std::lower_bound(cum.begin(), cum.begin() + std::min<uint64_t>(H, n) + 1, target) - cum.begin());
    }
    hot_cut[8] = (uint32_t)std::min<uint64_t>(H, n);
    // policy 2: warm rings: ring r covers ranks [H + r*8*W', ...) cut into 8 equal-mass pieces; simpler: pieces of equal mass
    // = the mass of one hot sub-range, up to 64 pieces, each at most W counters wide
    std::vector<uint32_t> warm_cut;
    if (policy == 2) {
        uint32_t at = hot_cut[8];
        warm_cut.push_back(at);
        const uint64_t piece = cum[hot_cut[8]] / 8;
        while (warm_cut.size() <= kWarm && at < n) {
            uint64_t target = cum[at] + piece;
            uint32_t nx = (uint32_t)(std::lower_bound(cum.begin() + at, cum.end(), target) - cum.begin());
            nx = std::min<uint32_t>(std::min<uint64_t>(nx, (uint64_t)at + W), (uint32_t)n);
            if (nx <= at) nx = at + 1;
            warm_cut.push_back(nx);
            at = nx;
        }
    }
    auto band_of = [&](uint32_t r) -> uint32_t { // key: < 1000 pinned or hot, >= 1000 cold
        if (policy == 0) {
            if (r < W) return 0;
            const uint64_t j = r / W;
            if (j <= kWarm) return (uint32_t)j;
            return 1000 + (uint32_t)(63 - __builtin_clzll(j / (kWarm + 1)));
        }
        if (r < hot_cut[8]) return (uint32_t)(std::upper_bound(hot_cut.begin(), hot_cut.end(), r) - hot_cut.begin() - 1); // 0..7
        if (policy == 2) {
            if (r < warm_cut.back()) return 8 + (uint32_t)(std::upper_bound(warm_cut.begin(), warm_cut.end(), r) - warm_cut.begin() - 1);
            return 1000 + (uint32_t)(63 - __builtin_clzll((uint64_t)r / W + 1));
        }
        const uint64_t j = (r - hot_cut[8]) / W;
        if (j < kWarm) return 8 + (uint32_t)j;
        return 1000 + (uint32_t)(63 - __builtin_clzll(j / kWarm + 1));
    };
    auto xcd_of = [&](uint32_t key) -> int { // -1 = flexible
        if (key >= 1000) return -1;
        if (policy == 0) return key == 0 ? -1 : (int)(key & 7u);
        return key < 8 ? (int)key : (int)((key - 8) & 7u);
    };
    // ---- cut hub rows
    std::vector<Chunk> chunks;
    std::vector<uint64_t> first_chunk(n + 1, 0);
    uint64_t direct_edges = 0;
    for (uint64_t d = 0; d < n; d++) {
        first_chunk[d] = chunks.size();
        const uint64_t b = rp[d], e = rp[d + 1];
        if (e - b <= direct_max) {
            direct_edges += e - b;
            continue;
        }
        uint64_t i = b;
        while (i < e) {
            const uint32_t b0 = band_of(rs[i]);
            uint64_t j = i + 1;
            while (j < e && j - i < chunk) {
                if (j - i >= minc && band_of(rs[j]) != b0) break;
                j++;
            }
            if (e - j < minc && e - i <= chunk) j = e;
            chunks.push_back({i, (uint32_t)(j - i), b0, (uint32_t)d});
            i = j;
        }
    }
    first_chunk[n] = chunks.size();
    // ---- groups: pinned by key, flexible ones (hot in policy 0, cold everywhere) to the least loaded group
    std::vector<uint8_t> grp_of(chunks.size(), 0);
    uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> flex;
    for (size_t k = 0; k < chunks.size(); k++) {
        const int x = xcd_of(chunks[k].key);
        if (x >= 0) {
            grp_of[k] = (uint8_t)x;
            load[x] += chunks[k].len + 4;
        } else {
            flex.push_back((uint32_t)k);
        }
    }
    uint64_t pinned_min = ~0ull, pinned_max = 0;
    for (int x = 0; x < 8; x++) pinned_min = std::min(pinned_min, load[x]), pinned_max = std::max(pinned_max, load[x]);
    // cold first (as slow as warm), then hot: both to the least loaded group, in (key, longer first) order
    std::stable_sort(flex.begin(), flex.end(), [&](uint32_t a, uint32_t b) {
        const bool ca = chunks[a].key >= 1000, cb = chunks[b].key >= 1000;
        if (ca != cb) return ca;
        return chunks[a].key < chunks[b].key;
    });
    for (uint32_t k : flex) {
        int best = 0;
        for (int x = 1; x < 8; x++)
            if (load[x] < load[best]) best = x;
        grp_of[k] = (uint8_t)best;
        load[best] += chunks[k].len + 4;
    }
    std::vector<uint32_t> corder(chunks.size());
    std::iota(corder.begin(), corder.end(), 0u);
    std::stable_sort(corder.begin(), corder.end(), [&](uint32_t a, uint32_t b) {
        if (grp_of[a] != grp_of[b]) return grp_of[a] < grp_of[b];
        if (chunks[a].key != chunks[b].key) return chunks[a].key < chunks[b].key;
        return chunks[a].len > chunks[b].len;
    });
    std::vector<uint32_t> vid_of(chunks.size());
    uint64_t gbeg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t r = 0; r < corder.size(); r++) {
        vid_of[corder[r]] = (uint32_t)r;
        gbeg[grp_of[corder[r]] + 1] = r + 1;
    }
    for (int x = 1; x <= 8; x++) gbeg[x] = std::max(gbeg[x], gbeg[x - 1]);
    uint64_t lmin = ~0ull, lmax = 0;
    for (int x = 0; x < 8; x++) lmin = std::min(lmin, load[x]), lmax = std::max(lmax, load[x]);
    std::printf("scale %d policy %d: n = %llu, m = %llu; chunks = %zu (%.1f sources each), direct edges = %llu; group load min/max = %.3f / %.3f of mean (pinned only: %.3f / %.3f)\n",
                scale, policy, (unsigned long long)n, (unsigned long long)m, chunks.size(), chunks.empty() ? 0.0 : (double)(m - direct_edges) / chunks.size(),
                (unsigned long long)direct_edges, lmin * 8.0 / std::accumulate(load, load + 8, 0ull), lmax * 8.0 / std::accumulate(load, load + 8, 0ull),
                pinned_min * 8.0 / std::max<uint64_t>(1, std::accumulate(load, load + 8, 0ull)), pinned_max * 8.0 / std::max<uint64_t>(1, std::accumulate(load, load + 8, 0ull)));
    if (policy) {
        std::printf("  hot sub-ranges (ranks):");
        for (int x = 0; x < 8; x++) std::printf(" %u", hot_cut[x + 1] - hot_cut[x]);
        std::printf("\n");
    }
    hbs_free(g);
    // ---- replay
    const uint64_t A_rd = 0, A_wr = n * 64, A_part = 2 * n * 64, A_src = A_part + chunks.size() * 64, A_state = A_src + m * 4 + chunks.size() * 8;
    Tally tl[8], tn[8];
    const uint64_t B = (uint64_t)32 * bpc;
#pragma omp parallel for schedule(dynamic, 1)
    for (int x = 0; x < 8; x++) {
        Cache c(4ull << 20);
        Tally &t = tl[x];
        auto touch = [&](uint64_t addr, int cls) {
            if (c.access(addr >> 7)) t.hit[cls]++;
            else t.miss[cls]++;
        };
        const uint64_t lo = gbeg[x], hi = gbeg[x + 1], ntiles = (hi - lo + 63) / 64;
        for (uint64_t step = 0; step * B < ntiles; step++) {
            const uint64_t t0 = step * B, t1 = std::min(ntiles, t0 + B);
            for (uint32_t round = 0; round < 5; round++)
                for (uint64_t tile = t0; tile < t1; tile++)
                    for (uint64_t r = lo + tile * 64; r < std::min(hi, lo + tile * 64 + 64); r++) {
                        const Chunk &ch = chunks[corder[r]];
                        const uint64_t b = ch.beg + (uint64_t)round * 16, en = std::min<uint64_t>(ch.beg + ch.len, b + 16);
                        if (b >= en) continue;
                        touch(A_src + b * 4, 2);
                        for (uint64_t k = b; k < en; k++) touch(A_rd + (uint64_t)rs[k] * 64, 0);
                    }
            for (uint64_t tile = t0; tile < t1; tile++)
                for (uint64_t r = lo + tile * 64; r < std::min(hi, lo + tile * 64 + 64); r += 2) touch(A_part + r * 64, 3);
        }
    }
    {
        const uint64_t ntiles = (n + 63) / 64, G = 128 * 8;
#pragma omp parallel for schedule(dynamic, 1)
        for (int x = 0; x < 8; x++) {
            Cache c(4ull << 20);
            Tally &t = tn[x];
            auto touch = [&](uint64_t addr, int cls) {
                if (c.access(addr >> 7)) t.hit[cls]++;
                else t.miss[cls]++;
            };
            for (uint64_t step = 0; step * G < ntiles; step++) {
                for (uint32_t round = 0; round < 9; round++)
                    for (uint64_t b = (uint64_t)x; b < G; b += 8) {
                        const uint64_t tile = step * G + b;
                        if (tile >= ntiles) break;
                        for (uint64_t r = tile * 64; r < std::min(n, tile * 64 + 64); r++) {
                            if (round == 0 && (r & 1) == 0) touch(A_rd + r * 64, 2), touch(A_state + r * 32, 2);
                            const uint64_t nch = first_chunk[r + 1] - first_chunk[r];
                            if (nch) { // hub row: one partial per chunk (the real plan reads them through a tree)
                                for (uint64_t k = first_chunk[r] + (uint64_t)round * 8; k < std::min(first_chunk[r + 1], first_chunk[r] + (uint64_t)round * 8 + 8); k++)
                                    touch(A_part + (uint64_t)vid_of[k] * 64, 1);
                            } else {
                                const uint64_t bb = rp[r] + (uint64_t)round * 8, en = std::min<uint64_t>(rp[r + 1], bb + 8);
                                if (bb >= en) continue;
                                touch(A_src + bb * 4, 2);
                                for (uint64_t k = bb; k < en; k++) touch(A_rd + (uint64_t)rs[k] * 64, 0);
                            }
                        }
                    }
                for (uint64_t b = (uint64_t)x; b < G; b += 8) {
                    const uint64_t tile = step * G + b;
                    if (tile >= ntiles) break;
                    for (uint64_t r = tile * 64; r < std::min(n, tile * 64 + 64); r += 2) touch(A_wr + r * 64, 3);
                }
            }
        }
    }
    const char *names[4] = {"counter gathers", "partial reads", "streams", "writes"};
    uint64_t total_miss = 0;
    for (int part = 0; part < 2; part++) {
        Tally *t = part ? tn : tl;
        std::printf("%s\n", part ? "node-row launch" : "level-1 hub-chunk launch");
        uint64_t M = 0, Hh = 0, xmin = ~0ull, xmax = 0;
        for (int cls = 0; cls < 4; cls++) {
            uint64_t h = 0, mm = 0;
            for (int x = 0; x < 8; x++) h += t[x].hit[cls], mm += t[x].miss[cls];
            if (h + mm) std::printf("  %-18s %11llu accesses  hit %5.1f %%  misses %10llu\n", names[cls], (unsigned long long)(h + mm), 100.0 * h / (h + mm), (unsigned long long)mm);
            M += mm;
            Hh += h;
        }
        for (int x = 0; x < 8; x++) {
            uint64_t a = 0;
            for (int cls = 0; cls < 4; cls++) a += t[x].miss[cls] * 21 + t[x].hit[cls] * 6;
            xmin = std::min(xmin, a);
            xmax = std::max(xmax, a);
        }
        std::printf("  all: hit %.1f %%, misses %llu; per-XCD cost model min %.3f ms, max %.3f ms\n", 100.0 * Hh / std::max<uint64_t>(1, Hh + M), (unsigned long long)M, xmin * 8e-9, xmax * 8e-9);
        total_miss += M;
    }
    std::printf("total misses per dense pass: %llu\n", (unsigned long long)total_miss);
    return 0;
}
