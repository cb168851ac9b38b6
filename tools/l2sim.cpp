// l2sim.cpp - CPU model of the per-XCD L2 behaviour of one DENSE pass over a device plan (no GPU needed).
//
// Why: the dense pass is bounded by the rate of gathers that miss L2; the plan (device order, hub
// chunks cut at slice boundaries, XCD groups) decides how many miss.  This tool replays the access stream of the
// level-1 hub-chunk launch and of the node-row launch through 8 set-associative LRU caches (4 MiB, 128-byte lines,
// 16 ways) with the block -> XCD -> tile mapping of hb_kernels.hip.h, and reports hits / misses per source class.
// It is a planning aid: compare layouts here first, measure on the GPU after (profiles/r02*_l2sim*).
//
// build:  g++ -O2 -fopenmp -std=c++17 -Iinclude -Istract_amd/csrc tools/l2sim.cpp stract_amd/csrc/hb_host.cpp \
//             stract_amd/csrc/hb_synth.cpp -o tools/l2sim.bin
// usage:  tools/l2sim.bin <scale> <m_target> [band_w_log2] [minc] [direct_max] [chunk] [hub blocks/CU] [node blocks/CU] [streams bypass L2]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    std::vector<uint64_t> tag;   // sets * ways, 0 = empty (line + 1 stored)
    std::vector<uint32_t> stamp; // LRU time
    uint32_t clock = 0;
    explicit Cache(uint64_t bytes) : sets((uint32_t)(bytes / 128 / kWays)), tag((size_t)sets * kWays, 0), stamp((size_t)sets * kWays, 0) {}
    // returns true on hit
    bool access(uint64_t line, bool allocate = true)
    {
        uint64_t h = line * 0x9E3779B97F4A7C15ull;
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
        if (!allocate) return false;
        t[victim] = line + 1;
        st[victim] = clock;
        return false;
    }
};

// address map (bytes): [regs rd | regs wr | part | src | row_ptr | ksum.. ]
struct Layout {
    uint64_t rd, wr, part, src, rowptr, state;
};

struct Tally {
    uint64_t hit[8] = {0, 0, 0, 0, 0, 0, 0, 0}, miss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
const char *kClass[8] = {"gather slice 0 (hot)", "gather slices 1-7", "gather slices 8-64", "gather cold", "gather partial (virtual src)",
                         "index / row_ptr stream", "own counter / state stream", "writes (allocate)"};
inline int gather_class(uint32_t s, uint64_t n_pad)
{
    if (s >= n_pad) return 4;
    const uint32_t sl = s >> 16;
    return sl == 0 ? 0 : (sl < 8 ? 1 : (sl <= 64 ? 2 : 3));
}
} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <scale> <m_target> [band_w_log2] [minc] [direct_max] [chunk] [blocks_per_cu]\n", argv[0]);
        return 2;
    }
    const int scale = std::atoi(argv[1]);
    const uint64_t m_target = std::strtoull(argv[2], nullptr, 10);
    PlanTune pt;
    if (argc > 3 && std::atoi(argv[3]) > 0) pt.band_w = std::atoi(argv[3]) == 1 ? 0 : 1u << std::atoi(argv[3]);
    if (argc > 4 && std::atoi(argv[4]) > 0) pt.minc = (uint32_t)std::atoi(argv[4]);
    if (argc > 5 && std::atoi(argv[5]) > 0) pt.direct_max = (uint32_t)std::atoi(argv[5]);
    if (argc > 6 && std::atoi(argv[6]) > 0) pt.chunk = (uint32_t)std::atoi(argv[6]);
    const int bpc = argc > 7 ? std::atoi(argv[7]) : 2;
    const int node_bpc = argc > 8 ? std::atoi(argv[8]) : 64;
    const bool stream_bypass = argc > 9 && std::atoi(argv[9]) != 0; // model: index / state / write streams do not allocate in L2
    hbs_graph *g = hbs_rmat(scale, m_target, 0x5712AC7ull, 0);
    if (!g) return 1;
    const uint64_t n = hbs_num_nodes(g);
    std::vector<uint32_t> outdeg;
    count_out_degree(hbs_row_ptr(g), hbs_src(g), n, &outdeg);
    Plan p;
    std::string e = build_plan(n, hbs_row_ptr(g), hbs_src(g), outdeg, true, pt, &p);
    if (!e.empty()) {
        std::fprintf(stderr, "plan: %s\n", e.c_str());
        return 1;
    }
    std::printf("scale %d: n = %llu, m = %llu; n_pad = %llu, virtual rows = %llu, level-1 edges = %llu, direct edges = %llu, levels = %zu\n", scale,
                (unsigned long long)n, (unsigned long long)hbs_num_edges(g), (unsigned long long)p.n_pad, (unsigned long long)p.nv,
                (unsigned long long)p.level1_edges, (unsigned long long)p.direct_edges, p.level_begin.size() - 1);
    hbs_free(g);
    Layout L;
    L.rd = 0;
    L.wr = L.rd + p.n_pad * 64;
    L.part = L.wr + p.n_pad * 64;
    L.src = L.part + p.nv * 64;
    L.rowptr = L.src + p.src.size() * 4;
    L.state = L.rowptr + p.row_ptr.size() * 8;
    const int kBlocksPerXcd = 32 * bpc; // 256 CUs / 8 XCDs x blocks per CU
    const uint64_t *rp = p.row_ptr.data();
    const uint32_t *src = p.src.data();

    Tally tl[8], tn[8];
    // ---------------- level-1 launch: per XCD group, blocks k take tiles k, k + B, ...; all blocks advance together
#pragma omp parallel for schedule(dynamic, 1)
    for (int x = 0; x < 8; x++) {
        Cache c(4ull << 20);
        Tally &t = tl[x];
        const uint64_t lo = p.xcd_groups == 8 ? p.xcd_begin[x] : p.level_begin[0] + (p.level_begin[1] - p.level_begin[0]) * x / 8 / 64 * 64;
        const uint64_t hi = p.xcd_groups == 8 ? p.xcd_begin[x + 1] : (x == 7 ? p.level_begin[1] : p.level_begin[0] + (p.level_begin[1] - p.level_begin[0]) * (x + 1) / 8 / 64 * 64);
        const uint64_t ntiles = (hi - lo + 63) / 64;
        auto touch = [&](uint64_t addr, int cls) {
            if (c.access(addr >> 7, !(stream_bypass && cls >= 5))) t.hit[cls]++;
            else t.miss[cls]++;
        };
        for (uint64_t step = 0; step * kBlocksPerXcd < ntiles; step++) {
            const uint64_t t0 = step * kBlocksPerXcd, t1 = std::min<uint64_t>(ntiles, t0 + kBlocksPerXcd);
            // row pointers of the tiles of this step
            for (uint64_t tile = t0; tile < t1; tile++) touch(L.rowptr + (lo + tile * 64) * 8, 5), touch(L.rowptr + (lo + tile * 64 + 32) * 8, 5);
            for (uint32_t round = 0; round < 64 / 16 + 1; round++) { // 16 gathers per quad and round (unroll 4 x 4 lanes)
                for (uint64_t tile = t0; tile < t1; tile++) {
                    for (uint64_t r = lo + tile * 64; r < std::min(hi, lo + tile * 64 + 64); r++) {
                        const uint64_t b = rp[r] + (uint64_t)round * 16, en = std::min<uint64_t>(rp[r + 1], b + 16);
                        if (b >= en) continue;
                        touch(L.src + b * 4, 5);
                        for (uint64_t k = b; k < en; k++) touch((src[k] >= p.n_pad ? L.part - p.n_pad * 64 : L.rd) + (uint64_t)src[k] * 64, gather_class(src[k], p.n_pad));
                    }
                }
            }
            for (uint64_t tile = t0; tile < t1; tile++)
                for (uint64_t r = lo + tile * 64; r < std::min(hi, lo + tile * 64 + 64); r += 2) touch(L.part + (r - p.n_pad) * 64, 7);
        }
    }
    // ---------------- node-row launch: plain grid stride, block b on XCD b % 8
    {
        const uint64_t ntiles = p.n_pad / 64, G = (uint64_t)std::min(32 * node_bpc, 32 * 4) * 8; // resident: 4 workgroups per CU
#pragma omp parallel for schedule(dynamic, 1)
        for (int x = 0; x < 8; x++) {
            Cache c(4ull << 20);
            Tally &t = tn[x];
            auto touch = [&](uint64_t addr, int cls) {
                if (c.access(addr >> 7, !(stream_bypass && cls >= 5))) t.hit[cls]++;
                else t.miss[cls]++;
            };
            for (uint64_t step = 0; step * G < ntiles; step++) {
                for (uint32_t round = 0; round < 64 / 8 + 1; round++) { // unroll 2 x 4 lanes
                    for (uint64_t b = (uint64_t)x; b < G; b += 8) {
                        const uint64_t tile = step * G + b;
                        if (tile >= ntiles) break;
                        for (uint64_t r = tile * 64; r < tile * 64 + 64; r++) {
                            if (round == 0 && (r & 15) == 0) touch(L.rowptr + r * 8, 5);
                            if (round == 0 && (r & 1) == 0) touch(L.rd + r * 64, 6), touch(L.state + r * 24, 6);
                            const uint64_t bb = rp[r] + (uint64_t)round * 8, en = std::min<uint64_t>(rp[r + 1], bb + 8);
                            if (bb >= en) continue;
                            touch(L.src + bb * 4, 5);
                            for (uint64_t k = bb; k < en; k++)
                                touch((src[k] >= p.n_pad ? L.part - p.n_pad * 64 : L.rd) + (uint64_t)src[k] * 64, gather_class(src[k], p.n_pad));
                        }
                    }
                }
                for (uint64_t b = (uint64_t)x; b < G; b += 8) {
                    const uint64_t tile = step * G + b;
                    if (tile >= ntiles) break;
                    for (uint64_t r = tile * 64; r < tile * 64 + 64; r += 2) touch(L.wr + r * 64, 7);
                }
            }
        }
    }
    auto report = [&](const char *name, Tally *t) {
        uint64_t H = 0, M = 0, gh = 0, gm = 0;
        std::printf("%s\n", name);
        for (int cls = 0; cls < 8; cls++) {
            uint64_t h = 0, m = 0;
            for (int x = 0; x < 8; x++) h += t[x].hit[cls], m += t[x].miss[cls];
            if (h + m == 0) continue;
            std::printf("  %-30s %11llu accesses  hit %5.1f %%  misses %10llu\n", kClass[cls], (unsigned long long)(h + m), 100.0 * h / (h + m), (unsigned long long)m);
            H += h;
            M += m;
            if (cls <= 4) gh += h, gm += m;
        }
        std::printf("  %-30s %11llu accesses  hit %5.1f %%  misses %10llu   (gathers only: hit %.1f %%, misses %llu)\n", "all", (unsigned long long)(H + M),
                    100.0 * H / (H + M), (unsigned long long)M, 100.0 * gh / std::max<uint64_t>(gh + gm, 1), (unsigned long long)gm);
        uint64_t lo = ~0ull, hi = 0;
        for (int x = 0; x < 8; x++) {
            uint64_t a = 0;
            for (int cls = 0; cls < 8; cls++) a += t[x].miss[cls] * 21 + t[x].hit[cls] * 6;
            lo = std::min(lo, a);
            hi = std::max(hi, a);
        }
        std::printf("  per-XCD cost model (21 ps / miss + 6 ps / hit, x8 since one XCD has 1/8 of the rate): min %.3f ms, max %.3f ms\n", lo * 8e-9, hi * 8e-9);
    };
    report("level-1 hub-chunk launch (XCD groups)", tl);
    report("node-row launch", tn);
    return 0;
}
