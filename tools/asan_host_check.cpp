// asan_host_check.cpp - runs the host stages (ingest semantics, planner) of the library under
// AddressSanitizer / UBSan on random inputs and re-checks the planner's coverage invariant in C++.
// Build + run (CPU only):
//   g++ -O1 -g -std=c++17 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
//       tools/asan_host_check.cpp stract_amd/csrc/hb_host.cpp -o /tmp/asan_host_check && /tmp/asan_host_check
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <set>

#include "../stract_amd/csrc/hb_internal.h"

using namespace hb;

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); failures++; } } while (0)

int main()
{
    std::mt19937_64 rng(12345);
    for (int iter = 0; iter < 300; iter++) {
        const uint64_t idspace = 2 + rng() % 700;
        const uint64_t m = rng() % (6 * idspace);
        std::vector<hb_edge> edges(m);
        for (auto &e : edges) {
            e.from.lo = 1 + rng() % idspace; e.from.hi = (rng() % 5 == 0) ? rng() % 3 : 0;
            e.to.lo = 1 + rng() % idspace;   e.to.hi = (rng() % 5 == 0) ? rng() % 3 : 0;
            e.rel_flags = (rng() % 6 == 0) ? (1ull << (8 + rng() % 15)) : 0;
        }
        if (iter % 7 == 0 && m > 10) // one mega hub
            for (uint64_t i = 0; i < m / 2; i++) edges[i].to = edges[0].to;
        DenseGraph g;
        std::string err = ingest_edges(nullptr, 0, edges.data(), m, &g);
        CHECK(err.empty(), "ingest: %s", err.c_str());
        const uint64_t n = g.ids.size();
        CHECK(g.row_ptr.size() == n + 1 && g.row_ptr[n] == g.src.size(), "csr sizes");
        // reference semantics restated with std::set
        std::set<std::pair<std::pair<uint64_t, uint64_t>, std::pair<uint64_t, uint64_t>>> seen;
        uint64_t want_eff = 0;
        for (auto &e : edges) {
            auto k = std::make_pair(std::make_pair(e.from.hi, e.from.lo), std::make_pair(e.to.hi, e.to.lo));
            if (!seen.insert(k).second) continue;
            if (!(e.rel_flags & HB_SKIPPED_REL_MASK)) want_eff++;
        }
        CHECK(seen.size() == g.m_unique && want_eff == g.src.size(), "dedup/filter counts %zu %llu %llu %zu", seen.size(),
              (unsigned long long)g.m_unique, (unsigned long long)want_eff, g.src.size());
        std::vector<uint32_t> outdeg;
        count_out_degree(g.row_ptr.data(), g.src.data(), n, &outdeg);
        PlanTune t;
        const uint32_t chunks[] = {4, 5, 8, 16, 64};
        t.chunk = chunks[rng() % 5];
        t.band_w = (rng() % 4 == 0) ? 0 : (1u << (2 + rng() % 6));
        t.minc = 1 + rng() % 8;
        t.direct_max = rng() % (t.chunk + 1);
        t.xcd_map = rng() % 2;
        t.world = 1 + rng() % 4;
        DenseGraph gl = g;
        const uint64_t rank = rng() % t.world;
        if (t.world > 1) keep_owned_rows(&gl, t.world, rank);
        Plan p;
        err = build_plan(n, gl.row_ptr.data(), gl.src.data(), outdeg, rng() % 2, t, &p);
        CHECK(err.empty(), "plan: %s", err.c_str());
        if (!err.empty() || n == 0) continue;
        const uint64_t rows = p.n_pad + p.nv;
        CHECK(p.row_ptr.size() == rows + 1 && p.row_ptr[rows] == p.src.size(), "plan sizes");
        std::function<void(uint64_t, std::vector<uint32_t> &)> expand = [&](uint64_t row, std::vector<uint32_t> &out) {
            CHECK(p.row_ptr[row + 1] - p.row_ptr[row] <= std::max<uint32_t>(t.chunk, 4), "row longer than chunk");
            for (uint64_t k = p.row_ptr[row]; k < p.row_ptr[row + 1]; k++) {
                const uint32_t s = p.src[k];
                if (s >= p.n_pad) { CHECK(s < rows && s < row + rows, "vid range"); expand(s, out); }
                else out.push_back(s);
            }
        };
        for (uint64_t d = 0; d < p.n_pad; d++) {
            const uint32_t sid = p.order[d];
            std::vector<uint32_t> got, want;
            expand(d, got);
            if (sid != kNone)
                for (uint64_t k = gl.row_ptr[sid]; k < gl.row_ptr[sid + 1]; k++) want.push_back(p.dev_of[gl.src[k]]);
            std::sort(got.begin(), got.end());
            std::sort(want.begin(), want.end());
            CHECK(got == want, "coverage of row %llu (iter %d)", (unsigned long long)d, iter);
        }
    }
    std::printf(failures ? "asan_host_check: %d FAILURES\n" : "asan_host_check: ok\n", failures);
    return failures ? 1 : 0;
}
