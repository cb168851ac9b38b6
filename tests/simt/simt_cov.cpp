// tests/simt/simt_cov.cpp - TEST INFRASTRUCTURE: which basic blocks of the library's sources (device code included) do the tests reach?
// The coverage build (`make cov`) compiles stract_amd/csrc/*.hip with -fsanitize-coverage=trace-pc-guard,pc-table; these are the
// callbacks.  Every process writes <HB_SIMT_COV_DIR>/<pid>.cov at exit: one line per instrumented block, "offset-in-the-library hit".
// tools/simt_coverage.py merges the files and turns offsets into source lines (llvm-symbolizer).
#include <dlfcn.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
struct Module {
    uint32_t *g0, *g1;
    const uintptr_t *p0 = nullptr, *p1 = nullptr;
    uint32_t first_id;
};
std::vector<Module> &mods()
{
    static std::vector<Module> *m = new std::vector<Module>(); // never destroyed: dump() runs among the static destructors
    return *m;
}
std::vector<uint8_t> &hits()
{
    static std::vector<uint8_t> *h = new std::vector<uint8_t>(1, 0);
    return *h;
}
std::vector<std::pair<const uintptr_t *, const uintptr_t *>> &tables();
void dump()
{
    const char *dir = std::getenv("HB_SIMT_COV_DIR");
    if (!dir) return;
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%d.cov", dir, (int)getpid());
    FILE *f = std::fopen(path, "w");
    if (!f) return;
    for (Module &m : mods()) {
        for (auto &t : tables())
            if (!m.p0 && (size_t)(t.second - t.first) == 2 * (size_t)(m.g1 - m.g0)) {
                m.p0 = t.first;
                m.p1 = t.second;
                t.second = t.first; // used
            }
        if (!m.p0) continue;
        const size_t n = (size_t)(m.g1 - m.g0);
        for (size_t i = 0; i < n && m.p0 + 2 * i < m.p1; i++) {
            Dl_info info;
            uintptr_t pc = m.p0[2 * i];
            uintptr_t base = dladdr((void *)pc, &info) ? (uintptr_t)info.dli_fbase : 0;
            std::fprintf(f, "%lx %d\n", (unsigned long)(pc - base), (int)hits()[m.first_id + i]);
        }
    }
    std::fclose(f);
}
} // namespace

extern "C" void __sanitizer_cov_trace_pc_guard_init(uint32_t *start, uint32_t *stop)
{
    if (start == stop || *start) return;
    static bool registered = false;
    if (!registered) {
        registered = true;
        std::atexit(dump);
    }
    Module m;
    m.g0 = start;
    m.g1 = stop;
    m.first_id = (uint32_t)hits().size();
    for (uint32_t *g = start; g < stop; g++) *g = (uint32_t)(m.first_id + (g - start));
    hits().resize(hits().size() + (size_t)(stop - start), 0);
    mods().push_back(m);
}
// (the two initialisers of a module may arrive in either order: the tables are paired with the guard ranges when the file is written)
namespace {
std::vector<std::pair<const uintptr_t *, const uintptr_t *>> &tables()
{
    static auto *t = new std::vector<std::pair<const uintptr_t *, const uintptr_t *>>();
    return *t;
}
} // namespace
extern "C" void __sanitizer_cov_pcs_init(const uintptr_t *beg, const uintptr_t *end)
{
    for (auto &t : tables())
        if (t.first == beg) return;
    tables().push_back({beg, end});
}
extern "C" void __sanitizer_cov_trace_pc_guard(uint32_t *guard)
{
    const uint32_t id = *guard;
    if (id) hits()[id] = 1;
}
