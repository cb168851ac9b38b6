// tests/simt/simt_core.cpp - TEST INFRASTRUCTURE (see simt.h): the fiber scheduler of the SIMT interpreter.
#include "simt.h"

#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__x86_64__)
#error "the SIMT interpreter's context switch is written for x86-64"
#endif

// void simt_switch(void **save_sp, void *load_sp): callee-saved registers + stack pointer
extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(".text\n"
    ".globl simt_switch\n"
    ".type simt_switch,@function\n"
    "simt_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size simt_switch, .-simt_switch\n");

// AddressSanitizer build (`make asan`): tell the runtime about every stack switch
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SIMT_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif

namespace simt {
thread_local Idx threadIdx_, blockIdx_, blockDim_, gridDim_;

namespace {
constexpr size_t kStack = 512 * 1024;
enum State { RUN, WAIT, DONE };
struct Lane {
    void *sp = nullptr;
    State st = DONE;
    int kind = 0;
    uint64_t value = 0, arg = 0, result = 0;
    const void *site = nullptr;
};
struct Run {
    std::vector<Lane> lanes;
    char *stacks = nullptr;
    size_t nstacks = 0;
    void *sched_sp = nullptr;
    int cur = -1;
    const std::function<void()> *body = nullptr;
};
thread_local Run g;
Stats g_stats;

[[noreturn]] void die(const char *what)
{
    std::fprintf(stderr, "[simt] %s (block %u, lane %d)\n", what, blockIdx_.x, g.cur);
    std::abort();
}

#ifdef SIMT_ASAN
thread_local const void *g_sched_bottom = nullptr;
thread_local size_t g_sched_size = 0;
#endif
// lane -> scheduler
void to_scheduler(void **save_sp, bool final_switch)
{
#ifdef SIMT_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(final_switch ? nullptr : &fake, g_sched_bottom, g_sched_size);
#endif
    simt_switch(save_sp, g.sched_sp);
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(fake, &g_sched_bottom, &g_sched_size);
#endif
    (void)final_switch;
}

void lane_entry()
{
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
    (*g.body)();
    g.lanes[g.cur].st = DONE;
    void *dummy;
    to_scheduler(&dummy, true);
    die("a finished lane was resumed");
}

void prepare_lane(int i)
{
    char *top = g.stacks + (size_t)(i + 1) * kStack;
#ifdef SIMT_ASAN
    __asan_unpoison_memory_region(top - kStack, kStack); // frames of the lane that used this stack before never returned
#endif
    uint64_t *sp = (uint64_t *)top;
    *--sp = 0;                     // fake return address of lane_entry (it never returns)
    *--sp = (uint64_t)&lane_entry; // popped by simt_switch's `ret`
    for (int k = 0; k < 6; k++) *--sp = 0; // rbp rbx r12 r13 r14 r15
    g.lanes[i].sp = sp;
    g.lanes[i].st = RUN;
}

// serve the lanes of wave [w0, w1) that wait at the lowest code address
bool serve_wave(int w0, int w1)
{
    const void *site = nullptr;
    for (int i = w0; i < w1; i++) {
        const Lane &L = g.lanes[i];
        if (L.st == WAIT && L.kind != K_BLOCK_SYNC && (!site || L.site < site)) site = L.site;
    }
    if (!site) return false;
    bool in[64];
    int kind = 0, members = 0, live = 0;
    for (int i = w0; i < w1; i++) {
        const Lane &L = g.lanes[i];
        live += L.st != DONE;
        in[i - w0] = L.st == WAIT && L.kind != K_BLOCK_SYNC && L.site == site;
        if (in[i - w0]) {
            if (kind && kind != L.kind) die("lanes wait at one address with different operations");
            kind = L.kind;
            members++;
        }
    }
    if (members < live) __atomic_fetch_add(&g_stats.partial_groups, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_stats.collectives, 1, __ATOMIC_RELAXED);
    uint64_t ballot = 0;
    if (kind == K_BALLOT)
        for (int i = w0; i < w1; i++)
            if (in[i - w0] && g.lanes[i].value) ballot |= 1ull << (i - w0);
    uint64_t res[64];
    for (int i = w0; i < w1; i++) {
        if (!in[i - w0]) continue;
        const Lane &L = g.lanes[i];
        const int l = i - w0;
        int src = l;
        switch (kind) {
        case K_BALLOT: res[l] = ballot; continue;
        case K_WAVE_SYNC: res[l] = 0; continue;
        case K_SHFL: src = (int)(L.arg & 63); break;
        case K_SHFL_UP: src = l >= (int)L.arg ? l - (int)L.arg : l; break;
        case K_SHFL_DOWN: src = l + (int)L.arg < 64 ? l + (int)L.arg : l; break;
        case K_SHFL_XOR: src = (l ^ (int)L.arg) & 63; break;
        case K_DPP: src = (l & ~3) | (int)((L.arg >> (2 * (l & 3))) & 3); break;
        default: die("unknown cross-lane operation");
        }
        if (src < w1 - w0 && in[src]) {
            res[l] = g.lanes[w0 + src].value;
        } else { // the machine returns 0 for a DPP read of an inactive lane (bound_ctrl) and the own value for a shuffle out of range
            res[l] = kind == K_DPP ? 0 : L.value;
            if (src != l) __atomic_fetch_add(&g_stats.reads_of_inactive_lanes, 1, __ATOMIC_RELAXED);
        }
    }
    for (int i = w0; i < w1; i++)
        if (in[i - w0]) {
            g.lanes[i].result = res[i - w0];
            g.lanes[i].st = RUN;
        }
    return true;
}

// HB_SIMT_ORDER: in which order the workgroups of a launch, and the runnable lanes of a workgroup between two cross-lane
// operations, take their turns: "forward" (default), "reverse", or "shuffle:<seed>" (a different permutation per launch / per
// sweep).  The machine promises no order at all; results that depend on one are races the default order would hide.
struct Order {
    int mode = 0; // 0 forward, 1 reverse, 2 shuffle
    uint64_t state = 0x9E3779B97F4A7C15ull;
    Order()
    {
        const char *e = std::getenv("HB_SIMT_ORDER");
        if (!e) return;
        if (!std::strcmp(e, "reverse")) mode = 1;
        else if (!std::strncmp(e, "shuffle", 7)) {
            mode = 2;
            if (e[7] == ':') state ^= std::strtoull(e + 8, nullptr, 10) * 0xD1B54A32D192ED03ull;
        }
    }
    static uint64_t next(uint64_t &st)
    {
        st ^= st << 13;
        st ^= st >> 7;
        st ^= st << 17;
        return st;
    }
    // `st`: the generator to draw from (the launching thread's for the workgroup order, every host thread's own for the lanes)
    void fill(std::vector<unsigned> &v, unsigned n, uint64_t &st) const
    {
        v.resize(n);
        for (unsigned i = 0; i < n; i++) v[i] = mode == 1 ? n - 1 - i : i;
        if (mode == 2)
            for (unsigned i = n; i > 1; i--) std::swap(v[i - 1], v[(unsigned)(next(st) % i)]);
    }
    void fill(std::vector<unsigned> &v, unsigned n) { fill(v, n, state); }
};
Order &order()
{
    static Order o;
    return o;
}

void run_block(unsigned nthreads)
{
    for (unsigned i = 0; i < nthreads; i++) prepare_lane((int)i);
    thread_local std::vector<unsigned> turn;
    thread_local uint64_t lane_rng = order().state ^ ((uint64_t)(uintptr_t)&turn * 0x9E3779B97F4A7C15ull) ^ 0x1234567ull;
    order().fill(turn, nthreads, lane_rng);
    for (;;) {
        bool ran = false;
        if (order().mode == 2) order().fill(turn, nthreads, lane_rng);
        for (unsigned k = 0; k < nthreads; k++) {
            const unsigned i = turn[k];
            if (g.lanes[i].st != RUN) continue;
            g.cur = (int)i;
            threadIdx_.x = i;
#ifdef SIMT_ASAN
            void *fake = nullptr;
            __sanitizer_start_switch_fiber(&fake, g.stacks + (size_t)i * kStack, kStack);
#endif
            simt_switch(&g.sched_sp, g.lanes[i].sp);
#ifdef SIMT_ASAN
            __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
            ran = true;
        }
        bool released = false, any_live = false, all_at_barrier = true;
        for (unsigned w0 = 0; w0 < nthreads; w0 += 64) released |= serve_wave((int)w0, (int)std::min<unsigned>(w0 + 64, nthreads));
        for (unsigned i = 0; i < nthreads; i++) {
            const Lane &L = g.lanes[i];
            if (L.st == DONE) continue;
            any_live = true;
            if (!(L.st == WAIT && L.kind == K_BLOCK_SYNC)) all_at_barrier = false;
        }
        if (!any_live) return;
        if (released) continue;
        if (all_at_barrier) { // __syncthreads: every live lane of the workgroup is there
            for (unsigned i = 0; i < nthreads; i++)
                if (g.lanes[i].st == WAIT) g.lanes[i].st = RUN;
            continue;
        }
        if (!ran) die("deadlock: no lane can run and no group can be served");
    }
}
} // namespace

uint64_t collective(int kind, uint64_t value, uint64_t arg)
{
    if (g.cur < 0) die("cross-lane operation outside a kernel");
    Lane &L = g.lanes[g.cur];
    L.kind = kind;
    L.value = value;
    L.arg = arg;
    L.site = __builtin_return_address(0);
    L.st = WAIT;
    to_scheduler(&L.sp, false);
    return g.lanes[g.cur].result;
}

namespace {
// the workgroups [next, grid) of one launch, drawn by whoever is free
void run_blocks(unsigned grid, unsigned block, const std::function<void()> &body, const std::vector<unsigned> &blocks, unsigned *next)
{
    if (g.cur >= 0) die("nested launch");
    if (g.nstacks < block) {
        if (g.stacks) munmap(g.stacks, g.nstacks * kStack);
        g.stacks = (char *)mmap(nullptr, (size_t)block * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g.stacks == (char *)MAP_FAILED) die("mmap of the lane stacks failed");
        g.nstacks = block;
    }
    g.lanes.assign(block, Lane());
    g.body = &body;
    const Idx keep_t = threadIdx_, keep_b = blockIdx_, keep_bd = blockDim_, keep_gd = gridDim_;
    blockDim_ = Idx{block, 1, 1};
    gridDim_ = Idx{grid, 1, 1};
    for (;;) {
        const unsigned k = __atomic_fetch_add(next, 1u, __ATOMIC_RELAXED);
        if (k >= grid) break;
        blockIdx_ = Idx{blocks[k], 0, 0};
        __atomic_fetch_add(&g_stats.blocks, 1, __ATOMIC_RELAXED);
        run_block(block);
    }
    g.cur = -1;
    g.body = nullptr;
    threadIdx_ = keep_t;
    blockIdx_ = keep_b;
    blockDim_ = keep_bd;
    gridDim_ = keep_gd;
}
} // namespace

// HB_SIMT_THREADS=N (default 1): the workgroups of a launch are shared out among N host threads, i.e. they really run at the same
// time - with the ThreadSanitizer build (`make tsan`) every pair of conflicting non-atomic accesses of two workgroups of one
// launch is reported as the data race it would be on the machine.  (The launch still returns only when all its workgroups are done.)
// The N - 1 helpers are started once and wait for work.
namespace {
struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    uint64_t generation = 0;
    unsigned busy = 0;
    // the current launch
    unsigned grid = 0, block = 0, next = 0;
    const std::function<void()> *body = nullptr;
    const std::vector<unsigned> *blocks = nullptr;

    void helper()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return generation != seen; });
                seen = generation;
            }
            run_blocks(grid, block, *body, *blocks, &next);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--busy == 0) cv_done.notify_all();
            }
        }
    }
    void run(unsigned helpers, unsigned g_, unsigned b_, const std::function<void()> &f, const std::vector<unsigned> &order_)
    {
        while (threads.size() < helpers) {
            threads.emplace_back([this] { helper(); });
            threads.back().detach();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            grid = g_;
            block = b_;
            next = 0;
            body = &f;
            blocks = &order_;
            busy = (unsigned)threads.size();
            generation++;
        }
        cv_work.notify_all();
        run_blocks(g_, b_, f, order_, &next);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return busy == 0; });
    }
};
Pool &pool()
{
    static Pool *p = new Pool(); // never destroyed: its detached helpers outlive main()
    return *p;
}
} // namespace

void launch(unsigned grid, unsigned block, const std::function<void()> &body)
{
    if (block == 0 || grid == 0) return;
    static const unsigned nthreads = [] {
        const char *e = std::getenv("HB_SIMT_THREADS");
        const long v = e ? std::atol(e) : 1;
        return (unsigned)(v < 1 ? 1 : v > 64 ? 64 : v);
    }();
    __atomic_fetch_add(&g_stats.launches, 1, __ATOMIC_RELAXED);
    std::vector<unsigned> blocks;
    order().fill(blocks, grid);
    if (nthreads == 1) {
        unsigned next = 0;
        run_blocks(grid, block, body, blocks, &next);
        return;
    }
    static std::mutex one_launch_at_a_time; // (host threads of the library may launch concurrently; the pool serves one launch)
    std::lock_guard<std::mutex> lk(one_launch_at_a_time);
    pool().run(nthreads - 1, grid, block, body, blocks);
}

Stats &stats() { return g_stats; }
} // namespace simt
