// hb_ampc.hip - GPU-resident shard of the AMPC counter table with HyperLogLog64Upsert semantics
// (include/hb_ampc.h cites the reference operations this serves).
#include "hb_guard_alloc.h" // FIRST: no-op unless built with -DHB_GUARD_ALLOC=<mode> (debug allocators: guard pages / poison / red zones)
#include "hb_pool.h"        // then: every hipMalloc / hipFree below goes through the caching device allocator (shipped build)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/hb_ampc.h"
#include "hb_regs.hip.h"
#include "hb_table.hip.h"

namespace {
thread_local std::string g_hbu_error;
using hbt::kEmpty;
using hbt::Table;
using hbt::u128;

__device__ __forceinline__ u128 make_key(const hb_u128 &v) { return ((u128)v.hi << 64) | (u128)v.lo; }

// ---- the key index lives on the device [r5] (rounds 2-4: a std::unordered_map on the host, one probe per pair on one core) ----
// key -> slot = hb_table.hip.h (the ingest's endpoint table: open addressing, one compare-and-swap per new key, slot ids in
// order of first arrival).  A batch: keys + values cross the link once; every pair finds or claims its slot; a STABLE radix
// sort of (slot, position) groups the pairs of one key in batch order; one quad per group applies them in that order.
__global__ __launch_bounds__(256) void slots_insert_kernel(const hb_u128 *keys, uint32_t count, Table t, uint32_t *slot)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) slot[i] = hbt::table_get(t, make_key(keys[i]), kEmpty);
}
// read-only: slots >= committed are keys of a batch that failed half-way (never visible)
__global__ __launch_bounds__(256) void slots_find_kernel(const hb_u128 *keys, uint32_t count, Table t, uint32_t committed, uint32_t *slot, uint8_t *found)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        uint32_t s = hbt::table_find(t, make_key(keys[i]));
        if (s >= committed) s = kEmpty;
        slot[i] = s;
        if (found) found[i] = s != kEmpty;
    }
}
__global__ __launch_bounds__(256) void table_clear_kernel(uint32_t *pids, uint64_t slots)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * 256) pids[i] = kEmpty;
}
// grow / repair: every entry whose slot id is below `keep` goes to the new table with its id
__global__ __launch_bounds__(256) void rehash_kernel(const u128 *old_keys, const uint32_t *old_pids, uint64_t old_slots, uint32_t keep, Table t)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < old_slots; i += (uint64_t)gridDim.x * 256) {
        const uint32_t p = old_pids[i];
        if (p < keep) (void)hbt::table_get(t, old_keys[i], p);
    }
}
// head of a group in the sorted order: the first pair of its slot
struct HeadFlag {
    const uint32_t *sorted;
    __host__ __device__ uint8_t operator()(uint32_t i) const { return (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0; }
};

// One quad per key group: its pairs (positions perm[begin .. end) of the batch, batch order kept) are applied in
// order to the stored counter - absent (fresh: slot >= first_new) keys take the first pair as is.  MODE 0 = upsert, 1 = set.
template <int MODE>
__global__ __launch_bounds__(256) void upsert_kernel(uint4 *table, const uint32_t *sorted_slot, const uint32_t *heads, const uint32_t *d_groups, uint32_t count,
                                                     uint32_t first_new, const uint32_t *perm, const uint4 *values, uint8_t *actions)
{
    const uint32_t groups = *d_groups;
    const int q = (int)(threadIdx.x & 3), qshift = (int)((threadIdx.x & 63) & ~3);
    const uint32_t stride = gridDim.x * 64;
    for (uint32_t g0 = blockIdx.x * 64; g0 < groups; g0 += stride) { // block-uniform trip count; every lane of a wave stays in
        const uint32_t gidx = g0 + (threadIdx.x >> 2);
        const bool valid = gidx < groups;
        uint32_t b = 0, e = 0, slot = 0;
        bool fresh = false;
        if (valid) {
            b = heads[gidx];
            e = gidx + 1 < groups ? heads[gidx + 1] : count;
            slot = sorted_slot[b];
            fresh = slot >= first_new;
        }
        uint4 cur = make_uint4(0, 0, 0, 0);
        if (valid && !fresh) cur = table[(uint64_t)slot * 4 + q];
        // the ballots below need every lane of the wave in the loop: iterate to the longest group of the wave
        uint32_t len = e - b, maxlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));
        for (uint32_t i = 0; i < maxlen; i++) {
            const bool act = valid && i < len;
            const uint32_t pos = act ? perm[b + i] : 0u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (act) v = values[(uint64_t)pos * 4 + q];
            hbk::Acc acc;
            hbk::acc_zero(acc);
            hbk::acc_merge(acc, cur);
            if (MODE == 0 && !(fresh && i == 0)) hbk::acc_merge(acc, v);
            const uint4 merged = (MODE == 1 || (fresh && i == 0)) ? v : hbk::acc_value(acc);
            const uint64_t bal = __ballot(act && hbk::u4_ne(merged, cur));
            const bool changed = ((bal >> qshift) & 0xFull) != 0;
            if (act) {
                if (MODE == 0 && q == 0) actions[pos] = (fresh && i == 0) ? HBU_INSERTED : (changed ? HBU_MERGED : HBU_NO_CHANGE);
                cur = merged;
            }
        }
        if (valid) table[(uint64_t)slot * 4 + q] = cur;
    }
}

__global__ __launch_bounds__(256) void get_kernel(const uint4 *table, const uint32_t *slots, uint32_t count, uint4 *out)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = t >> 2;
    if (i >= count) return;
    const uint32_t s = slots[i];
    out[t] = s == 0xFFFFFFFFu ? make_uint4(0, 0, 0, 0) : table[(uint64_t)s * 4 + (t & 3)];
}
} // namespace

struct hbu_table {
    int device = 0;
    hipStream_t stream = nullptr;
    // key index (device): open-addressing table of `slots` entries, load factor <= 1/2
    u128 *d_keys = nullptr;
    uint32_t *d_pids = nullptr;
    uint64_t slots = 0;
    unsigned long long *d_next = nullptr; // slot ids handed out so far (device counter of the index)
    unsigned long long *h_word = nullptr; // pinned: read-backs of the counter / the group count
    uint64_t committed = 0;               // keys visible to the caller (= d_next outside a failed batch)
    bool broken = false;                  // a failed batch could not be undone: every further call is refused
    uint4 *d_table = nullptr;
    uint64_t cap = 0; // counters allocated
    std::string err;
    // work memory of one batch, kept between calls (a mapper sends thousands of equally sized batches)
    void *d_work = nullptr;
    size_t work_bytes = 0;
};

namespace {
int fail(hbu_table *t, int code, const std::string &msg)
{
    (t ? t->err : g_hbu_error) = msg;
    return code;
}
#define HBU_HIP(call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(t, e_ == hipErrorOutOfMemory ? HB_ERR_NOMEM : HB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class F>
int guarded(hbu_table *t, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        try { return fail(t, HB_ERR_NOMEM, "out of host memory"); } catch (...) { return HB_ERR_NOMEM; }
    } catch (...) {
        try { return fail(t, HB_ERR_INVALID, "unexpected C++ exception"); } catch (...) { return HB_ERR_INVALID; }
    }
}

unsigned grid_for(uint64_t items) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>((items + 255) / 256, 1), 1u << 16); }
Table table_of(const hbu_table *t) { return Table{t->d_keys, t->d_pids, t->slots - 1, t->d_next}; }

// counters for `need` keys
int reserve(hbu_table *t, uint64_t need)
{
    if (need <= t->cap) return HB_OK;
    const uint64_t cap = std::max<uint64_t>(need, std::max<uint64_t>(2 * t->cap, 1024));
    uint4 *n = nullptr;
    if (hipMalloc((void **)&n, cap * 64) != hipSuccess) return fail(t, HB_ERR_NOMEM, "hipMalloc(counter table) failed");
    hipError_t e = hipMemsetAsync(n, 0, cap * 64, t->stream);
    if (e == hipSuccess && t->cap) e = hipMemcpyAsync(n, t->d_table, t->cap * 64, hipMemcpyDeviceToDevice, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) {
        (void)hipFree(n);
        return fail(t, HB_ERR_HIP, hipGetErrorString(e));
    }
    if (t->d_table) (void)hipFree(t->d_table);
    t->d_table = n;
    t->cap = cap;
    return HB_OK;
}

// a key index with room for `keys` keys at load factor <= 1/2, holding the entries with ids < keep of the present one
int rebuild_index(hbu_table *t, uint64_t keys, uint64_t keep)
{
    uint64_t slots = 1024;
    while (slots < 2 * keys) slots <<= 1;
    u128 *nk = nullptr;
    uint32_t *np = nullptr;
    if (hipMalloc((void **)&nk, slots * sizeof(u128)) != hipSuccess || hipMalloc((void **)&np, slots * sizeof(uint32_t)) != hipSuccess) {
        if (nk) (void)hipFree(nk);
        (void)hipGetLastError();
        return fail(t, HB_ERR_NOMEM, "hipMalloc(key index) failed");
    }
    hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for(slots)), dim3(256), 0, t->stream, np, slots);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && t->slots && keep) {
        const Table nt{nk, np, slots - 1, t->d_next};
        hipLaunchKernelGGL(rehash_kernel, dim3(grid_for(t->slots)), dim3(256), 0, t->stream, (const u128 *)t->d_keys, (const uint32_t *)t->d_pids, t->slots,
                           (uint32_t)keep, nt);
        e = hipGetLastError();
    }
    const unsigned long long next = keep;
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_next, &next, sizeof(next), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) {
        (void)hipFree(nk);
        (void)hipFree(np);
        return fail(t, HB_ERR_HIP, hipGetErrorString(e));
    }
    if (t->d_keys) (void)hipFree(t->d_keys);
    if (t->d_pids) (void)hipFree(t->d_pids);
    t->d_keys = nk;
    t->d_pids = np;
    t->slots = slots;
    return HB_OK;
}

int work_memory(hbu_table *t, size_t bytes)
{
    if (bytes <= t->work_bytes) return HB_OK;
    if (t->d_work) (void)hipFree(t->d_work);
    t->d_work = nullptr;
    t->work_bytes = 0;
    if (hipMalloc(&t->d_work, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(t, HB_ERR_NOMEM, "hipMalloc(batch work memory) failed");
    }
    t->work_bytes = bytes;
    return HB_OK;
}
struct Carve { // consecutive 256-byte aligned pieces of the work buffer
    char *p;
    size_t used = 0;
    template <class T>
    T *take(size_t count)
    {
        T *r = (T *)((uintptr_t)p + used); // (integer arithmetic: the sizing pass carves from a null base)
        used += (count * sizeof(T) + 255) & ~(size_t)255;
        return r;
    }
};

// shared body of batch_set / batch_upsert
int apply(hbu_table *t, const hb_u128 *keys, const uint8_t *counters, uint64_t count, uint8_t *actions, bool upsert)
{
    if (!t || (count && (!keys || !counters)) || (upsert && count && !actions)) return t ? fail(t, HB_ERR_INVALID, "NULL argument") : HB_ERR_INVALID;
    if (t->broken) return fail(t, HB_ERR_INVALID, "the table is unusable: an earlier failed batch could not be undone");
    // the kernels index 4 threads per pair / group with 32-bit thread ids
    if (count >= (1ull << 30)) return fail(t, HB_ERR_LIMIT, "batch too large (< 2^30 pairs per call)");
    if (!count) return HB_OK;
    HBU_HIP(hipSetDevice(t->device));
    if (t->committed + count >= 0xFFFFFFFEull) return fail(t, HB_ERR_LIMIT, "too many keys in one table (< 2^32)");
    // ---- everything that can fail for lack of memory comes BEFORE the index changes: room for count new keys (every pair might
    // bring one), their counters, the batch's work memory
    int rc;
    // (slots = the power of two >= 2 x keys: the rounding is what makes repeated growth geometric)
    if (2 * (t->committed + count) > t->slots && (rc = rebuild_index(t, t->committed + count, t->committed))) return rc;
    if ((rc = reserve(t, t->committed + count))) return rc;
    size_t sort_bytes = 0, select_bytes = 0;
    const uint32_t n32 = (uint32_t)count;
    {
        uint32_t *nul = nullptr;
        auto iota = rocprim::make_counting_iterator<uint32_t>(0);
        HBU_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t *)nul, nul, iota, nul, (size_t)count, 0, 32, t->stream));
        auto flags = rocprim::make_transform_iterator(iota, HeadFlag{nul});
        HBU_HIP(rocprim::select(nullptr, select_bytes, iota, flags, nul, nul, (size_t)count, t->stream));
    }
    const size_t tmp_bytes = std::max(sort_bytes, select_bytes);
    auto layout = [&](Carve &c, hb_u128 *&dk, uint4 *&dv, uint32_t *&slot, uint32_t *&slot_s, uint32_t *&perm, uint32_t *&heads, uint32_t *&groups,
                      uint8_t *&act, char *&tmp) {
        dk = c.take<hb_u128>(count);
        dv = c.take<uint4>(count * 4);
        slot = c.take<uint32_t>(count);
        slot_s = c.take<uint32_t>(count);
        perm = c.take<uint32_t>(count);
        heads = c.take<uint32_t>(count);
        groups = c.take<uint32_t>(2);
        act = c.take<uint8_t>(count);
        tmp = c.take<char>(tmp_bytes);
    };
    hb_u128 *d_k;
    uint4 *d_v;
    uint32_t *d_slot, *d_slot_s, *d_perm, *d_heads, *d_groups;
    uint8_t *d_act;
    char *d_tmp;
    {
        Carve probe{nullptr};
        layout(probe, d_k, d_v, d_slot, d_slot_s, d_perm, d_heads, d_groups, d_act, d_tmp);
        if ((rc = work_memory(t, probe.used))) return rc;
    }
    Carve carve{(char *)t->d_work};
    layout(carve, d_k, d_v, d_slot, d_slot_s, d_perm, d_heads, d_groups, d_act, d_tmp);
    // ---- from here on a failure is a HIP error; the keys this batch may have put into the index are taken out again
    // (the index is rebuilt from the entries below `committed`) before the error is returned: the batch is transactional
    const uint32_t first_new = (uint32_t)t->committed;
    auto run = [&]() -> hipError_t {
        hipError_t e = hipMemcpyAsync(d_k, keys, count * sizeof(hb_u128), hipMemcpyHostToDevice, t->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_v, counters, count * 64, hipMemcpyHostToDevice, t->stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(slots_insert_kernel, dim3(grid_for(count)), dim3(256), 0, t->stream, (const hb_u128 *)d_k, n32, table_of(t), d_slot);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        auto iota = rocprim::make_counting_iterator<uint32_t>(0);
        size_t b = tmp_bytes;
        if ((e = rocprim::radix_sort_pairs(d_tmp, b, (const uint32_t *)d_slot, d_slot_s, iota, d_perm, (size_t)count, 0, 32, t->stream)) != hipSuccess) return e;
        auto flags = rocprim::make_transform_iterator(iota, HeadFlag{d_slot_s});
        b = tmp_bytes;
        if ((e = rocprim::select(d_tmp, b, iota, flags, d_heads, d_groups, (size_t)count, t->stream)) != hipSuccess) return e;
        const unsigned blocks = (unsigned)std::min<uint64_t>((count + 63) / 64, 1u << 16);
        if (upsert)
            hipLaunchKernelGGL(upsert_kernel<0>, dim3(blocks), dim3(256), 0, t->stream, t->d_table, (const uint32_t *)d_slot_s, (const uint32_t *)d_heads,
                               (const uint32_t *)d_groups, n32, first_new, (const uint32_t *)d_perm, (const uint4 *)d_v, d_act);
        else
            hipLaunchKernelGGL(upsert_kernel<1>, dim3(blocks), dim3(256), 0, t->stream, t->d_table, (const uint32_t *)d_slot_s, (const uint32_t *)d_heads,
                               (const uint32_t *)d_groups, n32, first_new, (const uint32_t *)d_perm, (const uint4 *)d_v, d_act);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (upsert && (e = hipMemcpyAsync(actions, d_act, count, hipMemcpyDeviceToHost, t->stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(t->h_word, t->d_next, sizeof(unsigned long long), hipMemcpyDeviceToHost, t->stream)) != hipSuccess) return e;
        return hipStreamSynchronize(t->stream);
    };
    const hipError_t e = run();
    if (e != hipSuccess) {
        const std::string why = hipGetErrorString(e);
        (void)hipGetLastError();
        if (rebuild_index(t, std::max<uint64_t>(t->slots / 2, 512), t->committed)) t->broken = true;
        return fail(t, e == hipErrorOutOfMemory ? HB_ERR_NOMEM : HB_ERR_HIP, why);
    }
    t->committed = *t->h_word;
    return HB_OK;
}
} // namespace

extern "C" {

const char *hbu_last_error(const hbu_table *t) { return t ? t->err.c_str() : g_hbu_error.c_str(); }

int hbu_create(int32_t device, uint64_t capacity_hint, hbu_table **out)
{
    return guarded(nullptr, [&]() -> int {
        hbu_table *t = nullptr;
        if (!out) return fail(t, HB_ERR_INVALID, "out == NULL");
        *out = nullptr;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(t, HB_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
        int dev = device;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (dev >= ndev) return fail(t, HB_ERR_INVALID, "device ordinal out of range");
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(t, HB_ERR_NO_DEVICE, "kernels are built for gfx950 only");
        hbu_table *tab = new hbu_table();
        tab->device = dev;
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&tab->stream, hipStreamNonBlocking) != hipSuccess) {
            delete tab;
            return fail(t, HB_ERR_HIP, "stream creation failed");
        }
        int rc = HB_OK;
        if (hipMalloc((void **)&tab->d_next, 2 * sizeof(unsigned long long)) != hipSuccess || hipHostMalloc((void **)&tab->h_word, 2 * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(tab, HB_ERR_NOMEM, "allocation of the index counter failed");
        }
        if (!rc) rc = rebuild_index(tab, std::max<uint64_t>(capacity_hint, 1), 0);
        if (!rc) rc = reserve(tab, std::max<uint64_t>(capacity_hint, 1));
        if (rc) {
            g_hbu_error = tab->err;
            hbu_destroy(tab);
            return rc;
        }
        *out = tab;
        return HB_OK;
    });
}

void hbu_destroy(hbu_table *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    for (void *p : {(void *)t->d_table, (void *)t->d_keys, (void *)t->d_pids, (void *)t->d_next, t->d_work})
        if (p) (void)hipFree(p);
    if (t->h_word) (void)hipHostFree(t->h_word);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

int hbu_len(const hbu_table *t, uint64_t *keys)
{
    if (!t || !keys) return HB_ERR_INVALID;
    *keys = t->committed;
    return HB_OK;
}

int hbu_batch_set(hbu_table *t, const hb_u128 *keys, const uint8_t *counters, uint64_t count)
{
    return guarded(t, [&]() -> int { return apply(t, keys, counters, count, nullptr, false); });
}

int hbu_batch_upsert(hbu_table *t, const hb_u128 *keys, const uint8_t *counters, uint64_t count, uint8_t *actions)
{
    return guarded(t, [&]() -> int { return apply(t, keys, counters, count, actions, true); });
}

int hbu_batch_get(hbu_table *t, const hb_u128 *keys, uint64_t count, uint8_t *counters_out, uint8_t *found)
{
    return guarded(t, [&]() -> int {
        if (!t || (count && (!keys || !counters_out))) return t ? fail(t, HB_ERR_INVALID, "NULL argument") : HB_ERR_INVALID;
        if (count >= (1ull << 30)) return fail(t, HB_ERR_LIMIT, "batch too large (< 2^30 keys per call)"); // 4 threads per key, 32-bit ids
        if (!count) return HB_OK;
        if (t->broken) return fail(t, HB_ERR_INVALID, "the table is unusable: an earlier failed batch could not be undone");
        HBU_HIP(hipSetDevice(t->device));
        Carve probe{nullptr};
        (void)probe.take<hb_u128>(count);
        (void)probe.take<uint32_t>(count);
        (void)probe.take<uint4>(count * 4);
        (void)probe.take<uint8_t>(count);
        int rc = work_memory(t, probe.used);
        if (rc) return rc;
        Carve carve{(char *)t->d_work};
        hb_u128 *d_k = carve.take<hb_u128>(count);
        uint32_t *d_slots = carve.take<uint32_t>(count);
        uint4 *d_out = carve.take<uint4>(count * 4);
        uint8_t *d_found = carve.take<uint8_t>(count);
        HBU_HIP(hipMemcpyAsync(d_k, keys, count * sizeof(hb_u128), hipMemcpyHostToDevice, t->stream));
        hipLaunchKernelGGL(slots_find_kernel, dim3(grid_for(count)), dim3(256), 0, t->stream, (const hb_u128 *)d_k, (uint32_t)count, table_of(t),
                           (uint32_t)t->committed, d_slots, d_found);
        hipLaunchKernelGGL(get_kernel, dim3((unsigned)((count * 4 + 255) / 256)), dim3(256), 0, t->stream, (const uint4 *)t->d_table,
                           (const uint32_t *)d_slots, (uint32_t)count, d_out);
        HBU_HIP(hipGetLastError());
        HBU_HIP(hipMemcpyAsync(counters_out, d_out, count * 64, hipMemcpyDeviceToHost, t->stream));
        if (found) HBU_HIP(hipMemcpyAsync(found, d_found, count, hipMemcpyDeviceToHost, t->stream));
        HBU_HIP(hipStreamSynchronize(t->stream));
        return HB_OK;
    });
}

} // extern "C"
