// hb_ampc.hip - GPU-resident shard of the AMPC counter table with HyperLogLog64Upsert semantics
// (include/hb_ampc.h cites the reference operations this serves).
#include "hb_guard_alloc.h" // FIRST: no-op unless built with -DHB_GUARD_ALLOC=<mode> (debug allocators: guard pages / poison / red zones)
#include "hb_pool.h"        // then: every hipMalloc / hipFree below goes through the caching device allocator (shipped build)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hb_ampc.h"
#include "hb_regs.hip.h"

namespace {
thread_local std::string g_hbu_error;

struct KeyHash {
    size_t operator()(const hb_u128 &k) const
    {
        uint64_t x = k.lo ^ (k.hi * 0x9E3779B97F4A7C15ull);
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdull;
        x ^= x >> 33;
        return (size_t)x;
    }
};
struct KeyEq {
    bool operator()(const hb_u128 &a, const hb_u128 &b) const { return a.lo == b.lo && a.hi == b.hi; }
};

// One quad per key group: its pairs (positions perm[begin .. end) of the batch, batch order kept) are applied in
// order to the stored counter - absent (fresh) keys take the first pair as is.  MODE 0 = upsert, 1 = set.
template <int MODE>
__global__ __launch_bounds__(256) void upsert_kernel(uint4 *table, const uint32_t *group_slot, const uint32_t *group_begin,
                                                     const uint8_t *group_fresh, uint32_t groups, const uint32_t *perm,
                                                     const uint4 *values, uint8_t *actions)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t gidx = t >> 2;
    const int q = (int)(t & 3), qshift = (int)((threadIdx.x & 63) & ~3);
    const bool valid = gidx < groups;
    uint32_t b = 0, e = 0, slot = 0;
    bool fresh = false;
    if (valid) {
        b = group_begin[gidx];
        e = group_begin[gidx + 1];
        slot = group_slot[gidx];
        fresh = group_fresh[gidx] != 0;
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
    std::unordered_map<hb_u128, uint32_t, KeyHash, KeyEq> slot_of;
    uint4 *d_table = nullptr;
    uint64_t cap = 0; // counters allocated
    std::string err;
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
        if (e_ != hipSuccess) return fail(t, HB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
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

// shared body of batch_set / batch_upsert
int apply(hbu_table *t, const hb_u128 *keys, const uint8_t *counters, uint64_t count, uint8_t *actions, bool upsert)
{
    if (!t || (count && (!keys || !counters)) || (upsert && count && !actions)) return t ? fail(t, HB_ERR_INVALID, "NULL argument") : HB_ERR_INVALID;
    // the kernels index 4 threads per pair / group with 32-bit thread ids
    if (count >= (1ull << 30)) return fail(t, HB_ERR_LIMIT, "batch too large (< 2^30 pairs per call)");
    if (!count) return HB_OK;
    HBU_HIP(hipSetDevice(t->device));
    // key -> slot (new keys get the next slots), WITHOUT touching the table's key map yet: the device table is grown
    // first, so that a failed allocation cannot leave keys registered whose slots were never written or allocated
    std::vector<uint32_t> slot(count), perm(count);
    const uint32_t first_new = (uint32_t)t->slot_of.size();
    std::unordered_map<hb_u128, uint32_t, KeyHash, KeyEq> fresh;
    for (uint64_t i = 0; i < count; i++) {
        auto it = t->slot_of.find(keys[i]);
        if (it != t->slot_of.end()) {
            slot[i] = it->second;
            continue;
        }
        if ((uint64_t)first_new + fresh.size() >= 0xFFFFFFFEull) return fail(t, HB_ERR_LIMIT, "too many keys in one table (< 2^32)");
        slot[i] = fresh.emplace(keys[i], first_new + (uint32_t)fresh.size()).first->second;
    }
    int rc = reserve(t, (uint64_t)first_new + fresh.size());
    if (rc) return rc;
    // the map update is transactional: whatever fails below (host or device allocation, copies, the launch), the
    // keys this batch introduced are forgotten again (their slots are >= first_new and nothing else refers to them)
    struct Rollback {
        hbu_table *t;
        const std::unordered_map<hb_u128, uint32_t, KeyHash, KeyEq> *added;
        bool armed = true;
        ~Rollback()
        {
            if (armed)
                for (const auto &kv : *added) t->slot_of.erase(kv.first);
        }
    } rollback{t, &fresh};
    for (const auto &kv : fresh) t->slot_of.emplace(kv.first, kv.second);
    std::iota(perm.begin(), perm.end(), 0u);
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return slot[a] < slot[b]; });
    std::vector<uint32_t> gslot, gbegin;
    std::vector<uint8_t> gfresh;
    for (uint64_t i = 0; i < count; i++) {
        const uint32_t s = slot[perm[i]];
        if (gslot.empty() || gslot.back() != s) {
            gslot.push_back(s);
            gbegin.push_back((uint32_t)i);
            gfresh.push_back(s >= first_new ? 1 : 0);
        }
    }
    gbegin.push_back((uint32_t)count);
    const uint32_t groups = (uint32_t)gslot.size();
    uint32_t *d_gslot = nullptr, *d_gbegin = nullptr, *d_perm = nullptr;
    uint8_t *d_gfresh = nullptr, *d_actions = nullptr;
    uint4 *d_values = nullptr;
    auto cleanup = [&]() {
        for (void *p : {(void *)d_gslot, (void *)d_gbegin, (void *)d_perm, (void *)d_gfresh, (void *)d_actions, (void *)d_values})
            if (p) (void)hipFree(p);
    };
    hipError_t e = hipMalloc((void **)&d_gslot, groups * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_gbegin, (groups + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_perm, count * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_gfresh, groups);
    if (e == hipSuccess) e = hipMalloc((void **)&d_actions, count);
    if (e == hipSuccess) e = hipMalloc((void **)&d_values, count * 64);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gslot, gslot.data(), groups * 4, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gbegin, gbegin.data(), (groups + 1) * 4, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_perm, perm.data(), count * 4, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gfresh, gfresh.data(), groups, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_values, counters, count * 64, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)(((uint64_t)groups * 4 + 255) / 256);
        if (upsert)
            hipLaunchKernelGGL(upsert_kernel<0>, dim3(blocks), dim3(256), 0, t->stream, t->d_table, (const uint32_t *)d_gslot, (const uint32_t *)d_gbegin,
                               (const uint8_t *)d_gfresh, groups, (const uint32_t *)d_perm, (const uint4 *)d_values, d_actions);
        else
            hipLaunchKernelGGL(upsert_kernel<1>, dim3(blocks), dim3(256), 0, t->stream, t->d_table, (const uint32_t *)d_gslot, (const uint32_t *)d_gbegin,
                               (const uint8_t *)d_gfresh, groups, (const uint32_t *)d_perm, (const uint4 *)d_values, d_actions);
        e = hipGetLastError();
    }
    if (e == hipSuccess && upsert) e = hipMemcpyAsync(actions, d_actions, count, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    cleanup();
    if (e != hipSuccess) return fail(t, e == hipErrorOutOfMemory ? HB_ERR_NOMEM : HB_ERR_HIP, hipGetErrorString(e));
    rollback.armed = false;
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
        tab->slot_of.reserve((size_t)capacity_hint);
        int rc = reserve(tab, std::max<uint64_t>(capacity_hint, 1));
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
    if (t->d_table) (void)hipFree(t->d_table);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

int hbu_len(const hbu_table *t, uint64_t *keys)
{
    if (!t || !keys) return HB_ERR_INVALID;
    *keys = t->slot_of.size();
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
        HBU_HIP(hipSetDevice(t->device));
        std::vector<uint32_t> slots(count);
        for (uint64_t i = 0; i < count; i++) {
            auto it = t->slot_of.find(keys[i]);
            slots[i] = it == t->slot_of.end() ? 0xFFFFFFFFu : it->second;
            if (found) found[i] = it == t->slot_of.end() ? 0 : 1;
        }
        uint32_t *d_slots = nullptr;
        uint4 *d_out = nullptr;
        hipError_t e = hipMalloc((void **)&d_slots, count * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, count * 64);
        if (e == hipSuccess) e = hipMemcpyAsync(d_slots, slots.data(), count * 4, hipMemcpyHostToDevice, t->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(get_kernel, dim3((unsigned)((count * 4 + 255) / 256)), dim3(256), 0, t->stream, (const uint4 *)t->d_table,
                               (const uint32_t *)d_slots, (uint32_t)count, d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(counters_out, d_out, count * 64, hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
        if (d_slots) (void)hipFree(d_slots);
        if (d_out) (void)hipFree(d_out);
        if (e != hipSuccess) return fail(t, HB_ERR_HIP, hipGetErrorString(e));
        return HB_OK;
    });
}

} // extern "C"
