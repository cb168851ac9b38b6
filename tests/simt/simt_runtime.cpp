// tests/simt/simt_runtime.cpp - TEST INFRASTRUCTURE (tests/simt/simt.h): the runtime half of the fake <hip/hip_runtime.h>.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

namespace {
std::mutex g_mu;
std::map<void *, size_t> g_live;
size_t g_in_use = 0;
const size_t kTotal = (size_t)64 << 30;
double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
} // namespace
struct simt_stream {
    int id;
};
struct simt_event {
    double t_ms;
};

hipError_t hipMalloc(void **p, size_t n)
{
    if (!p) return hipErrorInvalidValue;
    const size_t bytes = (n ? n : 1);
    void *q = nullptr;
#if defined(__has_feature) && __has_feature(address_sanitizer)
    // exactly the bytes asked for: the sanitizer's red zone starts right behind the last one (64-byte aligned like a counter row)
    if (posix_memalign(&q, 64, bytes) != 0 || !q) {
#else
    if (posix_memalign(&q, 256, (bytes + 255) / 256 * 256) != 0 || !q) {
#endif
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    if (!std::getenv("HB_SIMT_NO_POISON")) std::memset(q, 0xA5, bytes); // fresh device memory holds anything
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[q] = bytes;
    g_in_use += bytes;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return hipErrorInvalidValue;
    g_in_use -= it->second;
    g_live.erase(it);
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    *p = std::malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p)
{
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b)
{
    std::lock_guard<std::mutex> lk(g_mu);
    *total_b = kTotal;
    *free_b = kTotal > g_in_use ? kTotal - g_in_use : 0;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d)
{
    *d = 0;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d)
{
    if (d != 0) return hipErrorInvalidValue;
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "SIMT interpreter (host)");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950 (SIMT interpreter, host)");
    const char *e = std::getenv("HB_SIMT_CUS");
    p->multiProcessorCount = e ? std::atoi(e) : 2; // few compute units = few workgroups per launch
    p->totalGlobalMem = kTotal;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "error (SIMT interpreter)"; }
hipError_t hipStreamCreate(hipStream_t *s)
{
    *s = new simt_stream{0};
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = new simt_event{0.0};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    e->t_ms = now_ms();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = (float)(b->t_ms - a->t_ms);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
    if (n) std::memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, n, k); }
hipError_t hipMemset(void *dst, int v, size_t n)
{
    if (n) std::memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { return hipMemset(dst, v, n); }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
