// tests/simt/include/hip/hip_runtime.h - TEST INFRASTRUCTURE (tests/simt/simt.h): the slice of the HIP language and runtime that
// stract_amd/csrc uses, mapped onto the SIMT interpreter and host memory, so that the UNCHANGED device sources compile with the
// host compiler (clang, -x c++).  Device memory = host memory (poisoned with 0xA5 when allocated), streams are synchronous,
// events are clock readings.  The one device it reports calls itself "gfx950 (SIMT interpreter, host)".
#pragma once
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../simt.h"

#define HB_SIMT_INTERPRETER 1

// ---- language ---------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...) // (__attribute__(()) is an empty attribute list)
#define __shared__ thread_local // one OS thread runs all lanes of a workgroup; (block-scope thread_local is static)
#define HIP_SYMBOL(x) x
#define threadIdx simt::threadIdx_
#define blockIdx simt::blockIdx_
#define blockDim simt::blockDim_
#define gridDim simt::gridDim_

struct uint2 {
    uint32_t x, y;
};
struct alignas(16) uint4 {
    uint32_t x, y, z, w;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- cross-lane operations -----------------------------------------------------------------------------------
template <class T>
static __forceinline__ uint64_t simt_bits(T v)
{
    static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits");
    uint64_t b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T>
static __forceinline__ T simt_from(uint64_t b)
{
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}
static __forceinline__ uint64_t __ballot(int pred) { return simt::collective(simt::K_BALLOT, pred ? 1 : 0, 0); }
template <class T>
static __forceinline__ T __shfl(T v, int src, int = 64) { return simt_from<T>(simt::collective(simt::K_SHFL, simt_bits(v), (uint64_t)(unsigned)src)); }
template <class T>
static __forceinline__ T __shfl_up(T v, unsigned d, int = 64) { return simt_from<T>(simt::collective(simt::K_SHFL_UP, simt_bits(v), d)); }
template <class T>
static __forceinline__ T __shfl_down(T v, unsigned d, int = 64) { return simt_from<T>(simt::collective(simt::K_SHFL_DOWN, simt_bits(v), d)); }
template <class T>
static __forceinline__ T __shfl_xor(T v, int m, int = 64) { return simt_from<T>(simt::collective(simt::K_SHFL_XOR, simt_bits(v), (uint64_t)(unsigned)m)); }
static __forceinline__ void __syncthreads() { (void)simt::collective(simt::K_BLOCK_SYNC, 0, 0); }
// quad_perm DPP only (ctrl < 0x100: lane q of a quad reads lane (ctrl >> 2q) & 3 of it; an inactive source reads as 0)
static __forceinline__ int simt_mov_dpp(int v, int ctrl) { return (int)(uint32_t)simt::collective(simt::K_DPP, (uint32_t)v, (uint64_t)(unsigned)ctrl); }
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) simt_mov_dpp((v), (ctrl))
// wavefront-scope fences / barriers: where the lanes of a wave hand LDS data to each other
#define __builtin_amdgcn_fence(order, scope) ((void)simt::collective(simt::K_WAVE_SYNC, 0, 0))
#define __builtin_amdgcn_wave_barrier() ((void)simt::collective(simt::K_WAVE_SYNC, 0, 0))
static __forceinline__ void __threadfence() {}
static __forceinline__ void __threadfence_system() {}

static __forceinline__ double __hiloint2double(int hi, int lo) { return simt_from<double>(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }
static __forceinline__ int __double2loint(double d) { return (int)(uint32_t)simt_bits(d); }
static __forceinline__ int __double2hiint(double d) { return (int)(uint32_t)(simt_bits(d) >> 32); }
static __forceinline__ long long __double_as_longlong(double d) { return (long long)simt_bits(d); }
static __forceinline__ double __longlong_as_double(long long v) { return simt_from<double>((uint64_t)v); }
// the device library's integer min / max
template <class T>
static __forceinline__ T max(T a, T b) { return a < b ? b : a; }
template <class T>
static __forceinline__ T min(T a, T b) { return b < a ? b : a; }
#define HB_DRAIN_VMEM() __asm__ __volatile__("" ::: "memory") // (hb_ingest.hip: s_waitcnt vmcnt(0) on the machine)
static __forceinline__ int __popc(unsigned v) { return __builtin_popcount(v); }
static __forceinline__ int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static __forceinline__ int __ffs(int v) { return __builtin_ffs(v); }
static __forceinline__ int __ffsll(long long v) { return __builtin_ffsll(v); }
static __forceinline__ int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static __forceinline__ int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

// ---- atomics ----------------------------------------------------------------------------------------------------------
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
// (__hip_atomic_load / _store / _fetch_* are clang builtins on the host too.)  One of them is a hand-over point: the pass-0
// kernel clears its LDS scratch counters and then max-accumulates into them with no fence in between - on the machine the
// wave's LDS instructions complete in order, here the first atomic of a lane must wait until every lane has done its
// clearing.  So a workgroup-scope fetch_max first lets the other lanes of the wave catch up (a yield, served like any
// other cross-lane operation: partial groups are fine).
template <class T, class U>
static __forceinline__ T simt_lds_fetch_max(T *p, U v)
{
    (void)simt::collective(simt::K_WAVE_SYNC, 0, 0);
    const T old = *p;
    if ((T)v > old) *p = (T)v;
    return old;
}
#define __hip_atomic_fetch_max(p, v, order, scope) simt_lds_fetch_max((p), (v))
// real (relaxed) atomics: the workgroups of a launch may run on several host threads (HB_SIMT_THREADS, simt_core.cpp)
template <class T, class U>
static __forceinline__ T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static __forceinline__ T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static __forceinline__ T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static __forceinline__ T atomicMax(T *p, U v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v > old && !__atomic_compare_exchange_n(p, &old, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
template <class T, class U>
static __forceinline__ T atomicMin(T *p, U v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v < old && !__atomic_compare_exchange_n(p, &old, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
template <class T, class U, class V>
static __forceinline__ T atomicCAS(T *p, U cmp, V v)
{
    T expected = (T)cmp;
    __atomic_compare_exchange_n(p, &expected, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected; // the value found
}
template <class T>
static __forceinline__ T __builtin_nontemporal_load_simt(const T *p) { return *p; }

// ---- runtime ------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotReady = 600,
    hipErrorNoDevice = 100,
};
typedef struct simt_stream *hipStream_t;
typedef struct simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s = nullptr);
hipError_t hipMemset(void *dst, int v, size_t n);
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s = nullptr);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
template <class T>
static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
template <class T>
static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned flags = 0) { return hipHostMalloc((void **)p, n, flags); }
template <class T>
static inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost)
{
    std::memcpy(dst, (const char *)&sym + off, n);
    return hipSuccess;
}
template <class T>
static inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice)
{
    std::memcpy((char *)&sym + off, src, n);
    return hipSuccess;
}

// kernel launch: every workgroup of the grid, one after the other, through the interpreter
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    simt::launch(dim3(grid).x, dim3(block).x, [&]() { kernel(__VA_ARGS__); })
