// tests/simt/include/rccl/rccl.h - TEST INFRASTRUCTURE (tests/simt/simt.h): RCCL's names for the host-interpreted build.  There is
// one interpreted device (contexts under the interpreter use logical ranks, hb_set_collectives, or a one-rank communicator).
#pragma once
#include <hip/hip_runtime.h>
typedef struct simt_comm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
    char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
// a communicator of ONE rank works (every collective is a copy onto itself), so that the library's RCCL call path - the
// calls, their order, the buffers they name - runs under the interpreter as it does in the one-rank GPU tests
static inline const char *ncclGetErrorString(ncclResult_t) { return "the SIMT interpreter has one device: only a one-rank communicator exists"; }
static inline size_t simt_nccl_size(ncclDataType_t t) { return t == ncclInt8 || t == ncclUint8 ? 1 : t == ncclFloat16 ? 2 : (t == ncclInt32 || t == ncclUint32 || t == ncclFloat32) ? 4 : 8; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof(*id));
    return ncclSuccess;
}
static inline ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank)
{
    if (nranks != 1 || rank != 0) return ncclInvalidUsage;
    *comm = (ncclComm_t)(uintptr_t)1;
    return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
static inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
static inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }
static inline ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t)
{
    if (send != recv) std::memmove(recv, send, count * simt_nccl_size(t));
    return ncclSuccess;
}
static inline ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm_t, hipStream_t)
{
    if (send != recv) std::memmove(recv, send, sendcount * simt_nccl_size(t));
    return ncclSuccess;
}
static inline ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t, hipStream_t)
{
    if (root != 0) return ncclInvalidArgument;
    if (send != recv) std::memmove(recv, send, count * simt_nccl_size(t));
    return ncclSuccess;
}
