// tests/emu: an in-process stand-in for the handful of RCCL calls libnsparse_dist makes -- ranks are THREADS of one
// process (ncclCommInitAll: one thread per fake device, as samples/amb_dist.cpp runs them; ncclCommInitRank: ranks that
// share a unique id), a collective is a rendezvous of those threads plus memcpy.  Test infrastructure: it lets the
// native multi-rank control flow (partition, per-rank conversion, all-gather in place / staged + gap closing, the
// broadcasts of the SpGEMM gather) execute at world > 1 on a box without GPUs.  Nothing about xGMI is modelled.
#pragma once
#include <hip/hip_runtime.h>
typedef enum {
    ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
    ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7
} ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
struct ncclComm;
typedef ncclComm *ncclComm_t;
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *async_error);
const char *ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st);
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t st);
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t st);
}
