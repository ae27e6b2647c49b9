// rccl.h of tests/hip_emul -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h): the stand-in runs single-rank solves;
// the collective entry points exist so that the sources compile and fail loudly if a sharded solve reaches them.
#pragma once
#include "../hip/hip_runtime.h"
typedef struct hip_emul_nccl_comm* ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledError = 1 };
enum ncclDataType_t { ncclDouble = 8 };
enum ncclRedOp_t { ncclSum = 0 };
struct ncclUniqueId { char internal[128]; };
inline const char* ncclGetErrorString(ncclResult_t) { return "hip_emul: no collectives on the CPU stand-in"; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId*) { return ncclUnhandledError; }
inline ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int) { return ncclUnhandledError; }
inline ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
inline ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) { return ncclUnhandledError; }
