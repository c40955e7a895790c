// Run-time binding of RCCL for the gather of sharded results: libsosrt.so has no link-time dependency on
// RCCL; the library is resolved when sosrt_comm_init is first called -- the copy already in the process if there is one
// (PyTorch ships its own librccl.so), the ROCm one otherwise.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace sosrt {

struct Rccl {
    // subset of rccl.h (ROCm 7.2: ncclResult_t, ncclComm_t, ncclUniqueId by value, ncclDataType_t ncclFloat64 = 8)
    typedef struct ncclComm* comm_t;
    struct UniqueId { char internal[128]; };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    // nullptr on success, else a message
    const char* load();
};
Rccl& rccl();

}  // namespace sosrt
