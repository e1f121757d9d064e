// RCCL behind the C ABI: the two collectives of the K-sharded GGL iteration on the ctx's own stream.
// librccl is resolved at run time (dlopen of the SONAME librccl.so.1, which is also what PyTorch-ROCm ships: whichever
// copy the process has loaded serves both), so libggl_hip.so has no link-time dependency on it and single-GPU users
// never touch it.  Only what is needed of rccl.h is declared here (ABI of RCCL 2.x: rccl.h:40-43,187,220,260,339,448,467,611).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ggl {

struct RcclApi {
    static constexpr int UNIQUE_ID_BYTES = 128;
    struct UniqueId { char internal[UNIQUE_ID_BYTES]; };
    typedef void* Comm;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;      // optional (ggl_comm_count)
    int (*AllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static constexpr int Float64 = 8, Sum = 0;
    const char* load_error = nullptr;
};

// the process-wide table (loaded on first use); nullptr with *err set if librccl cannot be loaded
const RcclApi* rccl_api(const char** err);

}  // namespace ggl
