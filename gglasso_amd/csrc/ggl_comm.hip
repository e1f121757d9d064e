#include "ggl_comm.hpp"

#include <dlfcn.h>

#include <mutex>

namespace ggl {

const RcclApi* rccl_api(const char** err)
{
    static RcclApi api;
    static bool tried = false, ok = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!tried) {
        tried = true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            api.load_error = "librccl.so.1 could not be loaded (dlopen)";
        } else {
            api.GetUniqueId = (int (*)(RcclApi::UniqueId*))dlsym(h, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(RcclApi::Comm*, int, RcclApi::UniqueId, int))dlsym(h, "ncclCommInitRank");
            api.CommDestroy = (int (*)(RcclApi::Comm))dlsym(h, "ncclCommDestroy");
            api.AllReduce = (int (*)(const void*, void*, size_t, int, int, RcclApi::Comm, hipStream_t))dlsym(h, "ncclAllReduce");
            api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
            api.CommCount = (int (*)(RcclApi::Comm, int*))dlsym(h, "ncclCommCount");
            ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
            if (!ok) api.load_error = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
        }
    }
    if (!ok) {
        if (err) *err = api.load_error;
        return nullptr;
    }
    return &api;
}

}  // namespace ggl
