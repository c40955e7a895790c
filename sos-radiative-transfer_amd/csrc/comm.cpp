#include "comm.hpp"

#include <dlfcn.h>

namespace sosrt {

Rccl& rccl() {
    static Rccl r;
    return r;
}

const char* Rccl::load() {
    if (handle) return nullptr;
    void* h = nullptr;
    // the copy the process already uses (PyTorch's, or one the caller linked), then the system one
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    }
    if (!h) return "librccl.so not found (dlopen)";
    auto sym = [&](const char* n) { return dlsym(h, n); };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString)
        return "librccl.so lacks a symbol of the nccl API";
    handle = h;
    return nullptr;
}

}  // namespace sosrt
