// occ_comm.hpp -- RCCL, called directly (no PyTorch): the one-shot broadcast of the fixed problem arrays to the other
// GPUs of a group (chains are independent, so set-up is the only communication) and the few host-side
// collectives a multi-process launch needs (barrier, max over ranks, small metadata).  librccl is opened at run time
// (dlopen), so the engine itself does not depend on it: a single-GPU user never loads it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <string>

namespace occ {

struct RcclApi {
    void *lib = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok() const { return lib != nullptr; }
};

inline RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        // OCC_RCCL_LIB: a library to try first; the word `none` tries nothing at all (the branch a machine without
        // librccl takes -- occ_create_group then hands over by hipMemcpyPeer, init_comm by the file rendezvous)
        const char *first = std::getenv("OCC_RCCL_LIB");
        if (first && std::string(first) == "none") {
            a.err = "librccl could not be opened: OCC_RCCL_LIB=none";
            return a;
        }
        const char *names[] = {first, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string why;
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            a.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            const char *e = dlerror();  // (one call: it returns the message AND clears it)
            why += std::string(why.empty() ? "" : "; ") + nm + ": " + (e ? e : "not found");
        }
        if (!a.lib) {
            a.err = "librccl could not be opened: " + (why.empty() ? std::string("not found") : why);
            return a;
        }
#define OCC_RCCL_SYM(field, name)                                                           \
        a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, #name));                 \
        if (!a.field) { a.err = "librccl lacks " #name; dlclose(a.lib); a.lib = nullptr; return a; }
        OCC_RCCL_SYM(GetUniqueId, ncclGetUniqueId)
        OCC_RCCL_SYM(CommInitRank, ncclCommInitRank)
        OCC_RCCL_SYM(CommInitAll, ncclCommInitAll)
        OCC_RCCL_SYM(CommDestroy, ncclCommDestroy)
        OCC_RCCL_SYM(Broadcast, ncclBroadcast)
        OCC_RCCL_SYM(AllReduce, ncclAllReduce)
        OCC_RCCL_SYM(GroupStart, ncclGroupStart)
        OCC_RCCL_SYM(GroupEnd, ncclGroupEnd)
        OCC_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef OCC_RCCL_SYM
        return a;
    }();
    return api;
}

}  // namespace occ

// One rank of a multi-process group (one process per GPU).
struct occ_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    void *stage = nullptr;  // device staging buffer for host-side collectives
    size_t stage_bytes = 0;
    std::string err;
};
