// comm.h -- the communicator of cell-partitioned runs (included once, by engine.hip).
//
// The reference's only inter-process mechanism is Rmpi::mpi.applyLB over restarts (reference R/bayesian.R:262-263),
// with no communication while iterating.  A single factorisation whose cells are partitioned over the GPUs of a node
// needs ONE exchange per step: the sum over partitions of [sw | rowSums(eh) | scalars].  Two
// kinds of communicator carry it:
//   * RCCL (one process per GPU, xGMI): ncclAllReduce(sum, fp64) enqueued from C++ on a stream of the engine, so the
//     device-driven loop needs no host round trip and no Python between steps.  librccl is opened at run time
//     (dlopen "librccl.so.1"): the library still loads on a box without RCCL, and in a process that already carries
//     a copy (PyTorch ships one under the same SONAME) that copy is the one used.  VBNMF_RCCL_LIB names another
//     library exporting the same eight symbols (test infrastructure only: tests/fake_rccl).
//   * local group: the partition engines live in ONE process on ONE device (tests, single-GPU rehearsals of a
//     partitioned run: RCCL refuses two ranks on a device); the sum is a kernel (k_group_sum) in partition order.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "common.h"

struct vbnmf_engine;

namespace vbnmf {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

inline RcclApi &rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // VBNMF_RCCL_LIB=<path> names the library to open instead (tests/fake_rccl: a stand-in that accepts several ranks
        // on one device, so the multi-rank protocol can be rehearsed on a one-GPU box); it must open, there is no second try.
        const char *forced = getenv("VBNMF_RCCL_LIB");
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (forced && *forced) {
            api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *nm : names) {
                api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
                if (api.handle) break;
            }
        }
        if (!api.handle) { api.error = std::string(forced && *forced ? "VBNMF_RCCL_LIB could not be opened: " : "librccl could not be opened: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *nm) { void *p = dlsym(api.handle, nm); if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + nm; return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

}  // namespace vbnmf

struct vbnmf_comm {
    int kind = 0;                       // 0: RCCL, 1: local group
    int nranks = 1, rank = 0, device = 0;
    ncclComm_t nc = nullptr;            // kind 0
    // kind 1: the partition engines in attach order, and the device arrays of their send / receive pointers
    std::vector<vbnmf_engine *> members;
    const double **d_send_big = nullptr, **d_send_small = nullptr;
    double **d_recv_big = nullptr, **d_recv_small = nullptr;
    bool tables_ready = false;
};
