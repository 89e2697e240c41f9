// oval_comm.cpp — the one collective of the pattern path in the C ABI: the sum
// of the batch counters over the GPUs of a node, RCCL over xGMI.  Objects shard
// by worker key range (internal/runner/runner.go:117-124), so nothing else
// ever crosses GPUs.
//
// librccl is opened at first use, not linked: a single-GPU caller never needs
// it, and the library must load where RCCL is not installed.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/oval_hip.h"
#include "oval_capi_internal.h"
#include "oval_kernels.h"

struct oval_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0, device = 0;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // optional: oval_comm_info
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
    char why[256] = {0};
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            std::snprintf(r.why, sizeof r.why, "librccl.so.1 could not be opened: %s", dlerror());
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.handle, "ncclCommUserRank"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
        if (!r.ok) std::snprintf(r.why, sizeof r.why, "librccl lacks one of the five entry points used here");
    });
    return r;
}

int rccl_fail(const char* what, ncclResult_t e)
{
    return oval_internal_fail(OVAL_E_HIP, "%s: %s", what, rccl().GetErrorString(e));
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == OVAL_COMM_ID_BYTES, "the ABI carries the RCCL unique ID as 128 opaque bytes");

extern "C" {

int oval_comm_unique_id(uint8_t* id_out)
try {
    if (!id_out) return oval_internal_fail(OVAL_E_INVALID, "id_out is NULL");
    Rccl& r = rccl();
    if (!r.ok) return oval_internal_fail(OVAL_E_NODEVICE, "%s", r.why);
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
    std::memcpy(id_out, &id, sizeof id);
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_comm_create(oval_ctx* ctx, int n_ranks, int rank, const uint8_t* id, oval_comm** out)
try {
    if (!ctx || !id || !out) return oval_internal_fail(OVAL_E_INVALID, "NULL argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return oval_internal_fail(OVAL_E_INVALID, "rank %d of %d ranks", rank, n_ranks);
    Rccl& r = rccl();
    if (!r.ok) return oval_internal_fail(OVAL_E_NODEVICE, "%s", r.why);
    const int device = oval_internal_ctx_device(ctx);
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return oval_internal_fail(OVAL_E_HIP, "hipSetDevice: %s", hipGetErrorString(he));
    oval_comm* c = new (std::nothrow) oval_comm;
    if (!c) return oval_internal_fail(OVAL_E_NOMEM, "out of host memory");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    const ncclResult_t e = r.CommInitRank(&c->comm, n_ranks, uid, rank);
    if (e != ncclSuccess) {
        delete c;
        return rccl_fail("ncclCommInitRank", e);
    }
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    *out = c;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_comm_destroy(oval_comm* comm)
try {
    if (!comm) return OVAL_OK;
    (void)hipSetDevice(comm->device);
    const ncclResult_t e = rccl().CommDestroy(comm->comm);
    delete comm;
    return e == ncclSuccess ? OVAL_OK : rccl_fail("ncclCommDestroy", e);
} catch (...) {
    return oval_internal_caught();
}

int oval_comm_info(const oval_comm* comm, int* n_ranks_out, int* rank_out)
try {
    if (!comm) return oval_internal_fail(OVAL_E_INVALID, "comm is NULL");
    Rccl& r = rccl();
    int n = comm->n_ranks, me = comm->rank;
    if (r.CommCount && r.CommUserRank) {
        // what the communicator itself says, not what it was asked for
        ncclResult_t e = r.CommCount(comm->comm, &n);
        if (e != ncclSuccess) return rccl_fail("ncclCommCount", e);
        e = r.CommUserRank(comm->comm, &me);
        if (e != ncclSuccess) return rccl_fail("ncclCommUserRank", e);
    }
    if (n_ranks_out) *n_ranks_out = n;
    if (rank_out) *rank_out = me;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_allreduce_counters(oval_ctx* ctx, oval_comm* comm, uint64_t* counters_dev, size_t n)
try {
    if (!ctx || !comm || (!counters_dev && n)) return oval_internal_fail(OVAL_E_INVALID, "NULL argument");
    if (((uintptr_t)counters_dev) % 8) return oval_internal_fail(OVAL_E_INVALID, "counters_dev is not 8-byte aligned");
    if (n == 0) return OVAL_OK;
    if (oval_internal_ctx_device(ctx) != comm->device)
        return oval_internal_fail(OVAL_E_INVALID, "the communicator was created on device %d, the context runs on %d",
                                  comm->device, oval_internal_ctx_device(ctx));
    const ncclResult_t e = rccl().AllReduce(counters_dev, counters_dev, n, ncclUint64, ncclSum, comm->comm,
                                            oval_internal_ctx_stream(ctx));
    return e == ncclSuccess ? OVAL_OK : rccl_fail("ncclAllReduce", e);
} catch (...) {
    return oval_internal_caught();
}

int oval_gather_first_bad(oval_ctx* ctx, const oval_obj_result* results_dev, uint64_t n_obj, uint32_t* table_dev)
try {
    if (!ctx || (n_obj && (!results_dev || !table_dev))) return oval_internal_fail(OVAL_E_INVALID, "NULL argument");
    if (n_obj >= 0xFFFFFFF0ull) return oval_internal_fail(OVAL_E_LAYOUT, "too many objects (%llu)", (unsigned long long)n_obj);
    if (((uintptr_t)results_dev) % 16 || ((uintptr_t)table_dev) % 4)
        return oval_internal_fail(OVAL_E_INVALID, "results_dev must be 16-byte aligned, table_dev 4-byte aligned");
    const hipError_t e = oval::launch_gather_first_bad(oval_internal_ctx_stream(ctx), results_dev, (uint32_t)n_obj, table_dev);
    if (e != hipSuccess) return oval_internal_fail(OVAL_E_HIP, "gather_first_bad: %s", hipGetErrorString(e));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_allreduce_first_bad(oval_ctx* ctx, oval_comm* comm, uint32_t* table_dev, size_t n)
try {
    if (!ctx || !comm || (!table_dev && n)) return oval_internal_fail(OVAL_E_INVALID, "NULL argument");
    if (((uintptr_t)table_dev) % 4) return oval_internal_fail(OVAL_E_INVALID, "table_dev is not 4-byte aligned");
    if (n == 0) return OVAL_OK;
    if (oval_internal_ctx_device(ctx) != comm->device)
        return oval_internal_fail(OVAL_E_INVALID, "the communicator was created on device %d, the context runs on %d",
                                  comm->device, oval_internal_ctx_device(ctx));
    const ncclResult_t e = rccl().AllReduce(table_dev, table_dev, n, ncclUint32, ncclMin, comm->comm,
                                            oval_internal_ctx_stream(ctx));
    return e == ncclSuccess ? OVAL_OK : rccl_fail("ncclAllReduce", e);
} catch (...) {
    return oval_internal_caught();
}

}  // extern "C"
