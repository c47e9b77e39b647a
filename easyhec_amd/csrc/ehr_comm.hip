// ehr_comm.hip -- the ONE collective of a data-parallel step (the reference's DDP gradient exchange,
// /root/reference/easyhec/trainer/base.py:349-352, launched as tools/run_easyhec.py:41-50 does: one process per GPU),
// issued straight on an RCCL communicator this library creates itself: ncclAllReduce(sum) of the 8-float exchange vector
// [d sum(loss)/d dof (6), sum(loss), n_views] on the launch chain's own stream, between ehr_solver_step(defer_adam = 1)
// and ehr_pose_adam.  No torch.distributed on the step path, no second stream, no event hand-offs, and the whole
// data-parallel step can be captured in a hipGraph.  RCCL is resolved at run time (dlopen: the library a host process
// already holds, e.g. PyTorch's, is reused) so that libehr_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "ehr_host.h"

namespace ehr {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load() {
    if (g_rccl.lib) return EHR_OK;
    // First the copy the host process already holds (PyTorch bundles its own librccl.so: a second RCCL beside it would
    // set up the GPU's transports twice), found with RTLD_NOLOAD; only then a fresh load.
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (h) break;
    }
    if (!h)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) return fail(EHR_ERR_INVALID, "RCCL not found (librccl.so): %s", dlerror());
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString)
        return fail(EHR_ERR_INVALID, "librccl.so lacks an expected symbol");
    g_rccl = a;
    return EHR_OK;
}

#define EHR_NCCL(call)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return ::ehr::fail(EHR_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

}  // namespace ehr
using namespace ehr;

extern "C" {

int ehr_comm_unique_id(void* id128) {
    if (!id128) return fail(EHR_ERR_INVALID, "ehr_comm_unique_id: NULL");
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "include/ehr.h promises 128 bytes");
    ncclUniqueId id;
    EHR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return EHR_OK;
}

int ehr_comm_init(ehr_ctx* ctx, const void* id128, int nranks, int rank) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(EHR_ERR_INVALID, "ehr_comm_init: bad argument");
    int rc = rccl_load();
    if (rc) return rc;
    if (ctx->comm) {
        if (ctx->gexec) {  // a captured data-parallel step holds the old communicator in its all-reduce node
            EHR_HIP(hipGraphExecDestroy(ctx->gexec));
            ctx->gexec = nullptr;
        }
        EHR_NCCL(g_rccl.CommDestroy((ncclComm_t)ctx->comm));
        ctx->comm = nullptr;
    }
    int cur = 0;
    EHR_HIP(hipGetDevice(&cur));
    EHR_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    (void)hipSetDevice(cur);
    if (r != ncclSuccess) return fail(EHR_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    ctx->comm = (void*)comm;
    ctx->comm_ranks = nranks;
    return EHR_OK;
}

int ehr_comm_allreduce(ehr_ctx* ctx, float* red, int count, void* stream) {
    if (!ctx || !red || count <= 0) return fail(EHR_ERR_INVALID, "ehr_comm_allreduce: bad argument");
    if (!ctx->comm) return fail(EHR_ERR_INVALID, "ehr_comm_allreduce: call ehr_comm_init first");
    EHR_NCCL(g_rccl.AllReduce(red, red, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)ctx->comm, (hipStream_t)stream));
    return EHR_OK;
}

int ehr_comm_destroy(ehr_ctx* ctx) {
    if (!ctx || !ctx->comm) return EHR_OK;
    ncclComm_t c = (ncclComm_t)ctx->comm;
    ctx->comm = nullptr;
    if (ctx->gexec) {
        (void)hipGraphExecDestroy(ctx->gexec);
        ctx->gexec = nullptr;
    }
    EHR_NCCL(g_rccl.CommDestroy(c));
    return EHR_OK;
}

}  // extern "C"
