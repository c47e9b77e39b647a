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
#include "ehr_pose_core.h"

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

// ---- one-shot exchange over peer memory (round 6; SURVEY 5 / 8e's "later option") -------------------------------------------
// An 8-rank ncclAllReduce of 32 bytes costs 10-30 us and is followed by a launch of its own for Adam; the step it sits in is
// ~70 us.  Here every rank owns a MAILBOX in its device memory (uncached, exported with hipIpcGetMemHandle, opened by every
// peer), and the exchange is ONE single-workgroup kernel per rank and step: lane p stores this rank's 8 floats and then the
// exchange's sequence number (release, system scope) into slot `rank` of peer p's mailbox -- over xGMI for a peer GPU --, lane p
// then waits for slot p of its own mailbox to show the same number (acquire), eight lanes add the slots up IN RANK ORDER (every
// rank adds the same numbers in the same order: bit-identical sums, hence bit-identical Adam steps), and the kernel goes on to
// the Adam update.  Slots alternate by the parity of the sequence number: a rank can only be two exchanges ahead of a peer
// after that peer has read the older one.  A wait that does not end (a peer died) is reported -- NaN sums, Adam untouched --
// after ~half a minute, never a hang.  Kernel only: capturable in a hipGraph like the rest of the chain.
constexpr int P2P_MAX = EHR_P2P_MAX_RANKS;
constexpr int P2P_SLOT = 64;                       // bytes: 8 floats | sequence number | padding (a slot never shares a line)
constexpr int P2P_HDR = 128;                       // [0] exchanges completed by the owner
constexpr size_t P2P_BYTES = P2P_HDR + 2 * (size_t)P2P_MAX * P2P_SLOT;

struct P2PPeers {
    char* mail[P2P_MAX];
};

__global__ void __launch_bounds__(64)
p2p_exchange_adam_kernel(P2PPeers peers, char* __restrict__ mine, int nranks, int rank, float* __restrict__ red, int with_adam,
                         float* __restrict__ dof, float* __restrict__ m, float* __restrict__ v, int* __restrict__ step, float lr,
                         float b1, float b2, float eps, float wd, float* __restrict__ loss_out, float* __restrict__ grad_out) {
    __shared__ float s_red[8];
    __shared__ int s_bad;
    const int p = threadIdx.x;
    int* const seqp = reinterpret_cast<int*>(mine);
    const int seq = seqp[0] + 1;  // (every rank counts its exchanges alike: one per step)
    const size_t base = P2P_HDR + (size_t)(seq & 1) * P2P_MAX * P2P_SLOT;
    AdamState st;
    if (with_adam) st = pose_adam_fetch(dof, m, v, step);  // (requested before the wait)
    if (p == 0) s_bad = 0;
    __syncthreads();
    if (p < nranks) {
        float* const dst = reinterpret_cast<float*>(peers.mail[p] + base + (size_t)rank * P2P_SLOT);
#pragma unroll
        for (int k = 0; k < 8; k++) __hip_atomic_store(&dst[k], red[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(reinterpret_cast<int*>(dst + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const int* const flag = reinterpret_cast<const int*>(mine + base + (size_t)p * P2P_SLOT + 32);
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 24)) {  // (~half a minute: far beyond a peer that is only late -- a re-plan after a reported step
                                        //  takes a second --; a rank that gave up would leave the replicated state for good)
                s_bad = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (p < 8) {
        float sum = 0.f;
        for (int q = 0; q < nranks; q++)  // rank order, on every rank
            sum += __hip_atomic_load(reinterpret_cast<const float*>(mine + base + (size_t)q * P2P_SLOT) + p, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM);
        if (s_bad) sum = __int_as_float(0x7fc00000);
        s_red[p] = sum;
        red[p] = sum;  // (in place, like the all-reduce it replaces)
    }
    if (p == 0) seqp[0] = seq;
    __syncthreads();
    if (with_adam) pose_adam_apply(st, dof, m, v, step, s_red, lr, b1, b2, eps, wd, loss_out, grad_out);
}

}  // namespace ehr
using namespace ehr;

extern "C" {

int ehr_comm_p2p_export(ehr_ctx* ctx, void* handle64) {
    if (!ctx || !handle64) return fail(EHR_ERR_INVALID, "ehr_comm_p2p_export: NULL");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "include/ehr.h promises 64 bytes");
    int cur = 0;
    EHR_HIP(hipGetDevice(&cur));
    EHR_HIP(hipSetDevice(ctx->device));
    if (!ctx->p2p_mail) {
        // uncached device memory: a peer's stores arrive in memory and this rank's polling loads must see them there
        hipError_t e = hipExtMallocWithFlags(&ctx->p2p_mail, P2P_BYTES, hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipSetDevice(cur);
            return fail(EHR_ERR_HIP, "ehr_comm_p2p_export: hipExtMallocWithFlags failed: %s", hipGetErrorString(e));
        }
    }
    hipError_t e = hipMemset(ctx->p2p_mail, 0, P2P_BYTES);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, ctx->p2p_mail);
    (void)hipSetDevice(cur);
    if (e != hipSuccess) return fail(EHR_ERR_HIP, "ehr_comm_p2p_export: %s", hipGetErrorString(e));
    memcpy(handle64, &h, sizeof(h));
    return EHR_OK;
}

int ehr_comm_p2p_open(ehr_ctx* ctx, const void* handles, int nranks, int rank) {
    if (!ctx || !handles || nranks < 1 || nranks > P2P_MAX || rank < 0 || rank >= nranks)
        return fail(EHR_ERR_INVALID, "ehr_comm_p2p_open: bad argument (at most %d ranks)", P2P_MAX);
    if (!ctx->p2p_mail) return fail(EHR_ERR_INVALID, "ehr_comm_p2p_open: call ehr_comm_p2p_export first");
    if (ctx->gexec) {  // a captured data-parallel step holds the old peers in its kernel arguments
        EHR_HIP(hipGraphExecDestroy(ctx->gexec));
        ctx->gexec = nullptr;
    }
    int cur = 0;
    EHR_HIP(hipGetDevice(&cur));
    EHR_HIP(hipSetDevice(ctx->device));
    for (int q = 0; q < ctx->p2p_ranks; q++)
        if (q != ctx->p2p_rank && ctx->p2p_peer[q]) (void)hipIpcCloseMemHandle(ctx->p2p_peer[q]);
    ctx->p2p_ranks = 0;
    int rc = EHR_OK;
    for (int q = 0; q < nranks && rc == EHR_OK; q++) {
        ctx->p2p_peer[q] = nullptr;
        if (q == rank) {
            ctx->p2p_peer[q] = ctx->p2p_mail;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + 64 * (size_t)q, sizeof(h));
        hipError_t e = hipIpcOpenMemHandle(&ctx->p2p_peer[q], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            ctx->p2p_peer[q] = nullptr;
            rc = fail(EHR_ERR_HIP, "ehr_comm_p2p_open: hipIpcOpenMemHandle of rank %d's mailbox failed: %s", q, hipGetErrorString(e));
        }
    }
    if (rc != EHR_OK) {
        for (int q = 0; q < nranks; q++)
            if (q != rank && ctx->p2p_peer[q]) (void)hipIpcCloseMemHandle(ctx->p2p_peer[q]);
        (void)hipSetDevice(cur);
        return rc;
    }
    (void)hipSetDevice(cur);
    ctx->p2p_ranks = nranks;
    ctx->p2p_rank = rank;
    return EHR_OK;
}

int ehr_comm_p2p_step(ehr_ctx* ctx, float* red, float* dof, float* m, float* v, int32_t* step, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float* loss_out, float* grad_out, void* stream) {
    if (!ctx || !red) return fail(EHR_ERR_INVALID, "ehr_comm_p2p_step: NULL");
    if (ctx->p2p_ranks < 1) return fail(EHR_ERR_INVALID, "ehr_comm_p2p_step: call ehr_comm_p2p_open first");
    const int with_adam = dof != nullptr;
    if (with_adam && (!m || !v || !step)) return fail(EHR_ERR_INVALID, "ehr_comm_p2p_step: dof without its optimiser state");
    P2PPeers peers;
    for (int q = 0; q < P2P_MAX; q++) peers.mail[q] = q < ctx->p2p_ranks ? (char*)ctx->p2p_peer[q] : nullptr;
    p2p_exchange_adam_kernel<<<1, 64, 0, (hipStream_t)stream>>>(peers, (char*)ctx->p2p_mail, ctx->p2p_ranks, ctx->p2p_rank, red,
                                                               with_adam, dof, m, v, step, lr, beta1, beta2, eps, weight_decay,
                                                               loss_out, grad_out);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_comm_p2p_close(ehr_ctx* ctx) {
    if (!ctx) return EHR_OK;
    if (ctx->gexec && ctx->p2p_ranks) {
        (void)hipGraphExecDestroy(ctx->gexec);
        ctx->gexec = nullptr;
    }
    for (int q = 0; q < ctx->p2p_ranks; q++)
        if (q != ctx->p2p_rank && ctx->p2p_peer[q]) (void)hipIpcCloseMemHandle(ctx->p2p_peer[q]);
    ctx->p2p_ranks = 0;
    if (ctx->p2p_mail) {
        (void)hipFree(ctx->p2p_mail);
        ctx->p2p_mail = nullptr;
    }
    return EHR_OK;
}

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
