// ehr_host.h -- host-side plumbing shared by the translation units of libehr_hip.so (context, errors, scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/ehr.h"

namespace ehr {

void set_error(const std::string& msg);
int fail(int code, const char* fmt, ...);

#define EHR_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return ::ehr::fail(EHR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define EHR_LAUNCH_CHECK()                                                                              \
    do {                                                                                                \
        hipError_t e_ = hipGetLastError();                                                              \
        if (e_ != hipSuccess)                                                                           \
            return ::ehr::fail(EHR_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// A device buffer that only ever grows (ctx scratch: "sized lazily and grown, never shrunk").
struct Scratch {
    void* ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // may hipFree + hipMalloc (synchronises); never called on the fused hot path
    void release();
};

}  // namespace ehr

// Per-device rasterizer context (replaces nvdiffrast's RasterizeCudaContext).
struct ehr_ctx {
    int device = 0;
    // bin queues: counts/cursors/offsets are indexed by (image, tile, link)
    ehr::Scratch counts;    // int32 [2 * nkeys + 4]: counts | cursors | {total, overflow, nonempty, pad}
    ehr::Scratch offsets;   // int32 [nkeys]
    ehr::Scratch entries;   // int32 [entries_cap]
    size_t entries_cap = 0; // in entries
    int* host_pinned = nullptr;  // 4 ints, pinned, for the synchronous size read-back of the drop-in rasterize
    // fused path plan
    int pB = 0, pL = 0, pV = 0, pT = 0, pH = 0, pW = 0;
    int num_cus = 256;
    ehr::Scratch posc;       // float4 [B * V] clip-space vertices of the current step
    ehr::Scratch tile_part;  // float [B * NT * (1 + 12 * L)] per-tile partial loss + MVP gradients
    ehr::Scratch tile_list;  // int32 [2 * B * NT]: per-tile entry totals | work list of non-empty tiles
    // space-explorer scoring (ehr_mask_variance) keeps its own scratch so that it never disturbs a solver plan
    ehr::Scratch sc_counts, sc_offsets, sc_entries, sc_posc;
    size_t sc_entries_cap = 0;
    // side stream: the empty-tile streaming kernel overlaps the queue fill + tile kernels
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fill = nullptr;
    // natively captured launch chain (ehr_graph_*): capture stream and the instantiated graph
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    bool capturing = false;
    // measurement hook (ehr_fused_timing): EHR_FUSED_STAGES + 1 events per recorded call
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
};
