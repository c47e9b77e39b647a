// ehr_host.h -- host-side plumbing shared by the translation units of libehr_hip.so (context, errors, scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/ehr.h"

struct ehr_ctx;

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
    unsigned long long moves = 0;  // bumped whenever this buffer moves: captured graphs hold raw pointers
    // set once a call that used this buffer was recorded into somebody else's stream capture (torch.cuda.graphs): the graph
    // holds the raw pointer, so the buffer must never move again -- a reserve() that would have to grow it fails loudly
    // instead of leaving the graph's replays writing into freed memory
    bool pinned = false;
};

// Device fills / copies of the drop-in ops as plain kernels (ehr_raster.hip): a hipMemsetAsync / hipMemcpyAsync recorded
// inside a torch.cuda.graphs capture of a solver step becomes a memset / memcpy graph node, and a graph holding such
// nodes faulted on its SECOND replay on this stack (ROCm 7.0 / MI355X; kernels only: replays fine).
int zero_words(void* dst, size_t nwords, hipStream_t stream);
int fill_words(void* dst, size_t nwords, unsigned value, hipStream_t stream);
int copy_words(void* dst, const void* src, size_t nwords, hipStream_t stream);

struct StepHead;
struct StepTail;
struct RasterShape {  // what makes two drop-in rasterize calls "the same frame again" for the sync-free size read-back
    int B, V, T, H, W, ranged;
    bool operator==(const RasterShape& o) const { return B == o.B && V == o.V && T == o.T && H == o.H && W == o.W && ranged == o.ranged; }
};
#define VB_LOSS_SLOTS 32          // partial frame-loss sums per view (spreads same-address atomics)
#define VB_LOSS_STRIDE 16         // i64 between two of them: one 128-byte line each (atomics on one line serialise)
#define VB_MAX_UNITS 512          // views x links one context plans for
#define VB_SPILL_ITEMS (1 << 20)  // pool of blended-pair items for tiles that overflow their LDS list (16 MB)
int vbuf_plan(ehr_ctx* ctx, int B, int L, int V, int T, int H, int W, float slack, const float* verts,
              const int32_t* tris, const int32_t* tri_link, const int32_t* opp);
int vbuf_meta_read(ehr_ctx* ctx, int* meta4);
int vbuf_bind_ref(ehr_ctx* ctx, const float* ref, hipStream_t stream);
int vbuf_score(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* vert_link, const float* mvp, int Q, int S,
               int L, int V, int T, int H, int W, long long* score, unsigned char* count, hipStream_t stream, int* handled);
int vbuf_chain(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link, const int32_t* vert_link,
               const int32_t* opp, float* mvp, const float* ref, int B, int L, int V, int T, int H, int W, float* mask,
               float* loss, float* grad_mvp, const StepHead* head, const StepTail* tail, hipStream_t stream);

}  // namespace ehr

// Per-device rasterizer context (replaces nvdiffrast's RasterizeCudaContext).
struct ehr_ctx {
    int device = 0;
    // bin queues: counts/cursors/offsets are indexed by (image, tile, link)
    ehr::Scratch counts;    // int32 [2 * nkeys + 48]: counts | cursors | meta {total, overflow, nonempty, ...}.  ALL ZERO between
                            // calls: the last kernel of a drop-in rasterize call zeroes every word the call dirtied
                            // (a fill kernel per call was a tenth of the three-op step's launches)
    unsigned long long counts_clean = ~0ull;  // == counts.moves: the buffer is known to be all zero
    bool vb_slow_needed = false;  // a solver step met a triangle for the general path: vb_slow_kernel is launched from now on
    ehr::Scratch ranges;    // int32 [2 * B]: the per-image triangle ranges of a range-mode call
    std::vector<int32_t> ranges_last;            // what `ranges` holds on the device (host copy): the same ranges again are not
    unsigned long long ranges_moves = ~0ull;     // uploaded again -- a solve passes them every step, and an upload recorded
                                                 // into a stream capture would be a memcpy node (see zero_words)
    ehr::Scratch rkeys;     // u64 [B * H * W]: key image of the drop-in rasterizer's direct form; ALL ONES between calls
    unsigned long long rkeys_clean = ~0ull;  // == rkeys.moves: known to be all ones
    ehr::Scratch offsets;   // int32 [nkeys]
    ehr::Scratch entries;   // int32 [entries_cap]
    size_t entries_cap = 0; // in entries
    int* host_pinned = nullptr;  // 8 ints, pinned: size read-backs of the scoring op ([0..5]) and the drop-in rasterize ([6..7])
    hipEvent_t ev_size[2] = {nullptr, nullptr};  // drop-in rasterize: "the size in host_pinned[k] has arrived"
    bool size_valid[2] = {false, false};
    ehr::RasterShape size_shape[2] = {};         // shape of the call that produced it
    int size_slot = 0;
    // fused path plan
    int pB = 0, pL = 0, pV = 0, pT = 0, pH = 0, pW = 0;
    int num_cus = 256;
    // launch chain of the fused op (ehr_vbuf.hip); its own scratch, never shared with the drop-in ops
    ehr::Scratch vb_clus;    // i32 cluster index: ctri [NC][64] | clink [NC] | coff [L + 1] (static, built by the plan)
    ehr::Scratch vb_heavy;   // heavy-job scheduling hint carried from step to step (generation, lists, stamps)
    ehr::Scratch vb_idx;     // int4 [T] padded triangle indices | int4 [T] padded edge topology (static)
    const void* vb_plan_tris = nullptr;  // the scene the static index was built for
    const void* vb_plan_opp = nullptr;
    const void* vb_plan_verts = nullptr;
    int vb_nc = 0;           // number of clusters
    int vb_jcap = 0;         // job slots (of one chunk of views)
    int vb_chunk = 0;        // views per pass of the chain (plan time: LDS tables of the job kernel, scratch budget)
    ehr::Scratch vb_boxes;   // uint2 pixel boxes of the current step: tbox [B][NC][64] | cbox [B][NC]
    ehr::Scratch vb_units;   // i32 [B][L][4] pixel boxes of the links (re-armed by the finish kernel)
    ehr::Scratch vb_acc;     // i64 [B][12 L + VB_LOSS_SLOTS * VB_LOSS_STRIDE] fixed-point sums, then the meta words
    ehr::Scratch vb_posc;    // float4 [B][V] clip-space vertices (eager plans only)
    bool vb_lazy = false;    // the plan computes clip-space vertices where they are looked up (VbLazy in ehr_vbuf.hip): V > 1.5 T
    ehr::Scratch vb_jobs;    // per (view, link, tile) job slot: value tile | blended pairs | count | spill base
    ehr::Scratch vb_spill;   // blended pairs of jobs that exceed their slot
    int vb_spill_cap = 0;    // ... in items
    ehr::Scratch vb_refsum;  // cached sums of the bound reference mask: tsum i64 [B][nt] | vtot i64 [B] | flag
    const float* vb_ref = nullptr;  // the reference mask those sums belong to (ehr_fused_bind_ref), or NULL
    ehr::Scratch vb_hstate;  // i32 [16], survives re-plans: [0] Adam's step counter + 1 as the previous solver step's head saw
                             // it, [1] whether that head advanced the history cursor, [2] where it left it (a REPORTED step's row is reused)
    // space-explorer scoring (ehr_mask_variance) keeps its own scratch so that it never disturbs a solver plan
    ehr::Scratch sc_counts, sc_offsets, sc_entries, sc_posc;
    size_t sc_entries_cap = 0;
    // ... and, for the coverage-only chain of the scoring op (ehr_vbuf.hip: vbuf_score), the static cluster index of its mesh
    ehr::Scratch sc_clus, sc_misc;
    int sc_nc = 0;
    const void* sc_key[3] = {nullptr, nullptr, nullptr};  // (verts, tris, vert_link) the index was built for
    int sc_key_n[3] = {0, 0, 0};                          // (V, T, L)
    unsigned long long sc_hash = 0;                       // content hash of those arrays (an in-place edit rebuilds the index)
    bool sc_mixed = false;                                // that mesh cannot take the chain (links not grouped)
    // RCCL communicator of the data-parallel exchange (ehr_comm_*; an ncclComm_t), created by the library itself
    void* comm = nullptr;
    int comm_ranks = 0;
    // ... and the one-shot exchange over peer memory (ehr_comm_p2p_*): this rank's mailbox and the peers' (opened IPC handles)
    void* p2p_mail = nullptr;
    void* p2p_peer[EHR_P2P_MAX_RANKS] = {};
    int p2p_ranks = 0, p2p_rank = 0;
    // natively captured launch chain (ehr_graph_*): capture stream and the instantiated graph
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    unsigned long long gexec_reallocs = 0;  // scratch_moves() when the graph was instantiated
    // how often any scratch of THIS context moved (other contexts of the process do not disturb a captured graph)
    unsigned long long scratch_moves() const {
        unsigned long long n = 0;
        for (const ehr::Scratch* s : {&counts, &offsets, &entries, &vb_clus, &vb_heavy, &vb_idx, &vb_boxes, &vb_units, &vb_acc,
                                      &vb_posc, &vb_jobs, &vb_spill, &vb_refsum, &vb_hstate, &sc_counts, &sc_offsets, &sc_entries, &sc_posc, &sc_clus, &sc_misc})
            n += s->moves;
        return n;
    }
    bool capturing = false;
    // measurement hook (ehr_fused_timing): EHR_FUSED_STAGES + 1 events per recorded call
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
};
