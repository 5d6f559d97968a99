// Internal declarations shared by the translation units of libpedp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/pedp.h"

void pedp_set_error(const char *fmt, ...);
uint64_t pedp_next_generation();
// Host -> device copy of a caller's (pageable) array on the context's stream.  Large copies go
// through a pinned staging buffer of the context: a fresh pageable buffer -- a new numpy array per
// frame -- otherwise pays for pinning its pages on every call (measured 25-30 ms for 8.8 MB against
// ~1.5 ms staged).  The source may be reused as soon as the call returns.
struct pedp_ctx_s;
int pedp_upload(pedp_ctx_s *c, void *dst, const void *src, size_t bytes);
// The way back: device -> caller's (pageable) array through the same staging buffers.  Returns
// after the data are in `dst` (the stream has been drained up to and including the copy).
int pedp_download(pedp_ctx_s *c, void *dst, const void *src, size_t bytes);
// `bytes` of device memory into the context's page-locked download buffer (valid until the next download through it);
// returns when the copy is complete: host code that only READS a result needs neither an array of its own nor a staged copy
int pedp_download_view(pedp_ctx_s *c, const void *src, size_t bytes, const void **view);
int pedp_download_parts(pedp_ctx_s *c, const void *src, size_t span, int n, const size_t *off, void *const *dst, const size_t *bytes);
struct pedp_ctx_s;
int pedp_sort_keys64_begin(pedp_ctx_s *c, int64_t N, int bits, unsigned long long **d_keys);
int pedp_sort_keys64_run(pedp_ctx_s *c, int64_t N, int bits, int32_t *d_perm);

#define PEDP_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            pedp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return PEDP_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define PEDP_REQUIRE(cond, ...)             \
    do {                                    \
        if (!(cond)) {                      \
            pedp_set_error(__VA_ARGS__);    \
            return PEDP_ERR_BAD_ARG;        \
        }                                   \
    } while (0)

// A growable device scratch buffer owned by the context (no hipMalloc on the hot path
// once sizes have settled).
struct pedp_scratch {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
};

// Device buffers of cloud handles that have been destroyed, kept for the next cloud of about that size: a scene
// cloud per camera frame would otherwise pay hipMalloc and a device-synchronising hipFree for each of its
// buffers.  Reuse is ordered by the context's stream (every use of a cloud's buffers is enqueued there).
struct pedp_pool {
    struct entry { void *p; size_t cap; };
    std::vector<entry> free_;
    std::mutex mu;  // a handle may be destroyed on another thread (Python drops the last reference wherever it likes)
    void *take(size_t bytes, size_t *cap);   // a pooled buffer of cap >= bytes (not wastefully larger), or a new one
    void give(void *p, size_t cap);
    void clear();
};

#define PEDP_MAX_SUB 8

// What a captured registration graph depends on besides device-memory contents: if any of it
// changes, the graph is captured again.
struct pedp_icp_graph_key {
    // clouds are identified by their generation id, not by the handle address: a destroyed
    // cloud's address can be handed out again by the next pedp_cloud_create
    uint64_t src_gen = 0, tgt_gen = 0;
    const void *ws = nullptr;
    int64_t Ns = 0, Nt = 0;
    int max_iter = 0, qt = 0, estimator = 0;
    int seg = 0, poses = 0;  // fused-path group graphs: passes per replay, poses per launch
    double r = 0.0, rel_fitness = 0.0, rel_rmse = 0.0;
};

struct pedp_comm_s;  // pedp_comm.hip: RCCL communicator of this rank
int pedp_comm_allreduce_sum_f64(pedp_ctx_t c, double *buf, int64_t n);
struct pedp_cloud_s;
int pedp_cloud_host_stats(pedp_ctx_t c, pedp_cloud_s *cl);  // makes centroid / lo / hi / Tn / T2 valid on the host
bool pedp_ctx_is_live(pedp_ctx_t c);

struct pedp_ctx_s {
    int device = 0;
    pedp_comm_s *comm = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    // ray casting
    pedp_scratch ray_keys;   // N x u64 packed (t_bits << 32 | prim_id)
    pedp_scratch ray_in;     // staging for host-memory calls
    pedp_scratch ray_out;
    pedp_scratch ray_aux;    // shared-origin flag + per-call shared-origin pair records
    pedp_scratch ray_order;  // direction order of the last frame's rays (kept between calls) + the samples it is checked by
    int64_t ray_order_n = -1;
    pedp_scratch ray_rast;   // variant 4: header, chain heads, nodes, item table
    void *rast_hdr_ready = nullptr;  // the buffer whose header holds its start values
    int *rast_status = nullptr;      // pinned, written by the last kernel of a variant-4 cast: [0] why the grid failed (0: it did not), [1] ray count
    unsigned rast_seq = 0;           // casts enqueued: picks one of the two headers
    int rast_avoid_n = -2;
    int ray_last_variant = 0;        // variant the last pedp_raycast ran           // ray count for which the grid is known not to work: variant 3 instead
    int ray_tri_chunks = 0;  // 0 = auto
    int ray_variant = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // sweep timing
    hipEvent_t ev_copy = nullptr;             // pedp_cloud_create_device: behind its copies (the host waits on it)
    bool ray_timed = false;
    bool nn_timed = false;
    hipEvent_t nn_ev0 = nullptr, nn_ev1 = nullptr;
    // sampled timing of a registration's sweep kernels (pedp_icp_configure timed_pass = -2): event
    // pairs around the sweep kernel of every fourth pass, created on first use
    hipEvent_t nn_evs[16] = {};
    int nn_pairs = 0;  // pairs recorded by the last timed registration (0: the single pair above)
    int nn_span_launches = 1;  // timed_pass = -3: launches between the single pair's two events
    // ICP
    pedp_scratch icp_ws;
    pedp_scratch ops;        // point-cloud operations (voxel grid, DBSCAN, kNN, plane RANSAC)
    pedp_scratch sort_ws;    // spatial order of a cloud: keys + radix-sort buffers (pooled, stream-ordered)
    pedp_pool cloud_pool;    // buffers of destroyed cloud handles
    pedp_scratch ops_small;  // partial bounds and counters of the point-cloud operations
    pedp_scratch chain;      // pedp_preprocess_source: the clouds between its stages
    void *avg_host = nullptr;  // pinned: preprocess_source's voxel means of the normals, written by the kernel itself (zero copy)
    size_t avg_host_cap = 0;
    pedp_scratch ops_in;     // a large cloud's points, uploaded ahead of the workspace sizing (its box comes from the device copy)
    pedp_scratch proj, proj_out;  // fused heat-map projection: selection, rays, hit records / compacted outputs
    bool icp_exhaustive = false;  // pedp_icp_configure: no culling (all-pairs sweep every pass)
    int icp_timed_pass = -1;      // pedp_icp_configure: HIP events around the sweep kernel of this pass
    long long icp_last_cand = 0, icp_last_fb = 0, icp_last_passes = 0, icp_last_nt = 0, icp_last_planned = 0;
    void *icp_pending = nullptr;   // pedp_icp_begin's job until pedp_icp_end collects it (an IcpJob of pedp_icp.hip)  // last pedp_icp
    pedp_ctx_s *sub[PEDP_MAX_SUB] = {};  // sub-contexts (own stream + workspace) for batched registrations
    hipGraphExec_t icp_graph = nullptr;  // sub-contexts: one whole registration, replayed per start pose
    pedp_icp_graph_key icp_graph_key;
    // fused path: one graph per group size 1, 2, 4, ... 32 (poses share launches) and per stretch of passes
    // [0] one pass (evaluation), [1] two (single-iteration probes), [2] the rest -- a frame alternates between them
    hipGraphExec_t icp_bgraph[18] = {};
    pedp_icp_graph_key icp_bgraph_key[18];
    void *pinned = nullptr;  // small pinned host block for result read-back
    size_t pinned_cap = 0;
    void *stage[2] = {nullptr, nullptr};  // pinned staging buffers of pedp_upload [0] / pedp_download [1], grown on demand
    size_t stage_cap[2] = {0, 0};
    bool stage_busy = false;              // an upload's DMA may still be reading stage[0]
};

// Triangle record: 12 floats (48 B, 16-B aligned so a wave fetches it with scalar
// dwordx4/x8 loads): v0.xyz, e1.xyz, e2.xyz, 3 pad.
#define PEDP_TRI_STRIDE 12

struct pedp_mesh_s {
    pedp_ctx_t ctx = nullptr;
    int device = 0;  // copied at creation: destroy must not dereference a context that may be gone
    int64_t V = 0, F = 0;
    float *tri = nullptr;  // F_padded x PEDP_TRI_STRIDE floats on the device
    float *tri2 = nullptr; // F_padded/2 pair-interleaved general-origin records (24 floats each)
    void *spheres = nullptr;  // n_clusters x float4 bounding spheres of 16-triangle clusters
    int64_t n_clusters = 0;
    void *super_spheres = nullptr;  // bounding sphere of every 64 clusters (one cull-mask word)
    int64_t n_super = 0;
    int64_t F_padded = 0;  // multiple of 8; pad records can never be hit (all zero => det == 0)
    // posable meshes keep the model-frame float64 vertices, the float32 posed vertices and the
    // index buffer resident so that a new pose rebuilds the records on the device
    double *verts64 = nullptr;
    float *verts32 = nullptr;
    uint32_t *idx = nullptr;
};

struct pedp_cloud_s {
    pedp_ctx_t ctx = nullptr;
    int device = 0;     // copied at creation: destroy must not dereference a context that may be gone
    uint64_t gen = 0;   // process-wide creation counter (never reused), keys cached graphs
    int64_t N = 0;
    bool has_normals = false;
    double *pts = nullptr;      // N x 3 f64 (device)
    double *normals = nullptr;  // N x 3 f64 (device) or null
    size_t pts_cap = 0, normals_cap = 0, perm_cap = 0, sph_cap = 0, box_cap = 0;  // capacities (the buffers may come from the pool)
    double *d_box = nullptr;    // device: lo xyz, hi xyz of the points (what the spatial order is laid over)
    // centroid, box and the filter's magnitudes ON THE HOST: at creation for clouds made from host arrays, on first
    // use as a registration TARGET for clouds made from device memory (a scene cloud never needs them: no read-back
    // and no synchronisation between the camera frame and its registration)
    bool host_stats = false;
    // Set at creation (host pass over the points): centroid and the two magnitudes of the
    // centred cloud that the NN filter's error bound needs when this cloud is the target.
    double centroid[3] = {0, 0, 0};
    float Tn = 0.f;  // max |t'|_1
    float T2 = 0.f;  // max |t'|_2^2
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};  // axis-aligned bounding box of the points
    // Built on first use in ICP: spatial order of the points (device int32[N]); as a target
    // additionally the sorted float4 operand (x', y', z', |t'|^2), padded, and one bounding
    // sphere per 16-row tile.
    void *perm = nullptr;       // built from the cloud alone (its own bounding box): never rebuilt
    void *chunk_sph = nullptr;  // with it: bounding sphere (float64 x 4) of every 128 consecutive points of that order
    void *tgt4 = nullptr;
    void *tile_sph = nullptr;   // one sphere per 16 sorted rows
    void *tile_sph4 = nullptr;  // one sphere per 64 sorted rows
    void *tile_sphw = nullptr;  // one sphere per 1024 sorted rows (one cull-mask word of 16-row tiles)
    void *tgt_s = nullptr;      // sorted rows as float64 x 3 (exact re-scoring)
    void *tgt_bf = nullptr;     // the sorted operand as bf16 pieces for the dense sweep's v_mfma_f32_16x16x32_bf16 (1 KB per 16-row tile; built on first use)
    int64_t tgt4_pad = 0;
};

// pedp_icp.hip: frees the job of a pedp_icp_begin that was never ended (called by pedp_ctx_destroy)
void pedp_icp_drop_pending(pedp_ctx_t c);
