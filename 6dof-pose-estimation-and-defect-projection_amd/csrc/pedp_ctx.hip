// Context, error reporting and resource handles of libpedp_hip.so.
#include "pedp_internal.h"
#include <cmath>
#include <new>

#include <atomic>
#include <mutex>
#include <set>

static thread_local char g_err[512] = "";
static std::atomic<uint64_t> g_generation{0};

uint64_t pedp_next_generation() { return ++g_generation; }

// live contexts: a cloud handle destroyed after its context hands its buffers to hipFree, not to a pool that is gone
static std::mutex g_ctx_mutex;
static std::set<pedp_ctx_s *> g_ctx_live;
bool pedp_ctx_is_live(pedp_ctx_t c) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return g_ctx_live.count(c) != 0;
}

void *pedp_pool::take(size_t bytes, size_t *cap) {
    std::lock_guard<std::mutex> lock(mu);
    int best = -1;
    for (int i = 0; i < (int)free_.size(); ++i)
        if (free_[i].cap >= bytes && free_[i].cap <= 2 * bytes + 4096 && (best < 0 || free_[i].cap < free_[best].cap)) best = i;
    if (best >= 0) {
        void *p = free_[best].p;
        *cap = free_[best].cap;
        free_.erase(free_.begin() + best);
        return p;
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 16 + 256;   // a little headroom: frames of a stream differ by a few points
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *cap = want;
    return p;
}
void pedp_pool::give(void *p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(mu);
    if (free_.size() >= 24) {  // keep the pool small: drop the oldest entry
        (void)hipFree(free_.front().p);
        free_.erase(free_.begin());
    }
    free_.push_back({p, cap});
}
void pedp_pool::clear() {
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : free_) (void)hipFree(e.p);
    free_.clear();
}

void pedp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace { constexpr size_t STAGE_MIN = 256u << 10, STAGE_MAX = 64u << 20; }

// one pinned staging buffer per context and direction, grown on demand up to STAGE_MAX
static int stage_reserve(pedp_ctx_s *c, int which, size_t bytes) {
    if (c->stage_cap[which] >= bytes) return PEDP_OK;
    if (c->stage[which]) {
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        (void)hipHostFree(c->stage[which]);
        c->stage[which] = nullptr;
        c->stage_cap[which] = 0;
    }
    size_t want = bytes + bytes / 4;
    if (want > STAGE_MAX) want = STAGE_MAX;
    PEDP_HIP_CHECK(hipHostMalloc(&c->stage[which], want, hipHostMallocDefault));
    c->stage_cap[which] = want;
    return PEDP_OK;
}

// page-locked host memory (hipHostMalloc / hipHostRegister, whoever made it)?
static bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary pageable pointer: not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int pedp_upload(pedp_ctx_s *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return PEDP_OK;
    if (bytes < STAGE_MIN) {
        PEDP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return PEDP_OK;
    }
    if (host_is_pinned(src)) {  // the caller's buffer is page-locked already (e.g. a pinned torch tensor): no staging copy
        PEDP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return PEDP_OK;          // (every entry point synchronises the stream before it hands the buffer back)
    }
    int rc = stage_reserve(c, 0, bytes < STAGE_MAX ? bytes : STAGE_MAX);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += c->stage_cap[0]) {
        const size_t n = bytes - off < c->stage_cap[0] ? bytes - off : c->stage_cap[0];
        // the DMA that last read the buffer (an earlier upload on this stream) must be done
        if (c->stage_busy || off > 0) PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        // in pieces: the DMA of one piece runs while the host copies the next into its own part of the staging buffer (a
        // caller's array that has left the CPU caches is read at DRAM speed -- about as long as the DMA takes)
        constexpr size_t PIECE = 512u << 10;
        for (size_t p = 0; p < n; p += PIECE) {
            const size_t m = n - p < PIECE ? n - p : PIECE;
            memcpy((char *)c->stage[0] + p, (const char *)src + off + p, m);
            PEDP_HIP_CHECK(hipMemcpyAsync((char *)dst + off + p, (char *)c->stage[0] + p, m, hipMemcpyHostToDevice, c->stream));
        }
        c->stage_busy = true;
    }
    return PEDP_OK;
}

int pedp_download_view(pedp_ctx_s *c, const void *src, size_t bytes, const void **view) {
    int rc = stage_reserve(c, 1, bytes > STAGE_MIN ? bytes : STAGE_MIN);
    if (rc) return rc;
    if (bytes > c->stage_cap[1]) { pedp_set_error("pedp_download_view: %zu bytes exceed the staging buffer", bytes); return PEDP_ERR_BAD_ARG; }
    if (bytes) PEDP_HIP_CHECK(hipMemcpyAsync(c->stage[1], src, bytes, hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->stage_busy = false;
    *view = c->stage[1];
    return PEDP_OK;
}

int pedp_download(pedp_ctx_s *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return PEDP_OK;
    if (bytes < STAGE_MIN) {
        PEDP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->stage_busy = false;
        return PEDP_OK;
    }
    if (host_is_pinned(dst)) {
        PEDP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->stage_busy = false;
        return PEDP_OK;
    }
    int rc = stage_reserve(c, 1, bytes < STAGE_MAX ? bytes : STAGE_MAX);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += c->stage_cap[1]) {
        const size_t n = bytes - off < c->stage_cap[1] ? bytes - off : c->stage_cap[1];
        PEDP_HIP_CHECK(hipMemcpyAsync(c->stage[1], (const char *)src + off, n, hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->stage_busy = false;
        memcpy((char *)dst + off, c->stage[1], n);
    }
    return PEDP_OK;
}

// Several results that lie in ONE device block (parts i at src + off[i], bytes[i] long, dst[i] == nullptr: skipped) in one
// copy through the pinned staging buffer and one wait, instead of a copy and a wait per result.  Blocks beyond the
// staging buffer's limit go part by part.
int pedp_download_parts(pedp_ctx_s *c, const void *src, size_t span, int n, const size_t *off, void *const *dst, const size_t *bytes) {
    if (span == 0) return PEDP_OK;
    if (span > STAGE_MAX) {
        for (int i = 0; i < n; ++i)
            if (dst[i]) { int rc = pedp_download(c, dst[i], (const char *)src + off[i], bytes[i]); if (rc) return rc; }
        return PEDP_OK;
    }
    bool all_pinned = true;   // page-locked destinations (pedp_host_alloc, a pinned torch tensor): straight there, one wait
    for (int i = 0; i < n && all_pinned; ++i)
        if (dst[i] && bytes[i] && !host_is_pinned(dst[i])) all_pinned = false;
    if (all_pinned) {
        for (int i = 0; i < n; ++i)
            if (dst[i] && bytes[i]) PEDP_HIP_CHECK(hipMemcpyAsync(dst[i], (const char *)src + off[i], bytes[i], hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->stage_busy = false;
        return PEDP_OK;
    }
    int rc = stage_reserve(c, 1, span);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipMemcpyAsync(c->stage[1], src, span, hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->stage_busy = false;
    for (int i = 0; i < n; ++i)
        if (dst[i]) memcpy(dst[i], (const char *)c->stage[1] + off[i], bytes[i]);
    return PEDP_OK;
}

int pedp_scratch::reserve(size_t bytes) {
    if (bytes <= cap) return PEDP_OK;
    release();
    size_t want = bytes + bytes / 4 + 256;
    PEDP_HIP_CHECK(hipMalloc(&ptr, want));
    cap = want;
    return PEDP_OK;
}

void pedp_scratch::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

extern "C" {

int pedp_version(void) { return 100; }

const char *pedp_last_error(void) { return g_err; }

int pedp_device_count(int *count) {
    PEDP_REQUIRE(count, "pedp_device_count: null output");
    PEDP_HIP_CHECK(hipGetDeviceCount(count));
    return PEDP_OK;
}

int pedp_ctx_create(int device, void *stream, pedp_ctx_t *out) {
    PEDP_REQUIRE(out, "pedp_ctx_create: null output");
    *out = nullptr;
    int n = 0;
    PEDP_HIP_CHECK(hipGetDeviceCount(&n));
    PEDP_REQUIRE(device >= 0 && device < n, "pedp_ctx_create: device %d out of range (%d visible)", device, n);
    PEDP_HIP_CHECK(hipSetDevice(device));
    pedp_ctx_s *c = new (std::nothrow) pedp_ctx_s();
    if (!c) { pedp_set_error("pedp_ctx_create: out of host memory"); return PEDP_ERR_ALLOC; }
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; pedp_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return PEDP_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->nn_ev0) != hipSuccess || hipEventCreate(&c->nn_ev1) != hipSuccess) {
        pedp_set_error("pedp_ctx_create: hipEventCreate failed");
        pedp_ctx_destroy(c);
        return PEDP_ERR_HIP;
    }
    c->pinned_cap = 1 << 17;  // [0, 64 K): the registrations' states and the operations' counters; above: preprocess_source's average normal (64 K) and refit plane (65 K), the jet table (68 K)
    if (hipHostMalloc(&c->pinned, c->pinned_cap, hipHostMallocDefault) != hipSuccess) {
        pedp_set_error("pedp_ctx_create: hipHostMalloc failed");
        pedp_ctx_destroy(c);
        return PEDP_ERR_ALLOC;
    }
    { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_ctx_live.insert(c); }
    *out = c;
    return PEDP_OK;
}

void pedp_ctx_destroy(pedp_ctx_t c) {
    if (!c) return;
    { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_ctx_live.erase(c); }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)pedp_comm_destroy(c);
    for (int k = 0; k < PEDP_MAX_SUB; ++k) {
        if (c->sub[k]) pedp_ctx_destroy(c->sub[k]);
        c->sub[k] = nullptr;
    }
    pedp_icp_drop_pending(c);   // (a pedp_icp_begin nobody ended: the stream has been waited for above)
    if (c->icp_graph) (void)hipGraphExecDestroy(c->icp_graph);
    for (hipGraphExec_t g : c->icp_bgraph)
        if (g) (void)hipGraphExecDestroy(g);
    c->ray_keys.release();
    c->ray_in.release();
    c->ray_out.release();
    c->ray_aux.release();
    c->ray_order.release();
    c->ray_rast.release();
    c->ops_in.release();
    c->ops_small.release();
    c->chain.release();
    c->cloud_pool.clear();
    c->sort_ws.release();
    if (c->rast_status) (void)hipHostFree(c->rast_status);
    c->icp_ws.release();
    c->proj.release();
    c->ops.release();
    c->proj_out.release();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->avg_host) (void)hipHostFree(c->avg_host);
    for (int k = 0; k < 2; ++k)
        if (c->stage[k]) (void)hipHostFree(c->stage[k]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->nn_ev0) (void)hipEventDestroy(c->nn_ev0);
    if (c->nn_ev1) (void)hipEventDestroy(c->nn_ev1);
    for (hipEvent_t e : c->nn_evs)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pedp_host_alloc(size_t bytes, void **out) {
    PEDP_REQUIRE(out, "pedp_host_alloc: null output");
    *out = nullptr;
    PEDP_HIP_CHECK(hipHostMalloc(out, bytes > 0 ? bytes : 1, hipHostMallocDefault));
    return PEDP_OK;
}

void pedp_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int pedp_ctx_synchronize(pedp_ctx_t c) {
    PEDP_REQUIRE(c, "pedp_ctx_synchronize: null context");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->stage_busy = false;
    return PEDP_OK;
}

// ---------------------------------------------------------------- clouds

}  // extern "C"

namespace {

constexpr int CS_BLOCKS = 512, CS_THREADS = 256;

// per-workgroup partial sums and bounds of a device-resident cloud (fixed tree: deterministic)
__global__ __launch_bounds__(CS_THREADS) void cloud_sum_kernel(const double *__restrict__ pts, int64_t N,
                                                               double *__restrict__ part /* CS_BLOCKS x 9 */) {
    __shared__ double sh[CS_THREADS / 64][9];
    const double big = 1.7976931348623157e308;
    double v[9] = {0, 0, 0, big, big, big, -big, -big, -big};
    for (int64_t i = (int64_t)blockIdx.x * CS_THREADS + threadIdx.x; i < N; i += (int64_t)CS_BLOCKS * CS_THREADS)
        for (int k = 0; k < 3; ++k) {
            const double x = pts[3 * i + k];
            v[k] += x;
            v[3 + k] = x < v[3 + k] ? x : v[3 + k];
            v[6 + k] = x > v[6 + k] ? x : v[6 + k];
        }
    for (int k = 0; k < 9; ++k) {
        double x = v[k];
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(x, off, 64);
            x = k < 3 ? x + o : (k < 6 ? (o < x ? o : x) : (o > x ? o : x));
        }
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int k = threadIdx.x;
        double x = sh[0][k];
        for (int w = 1; w < CS_THREADS / 64; ++w) {
            const double o = sh[w][k];
            x = k < 3 ? x + o : (k < 6 ? (o < x ? o : x) : (o > x ? o : x));
        }
        part[blockIdx.x * 9 + k] = x;
    }
}

// the box alone, folded on the device into the cloud's own six doubles: nothing comes back to the host
__global__ __launch_bounds__(64) void cloud_box_fold_kernel(const double *__restrict__ part /* CS_BLOCKS x 9 */, double *__restrict__ box) {
    const int lane = threadIdx.x;
    double lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = part[lane * 9 + 3 + k]; hi[k] = part[lane * 9 + 6 + k]; }
    for (int b = lane + 64; b < CS_BLOCKS; b += 64)
        for (int k = 0; k < 3; ++k) {
            const double l2 = part[b * 9 + 3 + k], h2 = part[b * 9 + 6 + k];
            lo[k] = l2 < lo[k] ? l2 : lo[k];
            hi[k] = h2 > hi[k] ? h2 : hi[k];
        }
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off >= 1; off >>= 1) {
            const double l2 = __shfl_xor(lo[k], off, 64), h2 = __shfl_xor(hi[k], off, 64);
            lo[k] = l2 < lo[k] ? l2 : lo[k];
            hi[k] = h2 > hi[k] ? h2 : hi[k];
        }
    if (lane == 0)
        for (int k = 0; k < 3; ++k) { box[k] = lo[k]; box[3 + k] = hi[k]; }
}

// largest |t'|_1 and |t'|_2^2 of the centred float32 coordinates (the NN filter's magnitudes)
__global__ __launch_bounds__(CS_THREADS) void cloud_norm_kernel(const double *__restrict__ pts, int64_t N, double cx, double cy,
                                                                double cz, float *__restrict__ part /* CS_BLOCKS x 2 */) {
    __shared__ float sh[CS_THREADS / 64][2];
    float n1m = 0.f, n2m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * CS_THREADS + threadIdx.x; i < N; i += (int64_t)CS_BLOCKS * CS_THREADS) {
        const float x = (float)(pts[3 * i] - cx), y = (float)(pts[3 * i + 1] - cy), z = (float)(pts[3 * i + 2] - cz);
        n1m = fmaxf(n1m, fabsf(x) + fabsf(y) + fabsf(z));
        n2m = fmaxf(n2m, x * x + y * y + z * z);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        n1m = fmaxf(n1m, __shfl_xor(n1m, off, 64));
        n2m = fmaxf(n2m, __shfl_xor(n2m, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = n1m; sh[threadIdx.x >> 6][1] = n2m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < CS_THREADS / 64; ++w) { n1m = fmaxf(n1m, sh[w][0]); n2m = fmaxf(n2m, sh[w][1]); }
        part[blockIdx.x * 2] = fmaxf(n1m, sh[0][0]);
        part[blockIdx.x * 2 + 1] = fmaxf(n2m, sh[0][1]);
    }
}

pedp_cloud_s *new_cloud(pedp_ctx_t c, int64_t N) {
    pedp_cloud_s *cl = new (std::nothrow) pedp_cloud_s();
    if (!cl) { pedp_set_error("pedp_cloud_create: out of host memory"); return nullptr; }
    cl->ctx = c;
    cl->device = c->device;
    cl->gen = pedp_next_generation();
    cl->N = N;
    return cl;
}

}  // namespace

// Centroid, bounding box and the filter's magnitudes of a cloud made from device memory, on the host: two
// reductions and two read-backs, paid when -- and only if -- the cloud is used as a registration target.
int pedp_cloud_host_stats(pedp_ctx_t c, pedp_cloud_s *cl) {
    if (cl->host_stats || cl->N == 0) { cl->host_stats = true; return PEDP_OK; }
    const int64_t N = cl->N;
    double part[CS_BLOCKS * 9];
    float fpart[CS_BLOCKS * 2];
    int st = c->ops.reserve(sizeof(double) * CS_BLOCKS * 9 + sizeof(float) * CS_BLOCKS * 2 + 512);
    if (st) return st;
    double *d_part = (double *)c->ops.ptr;
    float *d_fpart = (float *)(d_part + CS_BLOCKS * 9);
    hipLaunchKernelGGL(cloud_sum_kernel, dim3(CS_BLOCKS), dim3(CS_THREADS), 0, c->stream, (const double *)cl->pts, N, d_part);
    PEDP_HIP_CHECK(hipMemcpyAsync(part, d_part, sizeof(part), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    double sum[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) { cl->lo[k] = part[3 + k]; cl->hi[k] = part[6 + k]; }
    for (int b = 0; b < CS_BLOCKS; ++b)
        for (int k = 0; k < 3; ++k) {
            sum[k] += part[b * 9 + k];
            if (part[b * 9 + 3 + k] < cl->lo[k]) cl->lo[k] = part[b * 9 + 3 + k];
            if (part[b * 9 + 6 + k] > cl->hi[k]) cl->hi[k] = part[b * 9 + 6 + k];
        }
    for (int k = 0; k < 3; ++k) cl->centroid[k] = sum[k] / (double)N;
    hipLaunchKernelGGL(cloud_norm_kernel, dim3(CS_BLOCKS), dim3(CS_THREADS), 0, c->stream, (const double *)cl->pts, N,
                       cl->centroid[0], cl->centroid[1], cl->centroid[2], d_fpart);
    PEDP_HIP_CHECK(hipMemcpyAsync(fpart, d_fpart, sizeof(fpart), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    float Tn = 0.f, T2 = 0.f;
    for (int b = 0; b < CS_BLOCKS; ++b) { Tn = fmaxf(Tn, fpart[2 * b]); T2 = fmaxf(T2, fpart[2 * b + 1]); }
    cl->Tn = Tn * 1.0001f;
    cl->T2 = T2 * 1.0001f;
    cl->host_stats = true;
    return PEDP_OK;
}

extern "C" {

int pedp_cloud_create(pedp_ctx_t c, const double *pts, const double *normals, int64_t N,
                      pedp_cloud_t *out) {
    PEDP_REQUIRE(c && out, "pedp_cloud_create: null context/output");
    *out = nullptr;
    PEDP_REQUIRE(N >= 0 && N < (int64_t)0x7FFFFFF0, "pedp_cloud_create: N=%lld out of range", (long long)N);
    PEDP_REQUIRE(pts || N == 0, "pedp_cloud_create: null points");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    pedp_cloud_s *cl = new_cloud(c, N);
    if (!cl) return PEDP_ERR_ALLOC;
    if (N > 0) {
        double sum[3] = {0, 0, 0};
        for (int64_t i = 0; i < N; ++i) { sum[0] += pts[3 * i]; sum[1] += pts[3 * i + 1]; sum[2] += pts[3 * i + 2]; }
        for (int k = 0; k < 3; ++k) cl->centroid[k] = sum[k] / (double)N;
        for (int k = 0; k < 3; ++k) { cl->lo[k] = pts[k]; cl->hi[k] = pts[k]; }
        float Tn = 0.f, T2 = 0.f;
        for (int64_t i = 0; i < N; ++i) {
            for (int k = 0; k < 3; ++k) {
                if (pts[3 * i + k] < cl->lo[k]) cl->lo[k] = pts[3 * i + k];
                if (pts[3 * i + k] > cl->hi[k]) cl->hi[k] = pts[3 * i + k];
            }
            float x = (float)(pts[3 * i] - cl->centroid[0]), y = (float)(pts[3 * i + 1] - cl->centroid[1]),
                  z = (float)(pts[3 * i + 2] - cl->centroid[2]);
            float n1 = fabsf(x) + fabsf(y) + fabsf(z), n2 = x * x + y * y + z * z;
            if (n1 > Tn) Tn = n1;
            if (n2 > T2) T2 = n2;
        }
        cl->Tn = Tn * 1.0001f;
        cl->T2 = T2 * 1.0001f;
    }
    cl->host_stats = true;
    const size_t bytes = sizeof(double) * 3 * (size_t)(N > 0 ? N : 1);
    bool ok = (cl->pts = (double *)c->cloud_pool.take(bytes, &cl->pts_cap)) != nullptr;
    ok = ok && (cl->d_box = (double *)c->cloud_pool.take(64, &cl->box_cap)) != nullptr;
    if (ok && N > 0 && pedp_upload(c, cl->pts, pts, sizeof(double) * 3 * (size_t)N) != PEDP_OK) ok = false;
    if (ok && normals) {
        cl->has_normals = true;
        ok = (cl->normals = (double *)c->cloud_pool.take(bytes, &cl->normals_cap)) != nullptr;
        if (ok && N > 0 && pedp_upload(c, cl->normals, normals, sizeof(double) * 3 * (size_t)N) != PEDP_OK) ok = false;
    }
    if (ok) {  // the box on the device too (the spatial order reads it there)
        double *hb = (double *)((char *)c->pinned + 12288);
        for (int k = 0; k < 3; ++k) { hb[k] = cl->lo[k]; hb[3 + k] = cl->hi[k]; }
        ok = hipMemcpyAsync(cl->d_box, hb, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    }
    if (ok) ok = hipStreamSynchronize(c->stream) == hipSuccess;   // the caller's arrays (and the pinned block) are free again
    if (!ok) {
        if (!pedp_last_error()[0]) pedp_set_error("pedp_cloud_create: allocation or copy failed");
        pedp_cloud_destroy(cl);
        return PEDP_ERR_HIP;
    }
    *out = cl;
    return PEDP_OK;
}

// Cloud from device memory (a scene back-projected on the GPU): one device-to-device copy and the bounding box
// by two small kernels, all on the context's stream -- no read-back, buffers from the pool.  The call returns when the
// COPIES are complete (the source may be freed or overwritten); the box kernels may still be running.
int pedp_cloud_create_device(pedp_ctx_t c, const double *d_pts, const double *d_normals, int64_t N, pedp_cloud_t *out) {
    PEDP_REQUIRE(c && out, "pedp_cloud_create_device: null context/output");
    *out = nullptr;
    PEDP_REQUIRE(N >= 0 && N < (int64_t)0x7FFFFF00, "pedp_cloud_create_device: N out of range");
    PEDP_REQUIRE(d_pts || N == 0, "pedp_cloud_create_device: null points");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    pedp_cloud_s *cl = new_cloud(c, N);
    if (!cl) return PEDP_ERR_ALLOC;
    const size_t bytes = sizeof(double) * 3 * (size_t)(N > 0 ? N : 1);
    hipError_t e = hipSuccess;
    bool ok = (cl->pts = (double *)c->cloud_pool.take(bytes, &cl->pts_cap)) != nullptr;
    ok = ok && (cl->d_box = (double *)c->cloud_pool.take(64, &cl->box_cap)) != nullptr;
    if (ok && N > 0) e = hipMemcpyAsync(cl->pts, d_pts, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToDevice, c->stream);
    if (ok && e == hipSuccess && d_normals) {
        cl->has_normals = true;
        ok = (cl->normals = (double *)c->cloud_pool.take(bytes, &cl->normals_cap)) != nullptr;
        if (ok && N > 0) e = hipMemcpyAsync(cl->normals, d_normals, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToDevice, c->stream);
    }
    // the caller's arrays are free again when the call returns: the host waits for the copies (an event behind them; a
    // few microseconds, no read-back) -- torch hands a freed tensor's block to the next allocation on ITS stream, which
    // this context's stream is not ordered with
    if (ok && e == hipSuccess && N > 0) e = hipEventRecord(c->ev_copy, c->stream);
    const bool wait_copies = ok && e == hipSuccess && N > 0;
    if (ok && e == hipSuccess && N > 0) {
        const int st = c->sort_ws.reserve(sizeof(double) * CS_BLOCKS * 9 + 512);   // (the order's scratch: next in line on this stream)
        if (st) { pedp_cloud_destroy(cl); return st; }
        double *d_part = (double *)c->sort_ws.ptr;
        hipLaunchKernelGGL(cloud_sum_kernel, dim3(CS_BLOCKS), dim3(CS_THREADS), 0, c->stream, (const double *)cl->pts, N, d_part);
        hipLaunchKernelGGL(cloud_box_fold_kernel, dim3(1), dim3(64), 0, c->stream, (const double *)d_part, cl->d_box);
        e = hipGetLastError();
    } else if (N == 0) {
        cl->host_stats = true;
    }
    if (wait_copies && e == hipSuccess) e = hipEventSynchronize(c->ev_copy);   // (the box kernels are already queued behind)
    if (!ok || e != hipSuccess) {
        pedp_set_error("pedp_cloud_create_device: %s", ok ? hipGetErrorString(e) : "allocation failed");
        pedp_cloud_destroy(cl);
        return PEDP_ERR_HIP;
    }
    *out = cl;
    return PEDP_OK;
}

void pedp_cloud_destroy(pedp_cloud_t cl) {
    if (!cl) return;
    (void)hipSetDevice(cl->device);
    // points, normals, order, chunk spheres and box go back to the context's pool (the next frame's cloud takes
    // them: reuse is ordered by the stream); everything else, and all of it if the context is gone, is freed
    // the context's liveness is checked and its pool used under ONE lock (a context destroyed in between, or one
    // re-created at the same address for another device, never gets these buffers)
    std::lock_guard<std::mutex> live(g_ctx_mutex);
    pedp_pool *pool = g_ctx_live.count(cl->ctx) != 0 && cl->ctx->device == cl->device ? &cl->ctx->cloud_pool : nullptr;
    auto drop = [&](void *p, size_t cap) {
        if (!p) return;
        if (pool && cap) pool->give(p, cap);
        else (void)hipFree(p);
    };
    drop(cl->pts, cl->pts_cap);
    drop(cl->normals, cl->normals_cap);
    drop(cl->perm, cl->perm_cap);
    drop(cl->chunk_sph, cl->sph_cap);
    drop(cl->d_box, cl->box_cap);
    if (cl->tgt4) (void)hipFree(cl->tgt4);
    if (cl->tile_sph) (void)hipFree(cl->tile_sph);
    if (cl->tile_sph4) (void)hipFree(cl->tile_sph4);
    if (cl->tile_sphw) (void)hipFree(cl->tile_sphw);
    if (cl->tgt_s) (void)hipFree(cl->tgt_s);
    if (cl->tgt_bf) (void)hipFree(cl->tgt_bf);
    delete cl;
}

int pedp_cloud_size(pedp_cloud_t cl, int64_t *N, int *has_normals) {
    PEDP_REQUIRE(cl, "pedp_cloud_size: null cloud");
    if (N) *N = cl->N;
    if (has_normals) *has_normals = cl->has_normals ? 1 : 0;
    return PEDP_OK;
}

}  // extern "C"
