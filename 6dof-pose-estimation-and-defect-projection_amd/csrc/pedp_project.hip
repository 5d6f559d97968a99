// Fused defect projection: pixel selection, ray generation, closest hit and the
// compaction of the hits of one heat map, all on the device.  Mirrors
//   heatmap_to_points          src/defect_projection.py:165-179   (np.where order = row-major)
//   compute_rays               src/defect_projection.py:196-223   (float64 normalised directions)
//   intersect_rays_with_mesh   src/defect_projection.py:225-266   (float32 cast, closest hit,
//                                                                  t != inf filter, o + d * t in f64)
// The sweep itself is pedp_raycast (pedp_ray.hip) on device-resident rays.
//
// Both compactions keep the input order, so they are a count / scan / scatter over fixed
// 2048-element blocks: one count kernel, one single-workgroup scan of the block totals, one scatter
// that redoes the block-local scan in LDS.  All of it is HBM-bound streaming over at most
// width * height elements (8 B per pixel in, 24 B per selected ray out).
#include "pedp_internal.h"

namespace {

constexpr int CP_THREADS = 256;
constexpr int CP_PER_THREAD = 8;
constexpr int CP_BLOCK = CP_THREADS * CP_PER_THREAD;

struct Cam {
    double fx, fy, cx, cy;
    int width;
};

// MODE 0: element i is pixel i of the heat map, kept when heatmap[i] > threshold
// MODE 1: element i is selected ray i, kept when its t_hit is finite (t != inf)
template <int MODE, typename H>
__device__ __forceinline__ bool keep(const H *heat, const float *t, double threshold, int64_t i) {
    if (MODE == 0) return (double)heat[i] > threshold;
    return t[i] != __builtin_inff();
}

__device__ __forceinline__ int block_exclusive_scan(int v, int *lds, int *total) {
    // 256 threads: wave-level inclusive scan with DPP-free shuffles, then across the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}

// MODE 1 with sel != nullptr also takes the smallest and largest intensity of the block's hits (one pair per block;
// min / max are order-free) for the colour normalisation
template <int MODE, typename H>
__global__ __launch_bounds__(CP_THREADS) void compact_count_kernel(const H *__restrict__ heat,
                                                                   const float *__restrict__ t, double threshold,
                                                                   int64_t n, int *__restrict__ block_tot,
                                                                   const int *__restrict__ sel, double *__restrict__ block_mm) {
    __shared__ int lds[4];
    __shared__ double mm[2][4];
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_PER_THREAD;
    int cnt = 0;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int k = 0; k < CP_PER_THREAD; ++k)
        if (base + k < n && keep<MODE>(heat, t, threshold, base + k)) {
            ++cnt;
            if (MODE == 1 && sel) {
                const double v = (double)heat[sel[base + k]];
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
        }
    int total;
    (void)block_exclusive_scan(cnt, lds, &total);
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
    if (MODE == 1 && sel) {
        for (int off = 32; off >= 1; off >>= 1) {
            const double a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        if ((threadIdx.x & 63) == 0) { mm[0][threadIdx.x >> 6] = lo; mm[1][threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) { lo = mm[0][w] < lo ? mm[0][w] : lo; hi = mm[1][w] > hi ? mm[1][w] : hi; }
            block_mm[2 * blockIdx.x] = lo;
            block_mm[2 * blockIdx.x + 1] = hi;
        }
    }
}
__global__ __launch_bounds__(256) void minmax_fold_kernel(const double *__restrict__ block_mm, int n_blocks, double *__restrict__ out) {
    __shared__ double mm[2][256];
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int b = threadIdx.x; b < n_blocks; b += 256) {
        lo = block_mm[2 * b] < lo ? block_mm[2 * b] : lo;
        hi = block_mm[2 * b + 1] > hi ? block_mm[2 * b + 1] : hi;
    }
    mm[0][threadIdx.x] = lo;
    mm[1][threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256; ++w) { lo = mm[0][w] < lo ? mm[0][w] : lo; hi = mm[1][w] > hi ? mm[1][w] : hi; }
        out[0] = lo;
        out[1] = hi;
    }
}

// exclusive scan of the block totals in place; the grand total goes to *out_total
__global__ __launch_bounds__(1024) void compact_scan_kernel(int *__restrict__ block_tot, int n_blocks,
                                                            int *__restrict__ out_total) {
    __shared__ int part[1024];
    const int per = (n_blocks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, n_blocks);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_tot[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { int v = part[i]; part[i] = run; run += v; }
        *out_total = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { int v = block_tot[i]; block_tot[i] = run; run += v; }
}

__device__ __forceinline__ void pixel_direction(const Cam &cam, int64_t pix, double d[3]) {
    const int64_t y = pix / cam.width, x = pix - y * cam.width;
    const double xn = __ddiv_rn(__dsub_rn((double)x, cam.cx), cam.fx);
    const double yn = __ddiv_rn(__dsub_rn((double)y, cam.cy), cam.fy);
    const double len = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(xn, xn), __dmul_rn(yn, yn)), 1.0));
    d[0] = __ddiv_rn(xn, len);
    d[1] = __ddiv_rn(yn, len);
    d[2] = __ddiv_rn(1.0, len);
}

// stage 1: selected pixels -> pixel list + float32 [origin | direction] rows
template <typename H>
__global__ __launch_bounds__(CP_THREADS) void select_scatter_kernel(const H *__restrict__ heat, double threshold,
                                                                    int64_t n, const int *__restrict__ block_off, Cam cam,
                                                                    float ox, float oy, float oz, int *__restrict__ sel,
                                                                    float *__restrict__ rays6) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_PER_THREAD;
    unsigned flags = 0;
    for (int k = 0; k < CP_PER_THREAD; ++k)
        if (base + k < n && keep<0, H>(heat, nullptr, threshold, base + k)) flags |= 1u << k;
    int pos = block_off[blockIdx.x] + block_exclusive_scan(__popc(flags), lds, nullptr);
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        if (!(flags >> k & 1u)) continue;
        double d[3];
        pixel_direction(cam, base + k, d);
        sel[pos] = (int)(base + k);
        float *r = rays6 + 6 * (int64_t)pos;
        r[0] = ox; r[1] = oy; r[2] = oz;
        r[3] = (float)d[0]; r[4] = (float)d[1]; r[5] = (float)d[2];
        ++pos;
    }
}

// stage 2: rays that hit -> points (o + d * t in float64), intensities, pixels, triangle ids
struct Post { double m[12]; int on; };
template <typename H>
__global__ __launch_bounds__(CP_THREADS) void hit_scatter_kernel(const float *__restrict__ t, const uint32_t *__restrict__ id,
                                                                 const int *__restrict__ sel, const H *__restrict__ heat,
                                                                 int64_t n, const int *__restrict__ block_off, Cam cam,
                                                                 double ox, double oy, double oz, double *__restrict__ points,
                                                                 double *__restrict__ intens, int32_t *__restrict__ pixels,
                                                                 uint32_t *__restrict__ prim, Post post,
                                                                 const double *__restrict__ lut, const double *__restrict__ mm,
                                                                 double *__restrict__ colors) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_PER_THREAD;
    unsigned flags = 0;
    for (int k = 0; k < CP_PER_THREAD; ++k)
        if (base + k < n && keep<1, H>(nullptr, t, 0.0, base + k)) flags |= 1u << k;
    int64_t pos = block_off[blockIdx.x] + block_exclusive_scan(__popc(flags), lds, nullptr);
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        if (!(flags >> k & 1u)) continue;
        const int pix = sel[base + k];
        double d[3];
        pixel_direction(cam, pix, d);
        const double th = (double)t[base + k];
        const double px = __dadd_rn(ox, __dmul_rn(d[0], th)), py = __dadd_rn(oy, __dmul_rn(d[1], th)),
                     pz = __dadd_rn(oz, __dmul_rn(d[2], th));
        if (post.on) {
            for (int r = 0; r < 3; ++r)
                points[3 * pos + r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(post.m[4 * r], px), __dmul_rn(post.m[4 * r + 1], py)),
                                                          __dmul_rn(post.m[4 * r + 2], pz)), post.m[4 * r + 3]);
        } else {
            points[3 * pos] = px;
            points[3 * pos + 1] = py;
            points[3 * pos + 2] = pz;
        }
        const double inten = (double)heat[pix];
        intens[pos] = inten;
        if (colors) {  // jet(min-max normalised intensity): index trunc(s * 256) clipped to 0..255, NaN -> black
            const double sc = __dmul_rn(__ddiv_rn(__dsub_rn(inten, mm[0]), __dsub_rn(mm[1], mm[0])), 256.0);
            if (sc != sc) {
                colors[3 * pos] = colors[3 * pos + 1] = colors[3 * pos + 2] = 0.0;
            } else {
                const int e = sc <= 0.0 ? 0 : (sc >= 255.0 ? 255 : (int)sc);
                colors[3 * pos] = lut[3 * e];
                colors[3 * pos + 1] = lut[3 * e + 1];
                colors[3 * pos + 2] = lut[3 * e + 2];
            }
        }
        if (pixels) { pixels[2 * pos] = pix % cam.width; pixels[2 * pos + 1] = pix / cam.width; }
        if (prim) prim[pos] = id[base + k];
        ++pos;
    }
}

inline size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" int pedp_project_heatmap(pedp_ctx_t c, pedp_mesh_t mesh, const pedp_pinhole *cam, const double *heatmap,
                                    double threshold, const double origin[3], int mem, int64_t capacity,
                                    double *points, double *intensities, int32_t *pixels, uint32_t *prim_id,
                                    int64_t *n_rays, int64_t *n_hits) {
    return pedp_project_heatmap_ex(c, mesh, cam, heatmap, threshold, origin, mem, capacity, points, intensities, pixels, prim_id, n_rays,
                                   n_hits, nullptr);
}

namespace {
template <typename H>
int project_impl(pedp_ctx_t c, pedp_mesh_t mesh, const pedp_pinhole *cam, const H *heatmap, int heat_mem, double threshold,
                 const double origin[3], int mem, int64_t capacity, double *points, double *intensities, int32_t *pixels,
                 uint32_t *prim_id, int64_t *n_rays, int64_t *n_hits, const double *jet_lut, double *colors, const double *post_T);
}

extern "C" int pedp_project_heatmap_ex(pedp_ctx_t c, pedp_mesh_t mesh, const pedp_pinhole *cam, const void *heatmap,
                                       double threshold, const double origin[3], int mem, int64_t capacity,
                                       double *points, double *intensities, int32_t *pixels, uint32_t *prim_id,
                                       int64_t *n_rays, int64_t *n_hits, const pedp_project_opts *opts) {
    PEDP_REQUIRE(c && mesh && cam && origin && n_rays && n_hits, "pedp_project_heatmap: null argument");
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "pedp_project_heatmap: bad mem flag %d", mem);
    const int heat_mem = opts ? opts->heat_mem : mem;
    PEDP_REQUIRE(heat_mem == PEDP_HOST || heat_mem == PEDP_DEVICE, "pedp_project_heatmap: bad heat_mem flag %d", heat_mem);
    PEDP_REQUIRE(!opts || (opts->jet_lut == nullptr) == (opts->colors == nullptr), "pedp_project_heatmap: jet_lut and colors go together");
    const double *lut = opts ? opts->jet_lut : nullptr, *post = opts ? opts->post : nullptr;
    double *colors = opts ? opts->colors : nullptr;
    if (opts && opts->heat_f32)
        return project_impl<float>(c, mesh, cam, (const float *)heatmap, heat_mem, threshold, origin, mem, capacity, points, intensities,
                                   pixels, prim_id, n_rays, n_hits, lut, colors, post);
    return project_impl<double>(c, mesh, cam, (const double *)heatmap, heat_mem, threshold, origin, mem, capacity, points, intensities,
                                pixels, prim_id, n_rays, n_hits, lut, colors, post);
}

namespace {
template <typename H>
int project_impl(pedp_ctx_t c, pedp_mesh_t mesh, const pedp_pinhole *cam, const H *heatmap, int heat_mem, double threshold,
                 const double origin[3], int mem, int64_t capacity, double *points, double *intensities, int32_t *pixels,
                 uint32_t *prim_id, int64_t *n_rays, int64_t *n_hits, const double *jet_lut, double *colors, const double *post_T) {
    PEDP_REQUIRE(mesh->ctx == c, "pedp_project_heatmap: mesh belongs to another context");
    PEDP_REQUIRE(cam->width >= 0 && cam->height >= 0, "pedp_project_heatmap: negative image size");
    const int64_t n_pix = (int64_t)cam->width * cam->height;
    PEDP_REQUIRE(n_pix < (int64_t)1 << 31, "pedp_project_heatmap: image too large");
    PEDP_REQUIRE(capacity >= 0, "pedp_project_heatmap: negative capacity");
    *n_rays = 0;
    *n_hits = 0;
    if (n_pix == 0) return PEDP_OK;
    PEDP_REQUIRE(heatmap, "pedp_project_heatmap: null heat map");
    PEDP_REQUIRE(cam->fx != 0.0 && cam->fy != 0.0, "pedp_project_heatmap: zero focal length");
    PEDP_HIP_CHECK(hipSetDevice(c->device));

    // stage-1 scratch: [heat map copy (host mode)][block offsets][pixel list][counters]
    const int n_blocks1 = (int)((n_pix + CP_BLOCK - 1) / CP_BLOCK);
    const size_t sz_heat = heat_mem == PEDP_HOST ? a256(sizeof(H) * (size_t)n_pix) : 0;
    const size_t sz_boff = a256(sizeof(int) * (size_t)n_blocks1);
    const size_t sz_sel = a256(sizeof(int) * (size_t)n_pix);
    const size_t sz_mm = a256(sizeof(double) * 2 * ((size_t)n_blocks1 + 1)), sz_lut = a256(sizeof(double) * 768);
    int st = c->proj.reserve(sz_heat + 2 * sz_boff + sz_sel + sz_mm + sz_lut + 256);
    if (st) return st;
    char *b = (char *)c->proj.ptr;
    const H *d_heat = heatmap;
    if (heat_mem == PEDP_HOST) {
        { int up_ = pedp_upload(c, b, heatmap, sizeof(H) * (size_t)n_pix); if (up_) return up_; }
        d_heat = (const H *)b;
    }
    int *boff1 = (int *)(b + sz_heat);
    int *boff2 = (int *)(b + sz_heat + sz_boff);
    int *sel = (int *)(b + sz_heat + 2 * sz_boff);
    double *block_mm = (double *)(b + sz_heat + 2 * sz_boff + sz_sel);  // per-block (min, max) of the hits' intensities, then the pair itself
    double *d_lut = (double *)(b + sz_heat + 2 * sz_boff + sz_sel + sz_mm);
    int *counters = (int *)(b + sz_heat + 2 * sz_boff + sz_sel + sz_mm + sz_lut);
    if (jet_lut) {  // 6 KB through the pinned block (behind the counters' words)
        double *h_lut = (double *)((char *)c->pinned + 69632);
        if (c->stage_busy) PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        memcpy(h_lut, jet_lut, sizeof(double) * 768);
        PEDP_HIP_CHECK(hipMemcpyAsync(d_lut, h_lut, sizeof(double) * 768, hipMemcpyHostToDevice, c->stream));
    }
    int *h_counters = (int *)((char *)c->pinned + 8192);

    const Cam cm{cam->fx, cam->fy, cam->cx, cam->cy, cam->width};
    hipLaunchKernelGGL((compact_count_kernel<0, H>), dim3(n_blocks1), dim3(CP_THREADS), 0, c->stream, d_heat, (const float *)nullptr,
                       threshold, n_pix, boff1, (const int *)nullptr, (double *)nullptr);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, c->stream, boff1, n_blocks1, counters);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipMemcpyAsync(h_counters, counters, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int64_t n_sel = h_counters[0];
    *n_rays = n_sel;
    if (n_sel == 0) return PEDP_OK;

    // stage-2 scratch: [rays6][t_hit][prim_id] (+ compacted outputs in host mode)
    const size_t sz_rays = a256(sizeof(float) * 6 * (size_t)n_sel);
    const size_t sz_t = a256(sizeof(float) * (size_t)n_sel);
    st = c->proj_out.reserve(sz_rays + 2 * sz_t);
    if (st) return st;
    char *o = (char *)c->proj_out.ptr;
    float *rays6 = (float *)o;
    float *t_hit = (float *)(o + sz_rays);
    uint32_t *ids = (uint32_t *)(o + sz_rays + sz_t);
    hipLaunchKernelGGL(select_scatter_kernel<H>, dim3(n_blocks1), dim3(CP_THREADS), 0, c->stream, d_heat, threshold, n_pix,
                       boff1, cm, (float)origin[0], (float)origin[1], (float)origin[2], sel, rays6);
    PEDP_HIP_CHECK(hipGetLastError());
    int rc = pedp_raycast(c, mesh, rays6, n_sel, PEDP_DEVICE, t_hit, ids, nullptr);
    if (rc) return rc;

    const int n_blocks2 = (int)((n_sel + CP_BLOCK - 1) / CP_BLOCK);
    hipLaunchKernelGGL((compact_count_kernel<1, H>), dim3(n_blocks2), dim3(CP_THREADS), 0, c->stream, d_heat, (const float *)t_hit, 0.0,
                       n_sel, boff2, colors ? (const int *)sel : (const int *)nullptr, block_mm);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, c->stream, boff2, n_blocks2, counters + 1);
    if (colors)
        hipLaunchKernelGGL(minmax_fold_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)block_mm, n_blocks2, block_mm + 2 * (size_t)n_blocks1);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipMemcpyAsync(h_counters + 1, counters + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int64_t n_hit = h_counters[1];
    *n_hits = n_hit;
    if (n_hit == 0) return PEDP_OK;
    PEDP_REQUIRE(n_hit <= capacity, "pedp_project_heatmap: %lld hits exceed the output capacity %lld",
                 (long long)n_hit, (long long)capacity);
    PEDP_REQUIRE(points && intensities, "pedp_project_heatmap: null output arrays");

    double *d_pts = points, *d_int = intensities, *d_col = colors;
    int32_t *d_pix = pixels;
    uint32_t *d_prim = prim_id;
    pedp_scratch &ob = c->ray_in;  // free here: pedp_raycast ran in device mode
    const size_t s_pts = a256(sizeof(double) * 3 * (size_t)n_hit), s_int = a256(sizeof(double) * (size_t)n_hit);
    const size_t s_col = colors ? s_pts : 0, s_pix = a256(sizeof(int32_t) * 2 * (size_t)n_hit), s_prim = a256(sizeof(uint32_t) * (size_t)n_hit);
    if (mem == PEDP_HOST) {
        st = ob.reserve(s_pts + s_int + s_col + s_pix + s_prim);
        if (st) return st;
        char *q = (char *)ob.ptr;
        d_pts = (double *)q;
        d_int = (double *)(q + s_pts);
        d_col = colors ? (double *)(q + s_pts + s_int) : nullptr;
        d_pix = pixels ? (int32_t *)(q + s_pts + s_int + s_col) : nullptr;
        d_prim = prim_id ? (uint32_t *)(q + s_pts + s_int + s_col + s_pix) : nullptr;
    }
    Post post;
    post.on = post_T ? 1 : 0;
    for (int k = 0; k < 12; ++k) post.m[k] = post_T ? post_T[k] : 0.0;
    hipLaunchKernelGGL(hit_scatter_kernel<H>, dim3(n_blocks2), dim3(CP_THREADS), 0, c->stream, (const float *)t_hit, (const uint32_t *)ids,
                       (const int *)sel, d_heat, n_sel, (const int *)boff2, cm, origin[0], origin[1], origin[2], d_pts, d_int, d_pix, d_prim,
                       post, (const double *)d_lut, (const double *)(block_mm + 2 * (size_t)n_blocks1), d_col);
    PEDP_HIP_CHECK(hipGetLastError());
    if (mem == PEDP_HOST) {  // the results lie in one block: one copy, one wait
        const size_t off[5] = {0, s_pts, s_pts + s_int, s_pts + s_int + s_col, s_pts + s_int + s_col + s_pix};
        void *const dst[5] = {points, intensities, colors, pixels, prim_id};
        const size_t bytes[5] = {sizeof(double) * 3 * (size_t)n_hit, sizeof(double) * (size_t)n_hit, sizeof(double) * 3 * (size_t)n_hit,
                                 sizeof(int32_t) * 2 * (size_t)n_hit, sizeof(uint32_t) * (size_t)n_hit};
        int last = 1;
        for (int k = 2; k < 5; ++k)
            if (dst[k]) last = k;
        int dn_ = pedp_download_parts(c, ob.ptr, off[last] + bytes[last], 5, off, dst, bytes);
        if (dn_) return dn_;
    }
    return PEDP_OK;
}
}  // namespace
