// Ray / triangle closest-hit sweep for gfx950 (MI355X).
//
// Replaces RaycastingScene.add_triangles + cast_rays as called by
// src/defect_projection.py:245-256 (one camera ray per heat-map pixel against every
// triangle of the posed mesh).  Arithmetic contract: oracle/ray.c (Moeller-Trumbore,
// division-deferred, every operation one fp32 rounding in a fixed order) -- results are
// bit-identical to it: t_hit, primitive ids and (u, v).
//
// Mapping (ray-per-lane variant): one ray per lane, its origin/direction in VGPRs for
// the whole sweep.  The triangle index is wave-uniform, so a triangle record is fetched
// with scalar loads (s_load_dwordx8 + x4 of a 48-B record) and feeds the VALU as SGPR
// operands: the scalar cache is the broadcast, no LDS traffic and no VGPRs are spent on
// triangle data.  The grid is (ray blocks) x (triangle chunks); chunk c is always served
// by workgroups with blockIdx % 8 == c % 8, i.e. by one XCD, so each XCD's L2 only ever
// holds its own 1/8 of the triangle buffer.  Partial results meet in one packed
// 64-bit atomicMin per ray and chunk: key = (bits(t) << 32) | triangle id, which orders
// by t (t >= 0, so IEEE bits are monotone) and then by triangle index -- exactly the
// oracle's tie rule, independent of execution order.
//
// Variant 2 (triangle-per-lane) is for few rays against a big mesh: lanes own
// consecutive triangles (coalesced 16-B loads of the record buffer), the ray is
// wave-uniform, and the packed key is min-reduced across the 64 lanes with DPP/shuffles.
#include "pedp_internal.h"

namespace {

constexpr unsigned long long KEY_MISS = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ float mulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fmar(float a, float b, float c) { return __fmaf_rn(a, b, c); }
__device__ __forceinline__ float subr(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float addr(float a, float b) { return __fadd_rn(a, b); }

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmar(az, bz, fmar(ay, by, mulr(ax, bx)));
}

struct Ray {
    float ox, oy, oz, dx, dy, dz;
};

struct MT {
    float det, un, vn, tn;
};

// The oracle's operation order (oracle/ray.c header comment), verbatim.
__device__ __forceinline__ MT mt_eval(const Ray &r, float v0x, float v0y, float v0z, float e1x,
                                      float e1y, float e1z, float e2x, float e2y, float e2z) {
    MT m;
    float px = fmar(r.dy, e2z, -mulr(r.dz, e2y));
    float py = fmar(r.dz, e2x, -mulr(r.dx, e2z));
    float pz = fmar(r.dx, e2y, -mulr(r.dy, e2x));
    m.det = dot3(e1x, e1y, e1z, px, py, pz);
    float sx = subr(r.ox, v0x), sy = subr(r.oy, v0y), sz = subr(r.oz, v0z);
    m.un = dot3(sx, sy, sz, px, py, pz);
    float qx = fmar(sy, e1z, -mulr(sz, e1y));
    float qy = fmar(sz, e1x, -mulr(sx, e1z));
    float qz = fmar(sx, e1y, -mulr(sy, e1x));
    m.vn = dot3(r.dx, r.dy, r.dz, qx, qy, qz);
    m.tn = dot3(e2x, e2y, e2z, qx, qy, qz);
    return m;
}

// Exact acceptance predicate of the oracle (det != 0 is checked next to it).
__device__ __forceinline__ bool mt_pass(const MT &m) {
    unsigned sg = __float_as_uint(m.det) & 0x80000000u;
    float U = __uint_as_float(__float_as_uint(m.un) ^ sg);
    float V = __uint_as_float(__float_as_uint(m.vn) ^ sg);
    float T = __uint_as_float(__float_as_uint(m.tn) ^ sg);
    float W = addr(U, V);
    return (U >= 0.0f) & (V >= 0.0f) & (T >= 0.0f) & (W <= fabsf(m.det));
}

// Branch-free score for the hot loop: score >= 0 <=> mt_pass for all non-NaN operands.
// With s = sign(det): U,V,T >= 0 <=> min3(un,vn,tn) >= 0 (s > 0) or max3(un,vn,tn) <= 0
// (s < 0); U + V == (un + vn)^s exactly, so once U,V >= 0 holds, U + V <= |det| <=>
// fl(|det| - |un + vn|) >= 0 (a difference has the sign of the exact difference).  With
// a NaN operand the score may pass where mt_pass does not, never the other way round:
// the accept path re-applies mt_pass, so the score only has to be a superset.
__device__ __forceinline__ float mt_score(const MT &m) {
    float lo = fminf(fminf(m.un, m.vn), m.tn);
    float hi = fmaxf(fmaxf(m.un, m.vn), m.tn);
    float sd = (__float_as_int(m.det) < 0) ? -hi : lo;
    float R = subr(fabsf(m.det), fabsf(addr(m.un, m.vn)));
    return fminf(sd, R);
}

__device__ __forceinline__ unsigned long long mt_key(const MT &m, unsigned id) {
    unsigned sg = __float_as_uint(m.det) & 0x80000000u;
    float T = __uint_as_float(__float_as_uint(m.tn) ^ sg);
    float t = __fdiv_rn(T, fabsf(m.det));
    unsigned tb = __float_as_uint(t) & 0x7FFFFFFFu;
    return ((unsigned long long)tb << 32) | (unsigned long long)id;
}

// ------------------------------------------------------------------ mesh setup
__global__ void tri_setup_kernel(const float *__restrict__ verts, const uint32_t *__restrict__ tris,
                                 int64_t F, int64_t F_padded, float *__restrict__ rec) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F_padded) return;
    float *r = rec + f * PEDP_TRI_STRIDE;
    if (f >= F) {
#pragma unroll
        for (int k = 0; k < PEDP_TRI_STRIDE; ++k) r[k] = 0.0f;  // det == 0: can never be hit
        return;
    }
    const float *a = verts + 3 * (int64_t)tris[3 * f + 0];
    const float *b = verts + 3 * (int64_t)tris[3 * f + 1];
    const float *c = verts + 3 * (int64_t)tris[3 * f + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r[k] = a[k];
        r[3 + k] = subr(b[k], a[k]);
        r[6 + k] = subr(c[k], a[k]);
    }
    r[9] = r[10] = r[11] = 0.0f;
}

// ------------------------------------------------------------------ sweep, ray per lane
constexpr int RPL_BLOCK = 256;
constexpr int RPL_UNROLL = 4;

__global__ __launch_bounds__(RPL_BLOCK) void ray_sweep_rpl_kernel(
    const float4 *__restrict__ tri, int groups_total, int groups_per_chunk, int n_chunks,
    const float *__restrict__ rays6, int64_t N, unsigned long long *__restrict__ keys) {
    const int b = blockIdx.x;
    const int chunk = b % n_chunks;  // n_chunks % 8 == 0, so chunk % 8 == b % 8: one XCD per chunk
    const int64_t rb = b / n_chunks;
    const int64_t ray = rb * RPL_BLOCK + threadIdx.x;
    const int64_t rl = ray < N ? ray : N - 1;  // tail lanes re-run the last ray, never store
    Ray r;
    r.ox = rays6[6 * rl + 0]; r.oy = rays6[6 * rl + 1]; r.oz = rays6[6 * rl + 2];
    r.dx = rays6[6 * rl + 3]; r.dy = rays6[6 * rl + 4]; r.dz = rays6[6 * rl + 5];

    int g0 = chunk * groups_per_chunk;
    int g1 = g0 + groups_per_chunk;
    if (g1 > groups_total) g1 = groups_total;
    unsigned long long best = KEY_MISS;
    if (g0 < g1) {
        // Software pipeline: the records of group g+1 are requested (scalar loads into
        // SGPRs) before group g is evaluated, so the scalar-cache latency hides behind
        // ~4 x 35 VALU instructions.  The accept path (division, 64-bit min) is rare:
        // one wave-level branch per group.
        float4 cur[RPL_UNROLL * 3], nxt[RPL_UNROLL * 3];
        {
            const float4 *t = tri + (size_t)g0 * (RPL_UNROLL * 3);
#pragma unroll
            for (int k = 0; k < RPL_UNROLL * 3; ++k) nxt[k] = t[k];
        }
        for (int g = g0; g < g1; ++g) {
#pragma unroll
            for (int k = 0; k < RPL_UNROLL * 3; ++k) cur[k] = nxt[k];
            {
                int gn = g + 1 < g1 ? g + 1 : g;
                const float4 *t = tri + (size_t)gn * (RPL_UNROLL * 3);
#pragma unroll
                for (int k = 0; k < RPL_UNROLL * 3; ++k) nxt[k] = t[k];
            }
            MT m[RPL_UNROLL];
#pragma unroll
            for (int k = 0; k < RPL_UNROLL; ++k) {
                float4 a = cur[3 * k + 0], bq = cur[3 * k + 1], c = cur[3 * k + 2];
                m[k] = mt_eval(r, a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x);
            }
            static_assert(RPL_UNROLL == 4, "accept test below is written for 4 records");
            const float sc = fmaxf(fmaxf(mt_score(m[0]), mt_score(m[1])), fmaxf(mt_score(m[2]), mt_score(m[3])));
            if (__builtin_amdgcn_ballot_w64(sc >= 0.0f) != 0) {  // wave-uniform, rarely taken
#pragma unroll
                for (int k = 0; k < RPL_UNROLL; ++k) {
                    if (mt_pass(m[k]) && m[k].det != 0.0f) {
                        unsigned long long key = mt_key(m[k], (unsigned)(g * RPL_UNROLL + k));
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
    if (ray < N && best != KEY_MISS) atomicMin(&keys[ray], best);
}

// ------------------------------------------------------------------ sweep, triangle per lane
// One workgroup = 4 waves; each wave takes TPL_RAYS wave-uniform rays and strides over a
// triangle chunk with lanes on consecutive triangles.
constexpr int TPL_BLOCK = 256;
constexpr int TPL_RAYS = 4;

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFull), off, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(TPL_BLOCK) void ray_sweep_tpl_kernel(
    const float4 *__restrict__ tri, int64_t F_padded, int tris_per_chunk, int n_chunks,
    const float *__restrict__ rays6, int64_t N, unsigned long long *__restrict__ keys) {
    const int b = blockIdx.x;
    const int chunk = b % n_chunks;  // n_chunks % 8 == 0, so chunk % 8 == b % 8: one XCD per chunk
    const int64_t rb = b / n_chunks;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray0 = (rb * (TPL_BLOCK / 64) + wave) * TPL_RAYS;
    if (ray0 >= N) return;  // wave-uniform
    Ray r[TPL_RAYS];
#pragma unroll
    for (int k = 0; k < TPL_RAYS; ++k) {
        int64_t rl = ray0 + k < N ? ray0 + k : N - 1;
        // wave-uniform address: the compiler keeps these in SGPRs
        r[k].ox = rays6[6 * rl + 0]; r[k].oy = rays6[6 * rl + 1]; r[k].oz = rays6[6 * rl + 2];
        r[k].dx = rays6[6 * rl + 3]; r[k].dy = rays6[6 * rl + 4]; r[k].dz = rays6[6 * rl + 5];
    }
    unsigned long long best[TPL_RAYS];
#pragma unroll
    for (int k = 0; k < TPL_RAYS; ++k) best[k] = KEY_MISS;
    int64_t f0 = (int64_t)chunk * tris_per_chunk;
    int64_t f1 = f0 + tris_per_chunk;
    if (f1 > F_padded) f1 = F_padded;
    for (int64_t f = f0 + lane; f < f1; f += 64) {
        float4 a = tri[3 * f + 0], bq = tri[3 * f + 1], c = tri[3 * f + 2];
#pragma unroll
        for (int k = 0; k < TPL_RAYS; ++k) {
            MT m = mt_eval(r[k], a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x);
            if (mt_pass(m) && m.det != 0.0f) {
                unsigned long long key = mt_key(m, (unsigned)f);
                best[k] = key < best[k] ? key : best[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < TPL_RAYS; ++k) {
        unsigned long long v = wave_min_u64(best[k]);  // the wavefront-wide min-t reduction
        if (lane == 0 && ray0 + k < N && v != KEY_MISS) atomicMin(&keys[ray0 + k], v);
    }
}

// ------------------------------------------------------------------ finalize
__global__ void ray_finalize_kernel(const float4 *__restrict__ tri, const float *__restrict__ rays6,
                                    int64_t N, const unsigned long long *__restrict__ keys,
                                    float *__restrict__ t_hit, uint32_t *__restrict__ prim_id,
                                    float *__restrict__ uv) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    unsigned long long key = keys[i];
    unsigned id = (unsigned)(key & 0xFFFFFFFFull);
    if (key == KEY_MISS) {
        t_hit[i] = __uint_as_float(0x7F800000u);
        prim_id[i] = 0xFFFFFFFFu;
        if (uv) { uv[2 * i] = 0.0f; uv[2 * i + 1] = 0.0f; }
        return;
    }
    t_hit[i] = __uint_as_float((unsigned)(key >> 32));
    prim_id[i] = id;
    if (uv) {
        Ray r;
        r.ox = rays6[6 * i + 0]; r.oy = rays6[6 * i + 1]; r.oz = rays6[6 * i + 2];
        r.dx = rays6[6 * i + 3]; r.dy = rays6[6 * i + 4]; r.dz = rays6[6 * i + 5];
        float4 a = tri[3 * (size_t)id + 0], bq = tri[3 * (size_t)id + 1], c = tri[3 * (size_t)id + 2];
        MT m = mt_eval(r, a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x);
        unsigned sg = __float_as_uint(m.det) & 0x80000000u;
        float U = __uint_as_float(__float_as_uint(m.un) ^ sg);
        float V = __uint_as_float(__float_as_uint(m.vn) ^ sg);
        float ad = fabsf(m.det);
        uv[2 * i] = __fdiv_rn(U, ad);
        uv[2 * i + 1] = __fdiv_rn(V, ad);
    }
}

}  // namespace

extern "C" {

int pedp_mesh_create(pedp_ctx_t c, const float *verts, int64_t V, const uint32_t *tris, int64_t F,
                     pedp_mesh_t *out) {
    PEDP_REQUIRE(c && out, "pedp_mesh_create: null context/output");
    *out = nullptr;
    PEDP_REQUIRE(V >= 0 && F >= 0 && F < (int64_t)0x7FFFFF00, "pedp_mesh_create: sizes out of range");
    PEDP_REQUIRE((verts || V == 0) && (tris || F == 0), "pedp_mesh_create: null arrays");
    for (int64_t i = 0; i < 3 * F; ++i)
        PEDP_REQUIRE((int64_t)tris[i] < V, "pedp_mesh_create: triangle %lld references vertex %u >= V=%lld",
                     (long long)(i / 3), tris[i], (long long)V);
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    pedp_mesh_s *m = new (std::nothrow) pedp_mesh_s();
    if (!m) { pedp_set_error("pedp_mesh_create: out of host memory"); return PEDP_ERR_ALLOC; }
    m->ctx = c;
    m->V = V;
    m->F = F;
    m->F_padded = ((F + 63) / 64) * 64;
    if (m->F_padded == 0) m->F_padded = 64;
    float *d_verts = nullptr;
    uint32_t *d_tris = nullptr;
    hipError_t e = hipMalloc((void **)&m->tri, sizeof(float) * PEDP_TRI_STRIDE * (size_t)m->F_padded);
    if (e == hipSuccess) e = hipMalloc((void **)&d_verts, sizeof(float) * 3 * (size_t)(V ? V : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_tris, sizeof(uint32_t) * 3 * (size_t)(F ? F : 1));
    if (e == hipSuccess && V) e = hipMemcpyAsync(d_verts, verts, sizeof(float) * 3 * (size_t)V, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && F) e = hipMemcpyAsync(d_tris, tris, sizeof(uint32_t) * 3 * (size_t)F, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        int grid = (int)((m->F_padded + 255) / 256);
        hipLaunchKernelGGL(tri_setup_kernel, dim3(grid), dim3(256), 0, c->stream, d_verts, d_tris, F,
                           m->F_padded, m->tri);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_verts) (void)hipFree(d_verts);
    if (d_tris) (void)hipFree(d_tris);
    if (e != hipSuccess) {
        pedp_set_error("pedp_mesh_create: %s", hipGetErrorString(e));
        pedp_mesh_destroy(m);
        return PEDP_ERR_HIP;
    }
    *out = m;
    return PEDP_OK;
}

void pedp_mesh_destroy(pedp_mesh_t m) {
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->tri) (void)hipFree(m->tri);
    delete m;
}

int pedp_mesh_size(pedp_mesh_t m, int64_t *V, int64_t *F) {
    PEDP_REQUIRE(m, "pedp_mesh_size: null mesh");
    if (V) *V = m->V;
    if (F) *F = m->F;
    return PEDP_OK;
}

int pedp_raycast_configure(pedp_ctx_t c, int tri_chunks, int variant) {
    PEDP_REQUIRE(c, "pedp_raycast_configure: null context");
    PEDP_REQUIRE(tri_chunks >= 0 && tri_chunks % 8 == 0, "pedp_raycast_configure: tri_chunks must be a multiple of 8");
    PEDP_REQUIRE(variant >= 0 && variant <= 2, "pedp_raycast_configure: variant must be 0, 1 or 2");
    c->ray_tri_chunks = tri_chunks;
    c->ray_variant = variant;
    return PEDP_OK;
}

int pedp_raycast(pedp_ctx_t c, pedp_mesh_t mesh, const float *rays6, int64_t N, int mem,
                 float *t_hit, uint32_t *prim_id, float *uv) {
    PEDP_REQUIRE(c && mesh, "pedp_raycast: null context/mesh");
    PEDP_REQUIRE(mesh->ctx == c, "pedp_raycast: mesh belongs to another context");
    PEDP_REQUIRE(N >= 0 && N < (int64_t)1 << 40, "pedp_raycast: N out of range");
    PEDP_REQUIRE(mem == PEDP_HOST || mem == PEDP_DEVICE, "pedp_raycast: bad mem flag %d", mem);
    if (N == 0) return PEDP_OK;
    PEDP_REQUIRE(rays6 && t_hit && prim_id, "pedp_raycast: null arrays");
    PEDP_HIP_CHECK(hipSetDevice(c->device));

    const float *d_rays = rays6;
    float *d_t = t_hit, *d_uv = uv;
    uint32_t *d_id = prim_id;
    if (mem == PEDP_HOST) {
        int st = c->ray_in.reserve(sizeof(float) * 6 * (size_t)N);
        if (st) return st;
        st = c->ray_out.reserve(sizeof(float) * 4 * (size_t)N);
        if (st) return st;
        PEDP_HIP_CHECK(hipMemcpyAsync(c->ray_in.ptr, rays6, sizeof(float) * 6 * (size_t)N, hipMemcpyHostToDevice, c->stream));
        d_rays = (const float *)c->ray_in.ptr;
        d_t = (float *)c->ray_out.ptr;
        d_id = (uint32_t *)(d_t + N);
        d_uv = uv ? (float *)(d_id + N) : nullptr;
    }
    int st = c->ray_keys.reserve(sizeof(unsigned long long) * (size_t)N);
    if (st) return st;
    unsigned long long *keys = (unsigned long long *)c->ray_keys.ptr;
    PEDP_HIP_CHECK(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * (size_t)N, c->stream));

    int variant = c->ray_variant;
    if (variant == 0) variant = (N < 16384) ? 2 : 1;
    const float4 *tri = (const float4 *)mesh->tri;
    PEDP_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    if (variant == 1) {
        int64_t ray_blocks = (N + RPL_BLOCK - 1) / RPL_BLOCK;
        int groups_total = (int)(mesh->F_padded / RPL_UNROLL);
        int n_chunks = c->ray_tri_chunks;
        if (n_chunks == 0) {
            // enough workgroups for >= 4 rounds over the chip, triangle chunks not below 2k
            n_chunks = 8;
            while (ray_blocks * n_chunks < 4 * 8 * (int64_t)c->num_cus && groups_total / (n_chunks * 2) >= 512) n_chunks *= 2;
        }
        int gpc = (groups_total + n_chunks - 1) / n_chunks;
        int64_t grid = ray_blocks * n_chunks;
        PEDP_REQUIRE(grid < (int64_t)0x7FFFFFFF, "pedp_raycast: grid too large");
        hipLaunchKernelGGL(ray_sweep_rpl_kernel, dim3((unsigned)grid), dim3(RPL_BLOCK), 0, c->stream, tri,
                           groups_total, gpc, n_chunks, d_rays, N, keys);
    } else {
        int64_t rays_per_block = (TPL_BLOCK / 64) * TPL_RAYS;
        int64_t ray_blocks = (N + rays_per_block - 1) / rays_per_block;
        int n_chunks = c->ray_tri_chunks;
        if (n_chunks == 0) {
            n_chunks = 8;
            while (ray_blocks * n_chunks < 4 * 8 * (int64_t)c->num_cus && mesh->F_padded / (n_chunks * 2) >= 1024) n_chunks *= 2;
        }
        int tpc = (int)((mesh->F_padded + n_chunks - 1) / n_chunks);
        tpc = ((tpc + 63) / 64) * 64;
        int64_t grid = ray_blocks * n_chunks;
        PEDP_REQUIRE(grid < (int64_t)0x7FFFFFFF, "pedp_raycast: grid too large");
        hipLaunchKernelGGL(ray_sweep_tpl_kernel, dim3((unsigned)grid), dim3(TPL_BLOCK), 0, c->stream, tri,
                           mesh->F_padded, tpc, n_chunks, d_rays, N, keys);
    }
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    c->ray_timed = true;
    {
        int64_t grid = (N + 255) / 256;
        hipLaunchKernelGGL(ray_finalize_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, tri, d_rays, N,
                           keys, d_t, d_id, d_uv);
        PEDP_HIP_CHECK(hipGetLastError());
    }
    if (mem == PEDP_HOST) {
        PEDP_HIP_CHECK(hipMemcpyAsync(t_hit, d_t, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipMemcpyAsync(prim_id, d_id, sizeof(uint32_t) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
        if (uv) PEDP_HIP_CHECK(hipMemcpyAsync(uv, d_uv, sizeof(float) * 2 * (size_t)N, hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PEDP_OK;
}

int pedp_raycast_last_sweep_ms(pedp_ctx_t c, float *ms) {
    PEDP_REQUIRE(c && ms, "pedp_raycast_last_sweep_ms: null argument");
    PEDP_REQUIRE(c->ray_timed, "pedp_raycast_last_sweep_ms: no sweep has run on this context");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    PEDP_HIP_CHECK(hipEventSynchronize(c->ev1));
    PEDP_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PEDP_OK;
}

}  // extern "C"
