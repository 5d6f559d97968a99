// Point-cloud operations of preprocess_source for gfx950: the Open3D calls the
// reference chains in src/pose_estimation.py:186-268, restated from the published open3d==0.18.0
// algorithms (oracle/cloudops.c is the CPU statement the tests compare with).
//
//   pedp_voxel_down_sample      pcd.voxel_down_sample(voxel_size)                   :204-205
//   pedp_cluster_dbscan         pcd.cluster_dbscan(eps, min_points)                 :284
//   pedp_knn_mean_distance      per-point part of pcd.remove_statistical_outlier    :308-312
//   pedp_segment_plane          pcd.segment_plane(threshold, 3, num_iterations)     :323-329
//
// Everything with an order-dependent result is made order-free or given the oracle's order:
//   * voxel averages: points are sorted by voxel key with a STABLE radix sort (rocPRIM), so a
//     voxel's members are summed in point order like Open3D's sequential accumulation;
//   * DBSCAN: clusters are the connected components of the core points (lock-free union-find,
//     smaller index wins, so the representative is the component's smallest core index whatever the
//     execution order); cluster ids rank the representatives -- Open3D's discovery order -- and a
//     border point takes the smallest id among its core neighbours -- the first cluster that
//     reaches it in Open3D's breadth-first sweep;
//   * k nearest neighbours: each thread keeps its k smallest squared distances sorted, so the mean
//     is summed in ascending order;
//   * plane RANSAC: one workgroup per iteration counts the inliers of its sampled plane; choosing
//     the best iteration, the final inliers and the refit are O(N) and done by the host in the
//     oracle's order.
// Neighbourhood queries (DBSCAN) walk a uniform grid with cell >= eps built by a sort on the cell id.
#include "pedp_internal.h"
#include <cmath>
#include <cstring>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

// rocPRIM sorts up to a million items by merging (about twenty launches whatever the key's width); with the limit at
// zero it always takes the radix passes, whose number follows the bits asked for
using RadixPasses = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

inline size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Carver {
    char *base;
    size_t off = 0;
    template <class T> T *take(size_t n) {
        T *p = (T *)(base + off);
        off = a256(off + sizeof(T) * n);
        return p;
    }
};

__device__ __forceinline__ double dist2(const double *a, const double *b) {
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;  // -ffp-contract=off: no FMA, the oracle's rounding
}

// ------------------------------------------------------------------ voxel grid
// key = (cx, cy, cz) packed most significant first into just the bits the box needs (sy, sz: the widths of cy and cz):
// the radix sort's passes follow the key's width
__global__ void voxel_key_kernel(const double *__restrict__ pts, int64_t N, double lox, double loy, double loz, double voxel, int sy,
                                 int sz, unsigned long long *__restrict__ key, int *__restrict__ val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned long long cx = (unsigned long long)(long long)floor((pts[3 * i] - lox) / voxel);
    const unsigned long long cy = (unsigned long long)(long long)floor((pts[3 * i + 1] - loy) / voxel);
    const unsigned long long cz = (unsigned long long)(long long)floor((pts[3 * i + 2] - loz) / voxel);
    key[i] = ((cx & 0x1FFFFFull) << (sy + sz)) | ((cy & ((1ull << sy) - 1ull)) << sz) | (cz & ((1ull << sz) - 1ull));
    val[i] = (int)i;
}

__global__ void voxel_average_kernel(const double *__restrict__ pts, const double *__restrict__ nrm,
                                     const int *__restrict__ sorted_idx, const unsigned *__restrict__ counts,
                                     const unsigned *__restrict__ offsets, const unsigned *__restrict__ n_runs,
                                     double *__restrict__ out_pts, double *__restrict__ out_nrm) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *n_runs) return;
    const unsigned a = offsets[v], n = counts[v];
    double s0 = 0, s1 = 0, s2 = 0, n0 = 0, n1 = 0, n2 = 0;
    for (unsigned j = 0; j < n; ++j) {  // members in point order (stable sort)
        const int64_t i = sorted_idx[a + j];
        s0 += pts[3 * i]; s1 += pts[3 * i + 1]; s2 += pts[3 * i + 2];
        if (nrm) { n0 += nrm[3 * i]; n1 += nrm[3 * i + 1]; n2 += nrm[3 * i + 2]; }
    }
    const double c = (double)n;
    out_pts[3 * (size_t)v] = s0 / c; out_pts[3 * (size_t)v + 1] = s1 / c; out_pts[3 * (size_t)v + 2] = s2 / c;
    if (nrm) { out_nrm[3 * (size_t)v] = n0 / c; out_nrm[3 * (size_t)v + 1] = n1 / c; out_nrm[3 * (size_t)v + 2] = n2 / c; }
}

// ------------------------------------------------------------------ uniform grid
struct Grid {
    double lo[3], cell;
    int dim[3];
};

__device__ __forceinline__ int grid_axis(double p, double lo, double cell, int dim) {
    int c = (int)floor((p - lo) / cell);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ void grid_cell_kernel(const double *__restrict__ pts, int64_t N, Grid g, unsigned *__restrict__ cell,
                                 int *__restrict__ val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int cx = grid_axis(pts[3 * i], g.lo[0], g.cell, g.dim[0]), cy = grid_axis(pts[3 * i + 1], g.lo[1], g.cell, g.dim[1]),
              cz = grid_axis(pts[3 * i + 2], g.lo[2], g.cell, g.dim[2]);
    cell[i] = (unsigned)(cx + g.dim[0] * (cy + g.dim[1] * cz));
    val[i] = (int)i;
}

// sorted by cell: where each cell's run starts and ends (empty cells keep the 0 / 0 they were cleared to), and the
// points gathered in that order
__global__ void grid_ranges_kernel(const unsigned *__restrict__ cell_sorted, const int *__restrict__ idx_sorted,
                                   const double *__restrict__ pts, int64_t N, int *__restrict__ cell_start,
                                   int *__restrict__ cell_end, double *__restrict__ sp) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const unsigned c = cell_sorted[j];
    if (j == 0 || cell_sorted[j - 1] != c) cell_start[c] = (int)j;
    if (j == N - 1 || cell_sorted[j + 1] != c) cell_end[c] = (int)j + 1;
    const int64_t i = idx_sorted[j];
    sp[3 * j] = pts[3 * i]; sp[3 * j + 1] = pts[3 * i + 1]; sp[3 * j + 2] = pts[3 * i + 2];
}

// The index by counting: a point's cell and its slot among the cell's points (an atomic count: the slots' order is
// whatever the hardware made it -- every reader of the index is indifferent to the order inside a cell: counts,
// minima, sets, and lists with explicit (distance, index) order), an exclusive scan of the counts, the points placed.
// Six launches; a sort of nine thousand pairs is a dozen.  begin[] has n_cells + 1 entries and is monotone, so cell
// c's run is begin[c] .. begin[c + 1] and the cells c .. c' of one x row are the one run begin[c] .. begin[c' + 1].
__global__ void grid_count_kernel(const double *__restrict__ pts, int64_t N, Grid g, unsigned *__restrict__ cell, int *__restrict__ slot,
                                  unsigned *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int cx = grid_axis(pts[3 * i], g.lo[0], g.cell, g.dim[0]), cy = grid_axis(pts[3 * i + 1], g.lo[1], g.cell, g.dim[1]),
              cz = grid_axis(pts[3 * i + 2], g.lo[2], g.cell, g.dim[2]);
    const unsigned c = (unsigned)(cx + g.dim[0] * (cy + g.dim[1] * cz));
    cell[i] = c;
    slot[i] = (int)atomicAdd(&count[c], 1u);
}
__global__ void grid_place_kernel(const double *__restrict__ pts, int64_t N, const unsigned *__restrict__ cell,
                                  const int *__restrict__ slot, const int *__restrict__ begin, double *__restrict__ sp,
                                  int *__restrict__ idx_sorted, unsigned *__restrict__ cell_s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned c = cell[i];
    const int64_t j = (int64_t)begin[c] + slot[i];
    sp[3 * j] = pts[3 * i]; sp[3 * j + 1] = pts[3 * i + 1]; sp[3 * j + 2] = pts[3 * i + 2];
    idx_sorted[j] = (int)i;
    cell_s[j] = c;
}

// The cells x0 .. x1 of one x row have consecutive ids, so their points are ONE run of the cell-sorted order
// (rs = re: the row is empty).
__device__ __forceinline__ void row_run(const int *__restrict__ cell_start, const int *__restrict__ cell_end, int base, int x0, int x1,
                                        int &rs, int &re) {
    rs = cell_start[base + x0];
    re = cell_end[base + x1];
}

// calls f(q) for every sorted position q whose point lies in one of the 27 cells around p
template <class F>
__device__ __forceinline__ void for_neighbours(const Grid &g, const double *p, const int *__restrict__ cell_start,
                                               const int *__restrict__ cell_end, F f) {
    const int cx = grid_axis(p[0], g.lo[0], g.cell, g.dim[0]), cy = grid_axis(p[1], g.lo[1], g.cell, g.dim[1]),
              cz = grid_axis(p[2], g.lo[2], g.cell, g.dim[2]);
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dim[2] - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dim[1] - 1); ++y)
            for (int x = max(cx - 1, 0); x <= min(cx + 1, g.dim[0] - 1); ++x) {
                const int c = x + g.dim[0] * (y + g.dim[1] * z);
                for (int q = cell_start[c]; q < cell_end[c]; ++q) f(q);
            }
}

// ------------------------------------------------------------------ DBSCAN
// The grid cell is eps / sqrt(3) less a hair: two points of one cell are closer than eps, so the core points of a
// cell form a clique -- they are one component without looking -- and a cell pair needs ONE edge between two of
// their core points to be united.  Points within eps lie at most DB_R = 2 cells apart per axis: the neighbourhood
// is 25 x rows of 5 cells, each row one run of the cell-sorted points (row_run), so lanes 0..24 of a
// wave fetch the 25 runs at once and the wave takes the non-empty ones 64 points at a time.  (With cells of eps
// every core point united itself with every core neighbour: 350 k finds and compare-and-swaps for 9 k points,
// 340 us; cells as cliques leave about a tenth of that.)
// A cloud whose extent over eps is beyond the grid's cell budget gets cells of eps or more (make_grid doubles them until
// they fit): no cliques there, 9 rows of 3 cells, and every core point unites itself with every core neighbour
// (CLIQUE = false below).
struct DbCell { int x, y, z; };
__device__ __forceinline__ DbCell db_cell(const Grid &g, const double *p) {
    return {grid_axis(p[0], g.lo[0], g.cell, g.dim[0]), grid_axis(p[1], g.lo[1], g.cell, g.dim[1]),
            grid_axis(p[2], g.lo[2], g.cell, g.dim[2])};
}
// the run of row `r` (0 .. (2R+1)^2 - 1) around cell c: sorted positions rs .. re
template <int DB_R>
__device__ __forceinline__ void db_row(const Grid &g, const DbCell &c, int r, const int *__restrict__ cell_start,
                                       const int *__restrict__ cell_end, int &rs, int &re) {
    constexpr int DB_SIDE = 2 * DB_R + 1;
    const int y = c.y + r % DB_SIDE - DB_R, z = c.z + r / DB_SIDE - DB_R;
    rs = re = 0;
    if (r < DB_SIDE * DB_SIDE && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) {
        row_run(cell_start, cell_end, g.dim[0] * (y + g.dim[1] * z), max(c.x - DB_R, 0), min(c.x + DB_R, g.dim[0] - 1), rs, re);
    }
}
template <int DB_R, class F>
__device__ __forceinline__ void for_rows_wave(const Grid &g, const double *p, const int *__restrict__ cell_start,
                                              const int *__restrict__ cell_end, int lane, F f) {
    int rs, re;
    db_row<DB_R>(g, db_cell(g, p), lane, cell_start, cell_end, rs, re);
    unsigned long long busy = __builtin_amdgcn_ballot_w64(re > rs);
    while (busy) {  // wave-uniform
        const int b = __builtin_ctzll(busy);
        busy &= busy - 1;
        const int e = __builtin_amdgcn_readlane(re, b);
        for (int q0 = __builtin_amdgcn_readlane(rs, b); q0 < e; q0 += 64) f(q0 + lane, q0 + lane < e);  // wave-uniform trip count
    }
}

constexpr int DB_WPB = 4;  // point waves per workgroup
template <bool CLIQUE>
__global__ __launch_bounds__(DB_WPB * 64) void dbscan_core_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                                 const int *__restrict__ cell_start,
                                                                 const int *__restrict__ cell_end, double e2, int min_points,
                                                                 const int *__restrict__ idx_sorted,
                                                                 int *__restrict__ core /* by original index */,
                                                                 int *__restrict__ parent) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * DB_WPB + (threadIdx.x >> 6);
    if (j >= N) return;  // wave-uniform
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    int cnt = 0;
    for_rows_wave<CLIQUE ? 2 : 1>(g, p, cell_start, cell_end, lane, [&](int q, bool valid) {
        const bool in = valid && dist2(p, sp + 3 * (size_t)q) < e2;
        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
    });
    if (lane == 0) {
        const int i = idx_sorted[j];
        core[i] = cnt >= min_points;
        parent[i] = i;
    }
}

// a thread per cell: rep[c] = the smallest index among the cell's core points (none: INT_MAX), and every core point
// of the cell hangs under it -- the clique, united without an atomic
__global__ void dbscan_cell_kernel(int64_t n_cells, const int *__restrict__ cell_start, const int *__restrict__ cell_end,
                                   const int *__restrict__ idx_sorted,
                                   const int *__restrict__ core, int *__restrict__ rep, int *__restrict__ parent) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const int a = cell_start[c], e = cell_end[c];
    int r = 0x7FFFFFFF;
    for (int q = a; q < e; ++q) {
        const int i = idx_sorted[q];
        if (core[i]) r = min(r, i);
    }
    rep[c] = r;
    if (r == 0x7FFFFFFF) return;
    for (int q = a; q < e; ++q) {
        const int i = idx_sorted[q];
        if (core[i]) parent[i] = r;
    }
}

// Find with path halving.  parent[x] <= x always points at a member of x's component, and only roots are
// ever hooked (atomicCAS on parent[root] == root in the union loop), so overwriting a NON-root's parent with
// its grandparent -- a plain store, whoever wins a race stores an ancestor -- keeps every invariant and the
// component's smallest index as its root; without it the index-ordered hooking grows chains hundreds of
// links long.
__device__ __forceinline__ int uf_find(int *parent, int a) {
    int r = a;
    while (true) {
        const int p = ((volatile int *)parent)[r];
        if (p == r) return r;
        const int gp = ((volatile int *)parent)[p];
        if (gp != p) ((volatile int *)parent)[r] = gp;
        r = p;
    }
}

// Uniting the cells.  With cliques a wave per CELL (the wave of the cell's first point; the others leave): the cell's
// core points wait in LDS, the wave walks the neighbourhood once, and a core neighbour q in a cell ABOVE this one
// (the pair is seen from the lower cell) within eps of any of them lists q's cell -- the first of every run of equal
// cells in a pass of 64.  The union-find is touched once per cell at the end (its loads and compare-and-swaps are
// device-coherent, a microsecond each): the listed cells' components are resolved to their roots, and every
// distinct root above the smallest of them (and of the cell's own) is hooked under that one, so the lanes'
// compare-and-swaps go to different words.  Without cliques a wave per core point i lists its core neighbours of
// smaller index (the pair is seen from the larger) and unites itself with them the same way.
constexpr int DB_LIST = 128;  // listed cells per wave before the union-find is visited
template <bool CLIQUE>
__global__ __launch_bounds__(DB_WPB * 64) void dbscan_union_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                                  const int *__restrict__ cell_start,
                                                                  const int *__restrict__ cell_end, const unsigned *__restrict__ cell_s,
                                                                  double e2, const int *__restrict__ idx_sorted,
                                                                  const int *__restrict__ core, const int *__restrict__ rep,
                                                                  int *__restrict__ parent) {
    __shared__ int list_all[DB_WPB][DB_LIST];
    __shared__ double mine_all[DB_WPB][64][3];
    const int lane = threadIdx.x & 63;
    int *list = list_all[threadIdx.x >> 6];
    double (*mine)[3] = mine_all[threadIdx.x >> 6];
    const int64_t j = (int64_t)blockIdx.x * DB_WPB + (threadIdx.x >> 6);
    if (j >= N) return;  // wave-uniform
    const unsigned ci = cell_s[j];
    int i = idx_sorted[j];
    if (CLIQUE) {
        if (cell_start[ci] != (int)j) return;  // wave-uniform: not the cell's first point
        i = rep[ci];
        if (i == 0x7FFFFFFF) return;  // no core point in the cell
    } else if (!core[i]) return;  // wave-uniform
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    const unsigned long long below = (1ull << lane) - 1ull;
    int listed = 0;
    auto unite = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the list has landed
        for (int l0 = 0; l0 < listed; l0 += 64) {
            const bool act = l0 + lane < listed;
            const int r = act ? uf_find(parent, CLIQUE ? rep[list[l0 + lane]] : list[l0 + lane]) : 0x7FFFFFFF, ri = uf_find(parent, i);
            int low = min(r, ri);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) low = min(low, __shfl_xor(low, off, 64));
#pragma unroll
            for (int round = 0; round < 2; ++round) {  // the listed roots, then i's own (lane 0)
                int a = round == 0 ? (act ? r : low) : (lane == 0 ? ri : low), b = low;
                while (a != b) {  // union: the larger root is hooked under the smaller one
                    if (a < b) { const int t = a; a = b; b = t; }
                    if (atomicCAS(&parent[a], a, b) == a) break;
                    a = uf_find(parent, a);
                    b = uf_find(parent, b);
                }
            }
        }
        listed = 0;
    };
    const int own_end = CLIQUE ? cell_end[ci] : (int)j + 1;
    for (int a0 = (int)j; a0 < own_end; a0 += 64) {  // the cell's points, 64 at a time (CLIQUE; else the one point)
        int n_mine = 1;
        if (CLIQUE) {
            const bool is_core = a0 + lane < own_end && core[idx_sorted[a0 + lane]];
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(is_core);
            n_mine = __builtin_popcountll(cm);
            if (!n_mine) continue;  // wave-uniform
            if (is_core) {
                const int slot = __builtin_popcountll(cm & below);
                mine[slot][0] = sp[3 * (size_t)(a0 + lane)]; mine[slot][1] = sp[3 * (size_t)(a0 + lane) + 1];
                mine[slot][2] = sp[3 * (size_t)(a0 + lane) + 2];
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        for_rows_wave<CLIQUE ? 2 : 1>(g, p, cell_start, cell_end, lane, [&](int q, bool valid) {
            unsigned cq = 0;
            int other = 0;
            bool act = false;
            if (valid) {
                cq = cell_s[q];
                other = idx_sorted[q];
                act = (CLIQUE ? cq > ci : other < i) && core[other];
            }
            if (CLIQUE) {
                if (act) {
                    const double pq[3] = {sp[3 * (size_t)q], sp[3 * (size_t)q + 1], sp[3 * (size_t)q + 2]};
                    bool near = false;
                    for (int a = 0; a < n_mine; ++a) near |= dist2(mine[a], pq) < e2;
                    act = near;
                }
                const unsigned key = act ? cq : 0xFFFFFFFFu, before = __shfl_up(key, 1, 64);
                act = act && (lane == 0 || before != key);  // one lane per run of equal cells (position order = cell order)
            } else act = act && dist2(p, sp + 3 * (size_t)q) < e2;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(act);
            if (!mask) return;  // wave-uniform
            if (listed + 64 > DB_LIST) unite();
            if (act) list[listed + __builtin_popcountll(mask & below)] = CLIQUE ? (int)cq : other;
            listed += __builtin_popcountll(mask);
        });
    }
    unite();
}

// roots of core components in index order -> flags for the rank scan
__global__ void dbscan_root_kernel(int64_t N, const int *__restrict__ core, int *__restrict__ parent, int *__restrict__ root,
                                   unsigned *__restrict__ is_rep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int r = -1;
    if (core[i]) r = uf_find(parent, (int)i);
    root[i] = r;
    is_rep[i] = (r == (int)i) ? 1u : 0u;
}

// labels: a component's rank among the roots in index order; a point that is not core takes the smallest rank among
// the core points within eps (none: -1, noise).  A wave per point; the core points' waves leave at once.
template <bool CLIQUE>
__global__ __launch_bounds__(DB_WPB * 64) void dbscan_label_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                                  const int *__restrict__ cell_start,
                                                                  const int *__restrict__ cell_end, double e2,
                                                                  const int *__restrict__ idx_sorted, const int *__restrict__ root,
                                                                  const unsigned *__restrict__ rank /* exclusive scan of is_rep */,
                                                                  int32_t *__restrict__ labels) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * DB_WPB + (threadIdx.x >> 6);
    if (j >= N) return;  // wave-uniform
    const int i = idx_sorted[j];
    if (root[i] >= 0) {  // wave-uniform
        if (lane == 0) labels[i] = (int32_t)rank[root[i]];
        return;
    }
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    int best = 0x7FFFFFFF;
    for_rows_wave<CLIQUE ? 2 : 1>(g, p, cell_start, cell_end, lane, [&](int q, bool valid) {
        if (!valid) return;
        const int rq = root[idx_sorted[q]];
        if (rq >= 0 && dist2(p, sp + 3 * (size_t)q) < e2) best = min(best, (int)rank[rq]);
    });
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
    if (lane == 0) labels[i] = best == 0x7FFFFFFF ? -1 : best;
}

// ------------------------------------------------------------------ k nearest neighbours (mean distance)
// One thread per query walks the grid shell by shell around its own cell: after the shell at
// Chebyshev distance rho every unvisited point is farther than rho * cell, so the search ends as
// soon as the k-th best squared distance is within (rho * cell)^2.  Near cells come first, so the
// k smallest squared distances -- kept sorted ascending in LDS, k x 64 doubles per workgroup --
// settle after little more than k insertions.  Candidates are fetched four at a time.
constexpr int KNN_THREADS = 64;
__global__ __launch_bounds__(KNN_THREADS) void knn_mean_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                               const int *__restrict__ cell_start,
                                                               const int *__restrict__ cell_end,
                                                               const int *__restrict__ idx_sorted, int k,
                                                               double *__restrict__ avg /* by original index */,
                                                               const int *__restrict__ gate /* run only if *gate != 0 */) {
    extern __shared__ double top[];  // element r of thread t at top[r * KNN_THREADS + t]
    const int t = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * KNN_THREADS + t;
    if (j >= N || *gate == 0) return;
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    const int m = (int)(N < k ? N : k);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    int have = 0;
    double worst = inf;  // the m-th best once m candidates are in
    auto offer = [&](double d) {
        if (!(d < worst)) return;
        int r = have < m ? have : m - 1;  // slot that opens up
        while (r > 0 && top[(r - 1) * KNN_THREADS + t] > d) {
            top[r * KNN_THREADS + t] = top[(r - 1) * KNN_THREADS + t];
            --r;
        }
        top[r * KNN_THREADS + t] = d;
        if (have < m) ++have;
        if (have == m) worst = top[(m - 1) * KNN_THREADS + t];
    };
    const int cx = grid_axis(p[0], g.lo[0], g.cell, g.dim[0]), cy = grid_axis(p[1], g.lo[1], g.cell, g.dim[1]),
              cz = grid_axis(p[2], g.lo[2], g.cell, g.dim[2]);
    const int far = max(max(max(cx, g.dim[0] - 1 - cx), max(cy, g.dim[1] - 1 - cy)), max(cz, g.dim[2] - 1 - cz));
    for (int rho = 0; rho <= far; ++rho) {
        for (int z = max(cz - rho, 0); z <= min(cz + rho, g.dim[2] - 1); ++z)
            for (int y = max(cy - rho, 0); y <= min(cy + rho, g.dim[1] - 1); ++y) {
                const bool face = (abs(z - cz) == rho) || (abs(y - cy) == rho);
                for (int x = max(cx - rho, 0); x <= min(cx + rho, g.dim[0] - 1); ++x) {
                    if (!face && abs(x - cx) != rho) continue;  // interior of the cube: visited by an earlier shell
                    const int c = x + g.dim[0] * (y + g.dim[1] * z);
                    const int e = cell_end[c];
                    for (int q = cell_start[c]; q < e; q += 4) {
                        double d[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) d[u] = (q + u < e) ? dist2(p, sp + 3 * (size_t)(q + u)) : inf;
#pragma unroll
                        for (int u = 0; u < 4; ++u) offer(d[u]);
                    }
                }
            }
        const double reach = (double)rho * g.cell * (1.0 - 1e-9);
        if (have == m && worst <= reach * reach) break;
    }
    double s = 0.0;
    for (int r = 0; r < m; ++r) s += sqrt(top[r * KNN_THREADS + t]);
    avg[idx_sorted[j]] = m > 0 ? s / (double)m : -1.0;
}

// The k smallest of the n_c squared distances in cand[] (a wave's LDS share), summed as square roots in
// ascending order -- the order the oracle's sorted top-k list is summed in.  `hi` is a value with
// chi = #{cand <= hi} >= m.  Three steps, none of them a serial extraction (75 wave-wide argmins cost
// more than everything else the query does):
//   1. halve [lo, hi] on the value until about m + 24 candidates lie below hi (counts by ballot),
//   2. compact those survivors to the front of cand[] (in place: writes trail the reads),
//   3. rank each survivor by counting the survivors below it ((d, position) order, so ties get distinct
//      ranks), write sqrt(d) of ranks < m to srt[rank], and sum srt[0..m) in order.
// Same multiset, same summation order, same bits as the extraction it replaces.
constexpr int KNN_KMAX = 304;  // k <= 300 (pedp_knn_mean_distance)
__device__ __forceinline__ double knn_select_sum(double *cand, int n_c, int m, double hi, int chi, double *srt, int lane) {
    const unsigned long long below = (1ull << lane) - 1ull;
    double lo = -1.0;  // #{cand <= lo} < m
    for (int it = 0; it < 64 && chi > m + 24; ++it) {
        const double mid = 0.5 * ((lo > 0.0 ? lo : 0.0) + hi);
        if (!(mid > lo && mid < hi)) break;
        int cnt = 0;
        for (int q0 = 0; q0 < n_c; q0 += 64) {
            const int q = q0 + lane;
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(q < n_c && cand[q] <= mid));
        }
        if (cnt >= m) { hi = mid; chi = cnt; } else lo = mid;
    }
    int S = 0;
    for (int q0 = 0; q0 < n_c; q0 += 64) {
        const int q = q0 + lane;
        const double d = q < n_c ? cand[q] : 0.0;
        const bool keep = q < n_c && d <= hi;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
        if (keep) cand[S + __builtin_popcountll(mask & below)] = d;  // S + ... <= q: behind every unread element
        S += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    for (int i = lane; i < S; i += 64) {
        const double d = cand[i];
        int r0 = 0, r1 = 0;
        int j = 0;
        for (; j + 1 < S; j += 2) {
            const double o0 = cand[j], o1 = cand[j + 1];
            r0 += (o0 < d || (o0 == d && j < i)) ? 1 : 0;
            r1 += (o1 < d || (o1 == d && j + 1 < i)) ? 1 : 0;
        }
        if (j < S) { const double o0 = cand[j]; r0 += (o0 < d || (o0 == d && j < i)) ? 1 : 0; }
        const int rank = r0 + r1;
        if (rank < m) srt[rank] = sqrt(d);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0.0;
    for (int r = 0; r < m; ++r) s += srt[r];  // every lane forms the same sum
    return s;
}

// One WAVE per query on the grid (clouds above KNN_SMALL_MAX points).  The candidates are the points of the cube of
// cells within rho of the query's cell: (2 rho + 1)^2 x rows, each one run of the cell-sorted points
// (row_run), so the lanes fetch the runs' bounds at once, and the candidates are then numbered through
// the runs and fetched 256 at a time -- four independent loads in flight per lane.  (Visiting the cells one after the
// other cost a memory latency per cell, most of them empty in a scanned sheet.)  The squared distances are parked
// in LDS; the cube is large enough when at least k of them lie within its reach rho * cell (every point outside
// the cube is farther: the k-th best is then settled -- the rule of the thread-per-query kernel's shells), else
// the next larger cube is gathered.  knn_select_sum does the rest.  A query whose cube holds more candidates than
// the LDS share raises `overflow`; the host then runs the thread-per-query kernel for the cloud.
constexpr int KNN_WCAP = 2048;  // candidates per query wave (16 KB)
constexpr int KNN_WPB = 4;      // query waves per workgroup
__global__ __launch_bounds__(KNN_WPB * 64) void knn_mean_wave_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                                    const int *__restrict__ cell_start,
                                                                    const int *__restrict__ cell_end,
                                                                    const int *__restrict__ idx_sorted, int k,
                                                                    double *__restrict__ avg, int *__restrict__ overflow) {
    __shared__ double cand_all[KNN_WPB][KNN_WCAP];
    __shared__ double srt_all[KNN_WPB][KNN_KMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * KNN_WPB + wave;
    if (j >= N) return;  // wave-uniform
    double *cand = cand_all[wave];
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    const int m = (int)(N < k ? N : k);
    const int cx = grid_axis(p[0], g.lo[0], g.cell, g.dim[0]), cy = grid_axis(p[1], g.lo[1], g.cell, g.dim[1]),
              cz = grid_axis(p[2], g.lo[2], g.cell, g.dim[2]);
    const int far = max(max(max(cx, g.dim[0] - 1 - cx), max(cy, g.dim[1] - 1 - cy)), max(cz, g.dim[2] - 1 - cz));
    int n_c = 0, within = 0;
    double reach2 = 0.0;
    for (int rho = 1;; ++rho) {
        n_c = 0;
        const int side = 2 * rho + 1, rows = side * side;
        const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.dim[0] - 1);
        for (int r0 = 0; r0 < rows; r0 += 64) {
            const int r = r0 + lane, y = cy + r % side - rho, z = cz + r / side - rho;
            int rs = 0, re = 0;
            if (r < rows && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2])
                row_run(cell_start, cell_end, g.dim[0] * (y + g.dim[1] * z), x0, x1, rs, re);
            const int cnt = re - rs;
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            const int first = incl - cnt, tot = __builtin_amdgcn_readlane(incl, 63);
            const unsigned long long busy = __builtin_amdgcn_ballot_w64(cnt > 0);
            if (n_c + tot <= KNN_WCAP) {  // wave-uniform; beyond it the query overflows below
                for (int c0 = 0; c0 < tot; c0 += 256) {
                    int src[4] = {0, 0, 0, 0};
                    for (unsigned long long bb = busy; bb; bb &= bb - 1) {  // the run each of the four candidates lies in
                        const int b = __builtin_ctzll(bb);
                        const int pb = __builtin_amdgcn_readlane(first, b), sb = __builtin_amdgcn_readlane(rs, b);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int idx = c0 + 64 * u + lane;
                            if (idx >= pb) src[u] = sb + (idx - pb);
                        }
                    }
                    double d[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) d[u] = dist2(p, sp + 3 * (size_t)(c0 + 64 * u + lane < tot ? src[u] : 0));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c0 + 64 * u + lane < tot) cand[n_c + c0 + 64 * u + lane] = d[u];
                }
            }
            n_c += tot;
        }
        if (n_c > KNN_WCAP) {  // wave-uniform
            if (lane == 0) atomicOr(overflow, 1);
            return;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS writes have landed
        const double reach = (double)rho * g.cell * (1.0 - 1e-9);
        reach2 = reach * reach;
        within = 0;
        for (int q0 = 0; q0 < n_c; q0 += 64) {
            const int q = q0 + lane;
            within += __builtin_popcountll(__builtin_amdgcn_ballot_w64(q < n_c && cand[q] <= reach2));
        }
        if (within >= m || rho >= far) break;  // rho >= far: the cube is the whole grid
    }
    double s;
    if (within >= m) s = knn_select_sum(cand, n_c, m, reach2, within, srt_all[wave], lane);
    else {  // the whole cloud without k points in reach: all n_c = N >= m candidates compete
        double mx = 0.0;
        for (int q = lane; q < n_c; q += 64) mx = fmax(mx, cand[q]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        s = knn_select_sum(cand, n_c, m, mx, n_c, srt_all[wave], lane);
    }
    if (lane == 0) avg[idx_sorted[j]] = m > 0 ? s / (double)m : -1.0;
}

// Small clouds (the largest cluster that reaches the outlier filter is a few thousand points): one
// WAVE per query.  The wave writes the squared distances to all N points into LDS and hands them to
// knn_select_sum with the largest of them as the first bound.  Same multiset and same summation
// order as the grid kernel and the oracle.
constexpr int KNN_SMALL_MAX = 4096;  // 32 KB of distances per query wave; beyond ~4k points the grid walk is faster
__global__ __launch_bounds__(64) void knn_mean_small_kernel(const double *__restrict__ pts, int N, int k,
                                                            double *__restrict__ avg) {
    extern __shared__ double dist[];  // N squared distances of this query
    __shared__ double srt[KNN_KMAX];
    const int lane = threadIdx.x, i = blockIdx.x;
    const double p[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    double mx = 0.0;
    for (int q = lane; q < N; q += 64) {
        const double d = dist2(p, pts + 3 * (size_t)q);
        dist[q] = d;
        mx = fmax(mx, d);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    const int m = N < k ? N : k;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const double s = knn_select_sum(dist, N, m, mx, N, srt, lane);
    if (lane == 0) avg[i] = m > 0 ? s / (double)m : -1.0;
}

// ------------------------------------------------------------------ normal estimation
// PointCloud::EstimateNormals with a hybrid search (radius, max_nn): the max_nn nearest points with
// d^2 < radius^2 in ascending (d^2, index) -- kept sorted in LDS per thread -- give the nine
// cumulants and the covariance; the normal is the eigenvector of the smallest eigenvalue by the
// non-iterative solver Open3D uses (FastEigen3x3, after Eberly's robust 3x3 symmetric solver).
// oracle/cloudops.c states the same arithmetic; acos/cos come from different libms.
__device__ inline void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ void eigenvector0(const double A[9], double e, double *out) {
    const double r0[3] = {A[0] - e, A[1], A[2]}, r1[3] = {A[1], A[4] - e, A[5]}, r2[3] = {A[2], A[5], A[8] - e};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01); cross3(r0, r2, c02); cross3(r1, r2, c12);
    const double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2_ = dot3(c12, c12);
    double b[3] = {c01[0], c01[1], c01[2]};
    double dm = d0;
    if (d1 > dm) { dm = d1; b[0] = c02[0]; b[1] = c02[1]; b[2] = c02[2]; }
    if (d2_ > dm) { dm = d2_; b[0] = c12[0]; b[1] = c12[1]; b[2] = c12[2]; }
    const double s = sqrt(dm);
    for (int k = 0; k < 3; ++k) out[k] = b[k] / s;
}
__device__ void eigenvector1(const double A[9], const double *ev0, double e, double *out) {
    double U[3], V[3];
    if (fabs(ev0[0]) > fabs(ev0[1])) {
        const double inv = 1.0 / sqrt(ev0[0] * ev0[0] + ev0[2] * ev0[2]);
        U[0] = -ev0[2] * inv; U[1] = 0.0; U[2] = ev0[0] * inv;
    } else {
        const double inv = 1.0 / sqrt(ev0[1] * ev0[1] + ev0[2] * ev0[2]);
        U[0] = 0.0; U[1] = ev0[2] * inv; U[2] = -ev0[1] * inv;
    }
    cross3(ev0, U, V);
    const double AU[3] = {(A[0] * U[0] + A[1] * U[1]) + A[2] * U[2], (A[1] * U[0] + A[4] * U[1]) + A[5] * U[2],
                          (A[2] * U[0] + A[5] * U[1]) + A[8] * U[2]};
    const double AV[3] = {(A[0] * V[0] + A[1] * V[1]) + A[2] * V[2], (A[1] * V[0] + A[4] * V[1]) + A[5] * V[2],
                          (A[2] * V[0] + A[5] * V[1]) + A[8] * V[2]};
    double m00 = dot3(U, AU) - e, m01 = dot3(U, AV), m11 = dot3(V, AV) - e;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0.0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
            for (int k = 0; k < 3; ++k) out[k] = m01 * U[k] - m00 * V[k];
        } else {
            for (int k = 0; k < 3; ++k) out[k] = U[k];
        }
    } else {
        if (fmax(a11, a01) > 0.0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
            for (int k = 0; k < 3; ++k) out[k] = m11 * U[k] - m01 * V[k];
        } else {
            for (int k = 0; k < 3; ++k) out[k] = U[k];
        }
    }
}
__device__ void smallest_eigenvector(const double cov[9], double out[3]) {
    double A[9];
    double mx = cov[0];
    for (int k = 1; k < 9; ++k) if (cov[k] > mx) mx = cov[k];
    out[0] = out[1] = out[2] = 0.0;
    if (mx == 0.0) return;
    for (int k = 0; k < 9; ++k) A[k] = cov[k] / mx;
    const double norm = (A[1] * A[1] + A[2] * A[2]) + A[5] * A[5];
    if (norm > 0.0) {
        const double q = ((A[0] + A[4]) + A[8]) / 3.0;
        const double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
        const double p = sqrt((((b00 * b00 + b11 * b11) + b22 * b22) + norm * 2.0) / 6.0);
        const double c00 = b11 * b22 - A[5] * A[5], c01 = A[1] * b22 - A[5] * A[2], c02 = A[1] * A[5] - b11 * A[2];
        const double det = ((b00 * c00 - A[1] * c01) + A[2] * c02) / ((p * p) * p);
        double half = det * 0.5;
        half = half < -1.0 ? -1.0 : (half > 1.0 ? 1.0 : half);
        const double angle = acos(half) / 3.0;
        const double beta2 = cos(angle) * 2.0, beta0 = cos(angle + 2.09439510239319549) * 2.0, beta1 = -(beta0 + beta2);
        const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (half >= 0.0) {
            eigenvector0(A, e2, v2);
            if (e2 < e0 && e2 < e1) { out[0] = v2[0]; out[1] = v2[1]; out[2] = v2[2]; return; }
            eigenvector1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) { out[0] = v1[0]; out[1] = v1[1]; out[2] = v1[2]; return; }
            cross3(v1, v2, out);
        } else {
            eigenvector0(A, e0, v0);
            if (e0 < e1 && e0 < e2) { out[0] = v0[0]; out[1] = v0[1]; out[2] = v0[2]; return; }
            eigenvector1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) { out[0] = v1[0]; out[1] = v1[1]; out[2] = v1[2]; return; }
            cross3(v0, v1, out);
        }
    } else {  // diagonal
        if (A[0] < A[4] && A[0] < A[8]) out[0] = 1.0;
        else if (A[4] < A[0] && A[4] < A[8]) out[1] = 1.0;
        else out[2] = 1.0;
    }
}

constexpr int NRM_THREADS = 64;
// KDTreeSearchParamHybrid(radius, max_nn): the max_nn nearest of the points closer than the radius
// (strictly, like nanoflann's radius search), sorted by (d^2, original index), kept per thread in LDS
// columns top_d / top_j [rank][64].  Returns how many there are.
__device__ __forceinline__ int hybrid_collect(const Grid &g, const double *p, const double *__restrict__ sp,
                                              const int *__restrict__ cell_start, const int *__restrict__ cell_end,
                                              const int *__restrict__ idx_sorted, double r2, int max_nn, double *top_d, int *top_j,
                                              int t) {
    int have = 0;
    for_neighbours(g, p, cell_start, cell_end, [&](int q) {
        const double d = dist2(p, sp + 3 * (size_t)q);
        if (!(d < r2)) return;
        const int jq = idx_sorted[q];
        int r = have < max_nn ? have : max_nn - 1;
        if (have == max_nn) {  // full: only a candidate ahead of the current last one enters
            const double dl = top_d[r * NRM_THREADS + t];
            if (!(d < dl || (d == dl && jq < top_j[r * NRM_THREADS + t]))) return;
        }
        while (r > 0) {
            const double dp = top_d[(r - 1) * NRM_THREADS + t];
            const int jp = top_j[(r - 1) * NRM_THREADS + t];
            if (!(dp > d || (dp == d && jp > jq))) break;
            top_d[r * NRM_THREADS + t] = dp;
            top_j[r * NRM_THREADS + t] = jp;
            --r;
        }
        top_d[r * NRM_THREADS + t] = d;
        top_j[r * NRM_THREADS + t] = jq;
        if (have < max_nn) ++have;
    });
    return have;
}

__global__ __launch_bounds__(NRM_THREADS) void normals_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                              const int *__restrict__ cell_start,
                                                              const int *__restrict__ cell_end,
                                                              const int *__restrict__ idx_sorted, double r2, int max_nn,
                                                              const double *__restrict__ pts /* original order */,
                                                              const double *__restrict__ prior, double *__restrict__ out) {
    extern __shared__ double lds[];  // [max_nn][64] squared distances, then [max_nn][64] indices
    double *top_d = lds;
    int *top_j = (int *)(lds + (size_t)max_nn * NRM_THREADS);
    const int t = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * NRM_THREADS + t;
    if (j >= N) return;
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    const int have = hybrid_collect(g, p, sp, cell_start, cell_end, idx_sorted, r2, max_nn, top_d, top_j, t);
    double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (have >= 3) {
        double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < have; ++q) {
            const double *x = pts + 3 * (size_t)top_j[q * NRM_THREADS + t];
            cu[0] += x[0]; cu[1] += x[1]; cu[2] += x[2];
            cu[3] += x[0] * x[0]; cu[4] += x[0] * x[1]; cu[5] += x[0] * x[2];
            cu[6] += x[1] * x[1]; cu[7] += x[1] * x[2]; cu[8] += x[2] * x[2];
        }
        for (int k = 0; k < 9; ++k) cu[k] /= (double)have;
        cov[0] = cu[3] - cu[0] * cu[0]; cov[4] = cu[6] - cu[1] * cu[1]; cov[8] = cu[8] - cu[2] * cu[2];
        cov[1] = cov[3] = cu[4] - cu[0] * cu[1];
        cov[2] = cov[6] = cu[5] - cu[0] * cu[2];
        cov[5] = cov[7] = cu[7] - cu[1] * cu[2];
    }
    double n[3];
    smallest_eigenvector(cov, n);
    const int64_t i = idx_sorted[j];
    if (sqrt(dot3(n, n)) == 0.0) {
        if (prior) { n[0] = prior[3 * i]; n[1] = prior[3 * i + 1]; n[2] = prior[3 * i + 2]; }
        else { n[0] = 0.0; n[1] = 0.0; n[2] = 1.0; }
    }
    if (prior && dot3(n, prior + 3 * i) < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    out[3 * i] = n[0]; out[3 * i + 1] = n[1]; out[3 * i + 2] = n[2];
}

// ------------------------------------------------------------------ FPFH (compute_fpfh_feature)
// Open3D 0.18 Feature.cpp: per point the SPFH histogram (3 x 11 bins of the Darboux-frame angles to
// the hybrid-search neighbours, the point itself skipped), then FPFH = own SPFH + the neighbours'
// SPFH weighted by 1 / squared distance, each third rescaled to 100.  Neighbour lists are written
// once ([rank][N], nearest first, ties by index) and read by both passes.
__global__ __launch_bounds__(NRM_THREADS) void hybrid_list_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                                  const int *__restrict__ cell_start,
                                                                  const int *__restrict__ cell_end,
                                                                  const int *__restrict__ idx_sorted, double r2, int max_nn,
                                                                  int *__restrict__ nbr_j, double *__restrict__ nbr_d,
                                                                  int *__restrict__ nbr_n) {
    extern __shared__ double lds[];
    double *top_d = lds;
    int *top_j = (int *)(lds + (size_t)max_nn * NRM_THREADS);
    const int t = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * NRM_THREADS + t;
    if (j >= N) return;
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    const int have = hybrid_collect(g, p, sp, cell_start, cell_end, idx_sorted, r2, max_nn, top_d, top_j, t);
    const int64_t i = idx_sorted[j];
    nbr_n[i] = have;
    for (int k = 0; k < have; ++k) {
        nbr_j[(size_t)k * N + i] = top_j[k * NRM_THREADS + t];
        nbr_d[(size_t)k * N + i] = top_d[k * NRM_THREADS + t];
    }
}

// The same lists, one WAVE per query: the 64 lanes fetch a cell's points together, the candidates
// closer than the radius are packed into LDS (ballot + prefix count), and the max_nn nearest are
// extracted one by one with a wave-wide lexicographic argmin over (d^2, original index) -- the order
// of hybrid_collect -- every lane keeping the minimum of its strided share.  A query with more
// candidates than the LDS share raises `overflow`; the host then tries the larger share and, past that,
// the thread-per-query kernel.
// WCAP candidates per query wave (12 B each), WPB query waves per workgroup: <1024, 4> = 48 KB (three
// workgroups per CU) first, <4096, 2> = 96 KB for clouds with denser neighbourhoods
template <int HYB_WCAP, int HYB_WPB>
__global__ __launch_bounds__(HYB_WPB * 64) void hybrid_list_wave_kernel(Grid g, const double *__restrict__ sp, int64_t N,
                                                                       const int *__restrict__ cell_start,
                                                                       const int *__restrict__ cell_end,
                                                                       const int *__restrict__ idx_sorted, double r2, int max_nn,
                                                                       int *__restrict__ nbr_j, double *__restrict__ nbr_d,
                                                                       int *__restrict__ nbr_n, int *__restrict__ overflow) {
    __shared__ double cd_all[HYB_WPB][HYB_WCAP];
    __shared__ int cj_all[HYB_WPB][HYB_WCAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * HYB_WPB + wave;
    if (j >= N) return;  // wave-uniform
    double *cd = cd_all[wave];
    int *cj = cj_all[wave];
    const double p[3] = {sp[3 * j], sp[3 * j + 1], sp[3 * j + 2]};
    const int cx = grid_axis(p[0], g.lo[0], g.cell, g.dim[0]), cy = grid_axis(p[1], g.lo[1], g.cell, g.dim[1]),
              cz = grid_axis(p[2], g.lo[2], g.cell, g.dim[2]);
    int n_c = 0;
    bool over = false;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dim[2] - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dim[1] - 1); ++y)
            for (int x = max(cx - 1, 0); x <= min(cx + 1, g.dim[0] - 1); ++x) {
                const int c = x + g.dim[0] * (y + g.dim[1] * z);
                const int e = cell_end[c];
                for (int q0 = cell_start[c]; q0 < e; q0 += 64) {  // wave-uniform bounds
                    const int q = q0 + lane;
                    double d = 0.0;
                    bool in = false;
                    if (q < e) { d = dist2(p, sp + 3 * (size_t)q); in = d < r2; }
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                    const int at = n_c + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (in && at < HYB_WCAP) { cd[at] = d; cj[at] = idx_sorted[q]; }
                    n_c += __builtin_popcountll(m);
                    over |= n_c > HYB_WCAP;
                }
            }
    if (over) {  // wave-uniform
        if (lane == 0) atomicOr(overflow, 1);
        return;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS writes have landed
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double lmin = inf;
    int lj = 0x7FFFFFFF, lidx = -1;
    for (int q = lane; q < n_c; q += 64) {
        const double d = cd[q];
        const int jq = cj[q];
        if (d < lmin || (d == lmin && jq < lj)) { lmin = d; lj = jq; lidx = q; }
    }
    const int have = n_c < max_nn ? n_c : max_nn;
    const int64_t i = idx_sorted[j];
    for (int r = 0; r < have; ++r) {
        double v = lmin;
        int vj = lj, owner = lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(v, off, 64);
            const int oj = __shfl_xor(vj, off, 64), oo = __shfl_xor(owner, off, 64);
            if (ov < v || (ov == v && oj < vj)) { v = ov; vj = oj; owner = oo; }
        }
        if (lane == owner) {  // hand the entry over and find the share's next minimum
            nbr_j[(size_t)r * N + i] = vj;
            nbr_d[(size_t)r * N + i] = v;
            cd[lidx] = inf;
            cj[lidx] = 0x7FFFFFFF;
            lmin = inf; lj = 0x7FFFFFFF; lidx = -1;
            for (int q = lane; q < n_c; q += 64) {
                const double d = cd[q];
                const int jq = cj[q];
                if (d < lmin || (d == lmin && jq < lj)) { lmin = d; lj = jq; lidx = q; }
            }
        }
    }
    if (lane == 0) nbr_n[i] = have;
}

// Feature.cpp ComputePairFeatures, operation for operation (oracle/features.c pair_features)
__device__ __forceinline__ void pair_features_dev(const double *p1, const double *n1, const double *p2, const double *n2,
                                                  double r[4]) {
    double dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    r[0] = r[1] = r[2] = 0.0;
    r[3] = sqrt(dot3(dp, dp));
    if (r[3] == 0.0) return;
    double a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
    const double angle1 = dot3(a, dp) / r[3], angle2 = dot3(b, dp) / r[3];
    if (acos(fabs(angle1)) > acos(fabs(angle2))) {
        for (int k = 0; k < 3; ++k) { a[k] = n2[k]; b[k] = n1[k]; dp[k] = -dp[k]; }
        r[2] = -angle2;
    } else {
        r[2] = angle1;
    }
    double v[3], w[3];
    cross3(dp, a, v);
    const double vn = sqrt(dot3(v, v));
    if (vn == 0.0) { r[0] = r[1] = r[2] = r[3] = 0.0; return; }
    for (int k = 0; k < 3; ++k) v[k] /= vn;
    cross3(a, v, w);
    r[1] = dot3(v, b);
    r[0] = atan2(dot3(w, b), dot3(a, b));
}

__device__ __forceinline__ int bin11(double x) {
    int h = (int)floor(x);
    return h < 0 ? 0 : (h >= 11 ? 10 : h);
}

__global__ __launch_bounds__(NRM_THREADS) void spfh_kernel(const double *__restrict__ pts, const double *__restrict__ nrm,
                                                           int64_t N, const int *__restrict__ nbr_j,
                                                           const int *__restrict__ nbr_n, double *__restrict__ spfh) {
    __shared__ double hist[33][NRM_THREADS];
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * NRM_THREADS + t;
    if (i >= N) return;
#pragma unroll
    for (int k = 0; k < 33; ++k) hist[k][t] = 0.0;
    const int n = nbr_n[i];
    if (n > 1) {
        const double incr = 100.0 / (double)(n - 1);
        const double p1[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, n1[3] = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]};
        for (int k = 1; k < n; ++k) {
            const size_t q = (size_t)nbr_j[(size_t)k * N + i];
            double pf[4];
            pair_features_dev(p1, n1, pts + 3 * q, nrm + 3 * q, pf);
            hist[bin11(11.0 * (pf[0] + 3.14159265358979323846) / (2.0 * 3.14159265358979323846))][t] += incr;
            hist[11 + bin11(11.0 * (pf[1] + 1.0) * 0.5)][t] += incr;
            hist[22 + bin11(11.0 * (pf[2] + 1.0) * 0.5)][t] += incr;
        }
    }
#pragma unroll
    for (int k = 0; k < 33; ++k) spfh[33 * i + k] = hist[k][t];
}

__global__ __launch_bounds__(NRM_THREADS) void fpfh_kernel(int64_t N, const int *__restrict__ nbr_j, const double *__restrict__ nbr_d,
                                                           const int *__restrict__ nbr_n, const double *__restrict__ spfh,
                                                           double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * NRM_THREADS + threadIdx.x;
    if (i >= N) return;
    double f[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) f[k] = 0.0;
    const int n = nbr_n[i];
    if (n > 1) {
        double sum[3] = {0.0, 0.0, 0.0};
        for (int k = 1; k < n; ++k) {
            const double dist = nbr_d[(size_t)k * N + i];
            if (dist == 0.0) continue;
            const double *h = spfh + 33 * (size_t)nbr_j[(size_t)k * N + i];
#pragma unroll
            for (int j = 0; j < 33; ++j) {
                const double val = h[j] / dist;
                sum[j / 11] += val;
                f[j] += val;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (sum[j] != 0.0) sum[j] = 100.0 / sum[j];
#pragma unroll
        for (int j = 0; j < 33; ++j) {
            f[j] *= sum[j / 11];
            f[j] += spfh[33 * i + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 33; ++j) out[33 * i + j] = f[j];
}

// nearest target feature of every source feature: squared L2 over the 33 components summed in order
// (float64, as KDTreeFlann::SearchKNN on the Feature matrix), ties to the lower index.  Grid = source
// blocks x target chunks: a workgroup holds 64 source features in registers (one per thread), streams
// its chunk of FM_CHUNK target features through LDS and leaves (d^2, index) per source and chunk; a
// second kernel takes the minimum over the chunks in chunk order (strict <, so the lowest index wins a tie).
constexpr int FM_TILE = 64;
constexpr int FM_CHUNK = 1024;
__global__ __launch_bounds__(64) void feature_match_kernel(const double *__restrict__ fs, int64_t Ns, const double *__restrict__ ft,
                                                           int64_t Nt, double *__restrict__ part_d, int32_t *__restrict__ part_j) {
    __shared__ double tile[FM_TILE][33];
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + t;
    const int64_t c0 = (int64_t)blockIdx.y * FM_CHUNK, c1 = c0 + FM_CHUNK < Nt ? c0 + FM_CHUNK : Nt;
    double a[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) a[k] = i < Ns ? fs[33 * i + k] : 0.0;
    double best = __longlong_as_double(0x7FF0000000000000ll);
    int bj = -1;
    for (int64_t j0 = c0; j0 < c1; j0 += FM_TILE) {
        const int m = (int)(c1 - j0 < FM_TILE ? c1 - j0 : FM_TILE);
        __syncthreads();
        for (int e = t; e < m * 33; e += 64) tile[e / 33][e % 33] = ft[33 * j0 + e];
        __syncthreads();
        for (int jj = 0; jj < m; ++jj) {
            double d = 0.0;
#pragma unroll
            for (int k = 0; k < 33; ++k) {
                const double e = a[k] - tile[jj][k];
                d += e * e;
            }
            if (d < best) { best = d; bj = (int)(j0 + jj); }
        }
    }
    if (i < Ns) {
        part_d[(size_t)blockIdx.y * Ns + i] = best;
        part_j[(size_t)blockIdx.y * Ns + i] = bj;
    }
}

__global__ void feature_match_fold_kernel(int64_t Ns, int n_chunks, const double *__restrict__ part_d,
                                          const int32_t *__restrict__ part_j, int32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ns) return;
    double best = __longlong_as_double(0x7FF0000000000000ll);
    int bj = -1;
    for (int c = 0; c < n_chunks; ++c) {
        const double d = part_d[(size_t)c * Ns + i];
        if (d < best) { best = d; bj = part_j[(size_t)c * Ns + i]; }
    }
    idx[i] = bj;
}

// ------------------------------------------------------------------ plane RANSAC
__host__ __device__ inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline void sample3(unsigned long long seed, long long t, long long N, long long out[3]) {
    unsigned long long s = splitmix64(seed ^ splitmix64((unsigned long long)t));
    int n = 0;
    while (n < 3) {
        s = splitmix64(s);
        const long long c = (long long)(s % (unsigned long long)N);
        bool dup = false;
        for (int q = 0; q < n; ++q) dup |= (out[q] == c);
        if (!dup) out[n++] = c;
    }
}
__host__ __device__ inline bool triangle_plane(const double *p0, const double *p1, const double *p2, double pl[4]) {
    const double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    const double n = sqrt((x * x + y * y) + z * z);
    if (!(n > 0.0)) return false;
    pl[0] = x / n; pl[1] = y / n; pl[2] = z / n;
    pl[3] = -((pl[0] * p0[0] + pl[1] * p0[1]) + pl[2] * p0[2]);
    return true;
}
__host__ __device__ inline double plane_dist(const double pl[4], const double *p) {
    return fabs(((pl[0] * p[0] + pl[1] * p[1]) + pl[2] * p[2]) + pl[3]);
}

// Two kernels.  The first forms every iteration's plane (a thread per iteration; a plane that does not exist --
// collinear sample -- gets a NaN offset, which no distance test passes, and the count -1).  The second counts
// inliers tile by tile: a workgroup holds 256 points in registers and runs RS_PLANES planes from LDS over them,
// so the cloud is read once per RS_PLANES iterations (a workgroup per iteration streaming the whole cloud
// moved 228 MB through L2 for 9.5 k points x 1000 iterations: 166 us).  Integer adds: any order, same counts.
constexpr int RS_PLANES = 64;
__global__ void ransac_plane_kernel(const double *__restrict__ pts, long long N, unsigned long long seed, int n_iter,
                                    double *__restrict__ planes, int *__restrict__ counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_iter) return;
    long long s[3];
    double pl[4];
    sample3(seed, t, N, s);
    const bool ok = triangle_plane(pts + 3 * s[0], pts + 3 * s[1], pts + 3 * s[2], pl);
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    for (int k = 0; k < 4; ++k) planes[4 * (size_t)t + k] = ok ? pl[k] : (k == 3 ? nan : 0.0);
    counts[t] = ok ? 0 : -1;
}
__global__ __launch_bounds__(256) void ransac_count_kernel(const double *__restrict__ pts, long long N, double threshold, int n_iter,
                                                           const double *__restrict__ planes, int *__restrict__ counts) {
    __shared__ double pl_s[RS_PLANES][4];
    __shared__ int cnt_s[RS_PLANES];
    const int t0 = blockIdx.y * RS_PLANES, nt = min(RS_PLANES, n_iter - t0);
    if (threadIdx.x < RS_PLANES) {
        cnt_s[threadIdx.x] = 0;
        if ((int)threadIdx.x < nt)
            for (int k = 0; k < 4; ++k) pl_s[threadIdx.x][k] = planes[4 * (size_t)(t0 + threadIdx.x) + k];
    }
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool have = i < N;
    double p[3] = {0.0, 0.0, 0.0};
    if (have) { p[0] = pts[3 * i]; p[1] = pts[3 * i + 1]; p[2] = pts[3 * i + 2]; }
    for (int t = 0; t < nt; ++t) {
        const double pl[4] = {pl_s[t][0], pl_s[t][1], pl_s[t][2], pl_s[t][3]};
        const int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64(have && plane_dist(pl, p) < threshold));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt_s[t], c);
    }
    __syncthreads();
    if ((int)threadIdx.x < nt && cnt_s[threadIdx.x]) atomicAdd(&counts[t0 + threadIdx.x], cnt_s[threadIdx.x]);
}

// the best iteration -- most inliers, the earliest on ties -- with its three sampled points and its plane (the one the
// counts were taken with), in one small block: out[0] = iteration (-1: no plane at all), out[1..9] = the points,
// out[10..13] = the plane.  The host-array entry reads the points back and forms the plane itself; the device-resident
// chain hands the block to plane_keep_kernel and never asks.
__global__ __launch_bounds__(256) void ransac_best_kernel(const int *__restrict__ counts, int n_iter, const double *__restrict__ pts,
                                                          long long N, unsigned long long seed, const double *__restrict__ planes,
                                                          double *__restrict__ out) {
    __shared__ int cnt_s[256], it_s[256];
    int bc = -1, bt = -1;
    for (int t = threadIdx.x; t < n_iter; t += 256)
        if (counts[t] > bc) { bc = counts[t]; bt = t; }  // ascending t: the earliest of the thread's share
    cnt_s[threadIdx.x] = bc;
    it_s[threadIdx.x] = bt;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const int oc = cnt_s[threadIdx.x + w], ot = it_s[threadIdx.x + w];
            if (oc > cnt_s[threadIdx.x] || (oc == cnt_s[threadIdx.x] && ot >= 0 && (it_s[threadIdx.x] < 0 || ot < it_s[threadIdx.x]))) {
                cnt_s[threadIdx.x] = oc;
                it_s[threadIdx.x] = ot;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int best = it_s[0];
        out[0] = (double)best;
        if (best >= 0) {
            long long s[3];
            sample3(seed, best, N, s);
            for (int q = 0; q < 3; ++q)
                for (int k = 0; k < 3; ++k) out[1 + 3 * q + k] = pts[3 * s[q] + k];
            for (int k = 0; k < 4; ++k) out[10 + k] = planes[4 * (size_t)best + k];
        }
    }
}

// GetPlaneFromPoints, oracle/cloudops.c pedp_oracle_plane_from_points: centroid of the inliers, their six second moments
// about it, the largest-determinant closed form.  Every sum is taken in BLOCKS of the cloud -- the 256 points [256 j, 256 j +
// 256) contribute their value if they are inliers and zero otherwise, a block is added in a fixed binary tree, t[i] += t[i + w]
// for w = 128 ... 1, the block sums are added in order of j -- the order the device takes (refit_* kernels below:
// pedp_preprocess_source refits without a trip to the host and without compacting the inliers); this host statement serves
// pedp_segment_plane.  Differs from Open3D's running sums by rounding only.  idx ascends.
inline double block_tree_sum(double *t) {
    for (int w = 128; w >= 1; w >>= 1)
        for (int i = 0; i < w; ++i) t[i] += t[i + w];
    return t[0];
}
__host__ __device__ inline void plane_closed_form(const double c[3], const double m[6], double pl[4]) {
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    const double xx = m[0], xy = m[1], xz = m[2], yy = m[3], yz = m[4], zz = m[5];
    const double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, cc;
    if (det_x >= det_y && det_x >= det_z) { a = det_x; b = xz * yz - xy * zz; cc = xy * yz - xz * yy; }
    else if (det_y >= det_z) { a = xz * yz - xy * zz; b = det_y; cc = xy * xz - yz * xx; }
    else { a = xy * yz - xz * yy; b = xy * xz - yz * xx; cc = det_z; }
    const double nrm = sqrt((a * a + b * b) + cc * cc);
    if (!(nrm > 0.0)) return;
    pl[0] = a / nrm; pl[1] = b / nrm; pl[2] = cc / nrm;
    pl[3] = -((pl[0] * c[0] + pl[1] * c[1]) + pl[2] * c[2]);
}
void blocked_sums(const double *pts, const int32_t *idx, int64_t n, const double *r0, int nq, double *out) {
    const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {0, 1, 2, 1, 2, 2};
    double t[6][256];
    for (int q = 0; q < nq; ++q) out[q] = 0.0;
    int64_t i = 0;
    while (i < n) {
        const int64_t blk = idx[i] / 256;
        for (int q = 0; q < nq; ++q)
            for (int k = 0; k < 256; ++k) t[q][k] = 0.0;
        for (; i < n && idx[i] / 256 == blk; ++i) {
            const double *p = pts + 3 * (int64_t)idx[i];
            const int k = (int)(idx[i] % 256);
            if (!r0) { t[0][k] = p[0]; t[1][k] = p[1]; t[2][k] = p[2]; continue; }
            const double r[3] = {p[0] - r0[0], p[1] - r0[1], p[2] - r0[2]};
            for (int q = 0; q < nq; ++q) t[q][k] = r[A[q]] * r[B[q]];
        }
        for (int q = 0; q < nq; ++q) out[q] += block_tree_sum(t[q]);
    }
}
void plane_from_points(const double *pts, const int32_t *idx, int64_t n, double pl[4]) {
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    if (n < 3) return;
    double c[3], m[6];
    blocked_sums(pts, idx, n, nullptr, 3, c);
    for (int k = 0; k < 3; ++k) c[k] /= (double)n;
    blocked_sums(pts, idx, n, c, 6, m);
    plane_closed_form(c, m, pl);
}

// ---- the same on the device: per block of 256 points the tree sums of NV values per inlier (and, NV = 3, the block's
// inlier count), then the blocks added in order.
// NV = 3: the coordinates; NV = 6: the products of the residuals about the centroid in res[0..2].  part: NV (+ 1) per block.
template <int NV>
__global__ __launch_bounds__(256) void refit_block_kernel(const double *__restrict__ pts, int64_t N, const double *__restrict__ best, double thr,
                                                          const double *__restrict__ res, double *__restrict__ part) {
    constexpr int NP = NV == 3 ? 4 : 6;
    __shared__ double t[NP][256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = 0.0;
    if (i < N && best[0] >= 0.0) {
        const double pl[4] = {best[10], best[11], best[12], best[13]};
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        const double p[3] = {x, y, z};
        if (plane_dist(pl, p) < thr) {
            if constexpr (NV == 3) { v[0] = x; v[1] = y; v[2] = z; v[3] = 1.0; }
            else {
                const double rx = x - res[0], ry = y - res[1], rz = z - res[2];
                v[0] = rx * rx; v[1] = rx * ry; v[2] = rx * rz; v[3] = ry * ry; v[4] = ry * rz; v[5] = rz * rz;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) t[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {  // t[i] += t[i + w], i < w: the oracle's tree
        if ((int)threadIdx.x < w)
#pragma unroll
            for (int k = 0; k < NP; ++k) t[k][threadIdx.x] += t[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < NP) part[NP * (size_t)blockIdx.x + threadIdx.x] = t[threadIdx.x][0];
}
// NV = 3: res[0..2] = centroid, res[7] = number of inliers.  NV = 6: res[3..6] = the plane (res[0..2], res[7] are read).
template <int NV>
__global__ __launch_bounds__(256) void refit_fold_kernel(const double *__restrict__ part, int64_t n_blocks, double *__restrict__ res) {
    constexpr int NP = NV == 3 ? 4 : 6;
    __shared__ double stage[NP * 1024];
    __shared__ double m[6];
    double s = 0.0;
    for (int64_t base = 0; base < n_blocks; base += 1024) {  // the block sums through LDS, then added in order: a lane per value
        const int nb = (int)(n_blocks - base < 1024 ? n_blocks - base : 1024);
        for (int j = threadIdx.x; j < NP * nb; j += 256) stage[j] = part[NP * (size_t)base + j];
        __syncthreads();
        if ((int)threadIdx.x < NP)
            for (int b = 0; b < nb; b += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = b + j < nb ? stage[NP * (b + j) + threadIdx.x] : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (b + j < nb) s += v[j];
            }
        __syncthreads();
    }
    if (NV == 3) {  // (the count is a sum of ones: exact in any order)
        if (threadIdx.x == 3) m[0] = s;
        __syncthreads();
        if (threadIdx.x < 3) res[threadIdx.x] = s / m[0];
        if (threadIdx.x == 3) res[7] = s;
        return;
    }
    if (threadIdx.x < 6) m[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double pl[4] = {0, 0, 0, 0};
        if (res[7] >= 3.0) {
            const double c[3] = {res[0], res[1], res[2]};
            plane_closed_form(c, m, pl);
        }
        for (int k = 0; k < 4; ++k) res[3 + k] = pl[k];
    }
}

// ------------------------------------------------------------------ compute_average_normal on the device
// src/pose_estimation.py:314-321: the normals' averages over a 10-unit voxel grid (voxel_down_sample: members summed in
// point order, divided by the count; voxels in ascending (ix, iy, iz)), then numpy's mean over the voxels (row by row).
// The sort-based grid above costs two dozen launches and two trips to the host for the 38 k points of a frame; here the
// voxels are cells of a dense grid (ids ascending in (ix, iy, iz)), members are counted and placed, and a thread per
// cell adds its members in ascending point index by repeated selection (a few dozen members).  No trip to the host:
// the grid's origin (the cloud's own minimum, less half a voxel: voxel_down_sample's) is folded on the device, its
// dimensions come from a box the caller knows to contain the cloud.  Same sums in the same order as the grid above.
struct AvgGrid { int dim[3]; double voxel; };
__global__ __launch_bounds__(256) void avgn_origin_kernel(const double *__restrict__ part /* n_part x 6 */, int n_part, double voxel,
                                                          double *__restrict__ org) {
    __shared__ double red[4][3];
    double lo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = (int)threadIdx.x < n_part ? part[6 * threadIdx.x + k] : __longlong_as_double(0x7FF0000000000000ll);
        for (int b = threadIdx.x + 256; b < n_part; b += 256) lo[k] = part[6 * b + k] < lo[k] ? part[6 * b + k] : lo[k];
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(lo[k], off, 64);
            lo[k] = o < lo[k] ? o : lo[k];
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) red[threadIdx.x >> 6][k] = lo[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = red[w][threadIdx.x] < v ? red[w][threadIdx.x] : v;
        org[threadIdx.x] = v - voxel * 0.5;
    }
}
__device__ __forceinline__ unsigned avgn_cell(const double *p, const double *org, const AvgGrid &g) {
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const long long v = (long long)floor((p[k] - org[k]) / g.voxel);
        c[k] = v < 0 ? 0 : (v >= g.dim[k] ? g.dim[k] - 1 : (int)v);  // (never clamps: the box contains the cloud)
    }
    return (unsigned)((c[0] * g.dim[1] + c[1]) * g.dim[2] + c[2]);
}
__global__ void avgn_count_kernel(const double *__restrict__ pts, int64_t N, const double *__restrict__ org, AvgGrid g,
                                  unsigned *__restrict__ cell, int *__restrict__ slot, unsigned *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned c = avgn_cell(pts + 3 * i, org, g);
    cell[i] = c;
    slot[i] = (int)atomicAdd(&count[c], 1u);
}
__global__ void avgn_place_kernel(int64_t N, const unsigned *__restrict__ cell, const int *__restrict__ slot, const unsigned *__restrict__ begin,
                                  int *__restrict__ member) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) member[begin[cell[i]] + (unsigned)slot[i]] = (int)i;
}
struct NonEmpty { __host__ __device__ unsigned operator()(unsigned n) const { return n != 0u ? 1u : 0u; } };
constexpr int AVGN_MAX_MEMBERS = 512, AVGN_WPB = 4;
// A wave per 64 consecutive cells; every non-empty cell of them is served by the whole wave: the members' indices go to
// LDS, every member's rank among them is the number of smaller indices, its normal goes to the LDS row of that rank, and
// three lanes -- one per component -- add the rows in order (eight reads in flight, then their additions).
__global__ __launch_bounds__(AVGN_WPB * 64) void avgn_mean_kernel(int64_t n_cells, const unsigned *__restrict__ begin,
                                                                   const unsigned *__restrict__ rank, const int *__restrict__ member,
                                                                   const double *__restrict__ nrm, double *__restrict__ means,
                                                                   int *__restrict__ overflow) {
    __shared__ int idx_s[AVGN_WPB][AVGN_MAX_MEMBERS];
    __shared__ double row_s[AVGN_WPB][3 * AVGN_MAX_MEMBERS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t c0 = ((int64_t)blockIdx.x * AVGN_WPB + wv) * 64, cl = c0 + lane;
    const unsigned b = cl < n_cells ? begin[cl] : 0u, e = cl < n_cells ? begin[cl + 1] : 0u;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(e > b);
    int *ids = idx_s[wv];
    double *rows = row_s[wv];
    while (todo) {  // wave-uniform
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned cb = __builtin_amdgcn_readlane(b, j), k = __builtin_amdgcn_readlane(e, j) - cb;
        if (k > (unsigned)AVGN_MAX_MEMBERS) { if (lane == 0) *overflow = 1; continue; }
        for (unsigned t = lane; t < k; t += 64) ids[t] = member[cb + t];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // (this wave's LDS writes are complete: lgkmcnt(0))
        __builtin_amdgcn_wave_barrier();
        for (unsigned t = lane; t < k; t += 64) {
            const int mine = ids[t];
            unsigned r = 0;
            for (unsigned u = 0; u < k; ++u) r += ids[u] < mine ? 1u : 0u;
            rows[3 * r] = nrm[3 * (size_t)mine];
            rows[3 * r + 1] = nrm[3 * (size_t)mine + 1];
            rows[3 * r + 2] = nrm[3 * (size_t)mine + 2];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane < 3) {
            double s = 0.0;
            for (unsigned t = 0; t < k; t += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = t + q < k ? rows[3 * (t + q) + lane] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (t + q < k) s += v[q];
            }
            means[3 * (size_t)rank[c0 + j] + lane] = s / (double)k;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// numpy's mean over axis 0 adds the rows one after the other: a serial chain of a few thousand additions, which one GPU
// thread takes 170 us over.  The means are therefore written straight into pinned host memory by the kernel above, and the
// host adds them after the stream's next wait (microseconds); this kernel only reports how many there are.
__global__ void avgn_rows_kernel(int64_t n_cells, const unsigned *__restrict__ rank, const unsigned *__restrict__ count,
                                 const int *__restrict__ overflow, double *__restrict__ res) {
    res[3] = (double)((int64_t)rank[n_cells - 1] + (count[n_cells - 1] != 0u ? 1 : 0));
    res[4] = (double)*overflow;
}

int check_cloud(pedp_ctx_t c, const double *pts, int64_t N, const char *who) {
    PEDP_REQUIRE(c, "%s: null context", who);
    PEDP_REQUIRE(N >= 0 && N < (int64_t)1 << 31, "%s: N out of range", who);
    PEDP_REQUIRE(pts || N == 0, "%s: null points", who);
    return PEDP_OK;
}

// Bounding box of device-resident points: per-workgroup partial minima / maxima (the host loop's
// comparisons, so NaN coordinates are passed over the same way), folded on the host from the pinned block.
// One small read-back instead of a host pass over the caller's array (0.5 ms for 365 k points).
constexpr int BND_BLOCKS = 256;
__global__ __launch_bounds__(256) void bounds_kernel(const double *__restrict__ pts, int64_t N, double *__restrict__ part) {
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = pts[3 * i + k];
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = __shfl_xor(lo[k], off, 64), b = __shfl_xor(hi[k], off, 64);
            if (a < lo[k]) lo[k] = a;
            if (b > hi[k]) hi[k] = b;
        }
    __shared__ double red[4][6];
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) {
            const double o = red[w][threadIdx.x];
            if (threadIdx.x < 3 ? o < v : o > v) v = o;
        }
        part[6 * blockIdx.x + threadIdx.x] = v;
    }
}

int bounds_device(pedp_ctx_t c, const double *d_pts, int64_t N, double *d_part /* 6 * BND_BLOCKS */, double lo[3], double hi[3]) {
    hipLaunchKernelGGL(bounds_kernel, dim3(BND_BLOCKS), dim3(256), 0, c->stream, d_pts, N, d_part);
    PEDP_HIP_CHECK(hipGetLastError());
    double *h = (double *)((char *)c->pinned + 16384);
    PEDP_HIP_CHECK(hipMemcpyAsync(h, d_part, sizeof(double) * 6 * BND_BLOCKS, hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) { lo[k] = h[k]; hi[k] = h[3 + k]; }
    for (int b = 1; b < BND_BLOCKS; ++b)
        for (int k = 0; k < 3; ++k) {
            if (h[6 * b + k] < lo[k]) lo[k] = h[6 * b + k];
            if (h[6 * b + 3 + k] > hi[k]) hi[k] = h[6 * b + 3 + k];
        }
    return PEDP_OK;
}

void bounds(const double *pts, int64_t N, double lo[3], double hi[3]) {
    for (int k = 0; k < 3; ++k) lo[k] = hi[k] = pts[k];
    for (int64_t i = 1; i < N; ++i)
        for (int k = 0; k < 3; ++k) {
            const double v = pts[3 * i + k];
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
}

// Large clouds: the points go up first (into their own scratch, so the workspace can still be sized by the grid)
// and their box comes from the device copy; small ones keep the host pass.  Returns the device copy or nullptr.
constexpr int64_t BND_DEVICE_MIN = 32768;
int bounds_of(pedp_ctx_t c, const double *pts, int64_t N, double lo[3], double hi[3], double **d_in) {
    *d_in = nullptr;
    if (N < BND_DEVICE_MIN) { bounds(pts, N, lo, hi); return PEDP_OK; }
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    int st = c->ops_in.reserve(a256(sizeof(double) * 3 * (size_t)N) + a256(sizeof(double) * 6 * BND_BLOCKS));
    if (st) return st;
    double *d = (double *)c->ops_in.ptr, *d_part = (double *)((char *)c->ops_in.ptr + a256(sizeof(double) * 3 * (size_t)N));
    { int up_ = pedp_upload(c, d, pts, sizeof(double) * 3 * (size_t)N); if (up_) return up_; }
    st = bounds_device(c, d, N, d_part, lo, hi);
    if (st) return st;
    *d_in = d;
    return PEDP_OK;
}

// grid over the bounding box with the given cell size, coarsened until it has at most 2^24 cells
int make_grid(const double lo[3], const double hi[3], double cell, Grid &g, int64_t &n_cells, const char *who) {
    g.cell = cell;
    while (true) {
        double cells = 1.0;
        for (int k = 0; k < 3; ++k) {
            PEDP_REQUIRE(std::isfinite(lo[k]) && std::isfinite(hi[k]), "%s: non-finite coordinates", who);
            g.lo[k] = lo[k];
            const double d = std::floor((hi[k] - lo[k]) / g.cell) + 1.0;
            g.dim[k] = d < 1.0 ? 1 : (d > 1e9 ? 1000000000 : (int)d);
            cells *= (double)g.dim[k];
        }
        if (cells <= 16777216.0) break;
        g.cell *= 2.0;  // a coarser grid is still a valid (slower) neighbourhood index
    }
    n_cells = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    return PEDP_OK;
}

#define PEDP_ROCPRIM(expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pedp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return PEDP_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

}  // namespace

// ------------------------------------------------------------------ device-to-device cores
// Every operation below takes its points where they already are -- in device memory -- and leaves its result
// there (inside the context's `ops` scratch, valid until the next operation).  The C entry points with host
// arrays are thin wrappers (upload, core, download); pedp_preprocess_source chains the cores without the
// points ever visiting the host.
namespace {

// a small device block of the context for partial bounds and counters (never inside `ops`, which the cores re-carve)
int small_block(pedp_ctx_t c, double **d_part) {
    int st = c->ops_small.reserve(a256(sizeof(double) * 6 * BND_BLOCKS) + 4096);
    if (st) return st;
    *d_part = (double *)c->ops_small.ptr;
    return PEDP_OK;
}

int voxel_core(pedp_ctx_t c, const double *d_pts, const double *d_nrm, int64_t N, double voxel_size, double **out_pts,
               double **out_nrm, int64_t *n_out, double *box_lo = nullptr, double *box_hi = nullptr) {
    double lo[3], hi[3], *d_part = nullptr;
    int rc = small_block(c, &d_part);
    if (rc) return rc;
    rc = bounds_device(c, d_pts, N, d_part, lo, hi);
    if (rc) return rc;
    for (int k = 0; k < 3 && box_lo && box_hi; ++k) { box_lo[k] = lo[k]; box_hi[k] = hi[k]; }  // the averages lie inside it
    int bits[3];
    for (int k = 0; k < 3; ++k) {
        PEDP_REQUIRE(std::isfinite(lo[k]) && std::isfinite(hi[k]), "pedp_voxel_down_sample: non-finite coordinates");
        lo[k] = lo[k] - voxel_size * 0.5;
        PEDP_REQUIRE((hi[k] - lo[k]) / voxel_size < 2097151.0, "pedp_voxel_down_sample: voxel_size is too small");
        // the largest cell index of the axis, one to spare for the division's rounding
        const unsigned long long top = (unsigned long long)std::floor((hi[k] - lo[k]) / voxel_size) + 1ull;
        bits[k] = 1;
        while (bits[k] < 21 && (top >> bits[k])) ++bits[k];
    }
    const int key_bits = bits[0] + bits[1] + bits[2];
    const unsigned n = (unsigned)N;
    size_t tmp_sort = 0, tmp_rle = 0, tmp_scan = 0;
    const bool passes = N >= 131072;  // below, the merge sort's launches are the shorter ones (a radix pass takes 20 us whatever the size)
    if (passes)
        PEDP_ROCPRIM(rocprim::radix_sort_pairs<RadixPasses>(nullptr, tmp_sort, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                            (int *)nullptr, (int *)nullptr, n, 0, key_bits, c->stream));
    else
        PEDP_ROCPRIM(rocprim::radix_sort_pairs(nullptr, tmp_sort, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr,
                                               (int *)nullptr, n, 0, key_bits, c->stream));
    PEDP_ROCPRIM(rocprim::run_length_encode(nullptr, tmp_rle, (unsigned long long *)nullptr, n, (unsigned long long *)nullptr,
                                            (unsigned *)nullptr, (unsigned *)nullptr, c->stream));
    PEDP_ROCPRIM(rocprim::exclusive_scan(nullptr, tmp_scan, (unsigned *)nullptr, (unsigned *)nullptr, 0u, n,
                                         rocprim::plus<unsigned>(), c->stream));
    size_t tmp = tmp_sort > tmp_rle ? tmp_sort : tmp_rle;
    if (tmp_scan > tmp) tmp = tmp_scan;
    const size_t need = a256(sizeof(double) * 3 * N) * 2 + a256(sizeof(unsigned long long) * N) * 3 + a256(sizeof(int) * N) * 2 +
                        a256(sizeof(unsigned) * N) * 2 + 256 + a256(tmp) + 4096;
    int st = c->ops.reserve(need);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    double *d_out = cv.take<double>(3 * (size_t)N), *d_outn = cv.take<double>(3 * (size_t)N);
    unsigned long long *key = cv.take<unsigned long long>(N), *key_s = cv.take<unsigned long long>(N),
                       *uniq = cv.take<unsigned long long>(N);
    int *val = cv.take<int>(N), *val_s = cv.take<int>(N);
    unsigned *counts = cv.take<unsigned>(N), *offsets = cv.take<unsigned>(N), *n_runs = cv.take<unsigned>(1);
    void *d_tmp = cv.take<char>(tmp);
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(grid), dim3(256), 0, c->stream, d_pts, N, lo[0], lo[1], lo[2], voxel_size, bits[1], bits[2],
                       key, val);
    if (passes) PEDP_ROCPRIM(rocprim::radix_sort_pairs<RadixPasses>(d_tmp, tmp_sort, key, key_s, val, val_s, n, 0, key_bits, c->stream));
    else PEDP_ROCPRIM(rocprim::radix_sort_pairs(d_tmp, tmp_sort, key, key_s, val, val_s, n, 0, key_bits, c->stream));
    PEDP_ROCPRIM(rocprim::run_length_encode(d_tmp, tmp_rle, key_s, n, uniq, counts, n_runs, c->stream));
    PEDP_ROCPRIM(rocprim::exclusive_scan(d_tmp, tmp_scan, counts, offsets, 0u, n, rocprim::plus<unsigned>(), c->stream));
    hipLaunchKernelGGL(voxel_average_kernel, dim3(grid), dim3(256), 0, c->stream, d_pts, d_nrm, val_s, counts, offsets, n_runs,
                       d_out, d_outn);
    PEDP_HIP_CHECK(hipGetLastError());
    unsigned *h_runs = (unsigned *)((char *)c->pinned + 8192);
    PEDP_HIP_CHECK(hipMemcpyAsync(h_runs, n_runs, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    *n_out = *h_runs;
    *out_pts = d_out;
    *out_nrm = d_nrm ? d_outn : nullptr;
    return PEDP_OK;
}

// the uniform grid of a cloud on the device: cells, sorted order, ranges, points gathered in that order
struct GridIndex {
    Grid g;
    double *sp;
    int *val_s, *cell_start, *cell_end;  // cell_end = cell_start + 1: one monotone array of n_cells + 1 run starts
    unsigned *cell_s;                    // cell of every sorted position
    void *tmp;  // the scan's scratch, free once the index stands
    size_t tmp_bytes;
};
// carves the index out of cv (the caller has reserved room: grid_index_bytes) and builds it
size_t grid_index_bytes(pedp_ctx_t c, int64_t N, int64_t n_cells, size_t *tmp_scan) {
    PEDP_ROCPRIM(rocprim::exclusive_scan(nullptr, *tmp_scan, (unsigned *)nullptr, (int *)nullptr, 0u, (size_t)n_cells + 1,
                                         rocprim::plus<unsigned>(), c->stream));
    return a256(sizeof(double) * 3 * N) + a256(sizeof(unsigned) * N) * 2 + a256(sizeof(int) * N) * 2 + a256(sizeof(int) * (n_cells + 1)) * 2 +
           a256(*tmp_scan < 256 ? 256 : *tmp_scan);
}
int grid_index_build(pedp_ctx_t c, Carver &cv, const double *d_pts, int64_t N, const Grid &g, int64_t n_cells, size_t tmp_scan,
                     GridIndex &ix) {
    ix.g = g;
    ix.sp = cv.take<double>(3 * (size_t)N);
    unsigned *cell = cv.take<unsigned>(N);
    ix.cell_s = cv.take<unsigned>(N);
    int *slot = cv.take<int>(N);
    ix.val_s = cv.take<int>(N);
    unsigned *count = cv.take<unsigned>(n_cells + 1);
    ix.cell_start = cv.take<int>(n_cells + 1);
    ix.cell_end = ix.cell_start + 1;
    ix.tmp = cv.take<char>(tmp_scan < 256 ? 256 : tmp_scan);
    ix.tmp_bytes = tmp_scan;
    const unsigned grid = (unsigned)((N + 255) / 256);
    PEDP_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(unsigned) * ((size_t)n_cells + 1), c->stream));
    hipLaunchKernelGGL(grid_count_kernel, dim3(grid), dim3(256), 0, c->stream, d_pts, N, g, cell, slot, count);
    PEDP_ROCPRIM(rocprim::exclusive_scan(ix.tmp, tmp_scan, count, ix.cell_start, 0u, (size_t)n_cells + 1, rocprim::plus<unsigned>(), c->stream));
    hipLaunchKernelGGL(grid_place_kernel, dim3(grid), dim3(256), 0, c->stream, d_pts, N, (const unsigned *)cell, (const int *)slot,
                       (const int *)ix.cell_start, ix.sp, ix.val_s, ix.cell_s);
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}

int dbscan_core(pedp_ctx_t c, const double *d_pts, int64_t N, double eps, int min_points, const double lo[3], const double hi[3],
                int32_t **out_labels) {
    Grid g;
    int64_t n_cells = 0;
    // cell = eps / sqrt(3) less a hair: a cell's diameter is below eps and neighbours within eps are never three cells apart
    int rc = make_grid(lo, hi, eps * 0.57735026918962576 * (1.0 - 1e-9), g, n_cells, "pedp_cluster_dbscan");
    if (rc) return rc;
    const bool clique = g.cell * 1.7320508075688772 < eps;
    if (!clique) {  // beyond the cell budget: cells of eps (or its next doubling that fits), neighbours one cell apart
        rc = make_grid(lo, hi, eps * (1.0 + 1e-9), g, n_cells, "pedp_cluster_dbscan");
        if (rc) return rc;
    }
    const unsigned n = (unsigned)N;
    size_t tmp_sort = 0, tmp_scan = 0;
    const size_t ix_bytes = grid_index_bytes(c, N, n_cells, &tmp_sort);
    PEDP_ROCPRIM(rocprim::exclusive_scan(nullptr, tmp_scan, (unsigned *)nullptr, (unsigned *)nullptr, 0u, n,
                                         rocprim::plus<unsigned>(), c->stream));
    const size_t need = ix_bytes + a256(sizeof(unsigned) * N) * 2 + a256(sizeof(int) * N) * 4 + a256(sizeof(int) * n_cells) + a256(tmp_scan) + 4096;
    int st = c->ops.reserve(need);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    GridIndex ix;
    rc = grid_index_build(c, cv, d_pts, N, g, n_cells, tmp_sort, ix);
    if (rc) return rc;
    unsigned *is_rep = cv.take<unsigned>(N), *rank = cv.take<unsigned>(N);
    int *core = cv.take<int>(N), *parent = cv.take<int>(N), *root = cv.take<int>(N), *rep = cv.take<int>(n_cells);
    int32_t *d_labels = cv.take<int32_t>(N);
    void *d_tmp = cv.take<char>(tmp_scan);
    const unsigned grid = (unsigned)((N + 255) / 256);
    const double e2 = eps * eps;
    const unsigned grid_w = (unsigned)((N + DB_WPB - 1) / DB_WPB);
    if (clique) {
        hipLaunchKernelGGL(dbscan_core_kernel<true>, dim3(grid_w), dim3(DB_WPB * 64), 0, c->stream, g, ix.sp, N, ix.cell_start, ix.cell_end, e2,
                           min_points, ix.val_s, core, parent);
        hipLaunchKernelGGL(dbscan_cell_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, c->stream, n_cells, ix.cell_start,
                           ix.cell_end, ix.val_s, core, rep, parent);
        hipLaunchKernelGGL(dbscan_union_kernel<true>, dim3(grid_w), dim3(DB_WPB * 64), 0, c->stream, g, ix.sp, N, ix.cell_start, ix.cell_end,
                           ix.cell_s, e2, ix.val_s, core, rep, parent);
    } else {
        hipLaunchKernelGGL(dbscan_core_kernel<false>, dim3(grid_w), dim3(DB_WPB * 64), 0, c->stream, g, ix.sp, N, ix.cell_start, ix.cell_end, e2,
                           min_points, ix.val_s, core, parent);
        hipLaunchKernelGGL(dbscan_union_kernel<false>, dim3(grid_w), dim3(DB_WPB * 64), 0, c->stream, g, ix.sp, N, ix.cell_start, ix.cell_end,
                           ix.cell_s, e2, ix.val_s, core, rep, parent);
    }
    hipLaunchKernelGGL(dbscan_root_kernel, dim3(grid), dim3(256), 0, c->stream, N, core, parent, root, is_rep);
    PEDP_ROCPRIM(rocprim::exclusive_scan(d_tmp, tmp_scan, is_rep, rank, 0u, n, rocprim::plus<unsigned>(), c->stream));
    if (clique)
        hipLaunchKernelGGL(dbscan_label_kernel<true>, dim3(grid_w), dim3(DB_WPB * 64), 0, c->stream, g, ix.sp, N, ix.cell_start, ix.cell_end, e2, ix.val_s,
                           root, rank, d_labels);
    else
        hipLaunchKernelGGL(dbscan_label_kernel<false>, dim3(grid_w), dim3(DB_WPB * 64), 0, c->stream, g, ix.sp, N, ix.cell_start, ix.cell_end, e2, ix.val_s,
                           root, rank, d_labels);
    PEDP_HIP_CHECK(hipGetLastError());
    *out_labels = d_labels;
    return PEDP_OK;
}

int knn_core(pedp_ctx_t c, const double *d_pts, int64_t N, int k, const double lo[3], const double hi[3], double **out_avg) {
    if (N <= KNN_SMALL_MAX) {  // a wave per query, all distances in LDS
        int st0 = c->ops.reserve(a256(sizeof(double) * N) + 512);
        if (st0) return st0;
        double *d_avg0 = (double *)c->ops.ptr;
        PEDP_HIP_CHECK(hipFuncSetAttribute((const void *)knn_mean_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(double) * KNN_SMALL_MAX)));
        hipLaunchKernelGGL(knn_mean_small_kernel, dim3((unsigned)N), dim3(64), sizeof(double) * (size_t)N, c->stream, d_pts, (int)N, k,
                           d_avg0);
        PEDP_HIP_CHECK(hipGetLastError());
        *out_avg = d_avg0;
        return PEDP_OK;
    }
    // cell ~ the radius that holds k points if the cloud were spread over a sheet of the box's
    // largest extent (scanned surfaces are sheets); any cell size gives the same result
    double ext = 0.0;
    for (int d = 0; d < 3; ++d) ext = std::fmax(ext, hi[d] - lo[d]);
    double cell = ext * std::sqrt((double)k / (3.14159265358979 * (double)N));
    if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
    Grid g;
    int64_t n_cells = 0;
    int rc = make_grid(lo, hi, cell, g, n_cells, "pedp_knn_mean_distance");
    if (rc) return rc;
    size_t tmp_sort = 0;
    const size_t ix_bytes = grid_index_bytes(c, N, n_cells, &tmp_sort);
    int st = c->ops.reserve(ix_bytes + a256(sizeof(double) * N) + 4096);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    GridIndex ix;
    rc = grid_index_build(c, cv, d_pts, N, g, n_cells, tmp_sort, ix);
    if (rc) return rc;
    double *d_avg = cv.take<double>(N);
    // a wave per query; a cloud too dense for the LDS share of some query falls back to a thread per query: that kernel
    // is launched behind the first one and leaves at once unless the overflow flag is up (no trip to the host to ask)
    int *d_over = (int *)ix.tmp;  // the scan is done: its scratch is free
    PEDP_HIP_CHECK(hipMemsetAsync(d_over, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(knn_mean_wave_kernel, dim3((unsigned)((N + KNN_WPB - 1) / KNN_WPB)), dim3(KNN_WPB * 64), 0, c->stream, g, ix.sp, N,
                       ix.cell_start, ix.cell_end, ix.val_s, k, d_avg, d_over);
    const size_t lds = sizeof(double) * (size_t)k * KNN_THREADS;
    PEDP_HIP_CHECK(hipFuncSetAttribute((const void *)knn_mean_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(knn_mean_kernel, dim3((unsigned)((N + KNN_THREADS - 1) / KNN_THREADS)), dim3(KNN_THREADS), lds, c->stream, g, ix.sp,
                       N, ix.cell_start, ix.cell_end, ix.val_s, k, d_avg, (const int *)d_over);
    PEDP_HIP_CHECK(hipGetLastError());
    *out_avg = d_avg;
    return PEDP_OK;
}

int normals_core(pedp_ctx_t c, const double *d_pts, int64_t N, double radius, int max_nn, const double *d_prior, const double lo[3],
                 const double hi[3], double **out_normals) {
    Grid g;
    int64_t n_cells = 0;
    int rc = make_grid(lo, hi, radius * (1.0 + 1e-9), g, n_cells, "pedp_estimate_normals");
    if (rc) return rc;
    size_t tmp_sort = 0;
    const size_t ix_bytes = grid_index_bytes(c, N, n_cells, &tmp_sort);
    int st = c->ops.reserve(ix_bytes + a256(sizeof(double) * 3 * N) + 4096);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    GridIndex ix;
    rc = grid_index_build(c, cv, d_pts, N, g, n_cells, tmp_sort, ix);
    if (rc) return rc;
    double *d_out = cv.take<double>(3 * (size_t)N);
    const size_t lds = (sizeof(double) + sizeof(int)) * (size_t)max_nn * NRM_THREADS;
    PEDP_HIP_CHECK(hipFuncSetAttribute((const void *)normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(normals_kernel, dim3((unsigned)((N + NRM_THREADS - 1) / NRM_THREADS)), dim3(NRM_THREADS), lds, c->stream, g,
                       ix.sp, N, ix.cell_start, ix.cell_end, ix.val_s, radius * radius, max_nn, d_pts, d_prior, d_out);
    PEDP_HIP_CHECK(hipGetLastError());
    *out_normals = d_out;
    return PEDP_OK;
}

// compute_average_normal's voxel means on the device (kernels above), nothing read back here: the means go to the context's
// pinned block as the kernel writes them, their number and the overflow flag (h_res[3], [4]) are on their way to `h_res`
// (pinned) when the call returns; all valid after the stream's next synchronisation.  *enqueued = false: the dense grid does not fit (the caller takes the sort-based grid).
int average_normal_core(pedp_ctx_t c, const double *d_pts, const double *d_nrm, int64_t N, double voxel, const double box_lo[3],
                        const double box_hi[3], double *d_part /* partial bounds + room behind them */, double *h_res, bool *enqueued) {
    *enqueued = false;
    AvgGrid g;
    g.voxel = voxel;
    double cells = 1.0;
    for (int k = 0; k < 3; ++k) {
        const double d = std::floor((box_hi[k] - box_lo[k]) / voxel) + 3.0;  // origin >= box_lo - voxel / 2, points <= box_hi
        if (!(d >= 1.0) || d > 4194304.0) return PEDP_OK;
        g.dim[k] = (int)d;
        cells *= d;
    }
    if (cells > 4194304.0) return PEDP_OK;
    const int64_t n_cells = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    size_t tmp_a = 0, tmp_b = 0;
    PEDP_ROCPRIM(rocprim::exclusive_scan(nullptr, tmp_a, (unsigned *)nullptr, (unsigned *)nullptr, 0u, (size_t)n_cells + 1,
                                         rocprim::plus<unsigned>(), c->stream));
    auto flags_null = rocprim::make_transform_iterator((const unsigned *)nullptr, NonEmpty());
    PEDP_ROCPRIM(rocprim::exclusive_scan(nullptr, tmp_b, flags_null, (unsigned *)nullptr, 0u, (size_t)n_cells, rocprim::plus<unsigned>(), c->stream));
    const size_t tmp = tmp_a > tmp_b ? tmp_a : tmp_b;
    const int64_t max_rows = n_cells < N ? n_cells : N;
    if (sizeof(double) * 3 * (size_t)max_rows > c->avg_host_cap) {
        if (c->avg_host) { PEDP_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->avg_host); }
        c->avg_host = nullptr;
        c->avg_host_cap = 0;
        const size_t want = sizeof(double) * 3 * (size_t)max_rows + sizeof(double) * 3 * (size_t)max_rows / 4 + 4096;
        PEDP_HIP_CHECK(hipHostMalloc(&c->avg_host, want, hipHostMallocDefault));
        c->avg_host_cap = want;
    }
    int st = c->ops.reserve(a256(sizeof(unsigned) * ((size_t)n_cells + 1)) * 3 + a256(sizeof(unsigned) * N) + a256(sizeof(int) * N) * 2 +
                            a256(tmp) + 4096);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    unsigned *count = cv.take<unsigned>((size_t)n_cells + 1), *begin = cv.take<unsigned>((size_t)n_cells + 1),
             *rank = cv.take<unsigned>((size_t)n_cells + 1), *cell = cv.take<unsigned>(N);
    int *slot = cv.take<int>(N), *member = cv.take<int>(N);
    double *means = (double *)c->avg_host;  // (pinned host memory, written by the device)
    void *d_tmp = cv.take<char>(tmp);
    double *d_org = d_part + 6 * BND_BLOCKS + 16, *d_res = d_org + 4;  // behind the partial bounds and the best plane's block
    int *d_over = (int *)(d_res + 8);
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(bounds_kernel, dim3(BND_BLOCKS), dim3(256), 0, c->stream, d_pts, N, d_part);
    hipLaunchKernelGGL(avgn_origin_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)d_part, BND_BLOCKS, voxel, d_org);
    PEDP_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(unsigned) * ((size_t)n_cells + 1), c->stream));
    PEDP_HIP_CHECK(hipMemsetAsync(d_over, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(avgn_count_kernel, dim3(grid), dim3(256), 0, c->stream, d_pts, N, (const double *)d_org, g, cell, slot, count);
    PEDP_ROCPRIM(rocprim::exclusive_scan(d_tmp, tmp_a, count, begin, 0u, (size_t)n_cells + 1, rocprim::plus<unsigned>(), c->stream));
    hipLaunchKernelGGL(avgn_place_kernel, dim3(grid), dim3(256), 0, c->stream, N, (const unsigned *)cell, (const int *)slot,
                       (const unsigned *)begin, member);
    auto flags = rocprim::make_transform_iterator((const unsigned *)count, NonEmpty());
    PEDP_ROCPRIM(rocprim::exclusive_scan(d_tmp, tmp_b, flags, rank, 0u, (size_t)n_cells, rocprim::plus<unsigned>(), c->stream));
    hipLaunchKernelGGL(avgn_mean_kernel, dim3((unsigned)((n_cells + 64 * AVGN_WPB - 1) / (64 * AVGN_WPB))), dim3(64 * AVGN_WPB), 0, c->stream, n_cells, (const unsigned *)begin,
                       (const unsigned *)rank, (const int *)member, d_nrm, means, d_over);
    hipLaunchKernelGGL(avgn_rows_kernel, dim3(1), dim3(1), 0, c->stream, n_cells, (const unsigned *)rank, (const unsigned *)count,
                       (const int *)d_over, d_res);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipMemcpyAsync(h_res, d_res, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream));
    *enqueued = true;
    return PEDP_OK;
}

// plane RANSAC: inlier counts of every iteration -> the best iteration (most inliers, earliest on ties) and its plane
int plane_best_core(pedp_ctx_t c, const double *d_pts, int64_t N, double distance_threshold, int num_iterations, uint64_t seed,
                    int *best_t, double best[4], double *d_keep = nullptr /* 14 doubles on the device: the result stays there */) {
    int st = c->ops.reserve(a256(sizeof(int) * (size_t)num_iterations) + a256(sizeof(double) * (4 * (size_t)num_iterations + 32)) + 512);
    if (st) return st;
    int *d_cnt = (int *)c->ops.ptr;
    double *d_planes = (double *)((char *)c->ops.ptr + a256(sizeof(int) * (size_t)num_iterations));
    hipLaunchKernelGGL(ransac_plane_kernel, dim3((unsigned)((num_iterations + 63) / 64)), dim3(64), 0, c->stream, d_pts, (long long)N,
                       (unsigned long long)seed, num_iterations, d_planes, d_cnt);
    hipLaunchKernelGGL(ransac_count_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)((num_iterations + RS_PLANES - 1) / RS_PLANES)),
                       dim3(256), 0, c->stream, d_pts, (long long)N, distance_threshold, num_iterations, (const double *)d_planes, d_cnt);
    PEDP_HIP_CHECK(hipGetLastError());
    double *d_best = d_keep ? d_keep : d_planes + 4 * (size_t)num_iterations, *h_best = (double *)((char *)c->pinned + 8192);
    hipLaunchKernelGGL(ransac_best_kernel, dim3(1), dim3(256), 0, c->stream, (const int *)d_cnt, num_iterations, d_pts, (long long)N,
                       (unsigned long long)seed, (const double *)d_planes, d_best);
    PEDP_HIP_CHECK(hipGetLastError());
    if (d_keep) return PEDP_OK;
    PEDP_HIP_CHECK(hipMemcpyAsync(h_best, d_best, sizeof(double) * 10, hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int bt = (int)h_best[0];
    *best_t = bt;
    if (bt < 0) return PEDP_OK;
    const double *p3 = h_best + 1;  // the three sampled points
    triangle_plane(p3, p3 + 3, p3 + 6, best);
    return PEDP_OK;
}

// ---- selection in index order: flags -> exclusive scan -> scatter
__global__ void plane_keep_kernel(const double *__restrict__ pts, int64_t N, double a, double b, double cc, double d, double thr,
                                  int inliers, unsigned *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double pl[4] = {a, b, cc, d};
    const bool in = plane_dist(pl, pts + 3 * i) < thr;
    flag[i] = (in == (inliers != 0)) ? 1u : 0u;
}
// the largest cluster (np.unique + argmax: the lowest label among the largest) without a trip to the host: members per
// label, the label with most members, the flags of its points (no cluster at all: *label = -1, nothing is kept)
__global__ void label_count_kernel(const int32_t *__restrict__ labels, int64_t N, int *__restrict__ members) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = i < N ? labels[i] : -1;
    // one add per distinct label of the wave (most points carry the same label: a point's own add each queues
    // thousands of atomics on one word)
    unsigned long long left = __builtin_amdgcn_ballot_w64(l >= 0);
    while (left) {  // wave-uniform
        const int l0 = __builtin_amdgcn_readlane(l, __builtin_ctzll(left));
        const unsigned long long same = __builtin_amdgcn_ballot_w64(l == l0);
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(left)) atomicAdd(&members[l0], __builtin_popcountll(same));
        left &= ~same;
    }
}
__global__ __launch_bounds__(256) void label_largest_kernel(const int *__restrict__ members, int64_t N, int *__restrict__ label) {
    __shared__ int cnt_s[256], lab_s[256];
    int bc = 0, bl = -1;
    for (int64_t l = threadIdx.x; l < N; l += 256)
        if (members[l] > bc) { bc = members[l]; bl = (int)l; }  // ascending l: the lowest of the thread's share
    cnt_s[threadIdx.x] = bc;
    lab_s[threadIdx.x] = bl;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const int oc = cnt_s[threadIdx.x + w], ol = lab_s[threadIdx.x + w];
            if (oc > cnt_s[threadIdx.x] || (oc == cnt_s[threadIdx.x] && ol >= 0 && ol < lab_s[threadIdx.x])) {
                cnt_s[threadIdx.x] = oc;
                lab_s[threadIdx.x] = ol;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *label = lab_s[0];
}
// the same from the device block of ransac_best_kernel (no plane found: everything is kept)
__global__ void plane_keep_best_kernel(const double *__restrict__ pts, int64_t N, const double *__restrict__ best, double thr,
                                       unsigned *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double pl[4] = {best[10], best[11], best[12], best[13]};
    flag[i] = (best[0] < 0.0 || !(plane_dist(pl, pts + 3 * i) < thr)) ? 1u : 0u;
}
// remove_points_below_plane (src/pose_estimation.py:366-378): keep `(a x + b y + c z + d) / sqrt(a^2 + b^2 + c^2) <= 0`.
// The divisor is positive, so the sign is the numerator's: numpy's left-to-right sum of three products and d, one
// rounding each (no contraction in this build).  A flipped plane negates every term exactly: `>= 0` on the same sum.
__global__ void halfspace_keep_kernel(const double *__restrict__ pts, int64_t N, double a, double b, double cc, double d, int flipped,
                                      unsigned *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double s = ((a * pts[3 * i] + b * pts[3 * i + 1]) + cc * pts[3 * i + 2]) + d;
    flag[i] = (flipped ? s >= 0.0 : s <= 0.0) ? 1u : 0u;
}
__global__ void label_keep_kernel(const int32_t *__restrict__ labels, int64_t N, const int *__restrict__ label, unsigned *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = (*label >= 0 && labels[i] == *label) ? 1u : 0u;
}
__global__ void range_keep_kernel(const double *__restrict__ avg, int64_t N, double limit, unsigned *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = (avg[i] > 0.0 && avg[i] < limit) ? 1u : 0u;   // Open3D: 0 < mean distance < mean + ratio * std
}
__global__ void scatter_kept_kernel(const double *__restrict__ pts, const double *__restrict__ nrm, int64_t N,
                                    const unsigned *__restrict__ flag, const unsigned *__restrict__ pos, double *__restrict__ out_pts,
                                    double *__restrict__ out_nrm, unsigned *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (flag[i]) {
        const size_t o = pos[i];
        out_pts[3 * o] = pts[3 * i]; out_pts[3 * o + 1] = pts[3 * i + 1]; out_pts[3 * o + 2] = pts[3 * i + 2];
        if (nrm) { out_nrm[3 * o] = nrm[3 * i]; out_nrm[3 * o + 1] = nrm[3 * i + 1]; out_nrm[3 * o + 2] = nrm[3 * i + 2]; }
    }
    if (i == N - 1) *count = pos[i] + flag[i];
}
// d_flag (N unsigned, in the chain's own buffers) -> kept points (and normals) in index order at out_*; one 4-byte read-back
int select_core(pedp_ctx_t c, const double *d_pts, const double *d_nrm, int64_t N, const unsigned *d_flag, double *out_pts,
                double *out_nrm, int64_t *n_out) {
    *n_out = 0;
    if (N == 0) return PEDP_OK;
    size_t tmp_scan = 0;
    PEDP_ROCPRIM(rocprim::exclusive_scan(nullptr, tmp_scan, (unsigned *)nullptr, (unsigned *)nullptr, 0u, (unsigned)N,
                                         rocprim::plus<unsigned>(), c->stream));
    int st = c->ops.reserve(a256(sizeof(unsigned) * N) + a256(tmp_scan) + 512);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    unsigned *pos = cv.take<unsigned>(N), *count = cv.take<unsigned>(1);
    void *d_tmp = cv.take<char>(tmp_scan);
    PEDP_ROCPRIM(rocprim::exclusive_scan(d_tmp, tmp_scan, d_flag, pos, 0u, (unsigned)N, rocprim::plus<unsigned>(), c->stream));
    hipLaunchKernelGGL(scatter_kept_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_pts, d_nrm, N, d_flag, pos,
                       out_pts, out_nrm, count);
    PEDP_HIP_CHECK(hipGetLastError());
    unsigned *h = (unsigned *)((char *)c->pinned + 8192);
    PEDP_HIP_CHECK(hipMemcpyAsync(h, count, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    *n_out = *h;
    return PEDP_OK;
}

// The plane segment_plane returns -- the best plane refit over its inliers (plane_from_points' blocked sums) -- on the
// device, nothing read back here: res[3..6] the plane, res[7] the inlier count are on their way to `h_res` (pinned, 8
// doubles) when the call returns and valid after the stream's next synchronisation.
int refit_core(pedp_ctx_t c, const double *d_pts, int64_t N, const double *d_best, double thr, double *d_res, double *h_res) {
    const size_t n_blocks = (size_t)((N + 255) / 256);
    int st = c->ops.reserve(a256(sizeof(double) * 6 * n_blocks) + 1024);
    if (st) return st;
    double *part = (double *)c->ops.ptr;
    const unsigned grid = (unsigned)n_blocks;
    hipLaunchKernelGGL(refit_block_kernel<3>, dim3(grid), dim3(256), 0, c->stream, d_pts, N, d_best, thr, (const double *)d_res, part);
    hipLaunchKernelGGL(refit_fold_kernel<3>, dim3(1), dim3(256), 0, c->stream, (const double *)part, (int64_t)n_blocks, d_res);
    hipLaunchKernelGGL(refit_block_kernel<6>, dim3(grid), dim3(256), 0, c->stream, d_pts, N, d_best, thr, (const double *)d_res, part);
    hipLaunchKernelGGL(refit_fold_kernel<6>, dim3(1), dim3(256), 0, c->stream, (const double *)part, (int64_t)n_blocks, d_res);
    PEDP_HIP_CHECK(hipGetLastError());
    PEDP_HIP_CHECK(hipMemcpyAsync(h_res, d_res, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    return PEDP_OK;
}

// upload of a wrapper's host array into the context's input scratch (its own allocation: the cores re-carve `ops`)
int upload_in(pedp_ctx_t c, const double *pts, const double *second, int64_t N, double **d_a, double **d_b) {
    const size_t bytes = a256(sizeof(double) * 3 * (size_t)N);
    int st = c->ops_in.reserve(bytes * (second ? 2 : 1) + 256);
    if (st) return st;
    *d_a = (double *)c->ops_in.ptr;
    { int up_ = pedp_upload(c, *d_a, pts, sizeof(double) * 3 * (size_t)N); if (up_) return up_; }
    if (second) {
        *d_b = (double *)((char *)c->ops_in.ptr + bytes);
        int up_ = pedp_upload(c, *d_b, second, sizeof(double) * 3 * (size_t)N);
        if (up_) return up_;
    } else if (d_b) {
        *d_b = nullptr;
    }
    return PEDP_OK;
}

}  // namespace

extern "C" {

static int voxel_down_sample_impl(pedp_ctx_t c, const double *pts, const double *normals, int64_t N, double voxel_size,
                                  double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out, bool pts_on_device) {
    int rc = check_cloud(c, pts, N, "pedp_voxel_down_sample");
    if (rc) return rc;
    PEDP_REQUIRE(n_out, "pedp_voxel_down_sample: null count");
    *n_out = 0;
    PEDP_REQUIRE(voxel_size > 0.0, "pedp_voxel_down_sample: voxel_size <= 0");  // Open3D raises too
    if (N == 0) return PEDP_OK;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    double *d_pts = const_cast<double *>(pts), *d_nrm = nullptr;
    if (!pts_on_device) { rc = upload_in(c, pts, normals, N, &d_pts, &d_nrm); if (rc) return rc; }
    double *d_out = nullptr, *d_outn = nullptr;
    int64_t m = 0;
    rc = voxel_core(c, d_pts, d_nrm, N, voxel_size, &d_out, &d_outn, &m);
    if (rc) return rc;
    *n_out = m;
    PEDP_REQUIRE(m <= capacity, "pedp_voxel_down_sample: %lld voxels exceed the output capacity %lld", (long long)m,
                 (long long)capacity);
    PEDP_REQUIRE(out_pts && (out_normals || !normals), "pedp_voxel_down_sample: null output arrays");
    { int dn_ = pedp_download(c, out_pts, d_out, sizeof(double) * 3 * (size_t)m); if (dn_) return dn_; }
    if (normals) { int dn_ = pedp_download(c, out_normals, d_outn, sizeof(double) * 3 * (size_t)m); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_voxel_down_sample(pedp_ctx_t c, const double *pts, const double *normals, int64_t N, double voxel_size,
                           double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out) {
    return voxel_down_sample_impl(c, pts, normals, N, voxel_size, out_pts, out_normals, capacity, n_out, false);
}

int pedp_voxel_down_sample_device_in(pedp_ctx_t c, const double *d_pts, int64_t N, double voxel_size, double *out_pts,
                                     int64_t capacity, int64_t *n_out) {
    return voxel_down_sample_impl(c, d_pts, nullptr, N, voxel_size, out_pts, nullptr, capacity, n_out, true);
}

int pedp_cluster_dbscan(pedp_ctx_t c, const double *pts, int64_t N, double eps, int min_points, int32_t *labels) {
    int rc = check_cloud(c, pts, N, "pedp_cluster_dbscan");
    if (rc) return rc;
    PEDP_REQUIRE(eps > 0.0 && std::isfinite(eps), "pedp_cluster_dbscan: eps must be positive and finite");
    if (N == 0) return PEDP_OK;
    PEDP_REQUIRE(labels, "pedp_cluster_dbscan: null labels");
    double lo[3], hi[3];
    bounds(pts, N, lo, hi);
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    double *d_pts = nullptr;
    rc = upload_in(c, pts, nullptr, N, &d_pts, nullptr);
    if (rc) return rc;
    int32_t *d_labels = nullptr;
    rc = dbscan_core(c, d_pts, N, eps, min_points, lo, hi, &d_labels);
    if (rc) return rc;
    { int dn_ = pedp_download(c, labels, d_labels, sizeof(int32_t) * (size_t)N); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_knn_mean_distance(pedp_ctx_t c, const double *pts, int64_t N, int k, double *avg) {
    int rc = check_cloud(c, pts, N, "pedp_knn_mean_distance");
    if (rc) return rc;
    PEDP_REQUIRE(k >= 1 && k <= 300, "pedp_knn_mean_distance: k must be in 1..300");  // k x 64 doubles of LDS per workgroup
    if (N == 0) return PEDP_OK;
    PEDP_REQUIRE(avg, "pedp_knn_mean_distance: null output");
    double lo[3], hi[3];
    bounds(pts, N, lo, hi);
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    double *d_pts = nullptr;
    rc = upload_in(c, pts, nullptr, N, &d_pts, nullptr);
    if (rc) return rc;
    double *d_avg = nullptr;
    rc = knn_core(c, d_pts, N, k, lo, hi, &d_avg);
    if (rc) return rc;
    { int dn_ = pedp_download(c, avg, d_avg, sizeof(double) * (size_t)N); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_estimate_normals(pedp_ctx_t c, const double *pts, int64_t N, double radius, int max_nn, const double *prior,
                          double *normals) {
    int rc = check_cloud(c, pts, N, "pedp_estimate_normals");
    if (rc) return rc;
    PEDP_REQUIRE(radius > 0.0 && std::isfinite(radius), "pedp_estimate_normals: radius must be positive and finite");
    PEDP_REQUIRE(max_nn >= 1 && max_nn <= 128, "pedp_estimate_normals: max_nn must be in 1..128");
    if (N == 0) return PEDP_OK;
    PEDP_REQUIRE(normals, "pedp_estimate_normals: null output");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    double *d_pts = nullptr, *d_prior = nullptr;
    rc = upload_in(c, pts, prior, N, &d_pts, &d_prior);
    if (rc) return rc;
    double lo[3], hi[3];
    if (N < BND_DEVICE_MIN) {
        bounds(pts, N, lo, hi);
    } else {  // the box of a large cloud from its device copy: no host pass over the caller's array
        double *d_part = nullptr;
        rc = small_block(c, &d_part);
        if (rc) return rc;
        rc = bounds_device(c, d_pts, N, d_part, lo, hi);
        if (rc) return rc;
    }
    double *d_out = nullptr;
    rc = normals_core(c, d_pts, N, radius, max_nn, d_prior, lo, hi, &d_out);
    if (rc) return rc;
    { int dn_ = pedp_download(c, normals, d_out, sizeof(double) * 3 * (size_t)N); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_fpfh(pedp_ctx_t c, const double *pts, const double *normals, int64_t N, double radius, int max_nn, double *out) {
    int rc = check_cloud(c, pts, N, "pedp_fpfh");
    if (rc) return rc;
    PEDP_REQUIRE(radius > 0.0 && std::isfinite(radius), "pedp_fpfh: radius must be positive and finite");
    PEDP_REQUIRE(max_nn >= 1 && max_nn <= 128, "pedp_fpfh: max_nn must be in 1..128");
    if (N == 0) return PEDP_OK;
    PEDP_REQUIRE(normals && out, "pedp_fpfh: null normals / output (FPFH needs the cloud's normals)");
    PEDP_REQUIRE(N * (int64_t)max_nn < (int64_t)1 << 31, "pedp_fpfh: neighbour table too large");
    double lo[3], hi[3], *d_in = nullptr;
    rc = bounds_of(c, pts, N, lo, hi, &d_in);
    if (rc) return rc;
    Grid g;
    int64_t n_cells = 0;
    rc = make_grid(lo, hi, radius * (1.0 + 1e-9), g, n_cells, "pedp_fpfh");
    if (rc) return rc;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const unsigned n = (unsigned)N;
    size_t tmp_sort = 0;
    PEDP_ROCPRIM(rocprim::radix_sort_pairs(nullptr, tmp_sort, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr,
                                           (int *)nullptr, n, 0, 32, c->stream));
    const size_t need = a256(sizeof(double) * 3 * N) * 3 + a256(sizeof(unsigned) * N) * 2 + a256(sizeof(int) * N) * 3 +
                        a256(sizeof(int) * n_cells) * 2 + a256(tmp_sort) + a256(sizeof(int) * (size_t)N * max_nn) +
                        a256(sizeof(double) * (size_t)N * max_nn) + a256(sizeof(double) * 33 * N) * 2 + 8192;
    int st = c->ops.reserve(need);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    double *d_pts = cv.take<double>(3 * (size_t)N), *sp = cv.take<double>(3 * (size_t)N), *d_nrm = cv.take<double>(3 * (size_t)N);
    unsigned *cell_id = cv.take<unsigned>(N), *cell_s = cv.take<unsigned>(N);
    int *val = cv.take<int>(N), *val_s = cv.take<int>(N), *nbr_n = cv.take<int>(N);
    int *cell_start = cv.take<int>(2 * (size_t)n_cells), *cell_end = cell_start + n_cells;
    void *d_tmp = cv.take<char>(tmp_sort);
    int *nbr_j = cv.take<int>((size_t)N * max_nn);
    double *nbr_d = cv.take<double>((size_t)N * max_nn);
    double *spfh = cv.take<double>(33 * (size_t)N), *d_out = cv.take<double>(33 * (size_t)N);
    if (d_in) d_pts = d_in;  // already up (bounds_of)
    else { int up_ = pedp_upload(c, d_pts, pts, sizeof(double) * 3 * (size_t)N); if (up_) return up_; }
    { int up_ = pedp_upload(c, d_nrm, normals, sizeof(double) * 3 * (size_t)N); if (up_) return up_; }
    PEDP_HIP_CHECK(hipMemsetAsync(cell_start, 0, sizeof(int) * 2 * (size_t)n_cells, c->stream));
    const unsigned grid = (unsigned)((N + 255) / 256), grid64 = (unsigned)((N + NRM_THREADS - 1) / NRM_THREADS);
    hipLaunchKernelGGL(grid_cell_kernel, dim3(grid), dim3(256), 0, c->stream, d_pts, N, g, cell_id, val);
    PEDP_ROCPRIM(rocprim::radix_sort_pairs(d_tmp, tmp_sort, cell_id, cell_s, val, val_s, n, 0, 32, c->stream));
    hipLaunchKernelGGL(grid_ranges_kernel, dim3(grid), dim3(256), 0, c->stream, cell_s, val_s, d_pts, N, cell_start, cell_end, sp);
    // a wave per query, 1,024 candidates each; a denser neighbourhood somewhere sends the cloud to the 4,096-candidate
    // instantiation and, past that, to the thread-per-query kernel
    int *d_over = cv.take<int>(64), h_over = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        PEDP_HIP_CHECK(hipMemsetAsync(d_over, 0, sizeof(int), c->stream));
        if (attempt == 0)
            hipLaunchKernelGGL((hybrid_list_wave_kernel<1024, 4>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream, g, sp, N,
                               cell_start, cell_end, val_s, radius * radius, max_nn, nbr_j, nbr_d, nbr_n, d_over);
        else if (attempt == 1)
            hipLaunchKernelGGL((hybrid_list_wave_kernel<4096, 2>), dim3((unsigned)((N + 1) / 2)), dim3(128), 0, c->stream, g, sp, N,
                               cell_start, cell_end, val_s, radius * radius, max_nn, nbr_j, nbr_d, nbr_n, d_over);
        else {
            const size_t lds = (sizeof(double) + sizeof(int)) * (size_t)max_nn * NRM_THREADS;
            PEDP_HIP_CHECK(hipFuncSetAttribute((const void *)hybrid_list_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(hybrid_list_kernel, dim3(grid64), dim3(NRM_THREADS), lds, c->stream, g, sp, N, cell_start, cell_end,
                               val_s, radius * radius, max_nn, nbr_j, nbr_d, nbr_n);
        }
        PEDP_HIP_CHECK(hipGetLastError());
        PEDP_HIP_CHECK(hipMemcpyAsync(&h_over, d_over, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (!h_over) break;
    }
    hipLaunchKernelGGL(spfh_kernel, dim3(grid64), dim3(NRM_THREADS), 0, c->stream, d_pts, d_nrm, N, nbr_j, nbr_n, spfh);
    hipLaunchKernelGGL(fpfh_kernel, dim3(grid64), dim3(NRM_THREADS), 0, c->stream, N, nbr_j, nbr_d, nbr_n, spfh, d_out);
    PEDP_HIP_CHECK(hipGetLastError());
    { int dn_ = pedp_download(c, out, d_out, sizeof(double) * 33 * (size_t)N); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_feature_match(pedp_ctx_t c, const double *fs, int64_t Ns, const double *ft, int64_t Nt, int32_t *idx) {
    PEDP_REQUIRE(c, "pedp_feature_match: null context");
    PEDP_REQUIRE(Ns >= 0 && Nt >= 0 && Ns < (int64_t)1 << 31 && Nt < (int64_t)1 << 31, "pedp_feature_match: sizes out of range");
    if (Ns == 0) return PEDP_OK;
    PEDP_REQUIRE(fs && idx && (ft || Nt == 0), "pedp_feature_match: null arrays");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const int n_chunks = (int)((Nt + FM_CHUNK - 1) / FM_CHUNK > 0 ? (Nt + FM_CHUNK - 1) / FM_CHUNK : 1);
    PEDP_REQUIRE(n_chunks <= 65535, "pedp_feature_match: more than 67 M target features");
    int st = c->ops.reserve(a256(sizeof(double) * 33 * Ns) + a256(sizeof(double) * 33 * (Nt > 0 ? Nt : 1)) + a256(sizeof(int32_t) * Ns) +
                            a256(sizeof(double) * (size_t)n_chunks * Ns) + a256(sizeof(int32_t) * (size_t)n_chunks * Ns) + 1024);
    if (st) return st;
    Carver cv{(char *)c->ops.ptr};
    double *d_fs = cv.take<double>(33 * (size_t)Ns), *d_ft = cv.take<double>(33 * (size_t)(Nt > 0 ? Nt : 1));
    int32_t *d_idx = cv.take<int32_t>(Ns);
    double *part_d = cv.take<double>((size_t)n_chunks * Ns);
    int32_t *part_j = cv.take<int32_t>((size_t)n_chunks * Ns);
    { int up_ = pedp_upload(c, d_fs, fs, sizeof(double) * 33 * (size_t)Ns); if (up_) return up_; }
    if (Nt > 0) { int up_ = pedp_upload(c, d_ft, ft, sizeof(double) * 33 * (size_t)Nt); if (up_) return up_; }
    hipLaunchKernelGGL(feature_match_kernel, dim3((unsigned)((Ns + 63) / 64), (unsigned)n_chunks), dim3(64), 0, c->stream, d_fs, Ns,
                       d_ft, Nt, part_d, part_j);
    hipLaunchKernelGGL(feature_match_fold_kernel, dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, c->stream, Ns, n_chunks,
                       (const double *)part_d, (const int32_t *)part_j, d_idx);
    PEDP_HIP_CHECK(hipGetLastError());
    { int dn_ = pedp_download(c, idx, d_idx, sizeof(int32_t) * (size_t)Ns); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

int pedp_segment_plane(pedp_ctx_t c, const double *pts, int64_t N, double distance_threshold, int num_iterations,
                       uint64_t seed, double plane[4], int32_t *inliers, int64_t *n_inliers) {
    int rc = check_cloud(c, pts, N, "pedp_segment_plane");
    if (rc) return rc;
    PEDP_REQUIRE(plane && n_inliers, "pedp_segment_plane: null outputs");
    PEDP_REQUIRE(num_iterations >= 0 && num_iterations <= 10000000, "pedp_segment_plane: num_iterations out of range");
    plane[0] = plane[1] = plane[2] = plane[3] = 0.0;
    *n_inliers = 0;
    if (N < 3 || num_iterations == 0) return PEDP_OK;
    PEDP_REQUIRE(inliers, "pedp_segment_plane: null inlier array");
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    double *d_pts = nullptr;
    rc = upload_in(c, pts, nullptr, N, &d_pts, nullptr);
    if (rc) return rc;
    int best_t = -1;
    double best[4];
    rc = plane_best_core(c, d_pts, N, distance_threshold, num_iterations, seed, &best_t, best);
    if (rc) return rc;
    if (best_t < 0) return PEDP_OK;
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i)
        if (plane_dist(best, pts + 3 * i) < distance_threshold) inliers[n++] = (int32_t)i;
    plane_from_points(pts, inliers, n, plane);
    *n_inliers = n;
    return PEDP_OK;
}

// ------------------------------------------------------------------ preprocess_source, device resident
// The reference's scene preprocessing (src/pose_estimation.py:186-268, the branch without param['box'] /
// param['mesh'] / background): voxel grid -> table plane by RANSAC -> [normals of the down-sampled cloud] ->
// plane removed -> DBSCAN, largest cluster -> statistical outlier filter -> [normals, oriented like the first
// ones].  The same kernels as the single operations above, chained on the device: what crosses PCIe is the
// input cloud (if it is not on the device already), a few counters and small arrays the host decides on
// (RANSAC inlier counts, three sampled points, cluster labels of the ~7 k points left, their mean neighbour
// distances) and the processed cloud.  Results are bit for bit those of the step-by-step calls.
int pedp_preprocess_source(pedp_ctx_t c, const double *pts, int64_t N, int pts_on_device, const pedp_preprocess_params *prm,
                           double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out, int64_t stage_counts[4],
                           int *status) {
    return pedp_preprocess_source_ex(c, pts, N, pts_on_device, prm, out_pts, out_normals, capacity, n_out, stage_counts, status, nullptr);
}

int pedp_preprocess_source_ex(pedp_ctx_t c, const double *pts, int64_t N, int pts_on_device, const pedp_preprocess_params *prm,
                              double *out_pts, double *out_normals, int64_t capacity, int64_t *n_out, int64_t stage_counts[4],
                              int *status, double report[PEDP_PREPROCESS_REPORT_DOUBLES]) {
    int rc = check_cloud(c, pts, N, "pedp_preprocess_source");
    if (rc) return rc;
    PEDP_REQUIRE(prm && n_out && status && out_pts, "pedp_preprocess_source: null argument");
    PEDP_REQUIRE(prm->voxel_size > 0.0, "pedp_preprocess_source: voxel_size <= 0");
    PEDP_REQUIRE(prm->plane_iterations >= 1 && prm->plane_iterations <= 10000000, "pedp_preprocess_source: plane_iterations out of range");
    PEDP_REQUIRE(prm->cluster_eps > 0.0 && std::isfinite(prm->cluster_eps), "pedp_preprocess_source: cluster_eps must be positive");
    PEDP_REQUIRE(prm->outlier_neighbors >= 1 && prm->outlier_neighbors <= 300, "pedp_preprocess_source: outlier_neighbors must be in 1..300");
    PEDP_REQUIRE(!prm->first_frame || (prm->normal_radius > 0.0 && prm->normal_max_nn >= 1 && prm->normal_max_nn <= 128 && out_normals),
                 "pedp_preprocess_source: normals (first frame) need a radius, max_nn in 1..128 and an output array");
    *n_out = 0;
    *status = PEDP_PREPROCESS_OK;
    if (stage_counts) stage_counts[0] = stage_counts[1] = stage_counts[2] = stage_counts[3] = 0;
    if (N < 3) { *status = PEDP_PREPROCESS_DEGENERATE; return PEDP_OK; }
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    const bool nrm = prm->first_frame != 0;
    const bool box = (prm->flags & PEDP_PREPROCESS_BOX) != 0;
    const bool seen = box || report != nullptr;  // someone reads the refit plane and the average normal (box cut / INFO lines)
    const bool carry = nrm && !box;              // remove_points_below_plane returns points only (:375-377)
    const double avg_voxel = prm->average_normal_voxel > 0.0 ? prm->average_normal_voxel : 10.0;
    if (report) for (int k = 0; k < PEDP_PREPROCESS_REPORT_DOUBLES; ++k) report[k] = 0.0;
    double *d_in = const_cast<double *>(pts);
    if (!pts_on_device) { rc = upload_in(c, pts, nullptr, N, &d_in, nullptr); if (rc) return rc; }
    // ---- voxel grid
    double *v_pts = nullptr, *v_nrm = nullptr, *d_part = nullptr;
    int64_t m1 = 0;
    double box_lo[3], box_hi[3];  // of the input cloud: every later stage's points lie inside, so it can shape their grids
    rc = voxel_core(c, d_in, nullptr, N, prm->voxel_size, &v_pts, &v_nrm, &m1, box_lo, box_hi);
    if (rc) return rc;
    if (stage_counts) stage_counts[0] = m1;
    if (m1 < 3) { *status = PEDP_PREPROCESS_DEGENERATE; return PEDP_OK; }
    // the clouds between the stages: three (points, normals) pairs and a flag array, sized by the down-sampled cloud
    const size_t one = a256(sizeof(double) * 3 * (size_t)m1);
    rc = c->chain.reserve(6 * one + a256(sizeof(unsigned) * (size_t)m1) + a256(sizeof(int) * ((size_t)m1 + 1)) + 256);
    if (rc) return rc;
    rc = small_block(c, &d_part);
    if (rc) return rc;
    char *cb = (char *)c->chain.ptr;
    double *A_pts = (double *)cb, *A_nrm = (double *)(cb + one), *B_pts = (double *)(cb + 2 * one), *B_nrm = (double *)(cb + 3 * one),
           *C_pts = (double *)(cb + 4 * one), *C_nrm = (double *)(cb + 5 * one);
    unsigned *flag = (unsigned *)(cb + 6 * one);
    int *d_members = (int *)(cb + 6 * one + a256(sizeof(unsigned) * (size_t)m1));  // members per label, then the largest cluster's label
    PEDP_HIP_CHECK(hipMemcpyAsync(A_pts, v_pts, sizeof(double) * 3 * (size_t)m1, hipMemcpyDeviceToDevice, c->stream));
    // ---- table plane: the best of the sampled planes
    int best_t = -1;
    double best[4] = {0, 0, 0, 0}, *d_best = d_part + 6 * BND_BLOCKS;  // behind the partial bounds, untouched by the stages between
    rc = plane_best_core(c, A_pts, m1, prm->plane_distance, prm->plane_iterations, prm->seed, &best_t, best, d_best);
    if (rc) return rc;
    // ---- someone reads the plane segment_plane returns (the best plane refit over its inliers): refit on the device,
    // behind the plane kernels; the result is picked up after the stream's next wait
    double refit[4] = {0, 0, 0, 0};
    double *h_refit = (double *)((char *)c->pinned + 66560), *d_refit = d_part + 6 * BND_BLOCKS + 48;  // (behind the average normal's block)
    bool plane_found = false, refit_pending = false;
    if (seen) {
        rc = refit_core(c, A_pts, m1, d_best, prm->plane_distance, d_refit, h_refit);
        if (rc) return rc;
        refit_pending = true;
    }
    auto refit_collect = [&]() {  // after a wait on the stream
        if (!refit_pending) return;
        refit_pending = false;
        plane_found = h_refit[7] > 0.0;
        for (int k = 0; k < 4; ++k) refit[k] = h_refit[3 + k];
        if (report) { for (int k = 0; k < 4; ++k) report[k] = refit[k]; report[8] = h_refit[7]; }
    };
    double lo[3], hi[3];
    // ---- normals of the down-sampled cloud (they orient the final ones)
    if (nrm) {
        double *n_out_d = nullptr;
        rc = normals_core(c, A_pts, m1, prm->normal_radius, prm->normal_max_nn, nullptr, box_lo, box_hi, &n_out_d);
        if (rc) return rc;
        PEDP_HIP_CHECK(hipMemcpyAsync(A_nrm, n_out_d, sizeof(double) * 3 * (size_t)m1, hipMemcpyDeviceToDevice, c->stream));
    }
    // ---- compute_average_normal (:314-321): the normals' voxel averages on a 10-unit grid, their mean in row order
    // (numpy's reduction over axis 0 adds row by row).  The caller normalises it -- numpy's norm is the caller's.
    double avg[3] = {1.0, 1.0, 1.0};  // tracking frames: the reference's np.array([1, 1, 1]) (:217)
    double *h_avg = (double *)((char *)c->pinned + 65536);  // sum xyz, voxels, overflow: valid after the stream's next wait
    bool avg_pending = false;
    auto average_by_sort = [&]() -> int {  // the sort-based grid (a grid too large for dense cells, or a crowded voxel)
        double *g_pts = nullptr, *g_nrm = nullptr;
        int64_t m10 = 0;
        int st_ = voxel_core(c, A_pts, A_nrm, m1, avg_voxel, &g_pts, &g_nrm, &m10);
        if (st_) return st_;
        std::vector<double> vn(3 * (size_t)m10);
        { int dn_ = pedp_download(c, vn.data(), g_nrm, sizeof(double) * 3 * (size_t)m10); if (dn_) return dn_; }
        double sum[3] = {0.0, 0.0, 0.0};
        for (int64_t i = 0; i < m10; ++i)
            for (int k = 0; k < 3; ++k) sum[k] += vn[3 * (size_t)i + k];
        for (int k = 0; k < 3; ++k) avg[k] = sum[k] / (double)m10;
        if (report) report[9] = (double)m10;
        return PEDP_OK;
    };
    auto average_collect = [&]() -> int {  // after a wait on the stream
        if (!avg_pending) return PEDP_OK;
        avg_pending = false;
        if (h_avg[4] != 0.0) return average_by_sort();   // (A_pts / A_nrm are still the down-sampled cloud here)
        const int64_t m10 = (int64_t)h_avg[3];
        const double *vn = (const double *)c->avg_host;  // the voxel means, written by avgn_mean_kernel
        double sum[3] = {0.0, 0.0, 0.0};
        for (int64_t i = 0; i < m10; ++i)
            for (int k = 0; k < 3; ++k) sum[k] += vn[3 * (size_t)i + k];
        for (int k = 0; k < 3; ++k) avg[k] = sum[k] / (double)m10;
        if (report) report[9] = h_avg[3];
        return PEDP_OK;
    };
    if (seen && nrm) {
        rc = average_normal_core(c, A_pts, A_nrm, m1, avg_voxel, box_lo, box_hi, d_part, h_avg, &avg_pending);
        if (rc) return rc;
        if (!avg_pending) { rc = average_by_sort(); if (rc) return rc; }
    }
    int64_t m2 = m1;
    if (box) {
        PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
        rc = average_collect();
        if (rc) return rc;
        refit_collect();
        // ---- flip_plane_normal_if_needed (:342-359) + remove_points_below_plane (:366-378).  Only the SIGN of
        // dot(plane normal, average normal) is used; a dot product within rounding of zero is left to the step-by-step
        // path (status AMBIGUOUS), whose numpy arithmetic is then the reference's to the letter.
        if (!plane_found) { *status = PEDP_PREPROCESS_NO_CLUSTER; return PEDP_OK; }
        const double pn = std::sqrt((refit[0] * refit[0] + refit[1] * refit[1]) + refit[2] * refit[2]);
        const double an = nrm ? std::sqrt((avg[0] * avg[0] + avg[1] * avg[1]) + avg[2] * avg[2]) : 1.0;
        const double dot = ((refit[0] * avg[0] + refit[1] * avg[1]) + refit[2] * avg[2]) / (pn * an);
        if (!(std::fabs(dot) > 1e-9)) { *status = PEDP_PREPROCESS_AMBIGUOUS; return PEDP_OK; }
        const int flipped = dot < 0.0 ? 1 : 0;
        if (report) report[7] = (double)flipped;
        hipLaunchKernelGGL(halfspace_keep_kernel, dim3((unsigned)((m1 + 255) / 256)), dim3(256), 0, c->stream, A_pts, m1, refit[0], refit[1],
                           refit[2], refit[3], flipped, flag);
    } else {
        // ---- the plane's inliers removed (no plane found: nothing removed, like select_by_index of an empty list)
        hipLaunchKernelGGL(plane_keep_best_kernel, dim3((unsigned)((m1 + 255) / 256)), dim3(256), 0, c->stream, A_pts, m1,
                           (const double *)d_best, prm->plane_distance, flag);
    }
    rc = select_core(c, A_pts, carry ? A_nrm : nullptr, m1, flag, B_pts, B_nrm, &m2);
    if (rc) return rc;
    rc = average_collect();  // (select_core has waited for the stream)
    if (rc) return rc;
    refit_collect();
    if (report) for (int k = 0; k < 3; ++k) report[4 + k] = avg[k];
    if (stage_counts) stage_counts[1] = m2;
    if (m2 == 0) { *status = PEDP_PREPROCESS_NO_CLUSTER; return PEDP_OK; }
    // ---- DBSCAN, largest cluster (np.unique + argmax: the lowest label among the largest)
    int32_t *d_labels = nullptr;  // (the grid over the input's box: no trip to the host for this cloud's own)
    rc = dbscan_core(c, B_pts, m2, prm->cluster_eps, prm->cluster_min_points, box_lo, box_hi, &d_labels);
    if (rc) return rc;
    PEDP_HIP_CHECK(hipMemsetAsync(d_members, 0, sizeof(int) * (size_t)(m2 + 1), c->stream));
    hipLaunchKernelGGL(label_count_kernel, dim3((unsigned)((m2 + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *)d_labels, m2,
                       d_members);
    hipLaunchKernelGGL(label_largest_kernel, dim3(1), dim3(256), 0, c->stream, (const int *)d_members, m2, d_members + m2);
    hipLaunchKernelGGL(label_keep_kernel, dim3((unsigned)((m2 + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *)d_labels, m2,
                       (const int *)(d_members + m2), flag);
    int64_t m3 = 0;
    rc = select_core(c, B_pts, carry ? B_nrm : nullptr, m2, flag, C_pts, C_nrm, &m3);
    if (rc) return rc;
    if (stage_counts) stage_counts[2] = m3;
    if (m3 == 0) { *status = PEDP_PREPROCESS_NO_CLUSTER; return PEDP_OK; }  // every label was noise
    // ---- statistical outlier filter: mean distance to the k nearest on the device, Open3D's global step on the host
    // in index order (cloud_ops.statistical_outlier_indices: sequential sums), the cut again on the device
    if (m3 > KNN_SMALL_MAX) {
        rc = bounds_device(c, C_pts, m3, d_part, lo, hi);
        if (rc) return rc;
    }
    double *d_avg = nullptr;
    rc = knn_core(c, C_pts, m3, prm->outlier_neighbors, lo, hi, &d_avg);
    if (rc) return rc;
    // (read where the copy lands, in the context's page-locked buffer: a std::vector of its own cost an allocation, its
    // page faults and a staged copy -- 0.13 ms of host time in a frame's trace for 12 k doubles)
    const double *dist = nullptr;
    { const void *v_ = nullptr; int dn_ = pedp_download_view(c, d_avg, sizeof(double) * (size_t)m3, &v_); if (dn_) return dn_; dist = (const double *)v_; }
    int64_t valid = 0;
    double sum = 0.0;     // (index order, like the two separate loops before: the count is exact in any order)
    for (int64_t i = 0; i < m3; ++i) {
        valid += dist[i] >= 0.0 ? 1 : 0;
        sum += dist[i] > 0.0 ? dist[i] : 0.0;
    }
    int64_t m4 = 0;
    if (valid > 0) {
        const double mean = sum / (double)valid;
        double sq = 0.0;
        for (int64_t i = 0; i < m3; ++i) sq += dist[i] > 0.0 ? (dist[i] - mean) * (dist[i] - mean) : 0.0;
        const double sd = valid > 1 ? std::sqrt(sq / (double)(valid - 1)) : std::nan("");
        const double limit = mean + prm->outlier_std_ratio * sd;
        hipLaunchKernelGGL(range_keep_kernel, dim3((unsigned)((m3 + 255) / 256)), dim3(256), 0, c->stream, (const double *)d_avg, m3, limit,
                           flag);
        rc = select_core(c, C_pts, carry ? C_nrm : nullptr, m3, flag, A_pts, A_nrm, &m4);
        if (rc) return rc;
    }
    if (stage_counts) stage_counts[3] = m4;
    *n_out = m4;
    PEDP_REQUIRE(m4 <= capacity, "pedp_preprocess_source: %lld points exceed the output capacity %lld", (long long)m4, (long long)capacity);
    if (m4 == 0) return PEDP_OK;
    // ---- normals of the processed cloud, oriented like the ones it carries
    if (nrm) {
        rc = bounds_device(c, A_pts, m4, d_part, lo, hi);
        if (rc) return rc;
        double *n_fin = nullptr;
        rc = normals_core(c, A_pts, m4, prm->normal_radius, prm->normal_max_nn, carry ? A_nrm : nullptr, lo, hi, &n_fin);
        if (rc) return rc;
        { int dn_ = pedp_download(c, out_normals, n_fin, sizeof(double) * 3 * (size_t)m4); if (dn_) return dn_; }
    }
    { int dn_ = pedp_download(c, out_pts, A_pts, sizeof(double) * 3 * (size_t)m4); if (dn_) return dn_; }
    PEDP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PEDP_OK;
}

}  // extern "C"


// Stable sort of point indices by a small integer key (pedp_icp.hip: spatial order of a cloud by
// the Hilbert index of its grid cell).  Stable = equal keys keep ascending point index, so the
// order is a function of the data alone -- the ICP sums that follow it are run-to-run bit-stable.
// keys: N 64-bit keys below 2^bits on the device (overwritten); d_perm: N int32 out.
namespace {
__global__ void iota_kernel(int *__restrict__ v, int64_t N) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) v[i] = (int)i;
}
}  // namespace
// Two calls: _begin reserves the context's scratch for N 64-bit keys plus the sort's buffers and returns the key
// array for the caller's key kernel; _run sorts.  Everything is stream-ordered in the context's pooled scratch:
// no allocation once sizes have settled, no synchronisation.
namespace {
struct Sort64Layout {
    unsigned long long *keys, *keys_s;
    int *val;
    void *tmp;
    size_t tmp_bytes;
};
// The keys are Hilbert indices, most significant level first: their top 3 b bits are the index at b bits per axis.
// Only as many levels are sorted on as tell the points apart -- b = ceil(log2(N) / 3) + 1 bits per axis, a few
// points per cell; inside a cell the stable sort keeps the point order -- because the radix passes' number follows
// the bits (a 368 k-point frame: 24 bits, three passes, instead of six for all 48).  Large inputs take rocPRIM's radix
// passes, small ones its merge sort (fewer, shorter launches below ~10^5 items).
struct SortPlan { int begin_bit, end_bit; bool passes; };
SortPlan sort_plan(int64_t N, int bits) {
    int lg = 0;
    while (lg < 62 && ((int64_t)1 << lg) < N) ++lg;
    int per_axis = (lg + 2) / 3 + 1;
    per_axis = per_axis < 5 ? 5 : per_axis;
    const int used = 3 * per_axis < bits ? 3 * per_axis : bits;
    return {bits - used, bits, N >= 131072};
}
int sort64_layout(pedp_ctx_t c, int64_t N, int bits, Sort64Layout &L) {
    size_t tmp_sort = 0;
    const SortPlan sp = sort_plan(N, bits);
    if (sp.passes)
        PEDP_ROCPRIM(rocprim::radix_sort_pairs<RadixPasses>(nullptr, tmp_sort, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                            (int *)nullptr, (int *)nullptr, (unsigned)N, sp.begin_bit, sp.end_bit, c->stream));
    else
        PEDP_ROCPRIM(rocprim::radix_sort_pairs(nullptr, tmp_sort, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr,
                                               (int *)nullptr, (unsigned)N, sp.begin_bit, sp.end_bit, c->stream));
    const size_t need = 2 * a256(sizeof(unsigned long long) * N) + a256(sizeof(int) * N) + a256(tmp_sort) + 1024;
    int st = c->sort_ws.reserve(need);
    if (st) return st;
    Carver cv{(char *)c->sort_ws.ptr};
    L.keys = cv.take<unsigned long long>(N);
    L.keys_s = cv.take<unsigned long long>(N);
    L.val = cv.take<int>(N);
    L.tmp = cv.take<char>(tmp_sort);
    L.tmp_bytes = tmp_sort;
    return PEDP_OK;
}
}  // namespace
int pedp_sort_keys64_begin(pedp_ctx_t c, int64_t N, int bits, unsigned long long **d_keys) {
    Sort64Layout L;
    int st = sort64_layout(c, N, bits, L);
    if (st) return st;
    *d_keys = L.keys;
    return PEDP_OK;
}
int pedp_sort_keys64_run(pedp_ctx_t c, int64_t N, int bits, int32_t *d_perm) {
    if (N <= 0) return PEDP_OK;
    Sort64Layout L;
    int st = sort64_layout(c, N, bits, L);  // same sizes as _begin: the scratch does not move
    if (st) return st;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, L.val, N);
    const SortPlan sp = sort_plan(N, bits);
    if (sp.passes)
        PEDP_ROCPRIM(rocprim::radix_sort_pairs<RadixPasses>(L.tmp, L.tmp_bytes, L.keys, L.keys_s, L.val, (int *)d_perm, (unsigned)N,
                                                            sp.begin_bit, sp.end_bit, c->stream));
    else
        PEDP_ROCPRIM(rocprim::radix_sort_pairs(L.tmp, L.tmp_bytes, L.keys, L.keys_s, L.val, (int *)d_perm, (unsigned)N, sp.begin_bit,
                                               sp.end_bit, c->stream));
    PEDP_HIP_CHECK(hipGetLastError());
    return PEDP_OK;
}
