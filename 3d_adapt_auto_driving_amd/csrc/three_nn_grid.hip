// three_nn_grid.hip -- exact three_nn (K7 semantics, interpolate_gpu.cu:9-52) through a dense uniform
// grid over the KNOWN points, for large (n, m).
//
// The brute-force kernel (interp.hip) tests all n*m pairs: 67 M distance tests per scene at
// (16384, 4096), pure VALU that fights the MFMA kernels for issue slots when the two streams co-run.
// Here the m known points of a scene are binned on (x, z) into a G x G grid over their own bounding
// rectangle (G ~ sqrt(m/2), out-of-extent coordinates clamp to the border cells) and counting-sorted
// by cell, all inside ONE workgroup per scene (histogram + scan in LDS).  Cells of one grid row are
// contiguous in the sorted array, so a row segment of the search square is a single index range.
//
// A query walks Chebyshev rings R = 0, 1, 2, ... around its own (clamped) cell.  After ring R every
// unvisited point lies in a cell at Chebyshev distance >= R+1, hence at distance >= R*s in the (x, z)
// plane (clamping only moves cells closer than the points really are).  The walk stops as soon as the
// third-best squared distance is below (R*s)^2 (minus a 1e-5 relative margin for the rounding of the
// cell assignment), or when the square covers the grid.
//
// The reference result -- the three smallest f32 squared distances, lowest index first among equals --
// does not depend on visiting order once insertion compares (d, index) lexicographically: the strict
// '<' scan in index order keeps exactly the lowest indices among equal distances.  The distance is the
// same f32 expression, so dist2 and idx are bit-identical to the brute-force scan.
//
// Round 3 (LiDAR-shaped clouds: the FPS-picked known points of a sweep occupy an eighth of their bounding rectangle, up to
// 90 of them per cell of the round-2 grid, 300 candidates per query, every lane of a wave in a different cell: 1.67 ms for
// one geometry group against 0.26 ms on the uniform scene): (1) the grid is twice as fine (G ~ sqrt(2 m)); (2) the QUERIES are
// counting-sorted by cell as well (same workgroup, same histogram memory) and served in that order, so the lanes of a wave sit
// in the same or neighbouring cells, walk the same ranges and their candidate loads coalesce into one or two cache lines;
// (3) candidates are fetched four at a time.  Results are written at the query's original position: same bits as before.
#include "common.hpp"
#include <math.h>
#include <stdlib.h>

namespace prcnn {

constexpr int TG_MAX = 96;                 // grid edge limit: 9216 cells, 36 KB of LDS histogram
constexpr int TB = 1024;                   // build workgroup

struct GridParams {                         // one per scene, written by the build kernel
    double x0, z0, inv_s;
    float s;
    int g;
};

__device__ __forceinline__ int grid_coord(float v, double o, double inv_s, int g)
{
    double c = floor(((double)v - o) * inv_s);
    c = fmin(fmax(c, 0.0), (double)(g - 1));   // NaN -> 0
    return (int)c;
}

__global__ __launch_bounds__(TB) void tnn_build_kernel(int m, int g, const float *__restrict__ known,
                                                       GridParams *__restrict__ params, int *__restrict__ cell_start,
                                                       float4 *__restrict__ sorted, int n, const float *__restrict__ unknown,
                                                       int *__restrict__ order)
{
    extern __shared__ int hist[];          // g*g counters, then cursors
    __shared__ float red[4][TB / 64];
    __shared__ int wsum[TB / 64];
    __shared__ GridParams gp;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *__restrict__ kn = known + (long)b * m * 3;
    const int cells = g * g;

    // bounding rectangle of the finite points
    float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
    for (int k = t; k < m; k += TB) {
        const float x = kn[3 * k], z = kn[3 * k + 2];
        if (isfinite(x)) { xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); }
        if (isfinite(z)) { zmin = fminf(zmin, z); zmax = fmaxf(zmax, z); }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, off)); xmax = fmaxf(xmax, __shfl_xor(xmax, off));
        zmin = fminf(zmin, __shfl_xor(zmin, off)); zmax = fmaxf(zmax, __shfl_xor(zmax, off));
    }
    if (lane == 0) { red[0][w] = xmin; red[1][w] = xmax; red[2][w] = zmin; red[3][w] = zmax; }
    for (int c = t; c < cells; c += TB) hist[c] = 0;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < TB / 64; ++i) {
            xmin = fminf(xmin, red[0][i]); xmax = fmaxf(xmax, red[1][i]);
            zmin = fminf(zmin, red[2][i]); zmax = fmaxf(zmax, red[3][i]);
        }
        if (!(xmin <= xmax)) { xmin = 0.f; xmax = 0.f; }
        if (!(zmin <= zmax)) { zmin = 0.f; zmax = 0.f; }
        double ext = fmax((double)xmax - (double)xmin, (double)zmax - (double)zmin);
        if (!(ext > 1e-6)) ext = 1e-6;
        const double s = ext / g * (1.0 + 1e-9);
        gp.x0 = xmin; gp.z0 = zmin; gp.inv_s = 1.0 / s; gp.s = (float)s; gp.g = g;
        params[b] = gp;
    }
    __syncthreads();
    const double x0 = gp.x0, z0 = gp.z0, inv_s = gp.inv_s;

    for (int k = t; k < m; k += TB) {
        const int c = grid_coord(kn[3 * k + 2], z0, inv_s, g) * g + grid_coord(kn[3 * k], x0, inv_s, g);
        atomicAdd(&hist[c], 1);
    }
    __syncthreads();

    // exclusive scan of the cell counts: each thread owns a contiguous chunk
    const int per = (cells + TB - 1) / TB;
    const int lo = t * per, hi = min(lo + per, cells);
    int sum = 0;
    for (int c = lo; c < hi; ++c) sum += hist[c];
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int i = 0; i < w; ++i) base += wsum[i];
    int *__restrict__ cs = cell_start + (long)b * (TG_MAX * TG_MAX + 1);
    for (int c = lo; c < hi; ++c) {
        const int cnt = hist[c];
        cs[c] = base;
        hist[c] = base;                    // becomes the scatter cursor
        base += cnt;
    }
    if (t == TB - 1) cs[cells] = m;
    __syncthreads();

    float4 *__restrict__ so = sorted + (long)b * m;
    for (int k = t; k < m; k += TB) {
        const float x = kn[3 * k], y = kn[3 * k + 1], z = kn[3 * k + 2];
        const int c = grid_coord(z, z0, inv_s, g) * g + grid_coord(x, x0, inv_s, g);
        const int pos = atomicAdd(&hist[c], 1);
        so[pos] = make_float4(x, y, z, __int_as_float(k));
    }
    if (!order) return;

    // the queries, counting-sorted by the cell they fall into (clamped like the ring search clamps them): order[b][0..n)
    __syncthreads();
    const float *__restrict__ un = unknown + (long)b * n * 3;
    for (int c = t; c < cells; c += TB) hist[c] = 0;
    __syncthreads();
    for (int k = t; k < n; k += TB) {
        const int c = grid_coord(un[3 * k + 2], z0, inv_s, g) * g + grid_coord(un[3 * k], x0, inv_s, g);
        atomicAdd(&hist[c], 1);
    }
    __syncthreads();
    sum = 0;
    for (int c = lo; c < hi; ++c) sum += hist[c];
    incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    __syncthreads();                           // wsum is rewritten
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    base = incl - sum;
    for (int i = 0; i < w; ++i) base += wsum[i];
    for (int c = lo; c < hi; ++c) {
        const int cnt = hist[c];
        hist[c] = base;
        base += cnt;
    }
    __syncthreads();
    int *__restrict__ od = order + (long)b * n;
    for (int k = t; k < n; k += TB) {
        const int c = grid_coord(un[3 * k + 2], z0, inv_s, g) * g + grid_coord(un[3 * k], x0, inv_s, g);
        od[atomicAdd(&hist[c], 1)] = k;
    }
}

// The same build with both clouds in the workgroup's REGISTERS (round 4; n <= 16 * 1024 queries, m <= 4 * 1024 known points:
// every FP level of the backbone).  The kernel above reads the known points three times and the queries twice from global memory,
// each time as a loop of dependent (load -> LDS atomic) rounds; here every coordinate is loaded once, all loads in flight
// together, and the rest is LDS and ALU work.  Same cell assignment, same counting sort: the query kernel sees the same tables
// up to the order of points INSIDE a cell (atomics), which the (d, index) insertion makes irrelevant.
template <int NK, int NQ>
__global__ __launch_bounds__(TB) void tnn_build_reg_kernel(int m, int g, const float *__restrict__ known,
                                                           GridParams *__restrict__ params, int *__restrict__ cell_start,
                                                           float4 *__restrict__ sorted, int n, const float *__restrict__ unknown,
                                                           int *__restrict__ order)
{
    extern __shared__ int hist[];          // g*g counters, then cursors
    __shared__ float red[4][TB / 64];
    __shared__ int wsum[TB / 64];
    __shared__ GridParams gp;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *__restrict__ kn = known + (long)b * m * 3;
    const float *__restrict__ un = unknown + (long)b * n * 3;
    const int cells = g * g;
    float kx[NK], ky[NK], kz[NK], qx[NQ], qz[NQ];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int k = j * TB + t;
        kx[j] = ky[j] = kz[j] = 0.f;
        if (k < m) { kx[j] = kn[3 * k]; ky[j] = kn[3 * k + 1]; kz[j] = kn[3 * k + 2]; }
    }
    if (order) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int k = j * TB + t;
            qx[j] = qz[j] = 0.f;
            if (k < n) { qx[j] = un[3 * k]; qz[j] = un[3 * k + 2]; }
        }
    }
    // bounding rectangle of the finite known points
    float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        if (j * TB + t >= m) continue;
        if (isfinite(kx[j])) { xmin = fminf(xmin, kx[j]); xmax = fmaxf(xmax, kx[j]); }
        if (isfinite(kz[j])) { zmin = fminf(zmin, kz[j]); zmax = fmaxf(zmax, kz[j]); }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, off)); xmax = fmaxf(xmax, __shfl_xor(xmax, off));
        zmin = fminf(zmin, __shfl_xor(zmin, off)); zmax = fmaxf(zmax, __shfl_xor(zmax, off));
    }
    if (lane == 0) { red[0][w] = xmin; red[1][w] = xmax; red[2][w] = zmin; red[3][w] = zmax; }
    for (int c = t; c < cells; c += TB) hist[c] = 0;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < TB / 64; ++i) {
            xmin = fminf(xmin, red[0][i]); xmax = fmaxf(xmax, red[1][i]);
            zmin = fminf(zmin, red[2][i]); zmax = fmaxf(zmax, red[3][i]);
        }
        if (!(xmin <= xmax)) { xmin = 0.f; xmax = 0.f; }
        if (!(zmin <= zmax)) { zmin = 0.f; zmax = 0.f; }
        double ext = fmax((double)xmax - (double)xmin, (double)zmax - (double)zmin);
        if (!(ext > 1e-6)) ext = 1e-6;
        const double s = ext / g * (1.0 + 1e-9);
        gp.x0 = xmin; gp.z0 = zmin; gp.inv_s = 1.0 / s; gp.s = (float)s; gp.g = g;
        params[b] = gp;
    }
    __syncthreads();
    const double x0 = gp.x0, z0 = gp.z0, inv_s = gp.inv_s;
    int kc[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        kc[j] = grid_coord(kz[j], z0, inv_s, g) * g + grid_coord(kx[j], x0, inv_s, g);
        if (j * TB + t < m) atomicAdd(&hist[kc[j]], 1);
    }
    __syncthreads();
    // exclusive scan of the cell counts: each thread owns a contiguous chunk (cells of a grid row must stay contiguous)
    const int per = (cells + TB - 1) / TB;
    const int lo = min(t * per, cells), hi = min(lo + per, cells);
    int sum = 0;
    for (int c = lo; c < hi; ++c) sum += hist[c];
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int i = 0; i < w; ++i) base += wsum[i];
    int *__restrict__ cs = cell_start + (long)b * (TG_MAX * TG_MAX + 1);
    for (int c = lo; c < hi; ++c) {
        const int cnt = hist[c];
        cs[c] = base;
        hist[c] = base;                    // becomes the scatter cursor
        base += cnt;
    }
    if (t == TB - 1) cs[cells] = m;
    __syncthreads();
    float4 *__restrict__ so = sorted + (long)b * m;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int k = j * TB + t;
        if (k < m) so[atomicAdd(&hist[kc[j]], 1)] = make_float4(kx[j], ky[j], kz[j], __int_as_float(k));
    }
    if (!order) return;

    // the queries, counting-sorted by the cell they fall into (clamped like the ring search clamps them): order[b][0..n)
    __syncthreads();
    for (int c = t; c < cells; c += TB) hist[c] = 0;
    __syncthreads();
    int qc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        qc[j] = grid_coord(qz[j], z0, inv_s, g) * g + grid_coord(qx[j], x0, inv_s, g);
        if (j * TB + t < n) atomicAdd(&hist[qc[j]], 1);
    }
    __syncthreads();
    sum = 0;
    for (int c = lo; c < hi; ++c) sum += hist[c];
    incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    __syncthreads();                           // wsum is rewritten
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    base = incl - sum;
    for (int i = 0; i < w; ++i) base += wsum[i];
    for (int c = lo; c < hi; ++c) {
        const int cnt = hist[c];
        hist[c] = base;
        base += cnt;
    }
    __syncthreads();
    int *__restrict__ od = order + (long)b * n;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int k = j * TB + t;
        if (k < n) od[atomicAdd(&hist[qc[j]], 1)] = k;
    }
}

// one query against one cloud's tables (cs: cell starts, so: known points sorted by cell) -- in global memory or staged in LDS
template <typename CS, typename SO>
__device__ __forceinline__ void tnn_search(const GridParams &gp, float ux, float uy, float uz, CS cs, SO so,
                                           float &b1, float &b2, float &b3, int &i1, int &i2, int &i3)
{
    const int g = gp.g;
    const int ix = grid_coord(ux, gp.x0, gp.inv_s, g), iz = grid_coord(uz, gp.z0, gp.inv_s, g);
    b1 = INFINITY; b2 = INFINITY; b3 = INFINITY;
    i1 = 0; i2 = 0; i3 = 0;
    // branch-free insertion into the sorted triple ((b1,i1) <= (b2,i2) <= (b3,i3) lexicographically, so lt1 => lt2 => lt3).  Written
    // as an if / else-if chain (round 3) the compiler turned the triple into a dynamically indexed STACK array: 141 scratch loads and
    // 126 scratch stores in the kernel's ISA, every candidate a round trip through scratch memory.
    auto take = [&](const float4 pt) __attribute__((always_inline)) {
        const int k = __float_as_int(pt.w);
        const float d = sqdist3(ux, uy, uz, pt.x, pt.y, pt.z);
        const bool lt1 = d < b1 || (d == b1 && k < i1);
        const bool lt2 = d < b2 || (d == b2 && k < i2);
        const bool lt3 = d < b3 || (d == b3 && k < i3);
        b3 = lt2 ? b2 : (lt3 ? d : b3); i3 = lt2 ? i2 : (lt3 ? k : i3);
        b2 = lt1 ? b1 : (lt2 ? d : b2); i2 = lt1 ? i1 : (lt2 ? k : i2);
        b1 = lt1 ? d : b1;              i1 = lt1 ? k : i1;
    };
    auto scan = [&](int first, int last) __attribute__((always_inline)) {
        int q = first;
        for (; q + 4 <= last; q += 4) {                             // four independent 16-byte loads in flight
            const float4 p0 = so[q], p1 = so[q + 1], p2 = so[q + 2], p3 = so[q + 3];
            take(p0); take(p1); take(p2); take(p3);
        }
        for (; q < last; ++q) take(so[q]);
    };
    const int reach = max(max(ix, g - 1 - ix), max(iz, g - 1 - iz));   // ring that covers the whole grid
    for (int r = 0; r <= reach; ++r) {
        const int xlo = max(ix - r, 0), xhi = min(ix + r, g - 1);
        for (int dz = -r; dz <= r; ++dz) {
            const int z = iz + dz;
            if (z < 0 || z >= g) continue;
            const int row = z * g;
            if (dz == -r || dz == r) {
                scan(cs[row + xlo], cs[row + xhi + 1]);                // a full row segment: one range
            } else {
                if (ix - r >= 0) scan(cs[row + ix - r], cs[row + ix - r + 1]);
                if (ix + r < g) scan(cs[row + ix + r], cs[row + ix + r + 1]);
            }
        }
        const float bound = (float)r * gp.s;
        if (b3 < bound * bound * 0.99999f) break;
    }
}

__device__ __forceinline__ void tnn_store(int b, int n, int p, float b1, float b2, float b3, int i1, int i2, int i3,
                                          float *__restrict__ dist2, int *__restrict__ idx, float *__restrict__ weight)
{
    int *oi = idx + ((long)b * n + p) * 3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
    if (dist2) {
        float *od = dist2 + ((long)b * n + p) * 3;
        od[0] = b1; od[1] = b2; od[2] = b3;
    }
    if (weight) three_nn_weights(b1, b2, b3, weight + ((long)b * n + p) * 3);
}

__global__ __launch_bounds__(256) void tnn_query_kernel(int n, int m, const float *__restrict__ unknown,
                                                        const GridParams *__restrict__ params,
                                                        const int *__restrict__ cell_start,
                                                        const float4 *__restrict__ sorted, float *__restrict__ dist2,
                                                        int *__restrict__ idx, const int *__restrict__ order,
                                                        float *__restrict__ weight)
{
    const int b = blockIdx.y;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n) return;
    const int p = order ? order[(long)b * n + slot] : slot;       // queries in cell order: a wave's lanes are neighbours
    const GridParams gp = params[b];
    const float *u = unknown + ((long)b * n + p) * 3;
    float b1, b2, b3;
    int i1, i2, i3;
    tnn_search(gp, u[0], u[1], u[2], cell_start + (long)b * (TG_MAX * TG_MAX + 1), sorted + (long)b * m, b1, b2, b3, i1, i2, i3);
    tnn_store(b, n, p, b1, b2, b3, i1, i2, i3, dist2, idx, weight);
}

// The same search with the cloud's tables STAGED IN LDS (round 4; m <= 4096 known points: 64 KB of sorted points + <= 36 KB of cell
// starts per workgroup of 1024 queries).  A query's ring walk is a chain of dependent loads -- cell starts, then candidates, ring
// after ring; from global memory each link costs an L2 round trip and a batch of 8 clouds has only 8 waves per CU to hide it
// behind (62 us for 8 x 16384 queries); from LDS a link is ~100 ns.  Same visiting order, same insertion: same bits.
__global__ __launch_bounds__(1024) void tnn_query_lds_kernel(int n, int m, const float *__restrict__ unknown,
                                                             const GridParams *__restrict__ params,
                                                             const int *__restrict__ cell_start,
                                                             const float4 *__restrict__ sorted, float *__restrict__ dist2,
                                                             int *__restrict__ idx, const int *__restrict__ order,
                                                             float *__restrict__ weight)
{
    extern __shared__ float4 tnn_lds[];        // m points, then g*g + 1 cell starts
    const int b = blockIdx.y, t = threadIdx.x;
    const GridParams gp = params[b];
    const int cells1 = gp.g * gp.g + 1;
    float4 *lpts = tnn_lds;
    int *lcs = reinterpret_cast<int *>(tnn_lds + m);
    const float4 *__restrict__ so = sorted + (long)b * m;
    const int *__restrict__ cs = cell_start + (long)b * (TG_MAX * TG_MAX + 1);
    for (int i = t; i < m; i += 1024) lpts[i] = so[i];
    for (int i = t; i < cells1; i += 1024) lcs[i] = cs[i];
    __syncthreads();
    const int slot = blockIdx.x * 1024 + t;
    if (slot >= n) return;
    const int p = order ? order[(long)b * n + slot] : slot;
    const float *u = unknown + ((long)b * n + p) * 3;
    float b1, b2, b3;
    int i1, i2, i3;
    tnn_search(gp, u[0], u[1], u[2], lcs, lpts, b1, b2, b3, i1, i2, i3);
    tnn_store(b, n, p, b1, b2, b3, i1, i2, i3, dist2, idx, weight);
}

static size_t align_up256(size_t v) { return (v + 255) & ~(size_t)255; }

// Returns PRCNN_OK or an error; *used = 0 when the grid path declines (caller runs the brute-force scan).
int three_nn_grid(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                  hipStream_t st, int *used, float *weight)
{
    *used = 0;
    if (m < 1024 || n < 1024 || m > (1 << 20)) return PRCNN_OK;
    static const double cells_per_point = [] { const char *e = getenv("PRCNN_TNN_CELLS"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 2.0; }();
    int g = (int)ceil(sqrt((double)m * cells_per_point));          // round 2: sqrt(m / 2) (PRCNN_TNN_CELLS=0.5)
    if (g > TG_MAX) g = TG_MAX;
    const bool ordered = true;           // queries served in cell order (round 6: switches PRCNN_TNN_UNORDERED / PRCNN_TNN_NO_LDS removed)
    const size_t o_par = 0;
    const size_t o_cs = align_up256((size_t)b * sizeof(GridParams));
    const size_t o_sorted = o_cs + align_up256((size_t)b * (TG_MAX * TG_MAX + 1) * sizeof(int));
    const size_t o_order = o_sorted + align_up256((size_t)b * m * sizeof(float4));
    const size_t need = o_order + align_up256((size_t)b * n * sizeof(int));
    char *base = scratch_for(st, need, 4);
    if (!base) { set_error("three_nn: cannot allocate %zu bytes of grid scratch", need); return PRCNN_ELAUNCH; }
    GridParams *params = (GridParams *)(base + o_par);
    int *cs = (int *)(base + o_cs);
    float4 *sorted = (float4 *)(base + o_sorted);
    int *order = ordered ? (int *)(base + o_order) : nullptr;
    if (m <= 4 * TB && n <= 16 * TB)
        hipLaunchKernelGGL((tnn_build_reg_kernel<4, 16>), dim3(b), dim3(TB), (size_t)g * g * sizeof(int), st, m, g, known, params, cs,
                           sorted, n, unknown, order);
    else
        hipLaunchKernelGGL(tnn_build_kernel, dim3(b), dim3(TB), (size_t)g * g * sizeof(int), st, m, g, known, params, cs, sorted,
                           n, unknown, order);
    const size_t qlds = (size_t)m * sizeof(float4) + ((size_t)g * g + 1) * sizeof(int);
    const bool no_lds = false;
    if (!no_lds && qlds <= 128 * 1024 && n >= 2 * m) {
        const int rc = ensure_dynamic_lds((const void *)tnn_query_lds_kernel, qlds, "three_nn(query)");
        if (rc != PRCNN_OK) return rc;
        hipLaunchKernelGGL(tnn_query_lds_kernel, dim3(ceil_div(n, 1024), b), dim3(1024), qlds, st, n, m, unknown, params, cs,
                           sorted, dist2, idx, order, weight);
    } else
        hipLaunchKernelGGL(tnn_query_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, n, m, unknown, params, cs,
                           sorted, dist2, idx, order, weight);
    *used = 1;
    return check_launch("three_nn(grid)");
}

}  // namespace prcnn
