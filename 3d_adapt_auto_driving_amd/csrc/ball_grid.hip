// ball_grid.hip -- exact ball query (K1 semantics, ball_query_gpu.cu:9-45) through a hashed uniform
// grid, for large clouds.
//
// The brute-force kernel (ball_group.hip) tests all M*N pairs: 67 M distance tests per scene at
// N=16384, VALU-bound far above the few MB the op actually has to move.  Here the cloud is binned on
// (x, z) into cells of edge s = 1.001*r, hashed into H = 2^k >= 2N buckets per scene (no extent has to
// be known).  Buckets are LINKED LISTS built with one atomicExch per point -- no histogram, scan or
// scatter passes: node[k] = (x, y, z, next) is a single 16-byte record, so walking a bucket is one
// 16-byte load per candidate.  A centre visits only the <= 9 buckets of its 3x3 cell neighbourhood.
//
// The reference's result -- the FIRST nsample in-radius indices in index order, first hit
// back-filled -- does not depend on visiting order: every in-radius index is inserted into a per-lane
// sorted list (LDS, [slot][lane], keeps the nsample smallest), so the output is bit-identical to the
// brute-force scan.  The in-radius test is the same f32 expression.
//
// Cell coordinates are computed in f64 and the cell edge carries a 0.1 % margin, so a point with
// d^2 < r^2 can never fall outside the 3x3 neighbourhood through rounding.  Hash collisions only add
// candidates; a bucket reached through two neighbour cells is visited once.
#include "common.hpp"
#include <math.h>

namespace prcnn {

__device__ __forceinline__ unsigned cell_hash(int ix, int iz, unsigned mask)
{
    return (((unsigned)ix * 73856093u) ^ ((unsigned)iz * 19349663u)) & mask;
}

__device__ __forceinline__ int cell_coord(float v, double inv_s)
{
    double c = floor((double)v * inv_s);
    c = fmin(fmax(c, -1.0e9), 1.0e9);
    return (int)c;
}

// head[b][bucket] = index of the most recently inserted point (-1 = empty, set by the memset 0xFF)
__global__ __launch_bounds__(256) void grid_link_kernel(int n, unsigned mask, double inv_s,
                                                        const float *__restrict__ xyz, int *__restrict__ head,
                                                        float4 *__restrict__ node)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float *p = xyz + ((long)b * n + k) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const unsigned key = cell_hash(cell_coord(x, inv_s), cell_coord(z, inv_s), mask);
    const int prev = atomicExch(head + (long)b * (mask + 1) + key, k);
    node[(long)b * n + k] = make_float4(x, y, z, __int_as_float(prev));
}

constexpr int QT = 64;   // query threads per block (one wave): 512 blocks at b*m = 32768 centres

__global__ __launch_bounds__(QT) void grid_query_kernel(
    int n, int m, unsigned mask, double inv_s, float r2, int nsample, const float *__restrict__ new_xyz,
    const int *__restrict__ head, const float4 *__restrict__ node, int *__restrict__ idx, int write_empty)
{
    extern __shared__ int lst[];  // [nsample][QT]: per-lane ascending list of hit indices
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const int p = blockIdx.x * QT + t;
    if (p >= m) return;
    const float *c = new_xyz + ((long)b * m + p) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const int ix = cell_coord(cx, inv_s), iz = cell_coord(cz, inv_s);
    const int *__restrict__ hd = head + (long)b * (mask + 1);
    const float4 *__restrict__ nd = node + (long)b * n;
    int *mine = lst + t;

    // all nine bucket heads are fetched up front (independent loads), then the lists are walked
    unsigned seen[9];
    int first[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const unsigned key = cell_hash(ix + q / 3 - 1, iz + q % 3 - 1, mask);
        seen[q] = key;
        bool dup = false;
#pragma unroll
        for (int e = 0; e < q; ++e) dup |= (seen[e] == key);
        first[q] = dup ? -1 : hd[key];
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        for (int k = first[q]; k >= 0;) {
            const float4 pt = nd[k];
            const int cur = k;
            k = __float_as_int(pt.w);
            if (!(sqdist3(cx, cy, cz, pt.x, pt.y, pt.z) < r2)) continue;
            if (cnt == nsample && cur > mine[(nsample - 1) * QT]) continue;   // not among the nsample smallest
            int pos = cnt < nsample ? cnt : nsample - 1;
            while (pos > 0 && mine[(pos - 1) * QT] > cur) {
                mine[pos * QT] = mine[(pos - 1) * QT];
                --pos;
            }
            mine[pos * QT] = cur;
            if (cnt < nsample) ++cnt;
        }
    }
    int *out = idx + ((long)b * m + p) * nsample;
    if (cnt == 0) {
        if (write_empty)
            for (int l = 0; l < nsample; ++l) out[l] = 0;
        return;
    }
    const int lowest = mine[0];
    for (int l = 0; l < nsample; ++l) out[l] = l < cnt ? mine[l * QT] : lowest;
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// Returns PRCNN_OK, or an error; *used = 0 when the grid path declines (caller falls back).
int ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                    int *idx, int write_empty, hipStream_t st, int *used)
{
    *used = 0;
    if (!(radius > 0.f) || !isfinite(radius) || n < 4096 || m < 64 || nsample > 128 || b > 65535) return PRCNN_OK;
    unsigned H = 1;
    while (H < 2u * (unsigned)n) H <<= 1;
    const size_t o_head = 0;
    const size_t o_node = align_up((size_t)b * H * sizeof(int));
    const size_t need = o_node + align_up((size_t)b * n * sizeof(float4));
    char *base = scratch_for(st, need);
    if (!base) { set_error("ball_query: cannot allocate %zu bytes of grid scratch", need); return PRCNN_ELAUNCH; }
    int *head = (int *)(base + o_head);
    float4 *node = (float4 *)(base + o_node);

    const double inv_s = 1.0 / ((double)radius * 1.001);
    if (hipMemsetAsync(head, 0xFF, (size_t)b * H * sizeof(int), st) != hipSuccess) {
        set_error("ball_query: memset failed");
        return PRCNN_ELAUNCH;
    }
    hipLaunchKernelGGL(grid_link_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, n, H - 1, inv_s, xyz, head, node);
    const size_t lds = (size_t)nsample * QT * sizeof(int);
    hipLaunchKernelGGL(grid_query_kernel, dim3(ceil_div(m, QT), b), dim3(QT), lds, st, n, m, H - 1, inv_s,
                       radius * radius, nsample, new_xyz, head, node, idx, write_empty);
    *used = 1;
    return check_launch("ball_query(grid)");
}

}  // namespace prcnn
