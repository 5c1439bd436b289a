// ball_dense.hip -- exact ball query (K1 semantics, ball_query_gpu.cu:9-45) for clouds whose density varies by orders of
// magnitude (LiDAR sweeps: hundreds of points per cell near the sensor, none at 60 m).  Round 3.
//
// ball_grid.hip (round 1) hashes the cloud into (x, z) cells of edge 1.001 r whose buckets are LINKED LISTS, one lane per
// centre walking up to nine lists.  On the uniform synthetic scene a list holds a handful of nodes; on a LiDAR-shaped scene
// (synth.lidar_scene) the nine cells of a centre near the sensor hold ~2000 nodes, every hop is a dependent 16-byte load,
// and the slowest lane of a wave sets its pace: 3.5 ms for the r = 0.5 query of one geometry group (32 clouds) against
// 0.11 ms on the uniform scene (profiles/r03_geo_probe.md).
//
// Here the same hashed cells are stored CONTIGUOUSLY (a counting sort by bucket inside one workgroup per cloud: LDS
// histogram, scan, scatter of 16-byte (x, y, z, index) records; round 4: with the cloud in the workgroup's registers, see
// dense_build_reg_kernel) and a WAVE serves one centre:
//   * lanes 0..8 fetch the nine bucket ranges (duplicate buckets -- two cells hashing alike -- are visited once);
//   * the candidates are read 64 at a time, coalesced, tested with the reference's f32 expression (sqdist3), and
//   * the wave keeps the nsample SMALLEST hit indices sorted across its lanes (lane i = i-th smallest; an insertion is one
//     wave_shr:1 DPP move and two compares), which is the reference's "first nsample in index order" whatever the order
//     of visiting; a hit that is not below the current nsample-th smallest is dropped by one compare.
//   * The build hands out bucket slots in index-ordered CHUNKS (a barrier between chunks of n/16 points), so inside a bucket
//     the records are sorted by chunk: once nsample hits are known and a range has reached a chunk beyond that of the
//     nsample-th smallest, the rest of the range cannot contribute and is skipped -- in a full ball most of every range.
// Same results as the index-order scan bit for bit (tests/test_gpu_ops.py: all BASELINE configs[1] pairs on uniform, LiDAR-
// shaped and duplicate-point clouds; shadowed at batch 8 in tests/test_gpu_shadow.py).
#include "common.hpp"
#include <math.h>

namespace prcnn {

__device__ __forceinline__ unsigned dcell_hash(int ix, int iz, unsigned mask)
{
    return (((unsigned)ix * 73856093u) ^ ((unsigned)iz * 19349663u)) & mask;
}

__device__ __forceinline__ int dcell_coord(float v, double inv_s)
{
    double c = floor((double)v * inv_s);
    c = fmin(fmax(c, -1.0e9), 1.0e9);
    return (int)c;
}

constexpr int DB = 1024;        // build workgroup: one per cloud

// GENERIC build (16384 < n <= 65535; the fast one follows).  range[b][0..H): (first record, length) of every bucket;
// sorted[b][n]: (x, y, z, index) by bucket, by chunk inside
__global__ __launch_bounds__(DB) void dense_build_kernel(int n, unsigned mask, double inv_s, int chunk,
                                                         const float *__restrict__ xyz, int2 *__restrict__ range,
                                                         float4 *__restrict__ sorted, unsigned short *__restrict__ rank)
{
    extern __shared__ int cnt[];               // H counters, then the buckets' start offsets
    __shared__ int wsum[DB / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int H = (int)mask + 1;
    const float *__restrict__ pts = xyz + (long)b * n * 3;
    unsigned short *__restrict__ rk = rank + (long)b * n;
    for (int c = t; c < H; c += DB) cnt[c] = 0;
    __syncthreads();
    // slots inside a bucket are handed out chunk by chunk in index order: a bucket's records end up sorted by chunk
    for (int base = 0; base < n; base += chunk) {
        const int hi = min(base + chunk, n);
        for (int k = base + t; k < hi; k += DB) {
            const unsigned key = dcell_hash(dcell_coord(pts[3 * k], inv_s), dcell_coord(pts[3 * k + 2], inv_s), mask);
            rk[k] = (unsigned short)atomicAdd(&cnt[key], 1);
        }
        __syncthreads();
    }
    // exclusive scan of the H counts; thread t owns `per` consecutive buckets
    const int per = H / DB;                    // H is a power of two >= 2048
    const int lo = t * per;
    int sum = 0;
    for (int c = lo; c < lo + per; ++c) sum += cnt[c];
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int i = 0; i < w; ++i) run += wsum[i];
    int2 *__restrict__ st = range + (long)b * H;
    for (int c = lo; c < lo + per; ++c) {
        const int v = cnt[c];
        cnt[c] = run;
        st[c] = make_int2(run, v);
        run += v;
    }
    __syncthreads();
    float4 *__restrict__ so = sorted + (long)b * n;
    for (int k = t; k < n; k += DB) {
        const float x = pts[3 * k], y = pts[3 * k + 1], z = pts[3 * k + 2];
        const unsigned key = dcell_hash(dcell_coord(x, inv_s), dcell_coord(z, inv_s), mask);
        so[cnt[key] + (int)rk[k]] = make_float4(x, y, z, __int_as_float(k));
    }
}


// FAST build, n <= 16384 (every level of the RPN backbone): the cloud lives in REGISTERS (thread t holds points t, t + 1024, ...),
// so the 16 rank-assignment rounds are LDS atomics + barriers only (the generic kernel re-reads the points from global memory in
// every round: 16 dependent global round trips + 16 more in the scatter = most of the ~50 us it takes for 8 clouds, measured as
// "B = 1, C = 0: 79 us per call" in profiles/r04_query_and_group_sweep.md).  A chunk is one round = 1024 consecutive indices
// (chunk_shift 10).  The bucket ranges are laid out in the order (thread, slice) -- thread t owns buckets t, t + 1024, ... --
// so that the scan reads LDS conflict-free; the query reads a bucket's (start, length) pair, the layout order is free.
template <int PPT>
__global__ __launch_bounds__(DB) void dense_build_reg_kernel(int n, unsigned mask, double inv_s, const float *__restrict__ xyz,
                                                             int2 *__restrict__ range, float4 *__restrict__ sorted)
{
    extern __shared__ int cnt[];               // H counters, then the buckets' start offsets
    __shared__ int wsum[DB / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int H = (int)mask + 1;
    const float *__restrict__ pts = xyz + (long)b * n * 3;
    float px[PPT], py[PPT], pz[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = j * DB + t;
        px[j] = py[j] = pz[j] = 0.f;
        if (k < n) { px[j] = pts[3 * k]; py[j] = pts[3 * k + 1]; pz[j] = pts[3 * k + 2]; }
    }
    int4 *cnt4 = reinterpret_cast<int4 *>(cnt);
    for (int c = t; c < H / 4; c += DB) cnt4[c] = make_int4(0, 0, 0, 0);
    int kr[PPT];                               // bucket << 16 | rank inside the bucket (rank < 16384, bucket < 32768)
#pragma unroll
    for (int j = 0; j < PPT; ++j)
        kr[j] = (int)(dcell_hash(dcell_coord(px[j], inv_s), dcell_coord(pz[j], inv_s), mask) << 16);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {            // round j = index chunk j: a bucket's records end up sorted by chunk
        if (j * DB + t < n) kr[j] |= atomicAdd(&cnt[(unsigned)kr[j] >> 16], 1);
        if ((j + 1) * DB < n + DB) __syncthreads();
    }
    // exclusive scan of the H counts in the order (thread, slice): thread t owns buckets t + 1024 i
    const int per = H / DB;                    // H is a power of two >= 2048
    int sum = 0;
    for (int i = 0; i < per; ++i) sum += cnt[i * DB + t];
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int i = 0; i < w; ++i) run += wsum[i];
    int2 *__restrict__ st = range + (long)b * H;
    for (int i = 0; i < per; ++i) {
        const int c = i * DB + t;
        const int v = cnt[c];
        cnt[c] = run;
        st[c] = make_int2(run, v);
        run += v;
    }
    __syncthreads();
    float4 *__restrict__ so = sorted + (long)b * n;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = j * DB + t;
        if (k < n) so[cnt[(unsigned)kr[j] >> 16] + (kr[j] & 0xffff)] = make_float4(px[j], py[j], pz[j], __int_as_float(k));
    }
}

constexpr int DQ_WAVES = 4;     // centres per query workgroup (one wave each)
constexpr int DQ_FLAT = 192;    // up to this many candidates the nine ranges are read as ONE enumeration (sparse balls)

// lane i <- lane i-1, lane 0 <- `first`  (DPP wave_shr:1, GFX9 family)
__device__ __forceinline__ int wave_shr1(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

__global__ __launch_bounds__(64 * DQ_WAVES) void dense_query_kernel(
    int n, int m, unsigned mask, double inv_s, float r2, int nsample, int chunk_shift, const float *__restrict__ new_xyz,
    const int2 *__restrict__ range, const float4 *__restrict__ sorted, int *__restrict__ idx, int write_empty)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * DQ_WAVES + (threadIdx.x >> 6));
    if (p >= m) return;
    const float *c = new_xyz + ((long)b * m + p) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const int ix = dcell_coord(cx, inv_s), iz = dcell_coord(cz, inv_s);
    const int2 *__restrict__ st = range + (long)b * (mask + 1);
    const float4 *__restrict__ so = sorted + (long)b * n;

    // lanes 0..8: the nine buckets of the 3 x 3 cell neighbourhood, own cell first; a bucket reached twice is read once
    const int q = lane < 9 ? lane : 0;
    const int order = q == 0 ? 4 : (q <= 4 ? q - 1 : q);               // 4, 0, 1, 2, 3, 5, 6, 7, 8
    const unsigned key = dcell_hash(ix + order / 3 - 1, iz + order % 3 - 1, mask);
    bool dup = lane >= 9;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned ke = (unsigned)__builtin_amdgcn_readlane((int)key, e);
        dup |= (e < lane) && (ke == key);
    }
    int r_st = 0, r_len = 0;
    if (!dup) {
        const int2 r = st[key];
        r_st = r.x;
        r_len = r.y;
    }
    int S[9], L[9], total = 0;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        S[e] = __builtin_amdgcn_readlane(r_st, e);
        L[e] = __builtin_amdgcn_readlane(r_len, e);
        total += L[e];
    }

    int best = 0x7fffffff;        // lane i: the i-th smallest hit index so far
    int cnt = 0;                  // hits kept, <= nsample (wave-uniform)
    int T = 0x7fffffff;           // the nsample-th smallest once cnt == nsample: larger indices cannot enter

    auto absorb = [&](bool in, int id) {
        unsigned long long hits = __ballot(in);
        while (hits) {
            const int l = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int h = __builtin_amdgcn_readlane(id, l);
            if (h >= T) continue;
            const int up = wave_shr1(best, (int)0x80000000);
            best = best < h ? best : (up < h ? h : up);
            if (cnt < nsample) ++cnt;
            if (cnt == nsample) T = __builtin_amdgcn_readlane(best, nsample - 1);
        }
    };

    if (total <= DQ_FLAT) {
        // sparse neighbourhood: one enumeration over the nine ranges, 64 candidates per round
        for (int off = 0; off < total; off += 64) {
            int e = off + lane, addr = -1;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (addr < 0 && e < L[k]) addr = S[k] + e;
                e -= L[k];
            }
            const bool valid = off + lane < total;
            const float4 pt = so[valid ? addr : 0];
            const int id = __float_as_int(pt.w);
            absorb(valid && sqdist3(cx, cy, cz, pt.x, pt.y, pt.z) < r2 && id < T, id);
        }
    } else {
        // dense neighbourhood: range by range; a range is sorted by index chunk, so it ends for this centre at the first
        // round that starts in a chunk beyond the one of the nsample-th smallest hit
#pragma unroll 1
        for (int k = 0; k < 9; ++k) {
            const int len = L[k], s0 = S[k];
            for (int off = 0; off < len; off += 64) {
                const bool valid = off + lane < len;
                const float4 pt = so[s0 + (valid ? off + lane : off)];
                const int id = __float_as_int(pt.w);
                if (cnt == nsample && (__builtin_amdgcn_readfirstlane(id) >> chunk_shift) > (T >> chunk_shift)) break;
                absorb(valid && sqdist3(cx, cy, cz, pt.x, pt.y, pt.z) < r2 && id < T, id);
            }
        }
    }

    int *out = idx + ((long)b * m + p) * nsample;
    if (cnt == 0) {
        if (write_empty && lane < nsample) out[lane] = 0;
        return;
    }
    const int lowest = __builtin_amdgcn_readlane(best, 0);     // back-fill with the first hit (ball_query_gpu.cu:35-39)
    if (lane < nsample) out[lane] = lane < cnt ? best : lowest;
}

static size_t dalign(size_t v) { return (v + 255) & ~(size_t)255; }

// Returns PRCNN_OK, or an error; *used = 0 when this path declines (the caller falls back).
int ball_query_dense(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                     int *idx, int write_empty, hipStream_t st, int *used)
{
    *used = 0;
    if (!(radius > 0.f) || !isfinite(radius) || n < 2048 || n > 65535 || m < 64 || nsample > 64 || b > 65535) return PRCNN_OK;
    unsigned H = 2048;
    while (H < 2u * (unsigned)n && H < 32768u) H <<= 1;                 // 128 KB of LDS counters at most
    const bool fast = n <= 16 * DB;                                      // the cloud fits in the build workgroup's registers
    int chunk_shift = fast ? 10 : 6;                                     // fast: a chunk = a round of 1024 indices; generic: 16 chunks
    while (!fast && (1 << chunk_shift) * 16 < n) ++chunk_shift;
    const int chunk = 1 << chunk_shift;
    const size_t o_range = 0;
    const size_t o_sorted = dalign((size_t)b * H * sizeof(int2));
    const size_t o_rank = o_sorted + dalign((size_t)b * n * sizeof(float4));
    const size_t need = o_rank + (fast ? 0 : dalign((size_t)b * n * sizeof(unsigned short)));
    char *base = scratch_for(st, need, 9);
    if (!base) { set_error("ball_query: cannot allocate %zu bytes of grid scratch", need); return PRCNN_ELAUNCH; }
    int2 *range = (int2 *)(base + o_range);
    float4 *sorted = (float4 *)(base + o_sorted);
    unsigned short *rank = (unsigned short *)(base + o_rank);
    const double inv_s = 1.0 / ((double)radius * 1.001);
    const size_t lds = (size_t)H * sizeof(int);
    const void *build = fast ? (n <= 4 * DB ? (const void *)dense_build_reg_kernel<4> : (const void *)dense_build_reg_kernel<16>)
                             : (const void *)dense_build_kernel;
    if (lds > 48 * 1024) {
        const int rc = ensure_dynamic_lds(build, lds, "ball_query(dense build)");
        if (rc != PRCNN_OK) return rc;
    }
    if (!fast)
        hipLaunchKernelGGL(dense_build_kernel, dim3(b), dim3(DB), lds, st, n, H - 1, inv_s, chunk, xyz, range, sorted, rank);
    else if (n <= 4 * DB)
        hipLaunchKernelGGL(dense_build_reg_kernel<4>, dim3(b), dim3(DB), lds, st, n, H - 1, inv_s, xyz, range, sorted);
    else
        hipLaunchKernelGGL(dense_build_reg_kernel<16>, dim3(b), dim3(DB), lds, st, n, H - 1, inv_s, xyz, range, sorted);
    hipLaunchKernelGGL(dense_query_kernel, dim3(ceil_div(m, DQ_WAVES), b), dim3(64 * DQ_WAVES), 0, st, n, m, H - 1, inv_s,
                       radius * radius, nsample, chunk_shift, new_xyz, range, sorted, idx, write_empty);
    *used = 1;
    return check_launch("ball_query(dense)");
}

}  // namespace prcnn
