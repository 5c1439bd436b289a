// fps.hip -- furthest point sampling (K6) for gfx950.
//
// Reference behaviour restated: pointnet2_lib/pointnet2/src/sampling_gpu.cu:86-253 and
// cuda_utils.h:10-13.  The reference runs ONE block of bs = opt_n_threads(n) threads per cloud;
// thread t scans k = t, t+bs, ... keeping its first strict maximum, then a shared-memory tree
// (slot t takes slot t+s only if strictly larger, s = bs/2..1) picks the block winner.  The
// winner among equal maxima is therefore the point minimising the key
//        K(k) = ( bitrev_{log2 bs}(k mod bs) , k div bs )      (lexicographic)
// -- NOT the lowest index.  This file computes the same arg-max with that total order, so the
// launch shape is free: points live in registers (PPT per lane), the per-iteration arg-max is a
// DPP/permute wave reduction followed by one LDS exchange and ONE barrier, and small clouds get
// one wave each (no barrier at all), which is what the 100*B RoI clouds of the RCNN stage need.
#include "common.hpp"
#include <math.h>

namespace prcnn {

// (v, key) beats (bv, bkey): larger value, ties -> smaller key.  Branchless on purpose: the
// short-circuit form compiles to exec-mask branches inside the hot loop.
__device__ __forceinline__ bool better(float v, uint32_t key, float bv, uint32_t bkey)
{
    return (v > bv) | ((v == bv) & (key < bkey));
}

__device__ __forceinline__ void take_if_better(float v, uint32_t key, float &bv, uint32_t &bkey)
{
    const bool t = better(v, key, bv, bkey);
    bv = t ? v : bv;
    bkey = t ? key : bkey;
}

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int x)
{
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ void step_dpp(float &v, uint32_t &key)
{
    const float ov = __int_as_float(dpp_mov<CTRL>(__float_as_int(v)));
    const uint32_t ok = (uint32_t)dpp_mov<CTRL>((int)key);
    take_if_better(ov, ok, v, key);
}

// every lane of each 16-lane row ends up with the row's best (v, key): DPP only, no LDS
__device__ __forceinline__ void row16_argmax(float &v, uint32_t &key)
{
    step_dpp<0xB1>(v, key);   // quad_perm [1,0,3,2]  (lane ^ 1)
    step_dpp<0x4E>(v, key);   // quad_perm [2,3,0,1]  (lane ^ 2)
    step_dpp<0x141>(v, key);  // row_half_mirror: the two quads of an 8-lane group meet
    step_dpp<0x140>(v, key);  // row_mirror: the 8-lane halves of a 16-lane row meet
}

// wave-uniform best of the 64 lanes: rows reduced with DPP, the 4 row results read into SGPRs
__device__ __forceinline__ void wave_argmax(float &v, uint32_t &key)
{
    row16_argmax(v, key);
    float rv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)key, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r));
        const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)key, r);
        take_if_better(ov, ok, rv, rk);
    }
    v = rv;
    key = rk;
}

struct KeyCodec {
    int log2bs;  // virtual block = 1 << log2bs
    int sh;      // bits reserved for k >> log2bs
    __device__ __forceinline__ uint32_t encode(int k) const
    {
        const uint32_t low = (uint32_t)k & ((1u << log2bs) - 1u);
        const uint32_t rev = log2bs ? (__brev(low) >> (32 - log2bs)) : 0u;
        return (rev << sh) | ((uint32_t)k >> log2bs);
    }
    __device__ __forceinline__ int decode(uint32_t key) const
    {
        const uint32_t hi = key >> sh;
        const uint32_t rev = log2bs ? (__brev(hi) >> (32 - log2bs)) : 0u;
        return (int)(((key & ((1u << sh) - 1u)) << log2bs) | rev);
    }
};

// Register-resident FPS: one block (WAVES waves) per cloud, PPT points per lane.
// ORDERED: the virtual block size equals the real one (bs == 64*WAVES), so a lane's points
// k = t + i*T have keys increasing with i and a strict '>' scan already keeps the smallest key.
template <int WAVES, int PPT, bool ORDERED>
__global__ __launch_bounds__(64 * WAVES) void fps_reg_kernel(
    int n, int m, KeyCodec kc, const float *__restrict__ xyz, float *__restrict__ temp,
    int *__restrict__ idx)
{
    constexpr int T = 64 * WAVES;
    __shared__ float s_v[2][16];
    __shared__ uint32_t s_k[2][16];

    const int b = blockIdx.x;
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    float *__restrict__ mind = temp + (long)b * n;
    int *__restrict__ sel = idx + (long)b * m;
    const int t = threadIdx.x;

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
    uint32_t pk[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = t + i * T;
        if (k < n) {
            px[i] = cloud[3 * k]; py[i] = cloud[3 * k + 1]; pz[i] = cloud[3 * k + 2];
            pt[i] = mind[k];
            pk[i] = kc.encode(k);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            pt[i] = -INFINITY;  // can never win: the reference's "best" starts at -1
            pk[i] = 0xffffffffu;
        }
    }
    if (WAVES > 1 && t < 32) {  // unused exchange slots must lose every comparison
        (&s_v[0][0])[t] = -INFINITY;
        (&s_k[0][0])[t] = 0xffffffffu;
    }
    if (WAVES > 1) __syncthreads();

    int old = 0;
    if (t == 0) sel[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float ox = cloud[3 * old], oy = cloud[3 * old + 1], oz = cloud[3 * old + 2];
        float bv = -1.0f;
        uint32_t bkey = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float d = sqdist3(px[i], py[i], pz[i], ox, oy, oz);
            const float d2 = fminf(d, pt[i]);  // min(d, temp[k]) of sampling_gpu.cu:134
            pt[i] = d2;
            if (ORDERED) {
                const bool tk = d2 > bv;
                bv = tk ? d2 : bv;
                bkey = tk ? pk[i] : bkey;
            } else {
                take_if_better(d2, pk[i], bv, bkey);
            }
        }
        wave_argmax(bv, bkey);
        if (WAVES > 1) {
            const int buf = j & 1;
            if ((t & 63) == 0) { s_v[buf][t >> 6] = bv; s_k[buf][t >> 6] = bkey; }
            __syncthreads();
            bv = s_v[buf][t & 15];
            bkey = s_k[buf][t & 15];
            row16_argmax(bv, bkey);
        }
        // no candidate beat the reference's initial (-1, index 0): it would return 0
        old = (bkey == 0xffffffffu) ? 0 : kc.decode(bkey);
        old = __builtin_amdgcn_readfirstlane(old);
        if (t == 0) sel[j] = old;
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = t + i * T;
        if (k < n) mind[k] = pt[i];
    }
}

// Any-n fallback: running minima stay in `temp` (global), one 1024-thread block per cloud.
__global__ __launch_bounds__(1024) void fps_generic_kernel(
    int n, int m, KeyCodec kc, const float *__restrict__ xyz, float *__restrict__ temp,
    int *__restrict__ idx)
{
    constexpr int WAVES = 16;
    __shared__ float s_v[2][WAVES];
    __shared__ uint32_t s_k[2][WAVES];
    const int b = blockIdx.x;
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    float *__restrict__ mind = temp + (long)b * n;
    int *__restrict__ sel = idx + (long)b * m;
    const int t = threadIdx.x;
    int old = 0;
    if (t == 0) sel[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float ox = cloud[3 * old], oy = cloud[3 * old + 1], oz = cloud[3 * old + 2];
        float bv = -1.0f;
        uint32_t bkey = 0xffffffffu;
        for (int k = t; k < n; k += 1024) {
            const float d = sqdist3(cloud[3 * k], cloud[3 * k + 1], cloud[3 * k + 2], ox, oy, oz);
            const float d2 = fminf(d, mind[k]);
            mind[k] = d2;
            take_if_better(d2, kc.encode(k), bv, bkey);
        }
        wave_argmax(bv, bkey);
        const int buf = j & 1;
        if ((t & 63) == 0) { s_v[buf][t >> 6] = bv; s_k[buf][t >> 6] = bkey; }
        __syncthreads();
        bv = s_v[buf][t & 15];
        bkey = s_k[buf][t & 15];
        row16_argmax(bv, bkey);
        old = (bkey == 0xffffffffu) ? 0 : kc.decode(bkey);
        old = __builtin_amdgcn_readfirstlane(old);
        if (t == 0) sel[j] = old;
    }
}

static int host_opt_n_threads(int work_size)
{
    // cuda_utils.h:10-13 (double log ratio truncated, clamped to [1, 1024])
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

template <int WAVES, int PPT>
static void launch_reg(int b, int n, int m, KeyCodec kc, const float *xyz, float *temp, int *idx, hipStream_t st)
{
    if ((1 << kc.log2bs) == 64 * WAVES)
        hipLaunchKernelGGL((fps_reg_kernel<WAVES, PPT, true>), dim3(b), dim3(64 * WAVES), 0, st, n, m, kc, xyz, temp, idx);
    else
        hipLaunchKernelGGL((fps_reg_kernel<WAVES, PPT, false>), dim3(b), dim3(64 * WAVES), 0, st, n, m, kc, xyz, temp, idx);
}

}  // namespace prcnn

using namespace prcnn;

extern "C" int prcnn_opt_n_threads(int work_size) { return work_size > 0 ? host_opt_n_threads(work_size) : 1; }

extern "C" int prcnn_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                             int *idx, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "fps: bad sizes b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(n > 0, "fps: empty cloud with m=%d", m);
    PRCNN_REQUIRE(xyz && temp && idx, "fps: null pointer");
    hipStream_t st = (hipStream_t)stream;

    const int bs = host_opt_n_threads(n);
    KeyCodec kc;
    kc.log2bs = 0;
    while ((1 << kc.log2bs) < bs) ++kc.log2bs;
    const int nq = (n + bs - 1) / bs;  // values of k div bs: 0 .. nq-1
    kc.sh = 0;
    while ((1 << kc.sh) < nq) ++kc.sh;
    PRCNN_REQUIRE(kc.sh + kc.log2bs <= 31, "fps: n=%d too large for the 32-bit tie key", n);

    if (n <= 128) launch_reg<1, 2>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 256) launch_reg<1, 4>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 512) launch_reg<1, 8>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 1024 && b >= 128) launch_reg<1, 16>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 1024) launch_reg<4, 4>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 4096) launch_reg<16, 4>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 8192) launch_reg<16, 8>(b, n, m, kc, xyz, temp, idx, st);
    else if (n <= 16384) launch_reg<16, 16>(b, n, m, kc, xyz, temp, idx, st);
    else hipLaunchKernelGGL(fps_generic_kernel, dim3(b), dim3(1024), 0, st, n, m, kc, xyz, temp, idx);
    return check_launch("furthest_point_sampling");
}
