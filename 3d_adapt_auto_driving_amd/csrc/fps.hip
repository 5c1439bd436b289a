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
#include <stdlib.h>
#include <map>
#include <mutex>

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

// ---- two-pass arg-max: first the maximum VALUE (one v_max_f32 per candidate, one per DPP step), then the smallest key
// among the candidates that hold it (compare + select + v_min_u32).  Same total order as take_if_better -- larger value
// first, ties to the smaller key -- with a third of the instructions and much shorter dependency chains.
// The DPP permutation rides ON the max / min instruction (v_max_f32_dpp): one instruction per butterfly step.  Written as
// update_dpp + fmaxf the compiler emits v_mov, v_mov_dpp, a canonicalising v_max and the v_max -- 20 instructions for the four
// steps of a reduction that sits on the critical path of every FPS iteration.  (s_nop 1: a DPP operand written by the
// previous VALU instruction needs two wait states; the hazard recogniser does not look into inline assembly.)
#define PRCNN_DPP_OP(OP, TY, CTRL_TEXT)                                                                                       \
    asm("s_nop 1\n\t" OP " %0, %1, %1 " CTRL_TEXT " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v))
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v)
{
    float r;
    if constexpr (CTRL == 0xB1) PRCNN_DPP_OP("v_max_f32_dpp", float, "quad_perm:[1,0,3,2]");
    else if constexpr (CTRL == 0x4E) PRCNN_DPP_OP("v_max_f32_dpp", float, "quad_perm:[2,3,0,1]");
    else if constexpr (CTRL == 0x141) PRCNN_DPP_OP("v_max_f32_dpp", float, "row_half_mirror");
    else if constexpr (CTRL == 0x140) PRCNN_DPP_OP("v_max_f32_dpp", float, "row_mirror");
    else r = fmax_raw(v, __int_as_float(dpp_mov<CTRL>(__float_as_int(v))));
    return r;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_min(uint32_t v)
{
    uint32_t r;
    if constexpr (CTRL == 0xB1) PRCNN_DPP_OP("v_min_u32_dpp", uint32_t, "quad_perm:[1,0,3,2]");
    else if constexpr (CTRL == 0x4E) PRCNN_DPP_OP("v_min_u32_dpp", uint32_t, "quad_perm:[2,3,0,1]");
    else if constexpr (CTRL == 0x141) PRCNN_DPP_OP("v_min_u32_dpp", uint32_t, "row_half_mirror");
    else if constexpr (CTRL == 0x140) PRCNN_DPP_OP("v_min_u32_dpp", uint32_t, "row_mirror");
    else {
        const uint32_t o = (uint32_t)dpp_mov<CTRL>((int)v);
        r = o < v ? o : v;
    }
    return r;
}
__device__ __forceinline__ float wave_max_f32(float v)        // wave-uniform result
{
    v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v); v = dpp_max<0x141>(v); v = dpp_max<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)  // wave-uniform result
{
    v = dpp_min<0xB1>(v); v = dpp_min<0x4E>(v); v = dpp_min<0x141>(v); v = dpp_min<0x140>(v);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}
// (value, key) of a wave as ONE unsigned 64-bit word whose integer order is the arg-max order: the value's bits made
// monotone (negative floats flipped, positive ones offset), the key complemented so that the smaller key is the larger
// word.  The workgroup's winner is then a single LDS atomic max per wave instead of an exchange + a 16-entry reduction.
__device__ __forceinline__ unsigned long long pack_candidate(float v, uint32_t key)
{
    const uint32_t b = (uint32_t)__float_as_int(v);
    const uint32_t mono = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)mono << 32) | (unsigned long long)(~key);
}
__device__ __forceinline__ void unpack_candidate(unsigned long long w, float &v, uint32_t &key)
{
    const uint32_t mono = (uint32_t)(w >> 32);
    const uint32_t b = (mono & 0x80000000u) ? (mono & 0x7fffffffu) : ~mono;
    v = __int_as_float((int)b);
    key = ~(uint32_t)w;
}

// The squared distance of sampling_gpu.cu:133, `(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)`.
// HIPCC = false: the source's arithmetic, one rounding per operation (the parity contract).
// HIPCC = true (prcnn_set_fps_arithmetic(1), round 4): the arithmetic of the reference's KERNEL BINARY as hipcc 7.2 builds that file
// for gfx950 with its default contraction -- read off the disassembly of oracle/_ref/pointnet2_kernels_ref.so, the same in all
// eleven block-size instantiations: v_pk_mul (dx^2, dz^2), v_fma dy*dy + dx^2, v_add + dz^2, i.e. (fma(dy, dy, dx*dx)) + dz*dz.
// With it the picks equal the reference kernel's on every cloud of tests/test_gpu_reference_kernels.py, near-ties included.
template <bool HIPCC>
__device__ __forceinline__ float fps_dist(float px, float py, float pz, float ox, float oy, float oz)
{
    if (!HIPCC) return sqdist3(px, py, pz, ox, oy, oz);
    const float dx = px - ox, dy = py - oy, dz = pz - oz;
    return __fadd_rn(__fmaf_rn(dy, dy, __fmul_rn(dx, dx)), __fmul_rn(dz, dz));
}

struct KeyCodec {
    int log2bs;  // virtual block = 1 << log2bs
    int sh;      // bits reserved for k >> log2bs
    int hipcc;   // 1: distances as the reference's hipcc-built binary computes them (fps_dist<true>); wave-uniform
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
    int *__restrict__ idx, float *__restrict__ new_xyz = nullptr)
{
    // temp == NULL: the running minima start at the reference caller's fill value (1e10, pointnet2_utils.py:26) and are not
    // handed back; new_xyz != NULL: the selected points' coordinates are written as well (prcnn_fps_new_xyz: the caller's
    // fill + index cast + gather launches folded into this one)
    constexpr int T = 64 * WAVES;
    __shared__ unsigned long long s_best[3];
    const int b = blockIdx.x;
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    float *__restrict__ mind = temp ? temp + (long)b * n : nullptr;
    int *__restrict__ sel = idx + (long)b * m;
    float *__restrict__ nxyz = new_xyz ? new_xyz + (long)b * m * 3 : nullptr;
    const int t = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);   // latency-bound dependent chain: win issue arbitration (see fps_spec_kernel)

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
    uint32_t pk[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = t + i * T;
        if (k < n) {
            px[i] = cloud[3 * k]; py[i] = cloud[3 * k + 1]; pz[i] = cloud[3 * k + 2];
            pt[i] = mind ? mind[k] : 1e10f;
            pk[i] = kc.encode(k);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            pt[i] = -INFINITY;  // can never win: the reference's "best" starts at -1
            pk[i] = 0xffffffffu;
        }
    }
    if (WAVES > 1 && t < 3) s_best[t] = 0ull;
    if (WAVES > 1) __syncthreads();

    int old = 0;
    if (t == 0) sel[0] = 0;
    for (int j = 1; j < m; ++j) {
        float ox, oy, oz;
        if constexpr (WAVES == 1) {
            // one wave per cloud (the 800 RoI clouds of a batch): the pivot's coordinates come out of the REGISTERS that hold the
            // cloud (point k = lane + 64 i) instead of a dependent scalar load per iteration -- the memory round trip was most
            // of an iteration here (round 3: 138 -> ~50 us for 800 x (512 -> 128))
            const int pl = old & 63, pi = old >> 6;
            ox = oy = oz = 0.f;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float vx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[i]), pl));
                const float vy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[i]), pl));
                const float vz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[i]), pl));
                if (i == pi) { ox = vx; oy = vy; oz = vz; }
            }
        } else {
            ox = cloud[3 * old]; oy = cloud[3 * old + 1]; oz = cloud[3 * old + 2];
        }
        if (nxyz && t == 0) { nxyz[3 * (j - 1)] = ox; nxyz[3 * (j - 1) + 1] = oy; nxyz[3 * (j - 1) + 2] = oz; }
        float lv = -INFINITY;
        if (kc.hipcc) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float d2 = fminf(fps_dist<true>(px[i], py[i], pz[i], ox, oy, oz), pt[i]);
                pt[i] = d2;
                lv = fmaxf(lv, d2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float d = sqdist3(px[i], py[i], pz[i], ox, oy, oz);
                const float d2 = fminf(d, pt[i]);  // min(d, temp[k]) of sampling_gpu.cu:134
                pt[i] = d2;
                lv = fmaxf(lv, d2);
            }
        }
        float bv = wave_max_f32(lv);
        uint32_t lk = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const uint32_t c = pt[i] == bv ? pk[i] : 0xffffffffu;
            lk = c < lk ? c : lk;
        }
        uint32_t bkey = wave_min_u32(lk);
        if (WAVES > 1) {
            const int buf = j % 3;
            if ((t & 63) == 0) atomicMax(&s_best[buf], pack_candidate(bv, bkey));
            lds_barrier();     // orders LDS only: __syncthreads() would also drain wave 0's store of sel[j-1] every iteration
            const unsigned long long best = s_best[buf];
            if (t == 0) s_best[(j + 2) % 3] = 0ull;     // the slot of iteration j-1: every wave has read it (barrier j), next used at j+2
            unpack_candidate(best, bv, bkey);
        }
        // no candidate beat the reference's initial (-1, index 0): it would return 0
        old = (bkey == 0xffffffffu || !(bv > -1.0f)) ? 0 : kc.decode(bkey);
        old = __builtin_amdgcn_readfirstlane(old);
        if (t == 0) sel[j] = old;
        if (bv == 0.f) {
            // Every running minimum is 0: all distinct points have been picked, what is left are exact copies (a RoI that holds
            // fewer points than it is sampled to, roipool3d_kernel.cu:152-159).  The arg-max over equal values is the point
            // with the lowest tie key, point 0 -- in this and in every later iteration (min(d, 0) stays 0): the remaining
            // picks are 0 without computing them.  (`old` is already 0 here by the same rule.)
            for (int jj = j + 1 + t; jj < m; jj += T) {
                sel[jj] = 0;
                if (nxyz) { nxyz[3 * (jj - 1)] = cloud[0]; nxyz[3 * (jj - 1) + 1] = cloud[1]; nxyz[3 * (jj - 1) + 2] = cloud[2]; }
            }
            if (nxyz && t == 0) { nxyz[3 * j] = cloud[0]; nxyz[3 * j + 1] = cloud[1]; nxyz[3 * j + 2] = cloud[2]; }   // slot j is written by iteration j + 1 otherwise
            break;
        }
    }
    if (nxyz && t == 0 && m > 0) { nxyz[3 * (m - 1)] = cloud[3 * old]; nxyz[3 * (m - 1) + 1] = cloud[3 * old + 1]; nxyz[3 * (m - 1) + 2] = cloud[3 * old + 2]; }
    if (mind) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int k = t + i * T;
            if (k < n) mind[k] = pt[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Spatially pruned FPS (exact).  A new pivot can lower the running minimum of point k only if
// d(k, pivot)^2 < temp[k] <= V, where V = max_k temp[k] is the value of the pivot just selected.
// So only points inside the ball of radius sqrt(V) around the pivot need a distance evaluation, and
// V shrinks as sampling proceeds (about (area / j) after j picks).  The cloud is first ordered along a
// Morton curve over (x, z) (fps_order_kernel: one workgroup per cloud, LDS counting sort), so that each
// group of 64 consecutive points -- one register slot across the lanes of a wave, a "tile" -- is
// spatially compact.  Per iteration lane i < PPT of every wave tests the pivot against the bounding box
// of tile i (a lower bound of every distance in the tile, with a 1e-5 relative safety margin that
// covers f32 rounding of both sides); tiles that cannot change are skipped, and a wave none of whose
// tiles changed re-submits its cached best.  Selected indices, tie rule and running minima are
// bit-identical to the full scan: skipped points satisfy min(d, temp) == temp.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned part1by1(unsigned v)   // spread the low 16 bits to even positions
{
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// perm[b][s] = original index of the s-th point in Morton order of a 64 x 64 grid over the cloud's
// (x, z) bounding box.  Any permutation is valid for correctness; this one makes tiles compact.
__global__ __launch_bounds__(1024) void fps_order_kernel(int n, const float *__restrict__ xyz, int *__restrict__ perm)
{
    __shared__ int cnt[4096];
    __shared__ float red[4][16];
    __shared__ int wsum[16];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    float x0 = INFINITY, x1 = -INFINITY, z0 = INFINITY, z1 = -INFINITY;
    for (int k = t; k < n; k += 1024) {
        const float x = cloud[3 * k], z = cloud[3 * k + 2];
        x0 = fminf(x0, x); x1 = fmaxf(x1, x); z0 = fminf(z0, z); z1 = fmaxf(z1, z);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d, 64)); x1 = fmaxf(x1, __shfl_xor(x1, d, 64));
        z0 = fminf(z0, __shfl_xor(z0, d, 64)); z1 = fmaxf(z1, __shfl_xor(z1, d, 64));
    }
    if ((t & 63) == 0) { red[0][t >> 6] = x0; red[1][t >> 6] = x1; red[2][t >> 6] = z0; red[3][t >> 6] = z1; }
    for (int i = t; i < 4096; i += 1024) cnt[i] = 0;
    __syncthreads();
    x0 = red[0][0]; x1 = red[1][0]; z0 = red[2][0]; z1 = red[3][0];
    for (int w = 1; w < 16; ++w) {
        x0 = fminf(x0, red[0][w]); x1 = fmaxf(x1, red[1][w]); z0 = fminf(z0, red[2][w]); z1 = fmaxf(z1, red[3][w]);
    }
    const float sx = x1 > x0 ? 64.f / (x1 - x0) : 0.f, sz = z1 > z0 ? 64.f / (z1 - z0) : 0.f;
    auto code_of = [&](int k) {
        float fx = (cloud[3 * k] - x0) * sx, fz = (cloud[3 * k + 2] - z0) * sz;
        fx = fx == fx ? fx : 0.f; fz = fz == fz ? fz : 0.f;            // NaN-safe
        const int cx = min(63, max(0, (int)fx)), cz = min(63, max(0, (int)fz));
        return (int)(part1by1((unsigned)cx) | (part1by1((unsigned)cz) << 1));
    };
    for (int k = t; k < n; k += 1024) atomicAdd(&cnt[code_of(k)], 1);
    __syncthreads();
    // exclusive scan of the 4096 counters: 4 per thread
    int local = 0, c4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { c4[i] = cnt[4 * t + i]; local += c4[i]; }
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if ((t & 63) >= d) incl += o;
    }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < (t >> 6); ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) { cnt[4 * t + i] = run; run += c4[i]; }
    __syncthreads();
    for (int k = t; k < n; k += 1024) perm[(long)b * n + atomicAdd(&cnt[code_of(k)], 1)] = k;
}

// (fps_pruned_kernel, the one-pick-per-exchange kernel of round 3 -- 4.07 ms for 32 x (16384 -> 4096) against the speculative kernel's
// 1.69 -- was removed in round 6 together with its switch PRCNN_FPS_SEQUENTIAL; the pruning above lives on in fps_spec_kernel.)

// ---------------------------------------------------------------------------------------------
// SPECULATIVE multi-pick FPS (round 4) -- the same picks as the sequential scan, several per exchange.
//
// Round 3's fps_pruned_kernel paid one workgroup-wide exchange (LDS atomic, barrier, LDS read, scalar load of the pivot) per PICK:
// 0.99 us x 4095 picks = 4.07 ms for 16384 -> 4096, the longest launch of the step and the limiter on LiDAR-shaped scenes.
// An exchange can decide MORE than one pick.  Let the points be ordered by (running minimum desc, tie key asc) and let
// g0, g1, ... be the head of that order.  g0 is the next pick.  If adding g0 leaves g1's minimum unchanged
// (d(g1, g0) >= min(g1)) then g1 is the pick after it -- every other minimum can only have decreased -- and so on: g_q
// follows as long as it is unchanged by ALL of g0 .. g_{q-1}.  Once the sampled points are dense these heads are the centres
// of separate holes and rarely interact, so a round decides several picks and the distance updates of all of them run
// behind ONE barrier.
//
// What a wave publishes when its entries changed (16 waves, one LDS table; two barriers per round: table complete / verdict in):
//   E1 = its best point (value, key, coordinates); E2 = the best point of its OTHER lanes; and a bound B = the largest value
//   any of its unpublished points can have (third among the lanes' bests, and the second bests of the two publishing lanes).
// ONE wave then merges the 32 published entries: take the best remaining entry g;
// accept it if it is the first one, or if (a) its value is STRICTLY above max_w B_w -- then every unpublished point of the
// cloud comes after it in the order -- and (b) none of the pivots accepted in this round is closer to it than its minimum
// (d < min would change it).  The first entry that fails ends the round.  Accepted pivots update the registers (tile boxes
// prune as before, against the wave's own maximum), touched waves rebuild their entries, untouched ones re-publish.
// Same picks, same tie rule, same running minima as the sequential scan, bit for bit (tests/test_gpu_ops.py,
// tests/test_gpu_reference_kernels.py: lattices and duplicate clouds with thousands of exact ties included).
// ---------------------------------------------------------------------------------------------
constexpr int FS_RMAX = 16;     // picks per round at most

// coordinates of point (lane l, slot) -- both wave-uniform -- as three v_readlane behind a scalar compare ladder.  (Indexing the
// register arrays with the uniform slot made the compiler keep scratch copies of px / py / pz once the kernel was at its 128-VGPR
// limit; a per-lane select chain would cost 3 x PPT VALU instructions per rebuild.)
template <int PPT, int I = 0>
__device__ __forceinline__ void fs_pick3(const float (&px)[PPT], const float (&py)[PPT], const float (&pz)[PPT], int slot, int l,
                                         float &x, float &y, float &z)
{
    if (slot == I) {
        x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[I]), l));
        y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[I]), l));
        z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[I]), l));
    } else if constexpr (I + 1 < PPT) {
        fs_pick3<PPT, I + 1>(px, py, pz, slot, l, x, y, z);
    }
}

template <int PPT>
__global__ __launch_bounds__(1024) void fps_spec_kernel(
    int n, int m, KeyCodec kc, const float *__restrict__ xyz, const int *__restrict__ perm,
    float *__restrict__ temp, int *__restrict__ idx, float *__restrict__ new_xyz)
{
    // temp == NULL: the running minima start at the reference caller's fill value (1e10, pointnet2_utils.py:26) and are not handed back;
    // new_xyz != NULL: the coordinates of the picks are written as well (prcnn_fps_new_xyz: the caller's gather launch folded in)
    constexpr int SB = PPT == 16 ? 4 : (PPT == 8 ? 3 : 2);          // slot bits below the tie key
    __shared__ unsigned long long s_vk[32];                          // packed (value, key) of the published entries
    __shared__ float s_xyz[32][3];
    __shared__ float s_bound[16];
    __shared__ int s_rank[32], s_blk[32];                            // per entry: entries that precede it / one of them blocks it
    const int b = blockIdx.x;
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    const int *__restrict__ order = perm + (long)b * n;
    float *__restrict__ mind = temp ? temp + (long)b * n : nullptr;
    int *__restrict__ sel = idx + (long)b * m;
    float *__restrict__ nxyz = new_xyz ? new_xyz + (long)b * m * 3 : nullptr;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    __builtin_amdgcn_s_setprio(3);

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
    uint32_t pc[PPT];                                                 // (tie key << SB) | slot; ~0 for a slot beyond the cloud
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        // position in Morton order: a wave = 1024 consecutive positions.  (Round 6, measured and not kept: tiles or tile PAIRS dealt
        // round-robin to the waves so that a pivot's neighbourhood is spread over them -- with the lazy rebuild of round 5 in place:
        // pairs 1.91 / 2.32 ms, single tiles 2.04 / 2.41 ms against 1.69 / 2.46 ms, uniform / LiDAR-shaped, same picks.)
        const int s = w * (64 * PPT) + i * 64 + lane;
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY, z0 = INFINITY, z1 = -INFINITY;
        if (s < n) {
            const int k = order[s];
            px[i] = cloud[3 * k]; py[i] = cloud[3 * k + 1]; pz[i] = cloud[3 * k + 2];
            pt[i] = mind ? mind[k] : 1e10f;
            pc[i] = (kc.encode(k) << SB) | (uint32_t)i;
            x0 = x1 = px[i]; y0 = y1 = py[i]; z0 = z1 = pz[i];
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            pt[i] = -INFINITY;
            pc[i] = 0xffffffffu;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            x0 = fmin_raw(x0, __shfl_xor(x0, d, 64)); x1 = fmax_raw(x1, __shfl_xor(x1, d, 64));
            y0 = fmin_raw(y0, __shfl_xor(y0, d, 64)); y1 = fmax_raw(y1, __shfl_xor(y1, d, 64));
            z0 = fmin_raw(z0, __shfl_xor(z0, d, 64)); z1 = fmax_raw(z1, __shfl_xor(z1, d, 64));
        }
        if (lane == i) { bx0 = x0; bx1 = x1; by0 = y0; by1 = y1; bz0 = z0; bz1 = z1; }
    }

    // every group of PPT lanes holds the PPT tile boxes (lane -> tile lane % PPT): 64 / PPT pivots are box-tested per pass
    constexpr int GP = 64 / PPT;
    bx0 = __shfl(bx0, lane & (PPT - 1), 64); bx1 = __shfl(bx1, lane & (PPT - 1), 64);
    by0 = __shfl(by0, lane & (PPT - 1), 64); by1 = __shfl(by1, lane & (PPT - 1), 64);
    bz0 = __shfl(bz0, lane & (PPT - 1), 64); bz1 = __shfl(bz1, lane & (PPT - 1), 64);
    // pivots against the registers: tile boxes prune against `bound`, an upper bound of every running minimum of this wave
    // `touched`: one of the tiles that hold this wave's two PUBLISHED points went through an update -- only then are its entries rebuilt
    // (round 5).  A pivot that changes other points of the wave leaves the published points, their values and keys as they are, and
    // the published bound B (and `bound` below) stays an upper bound of everything unpublished, since running minima only fall: the
    // merge stays exact with the stale bound, it may only accept fewer entries.  Measured on the round model (profiles/fps_round_sim.py
    // lazy): 7.7 instead of 10.1 rebuilds per round on the uniform scene, 5.1 instead of 6.9 on the LiDAR-shaped one, the rounds
    // themselves 481 / 742 against 480 / 742.
    unsigned long long touched = 0ull, etiles = ~0ull;
    float bound = INFINITY;
    // which tiles can a pivot (per lane group: ox, oy, oz differ by group) change?  bit g * PPT + i: tile i, the group's pivot
    auto box_mask = [&](float ox, float oy, float oz, bool live) __attribute__((always_inline)) {
        const float dx = fmax_raw(fmax_raw(bx0 - ox, ox - bx1), 0.f);
        const float dy = fmax_raw(fmax_raw(by0 - oy, oy - by1), 0.f);
        const float dz = fmax_raw(fmax_raw(bz0 - oz, oz - bz1), 0.f);
        const float lb = dx * dx + dy * dy + dz * dz;
        return __ballot(live && !(lb * 0.99999f >= bound));           // empty boxes give lb = +inf
    };
    auto update = [&](float ox, float oy, float oz, unsigned long long mask) __attribute__((always_inline)) {
        if (mask != 0ull) {
            touched |= mask & etiles;
            // Tiles in PAIRS on packed f32 arithmetic (round 5): v_pk_add / v_pk_mul / v_pk_fma apply to each half exactly the operation
            // of the scalar form (one rounding each), so a tile's new minima are the same bits; a pair is updated when EITHER tile passed
            // the box test -- for the other one the update is the identity (no point of a pruned tile can come closer than its minimum:
            // that is what the pruning rests on).  8 packed + 2 min instructions per pair instead of 9 per tile.
            const pk_f32x2 o_x = {ox, ox}, o_y = {oy, oy}, o_z = {oz, oz};
            if (kc.hipcc) {
#pragma unroll
                for (int i = 0; i < PPT; i += 2)
                    if ((mask >> i) & 3ull) {
                        const pk_f32x2 dx = (pk_f32x2){px[i], px[i + 1]} - o_x, dy = (pk_f32x2){py[i], py[i + 1]} - o_y,
                                       dz = (pk_f32x2){pz[i], pz[i + 1]} - o_z;
                        const pk_f32x2 d = __builtin_elementwise_fma(dy, dy, dx * dx) + dz * dz;
                        pt[i] = fmin_raw(d.x, pt[i]); pt[i + 1] = fmin_raw(d.y, pt[i + 1]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < PPT; i += 2)
                    if ((mask >> i) & 3ull) {
                        const pk_f32x2 dx = (pk_f32x2){px[i], px[i + 1]} - o_x, dy = (pk_f32x2){py[i], py[i + 1]} - o_y,
                                       dz = (pk_f32x2){pz[i], pz[i + 1]} - o_z;
                        const pk_f32x2 d = (dx * dx + dy * dy) + dz * dz;
                        pt[i] = fmin_raw(d.x, pt[i]); pt[i + 1] = fmin_raw(d.y, pt[i + 1]);
                    }
            }
        }
    };
    if (t == 0) {
        sel[0] = 0;
        if (nxyz) { nxyz[0] = cloud[0]; nxyz[1] = cloud[1]; nxyz[2] = cloud[2]; }
    }
    if (t < 32) { s_rank[t] = 0; s_blk[t] = 0; }                      // (the first barrier A orders this before any atomic)
    int j = 1;                                                        // picks made so far
    if (m > 1) {                                                      // the given start point, index 0
        const float ox = cloud[0], oy = cloud[1], oz = cloud[2];
        update(ox, oy, oz, box_mask(ox, oy, oz, lane < PPT));
    }
    touched = ~0ull;                                                  // the entries have to be built
    while (j < m) {
        // ---- 1. this wave's entries: rebuilt when one of its tiles changed
        if (touched != 0ull) {
            float bv = -INFINITY, sv = -INFINITY;                     // best and second-best VALUE of this lane
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                sv = __builtin_amdgcn_fmed3f(bv, sv, pt[i]);          // = max(sv, min(bv, pt)) while sv <= bv: one v_med3_f32
                bv = fmax_raw(bv, pt[i]);
            }
            uint32_t lk = 0xffffffffu;                                // (key, slot) of this lane's best
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const uint32_t c = pt[i] == bv ? pc[i] : 0xffffffffu;
                lk = c < lk ? c : lk;
            }
            // best lane, best of the other lanes, third value
            const float v1 = wave_max_f32(bv);
            const uint32_t c1 = wave_min_u32(bv == v1 ? lk : 0xffffffffu);
            const unsigned long long m1 = __ballot(bv == v1 && lk == c1);
            const int l1 = m1 ? (int)__builtin_ctzll(m1) : 0;
            const float bv2 = lane == l1 ? -INFINITY : bv;
            const float v2 = wave_max_f32(bv2);
            const uint32_t c2 = wave_min_u32(bv2 == v2 ? lk : 0xffffffffu);
            const unsigned long long m2 = __ballot(lane != l1 && bv2 == v2 && lk == c2);
            const int l2 = m2 ? (int)__builtin_ctzll(m2) : l1;
            const float v3 = wave_max_f32((lane == l1 || lane == l2) ? -INFINITY : bv);
            const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), l1));
            const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), l2));
            const float wB = fmax_raw(v3, fmax_raw(s1, s2));
            bound = v1;
            const int sl1 = (int)(c1 & ((1u << SB) - 1u)), sl2 = (int)(c2 & ((1u << SB) - 1u));
            const bool has2 = m2 != 0ull && c2 != 0xffffffffu;
            etiles = (1ull << sl1) | (has2 ? (1ull << sl2) : 0ull);
            // ---- 2. publish (only a wave whose entries changed writes: the table keeps the others'): lane 0 the best point, lane 1
            // the best point of the other lanes
            float x1 = 0.f, y1 = 0.f, z1 = 0.f, x2 = 0.f, y2 = 0.f, z2 = 0.f;
            fs_pick3<PPT>(px, py, pz, __builtin_amdgcn_readfirstlane(sl1), l1, x1, y1, z1);
            fs_pick3<PPT>(px, py, pz, __builtin_amdgcn_readfirstlane(sl2), l2, x2, y2, z2);
            const float ex = lane == 0 ? x1 : x2, ey = lane == 0 ? y1 : y2, ez = lane == 0 ? z1 : z2;
            if (lane < 2) {
                const float ev = lane == 0 ? v1 : (has2 ? v2 : -INFINITY);
                const uint32_t ek = lane == 0 ? (c1 == 0xffffffffu ? 0xffffffffu : (c1 >> SB)) : (has2 ? (c2 >> SB) : 0xffffffffu);
                s_vk[2 * w + lane] = pack_candidate(ev, ek);
                s_xyz[2 * w + lane][0] = ex; s_xyz[2 * w + lane][1] = ey; s_xyz[2 * w + lane][2] = ez;
                if (lane == 0) s_bound[w] = wB;
            }
        }
        lds_barrier();                                                // A: the table is complete
        // ---- 3. merge (round 6 form: two barriers per round, no LDS atomic, no verdict broadcast).  All 32 x 32 pairs at once over the
        // 16 waves: wave w OWNS entries 2 w and 2 w + 1 -- lane (e = lane & 31, h = lane >> 5) compares entry i = 2 w + h with entry e:
        // does e precede i in the order (value desc, key asc), and does it lie closer to i than i's running minimum?  Entry i's RANK
        // (how many precede it) and its BLOCKED bit are a population count / a test of one half of a ballot: wave-uniform, written by
        // one lane.  Behind the second barrier EVERY wave reads the 32 ranks and derives the verdict itself -- the picks of the round
        // are the ranks 0 .. r-1 up to the first rank that fails -- so nobody waits for wave 0 and a third barrier; a forward
        // permute (ds_permute: lane e sends to lane rank(e)) puts pivot q's coordinates into lane q of every wave.
        // (Rounds 4-5: lane = entry i, wave = entry e, ranks summed through LDS atomics, the verdict by wave 0 into s_res, barrier B:
        // pairs 540 + verdict 760 + the barrier ~ 1500 of a round's 8200 cycles, profiles/r05_fps_round_stamps.txt.  History before that:
        // every wave merging the whole table by sequential extraction -- issue-bound, 5.2 ms; one wave walking its half of the table,
        // 16 dependent LDS round trips: 2.6 k of 10.2 k cycles.)
        const int left = m - j;
        const int e = lane & 31, h = lane >> 5;
        const unsigned long long pke = s_vk[e];
        float cv; uint32_t ck;                                        // entry e: what this lane is responsible for in the verdict
        unpack_candidate(pke, cv, ck);
        const float cx = s_xyz[e][0], cy = s_xyz[e][1], cz = s_xyz[e][2];
        const float gB = wave_max_f32(lane < 16 ? s_bound[lane] : -INFINITY);     // (read in front of A2: the next publish rewrites it behind A2)
        {
            const int i = 2 * w + h;
            const unsigned long long pki = s_vk[i];
            float iv; uint32_t ik;
            unpack_candidate(pki, iv, ik);
            const float ix = s_xyz[i][0], iy = s_xyz[i][1], iz = s_xyz[i][2];
            const bool before = pke > pki;
            const float d = kc.hipcc ? fps_dist<true>(ix, iy, iz, cx, cy, cz) : sqdist3(ix, iy, iz, cx, cy, cz);
            const unsigned long long bb = __ballot(before), kb = __ballot(before && (d < iv));
            if (e == 0) {
                s_rank[i] = __popc((unsigned)(h ? (bb >> 32) : bb));
                s_blk[i] = (unsigned)(h ? (kb >> 32) : kb) != 0u;
            }
        }
        lds_barrier();                                                // A2: every entry has its rank
        const int rank = s_rank[e];
        const bool blocked = s_blk[e] != 0;
        const bool valid = !(ck == 0xffffffffu || !(cv > -1.0f));      // (reference: best starts at -1, besti at 0)
        const bool pass = rank == 0 ? true : ((cv > gB) && (cv > 0.f) && valid && !blocked);
        // a round ends behind an invalid first entry (the reference then picks index 0: every running minimum is below -1 / NaN)
        const bool stop = !pass || (rank == 0 && !valid);
        int r = (int)wave_min_u32((stop && !(rank == 0)) ? (uint32_t)rank : ((rank == 0 && !valid) ? 1u : 32u));
        r = min(min(r, left), FS_RMAX);
        if (r < 1) r = 1;
        const float ox_ = valid ? cx : cloud[0], oy_ = valid ? cy : cloud[1], oz_ = valid ? cz : cloud[2];
        if (w == 0 && h == 0 && rank < r) {                           // the outputs: one wave writes them
            sel[j + rank] = valid ? kc.decode(ck) : 0;
            if (nxyz) { nxyz[3 * (j + rank)] = ox_; nxyz[3 * (j + rank) + 1] = oy_; nxyz[3 * (j + rank) + 2] = oz_; }
        }
        // pivot q -> lanes 3 q, 3 q + 1, 3 q + 2 of ONE register (x, y, z): lane e (first half) sends its three coordinates to the lanes
        // of its rank; whatever is no pivot goes to lane 63, which nobody reads; lanes nobody writes read 0, so the three answers OR
        // together.  (Three registers, lane q = pivot q, spilled fps_spec_kernel<16> at its 128-register limit.)
        const int dst = (h == 0 && rank < FS_RMAX) ? 3 * rank : 63;
        const float rv = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(ox_)) |
                                        __builtin_amdgcn_ds_permute(min(dst + 1, 63) << 2, __float_as_int(oy_)) |
                                        __builtin_amdgcn_ds_permute(min(dst + 2, 63) << 2, __float_as_int(oz_)));
        // ---- 4. running minima against this round's pivots (the last pick of the whole run does not update them: sampling_gpu.cu)
        j += r;
        const int apply_n = j >= m ? r - 1 : r;
        touched = 0ull;
        for (int q0 = 0; q0 < apply_n; q0 += GP) {
            // lane group g tests pivot q0 + g against the PPT tile boxes
            const int qg = q0 + lane / PPT;
            const int src = 3 * min(qg, FS_RMAX - 1);
            const float gx = __shfl(rv, src, 64), gy = __shfl(rv, src + 1, 64), gz = __shfl(rv, src + 2, 64);
            const unsigned long long masks = box_mask(gx, gy, gz, qg < apply_n);
            if (masks == 0ull) continue;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const unsigned long long mk = (masks >> (g * PPT)) & ((PPT == 64) ? ~0ull : ((1ull << PPT) - 1ull));
                if (mk != 0ull) {
                    const int q = q0 + g;
                    const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 3 * q));
                    const float oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 3 * q + 1));
                    const float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 3 * q + 2));
                    update(ox, oy, oz, mk);
                }
            }
        }
    }
    if (mind) {
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (pc[i] != 0xffffffffu) mind[kc.decode(pc[i] >> SB)] = pt[i];
    }
}

// ---------------------------------------------------------------------------------------------
// The speculative kernel for 16384 < n <= 32768 (round 5: tools/cfgs/double.yaml, NUM_POINTS 32768): TWO workgroups per cloud.
//
// 32768 points x (x, y, z, running minimum) do not fit one compute unit's registers (a 1024-thread workgroup has 128 VGPRs per thread:
// 16 points each), and streaming half of them from L2 for every pick is what fps_generic_kernel does (~3.5 us per pick).  Here the
// cloud in Morton order is cut in two and each half lives in the registers of a workgroup of its own, exactly as in fps_spec_kernel<16>;
// the two workgroups run the SAME rounds in lockstep:
//   * every wave publishes its two entries + its bound into the round's table in GLOBAL memory (64 entries, two tables alternating:
//     a table is rewritten two rounds later, when both workgroups are provably past reading it);
//   * one cross-workgroup barrier per round (a monotonic counter per cloud: agent-scope release add, acquire spin);
//   * each workgroup copies the 64 entries into its LDS and computes the verdict over all 64 x 64 pairs REDUNDANTLY -- the merge is
//     deterministic, both arrive at the same picks, no second exchange is needed; workgroup 0 writes the outputs;
//   * each updates the running minima of its own half against the round's pivots.
// Same picks, tie rule and running minima as the sequential scan (tests/test_gpu_ops.py: oracle-exact, lattices and duplicates).
// Placement: the halves of a cloud are blocks q and q + 8 -- dispatched to the same XCD, one right after the other, so a half never
// waits for a partner that cannot be scheduled behind workgroups that themselves wait (see fps_any).
// ---------------------------------------------------------------------------------------------
struct Fps2Table {                                                   // per cloud, in the library's scratch (zeroed `sync` per launch)
    unsigned long long vk[2][64];
    float xyz[2][64][4];
    float bound[2][32];
    unsigned int sync;
    unsigned int pad[63];
};

__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent_f32(const float *p)
{
    return __int_as_float(__hip_atomic_load(reinterpret_cast<const int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void st_agent_u64(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_f32(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<int *>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void fps_spec2_kernel(
    int nclouds, int n, int m, KeyCodec kc, const float *__restrict__ xyz, const int *__restrict__ perm,
    float *__restrict__ temp, int *__restrict__ idx, float *__restrict__ new_xyz, Fps2Table *__restrict__ tables)
{
    constexpr int PPT = 16, SB = 4, NE = 64;
    __shared__ unsigned long long s_vk[NE];
    __shared__ float s_xyz[NE][3];
    __shared__ float s_bound[32];
    __shared__ float s_res[64];
    __shared__ int s_rank[NE], s_blk[NE];
    __shared__ unsigned long long s_evk[32];                         // this workgroup's published entries between rebuilds (the registers are full)
    __shared__ float s_exyz[32][3], s_eb[16];
    // blocks q and q + 8 (same XCD) are the two halves of cloud (q / 16) * 8 + q % 8
    const int q_ = blockIdx.x;
    const int b = (q_ >> 4) * 8 + (q_ & 7), half = (q_ >> 3) & 1;
    if (b >= nclouds) return;
    const float *__restrict__ cloud = xyz + (long)b * n * 3;
    const int *__restrict__ order = perm + (long)b * n;
    float *__restrict__ mind = temp ? temp + (long)b * n : nullptr;
    int *__restrict__ sel = idx + (long)b * m;
    float *__restrict__ nxyz = new_xyz ? new_xyz + (long)b * m * 3 : nullptr;
    Fps2Table *__restrict__ tb = tables + b;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, gw = 16 * half + w;       // gw: this wave among the cloud's 32
    __builtin_amdgcn_s_setprio(3);

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
    uint32_t pc[PPT];
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int s = gw * (64 * PPT) + i * 64 + lane;
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY, z0 = INFINITY, z1 = -INFINITY;
        if (s < n) {
            const int k = order[s];
            px[i] = cloud[3 * k]; py[i] = cloud[3 * k + 1]; pz[i] = cloud[3 * k + 2];
            pt[i] = mind ? mind[k] : 1e10f;
            pc[i] = (kc.encode(k) << SB) | (uint32_t)i;
            x0 = x1 = px[i]; y0 = y1 = py[i]; z0 = z1 = pz[i];
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            pt[i] = -INFINITY;
            pc[i] = 0xffffffffu;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, d, 64)); x1 = fmaxf(x1, __shfl_xor(x1, d, 64));
            y0 = fminf(y0, __shfl_xor(y0, d, 64)); y1 = fmaxf(y1, __shfl_xor(y1, d, 64));
            z0 = fminf(z0, __shfl_xor(z0, d, 64)); z1 = fmaxf(z1, __shfl_xor(z1, d, 64));
        }
        if (lane == i) { bx0 = x0; bx1 = x1; by0 = y0; by1 = y1; bz0 = z0; bz1 = z1; }
    }
    constexpr int GP = 64 / PPT;
    bx0 = __shfl(bx0, lane & (PPT - 1), 64); bx1 = __shfl(bx1, lane & (PPT - 1), 64);
    by0 = __shfl(by0, lane & (PPT - 1), 64); by1 = __shfl(by1, lane & (PPT - 1), 64);
    bz0 = __shfl(bz0, lane & (PPT - 1), 64); bz1 = __shfl(bz1, lane & (PPT - 1), 64);
    unsigned long long touched = 0ull;
    float bound = INFINITY;
    auto box_mask = [&](float ox, float oy, float oz, bool live) __attribute__((always_inline)) {
        const float dx = fmax_raw(fmax_raw(bx0 - ox, ox - bx1), 0.f);
        const float dy = fmax_raw(fmax_raw(by0 - oy, oy - by1), 0.f);
        const float dz = fmax_raw(fmax_raw(bz0 - oz, oz - bz1), 0.f);
        const float lb = dx * dx + dy * dy + dz * dz;
        return __ballot(live && !(lb * 0.99999f >= bound));
    };
    auto update = [&](float ox, float oy, float oz, unsigned long long mask) __attribute__((always_inline)) {
        if (mask != 0ull) {
            touched |= mask;
            // tile pairs on packed f32 arithmetic, as in fps_spec_kernel
            const pk_f32x2 o_x = {ox, ox}, o_y = {oy, oy}, o_z = {oz, oz};
            if (kc.hipcc) {
#pragma unroll
                for (int i = 0; i < PPT; i += 2)
                    if ((mask >> i) & 3ull) {
                        const pk_f32x2 dx = (pk_f32x2){px[i], px[i + 1]} - o_x, dy = (pk_f32x2){py[i], py[i + 1]} - o_y,
                                       dz = (pk_f32x2){pz[i], pz[i + 1]} - o_z;
                        const pk_f32x2 d = __builtin_elementwise_fma(dy, dy, dx * dx) + dz * dz;
                        pt[i] = fmin_raw(d.x, pt[i]); pt[i + 1] = fmin_raw(d.y, pt[i + 1]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < PPT; i += 2)
                    if ((mask >> i) & 3ull) {
                        const pk_f32x2 dx = (pk_f32x2){px[i], px[i + 1]} - o_x, dy = (pk_f32x2){py[i], py[i + 1]} - o_y,
                                       dz = (pk_f32x2){pz[i], pz[i + 1]} - o_z;
                        const pk_f32x2 d = (dx * dx + dy * dy) + dz * dz;
                        pt[i] = fmin_raw(d.x, pt[i]); pt[i + 1] = fmin_raw(d.y, pt[i + 1]);
                    }
            }
        }
    };
    if (t == 0 && half == 0) {
        sel[0] = 0;
        if (nxyz) { nxyz[0] = cloud[0]; nxyz[1] = cloud[1]; nxyz[2] = cloud[2]; }
    }
    if (t < NE) { s_rank[t] = 0; s_blk[t] = 0; }
    int j = 1;
    if (m > 1) {
        const float ox = cloud[0], oy = cloud[1], oz = cloud[2];
        update(ox, oy, oz, box_mask(ox, oy, oz, lane < PPT));
    }
    touched = ~0ull;
    // this wave's published entries are kept in LDS between rebuilds; every round copies them into the round's table
    unsigned int round = 0;
    while (j < m) {
        if (touched != 0ull) {
            float bv = -INFINITY, sv = -INFINITY;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                sv = __builtin_amdgcn_fmed3f(bv, sv, pt[i]);
                bv = fmax_raw(bv, pt[i]);
            }
            uint32_t lk = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const uint32_t c = pt[i] == bv ? pc[i] : 0xffffffffu;
                lk = c < lk ? c : lk;
            }
            const float v1 = wave_max_f32(bv);
            const uint32_t c1 = wave_min_u32(bv == v1 ? lk : 0xffffffffu);
            const unsigned long long m1 = __ballot(bv == v1 && lk == c1);
            const int l1 = m1 ? (int)__builtin_ctzll(m1) : 0;
            const float bv2 = lane == l1 ? -INFINITY : bv;
            const float v2 = wave_max_f32(bv2);
            const uint32_t c2 = wave_min_u32(bv2 == v2 ? lk : 0xffffffffu);
            const unsigned long long m2 = __ballot(lane != l1 && bv2 == v2 && lk == c2);
            const int l2 = m2 ? (int)__builtin_ctzll(m2) : l1;
            const float v3 = wave_max_f32((lane == l1 || lane == l2) ? -INFINITY : bv);
            const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), l1));
            const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), l2));
            const float e_b = fmaxf(v3, fmaxf(s1, s2));
            bound = v1;
            const int sl1 = (int)(c1 & ((1u << SB) - 1u)), sl2 = (int)(c2 & ((1u << SB) - 1u));
            const bool has2 = m2 != 0ull && c2 != 0xffffffffu;
            float x1 = 0.f, y1 = 0.f, z1 = 0.f, x2 = 0.f, y2 = 0.f, z2 = 0.f;
            fs_pick3<PPT>(px, py, pz, __builtin_amdgcn_readfirstlane(sl1), l1, x1, y1, z1);
            fs_pick3<PPT>(px, py, pz, __builtin_amdgcn_readfirstlane(sl2), l2, x2, y2, z2);
            const float ev = lane == 0 ? v1 : (has2 ? v2 : -INFINITY);
            const uint32_t ek = lane == 0 ? (c1 == 0xffffffffu ? 0xffffffffu : (c1 >> SB)) : (has2 ? (c2 >> SB) : 0xffffffffu);
            if (lane < 2) {
                s_evk[2 * w + lane] = pack_candidate(ev, ek);
                s_exyz[2 * w + lane][0] = lane == 0 ? x1 : x2; s_exyz[2 * w + lane][1] = lane == 0 ? y1 : y2; s_exyz[2 * w + lane][2] = lane == 0 ? z1 : z2;
                if (lane == 0) s_eb[w] = e_b;
            }
        }
        const int par = (int)(round & 1u);
        if (lane < 2) {                                               // (the same lanes wrote the LDS copy: program order)
            const int e = 2 * gw + lane, le = 2 * w + lane;
            st_agent_u64(&tb->vk[par][e], s_evk[le]);
            st_agent_f32(&tb->xyz[par][e][0], s_exyz[le][0]); st_agent_f32(&tb->xyz[par][e][1], s_exyz[le][1]);
            st_agent_f32(&tb->xyz[par][e][2], s_exyz[le][2]);
            if (lane == 0) st_agent_f32(&tb->bound[par][gw], s_eb[w]);
        }
        // ---- the round's cross-workgroup barrier: both halves have published.  ADVICE r5: a workgroup barrier orders nothing beyond the
        // workgroup, and thread 0's agent-scope release waits for ITS wave's stores only.  So the table stores are agent-scope
        // write-throughs (st_agent_*: they are complete when they are acknowledged, nothing stays dirty in a cache that a release
        // would have to write back) and EVERY wave waits for its own acknowledgements before the workgroup barrier: when thread 0
        // signals, the whole table is at the agent's coherence point.  (A full agent-scope release fence per wave and round --
        // buffer_wbl2, an L2 write-back walk -- does the same and HALVED the kernel: double.yaml 5011 -> 2420 scenes/s.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(&tb->sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = 2u * (round + 1u);
            // bounded: the host side only launches grids that fit the chip twice over (fps2_capacity), so the partner IS resident or
            // becomes resident as other work retires; should it never arrive (a CU mask narrower than the one the capacity was
            // computed under, a partition change) the launch dies loudly after ~4 s of the 100-MHz clock instead of hanging the GPU
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&tb->sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 400000000ll) __builtin_trap();
            }
        }
        __syncthreads();
        if (t < NE) {
            s_vk[t] = ld_agent_u64(&tb->vk[par][t]);
            s_xyz[t][0] = ld_agent_f32(&tb->xyz[par][t][0]);
            s_xyz[t][1] = ld_agent_f32(&tb->xyz[par][t][1]);
            s_xyz[t][2] = ld_agent_f32(&tb->xyz[par][t][2]);
            if (t < 32) s_bound[t] = ld_agent_f32(&tb->bound[par][t]);
        }
        __syncthreads();                                              // A: the table is complete (in this workgroup's LDS)
        // ---- merge: lane = entry i, wave w compares it with entries 4 w .. 4 w + 3 (all 64 x 64 pairs over the 16 waves)
        const int left = m - j;
        const int i = lane;
        const unsigned long long pki = s_vk[i];
        float cv; uint32_t ck;
        unpack_candidate(pki, cv, ck);
        const float cx = s_xyz[i][0], cy = s_xyz[i][1], cz = s_xyz[i][2];
        {
            int rk = 0;
            bool bl = false;
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * w + u;
                const unsigned long long pke = s_vk[e];
                const float ex = s_xyz[e][0], ey = s_xyz[e][1], ez = s_xyz[e][2];
                const bool before = pke > pki;
                const float d = kc.hipcc ? fps_dist<true>(cx, cy, cz, ex, ey, ez) : sqdist3(cx, cy, cz, ex, ey, ez);
                rk += before ? 1 : 0;
                bl = bl || (before && (d < cv));
            }
            if (rk) atomicAdd(&s_rank[i], rk);
            if (bl) atomicOr(&s_blk[i], 1);
        }
        lds_barrier();                                                // A2
        if (w == 0) {
            const int rank = s_rank[i];
            const bool blocked = s_blk[i] != 0;
            const float gB = wave_max_f32(lane < 32 ? s_bound[lane] : -INFINITY);
            const bool valid = !(ck == 0xffffffffu || !(cv > -1.0f));
            const bool pass = rank == 0 ? true : ((cv > gB) && (cv > 0.f) && valid && !blocked);
            const bool stop = !pass || (rank == 0 && !valid);
            int r = (int)wave_min_u32((stop && !(rank == 0)) ? (uint32_t)rank : ((rank == 0 && !valid) ? 1u : 64u));
            r = min(min(r, left), FS_RMAX);
            if (r < 1) r = 1;
            if (rank < r) {
                const bool v0 = valid;
                const float ox = v0 ? cx : cloud[0], oy = v0 ? cy : cloud[1], oz = v0 ? cz : cloud[2];
                s_res[1 + 3 * rank] = ox; s_res[2 + 3 * rank] = oy; s_res[3 + 3 * rank] = oz;
                if (half == 0) {
                    sel[j + rank] = valid ? kc.decode(ck) : 0;
                    if (nxyz) { nxyz[3 * (j + rank)] = ox; nxyz[3 * (j + rank) + 1] = oy; nxyz[3 * (j + rank) + 2] = oz; }
                }
            }
            s_rank[i] = 0; s_blk[i] = 0;
            if (lane == 0) s_res[0] = __int_as_float(r);
        }
        lds_barrier();                                                // B
        const float rv = s_res[lane];
        const int r = __builtin_amdgcn_readfirstlane(__float_as_int(rv));
        j += r;
        const int apply_n = j >= m ? r - 1 : r;
        touched = 0ull;
        for (int q0 = 0; q0 < apply_n; q0 += GP) {
            const int qg = q0 + lane / PPT;
            const int src = 1 + 3 * min(qg, FS_RMAX - 1);
            const float gx = __shfl(rv, src, 64), gy = __shfl(rv, src + 1, 64), gz = __shfl(rv, src + 2, 64);
            const unsigned long long masks = box_mask(gx, gy, gz, qg < apply_n);
            if (masks == 0ull) continue;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const unsigned long long mk = (masks >> (g * PPT)) & ((1ull << PPT) - 1ull);
                if (mk != 0ull) {
                    const int q = q0 + g;
                    const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 1 + 3 * q));
                    const float oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 2 + 3 * q));
                    const float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 3 + 3 * q));
                    update(ox, oy, oz, mk);
                }
            }
        }
        ++round;
    }
    if (mind) {
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (pc[i] != 0xffffffffu) mind[kc.decode(pc[i] >> SB)] = pt[i];
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
            const float d = kc.hipcc ? fps_dist<true>(cloud[3 * k], cloud[3 * k + 1], cloud[3 * k + 2], ox, oy, oz)
                                     : sqdist3(cloud[3 * k], cloud[3 * k + 1], cloud[3 * k + 2], ox, oy, oz);
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
static void launch_reg(int b, int n, int m, KeyCodec kc, const float *xyz, float *temp, int *idx, hipStream_t st, float *new_xyz = nullptr)
{
    if ((1 << kc.log2bs) == 64 * WAVES)
        hipLaunchKernelGGL((fps_reg_kernel<WAVES, PPT, true>), dim3(b), dim3(64 * WAVES), 0, st, n, m, kc, xyz, temp, idx, new_xyz);
    else
        hipLaunchKernelGGL((fps_reg_kernel<WAVES, PPT, false>), dim3(b), dim3(64 * WAVES), 0, st, n, m, kc, xyz, temp, idx, new_xyz);
}

}  // namespace prcnn

using namespace prcnn;

extern "C" int prcnn_opt_n_threads(int work_size) { return work_size > 0 ? host_opt_n_threads(work_size) : 1; }

// 0 (default): the distance of sampling_gpu.cu:133 with one rounding per source operation (the parity contract: what the source says,
// whatever a compiler contracts); 1: the arithmetic of the reference's kernel binary as hipcc builds it for gfx950
// ((fma(dy, dy, dx*dx)) + dz*dz: see fps_dist).  Process-wide, read at launch; applies to every FPS entry point of the library.
static int g_fps_hipcc = 0;
extern "C" int prcnn_set_fps_arithmetic(int mode)
{
    PRCNN_REQUIRE(mode == 0 || mode == 1, "set_fps_arithmetic: mode %d", mode);
    g_fps_hipcc = mode;
    return PRCNN_OK;
}

namespace prcnn {
__global__ __launch_bounds__(256) void fps_fill_kernel(long count, float v, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = v;
}
__global__ __launch_bounds__(256) void fps_gather_xyz_kernel(int n, int m, const float *__restrict__ xyz, const int *__restrict__ idx,
                                                             float *__restrict__ new_xyz)
{
    const int chunks = (m + 255) / 256;
    const int b = blockIdx.x / chunks, j = (blockIdx.x % chunks) * 256 + threadIdx.x;
    if (j >= m) return;
    const float *p = xyz + ((long)b * n + idx[(long)b * m + j]) * 3;
    float *o = new_xyz + ((long)b * m + j) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}
}  // namespace prcnn

// new_xyz != NULL (prcnn_fps_new_xyz; temp may then be NULL): the speculative kernel (2048 < n <= 16384, m >= 256) writes the
// coordinates itself; every other route (few samples, small clouds, n > 32768 or a device that cannot hold the two-workgroup kernel) runs its kernel over an
// internal distance scratch filled with the reference caller's 1e10 and gathers the coordinates behind it -- same indices, same
// coordinates, two small launches more (round 5; ADVICE r4: those routes used to refuse)
// how many fps_spec2_kernel workgroups the current device holds at once (0: it refuses the dynamic LDS); cached per device
static int fps2_capacity(size_t pad)
{
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    int cap = 0, per_cu = 0, cus = 0;
    if (ensure_dynamic_lds((const void *)prcnn::fps_spec2_kernel, pad, "furthest_point_sampling(two workgroups)") == PRCNN_OK &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)prcnn::fps_spec2_kernel, 1024, pad) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
        cap = per_cu * cus;
    else
        (void)hipGetLastError();
    if (const char *e = getenv("PRCNN_FPS2_CAPACITY")) cap = atoi(e);    // tests: force the chunked launches / the fallback
    cache[dev] = cap;
    return cap;
}

static int fps_any(int b, int n, int m, const float *xyz, float *temp, int *idx, float *new_xyz, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "fps: bad sizes b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(n > 0, "fps: empty cloud with m=%d", m);
    PRCNN_REQUIRE(xyz && (temp || new_xyz) && idx, "fps: null pointer");
    hipStream_t st = (hipStream_t)stream;

    const int bs = host_opt_n_threads(n);
    KeyCodec kc;
    kc.hipcc = g_fps_hipcc;
    kc.log2bs = 0;
    while ((1 << kc.log2bs) < bs) ++kc.log2bs;
    const int nq = (n + bs - 1) / bs;  // values of k div bs: 0 .. nq-1
    kc.sh = 0;
    while ((1 << kc.sh) < nq) ++kc.sh;
    PRCNN_REQUIRE(kc.sh + kc.log2bs <= 31, "fps: n=%d too large for the 32-bit tie key", n);

    // large clouds: Morton ordering + pruned scan (exact).  It needs (n ints) of scratch per scene and pays
    // off when the sample count is large enough for the pruning radius to shrink.
    // the two-workgroup kernel spins on its partner: only where BOTH halves of every cloud of a launch are resident at once.  One such
    // workgroup holds a CU (84 KB of LDS); a launch takes at most HALF of what the device can hold (the other geometry stream may be
    // running the same kernel: per XCD at most one unpaired block per launch is resident, every other resident block has its partner
    // and makes progress), larger batches go as several launches.  A device that refuses the LDS (64 KB parts) or has fewer than 32
    // such slots keeps fps_generic_kernel
    static const size_t pad2 = (size_t)(getenv("PRCNN_FPS_LDS_PAD") ? atoi(getenv("PRCNN_FPS_LDS_PAD")) : 84) * 1024;
    const bool pair_shape = n > 16384 && n <= 32768 && m >= 256;
    const int cap2 = pair_shape ? fps2_capacity(pad2) : 0;
    const bool pair_ok = pair_shape && cap2 >= 32;
    const bool writes_xyz = (n > 2048 && n <= 16384 && m >= 256) || pair_ok;
    if (new_xyz && !writes_xyz) {
        if (!temp) {
            temp = (float *)scratch_for(st, (size_t)b * n * sizeof(float), 12);
            if (!temp) { set_error("fps_new_xyz: cannot allocate the distance scratch"); return PRCNN_ELAUNCH; }
            hipLaunchKernelGGL(fps_fill_kernel, dim3((unsigned)(((long)b * n + 255) / 256)), dim3(256), 0, st, (long)b * n, 1e10f, temp);
        }
        const int rc = fps_any(b, n, m, xyz, temp, idx, nullptr, stream);
        if (rc != PRCNN_OK) return rc;
        hipLaunchKernelGGL(fps_gather_xyz_kernel, dim3((unsigned)((long)((m + 255) / 256) * b)), dim3(256), 0, st, n, m, xyz, idx, new_xyz);
        return check_launch("fps_new_xyz(gather)");
    }
    if (pair_ok) {
        // two workgroups per cloud, blocks q and q + 8 (fps_spec2_kernel): the grid is padded to whole sets of 16 blocks
        int *perm = (int *)scratch_for(st, (size_t)b * n * sizeof(int), 1);
        Fps2Table *tables = (Fps2Table *)scratch_for(st, (size_t)b * sizeof(Fps2Table), 13);
        if (!perm || !tables) { set_error("fps: cannot allocate the ordering / exchange scratch"); return PRCNN_ELAUNCH; }
        hipLaunchKernelGGL(fps_order_kernel, dim3(b), dim3(1024), 0, st, n, xyz, perm);
        if (hipMemsetAsync(tables, 0, (size_t)b * sizeof(Fps2Table), st) != hipSuccess) {
            (void)hipGetLastError();
            set_error("fps: cannot reset the exchange tables");
            return PRCNN_ELAUNCH;
        }
        // (the placement hint of the one-workgroup kernel is REQUIRED here: with more than half of a CU's LDS per workgroup no two of
        //  these spinning workgroups share a CU, and every XCD dispatches a cloud's halves back to back)
        const int per = (cap2 / 2 / 16) * 8;                          // clouds per launch: whole sets of 16 blocks, half the capacity
        for (int off = 0; off < b; off += per) {
            const int nb = b - off < per ? b - off : per;
            const unsigned grid = (unsigned)((nb + 7) / 8) * 16u;
            hipLaunchKernelGGL(fps_spec2_kernel, dim3(grid), dim3(1024), pad2, st, nb, n, m, kc, xyz + (size_t)off * n * 3, perm + (size_t)off * n,
                               temp ? temp + (size_t)off * n : nullptr, idx + (size_t)off * m, new_xyz ? new_xyz + (size_t)off * m * 3 : nullptr,
                               tables + off);
        }
        return check_launch("furthest_point_sampling(two workgroups)");
    }
    if (n > 2048 && n <= 16384 && m >= 256) {
        int *perm = (int *)scratch_for(st, (size_t)b * n * sizeof(int), 1);
        if (!perm) { set_error("fps: cannot allocate ordering scratch"); return PRCNN_ELAUNCH; }
        hipLaunchKernelGGL(fps_order_kernel, dim3(b), dim3(1024), 0, st, n, xyz, perm);
        // Unused dynamic LDS as a placement hint: with more than half of a CU's LDS requested, two of these
        // latency-bound 1024-thread workgroups (the chains of two batches run on two streams) never share a CU.
        // (+1.2 % end to end; only when the batch is small enough that one workgroup per CU costs no concurrency)
        static const size_t pad_cfg = (size_t)(getenv("PRCNN_FPS_LDS_PAD") ? atoi(getenv("PRCNN_FPS_LDS_PAD")) : 84) * 1024;
        const size_t pad = b <= 128 ? pad_cfg : 0;
        // 16 waves per cloud: 8 waves x 32 points per lane runs 4.98 ms, 4 waves x 64 points 7.7 ms (16384 -> 4096, 4.2-4.3 ms here):
        // the per-iteration update of the touched tiles parallelises over waves, the exchange does not get cheaper with fewer
        const void *ss[3] = {(const void *)fps_spec_kernel<4>, (const void *)fps_spec_kernel<8>, (const void *)fps_spec_kernel<16>};
        if (pad)
            for (const void *k : ss) {
                const int rc = ensure_dynamic_lds(k, pad, "furthest_point_sampling(speculative)");
                if (rc != PRCNN_OK) return rc;
            }
        if (n <= 4096) hipLaunchKernelGGL((fps_spec_kernel<4>), dim3(b), dim3(1024), pad, st, n, m, kc, xyz, perm, temp, idx, new_xyz);
        else if (n <= 8192) hipLaunchKernelGGL((fps_spec_kernel<8>), dim3(b), dim3(1024), pad, st, n, m, kc, xyz, perm, temp, idx, new_xyz);
        else hipLaunchKernelGGL((fps_spec_kernel<16>), dim3(b), dim3(1024), pad, st, n, m, kc, xyz, perm, temp, idx, new_xyz);
        return check_launch("furthest_point_sampling(speculative)");
    }
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

extern "C" int prcnn_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                             int *idx, void *stream)
{
    PRCNN_REQUIRE(temp || b == 0 || m == 0, "fps: null pointer");
    return fps_any(b, n, m, xyz, temp, idx, nullptr, stream);
}

// FPS of many small clouds (n <= 1024: one wave per cloud, everything in registers) with the selected coordinates written
// alongside the indices: what furthest_point_sample + gather_operation (pointnet2_modules.py:40-46) produce, without the
// caller's 1e10 fill of the distance scratch, the index cast and the gather launch.  idx (b,m), new_xyz (b,m,3).
extern "C" int prcnn_fps_new_xyz(int b, int n, int m, const float *xyz, int *idx, float *new_xyz, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n > 0 && m >= 0, "fps_new_xyz: b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && idx && new_xyz, "fps_new_xyz: null pointer");
    if (n > 1024) return fps_any(b, n, m, xyz, nullptr, idx, new_xyz, stream);      // round 4: the speculative kernel writes them too
    hipStream_t st = (hipStream_t)stream;
    const int bs = host_opt_n_threads(n);
    KeyCodec kc;
    kc.hipcc = g_fps_hipcc;
    kc.log2bs = 0;
    while ((1 << kc.log2bs) < bs) ++kc.log2bs;
    const int nq = (n + bs - 1) / bs;
    kc.sh = 0;
    while ((1 << kc.sh) < nq) ++kc.sh;
    if (n <= 128) launch_reg<1, 2>(b, n, m, kc, xyz, nullptr, idx, st, new_xyz);
    else if (n <= 256) launch_reg<1, 4>(b, n, m, kc, xyz, nullptr, idx, st, new_xyz);
    else if (n <= 512) launch_reg<1, 8>(b, n, m, kc, xyz, nullptr, idx, st, new_xyz);
    else if (b >= 128) launch_reg<1, 16>(b, n, m, kc, xyz, nullptr, idx, st, new_xyz);
    else launch_reg<4, 4>(b, n, m, kc, xyz, nullptr, idx, st, new_xyz);             // few clouds: four waves each (as prcnn_furthest_point_sampling)
    return check_launch("fps_new_xyz");
}

// ---- spatial groups of a cloud for the consumers that sweep it per box (csrc/roipool.hip) -------------------------------------
// The cloud in the Morton order of fps_order_kernel, 64 points per group: pxyz (b, n) float4 = (x, y, z, original index as bits),
// aabb (b, n / 64, 2) float4 = per-group (min x, min y, min z, -) and (max x, max y, max z, -).  A box then tests 256 group boxes
// instead of 16384 points and reads the few groups that can hold a point of it as coalesced 16-byte lanes.
namespace prcnn {
__global__ __launch_bounds__(256) void point_groups_kernel(int n, const float *__restrict__ xyz, const int *__restrict__ perm,
                                                           float4 *__restrict__ pxyz, float4 *__restrict__ aabb)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g * 64 >= n) return;
    const long s = (long)b * n + g * 64 + lane;
    const int k = perm[s];
    const float *p = xyz + ((long)b * n + k) * 3;
    const float x = p[0], y = p[1], z = p[2];
    pxyz[s] = make_float4(x, y, z, __int_as_float(k));
    float x0 = x, x1 = x, y0 = y, y1 = y, z0 = z, z1 = z;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d, 64)); x1 = fmaxf(x1, __shfl_xor(x1, d, 64));
        y0 = fminf(y0, __shfl_xor(y0, d, 64)); y1 = fmaxf(y1, __shfl_xor(y1, d, 64));
        z0 = fminf(z0, __shfl_xor(z0, d, 64)); z1 = fmaxf(z1, __shfl_xor(z1, d, 64));
    }
    if (lane == 0) {
        aabb[((long)b * (n / 64) + g) * 2] = make_float4(x0, y0, z0, 0.f);
        aabb[((long)b * (n / 64) + g) * 2 + 1] = make_float4(x1, y1, z1, 0.f);
    }
}
}  // namespace prcnn

extern "C" int prcnn_point_groups(int b, int n, const float *xyz, float *pxyz, float *aabb, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n > 0 && n % 64 == 0 && n <= 65536 && b <= 65535, "point_groups: b=%d n=%d (n a multiple of 64, <= 65536)", b, n);
    if (b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && pxyz && aabb && (((uintptr_t)pxyz | (uintptr_t)aabb) & 15) == 0, "point_groups: null / misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    int *perm = (int *)scratch_for(st, (size_t)b * n * sizeof(int), 8);
    if (!perm) { set_error("point_groups: cannot allocate ordering scratch"); return PRCNN_ELAUNCH; }
    hipLaunchKernelGGL(fps_order_kernel, dim3(b), dim3(1024), 0, st, n, xyz, perm);
    hipLaunchKernelGGL(point_groups_kernel, dim3((n / 64 + 3) / 4, b), dim3(256), 0, st, n, xyz, perm, (float4 *)pxyz, (float4 *)aabb);
    return check_launch("point_groups");
}

// ---- the whole geometry chain of the RCNN's RoI clouds in ONE kernel (round 3) -----------------------------------------------
// rcnn_net.py:165-175 runs, per RoI, SA level 1 (sample 128 of the 512 pooled points, ball query r1 / 64) and SA level 2 (sample 32 of
// those 128, ball query r2 / 64).  As separate launches over the 800 RoI clouds of a batch that was FPS, limited ball query,
// representative map, FPS, ball query, representative map: six latency-bound kernels (one wave per cloud each, ~0.8 us per dependent
// FPS iteration) that were 0.41 ms of the proposal stream in the pipelined step and a quarter of a millisecond of host time.
// Here ONE WAVE serves a RoI from the pooled coordinates to both index tensors: the cloud stays in registers (8 points per lane)
// through sampling and the first ball query, the 128 sampled centres stay in registers (2 per lane) through the second pair; the hit
// lists are staged in LDS ([slot][centre]) and leave as coalesced rows.  Per operator the arithmetic is that of fps_reg_kernel /
// ball_query_kernel / dup_rep_kernel: same indices, bit for bit (tests/test_gpu_ops.py compares with the separate entry points,
// tests/test_gpu_shadow.py with the oracle).
namespace prcnn {

constexpr int RG_N = 512, RG_M1 = 128, RG_M2 = 32, RG_NS = 64;
constexpr int RG_LD1 = RG_M1 + 1, RG_LD2 = RG_M2 + 1;   // row strides of the staged hit lists (16-bit entries): odd, see the rows-out loops

// coordinates of point (lane l, slot) -- both wave-uniform, slot < D -- of the D first register slots: a scalar compare ladder in front
// of three v_readlane (fs_pick3 over a prefix of the arrays)
template <int PPT, int D, int I = 0>
__device__ __forceinline__ void roi_pick3(const float (&px)[PPT], const float (&py)[PPT], const float (&pz)[PPT], int slot, int l,
                                          float &x, float &y, float &z)
{
    if (I + 1 == D || slot == I) {
        x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[I]), l));
        y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[I]), l));
        z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[I]), l));
    } else if constexpr (I + 1 < D) {
        roi_pick3<PPT, D, I + 1>(px, py, pz, slot, l, x, y, z);
    }
}

// FPS of a cloud of n <= 64 PPT points held in registers (point k = lane + 64 i) -> sel[0..m) in LDS; all lanes in step.  Returns the
// number of picks made before only copies of picked points were left (m if that never happened): sel[that ..] = 0.
//
// Only the first `lim` points are distinct: D = ceil(lim / 64) register slots take part instead of PPT (round 5; a pooled RoI cloud has
// 20-80 distinct points of its 512 on the synthetic scenes, and a pick was ~130 VALU instructions of a single-wave dependent chain).
//   MOD = true:  point k >= lim is a copy of point k % lim (the pooled rows).  A copy has its source's coordinates, hence its source's
//     running minimum at every step, and the scan's pick is the arg-max with the smallest tie key: it is decided among the SOURCES when
//     each carries the smallest key of its copies; the index handed back is that copy's, as in the scan over all n.
//   MOD = false: the points k >= lim are all copies of point 0 (the sampled centres behind an exhausted level-1 scan).  Point 0 is the
//     first pivot: its minimum is 0 from the first step on and its key never decides a pick before the exit below.
// The exit: a best value of exactly 0 means only copies of picked points are left -- every later pick is point 0 (fps_reg_kernel).
template <int PPT, int D, bool MOD>
__device__ __forceinline__ int roi_fps(int n, int lim, int m, KeyCodec kc, const float (&px)[PPT], const float (&py)[PPT],
                                       const float (&pz)[PPT], int *__restrict__ s_sel, const int lane)
{
    float pt[D];
    uint32_t pk[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int k = lane + 64 * i;
        pt[i] = k < lim ? 1e10f : -INFINITY;
        uint32_t key = 0xffffffffu;
        if (k < lim) {
            key = kc.encode(k);
            if (MOD)
                for (int c = k + lim; c < n; c += lim) {
                    const uint32_t kc2 = kc.encode(c);
                    key = kc2 < key ? kc2 : key;
                }
        }
        pk[i] = key;
    }
    if (lane == 0) s_sel[0] = 0;
    float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[0]), 0));
    float oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[0]), 0));
    float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[0]), 0));
    for (int j = 1; j < m; ++j) {
        float lv = -INFINITY;
        if constexpr (D >= 2) {
            // slot PAIRS on packed f32 arithmetic (each half the scalar form's operations, one rounding each: same bits; see fps_spec_kernel)
            const pk_f32x2 o_x = {ox, ox}, o_y = {oy, oy}, o_z = {oz, oz};
#pragma unroll
            for (int i = 0; i < D; i += 2) {
                const pk_f32x2 dx = (pk_f32x2){px[i], px[i + 1]} - o_x, dy = (pk_f32x2){py[i], py[i + 1]} - o_y, dz = (pk_f32x2){pz[i], pz[i + 1]} - o_z;
                const pk_f32x2 d = kc.hipcc ? (pk_f32x2)(__builtin_elementwise_fma(dy, dy, dx * dx) + dz * dz) : (pk_f32x2)((dx * dx + dy * dy) + dz * dz);
                pt[i] = fmin_raw(d.x, pt[i]); pt[i + 1] = fmin_raw(d.y, pt[i + 1]);
                lv = fmax_raw(lv, fmax_raw(pt[i], pt[i + 1]));
            }
        } else {
            const float d = kc.hipcc ? fps_dist<true>(px[0], py[0], pz[0], ox, oy, oz) : sqdist3(px[0], py[0], pz[0], ox, oy, oz);
            pt[0] = fmin_raw(d, pt[0]);
            lv = pt[0];
        }
        const float bv = wave_max_f32(lv);
        uint32_t lk = 0xffffffffu;                               // this lane's smallest key among its slots at the best value, and its slot
        int ls = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const bool take = pt[i] == bv && pk[i] < lk;
            lk = take ? pk[i] : lk;
            ls = take ? i : ls;
        }
        // one lane at the best value (the rule once the points are distinct): its key is the answer, no second reduction
        const unsigned long long at = __ballot(lv == bv);
        uint32_t bkey;
        int wl;
        if (__builtin_popcountll(at) == 1) {
            wl = (int)__builtin_ctzll(at);
            bkey = (uint32_t)__builtin_amdgcn_readlane((int)lk, wl);
        } else {
            bkey = wave_min_u32(lk);
            const unsigned long long wm = __ballot(lk == bkey);
            wl = wm ? (int)__builtin_ctzll(wm) : 0;
        }
        const bool valid = !(bkey == 0xffffffffu || !(bv > -1.0f));
        int old = valid ? kc.decode(bkey) : 0;
        old = __builtin_amdgcn_readfirstlane(old);
        if (lane == 0) s_sel[j] = old;
        if (bv == 0.f) {
            for (int jj = j + 1 + lane; jj < m; jj += 64) s_sel[jj] = 0;
            return j;
        }
        // the next pivot: the winner's source sits in slot `ls` of lane `wl` (point 0 behind an invalid best)
        wl = valid ? wl : 0;
        const int slot = valid ? __builtin_amdgcn_readlane(ls, wl) : 0;
        roi_pick3<PPT, D>(px, py, pz, slot, wl, ox, oy, oz);
    }
    return m;
}

template <int PPT, bool MOD>
__device__ __forceinline__ int roi_fps_any(int n, int lim, int m, KeyCodec kc, const float (&px)[PPT], const float (&py)[PPT],
                                           const float (&pz)[PPT], int *__restrict__ s_sel, const int lane)
{
    static_assert(PPT == 8 || PPT == 2, "the two shapes of the RoI chain");
    if constexpr (PPT == 8) {
        if (lim > 256) return roi_fps<PPT, 8, MOD>(n, lim, m, kc, px, py, pz, s_sel, lane);
        if (lim > 128) return roi_fps<PPT, 4, MOD>(n, lim, m, kc, px, py, pz, s_sel, lane);
    }
    if (lim > 64) return roi_fps<PPT, 2, MOD>(n, lim, m, kc, px, py, pz, s_sel, lane);
    return roi_fps<PPT, 1, MOD>(n, lim, m, kc, px, py, pz, s_sel, lane);
}

// first `ns` in-range points (k < n_scan, index order) of CPL centres per lane among the PPT * 64 points in registers -> hit lists in
// LDS, s_hits[slot * stride + centre], counts in cnt[] (returned in registers); lanes whose centres are all full stop the scan early together
template <int PPT, int CPL>
__device__ __forceinline__ void roi_ball_query(int n_scan, int ns, float r2, const float (&px)[PPT], const float (&py)[PPT],
                                               const float (&pz)[PPT], const float (&cx)[CPL], const float (&cy)[CPL], const float (&cz)[CPL],
                                               const bool (&live)[CPL], unsigned short *__restrict__ s_hits, int stride, int (&cnt)[CPL],
                                               const int lane)
{
#pragma unroll
    for (int q = 0; q < CPL; ++q) cnt[q] = live[q] ? 0 : ns;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int base = 64 * i;
        if (base >= n_scan) break;
        bool full = true;
#pragma unroll
        for (int q = 0; q < CPL; ++q) full = full && cnt[q] >= ns;
        if (__all(full)) break;
        const int nb = min(64, n_scan - base);
        for (int l = 0; l < nb; ++l) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[i]), l));
            const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[i]), l));
            const float z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[i]), l));
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const float d2 = sqdist3(cx[q], cy[q], cz[q], x, y, z);
                if (d2 < r2 && cnt[q] < ns) {
                    s_hits[cnt[q] * stride + lane + 64 * q] = (unsigned short)(base + l);
                    ++cnt[q];
                }
            }
        }
    }
}

// rows of an index tensor out of the staged hit lists: slot s of centre c; slots past the hit count repeat the first hit, an empty ball is
// a row of zeros.  The lists sit in LDS as [slot][centre] with an ODD row stride: the ball query writes a slot of 64 centres side by
// side, this loop reads the 64 slots of one centre -- 64 consecutive rows -- and with an even stride (128 entries = 256 bytes) those
// were 64 addresses in ONE bank: 300 cycles per row, a fifth of the kernel.  cnt_lo / cnt_hi: the hit counts of centres lane / lane + 64.
template <int M>
__device__ __forceinline__ void roi_rows_out(int ns, const unsigned short *__restrict__ s_hits, int stride, int cnt_lo, int cnt_hi,
                                             int *__restrict__ out, const int lane)
{
    if (ns == 64) {                                               // a row per wave store: centre c = the iteration, slot = the lane
#pragma unroll 8
        for (int c = 0; c < M; ++c) {
            const int tot = __builtin_amdgcn_readlane(c < 64 ? cnt_lo : cnt_hi, c & 63);
            const int v = s_hits[(lane < tot ? lane : 0) * stride + c];
            out[c * 64 + lane] = tot == 0 ? 0 : v;
        }
    } else {
        for (int e = lane; e < M * ns; e += 64) {
            const int c = e / ns, s = e - c * ns;
            const int t_lo = __shfl(cnt_lo, c & 63, 64), t_hi = __shfl(cnt_hi, c & 63, 64);
            const int tot = c < 64 ? t_lo : t_hi;
            out[e] = tot == 0 ? 0 : (int)s_hits[(s < tot ? s : 0) * stride + c];
        }
    }
}

// The row lists of the two sampled levels (prcnn_rcnn_roi_geometry_packs): what prcnn_ball_pack_ex makes of idx1 (limit, crep = rep1)
// and of idx2 (rep = rep1, crep = rep2), written by the wave that has the hit lists in LDS anyway -- as separate launches the two
// packs re-derived them from 10240 index entries per cloud (1024-thread workgroups, a block scan, a binary search per row) and cost
// the step 26 us (uniform scene) / 57 us (LiDAR-shaped) of 1060 / 1540 (profiles/sensitivity_probe.py).  hdr: zero on entry.
struct RgPacks {
    unsigned int *rowinfo1; float4 *rowdxyz1; int *tilecloud1; unsigned int *hdr1;
    unsigned int *rowinfo2; float4 *rowdxyz2; int *tilecloud2; unsigned int *hdr2;
    unsigned int *rowinfo3; float4 *rowdxyz3; unsigned int *hdr3;    // optional third list (rows carry their cloud): see the kernel's end
    int *crows1; unsigned int *hdr_c1;                               // optional: the level-1 centres that are their own representatives, as rows
    // tilecloud* == NULL: the lists' rows carry their cloud -- descriptor (cloud << 16) | (centre << 9) | point -- and are drawn from the
    // list's ROW counter hdr[1], so that tiles are cut wherever the rows fall (csrc/sa_packed.hip reads such a list when it is given no
    // tilecloud): no padded last tile per cloud
};

__device__ __forceinline__ int wave_incl_scan(int v, const int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// One cloud's list: centre c (c = lane: keep_lo / tot_lo, c = lane + 64: keep_hi / tot_hi; M centres) lists keep[c] rows -- 0 for a
// centre that copies an earlier one, else max(hits, 1) -- row p of it = hit p of its staged list (point 0 of an empty ball), in centre
// order, cut into 64-row tiles drawn from the list's counter; the last tile is filled with copies of centre M - 1's first row (as
// ball_pack_kernel fills it).  pmap: staged hit -> point of the cloud (null: itself); cmap_*: centre -> point of the cloud.
template <int M>
__device__ __forceinline__ void roi_pack_out(int b, int keep_lo, int keep_hi, int tot_lo, int tot_hi, const unsigned short *__restrict__ s_hits,
                                             int stride, int *__restrict__ s_off, const float *__restrict__ cloud,
                                             const int *__restrict__ pmap, const int *__restrict__ cmap, const int *__restrict__ cmap2,
                                             unsigned int *__restrict__ rowinfo, float4 *__restrict__ rowdxyz, int *__restrict__ tilecloud,
                                             unsigned int *__restrict__ hdr, const int lane)
{
    const int in_lo = wave_incl_scan(keep_lo, lane);
    const int sum_lo = __builtin_amdgcn_readlane(in_lo, 63);
    const int in_hi = wave_incl_scan(M > 64 ? keep_hi : 0, lane) + sum_lo;
    const int total = __builtin_amdgcn_readlane(in_hi, 63);
    if (lane < M) s_off[lane] = in_lo - keep_lo;
    if (M > 64) s_off[lane + 64] = in_hi - keep_hi;
    const bool rowcloud = tilecloud == nullptr;
    const int nt = (total + 63) >> 6;
    int base = 0;
    if (lane == 0) {
        if (rowcloud) {
            base = (int)atomicAdd(&hdr[1], (unsigned int)total);  // the cloud's first ROW of the list
        } else {
            base = (int)atomicAdd(&hdr[0], (unsigned int)nt);     // the cloud's first TILE
            atomicAdd(&hdr[1], (unsigned int)total);
        }
    }
    base = __builtin_amdgcn_readfirstlane(base);
    __syncthreads();                                              // (one wave: orders the LDS writes above before the searches below)
    if (!rowcloud)
        for (int t = lane; t < nt; t += 64) tilecloud[base + t] = b;
    unsigned int *__restrict__ dst = rowinfo + (rowcloud ? (long)base : (long)base * 64);
    float4 *__restrict__ dx = rowdxyz + (rowcloud ? (long)base : (long)base * 64);
    const int rows_out = rowcloud ? total : nt * 64;
    for (int r0 = 0; r0 < rows_out; r0 += 64) {
        const int r = r0 + lane;                                  // (every lane stays in the loop: the shuffles below read all of them)
        int c = M - 1, p = 0;
        if (r < total) {
            int lo = 0, hi = M - 1;                               // the last centre whose offset is <= r
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= r) lo = mid; else hi = mid - 1;
            }
            c = lo; p = r - s_off[lo];
        }
        const int t_lo = __shfl(tot_lo, c & 63, 64), t_hi = __shfl(tot_hi, c & 63, 64);
        const int tot = (M > 64 && c >= 64) ? t_hi : t_lo;
        const int k = tot == 0 ? 0 : (int)s_hits[p * stride + c];
        const int pi = pmap ? pmap[k] : k;
        const int ci = cmap2 ? cmap[cmap2[c]] : cmap[c];
        const float *__restrict__ pt = cloud + 3 * pi, *__restrict__ ct = cloud + 3 * ci;
        if (r < rows_out) {
            dst[r] = rowcloud ? (((unsigned int)b << 16) | ((unsigned int)c << 9) | (unsigned int)k) : (((unsigned int)c << 16) | (unsigned int)k);
            dx[r] = make_float4(pt[0] - ct[0], pt[1] - ct[1], pt[2] - ct[2], 0.f);
        }
    }
}

__global__ __launch_bounds__(64) void rcnn_roi_geometry_kernel(
    KeyCodec kc1, KeyCodec kc2, float r1sq, float r2sq, int ns1, int ns2, const float *__restrict__ xyz /* (b, 512, 3) */,
    const int *__restrict__ limit /* (b) */, float *__restrict__ new_xyz1 /* (b,128,3) */, int *__restrict__ idx1 /* (b,128,ns1) */,
    int *__restrict__ rep1 /* (b,128) */, float *__restrict__ new_xyz2 /* (b,32,3) */, int *__restrict__ idx2 /* (b,32,ns2) */,
    int *__restrict__ rep2 /* (b,32) */, const RgPacks pk /* .rowinfo1 == NULL: no row lists */)
{
    // hit lists of the running ball query, [slot][centre], as 16-bit point numbers (< 512): 16.5 KB.  With 32-bit entries the workgroup
    // held 36 KB of LDS -- FOUR single-wave workgroups per CU, one per SIMD, and 1600 RoI clouds took two rounds of a chain that is
    // latency-bound from end to end (250 us per 1600 clouds); at 21 KB seven fit and every cloud of a launch is resident at once
    __shared__ unsigned short s_hits[RG_NS * RG_LD1];
    __shared__ int s_sel1[RG_M1], s_sel2[RG_M2];
    __shared__ int s_first[RG_N];
    __shared__ int s_rep1[RG_M1];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *__restrict__ cloud = xyz + (long)b * RG_N * 3;
    const int lim = limit ? min(max(limit[b], 1), RG_N) : RG_N;
    __builtin_amdgcn_s_setprio(3);

    // ---- level 1: sample 128 of the 512 pooled points
    float px[8], py[8], pz[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = lane + 64 * i;
        px[i] = cloud[3 * k]; py[i] = cloud[3 * k + 1]; pz[i] = cloud[3 * k + 2];
    }
    const int nd1 = roi_fps_any<8, true>(RG_N, lim, RG_M1, kc1, px, py, pz, s_sel1, lane);
    __syncthreads();
    // the centres that are their own representatives are the first nd1 (distinct picks; what follows are copies of centre 0): listed as
    // rows b * 128 + c for the per-point layer of the level above (prcnn_rows_gemm128_rows), which nobody asks for the other rows
    if (pk.crows1) {
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&pk.hdr_c1[1], (unsigned int)nd1);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int c = lane; c < nd1; c += 64) pk.crows1[base + c] = b * RG_M1 + c;
    }
    // the sampled centres: coordinates into registers (centre c = lane + 64 q) and out to new_xyz1
    float qx[2], qy[2], qz[2];
    int src1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + 64 * q, k = s_sel1[c];
        qx[q] = cloud[3 * k]; qy[q] = cloud[3 * k + 1]; qz[q] = cloud[3 * k + 2];
        float *o = new_xyz1 + ((long)b * RG_M1 + c) * 3;
        o[0] = qx[q]; o[1] = qy[q]; o[2] = qz[q];
        src1[q] = k >= lim ? k % lim : k;                        // the distinct pooled point behind this centre
    }
    // ---- ball query of level 1 over the DISTINCT pooled points only (prcnn_ball_query_limit)
    int cnt1[2];
    {
        const bool live[2] = {true, true};
        roi_ball_query<8, 2>(lim, ns1, r1sq, px, py, pz, qx, qy, qz, live, s_hits, RG_LD1, cnt1, lane);
    }
    // representative map of the centres: the first centre sampled from the same source (prcnn_dup_rep)
#pragma unroll
    for (int i = 0; i < 8; ++i) s_first[lane + 64 * i] = 0x7fffffff;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) atomicMin(&s_first[src1[q]], lane + 64 * q);
    __syncthreads();
    int own1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = s_first[src1[q]];
        s_rep1[lane + 64 * q] = r;
        rep1[(long)b * RG_M1 + lane + 64 * q] = r;
        own1[q] = r == lane + 64 * q;
    }
    if (idx1) roi_rows_out<RG_M1>(ns1, s_hits, RG_LD1, cnt1[0], cnt1[1], idx1 + (long)b * RG_M1 * ns1, lane);
    if (pk.rowinfo1) {
        __syncthreads();                                          // s_first is free: the list's offsets
        roi_pack_out<RG_M1>(b, own1[0] ? max(cnt1[0], 1) : 0, own1[1] ? max(cnt1[1], 1) : 0, cnt1[0], cnt1[1], s_hits, RG_LD1, s_first, cloud,
                            nullptr, s_sel1, nullptr, pk.rowinfo1, pk.rowdxyz1, pk.tilecloud1, pk.hdr1, lane);
    }
    __syncthreads();                                              // s_hits is reused below

    // ---- level 2: sample 32 of the 128 centres (held in registers as points k = lane + 64 q), ball query over all 128.
    // Behind an exhausted level-1 scan (nd1 < 128 picks, then copies of point 0) only the first nd1 centres are distinct.
    roi_fps_any<2, false>(RG_M1, nd1, RG_M2, kc2, qx, qy, qz, s_sel2, lane);
    __syncthreads();
    float cx[1] = {0.f}, cy[1] = {0.f}, cz[1] = {0.f};
    const bool has = lane < RG_M2;
    const int src2 = has ? s_rep1[s_sel2[lane]] : 0;            // the first level-1 centre with the same source as this one's pick
    // centre coordinates of level 2 by a cross-lane read of the registers that hold the 128 points
    {
        const int k = has ? s_sel2[lane] : 0;
        const int ql = k & 63, qi = k >> 6;
        const float x0 = __shfl(qx[0], ql), x1 = __shfl(qx[1], ql);
        const float y0 = __shfl(qy[0], ql), y1 = __shfl(qy[1], ql);
        const float z0 = __shfl(qz[0], ql), z1 = __shfl(qz[1], ql);
        cx[0] = qi ? x1 : x0; cy[0] = qi ? y1 : y0; cz[0] = qi ? z1 : z0;
        if (has) {
            float *o = new_xyz2 + ((long)b * RG_M2 + lane) * 3;
            o[0] = cx[0]; o[1] = cy[0]; o[2] = cz[0];
        }
    }
    int cnt2[1], cntd2 = 0;
    {
        // the scan runs over the nd1 distinct centres; the centres behind them are copies of centre 0: in range together with it, and
        // then the next hits in index order
        const bool live[1] = {has};
        roi_ball_query<2, 1>(nd1, ns2, r2sq, qx, qy, qz, cx, cy, cz, live, s_hits, RG_LD2, cnt2, lane);
        cntd2 = has ? cnt2[0] : 0;                                // hits among the distinct centres: the rows the level's list keeps
        const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx[0]), 0));
        const float y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy[0]), 0));
        const float z0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz[0]), 0));
        if (has && sqdist3(cx[0], cy[0], cz[0], x0, y0, z0) < r2sq)
            for (int k = nd1; k < RG_M1 && cnt2[0] < ns2; ++k) {
                s_hits[cnt2[0] * RG_LD2 + lane] = (unsigned short)k;
                ++cnt2[0];
            }
    }
    // representative map of level 2's centres through the map of level 1
    for (int i = lane; i < RG_M1; i += 64) s_first[i] = 0x7fffffff;
    __syncthreads();
    if (has) atomicMin(&s_first[src2], lane);
    __syncthreads();
    const int r2own = has ? s_first[src2] : -1;
    if (has) rep2[(long)b * RG_M2 + lane] = r2own;
    if (idx2) roi_rows_out<RG_M2>(ns2, s_hits, RG_LD2, has ? cnt2[0] : 0, 0, idx2 + (long)b * RG_M2 * ns2, lane);
    if (pk.rowinfo1) {
        __syncthreads();
        roi_pack_out<RG_M2>(b, r2own == lane ? max(cntd2, 1) : 0, 0, cntd2, 0, s_hits, RG_LD2, s_first, cloud, s_sel1, s_sel1, s_sel2,
                            pk.rowinfo2, pk.rowdxyz2, pk.tilecloud2, pk.hdr2, lane);
    }
    // the list of the level ABOVE (rcnn_net.py's GroupAll module: one group of all 32 centres, no centre subtraction): a row per centre of
    // level 2 that is its own representative -- what prcnn_ball_pack_ex makes of the index rows 0 .. 31 around the origin with rep = rep2,
    // every cloud a list of its own centre 0
    if (pk.rowinfo3) {
        const unsigned long long own = __ballot(has && r2own == lane);
        const int n3 = __builtin_popcountll(own);
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&pk.hdr3[1], (unsigned int)n3);
        base = __builtin_amdgcn_readfirstlane(base);
        if (has && r2own == lane) {
            const int r = base + (int)__builtin_popcountll(own & ((1ull << lane) - 1ull));
            pk.rowinfo3[r] = ((unsigned int)b << 16) | (unsigned int)lane;
            pk.rowdxyz3[r] = make_float4(cx[0] - 0.f, cy[0] - 0.f, cz[0] - 0.f, 0.f);
        }
    }
}

}  // namespace prcnn

/* RoI clouds xyz (b,512,3) whose points k >= limit[cloud] are copies of point k % limit[cloud] (pooled RoI rows) ->
 *   new_xyz1 (b,128,3), idx1 (b,128,ns1), rep1 (b,128): furthest_point_sample(128) + ball_query(r1, ns1) over the distinct points
 *                                                        (= prcnn_fps_new_xyz, prcnn_ball_query_limit, prcnn_dup_rep with `limit`);
 *   new_xyz2 (b,32,3), idx2 (b,32,ns2), rep2 (b,32): the same one level up over the 128 centres (prcnn_fps_new_xyz, prcnn_ball_query
 *                                                      with empty balls written as zeros, prcnn_dup_rep with prev = rep1).
 * ns1, ns2 <= 64.  The shape of rcnn_net.py:165-175 under default.yaml (RCNN.NUM_POINTS 512, SA_CONFIG NPOINTS [128, 32, -1]). */
static int roi_geometry_any(int b, int n, int m1, float r1, int ns1, int m2, float r2, int ns2, const float *xyz,
                            const int *limit, float *new_xyz1, int *idx1, int *rep1, float *new_xyz2, int *idx2, int *rep2,
                            const prcnn::RgPacks *packs, void *stream)
{
    PRCNN_REQUIRE(b >= 0 && n == RG_N && m1 == RG_M1 && m2 == RG_M2, "rcnn_roi_geometry: written for 512 -> 128 -> 32 points (got %d -> %d -> %d)", n, m1, m2);
    PRCNN_REQUIRE(ns1 >= 1 && ns1 <= RG_NS && ns2 >= 1 && ns2 <= RG_NS && r1 > 0.f && r2 > 0.f, "rcnn_roi_geometry: nsample must be 1..64, radii positive");
    if (b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && new_xyz1 && rep1 && new_xyz2 && rep2, "rcnn_roi_geometry: null pointer");
    PRCNN_REQUIRE((idx1 && idx2) || (!idx1 && !idx2 && packs->rowinfo1), "rcnn_roi_geometry: the index tensors may only be left out (both) when the row lists are asked for");
    auto codec = [](int npts) {
        const int bs = host_opt_n_threads(npts);
        KeyCodec kc;
        kc.hipcc = g_fps_hipcc;
        kc.log2bs = 0;
        while ((1 << kc.log2bs) < bs) ++kc.log2bs;
        const int nq = (npts + bs - 1) / bs;
        kc.sh = 0;
        while ((1 << kc.sh) < nq) ++kc.sh;
        return kc;
    };
    hipLaunchKernelGGL(rcnn_roi_geometry_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, codec(RG_N), codec(RG_M1), r1 * r1, r2 * r2,
                       ns1, ns2, xyz, limit, new_xyz1, idx1, rep1, new_xyz2, idx2, rep2, *packs);
    return check_launch("rcnn_roi_geometry");
}

extern "C" int prcnn_rcnn_roi_geometry(int b, int n, int m1, float r1, int ns1, int m2, float r2, int ns2, const float *xyz,
                                       const int *limit, float *new_xyz1, int *idx1, int *rep1, float *new_xyz2, int *idx2, int *rep2,
                                       void *stream)
{
    const prcnn::RgPacks none = {};
    return roi_geometry_any(b, n, m1, r1, ns1, m2, r2, ns2, xyz, limit, new_xyz1, idx1, rep1, new_xyz2, idx2, rep2, &none, stream);
}

/* prcnn_rcnn_roi_geometry + the distinct-row lists of both levels in the same launch (round 5):
 *   list 1 = prcnn_ball_pack_ex(b, b, 512, 128, ns1, idx1, limit, NULL, rep1, xyz, new_xyz1, ...),
 *   list 2 = prcnn_ball_pack_ex(b, b, 128, 32, ns2, idx2, NULL, rep1, rep2, new_xyz1, new_xyz2, ...)
 * -- the same rows per cloud in the same order, cut into the same tiles (the order of the CLOUDS' tiles in a list is whatever the
 * counter hands out, as it is for prcnn_ball_pack).  rowinfo* / rowdxyz* / tilecloud*: sized as for prcnn_ball_pack
 * (b * ceil(m * ns / 64) tiles); hdr1 / hdr2 (4 u32 each): zeroed here unless hdr_is_zero.  tilecloud1 == tilecloud2 == NULL: lists
 * whose rows carry their cloud (see RgPacks; the form the engine uses: prcnn_sa_packed_mlp reads it).  idx1 == idx2 == NULL: the index tensors
 * are not written (a caller that feeds the row lists to the packed MLP kernels has no use for them: 10240 words per cloud).
 * rowinfo3 / rowdxyz3 / hdr3 (optional, b * m2 rows at most): the list of the GroupAll level above -- every cloud one group (centre 0) of
 * its m2 level-2 centres, the centres that copy an earlier one dropped: prcnn_ball_pack_ex(b, b, m2, 1, m2, {0..m2-1}, NULL, rep2, NULL,
 * new_xyz2, origin, ...) in the row-carried form.
 * crows1 / hdr_c1 (optional, b * m1 entries at most): the level-1 centres that are their own representatives as rows cloud * m1 + centre,
 * hdr_c1[1] of them -- for prcnn_rows_gemm128_rows (the per-point layer of level 2 over exactly the rows its lists name). */
extern "C" int prcnn_rcnn_roi_geometry_packs(int b, int n, int m1, float r1, int ns1, int m2, float r2, int ns2, const float *xyz,
                                             const int *limit, float *new_xyz1, int *idx1, int *rep1, float *new_xyz2, int *idx2, int *rep2,
                                             unsigned int *rowinfo1, float *rowdxyz1, int *tilecloud1, unsigned int *hdr1,
                                             unsigned int *rowinfo2, float *rowdxyz2, int *tilecloud2, unsigned int *hdr2,
                                             unsigned int *rowinfo3, float *rowdxyz3, unsigned int *hdr3, int *crows1, unsigned int *hdr_c1,
                                             int hdr_is_zero, void *stream)
{
    PRCNN_REQUIRE(hdr1 && hdr2, "rcnn_roi_geometry_packs: null header");
    PRCNN_REQUIRE((crows1 != nullptr) == (hdr_c1 != nullptr), "rcnn_roi_geometry_packs: the centre rows come with their header");
    if (hdr_c1 && !hdr_is_zero && hipMemsetAsync(hdr_c1, 0, 4 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess) {
        set_error("rcnn_roi_geometry_packs: memset failed");
        return PRCNN_ELAUNCH;
    }
    PRCNN_REQUIRE((rowinfo3 != nullptr) == (rowdxyz3 != nullptr) && (rowinfo3 != nullptr) == (hdr3 != nullptr) && (!rowinfo3 || !tilecloud1),
                  "rcnn_roi_geometry_packs: the third list comes whole, and only with lists whose rows carry their cloud");
    if (!hdr_is_zero && (hipMemsetAsync(hdr1, 0, 4 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess ||
                         hipMemsetAsync(hdr2, 0, 4 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess ||
                         (hdr3 && hipMemsetAsync(hdr3, 0, 4 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess))) {
        set_error("rcnn_roi_geometry_packs: memset failed");
        return PRCNN_ELAUNCH;
    }
    if (b == 0) return PRCNN_OK;
    PRCNN_REQUIRE(rowinfo1 && rowdxyz1 && rowinfo2 && rowdxyz2, "rcnn_roi_geometry_packs: null pointer");
    PRCNN_REQUIRE((tilecloud1 && tilecloud2) || (!tilecloud1 && !tilecloud2 && b <= 65536), "rcnn_roi_geometry_packs: both lists with a tilecloud or none");
    PRCNN_REQUIRE((((uintptr_t)rowdxyz1 | (uintptr_t)rowdxyz2 | (uintptr_t)rowdxyz3) & 15) == 0, "rcnn_roi_geometry_packs: rowdxyz must be 16-byte aligned");
    const prcnn::RgPacks pk = {rowinfo1, (float4 *)rowdxyz1, tilecloud1, hdr1, rowinfo2, (float4 *)rowdxyz2, tilecloud2, hdr2,
                               rowinfo3, (float4 *)rowdxyz3, hdr3, crows1, hdr_c1};
    return roi_geometry_any(b, n, m1, r1, ns1, m2, r2, ns2, xyz, limit, new_xyz1, idx1, rep1, new_xyz2, idx2, rep2, &pk, stream);
}
