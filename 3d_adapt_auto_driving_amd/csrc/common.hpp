// common.hpp -- shared helpers for the gfx950 kernels of libprcnn_hip.so.
// Wave = 64 lanes everywhere (CDNA4); arithmetic that decides an index is compiled with
// -ffp-contract=off so that it matches the CPU oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/prcnn_hip.h"

namespace prcnn {

constexpr int WAVE = 64;

void set_error(const char *fmt, ...);
int check_launch(const char *what);

#define PRCNN_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            prcnn::set_error(__VA_ARGS__);       \
            return PRCNN_EINVAL;                 \
        }                                        \
    } while (0)

int current_device();
// raise a kernel's dynamic-LDS limit to `bytes` (> 64 KiB needs it) once per (device, kernel)
int ensure_dynamic_lds(const void *kernel, size_t bytes, const char *what);

// cached device scratch, one per (device, stream) (grown on demand; hipMalloc only on growth).  `slot` separates
// independent users (0: ball-query grid, 1: FPS ordering).
char *scratch_for(hipStream_t st, size_t bytes, int slot = 0, bool *fresh = nullptr);   // *fresh: the buffer is new (or changed owner): contents undefined

// Tile tickets of the persistent kernels: a PAIR of words per launch, [0] the draw counter, [1] the workgroups that have drawn
// their last ticket.  Thread 0 of every workgroup calls this once, after its last draw (on every exit path): the last one to arrive
// zeroes the pair, so the words are clean for their next user -- no fill in front of a launch, none inside a captured hipGraph
// (a memset node per ticketed launch cost the graph replay ~4 % of a step).  Launches that share a pair are ordered by their stream.
// No fence: the thread has CONSUMED the value of its last draw before it gets here (it decided to leave on it), both counters are
// only ever touched by device-scope atomics, and a kernel boundary orders the reset against the next launch.  (An agent-scope fence
// here is an L2 write-back per workgroup on a multi-XCD part: it cost rpn_tail_lin_kernel 10 us of 181.)
__device__ __forceinline__ void ticket_release(unsigned int *ticket)
{
    if (atomicAdd(ticket + 1, 1u) == gridDim.x * gridDim.y * gridDim.z - 1u) {
        atomicExch(ticket, 0u);
        atomicExch(ticket + 1, 0u);
    }
}

// The same with one draw counter per XCD (words [2..9] of the launch's 16-word ticket record): a workgroup draws from the counter of
// its own XCD first -- block b runs on XCD b % 8 (observed placement; a different one costs locality, never correctness) -- so that
// the tiles of one eighth of the problem, and the tables they gather from, stay in ONE L2 (the 8 L2s are 4 MB each and private).
struct XcdTickets {
    unsigned int *rec;        // the 16-word record
    unsigned int chunk;       // tiles per partition
    unsigned int tiles;
    unsigned int xcd, dead;   // this workgroup's XCD; how many partitions (starting at its own) it has found empty
    __device__ __forceinline__ unsigned int draw()      // thread 0 only; 0x7fffffff = nothing left anywhere
    {
        while (dead < 8u) {
            const unsigned int y = (xcd + dead) & 7u;
            const unsigned int k = atomicAdd(rec + 2 + y, 1u);
            const unsigned long tile = (unsigned long)y * chunk + k;
            if (k < chunk && tile < tiles) return (unsigned int)tile;
            ++dead;
        }
        return 0x7fffffffu;
    }
    // thread 0 only, no wait on the result: the next ticket of the partition the workgroup is on, as a tile index -- or a value
    // >= 0x40000000 when that partition has run out (the reader then calls draw(), which moves on to the next partition)
    __device__ __forceinline__ unsigned int issue()
    {
        if (dead >= 8u) return 0x7fffffffu;
        const unsigned int y = (xcd + dead) & 7u;
        const unsigned int k = atomicAdd(rec + 2 + y, 1u);
        const unsigned int lim = min(chunk, tiles > y * chunk ? tiles - y * chunk : 0u);      // tiles of partition y
        return k < lim ? y * chunk + k : 0x40000000u;                                        // (a select, not a branch)
    }
    __device__ __forceinline__ void release()           // thread 0 of every workgroup, once, after its last draw
    {
        if (atomicAdd(rec + 1, 1u) == gridDim.x * gridDim.y * gridDim.z - 1u) {
#pragma unroll
            for (int i = 0; i < 10; ++i) atomicExch(rec + i, 0u);
        }
    }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier, and the fence drains vmcnt as
// well: every wave then waits at the barrier for its outstanding GLOBAL loads and stores (an HBM round trip), which
// serialises "store this tile / prefetch the next tile" against the LDS hand-off between pipeline phases.  Use this
// where the only data exchanged between the waves goes through LDS.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Workgroups of a persistent (ticket-drawing) MFMA launch: PRCNN_MFMA_GRID, default in capi.hip
int mfma_grid_cap();

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// (a-b)^2 summed left to right, one rounding per operation (no fma): the distance form of
// ball_query_gpu.cu:33, sampling_gpu.cu:133, interpolate_gpu.cu:37.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// Interpolation weights of PointnetFPModule.forward (pointnet2_modules.py:139-144) from the three squared distances of three_nn
// (pointnet2_utils.py:97: the Python wrapper returns sqrt(dist2)): r_k = 1 / (sqrt(d2_k) + 1e-8), w_k = r_k / ((r_0 + r_1) + r_2).
// One rounding per operation (sqrtf -- NOT __fsqrt_rn, which compiles to the bare 1-ulp v_sqrt_f32 --, add, divide: all correctly rounded) -- the sequence torch executes as five elementwise /
// reduction launches per FP level; here it is the epilogue of the three_nn kernels (prcnn_three_nn_weights).
__device__ __forceinline__ void three_nn_weights(float d0, float d1, float d2, float *__restrict__ w)
{
    const float r0 = __fdiv_rn(1.0f, __fadd_rn(sqrtf(d0), 1e-8f));
    const float r1 = __fdiv_rn(1.0f, __fadd_rn(sqrtf(d1), 1e-8f));
    const float r2 = __fdiv_rn(1.0f, __fadd_rn(sqrtf(d2), 1e-8f));
    const float s = __fadd_rn(__fadd_rn(r0, r1), r2);
    w[0] = __fdiv_rn(r0, s); w[1] = __fdiv_rn(r1, s); w[2] = __fdiv_rn(r2, s);
}

// f32 trig contract: correctly rounded from the f64 value (see oracle/prcnn_oracle.h).
__device__ __forceinline__ float cos_f32(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float sin_f32(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float atan2_f32(float y, float x) { return (float)atan2((double)y, (double)x); }

// v_min_f32 / v_max_f32 as ONE instruction each.  fminf / fmaxf are llvm.minnum / maxnum, and for an operand that is not provably the
// result of an arithmetic instruction (a register-resident running minimum, a v_readlane, a bitcast) the backend puts a canonicalising
// `v_max_f32 x, x, x` in front -- sNaN quieting the hardware instruction does by itself in the IEEE mode compute kernels run in: 944 of the
// 6435 instructions of fps_spec_kernel<16> were that, 164 of the 2500 of sa_packed_mlp128_kernel (the pooling epilogues' maxima over MFMA results).  Same results for every input including quiet NaNs (the non-NaN operand is returned).
// PRECONDITIONS: (1) the kernel runs in IEEE mode (the default of every HIP compute kernel; `amdgpu-ieee=false` is not used
// anywhere in this build): that is what quiets a signalling NaN inside v_min / v_max -- for an sNaN operand the instruction then returns
// the QUIETED NaN where canonicalize + minnum would return the other operand, the one input class on which the two forms differ; no caller
// feeds one (operands are distances, running minima initialised to 1e10 / +-inf, MFMA results); (2) gfx9 encodings: this library is built
// for gfx950 only (csrc/Makefile), the #error below keeps it that way.  The asm is not volatile on purpose (pure function of its inputs:
// the compiler may CSE and schedule it); it hides the operation from constant folding, which no call site needs.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "fmin_raw / fmax_raw are written for the gfx9 VOP2 encodings (this library targets gfx950)"
#endif
__device__ __forceinline__ float fmin_raw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmax_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Packed f32 arithmetic: two components per v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.  Every component sees exactly the
// operation the scalar form would apply (one rounding each), so results are bit-identical; what changes is the VALU
// INSTRUCTION count of the tile builders, and a builder that shares its SIMD with a neighbour's MFMA stream waits for a
// slot between two MFMAs per instruction (profiles/r02_stage_stamps.md).
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
// acc + w * s per component, as fma(w, s, acc)
__device__ __forceinline__ float4 pk_fma4(const float4 w, float s, const float4 acc)
{
    const pk_f32x2 ss = {s, s};
    const pk_f32x2 lo = __builtin_elementwise_fma((pk_f32x2){w.x, w.y}, ss, (pk_f32x2){acc.x, acc.y});
    const pk_f32x2 hi = __builtin_elementwise_fma((pk_f32x2){w.z, w.w}, ss, (pk_f32x2){acc.z, acc.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 relu4(const float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
// relu(fma(wz, dz, fma(wy, dy, fma(wx, dx, base)))) per component: layer 1 of an SA scale applied to a neighbour offset
__device__ __forceinline__ float4 affine_relu4(const float4 base, const float4 wx, const float4 wy, const float4 wz, float dx, float dy, float dz)
{
    return relu4(pk_fma4(wz, dz, pk_fma4(wy, dy, pk_fma4(wx, dx, base))));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace prcnn
