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
char *scratch_for(hipStream_t st, size_t bytes, int slot = 0);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier, and the fence drains vmcnt as
// well: every wave then waits at the barrier for its outstanding GLOBAL loads and stores (an HBM round trip), which
// serialises "store this tile / prefetch the next tile" against the LDS hand-off between pipeline phases.  Use this
// where the only data exchanged between the waves goes through LDS.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// (a-b)^2 summed left to right, one rounding per operation (no fma): the distance form of
// ball_query_gpu.cu:33, sampling_gpu.cu:133, interpolate_gpu.cu:37.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// f32 trig contract: correctly rounded from the f64 value (see oracle/prcnn_oracle.h).
__device__ __forceinline__ float cos_f32(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float sin_f32(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float atan2_f32(float y, float x) { return (float)atan2((double)y, (double)x); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace prcnn
