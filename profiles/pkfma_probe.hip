// What does a v_pk_fma_f32 cost on gfx950, with a VGPR pair or an SGPR pair as the weight operand, against v_fma_f32?
// One workgroup of 256 threads per CU x 2 per CU, ITER iterations of 64 independent instructions per lane (8 accumulator pairs x 8).
// build: hipcc --offload-arch=gfx950 -O3 -o pkfma_probe profiles/pkfma_probe.hip ; run on the box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *w, int iters)
{
    f2 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f2){(float)threadIdx.x, (float)j};
    const f2 a = {out[threadIdx.x & 7], out[threadIdx.x & 7]};
    float s0 = w[0], s1 = w[1];
    s0 = __builtin_amdgcn_readfirstlane(s0); s1 = __builtin_amdgcn_readfirstlane(s1);
    const f2 wv = {out[1 + (threadIdx.x & 3)], out[2]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wv), "v"(a));
                else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "s"((f2){s0, s1}), "v"(a));
                else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(wv.x), "v"(a.x)); }
            }
    }
    float t = 0;
    for (int j = 0; j < 8; ++j) t += acc[j].x + acc[j].y;
    out[blockIdx.x * 256 + threadIdx.x + 16] = t;
}
int main()
{
    float *out, *w;
    hipMalloc(&out, 4 * (512 * 256 + 64)); hipMalloc(&w, 64);
    hipMemset(out, 0, 4 * (512 * 256 + 64)); hipMemset(w, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char *names[3] = {"v_pk_fma_f32, VGPR pair weight", "v_pk_fma_f32, SGPR pair weight (op_sel_hi as the compiler emits)", "v_fma_f32"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, out, w, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, out, w, iters);
            else hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, out, w, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                const double instr_per_simd = 2.0 * iters * 64;       // two waves per SIMD
                printf("%-70s %.3f ms: %.2f cycles per instruction and SIMD at 2.4 GHz\n", names[mode], ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
            }
        }
    }
    return 0;
}
