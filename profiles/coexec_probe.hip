// coexec_probe.hip -- do f32 MFMAs of one wave and f32 VALU work of ANOTHER wave of the same SIMD overlap on gfx950?
// One workgroup per CU, 8 waves = 2 per SIMD: waves 0-3 stream v_mfma_f32_32x32x2_f32 (two independent accumulators), waves 4-7 do
// mode 0: nothing | 1: v_fma_f32 chains | 2: v_pk_fma_f32 chains | 3: the same MFMA stream | 4: LDS reads | 5: v_mfma_f32_32x32x16_bf16.
// Prints the launch time per mode, and of the partner work alone.  build: hipcc --offload-arch=gfx950 -O3 -o profiles/_exp/coexec_probe
// profiles/coexec_probe.hip   (why co-resident MFMA kernels "take turns")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void probe(int mode, int mfma_on, int iters, float *out)
{
    __shared__ float lds[4096];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
    __syncthreads();
    float res = 0.f;
    if (w < 4) {
        if (mfma_on) {
            f32x16 a0 = {0}, a1 = {0};
            float x = (float)lane, y = 1.0f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                }
            }
            res = a0[0] + a1[0];
        }
    } else if (mode == 1) {
        float c[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        const float m = 1.0001f, a = 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[k]) : "v"(m), "v"(a));   // 128 per iteration (the compiler would pair them into v_pk_fma_f32)
        }
        for (int k = 0; k < 8; ++k) res += c[k];
    } else if (mode == 2) {
        f32x2 c[8];
        for (int k = 0; k < 8; ++k) c[k] = f32x2{(float)k, (float)lane};
        const f32x2 m = {1.0001f, 1.0002f}, a = {0.5f, 0.25f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = __builtin_elementwise_fma(c[k], m, a);   // 128 v_pk_fma_f32
        }
        for (int k = 0; k < 8; ++k) res += c[k].x + c[k].y;
    } else if (mode == 3) {
        f32x16 a0 = {0}, a1 = {0};
        float x = (float)lane, y = 1.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
        res = a0[0] + a1[0];
    } else if (mode == 4) {
        float s = 0.f;
        int p = lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { s += lds[(p + 64 * u) & 4095]; }
            p = (p + 1) & 4095;
        }
        res = s;
    } else if (mode == 5) {
        f32x16 a0 = {0}, a1 = {0};
        bf16x8 x, y;
        for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(float)(lane + k); y[k] = (__bf16)1.0f; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            }
        }
        res = a0[0] + a1[0];
    }
    if (res == 123.456f) out[threadIdx.x] = res;
}

// the same question INSIDE one wave: K independent v_fma_f32 (or LDS reads) behind every MFMA of the stream, waves 4-7 idle
template <int K, int LDS>
__global__ __launch_bounds__(512) void probe_same(int iters, float *out, const float *src)
{
    __shared__ float lds[4096];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
    __syncthreads();
    float res = 0.f;
    if (w < 4) {
        f32x16 a0 = {0}, a1 = {0};
        float x = (float)lane, y = 1.0f;
        float c[16];
        for (int k = 0; k < 16; ++k) c[k] = (float)(k + lane);
        const float m = 1.0001f, a = 0.5f;
        int p = lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (LDS == 1) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(4 * ((p + 64 * k) & 4095))); asm volatile("" :: "v"(v)); }
                    else if (LDS == 2) { float v; asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(4 * ((p + 64 * k) & 4095)), "s"(src)); asm volatile("" :: "v"(v)); }
                    else if (LDS == 3) { int sv; asm volatile("s_add_u32 %0, %1, 1" : "=s"(sv) : "s"(i) : "scc"); asm volatile("" :: "s"(sv)); }
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[k]) : "v"(m), "v"(a));
                }
            }
            if (LDS == 1) asm volatile("s_waitcnt lgkmcnt(0)");
            if (LDS == 2) asm volatile("s_waitcnt vmcnt(0)");
        }
        res = a0[0] + a1[0];
        for (int k = 0; k < 16; ++k) res += c[k];
    }
    if (res == 123.456f) out[threadIdx.x] = res;
}

// ONE accumulator per wave (dependent MFMAs back to back: sa_wide_fused / 32-row units): waves 0-3 alone, or waves 0-7 (two chains per SIMD)
__global__ __launch_bounds__(512) void probe_chain(int nwaves, int iters, float *out)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float res = 0.f;
    if (w < nwaves) {
        f32x16 a0 = {0};
        float x = (float)lane, y = 1.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        }
        res = a0[0];
    }
    if (res == 123.456f) out[threadIdx.x] = res;
}

static float run_chain(int nwaves, int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe_chain, dim3(256), dim3(512), 0, 0, nwaves, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe_chain, dim3(256), dim3(512), 0, 0, nwaves, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.f * 1000.f;
}

template <int K, int LDS>
static float run_same(int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe_same<K, LDS>), dim3(256), dim3(512), 0, 0, iters, out, out + 1024);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe_same<K, LDS>), dim3(256), dim3(512), 0, 0, iters, out, out + 1024);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.f * 1000.f;
}

static float run(int mode, int mfma_on, int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, mfma_on, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, mfma_on, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.f * 1000.f;
}

int main()
{
    float *out;
    hipMalloc(&out, 4096 + 4096 * 4 + 1024);
    const int iters = 2000;        // 16 MFMAs x 64 cycles = 1024 pipe cycles per iteration
    const char *names[] = {"nothing", "v_fma_f32 (128 per iteration)", "v_pk_fma_f32 (128 per iteration)", "the same f32 MFMA stream",
                           "LDS reads (32 per iteration)", "v_mfma_f32_32x32x16_bf16 (16 per iteration)"};
    printf("| waves 4-7 of every CU do | with the f32 MFMA stream on waves 0-3 (us) | alone (us) | sum if serialised (us) |\n|---|---|---|---|\n");
    const float t_mfma = run(0, 1, iters, out);
    for (int mode = 0; mode <= 5; ++mode) {
        const float both = run(mode, 1, iters, out);
        const float alone = mode == 0 ? 0.f : run(mode, 0, iters, out);
        printf("| %s | %.1f | %.1f | %.1f |\n", names[mode], both, alone, t_mfma + alone);
    }
    printf("\n| 16 MFMAs per iteration on ONE accumulator (dependent chain) | us |\n|---|---|\n");
    printf("| one wave per SIMD | %.1f |\n| two waves per SIMD (twice the MFMAs) | %.1f |\n", run_chain(4, iters, out), run_chain(8, iters, out));
    printf("\n| instructions of the SAME wave behind every MFMA (waves 4-7 idle) | v_fma_f32 (us) | ds_read_b32 (us) | global_load_dword (us) | s_add_u32 (us) |\n|---|---|---|---|---|\n");
    printf("| 0 | %.1f | %.1f | %.1f | %.1f |\n", run_same<0, 0>(iters, out), run_same<0, 1>(iters, out), run_same<0, 2>(iters, out), run_same<0, 3>(iters, out));
    printf("| 1 | %.1f | %.1f | %.1f | %.1f |\n", run_same<1, 0>(iters, out), run_same<1, 1>(iters, out), run_same<1, 2>(iters, out), run_same<1, 3>(iters, out));
    printf("| 2 | %.1f | %.1f | %.1f | %.1f |\n", run_same<2, 0>(iters, out), run_same<2, 1>(iters, out), run_same<2, 2>(iters, out), run_same<2, 3>(iters, out));
    printf("| 3 | %.1f | %.1f | %.1f | %.1f |\n", run_same<3, 0>(iters, out), run_same<3, 1>(iters, out), run_same<3, 2>(iters, out), run_same<3, 3>(iters, out));
    printf("| 4 | %.1f | %.1f | %.1f | %.1f |\n", run_same<4, 0>(iters, out), run_same<4, 1>(iters, out), run_same<4, 2>(iters, out), run_same<4, 3>(iters, out));
    printf("| 6 | %.1f | %.1f | %.1f | %.1f |\n", run_same<6, 0>(iters, out), run_same<6, 1>(iters, out), run_same<6, 2>(iters, out), run_same<6, 3>(iters, out));
    printf("| 8 | %.1f | %.1f | %.1f | %.1f |\n", run_same<8, 0>(iters, out), run_same<8, 1>(iters, out), run_same<8, 2>(iters, out), run_same<8, 3>(iters, out));
    printf("| 12 | %.1f | %.1f | %.1f | %.1f |\n", run_same<12, 0>(iters, out), run_same<12, 1>(iters, out), run_same<12, 2>(iters, out), run_same<12, 3>(iters, out));
    printf("| 16 | %.1f | %.1f | %.1f | %.1f |\n", run_same<16, 0>(iters, out), run_same<16, 1>(iters, out), run_same<16, 2>(iters, out), run_same<16, 3>(iters, out));
    return 0;
}
