// Pins the arithmetic of v_mfma_f32_16x16x4_f32 on the MI355X for a 128-deep chain (round 5: the 12 real columns 64..75 of the RPN
// regression head run on this instruction instead of a zero-padded 32x32x2 block).  k of step s, lane group kk = lane / 16: 32 kk + s.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o profiles/_exp/mfma16_probe profiles/mfma16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *A /*16x128*/, const float *B /*128x16*/, float *D /*16x16*/)
{
    const int l = threadIdx.x, n = l & 15, kk = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 32; ++s) {
        const float a = A[(l & 15) * 128 + 32 * kk + s];
        const float b = B[(32 * kk + s) * 16 + n];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * kk + r) * 16 + n] = acc[r];
}
int main()
{
    std::vector<float> A(16 * 128), B(128 * 16), D(256);
    srand(5);
    int bad[6] = {0, 0, 0, 0, 0, 0};
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto &v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (trial % 3 == 0 ? 100.f : 1.f);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float c[6];
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)A[i * 128 + k] * B[k * 16 + j];
                // 0: fma chain kk = 0..3 per step; 1: kk = 3..0; 2: pairs (0,1) then (2,3) summed apart; 3: products added unfused in order
                float a0 = 0, a1 = 0, a3 = 0; double a2 = 0; float p01, p23;
                float a4 = 0, a5 = 0;
                for (int s = 0; s < 32; ++s) {
                    for (int kk = 0; kk < 4; ++kk) a0 = fmaf(A[i * 128 + 32 * kk + s], B[(32 * kk + s) * 16 + j], a0);
                    for (int kk = 3; kk >= 0; --kk) a1 = fmaf(A[i * 128 + 32 * kk + s], B[(32 * kk + s) * 16 + j], a1);
                    for (int kk = 0; kk < 4; ++kk) a3 = a3 + A[i * 128 + 32 * kk + s] * B[(32 * kk + s) * 16 + j];
                    // 4: dot4 exact then one rounding into acc
                    double d4 = 0;
                    for (int kk = 0; kk < 4; ++kk) d4 += (double)A[i * 128 + 32 * kk + s] * B[(32 * kk + s) * 16 + j];
                    a4 = (float)((double)a4 + d4);
                    // 5: pairs: fma(a1,b1, a0*b0 exact?) -> t = fma(x1,y1, x0*y0) ; acc += t ...
                    p01 = fmaf(A[i * 128 + 32 + s], B[(32 + s) * 16 + j], A[i * 128 + s] * B[s * 16 + j]);
                    p23 = fmaf(A[i * 128 + 96 + s], B[(96 + s) * 16 + j], A[i * 128 + 64 + s] * B[(64 + s) * 16 + j]);
                    a5 = a5 + (p01 + p23);
                }
                (void)a2;
                c[0] = a0; c[1] = a1; c[2] = (float)ref; c[3] = a3; c[4] = a4; c[5] = a5;
                const float got = D[i * 16 + j];
                if (fabs(got - ref) > 1e-3 * (1 + fabs(ref))) { printf("LAYOUT WRONG trial %d (%d,%d): got %g ref %g\n", trial, i, j, got, ref); return 1; }
                for (int q = 0; q < 6; ++q) bad[q] += memcmp(&got, &c[q], 4) != 0;
            }
    }
    printf("mismatches of 51200 outputs: fma chain kk=0..3: %d | kk=3..0: %d | f64 dot rounded: %d | unfused in order: %d | exact dot4 per step: %d | pair sums: %d\n",
           bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    return 0;
}
