// Micro-benchmark: v_mfma_f32_16x16x4_f32 issue rate, alone and with VALU work between issues.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x4 d[8]; float m[8];
    const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = zero; m[i] = threadIdx.x; }
    float bb = b + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (VALU >= 1) { f32x4 v = d[i]; float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3])); m[i] = fminf(m[i], mn); }
            if (VALU >= 2) { m[i] = __builtin_fmaf(m[i], a, b); m[i] = __builtin_fmaf(m[i], a, b); m[i] = __builtin_fmaf(m[i], a, b); }
            d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + i, bb, zero, 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += d[i][0] + d[i][1] + d[i][2] + d[i][3] + m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 3; ++mode) for (int blocks : {256, 512, 1024, 2048}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double mf = (double)blocks * 4 * iters * 8;          // MFMAs
        printf("valu-mode %d blocks %d: %.3f ms, %.1f G mfma/s, %.1f TFLOP/s, %.1f cycles/mfma/SIMD @2.4GHz\n", mode, blocks, ms,
               mf / ms / 1e6, mf * 2048 / ms / 1e9, 1024.0 * 2.4e9 * ms * 1e-3 / mf);
    }
    return 0;
}
