// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950, alone and next to fp64 VALU work of the same wave / of other waves.
// Decides whether the RANSAC refit moments (a [hyps x n] 0/1 mask times an [n x 16] feature matrix) belong on the
// fp64 matrix pipe: one MFMA = 16 hyps x 16 features x 4 correspondences = 1024 FMA = 16 wave-wide v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define REP4(X) X X X X
template <int MODE>   // 0: mfma only (4 independent accumulators)  1: v_fma_f64 only (16 per step)  2: mfma + 16 v_fma_f64 per step
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double a = threadIdx.x * 0.001, b = 0.5 + threadIdx.x * 1e-6;
    f64x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = i * 0.25;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d3, 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < (MODE == 1 ? 4 : 4); ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], b, a);
        }
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    }
    double s = d0[0] + d1[1] + d2[2] + d3[3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; (void)hipMalloc(&d, 8192 * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) for (int wps : {1, 2, 4}) {       // waves per SIMD
        float ms = 0;
        const int blocks = 256 * wps;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters); else if (mode == 1) k<1><<<blocks, 256>>>(d, iters); else k<2><<<blocks, 256>>>(d, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        // per SIMD: wps waves x iters steps; a step = 4 mfma and/or 64 v_fma_f64
        double steps = (double)wps * iters;
        printf("mode %d (%s) waves/SIMD %d: %.3f ms, %.1f cycles per step (4 mfma_f64 | 64 v_fma_f64) @2.4GHz\n", mode,
               mode == 0 ? "mfma only" : mode == 1 ? "fma only" : "both", wps, ms, 2.4e9 * ms * 1e-3 / steps);
    }
    return 0;
}
