// Do MFMA and VALU work overlap on one SIMD when they come from DIFFERENT waves?  Half of the waves of every
// SIMD issue only v_mfma_f32_32x32x16_f16, the other half only v_min3_f32 (8 per MFMA of the partner).
// If the pair takes max(30, 34) cycles per step the pipes overlap across waves; if 64, they serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP8(X) X X X X X X X X
template <int MODE>   // 0: every wave does both (mfma + 8 min3 per step); 1: waves alternate roles (2 mfma | 16 min3 per step)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 d0 = {}, d1 = {};
    const f32x16 zero = {};
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 2.0\n v_mov_b32 v103, 0" ::: "v100","v101","v102","v103");
    // 512 threads = 8 waves = 2 per SIMD: waves w and w+4 share a SIMD (round-robin placement)
    const bool mfma_role = MODE == 0 || ((threadIdx.x >> 6) < 4);
    const bool valu_role = MODE == 0 || ((threadIdx.x >> 6) >= 4);
    for (int it = 0; it < iters; ++it) {
        if (mfma_role) { d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0); if (MODE == 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0); }
        if (valu_role) { asm volatile(REP8("v_min3_f32 v103, v100, v101, v102\n") ::: "v103"); if (MODE == 1) asm volatile(REP8("v_min3_f32 v103, v100, v101, v102\n") ::: "v103"); }
        asm volatile("" : "+v"(d0), "+v"(d1));
    }
    float r; asm volatile("v_mov_b32 %0, v103" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + r;
}
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode) for (int blocks : {256, 512, 1024}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 512>>>(d, iters); else k<1><<<blocks, 512>>>(d, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        // both modes execute, per SIMD and iteration, 2 MFMAs and 16 min3 in total (2 waves per SIMD per block)
        double steps = (double)blocks * 4 * iters * 2;
        printf("%s blocks %4d: %.3f ms, %.1f cycles per (mfma + 8 min3) per SIMD @2.4GHz\n", mode == 0 ? "same wave does both " : "roles split by wave", blocks, ms,
               1024.0 * 2.4e9 * ms * 1e-3 / steps);
    }
    return 0;
}
