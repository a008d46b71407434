// Does VALU work overlap with v_mfma_f32_32x32x16_f16 on one SIMD?  Per iteration: NM independent MFMAs
// (4 accumulators, C = 0) and NV VALU ops of a given kind on registers the MFMAs do not touch.
// Built with -mllvm -amdgpu-mfma-vgpr-form=1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP4(X) X X X X
#define REP8(X) X X X X X X X X
template <int MODE>   // 0: mfma only; 1: + 8 v_min3 per mfma; 2: + 8 v_fma per mfma; 3: 8 v_min3 only; 4: 8 v_fma only; 5: + 16 v_fma per mfma; 6: + 4 min3
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    const f32x16 zero = {};
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 2.0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0" ::: "v100","v101","v102","v103","v104","v105","v106");
    for (int it = 0; it < iters; ++it) {
#define VALU_MIN8 asm volatile(REP8("v_min3_f32 v103, v100, v101, v102\n") ::: "v103");
#define VALU_MIN4 asm volatile(REP4("v_min3_f32 v103, v100, v101, v102\n") ::: "v103");
#define VALU_FMA8 asm volatile(REP8("v_fma_f32 v104, v100, v101, v102\n") ::: "v104");
        if (MODE != 3 && MODE != 4) d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
        if (MODE == 1 || MODE == 3) { VALU_MIN8 } if (MODE == 2 || MODE == 4 || MODE == 5) { VALU_FMA8 } if (MODE == 5) { VALU_FMA8 } if (MODE == 6) { VALU_MIN4 }
        if (MODE != 3 && MODE != 4) d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
        if (MODE == 1 || MODE == 3) { VALU_MIN8 } if (MODE == 2 || MODE == 4 || MODE == 5) { VALU_FMA8 } if (MODE == 5) { VALU_FMA8 } if (MODE == 6) { VALU_MIN4 }
        if (MODE != 3 && MODE != 4) d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
        if (MODE == 1 || MODE == 3) { VALU_MIN8 } if (MODE == 2 || MODE == 4 || MODE == 5) { VALU_FMA8 } if (MODE == 5) { VALU_FMA8 } if (MODE == 6) { VALU_MIN4 }
        if (MODE != 3 && MODE != 4) d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
        if (MODE == 1 || MODE == 3) { VALU_MIN8 } if (MODE == 2 || MODE == 4 || MODE == 5) { VALU_FMA8 } if (MODE == 5) { VALU_FMA8 } if (MODE == 6) { VALU_MIN4 }
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    }
    float s = d0[0] + d1[1] + d2[2] + d3[3];
    float r; asm volatile("v_add_f32 %0, v103, v104" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + r;
}
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[] = {"mfma only", "mfma + 8 min3", "mfma + 8 fma", "8 min3 only", "8 fma only", "mfma + 16 fma", "mfma + 4 min3"};
    for (int mode = 0; mode < 7; ++mode) for (int blocks : {256, 1024, 2048}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            switch (mode) { case 0: k<0><<<blocks, 256>>>(d, iters); break; case 1: k<1><<<blocks, 256>>>(d, iters); break; case 2: k<2><<<blocks, 256>>>(d, iters); break;
                            case 3: k<3><<<blocks, 256>>>(d, iters); break; case 4: k<4><<<blocks, 256>>>(d, iters); break; case 5: k<5><<<blocks, 256>>>(d, iters); break; case 6: k<6><<<blocks, 256>>>(d, iters); break; }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double steps = (double)blocks * 4 * iters * 4;          // (mfma + valu group) steps
        printf("%-16s blocks %4d: %.3f ms, %.1f cycles per step per SIMD @2.4GHz\n", names[mode], blocks, ms, 1024.0 * 2.4e9 * ms * 1e-3 / steps);
    }
    return 0;
}
