// Does a VOP3-encoded v_fma_f32 (8-byte instruction, three distinct sources) issue as fast as
// the VOP2 v_fmac_f32?  Inline asm so the compiler cannot change the encodings.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16], y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = 0.5f * i; }
    float va = a + threadIdx.x * 1e-9f, vb = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(va), "v"(y[i]), "v"(x[i]));
            if (MODE == 2) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(va), "v"(y[i]), "v"(x[i]));
            if (MODE == 3) asm volatile("v_min_f32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(x[i]));
            if (MODE == 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(x[i]));
            if (MODE == 5) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(va), "v"(x[i]));
            if (MODE == 6) asm volatile("v_max_f32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(x[i]));
            if (MODE == 7) asm volatile("v_min_f32 %0, %1, %2" : "=v"(x[i]) : "v"(va), "v"(y[i]));      // no dependency chain
            if (MODE == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(va), "v"(y[i]), "v"(vb));   // no chain
            if (MODE == 9) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(x[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[10] = {"v_fmac_f32 (VOP2)", "v_fma_f32 (VOP3)", "v_min3_f32 (VOP3)", "v_min_f32 (VOP2)", "v_add_f32", "v_mul_f32", "v_max_f32", "v_min_f32 nochain", "v_fma_f32 nochain", "v_min_u32"};
    for (int mode = 0; mode < 10; ++mode) for (int blocks : {4096}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 3) k<3><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 4) k<4><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 5) k<5><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 6) k<6><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 7) k<7><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 8) k<8><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 9) k<9><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double ops = (double)blocks * 256 * iters * 16.0;
        printf("%s blocks %d: %.3f ms, %.2f T lane-instr/s\n", names[mode], blocks, ms, ops / ms / 1e9);
    }
    return 0;
}
