// Micro-benchmark: sustained fp32 VALU issue rate on MI355X for the instruction mixes the
// point-search kernel uses (scalar fma, sub/mul/fma mix, packed fma).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
    float2v p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = float2v{x[2*i], x[2*i+1]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 1) {   // sub, mul, fma, fma pattern (like the distance chain)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                float dx = x[i] - a, dy = x[i+1] - b, dz = x[i+2] - a;
                x[i+3] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) * 1e-9f + x[i+3];
                x[i] = dx + x[i+3]; x[i+1] = dy; x[i+2] = dz;
            }
        } else {
            float2v av{a, a}, bv{b, b};
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], av, bv);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kd(double* out, int iters, double a, double b) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    { double* dd; hipMalloc(&dd, 8192 * 256 * 8); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int blocks : {1024, 2048, 4096}) { float ms = 0; for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); kd<<<blocks, 256>>>(dd, 20000, 1.0001, 0.5); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
        double ops = (double)blocks * 256 * 20000 * 8.0; printf("fp64 fma blocks %d: %.3f ms, %.2f T lane-instr/s, %.1f TFLOP/s\n", blocks, ms, ops / ms / 1e9, 2 * ops / ms / 1e9); } }
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) for (int blocks : {1024, 2048, 4096, 8192}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        double lane_ops = (double)blocks * 256 * iters * (mode == 1 ? (4 * 9.0) : 16.0);   // VALU instr per lane (pk counts 8 instr)
        if (mode == 2) lane_ops = (double)blocks * 256 * iters * 8.0;
        double flops = (double)blocks * 256 * iters * (mode == 1 ? 4 * 11.0 : 32.0);
        printf("mode %d blocks %d: %.3f ms, %.2f T lane-instr/s, %.1f TFLOP/s\n", mode, blocks, ms, lane_ops / ms / 1e9, flops / ms / 1e9);
    }
    return 0;
}
