// Does a kernel pay a fixed "ramp" cost?  Same grid, growing trip counts, back-to-back vs idle gaps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {1024, 2048}) for (int iters : {5000, 10000, 20000, 40000, 80000}) {
        float ms = 0, best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0); k<<<blocks, 256>>>(d, iters, 1.0001f, 0.5f); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // 10 launches queued back to back
        (void)hipEventRecord(e0); for (int r = 0; r < 10; ++r) k<<<blocks, 256>>>(d, iters, 1.0001f, 0.5f); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 256 * iters * 32.0;
        printf("blocks %d iters %d: single %.3f ms (%.1f TF), back-to-back %.3f ms/launch (%.1f TF)\n", blocks, iters, best, fl / best / 1e9, ms / 10, fl / (ms / 10) / 1e9);
    }
    return 0;
}
