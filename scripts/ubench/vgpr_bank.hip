// VGPR bank conflicts on gfx950?  v_fma_f32 with the three sources in the same bank (reg % 4)
// versus in three different banks.  Explicit register numbers through inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(X) X X X X
__global__ __launch_bounds__(256) void k_same(float* out, int iters) {
    // sources v20, v24, v28 (all bank 0) -> dst v40..v43
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v28, 2.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v31, 2.0" ::: "v20","v24","v28","v21","v26","v31");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fma_f32 v40, v20, v24, v28\n v_fma_f32 v41, v20, v24, v28\n v_fma_f32 v42, v20, v24, v28\n v_fma_f32 v43, v20, v24, v28\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_diff(float* out, int iters) {
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v28, 2.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v31, 2.0" ::: "v20","v24","v28","v21","v26","v31");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fma_f32 v40, v21, v26, v31\n v_fma_f32 v41, v21, v26, v31\n v_fma_f32 v42, v21, v26, v31\n v_fma_f32 v43, v21, v26, v31\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_two_same(float* out, int iters) {   // src1, src2 same bank, src0 different
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v28, 2.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v31, 2.0" ::: "v20","v24","v28","v21","v26","v31");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fma_f32 v40, v21, v24, v28\n v_fma_f32 v41, v21, v24, v28\n v_fma_f32 v42, v21, v24, v28\n v_fma_f32 v43, v21, v24, v28\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_01_same(float* out, int iters) {   // src0, src1 same bank, src2 different
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v28, 2.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v31, 2.0" ::: "v20","v24","v28","v21","v26","v31");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fma_f32 v40, v20, v24, v31\n v_fma_f32 v41, v20, v24, v31\n v_fma_f32 v42, v20, v24, v31\n v_fma_f32 v43, v20, v24, v31\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_02_same(float* out, int iters) {   // src0, src2 same bank, src1 different
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v28, 2.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v31, 2.0" ::: "v20","v24","v28","v21","v26","v31");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fma_f32 v40, v20, v26, v28\n v_fma_f32 v41, v20, v26, v28\n v_fma_f32 v42, v20, v26, v28\n v_fma_f32 v43, v20, v26, v28\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_fmac_01(float* out, int iters) {   // v_fmac: src0, src1 same bank, accumulator elsewhere
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v45, 0" ::: "v20","v24","v41","v42","v43","v45");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fmac_f32 v41, v20, v24\n v_fmac_f32 v42, v20, v24\n v_fmac_f32 v43, v20, v24\n v_fmac_f32 v45, v20, v24\n")
                     ::: "v41","v42","v43","v45");
    }
    float r; asm volatile("v_mov_b32 %0, v41" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_fmac_diff(float* out, int iters) {
    asm volatile("v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v40, 0\n v_mov_b32 v44, 0\n v_mov_b32 v48, 0\n v_mov_b32 v52, 0" ::: "v21","v26","v40","v44","v48","v52");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fmac_f32 v40, v21, v26\n v_fmac_f32 v44, v21, v26\n v_fmac_f32 v48, v21, v26\n v_fmac_f32 v52, v21, v26\n")
                     ::: "v40","v44","v48","v52");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_fmac_0acc(float* out, int iters) {   // v_fmac: src0 and accumulator same bank, src1 different
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v40, 0\n v_mov_b32 v44, 0\n v_mov_b32 v48, 0\n v_mov_b32 v52, 0" ::: "v20","v26","v40","v44","v48","v52");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fmac_f32 v40, v20, v26\n v_fmac_f32 v44, v20, v26\n v_fmac_f32 v48, v20, v26\n v_fmac_f32 v52, v20, v26\n")
                     ::: "v40","v44","v48","v52");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_fmac_1acc(float* out, int iters) {   // v_fmac: src1 and accumulator same bank, src0 different
    asm volatile("v_mov_b32 v21, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v40, 0\n v_mov_b32 v44, 0\n v_mov_b32 v48, 0\n v_mov_b32 v52, 0" ::: "v21","v24","v40","v44","v48","v52");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_fmac_f32 v40, v21, v24\n v_fmac_f32 v44, v21, v24\n v_fmac_f32 v48, v21, v24\n v_fmac_f32 v52, v21, v24\n")
                     ::: "v40","v44","v48","v52");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_min_same(float* out, int iters) {   // VOP2 v_min with both sources in one bank
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v21, 1.0" ::: "v20","v24","v21");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_min_f32 v40, v20, v24\n v_min_f32 v41, v20, v24\n v_min_f32 v42, v20, v24\n v_min_f32 v43, v20, v24\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_min_diff(float* out, int iters) {
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v21, 1.0" ::: "v20","v24","v21");
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP4("v_min_f32 v40, v21, v24\n v_min_f32 v41, v21, v24\n v_min_f32 v42, v21, v24\n v_min_f32 v43, v21, v24\n")
                     ::: "v40","v41","v42","v43");
    }
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 4096;
    const char* names[11] = {"fma 3 sources same bank", "fma 3 different banks", "fma src1,src2 same bank", "min 2 sources same bank", "min 2 different banks", "fma src0,src1 same bank", "fma src0,src2 same bank", "fmac src0,src1 same bank (acc other)", "fmac all different", "fmac src0,acc same bank", "fmac src1,acc same bank"};
    for (int mode = 0; mode < 11; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k_same<<<blocks, 256>>>(d, iters);
            if (mode == 1) k_diff<<<blocks, 256>>>(d, iters);
            if (mode == 2) k_two_same<<<blocks, 256>>>(d, iters);
            if (mode == 3) k_min_same<<<blocks, 256>>>(d, iters);
            if (mode == 4) k_min_diff<<<blocks, 256>>>(d, iters);
            if (mode == 5) k_01_same<<<blocks, 256>>>(d, iters);
            if (mode == 6) k_02_same<<<blocks, 256>>>(d, iters);
            if (mode == 7) k_fmac_01<<<blocks, 256>>>(d, iters);
            if (mode == 8) k_fmac_diff<<<blocks, 256>>>(d, iters);
            if (mode == 9) k_fmac_0acc<<<blocks, 256>>>(d, iters);
            if (mode == 10) k_fmac_1acc<<<blocks, 256>>>(d, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double ops = (double)blocks * 256 * iters * 16.0;
        printf("%s: %.3f ms, %.2f T lane-instr/s\n", names[mode], ms, ops / ms / 1e9);
    }
    return 0;
}
