// v_sad_u16 issue rate against VGPR bank placement of its three sources (bank = reg mod 4) and
// against the accumulate-in-place form the descriptor kernel uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, INSTR)                                                                   \
__global__ __launch_bounds__(256) void NAME(float* out, int iters) {                          \
    asm volatile("v_mov_b32 v20, 1\n v_mov_b32 v21, 1\n v_mov_b32 v24, 3\n v_mov_b32 v25, 3\n v_mov_b32 v26, 5\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 7\n v_mov_b32 v40, 0" ::: "v20","v21","v24","v25","v26","v28","v29","v30","v31","v40"); \
    for (int it = 0; it < iters; ++it) { asm volatile(REP16(INSTR "\n") ::: "v28","v29","v30","v31","v40","v41","v42","v43","v44","v45","v46","v47"); }  \
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
}
KERNEL(k_distinct, "v_sad_u16 v40, v21, v26, v31")          // banks 1,2,3
KERNEL(k_s0s1,     "v_sad_u16 v40, v21, v25, v31")          // src0 & src1 bank 1
KERNEL(k_s0s2,     "v_sad_u16 v40, v21, v26, v29")          // src0 & src2 bank 1
KERNEL(k_s1s2,     "v_sad_u16 v40, v21, v26, v30")          // src1 & src2 bank 2
KERNEL(k_all,      "v_sad_u16 v40, v21, v25, v29")          // all bank 1
KERNEL(k_inplace4, "v_sad_u16 v40, v21, v26, v40\n v_sad_u16 v41, v21, v26, v41\n v_sad_u16 v42, v21, v26, v42\n v_sad_u16 v43, v21, v26, v43\n")   // 4 chains (x16 = 64 instr)
KERNEL(k_inplace8, "v_sad_u16 v40, v21, v26, v40\n v_sad_u16 v41, v21, v26, v41\n v_sad_u16 v42, v21, v26, v42\n v_sad_u16 v43, v21, v26, v43\n v_sad_u16 v44, v21, v26, v44\n v_sad_u16 v45, v21, v26, v45\n v_sad_u16 v46, v21, v26, v46\n v_sad_u16 v47, v21, v26, v47\n")
KERNEL(k_fma_ref,  "v_fma_f32 v40, v21, v26, v31")
typedef void (*kfn)(float*, int);
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    struct { const char* n; kfn f; double per; } ks[] = {{"distinct banks", k_distinct, 16}, {"src0=src1 bank", k_s0s1, 16}, {"src0=src2 bank", k_s0s2, 16},
        {"src1=src2 bank", k_s1s2, 16}, {"all one bank", k_all, 16}, {"in place, 4 chains", k_inplace4, 64}, {"in place, 8 chains", k_inplace8, 128}, {"v_fma_f32 ref", k_fma_ref, 16}};
    for (int blocks : {4096, 768}) {   // 768 blocks x 4 waves = 3 waves per SIMD, one round
        for (auto& k : ks) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0); hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            double ops = (double)blocks * 256 * iters * k.per;
            printf("blocks=%d %-22s %.3f ms, %.2f T lane-instr/s\n", blocks, k.n, ms, ops / ms / 1e9);
        }
    }
    return 0;
}
