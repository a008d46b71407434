// Issue rate of single VALU opcodes on gfx950 (explicit registers, operands in distinct banks).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, INSTR)                                                                   \
__global__ __launch_bounds__(256) void NAME(float* out, int iters) {                          \
    asm volatile("v_mov_b32 v21, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v31, 2.0\n v_mov_b32 v40, 0" ::: "v21","v26","v31","v40"); \
    for (int it = 0; it < iters; ++it) { asm volatile(REP16(INSTR "\n") ::: "v40","v41","v42","v43","vcc","s20","s21"); }  \
    float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r));                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
}
KERNEL(k_fma,   "v_fma_f32 v40, v21, v26, v31")
KERNEL(k_add,   "v_add_f32 v40, v21, v26")
KERNEL(k_sub,   "v_sub_f32 v40, v21, v26")
KERNEL(k_mul,   "v_mul_f32 v40, v21, v26")
KERNEL(k_min,   "v_min_f32 v40, v21, v26")
KERNEL(k_max,   "v_max_f32 v40, v21, v26")
KERNEL(k_min3,  "v_min3_f32 v40, v21, v26, v31")
KERNEL(k_med3,  "v_med3_f32 v40, v21, v26, v31")
KERNEL(k_cmp,   "v_cmp_lt_f32 vcc, v21, v26")
KERNEL(k_cmpe64,"v_cmp_lt_f32 s[20:21], v21, v26")
KERNEL(k_cnd,   "v_cndmask_b32 v40, v21, v26, vcc")
KERNEL(k_and,   "v_and_b32 v40, v21, v26")
KERNEL(k_or3,   "v_or3_b32 v40, v21, v26, v31")
KERNEL(k_mini,  "v_min_i32 v40, v21, v26")
KERNEL(k_addu,  "v_add_u32 v40, v21, v26")
KERNEL(k_mov,   "v_mov_b32 v40, v21")
KERNEL(k_pkmul, "v_pk_mul_f32 v[40:41], v[20:21], v[26:27]")
KERNEL(k_cmpclass, "v_cmp_class_f32 vcc, v21, v26")
KERNEL(k_sadu16, "v_sad_u16 v40, v21, v26, v31")
KERNEL(k_sadu8,  "v_sad_u8 v40, v21, v26, v31")
KERNEL(k_sadu32, "v_sad_u32 v40, v21, v26, v31")
KERNEL(k_dot2i,  "v_dot2_i32_i16 v40, v21, v26, v31")
KERNEL(k_dot4i,  "v_dot4_i32_i8 v40, v21, v26, v31")
KERNEL(k_dot2f,  "v_dot2_f32_f16 v40, v21, v26, v31")
KERNEL(k_pkfmah, "v_pk_fma_f16 v40, v21, v26, v31")
KERNEL(k_madu24, "v_mad_u32_u24 v40, v21, v26, v31")
typedef void (*kfn)(float*, int);
int main() {
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 4096;
    struct { const char* n; kfn f; } ks[] = {{"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_mul_f32", k_mul}, {"v_min_f32", k_min},
        {"v_max_f32", k_max}, {"v_min3_f32", k_min3}, {"v_med3_f32", k_med3}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmpe64},
        {"v_cndmask_b32", k_cnd}, {"v_and_b32", k_and}, {"v_or3_b32", k_or3}, {"v_min_i32", k_mini}, {"v_add_u32", k_addu}, {"v_mov_b32", k_mov},
        {"v_pk_mul_f32", k_pkmul}, {"v_cmp_class_f32", k_cmpclass}, {"v_sad_u16", k_sadu16}, {"v_sad_u8", k_sadu8}, {"v_sad_u32", k_sadu32},
        {"v_dot2_i32_i16", k_dot2i}, {"v_dot4_i32_i8", k_dot4i}, {"v_dot2_f32_f16", k_dot2f}, {"v_pk_fma_f16", k_pkfmah}, {"v_mad_u32_u24", k_madu24}};
    for (auto& k : ks) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double ops = (double)blocks * 256 * iters * 16.0;
        printf("%-22s %.3f ms, %.2f T lane-instr/s\n", k.n, ms, ops / ms / 1e9);
    }
    return 0;
}
