// Do the selection VALU ops of knn_candidates_f16_kernel overlap with v_mfma_f32_32x32x16_f16 on one SIMD?
//
// Every MFMA of the loop body has its own accumulator and an A operand the compiler cannot prove unchanged
// (an empty asm "+v" in front of it), so nothing is CSE'd: `make -C scripts/ubench check` (and
// tests/test_isa_lint.py) count the v_mfma instructions of each kernel's loop body in the emitted ISA.
// Modes (NM = 8 MFMAs per iteration in every MFMA mode; a "step" = 1 MFMA [+ one 16-score selection]):
//   0  MFMA only, 8 independent accumulators
//   1  serial:    MFMA -> min tree of ITS result -> compare + rare branch      (round 1's kernel structure)
//   2  pipelined: MFMA(i+1) issued, THEN the min tree of result(i)            (two accumulator tiles)
//   3  the 8 min trees + compares alone, on registers the loop keeps live      (VALU only)
//   4  pipelined, C = -thr folded into the MFMA (tree on s - thr, compare against 0 -> sign test)
//   5  pipelined, two tiles per compare: min3(tree0, tree1) then ONE compare + branch per two MFMAs
// Occupancy is set from the host by a dynamic-LDS reservation: 1, 2, 3, 4 waves per SIMD.
// Built with -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans like knn_mfma16.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float tree16(const f32x16& d) {
    float m0 = fminf(fminf(d[0], d[1]), d[2]), m1 = fminf(fminf(d[3], d[4]), d[5]);
    float m2 = fminf(fminf(d[6], d[7]), d[8]), m3 = fminf(fminf(d[9], d[10]), d[11]);
    float m4 = fminf(fminf(d[12], d[13]), d[14]);
    return fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), d[15]));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float thr_in, unsigned long long* clk) {
    extern __shared__ char reserve[];                      // occupancy knob only
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 a, b[4];
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(((threadIdx.x * 7 + i * 3) % 61) * 0.03125f);
        for (int g = 0; g < 4; ++g) b[g][i] = (_Float16)(((threadIdx.x * 5 + i + g) % 53) * 0.0625f + 1.0f);
    }
    const float thr = thr_in;                              // scores are >= 0 here, thr < 0: the branch is never taken
    f32x16 cthr; for (int r = 0; r < 16; ++r) cthr[r] = -thr;
    const f32x16 zero = {};
    float sink = 0.0f;
    f32x16 d[2] = {zero, zero};
    if (MODE == 3) for (int r = 0; r < 16; ++r) { d[0][r] = (float)(threadIdx.x + r); d[1][r] = (float)(threadIdx.x * 3 + r); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f32x16 e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(a)); e[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j & 3], zero, 0, 0, 0); }
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(e[j]));
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("" : "+v"(a));
                const f32x16 e = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j & 3], zero, 0, 0, 0);
                const float mn = tree16(e);
                if (mn < thr) { sink += mn; out[threadIdx.x] = sink; }
            }
        } else if (MODE == 2 || MODE == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("" : "+v"(a));
                const f32x16 prev = d[(j + 1) & 1];
                d[(j + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j & 3], MODE == 4 ? cthr : zero, 0, 0, 0);
                const float mn = tree16(prev);
                if (MODE == 4 ? (mn < 0.0f) : (mn < thr)) { sink += mn; out[threadIdx.x] = sink; }
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("" : "+v"(d[j & 1]));
                const float mn = tree16(d[j & 1]);
                if (mn < thr) { sink += mn; out[threadIdx.x] = sink; }
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                asm volatile("" : "+v"(a));
                const f32x16 p0 = d[0], p1 = d[1];
                d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j & 3], zero, 0, 0, 0);
                const float m0 = tree16(p0);
                asm volatile("" : "+v"(a));
                d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[(j + 1) & 3], zero, 0, 0, 0);
                const float m1 = tree16(p1);
                if (fminf(m0, m1) < thr) { sink += m0 + m1; out[threadIdx.x] = sink; }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink + d[0][0] + d[1][3] + tree16(d[0]) + tree16(d[1]);
    // shader clock held during the loop: delta(s_memtime) / delta(s_memrealtime) x 100 MHz (block 0 reports)
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

int main(int argc, char** argv) {
    float* dbuf; (void)hipMalloc(&dbuf, 8192 * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16); unsigned long long hclk[2];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const double ghz = argc > 2 ? atof(argv[2]) : 2.4;
    const char* names[] = {"mfma only", "serial mfma->tree->cmp", "pipelined tree(prev)", "trees only", "pipelined, C=-thr", "pipelined, cmp per 2"};
    for (int mode = 0; mode < 6; ++mode) for (int wps = 1; wps <= 4; ++wps) {
        const size_t lds = wps == 1 ? 100 * 1024 : wps == 2 ? 64 * 1024 : wps == 3 ? 44 * 1024 : 36 * 1024;   // 160 KiB / LDS = workgroups per CU
        const int blocks = 256 * wps * 4;
        float ms = 0, best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
#define L(M) case M: (void)hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k<M><<<blocks, 256, lds>>>(dbuf, iters, -1.0f, clk); break;
            switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) }
#undef L
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        // steps per SIMD = blocks/256 CUs (one wave of each block per SIMD) * iters * 8
        const double steps_per_simd = (double)blocks / 256.0 * iters * 8.0;
        (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        const double g = hclk[1] ? (double)hclk[0] / (double)hclk[1] * 0.1 : ghz;            // GHz held inside the kernel
        printf("%-24s %d waves/SIMD: %8.3f ms, %6.2f ns per step per SIMD, clock %.3f GHz -> %6.1f cycles per step\n", names[mode], wps, best, best * 1e6 / steps_per_simd, g, g * best * 1e6 / steps_per_simd);
    }
    return 0;
}
