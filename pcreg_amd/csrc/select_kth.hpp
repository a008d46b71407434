// pcreg_amd/csrc/select_kth.hpp -- K-th smallest of n non-negative doubles held in LDS, by one
// 256-thread workgroup, replacing the reference's full sort (AlignPoints_KNN.m:24,
// getSpacialHistogramDescriptors.m:82).  Non-negative IEEE doubles order like their bit
// patterns, so the search is a bisection on the 64-bit key: count(key <= mid) by every thread
// over its stride, one block reduction per step, at most 64 steps and usually ~45 because the
// search starts from the actual minimum and maximum.  (A digit-histogram radix select was
// measured first: all distances of a support share their leading bytes, so every LDS atomic of
// a pass lands on ONE bin and serialises -- 8 x n same-address ds_add, ~10x slower than this.)
#pragma once
#include <hip/hip_runtime.h>

namespace pcreg {

__device__ __forceinline__ unsigned long long kth_key(double d) { return (unsigned long long)__double_as_longlong(d); }

// returns the key of the K-th smallest (1 <= K <= n); *count_less = #{key < result}.
// s_u64[4] and s_i[4] are LDS scratch.  Every thread of the 256-thread block must call it.
__device__ inline unsigned long long block_select_kth(const double* __restrict__ sd, int n, int K,
                                                      unsigned long long* s_u64, int* s_i, int* count_less) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int i = tid; i < n; i += 256) { unsigned long long k = kth_key(sd[i]); lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if (lane == 0) { s_u64[wave] = lo; s_u64[4 + wave] = hi; }
    __syncthreads();
    lo = s_u64[0]; hi = s_u64[4];
#pragma unroll
    for (int w = 1; w < 4; ++w) { lo = s_u64[w] < lo ? s_u64[w] : lo; hi = s_u64[4 + w] > hi ? s_u64[4 + w] : hi; }
    __syncthreads();
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int i = tid; i < n; i += 256) c += kth_key(sd[i]) <= mid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) s_i[wave] = c;
        __syncthreads();
        c = s_i[0] + s_i[1] + s_i[2] + s_i[3];
        __syncthreads();
        if (c >= K) hi = mid; else lo = mid + 1;
    }
    int c = 0;
    for (int i = tid; i < n; i += 256) c += kth_key(sd[i]) < lo;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) s_i[wave] = c;
    __syncthreads();
    *count_less = s_i[0] + s_i[1] + s_i[2] + s_i[3];
    __syncthreads();
    return lo;
}

}  // namespace pcreg
