// pcreg_amd/csrc/select.hpp -- matchFeatures' post-search filters as device templates
// shared by the fp32 point search and the fp64 descriptor search:
//   threshold (removeWeakMatches), ratio test (removeAmbiguousMatches), and the
//   order-preserving compaction that keeps pairs ascending in the query index
//   (matchFeatures' output order; getMatches.m:51-56).
#pragma once
#include "common.hpp"

namespace pcreg {

template <typename T>
struct Top2T { T d1, d2; int i1, i2; };

template <typename T>
__device__ __forceinline__ bool lex_lt_t(T da, int ia, T db, int ib) {
    return da < db || (da == db && (unsigned)ia < (unsigned)ib);
}
// insert ordering by (dist, idx); -1 = empty slot; duplicates of an index are ignored
template <typename T>
__device__ __forceinline__ void top2_insert_lex_t(Top2T<T>& t, T d, int j) {
    if (j < 0 || j == t.i1 || j == t.i2) return;
    if (lex_lt_t(d, j, t.d2, t.i2)) {
        if (lex_lt_t(d, j, t.d1, t.i1)) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = j; }
        else { t.d2 = d; t.i2 = j; }
    }
}

// Merge R lists [R][Q][2] -> [Q][2]; list r starts r * rank_stride elements in (0: densely packed, Q * 2).
template <typename T>
__global__ void merge_top2_kernel_t(const int32_t* __restrict__ idx_in, const T* __restrict__ dist_in, int R, int Q,
                                    int32_t* __restrict__ idx, T* __restrict__ dist, size_t rank_stride = 0) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= Q) return;
    if (rank_stride == 0) rank_stride = (size_t)Q * 2;
    Top2T<T> t{(T)INFINITY, (T)INFINITY, -1, -1};
    for (int r = 0; r < R; ++r) {
        size_t o = (size_t)r * rank_stride + (size_t)qi * 2;
        top2_insert_lex_t(t, dist_in[o], idx_in[o]);
        top2_insert_lex_t(t, dist_in[o + 1], idx_in[o + 1]);
    }
    idx[(size_t)qi * 2] = t.i1; idx[(size_t)qi * 2 + 1] = t.i2;
    dist[(size_t)qi * 2] = t.d1; dist[(size_t)qi * 2 + 1] = t.d2;
}

template <typename T>
__global__ void filter_flag_kernel_t(const int32_t* __restrict__ idx, const T* __restrict__ dist, int Q, int M_total,
                                     T thr, T ratio, int32_t* __restrict__ flag, int32_t* __restrict__ block_cnt) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (qi < Q) {
        T d1 = dist[(size_t)qi * 2], d2 = dist[(size_t)qi * 2 + 1];
        int i1 = idx[(size_t)qi * 2];
        keep = i1 >= 0 && d1 <= thr;
        if (keep && M_total > 1) {
            T t1 = d1, t2 = d2;
            if (t2 < (T)1e-6) { t1 = (T)1; t2 = (T)1; }
            keep = (t1 / t2) <= ratio;
        }
        flag[qi] = keep;
    }
    __shared__ int s_cnt[4];
    unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of per-workgroup counts by one workgroup; *total = sum
static __global__ void scan_blocks_kernel(int32_t* __restrict__ block_cnt, int nblocks, int32_t* __restrict__ total) {
    __shared__ int s[256];
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        int i = b0 + threadIdx.x;
        int v = i < nblocks ? block_cnt[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_cnt[i] = carry + s[threadIdx.x] - v;
        int tot = s[255];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// The offset of workgroup b = the sum of the RAW counts of the workgroups before it, which every workgroup of a scatter
// kernel can add up itself (a few hundred ints): the compaction chains then need no scan launch in between.  Used up to
// kSelfPrefixMax workgroups (the reads grow with the square); the last workgroup also publishes the total.
constexpr int kSelfPrefixMax = 2048;
__device__ __forceinline__ int block_self_prefix_256(const int32_t* __restrict__ block_cnt, int b, int nblocks, int* s4 /*[4]*/,
                                                     int32_t* __restrict__ total) {
    int v = 0;
    for (int i = threadIdx.x; i < b; i += 256) v += block_cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    const int base = s4[0] + s4[1] + s4[2] + s4[3];
    if (total && b == nblocks - 1 && threadIdx.x == 0) *total = base + block_cnt[b];
    __syncthreads();
    return base;
}

static __global__ void filter_scatter_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ flag, int Q,
                                             const int32_t* __restrict__ block_off, int self_prefix, int32_t* __restrict__ n_cand,
                                             int32_t* __restrict__ cand_q, int32_t* __restrict__ cand_m) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = qi < Q && flag[qi];
    __shared__ int s_cnt[4];
    __shared__ int s_pre[4];
    const int pre = self_prefix ? block_self_prefix_256(block_off, blockIdx.x, gridDim.x, s_pre, n_cand) : block_off[blockIdx.x];
    unsigned long long b = __ballot(keep);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(b);
    __syncthreads();
    int base = pre;
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    if (keep) {
        int o = base + __popcll(b & ((1ull << lane) - 1ull));
        cand_q[o] = qi; cand_m[o] = idx[(size_t)qi * 2];
    }
}

// ---- ordered compaction in ONE launch for SMALL inputs (Q <= kCompactMax): a single workgroup of 1024 threads, every
// thread owns a contiguous run of <= 4 items; count -> block-wide exclusive scan -> ordered write.  One launch instead
// of flag + scan + scatter for the per-sphere getMatches sizes.  (Tried at the registration step's 50 k candidates too:
// 49 dependent strided reads per thread made it 276 us against 14 us for the three parallel launches; not used there.)
constexpr int kCompactMax = 4096;
constexpr int kCompactThreads = 1024;

// exclusive scan of one int per thread over a 1024-thread workgroup; returns the thread's offset, *total = sum
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* s_wave /*[16]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kCompactThreads / 64; ++w) { const int c = s_wave[w]; if (w < wave) base += c; tot += c; }
    *total = tot;
    return base + incl - v;
}

template <typename T>
__device__ __forceinline__ bool filter_keep(const int32_t* __restrict__ idx, const T* __restrict__ dist, int qi, int M_total, T thr, T ratio) {
    T d1 = dist[(size_t)qi * 2], d2 = dist[(size_t)qi * 2 + 1];
    bool keep = idx[(size_t)qi * 2] >= 0 && d1 <= thr;
    if (keep && M_total > 1) {
        T t1 = d1, t2 = d2;
        if (t2 < (T)1e-6) { t1 = (T)1; t2 = (T)1; }
        keep = (t1 / t2) <= ratio;
    }
    return keep;
}

template <typename T>
__global__ __launch_bounds__(kCompactThreads) void filter_compact_kernel(const int32_t* __restrict__ idx, const T* __restrict__ dist, int Q, int M_total,
                                                                         T thr, T ratio, int32_t* __restrict__ cand_q, int32_t* __restrict__ cand_m,
                                                                         int32_t* __restrict__ n_cand) {
    __shared__ int s_wave[kCompactThreads / 64];
    const int per = (Q + kCompactThreads - 1) / kCompactThreads;
    const int lo = min(Q, (int)threadIdx.x * per), hi = min(Q, lo + per);
    int cnt = 0;
    for (int qi = lo; qi < hi; ++qi) cnt += filter_keep<T>(idx, dist, qi, M_total, thr, ratio);
    int total;
    int o = block_exclusive_scan_1024(cnt, s_wave, &total);
    for (int qi = lo; qi < hi; ++qi)
        if (filter_keep<T>(idx, dist, qi, M_total, thr, ratio)) { cand_q[o] = qi; cand_m[o] = idx[(size_t)qi * 2]; ++o; }
    if (threadIdx.x == 0) *n_cand = total;
}

// threshold + ratio + ordered compaction; tmp must hold (Q + ceil(Q/256)) int32
template <typename T>
int run_filter_top2(const int32_t* idx, const T* dist, int Q, int M_total, T thr, T ratio, int32_t* cand_q,
                    int32_t* cand_m, int32_t* n_cand, int32_t* tmp, hipStream_t st) {
    if (Q == 0) { PCREG_HIP(hipMemsetAsync(n_cand, 0, sizeof(int32_t), st)); return PCREG_OK; }
    if (Q <= kCompactMax) {
        hipLaunchKernelGGL(filter_compact_kernel<T>, dim3(1), dim3(kCompactThreads), 0, st, idx, dist, Q, M_total, thr, ratio, cand_q, cand_m, n_cand);
        PCREG_HIP(hipGetLastError());
        return PCREG_OK;
    }
    int nb = (Q + 255) / 256;
    int32_t* flag = tmp; int32_t* bc = tmp + Q;
    hipLaunchKernelGGL(filter_flag_kernel_t<T>, dim3(nb), dim3(256), 0, st, idx, dist, Q, M_total, thr, ratio, flag, bc);
    const int self_prefix = nb <= kSelfPrefixMax;
    if (!self_prefix) hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, st, bc, nb, n_cand);
    hipLaunchKernelGGL(filter_scatter_kernel, dim3(nb), dim3(256), 0, st, idx, flag, Q, bc, self_prefix, n_cand, cand_q, cand_m);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
