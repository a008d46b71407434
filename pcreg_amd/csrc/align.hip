// pcreg_amd/csrc/align.hip -- AlignPoints_KNN (KNN-PCA local reference frame), batched.
//
// AlignPoints_KNN.m:17-59, one workgroup per support region:
//   :17     centroid                                   -> block reduction
//   :20-26  keep the K = round(0.85 N) points nearest to the centroid.  The reference
//           does a full stable sort; only the K-th order statistic is needed: one 256-bin
//           histogram over [min, max] of the distances held in LDS + an exact rank inside the
//           bin that holds the K-th (bisection, select_kth.hpp, only if that bin is crowded);
//           ties at the boundary go to the lowest indices (= stable sort order)
//   :30-34  pca(...,'Algorithm','eig')                 -> 3x3 covariance + Jacobi, then
//           MathWorks' sign convention (largest-|.| entry of each column positive)
//   :37-56  majority-sign disambiguation, y from det   -> block reduction of counts
//   :59     pts * coeff_unambig                        -> one coalesced pass
// The point set is read from HBM/L2 five times (centroid, distances, mean, covariance,
// projection): 5 * 24 B per point in, 24 B out -- the HBM-bound member of the family.
#include "common.hpp"
#include "select_kth.hpp"
#include <cfloat>

namespace pcreg {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double* s_red) {
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
    __syncthreads();
    return t;
}
__device__ __forceinline__ int block_sum_i(int v, int* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return t;
}

__device__ void jacobi_eig3(double (&A)[3][3], double (&V)[3][3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) V[r][c] = r == c ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        double dia = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= DBL_EPSILON * 1e-3 * dia) break;
#define PCREG_JROT(P, Q)                                                                    \
        if (A[P][Q] != 0.0) {                                                               \
            double th = (A[Q][Q] - A[P][P]) / (2.0 * A[P][Q]);                              \
            double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));                \
            double c = 1.0 / sqrt(t * t + 1.0), s = c * t;                                  \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = A[k][P], b = A[k][Q]; A[k][P] = c*a - s*b; A[k][Q] = s*a + c*b; } \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = A[P][k], b = A[Q][k]; A[P][k] = c*a - s*b; A[Q][k] = s*a + c*b; } \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = V[k][P], b = V[k][Q]; V[k][P] = c*a - s*b; V[k][Q] = s*a + c*b; } \
        }
        PCREG_JROT(0, 1) PCREG_JROT(0, 2) PCREG_JROT(1, 2)
#undef PCREG_JROT
    }
}

// N sums at once: per value the wave butterfly, then ((w0+w1)+w2)+w3 -- block_sum's order, one barrier pair
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* s_redn /*[4][N]*/) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum_d(v[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) s_redn[(threadIdx.x >> 6) * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = ((s_redn[k] + s_redn[N + k]) + s_redn[2 * N + k]) + s_redn[3 * N + k];
    __syncthreads();
}

__global__ __launch_bounds__(kBlock, 3) void align_points_knn_kernel(
    const double* __restrict__ pts, int ld, const int32_t* __restrict__ offsets, int C1, int C2,
    double* __restrict__ aligned, int ld_out, double* __restrict__ coeff_out, double* __restrict__ c_out,
    int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sd[];    // n distances to the centroid (their bit patterns order them)
    __shared__ double s_redn[4 * 6];
    __shared__ int s_redi[4];
    __shared__ unsigned long long s_u64[8];
    __shared__ double s_cu[9];       // coeff_unambig, row-major [r][col]
    __shared__ double s_coeff[9];
    constexpr int kSmall = 256;
    __shared__ unsigned long long s_small[kSmall];
    __shared__ int s_hist[256];
    __shared__ unsigned long long s_vk;
    __shared__ int s_nsmall, s_bin, s_below, s_nless, s_neq, s_base;

    const int b = blockIdx.x;
    const int off = offsets[b];
    const int n = offsets[b + 1] - off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* px = pts + off; const double* py = px + (size_t)ld; const double* pz = py + (size_t)ld;
    if (n < 2) { if (tid == 0) status[b] = 1; return; }

    // A pass over the support: thread t visits i = t, t + 256, ... in order; four points' loads are issued
    // before the first is used (the support is L2-resident, the passes are latency-bound).
#define PCREG_AL_PTS(...)                                                                       \
    for (int i0_ = tid; i0_ < n; i0_ += 4 * kBlock) {                                           \
        double X_[4], Y_[4], Z_[4];                                                             \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                      \
            const int ii_ = min(i0_ + u_ * kBlock, n - 1);                                      \
            X_[u_] = px[ii_]; Y_[u_] = py[ii_]; Z_[u_] = pz[ii_];                               \
        }                                                                                       \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                      \
            const int i = i0_ + u_ * kBlock;                                                    \
            if (i < n) { const double x0 = X_[u_], y0 = Y_[u_], z0 = Z_[u_]; __VA_ARGS__ }      \
        }                                                                                       \
    }

    // --- 1) centroid (:17)
    double c3[3] = {0, 0, 0};
    PCREG_AL_PTS(c3[0] += x0; c3[1] += y0; c3[2] += z0;)
    block_sum_n<3>(c3, s_redn);
    const double cx = c3[0] / n, cy = c3[1] / n, cz = c3[2] / n;

    // --- 2) distances to the centroid (:22-23); K-th smallest by ONE 256-bin histogram over [min, max]
    //        (the bin index is monotone in the distance) + an exact rank inside the bin that holds it
    const int K = (int)floor(n * 0.85 + 0.5);                                   // :20-21
    unsigned long long lo = ~0ull, hi = 0ull;
    PCREG_AL_PTS(const double x = x0 - cx; const double y = y0 - cy; const double z = z0 - cz;
                 const double d = sqrt(x * x + y * y + z * z); sd[i] = d;
                 const unsigned long long k = kth_key(d); lo = k < lo ? k : lo; hi = k > hi ? k : hi;)
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
        const unsigned long long a2 = __shfl_xor(lo, ofs), b2 = __shfl_xor(hi, ofs);
        lo = a2 < lo ? a2 : lo; hi = b2 > hi ? b2 : hi;
    }
    if (lane == 0) { s_u64[wave] = lo; s_u64[4 + wave] = hi; }
    for (int i = tid; i < 256; i += kBlock) s_hist[i] = 0;
    if (tid == 0) { s_nsmall = 0; s_vk = 0ull; s_nless = 0; s_neq = 0; s_base = 0; }
    __syncthreads();
    lo = s_u64[0]; hi = s_u64[4];
#pragma unroll
    for (int w = 1; w < 4; ++w) { lo = s_u64[w] < lo ? s_u64[w] : lo; hi = s_u64[4 + w] > hi ? s_u64[4 + w] : hi; }
    const double dlo = __longlong_as_double((long long)lo), dhi = __longlong_as_double((long long)hi);
    const double scale = dhi > dlo ? 256.0 / (dhi - dlo) : 0.0;
    auto bin_of = [&](double d) -> int { const int bb = (int)((d - dlo) * scale); return bb > 255 ? 255 : bb; };
    for (int i = tid; i < n; i += kBlock) atomicAdd(&s_hist[bin_of(sd[i])], 1);
    __syncthreads();
    if (wave == 0) {
        int c4[4], run = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) { c4[q4] = s_hist[lane * 4 + q4]; run += c4[q4]; }
        int incl = run;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const int t = __shfl_up(incl, ofs); if (lane >= ofs) incl += t; }
        int before = incl - run;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (before < K && K <= before + c4[q4]) { s_bin = lane * 4 + q4; s_below = before; }
            before += c4[q4];
        }
    }
    __syncthreads();
    const int bstar = s_bin, below = s_below;
    for (int i = tid; i < n; i += kBlock) {
        const double d = sd[i];
        if (bin_of(d) == bstar) { const int q = atomicAdd(&s_nsmall, 1); if (q < kSmall) s_small[q] = kth_key(d); }
    }
    __syncthreads();
    const int m = s_nsmall, Kp = K - below;
    unsigned long long vK; int n_less, n_eq;
    if (m <= kSmall) {
        for (int t = tid; t < m; t += kBlock) {
            const unsigned long long x = s_small[t];
            int less = 0, eq = 0;
            for (int u = 0; u < m; ++u) { const unsigned long long yv = s_small[u]; less += yv < x; eq += yv == x; }
            if (less < Kp && Kp <= less + eq) { s_vk = x; s_nless = below + less; s_neq = eq; }     // every writer writes the same
        }
        __syncthreads();
        vK = s_vk; n_less = s_nless; n_eq = s_neq;
    } else {                                         // crowded bin (many equal distances): bisection on the keys
        vK = block_select_kth(sd, n, K, s_u64, s_redi, &n_less);
        int c2 = 0;
        for (int i = tid; i < n; i += kBlock) c2 += kth_key(sd[i]) == vK;
        n_eq = block_sum_i(c2, s_redi);
    }
    const int take_eq = K - n_less;              // how many of the ties at vK belong to the K nearest
    // selection flags; ties at the K-th distance by ascending index (stable sort order, :24) -- ranked only
    // when the boundary really splits a group of equal distances
    if (n_eq == take_eq) {
        __syncthreads();
        for (int i = tid; i < n; i += kBlock) sd[i] = kth_key(sd[i]) <= vK ? 1.0 : 0.0;
        __syncthreads();
    } else {
        for (int i0 = 0; i0 < n; i0 += kBlock) {
            int i = i0 + tid;
            unsigned long long key = i < n ? kth_key(sd[i]) : ~0ull;
            bool eq = i < n && key == vK;
            unsigned long long bal = __ballot(eq);
            if (lane == 0) s_redi[wave] = __popcll(bal);
            __syncthreads();
            int rank = s_base;
            for (int w = 0; w < wave; ++w) rank += s_redi[w];
            rank += __popcll(bal & ((1ull << lane) - 1ull));
            bool sel = i < n && (key < vK || (eq && rank < take_eq));
            __syncthreads();
            if (i < n) sd[i] = sel ? 1.0 : 0.0;
            if (tid == 0) s_base += s_redi[0] + s_redi[1] + s_redi[2] + s_redi[3];
            __syncthreads();
        }
    }

    // --- 3) pca of the K selected, centroid-relative points (:30-34)
    double mx = 0, my = 0, mz = 0;
    if (!C1) {
        double a3[3] = {0, 0, 0};
        PCREG_AL_PTS(if (sd[i] != 0.0) { a3[0] += x0 - cx; a3[1] += y0 - cy; a3[2] += z0 - cz; })
        block_sum_n<3>(a3, s_redn);
        mx = a3[0] / K; my = a3[1] / K; mz = a3[2] / K;
    }
    double cv[6] = {0, 0, 0, 0, 0, 0};
    PCREG_AL_PTS(if (sd[i] != 0.0) {
        const double x = (x0 - cx) - mx; const double y = (y0 - cy) - my; const double z = (z0 - cz) - mz;
        cv[0] += x * x; cv[1] += x * y; cv[2] += x * z; cv[3] += y * y; cv[4] += y * z; cv[5] += z * z; })
    double dof = C1 ? (double)K : (double)(K - 1);
    if (dof < 1.0) dof = 1.0;
    block_sum_n<6>(cv, s_redn);
#pragma unroll
    for (int k = 0; k < 6; ++k) cv[k] = cv[k] / dof;
    if (tid == 0) {
        double A[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
        double V[3][3];
        jacobi_eig3(A, V);
        double ev[3] = {A[0][0], A[1][1], A[2][2]};
        int ord[3] = {0, 1, 2};                                  // descending eigenvalue, stable
        if (ev[ord[1]] > ev[ord[0]]) { int t = ord[0]; ord[0] = ord[1]; ord[1] = t; }
        if (ev[ord[2]] > ev[ord[0]]) { int t = ord[0]; ord[0] = ord[2]; ord[2] = t; }
        if (ev[ord[2]] > ev[ord[1]]) { int t = ord[1]; ord[1] = ord[2]; ord[2] = t; }
        for (int col = 0; col < 3; ++col) {
            double v0 = 0, v1 = 0, v2 = 0;
            for (int k = 0; k < 3; ++k) if (ord[col] == k) { v0 = V[0][k]; v1 = V[1][k]; v2 = V[2][k]; }
            double big = v0;                                     // largest-|.| entry positive (pca convention)
            if (fabs(v1) > fabs(big)) big = v1;
            if (fabs(v2) > fabs(big)) big = v2;
            double sg = big < 0 ? -1.0 : 1.0;
            s_coeff[0 * 3 + col] = sg * v0; s_coeff[1 * 3 + col] = sg * v1; s_coeff[2 * 3 + col] = sg * v2;
        }
    }
    __syncthreads();
    double co[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) co[k] = s_coeff[k];

    // --- sign disambiguation (:37-56)
    int posx = 0, posz = 0;
    PCREG_AL_PTS(
        if (C2) {
            posx += (x0 * co[0] + y0 * co[3] + z0 * co[6]) > 0;
            posz += (x0 * co[2] + y0 * co[5] + z0 * co[8]) > 0;
        } else if (sd[i] != 0.0) {
            const double x = (x0 - cx) - mx; const double y = (y0 - cy) - my; const double z = (z0 - cz) - mz;
            posx += (x * co[0] + y * co[3] + z * co[6]) > 0;
            posz += (x * co[2] + y * co[5] + z * co[8]) > 0;
        })
    posx = block_sum_i(posx, s_redi); posz = block_sum_i(posz, s_redi);
    if (tid == 0) {
        double xs = (2.0 * posx >= (double)n) ? 1.0 : -1.0;                    // :45,49 with k = N (:37)
        double zs = (2.0 * posz >= (double)n) ? 1.0 : -1.0;                    // :46,50
        double M[9];
        for (int r = 0; r < 3; ++r) { M[r * 3] = co[r * 3] * xs; M[r * 3 + 1] = co[r * 3 + 1]; M[r * 3 + 2] = co[r * 3 + 2] * zs; }
        double ys = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);   // :53
        for (int r = 0; r < 3; ++r) { s_cu[r * 3] = co[r * 3] * xs; s_cu[r * 3 + 1] = co[r * 3 + 1] * ys; s_cu[r * 3 + 2] = co[r * 3 + 2] * zs; }   // :56
        for (int r = 0; r < 3; ++r) for (int col = 0; col < 3; ++col) coeff_out[(size_t)b * 9 + r + 3 * col] = s_cu[r * 3 + col];
        c_out[(size_t)b * 3] = cx; c_out[(size_t)b * 3 + 1] = cy; c_out[(size_t)b * 3 + 2] = cz;
        status[b] = 0;
    }
    __syncthreads();
    double cu[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cu[k] = s_cu[k];
    double* ox = aligned + off; double* oy = ox + (size_t)ld_out; double* oz = oy + (size_t)ld_out;
    PCREG_AL_PTS(                                                               // :59
        ox[i] = x0 * cu[0] + y0 * cu[3] + z0 * cu[6];
        oy[i] = x0 * cu[1] + y0 * cu[4] + z0 * cu[7];
        oz[i] = x0 * cu[2] + y0 * cu[5] + z0 * cu[8];)
#undef PCREG_AL_PTS
}

}  // namespace

int launch_align_points_knn(const double* pts, int ld, const int32_t* offsets_dev, int B, int max_n, int C1, int C2,
                            double* aligned, int ld_out, double* coeff, double* c, int32_t* status, hipStream_t st) {
    PCREG_ARG(B >= 0 && max_n >= 0);
    if (B == 0) return PCREG_OK;
    size_t lds = (size_t)(max_n > 0 ? max_n : 1) * sizeof(double);
    if (lds > 60 * 1024) { set_error("AlignPoints_KNN support of %d points exceeds the LDS-resident limit (7680)", max_n); return PCREG_E_ARG; }
    hipLaunchKernelGGL(align_points_knn_kernel, dim3(B), dim3(kBlock), lds, st, pts, ld, offsets_dev, C1, C2, aligned,
                       ld_out, coeff, c, status);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
