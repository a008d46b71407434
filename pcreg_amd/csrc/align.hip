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
// The support is read from HBM once and kept in registers (24 B per point in, 24 B out): the HBM-bound member of
// the family, in practice a chain of ~11 block-wide steps per support whose latency is what the kernel pays.
#include "common.hpp"
#include "select_kth.hpp"
#include "wave_math.hpp"
#include <cfloat>
#include <cstdio>
#include <vector>

namespace pcreg {
namespace {


// Thread t keeps points t, t + NT, ... (PT of them) and their distances in registers: one load phase (every load of the
// support in flight at once: 72 KB per workgroup at n = 3000, what one CU needs outstanding to keep its share of HBM
// busy), then register-only arithmetic between the block-wide steps.  NT = 512 with PT <= 8 keeps two workgroups per CU
// (<= 128 VGPRs, no scratch: the cross-wave sums are butterflies, the Jacobi's V lives in LDS).  Round 1's kernel
// re-read the support from L2 in five passes (0.29 ms for 4096 x 3000; this one 0.20 ms).
__device__ __forceinline__ double opaque_f64(double x) { asm volatile("" : "+v"(x)); return x; }
// N block-wide sums: DPP wave sums (wave_math.hpp), then every thread adds the NW wave partials as the same pairwise
// tree (broadcast LDS reads: NW - 1 additions per value instead of a second 30-instruction butterfly), so all threads end
// with the same bits and only N values are live
template <int NW>
__device__ __forceinline__ double tree_sum_lds(const double* s, int stride) {
    if constexpr (NW == 1) return s[0];
    else {
        constexpr int H = NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 4 : 8));     // the largest power of two below NW
        return tree_sum_lds<H>(s, stride) + tree_sum_lds<NW - H>(s + H * stride, stride);
    }
}
template <int NW, int N>
__device__ __forceinline__ void block_sum_nw(double (&v)[N], double* s_redn /*[NW][N]*/) {
    static_assert(NW >= 1 && NW <= 16, "one wave partial per wave of a workgroup");
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum_dpp(v[k]);
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) s_redn[(threadIdx.x >> 6) * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = tree_sum_lds<NW>(s_redn + k, N);
    __syncthreads();
}
template <int NW>
__device__ __forceinline__ int block_sum_iw(int v, int* s_red) {
    v = wave_sum_dpp_i(v);
    const int lane = threadIdx.x & 63;
    if (lane == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[w];
    __syncthreads();
    return t;
}

#ifdef PCREG_EXPERIMENTS        // PCREG_ALIGN_TIMES=1: thread 0's clock at the phase boundaries, averaged by the launcher
#define PCREG_AL_STAMP(k) if (tstamp && tid == 0) tstamp[(size_t)blockIdx.x * 16 + k] = (long long)__builtin_readcyclecounter();
#else
#define PCREG_AL_STAMP(k)
#endif
template <int PT, int NT>
__global__ __launch_bounds__(NT, NT == 128 ? 2 : (NT == 256 && PT > 8 ? 3 : (NT == 1024 ? 2 : (PT <= 8 ? 4 : 2)))) void align_points_knn_reg_kernel(
    const double* __restrict__ pts, int ld, const int32_t* __restrict__ offsets, int C1, int C2,
    double* __restrict__ aligned, int ld_out, double* __restrict__ coeff_out, double* __restrict__ c_out,
    int32_t* __restrict__ status, long long* __restrict__ tstamp) {
    constexpr int NW = NT / 64;
    __shared__ double s_redn[NW * 6];
    __shared__ int s_redi[NW];
    __shared__ unsigned long long s_u64[2 * NW];
    __shared__ double s_cu[9];
    __shared__ double s_coeff[9];
    __shared__ double s_V[9];
    constexpr int kSmall = 256;
    __shared__ unsigned long long s_small[kSmall];
    __shared__ int s_hist[256];
    __shared__ unsigned long long s_vk;
    __shared__ int s_nsmall, s_nless, s_neq;

    const int b = blockIdx.x;
    const int off = offsets[b];
    const int n = offsets[b + 1] - off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* px = pts + off; const double* py = px + (size_t)ld; const double* pz = py + (size_t)ld;
    if (n < 2) { if (tid == 0) status[b] = 1; return; }

    PCREG_AL_STAMP(0)
    double X[PT], Y[PT], Z[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int ii = min(tid + r * NT, n - 1);
        X[r] = px[ii]; Y[r] = py[ii]; Z[r] = pz[ii];
    }
#define PCREG_RG(...) _Pragma("unroll") for (int r = 0; r < PT; ++r) { const int i = tid + r * NT; if (i < n) { __VA_ARGS__ } }

    // --- 1) centroid (:17)
    double c3[3] = {0, 0, 0};
    PCREG_RG(c3[0] += X[r]; c3[1] += Y[r]; c3[2] += Z[r];)
    block_sum_nw<NW>(c3, s_redn);
    const double cx = c3[0] / n, cy = c3[1] / n, cz = c3[2] / n;
    // every pass subtracts the centroid afresh: the three differences per point are NOT kept across passes (the
    // compiler would, for 6 VGPRs per point and a spilling kernel), hence one opaque copy of the centroid per pass
    const double cxa = opaque_f64(cx), cya = opaque_f64(cy), cza = opaque_f64(cz);

    PCREG_AL_STAMP(1)
    // --- 2) distances to the centroid (:22-23), K-th smallest: 256-bin histogram over [min, max] + exact rank in its bin
    const int K = (int)floor(n * 0.85 + 0.5);                                   // :20-21
    double D[PT];
    double dmin = DBL_MAX, dmax = 0.0;
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const double x = X[r] - cxa, y = Y[r] - cya, z = Z[r] - cza;
        D[r] = sqrt(x * x + y * y + z * z);
        if (tid + r * NT < n) { dmin = fmin(dmin, D[r]); dmax = fmax(dmax, D[r]); }
    }
    dmin = wave_min_dpp(dmin); dmax = wave_max_dpp(dmax);
    if (lane == 0) { s_u64[wave] = kth_key(dmin); s_u64[NW + wave] = kth_key(dmax); }
    for (int i = tid; i < 256; i += NT) s_hist[i] = 0;
    if (tid == 0) { s_nsmall = 0; s_vk = 0ull; s_nless = 0; s_neq = 0; }
    __syncthreads();
    double dlo = __longlong_as_double((long long)s_u64[0]), dhi = __longlong_as_double((long long)s_u64[NW]);     // broadcast reads
#pragma unroll
    for (int w = 1; w < NW; ++w) { dlo = fmin(dlo, __longlong_as_double((long long)s_u64[w])); dhi = fmax(dhi, __longlong_as_double((long long)s_u64[NW + w])); }
    const unsigned long long lo = kth_key(dlo), hi = kth_key(dhi);
    const double scale = dhi > dlo ? 256.0 / (dhi - dlo) : 0.0;
    auto bin_of = [&](double d) -> int { const int bb = (int)((d - dlo) * scale); return bb > 255 ? 255 : bb; };
    PCREG_RG(atomicAdd(&s_hist[bin_of(D[r])], 1);)
    __syncthreads();
    int bstar, below;
    {   // every wave finds the bin of the K-th and the number of entries below that bin itself (no broadcast through LDS)
        int c4[4], run = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) { c4[q4] = s_hist[lane * 4 + q4]; run += c4[q4]; }
        const int incl = wave_scan_incl_i(run);
        int before = incl - run, mybin = -1, mybelow = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (before < K && K <= before + c4[q4]) { mybin = lane * 4 + q4; mybelow = before; }
            before += c4[q4];
        }
        const unsigned long long has = __ballot(mybin >= 0);           // exactly one lane: 1 <= K <= n = the histogram's total
        const int src = has ? __ffsll((long long)has) - 1 : 0;
        bstar = __shfl(mybin, src); below = __shfl(mybelow, src);
        if (bstar < 0) bstar = 0;
    }
    PCREG_RG(if (bin_of(D[r]) == bstar) { const int q = atomicAdd(&s_nsmall, 1); if (q < kSmall) s_small[q] = kth_key(D[r]); })
    __syncthreads();
    const int m = s_nsmall, Kp = K - below;
    unsigned long long vK; int n_less, n_eq;
    if (m <= kSmall) {
        for (int t = tid; t < m; t += NT) {
            const unsigned long long x = s_small[t];
            int less = 0, eq = 0;
            for (int u = 0; u < m; ++u) { const unsigned long long yv = s_small[u]; less += yv < x; eq += yv == x; }
            if (less < Kp && Kp <= less + eq) { s_vk = x; s_nless = below + less; s_neq = eq; }     // every writer writes the same
        }
        __syncthreads();
        vK = s_vk; n_less = s_nless; n_eq = s_neq;
    } else {                                         // crowded bin (many equal distances): bisection on the keys
        unsigned long long blo = lo, bhi = hi;
        while (blo < bhi) {
            const unsigned long long mid = blo + ((bhi - blo) >> 1);
            int c = 0;
            PCREG_RG(c += kth_key(D[r]) <= mid;)
            c = block_sum_iw<NW>(c, s_redi);
            if (c >= K) bhi = mid; else blo = mid + 1;
        }
        vK = blo;
        int c1 = 0, c2 = 0;
        PCREG_RG(c1 += kth_key(D[r]) < vK; c2 += kth_key(D[r]) == vK;)
        n_less = block_sum_iw<NW>(c1, s_redi); n_eq = block_sum_iw<NW>(c2, s_redi);
    }
    const int take_eq = K - n_less;              // how many of the ties at vK belong to the K nearest
    // selection bits; ties at the K-th distance by ascending index (stable sort order, :24): index order is r-major,
    // then thread -- ranked only when the boundary really splits a group of equal distances
    unsigned sel = 0u;
    if (n_eq == take_eq) {
        PCREG_RG(sel |= (kth_key(D[r]) <= vK ? 1u : 0u) << r;)
    } else {
        int base = 0;
#pragma unroll
        for (int r = 0; r < PT; ++r) {                 // rare: two barriers per row of points
            const unsigned long long key = kth_key(D[r]);
            const bool in = tid + r * NT < n;
            const unsigned long long bal = __ballot(in && key == vK);
            if (lane == 0) s_redi[wave] = __popcll(bal);
            __syncthreads();
            int rank = base;
#pragma unroll
            for (int w = 0; w < NW; ++w) { if (w < wave) rank += s_redi[w]; base += s_redi[w]; }
            rank += __popcll(bal & ((1ull << lane) - 1ull));
            if (in && (key < vK || (key == vK && rank < take_eq))) sel |= 1u << r;
            __syncthreads();
        }
    }

    PCREG_AL_STAMP(2)
    // --- 3) pca of the K selected, centroid-relative points (:30-34)
    double mx = 0, my = 0, mz = 0;
    if (!C1) {
        double a3[3] = {0, 0, 0};
        const double cxb = opaque_f64(cx), cyb = opaque_f64(cy), czb = opaque_f64(cz);
        PCREG_RG(if ((sel >> r) & 1u) { a3[0] += X[r] - cxb; a3[1] += Y[r] - cyb; a3[2] += Z[r] - czb; })
        block_sum_nw<NW>(a3, s_redn);
        mx = a3[0] / K; my = a3[1] / K; mz = a3[2] / K;
    }
    PCREG_AL_STAMP(3)
    // (one pass for both moments, as in the descriptor kernel, was measured too: nine sums at once spill at 128 VGPRs
    // and the kernel is 3 % slower)
    double cv[6] = {0, 0, 0, 0, 0, 0};
    const double cxc = opaque_f64(cx), cyc = opaque_f64(cy), czc = opaque_f64(cz);
    PCREG_RG(if ((sel >> r) & 1u) {
        const double x = (X[r] - cxc) - mx; const double y = (Y[r] - cyc) - my; const double z = (Z[r] - czc) - mz;
        cv[0] += x * x; cv[1] += x * y; cv[2] += x * z; cv[3] += y * y; cv[4] += y * z; cv[5] += z * z; })
    double dof = C1 ? (double)K : (double)(K - 1);
    if (dof < 1.0) dof = 1.0;
    block_sum_nw<NW>(cv, s_redn);
    PCREG_AL_STAMP(4)
    if (tid == 0) {
        double a[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] = cv[k] / dof;
        jacobi_sym3(a, s_V);      // V in LDS: on registers the (12, 256) form spills and is 6 % slower, the descriptor kernel does not change
        double ev[3] = {a[0], a[3], a[5]};
        int ord[3] = {0, 1, 2};                                  // descending eigenvalue, stable
        if (ev[ord[1]] > ev[ord[0]]) { int t = ord[0]; ord[0] = ord[1]; ord[1] = t; }
        if (ev[ord[2]] > ev[ord[0]]) { int t = ord[0]; ord[0] = ord[2]; ord[2] = t; }
        if (ev[ord[2]] > ev[ord[1]]) { int t = ord[1]; ord[1] = ord[2]; ord[2] = t; }
        for (int col = 0; col < 3; ++col) {
            const int k = ord[col];
            const double v0 = s_V[k], v1 = s_V[3 + k], v2 = s_V[6 + k];
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

    PCREG_AL_STAMP(5)
    // --- sign disambiguation (:37-56)
    int posx = 0, posz = 0;
    const double cxd = opaque_f64(cx), cyd = opaque_f64(cy), czd = opaque_f64(cz);
    PCREG_RG(
        if (C2) {
            posx += (X[r] * co[0] + Y[r] * co[3] + Z[r] * co[6]) > 0;
            posz += (X[r] * co[2] + Y[r] * co[5] + Z[r] * co[8]) > 0;
        } else if ((sel >> r) & 1u) {
            const double x = (X[r] - cxd) - mx; const double y = (Y[r] - cyd) - my; const double z = (Z[r] - czd) - mz;
            posx += (x * co[0] + y * co[3] + z * co[6]) > 0;
            posz += (x * co[2] + y * co[5] + z * co[8]) > 0;
        })
    { const int both = block_sum_iw<NW>(posx | (posz << 16), s_redi); posx = both & 0xFFFF; posz = both >> 16; }   // n <= 8192
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
    PCREG_AL_STAMP(6)
    double* ox = aligned + off; double* oy = ox + (size_t)ld_out; double* oz = oy + (size_t)ld_out;
    PCREG_RG(                                                                   // :59
        ox[i] = X[r] * cu[0] + Y[r] * cu[3] + Z[r] * cu[6];
        oy[i] = X[r] * cu[1] + Y[r] * cu[4] + Z[r] * cu[7];
        oz[i] = X[r] * cu[2] + Y[r] * cu[5] + Z[r] * cu[8];)
    PCREG_AL_STAMP(7)
#undef PCREG_RG
}

}  // namespace

int launch_align_points_knn(const double* pts, int ld, const int32_t* offsets_dev, int B, int max_n, int C1, int C2,
                            double* aligned, int ld_out, double* coeff, double* c, int32_t* status, hipStream_t st) {
    PCREG_ARG(B >= 0 && max_n >= 0);
    if (B == 0) return PCREG_OK;
#define PCREG_AL_LAUNCH(PT, NT) hipLaunchKernelGGL((align_points_knn_reg_kernel<PT, NT>), dim3(B), dim3(NT), 0, st, pts, ld, offsets_dev, C1, C2, \
                                                   aligned, ld_out, coeff, c, status, tstamp)
    long long* tstamp = nullptr;
#ifdef PCREG_EXPERIMENTS
    if (debug_flag(kDbgAlignTimes)) { PCREG_HIP(hipMalloc((void**)&tstamp, (size_t)B * 16 * 8)); PCREG_HIP(hipMemsetAsync(tstamp, 0, (size_t)B * 16 * 8, st)); }
#endif
    if (max_n > 8192) { set_error("AlignPoints_KNN support of %d points exceeds the register-resident limit (8192)", max_n); return PCREG_E_ARG; }
    {
        if (max_n <= 1024) PCREG_AL_LAUNCH(4, 256);
        else if (max_n <= 2048) PCREG_AL_LAUNCH(4, 512);
#ifdef PCREG_EXPERIMENTS
        else if (max_n <= 3072 && debug_flag(kDbgAlignShape) == 1) PCREG_AL_LAUNCH(6, 512);
        else if (max_n <= 3072 && debug_flag(kDbgAlignShape) == 2) PCREG_AL_LAUNCH(24, 128);
#endif
        // four-wave workgroups, three per CU (<= 168 VGPRs): shorter barriers and a third support in flight per CU beat the
        // eight-wave form (two per CU at 128 VGPRs, spilling) by 1.3 x at 4096 x 3000 (A/B on one box: 0.155 vs 0.204 ms)
        else if (max_n <= 3072) PCREG_AL_LAUNCH(12, 256);
        else if (max_n <= 4096) PCREG_AL_LAUNCH(8, 512);
        else PCREG_AL_LAUNCH(16, 512);
        PCREG_HIP(hipGetLastError());
#ifdef PCREG_EXPERIMENTS
        if (tstamp) {                                  // mean cycles per phase over the supports (thread 0's clock)
            std::vector<long long> h((size_t)B * 16);
            PCREG_HIP(hipStreamSynchronize(st));
            PCREG_HIP(hipMemcpy(h.data(), tstamp, h.size() * 8, hipMemcpyDeviceToHost));
            double acc[8] = {0};
            for (int b = 0; b < B; ++b) for (int k = 1; k < 8; ++k) acc[k] += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + k - 1]);
            fprintf(stderr, "[pcreg] align phases (cycles, mean of %d): load+centroid %.0f select %.0f mean %.0f cov %.0f jacobi %.0f vote %.0f store %.0f\n",
                    B, acc[1] / B, acc[2] / B, acc[3] / B, acc[4] / B, acc[5] / B, acc[6] / B, acc[7] / B);
            PCREG_HIP(hipFree(tstamp));
        }
#endif
        return PCREG_OK;
    }
#undef PCREG_AL_LAUNCH
}

}  // namespace pcreg
