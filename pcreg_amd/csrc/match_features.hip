// pcreg_amd/csrc/match_features.hip -- descriptor matching on gfx950, fp64.
//
// getMatches.m:22-41  append the constant column, element-wise power  -> preprocess_*
// getMatches.m:51-56  matchFeatures (documented semantics, exact search):
//      row L2-normalisation                              -> normalize_rows_kernel
//      SAD / SSD all-pairs scores + two best per query   -> score_top2_kernel
//      threshold / ratio / Unique / ascending pairs      -> select.hpp + kernels below
//
// Layout: MATLAB hands over Q x D and M x D column-major matrices, i.e. for a fixed
// feature index the rows are contiguous -- exactly the "K-major" operand layout a
// register-tiled all-pairs kernel wants, so no transposition happens anywhere.
// score_top2_kernel: a 128 x 64 (query x model) tile per workgroup, 8 x 4 scores per
// lane, the feature dimension streamed through LDS in slabs of 16.  Scores accumulate
// over the feature index in ascending order with the oracle's operation order
// (SAD: s += |a-b|;  SSD: s = fma(a-b, a-b, s)), so the fp64 scores are the same bits.
// Arithmetic is IEEE double like the reference (descriptors are created with nan()/
// zeros(), getSpacialHistogramDescriptors.m:61-62).
#include "common.hpp"
#include "select.hpp"
#include <cfloat>
#include <algorithm>
#include <cstdlib>

namespace pcreg {
namespace {

constexpr int kBlock = 256;
constexpr int TQ = 8, TM = 4;               // scores per lane
constexpr int BQ = 16 * TQ;                 // 128 queries per workgroup
constexpr int BM = 16 * TM;                 // 64 model rows per tile
constexpr int DK = 16;                      // feature slab

// ---------------------------------------------------------------- preprocessing
// One lane per row keeps the oracle's summation order (features in ascending order); what the other lanes CAN do
// is fetch.  A workgroup owns 64 rows: all 256 threads stage a 64-row x 48-feature tile in LDS (coalesced along the
// rows, 12 loads per thread in flight, the next tile's loads issued before the current tile is consumed), and wave 0
// -- lane r = row r -- adds the tile's features in order.  The two descriptor sets share one launch.
constexpr int kRT = 64, kFT = 48;
struct RowSrc { const double* p; int ld; int rows; };           // first row of this workgroup, leading dimension, rows here
__device__ __forceinline__ RowSrc row_src(const double* f, int n, int ld, const double* f2, int n2, int ld2) {
    const int nb1 = (n + kRT - 1) / kRT;
    const int b = blockIdx.x;
    if (b < nb1) return RowSrc{f + (size_t)b * kRT, ld, min(kRT, n - b * kRT)};
    return RowSrc{f2 + (size_t)(b - nb1) * kRT, ld2, min(kRT, n2 - (b - nb1) * kRT)};
}
// MODE 0: sum |v| (vecnorm(.,1,2), getMatches.m:24); MODE 1: sum v^2 with fma (matchFeatures' normalizeX)
template <int MODE>
__device__ __forceinline__ double row_reduce(const RowSrc& src, int D, double (*tile)[kFT][kRT]) {
    const int tid = threadIdx.x, r = tid & (kRT - 1), fq = tid >> 6;          // thread loads features fq, fq + 4, ...
    const int ntiles = (D + kFT - 1) / kFT;
    double v[kFT / 4];
#define PCREG_ROW_FETCH(T)                                                                          \
    _Pragma("unroll") for (int k = 0; k < kFT / 4; ++k) {                                            \
        const int d = (T) * kFT + fq + 4 * k;                                                        \
        v[k] = (d < D && r < src.rows) ? src.p[(size_t)d * src.ld + r] : 0.0;                        \
    }
#define PCREG_ROW_STASH(BUF)                                                                        \
    _Pragma("unroll") for (int k = 0; k < kFT / 4; ++k) tile[BUF][fq + 4 * k][r] = v[k];
    double s = 0.0;
    PCREG_ROW_FETCH(0) PCREG_ROW_STASH(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) { PCREG_ROW_FETCH(t + 1) }
        if (tid < kRT) {
            const int dn = min(kFT, D - t * kFT);
            for (int d = 0; d < dn; ++d) { const double x = tile[t & 1][d][r]; if (MODE == 0) s += fabs(x); else s = fma(x, x, s); }
        }
        if (t + 1 < ntiles) { PCREG_ROW_STASH((t + 1) & 1) }
        __syncthreads();
    }
#undef PCREG_ROW_FETCH
#undef PCREG_ROW_STASH
    return s;                                                                  // valid in threads 0..63 (row r)
}
__global__ __launch_bounds__(256) void row_l1_kernel(const double* __restrict__ f, int n, int ld,
                                                     const double* __restrict__ f2, int n2, int ld2, int D, double* __restrict__ out) {
    __shared__ double tile[2][kFT][kRT];
    const RowSrc src = row_src(f, n, ld, f2, n2, ld2);
    const double s = row_reduce<0>(src, D, tile);
    const int nb1 = (n + kRT - 1) / kRT;
    const int row0 = (int)blockIdx.x < nb1 ? (int)blockIdx.x * kRT : n + ((int)blockIdx.x - nb1) * kRT;
    if ((int)threadIdx.x < src.rows) out[row0 + threadIdx.x] = s;
}
// deterministic mean of n values by one workgroup -> *out = factor * mean
__global__ void mean_kernel(const double* __restrict__ v, int n, double factor, double* __restrict__ out) {
    __shared__ double s[256];
    double a = 0;
    for (int i = threadIdx.x; i < n; i += 256) a += v[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = factor * (s[0] / (double)n);
}
// out (n x Dp, ld n) = [in, col] .^ factor                                  getMatches.m:25-26,36-37
__global__ void preprocess_kernel(const double* __restrict__ in, int n, int ld, int D, int Dp,
                                  const double* __restrict__ col, int change_metric, double factor,
                                  double* __restrict__ out) {
    size_t total = (size_t)n * Dp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int d = (int)(e / n), i = (int)(e % n);
        double v = d < D ? in[i + (size_t)d * ld] : *col;
        out[e] = change_metric ? pow(v, factor) : v;
    }
}
// matchFeatures' normalizeX: unit L2 rows, effectively-zero rows -> 0
__global__ __launch_bounds__(256) void normalize_rows_kernel(double* __restrict__ f, int n, int ld,
                                                             double* __restrict__ f2, int n2, int ld2, int D) {
    __shared__ double tile[2][kFT][kRT];
    __shared__ double s_nrm[kRT];
    const RowSrc src = row_src(f, n, ld, f2, n2, ld2);
    const double s = row_reduce<1>(src, D, tile);
    if (threadIdx.x < kRT) s_nrm[threadIdx.x] = sqrt(s);
    __syncthreads();
    double* p = const_cast<double*>(src.p);
    const int r = threadIdx.x & (kRT - 1);
    if (r >= src.rows) return;
    const double nrm = s_nrm[r];
    const bool zero = nrm <= (double)FLT_EPSILON;
    for (int d0 = threadIdx.x >> 6; d0 < D; d0 += 4 * 8) {                     // 8 loads in flight per thread
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int d = d0 + 4 * k; v[k] = d < D ? p[(size_t)d * src.ld + r] : 0.0; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int d = d0 + 4 * k; if (d < D) p[(size_t)d * src.ld + r] = zero ? 0.0 : v[k] / nrm; }
    }
}

// ---------------------------------------------------------------- all-pairs + top-2
// A: nA x D (queries), B: nB x D (model).  grid = (ceil(nA/BQ), S chunks of B).
// part_* layout [S][nA][2].
template <int METRIC>
__global__ __launch_bounds__(kBlock) void score_top2_kernel(const double* __restrict__ A, int nA, int lda,
                                                            const double* __restrict__ B, int nB, int ldb, int D,
                                                            int chunk, int32_t* __restrict__ part_idx,
                                                            double* __restrict__ part_dist) {
    __shared__ __attribute__((aligned(16))) double smem[DK * BQ + DK * BM];     // 24 KiB
    double* As = smem;               // [DK][BQ]
    double* Bs = smem + DK * BQ;     // [DK][BM]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * BQ;
    const int s = blockIdx.y;
    const int b_begin = s * chunk, b_end = min(nB, b_begin + chunk);

    Top2T<double> best[TQ];
#pragma unroll
    for (int r = 0; r < TQ; ++r) best[r] = Top2T<double>{INFINITY, INFINITY, -1, -1};

    for (int m0 = b_begin; m0 < b_end; m0 += BM) {
        double acc[TQ][TM];
#pragma unroll
        for (int r = 0; r < TQ; ++r)
#pragma unroll
            for (int c = 0; c < TM; ++c) acc[r][c] = 0.0;
        for (int d0 = 0; d0 < D; d0 += DK) {
            __syncthreads();
            // stage: for a fixed feature the rows are contiguous in memory (coalesced)
#pragma unroll
            for (int k = 0; k < DK * BQ / kBlock; ++k) {
                int e = k * kBlock + tid, r = e % BQ, dd = e / BQ;
                int gi = q0 + r, gd = d0 + dd;
                As[dd * BQ + r] = (gi < nA && gd < D) ? A[gi + (size_t)gd * lda] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < DK * BM / kBlock; ++k) {
                int e = k * kBlock + tid, r = e % BM, dd = e / BM;
                int gj = m0 + r, gd = d0 + dd;
                Bs[dd * BM + r] = (gj < b_end && gd < D) ? B[gj + (size_t)gd * ldb] : 0.0;
            }
            __syncthreads();
#pragma unroll 4
            for (int dd = 0; dd < DK; ++dd) {
                double a[TQ], b[TM];
#pragma unroll
                for (int r = 0; r < TQ; ++r) a[r] = As[dd * BQ + tx * TQ + r];
#pragma unroll
                for (int c = 0; c < TM; ++c) b[c] = Bs[dd * BM + ty * TM + c];
#pragma unroll
                for (int r = 0; r < TQ; ++r)
#pragma unroll
                    for (int c = 0; c < TM; ++c) {
                        double t = a[r] - b[c];
                        acc[r][c] = METRIC == PCREG_METRIC_SAD ? acc[r][c] + fabs(t) : fma(t, t, acc[r][c]);
                    }
            }
        }
        // this lane's model rows ascend with (m0, c): strict '<' keeps the lowest index
#pragma unroll
        for (int c = 0; c < TM; ++c) {
            int j = m0 + ty * TM + c;
            if (j < b_end) {
#pragma unroll
                for (int r = 0; r < TQ; ++r) {
                    double d = acc[r][c];
                    if (d < best[r].d2) {
                        if (d < best[r].d1) { best[r].d2 = best[r].d1; best[r].i2 = best[r].i1; best[r].d1 = d; best[r].i1 = j; }
                        else { best[r].d2 = d; best[r].i2 = j; }
                    }
                }
            }
        }
    }
    // merge the 16 ty-lists of every query through LDS (two halves of 64 queries)
    struct Cell { double d1, d2; int i1, i2; };
    Cell* cells = reinterpret_cast<Cell*>(smem);     // 64 x 16 x 24 B = 24 KiB
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TQ; ++r) {
            int ql = tx * TQ + r;                      // 0..127
            if ((ql >> 6) == half) cells[(ql & 63) * 16 + ty] = Cell{best[r].d1, best[r].d2, best[r].i1, best[r].i2};
        }
        __syncthreads();
        if (tid < 64) {
            int qi = q0 + half * 64 + tid;
            if (qi < nA) {
                Top2T<double> t{INFINITY, INFINITY, -1, -1};
                for (int y = 0; y < 16; ++y) {
                    Cell c = cells[tid * 16 + y];
                    top2_insert_lex_t(t, c.d1, c.i1);
                    top2_insert_lex_t(t, c.d2, c.i2);
                }
                size_t o = ((size_t)s * nA + qi) * 2;
                part_idx[o] = t.i1; part_idx[o + 1] = t.i2;
                part_dist[o] = t.d1; part_dist[o + 1] = t.d2;
            }
        }
    }
}

// gather rows of B (by cand_m) into a compact P x D column-major matrix (ld = cap)
__global__ void gather_rows_kernel(const double* __restrict__ B, int ldb, int D, const int32_t* __restrict__ cand_m,
                                   const int32_t* __restrict__ n_cand, int cap, double* __restrict__ out) {
    const int P = *n_cand;
    size_t total = (size_t)P * D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int d = (int)(e / P), k = (int)(e % P);
        out[k + (size_t)d * cap] = B[cand_m[k] + (size_t)d * ldb];
    }
}
// keep[k] = (first-best query of candidate k's model row == cand_q[k])
__global__ void unique_flag_kernel(const int32_t* __restrict__ back_idx, const int32_t* __restrict__ cand_q,
                                   const int32_t* __restrict__ n_cand, int32_t* __restrict__ keep) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < *n_cand) keep[k] = back_idx[(size_t)k * 2] == cand_q[k];
}
// ordered compaction of the kept candidates into 1-based pairs + matchMetric
__global__ void emit_pairs_kernel(const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m,
                                  const int32_t* __restrict__ keep, const int32_t* __restrict__ n_cand,
                                  const double* __restrict__ dist, uint32_t* __restrict__ pairs,
                                  double* __restrict__ metric, int32_t* __restrict__ n_pairs) {
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const int P = *n_cand;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < P; k0 += 256) {
        int k = k0 + threadIdx.x;
        bool kp = k < P && (keep == nullptr || keep[k] != 0);
        unsigned long long b = __ballot(kp);
        if (lane == 0) s_cnt[wave] = __popcll(b);
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wave; ++w) base += s_cnt[w];
        if (kp) {
            int o = base + __popcll(b & ((1ull << lane) - 1ull));
            pairs[(size_t)o * 2] = (uint32_t)cand_q[k] + 1u;
            pairs[(size_t)o * 2 + 1] = (uint32_t)cand_m[k] + 1u;
            if (metric) metric[o] = dist[(size_t)cand_q[k] * 2];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_pairs = s_base;
}

int pick_splits_desc(int n_tiles, int nB) {
    int S = (1024 + n_tiles - 1) / n_tiles;
    int maxS = (nB + BM - 1) / BM;
    if (S > maxS) S = maxS;
    return S < 1 ? 1 : S;
}

// top-2 of every row of A against all rows of B; tmp holds the [S][nA][2] partials
int run_score_top2(const double* A, int nA, int lda, const double* B, int nB, int ldb, int D, int metric,
                   int32_t* idx, double* dist, void* tmp, hipStream_t st) {
    int n_tiles = (nA + BQ - 1) / BQ;
    int S = pick_splits_desc(n_tiles, nB);
    int chunk = ((nB + S - 1) / S + BM - 1) / BM * BM;
    S = (nB + chunk - 1) / chunk;
    int32_t* part_idx = (int32_t*)tmp;
    double* part_dist = (double*)((char*)tmp + align_up((size_t)S * nA * 2 * sizeof(int32_t), 256));
    if (metric == PCREG_METRIC_SAD)
        hipLaunchKernelGGL(score_top2_kernel<PCREG_METRIC_SAD>, dim3(n_tiles, S), dim3(kBlock), 0, st, A, nA, lda, B, nB, ldb, D, chunk, part_idx, part_dist);
    else
        hipLaunchKernelGGL(score_top2_kernel<PCREG_METRIC_SSD>, dim3(n_tiles, S), dim3(kBlock), 0, st, A, nA, lda, B, nB, ldb, D, chunk, part_idx, part_dist);
    PCREG_HIP(hipGetLastError());
    hipLaunchKernelGGL(merge_top2_kernel_t<double>, dim3((nA + 255) / 256), dim3(256), 0, st, part_idx, part_dist, S, nA, idx, dist, (size_t)0);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
size_t score_tmp_bytes(int nA, int nB) {
    int n_tiles = (nA + BQ - 1) / BQ; if (n_tiles < 1) n_tiles = 1;
    int S = pick_splits_desc(n_tiles, nB > 0 ? nB : 1);
    size_t n = (size_t)(nA > 0 ? nA : 1);
    return align_up((size_t)S * n * 2 * sizeof(int32_t), 256) + align_up((size_t)S * n * 2 * sizeof(double), 256);
}

}  // namespace

// the certified u16 fast path (match_sad16.hip)
size_t sad16_workspace_bytes(int nA, int nB, int D);
int run_sad16_top2(const double* A, int nA, int lda, const double* B, int nB, int ldb, int D,
                   int32_t* idx, double* dist, void* ws, size_t ws_bytes, hipStream_t st, const int32_t* nA_live = nullptr);

// SAD runs on the certified u16 path unless pcreg_debug_set("match_exact", 1) (identical results either way)
static bool use_sad16(int metric) {
    return metric == PCREG_METRIC_SAD && debug_flag(kDbgMatchExact) == 0;
}

// workspace layout of launch_match_features (all sizes for capacity Q):
//   idx [Q][2] i32 | dist [Q][2] f64 | cand_q [Q] | cand_m [Q] | keep [Q] | n_cand | filter tmp
//   | back_idx [Q][2] | back_dist [Q][2] | gathered rows [Q x D] | score partials
size_t match_features_workspace_bytes(int Q, int M, int D) {
    size_t q = (size_t)(Q > 0 ? Q : 1);
    size_t b = 0;
    b += align_up(q * 2 * sizeof(int32_t), 256) + align_up(q * 2 * sizeof(double), 256);
    b += 3 * align_up(q * sizeof(int32_t), 256) + 256;
    b += align_up((q + q / 256 + 2) * sizeof(int32_t), 256);
    b += align_up(q * 2 * sizeof(int32_t), 256) + align_up(q * 2 * sizeof(double), 256);
    b += align_up(q * (size_t)(D > 0 ? D : 1) * sizeof(double), 256);
    size_t t1 = score_tmp_bytes(Q, M), t2 = score_tmp_bytes(Q, Q);
    size_t t3 = sad16_workspace_bytes(Q, M > Q ? M : Q, D);
    t1 = t1 > t2 ? t1 : t2;
    b += t1 > t3 ? t1 : t3;
    return b;
}

int launch_preprocess(const double* dS, int Q, int ldS, const double* dM, int M, int ldM, int D,
                      const pcreg_match_opts& o, double* outS, double* outM, void* ws, size_t ws_bytes,
                      hipStream_t st) {
    // ws: row L1 norms [Q+M] + the constant (1 double)
    size_t need = ((size_t)Q + M + 1) * sizeof(double);
    if (ws_bytes < need) { set_error("preprocess workspace too small"); return PCREG_E_WORKSPACE; }
    double* l1 = (double*)ws; double* col = l1 + (size_t)Q + M;
    int Dp = D + (o.unnormalize ? 1 : 0);
    if (o.unnormalize) {
        if (Q + M > 0) hipLaunchKernelGGL(row_l1_kernel, dim3((Q + kRT - 1) / kRT + (M + kRT - 1) / kRT), dim3(256), 0, st, dS, Q, ldS, dM, M, ldM, D, l1);
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, l1, Q + M, o.norm_factor, col);
    }
    if (Q > 0) hipLaunchKernelGGL(preprocess_kernel, dim3(1024), dim3(256), 0, st, dS, Q, ldS, D, Dp, col, o.change_metric, o.metric_factor, outS);
    if (M > 0) hipLaunchKernelGGL(preprocess_kernel, dim3(1024), dim3(256), 0, st, dM, M, ldM, D, Dp, col, o.change_metric, o.metric_factor, outM);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_normalize_rows(double* f, int n, int ld, int D, hipStream_t st) {
    if (n <= 0) return PCREG_OK;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((n + kRT - 1) / kRT), dim3(256), 0, st, f, n, ld, (double*)nullptr, 0, 0, D);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
// both descriptor sets in one launch
int launch_normalize_rows2(double* f, int n, int ld, double* f2, int n2, int ld2, int D, hipStream_t st) {
    if (n + n2 <= 0) return PCREG_OK;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((n + kRT - 1) / kRT + (n2 + kRT - 1) / kRT), dim3(256), 0, st, f, n, ld, f2, n2, ld2, D);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

// fS / fM must already be normalised when !prenormalized was requested (the host tier
// does that on its private copies).
int launch_match_features(const double* fS, int Q, int ldS, const double* fM, int M, int ldM, int D,
                          const pcreg_match_opts& o, uint32_t* pairs, double* metric, int32_t* P_dev,
                          void* ws, size_t ws_bytes, hipStream_t st) {
    PCREG_ARG(Q >= 0 && M >= 0 && D >= 1);
    if (Q == 0 || M == 0) { PCREG_HIP(hipMemsetAsync(P_dev, 0, sizeof(int32_t), st)); return PCREG_OK; }
    size_t need = match_features_workspace_bytes(Q, M, D);
    if (ws_bytes < need) { set_error("match workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    char* w = (char*)ws;
    size_t q = (size_t)Q;
    int32_t* idx = (int32_t*)w;        w += align_up(q * 2 * sizeof(int32_t), 256);
    double* dist = (double*)w;         w += align_up(q * 2 * sizeof(double), 256);
    int32_t* cand_q = (int32_t*)w;     w += align_up(q * sizeof(int32_t), 256);
    int32_t* cand_m = (int32_t*)w;     w += align_up(q * sizeof(int32_t), 256);
    int32_t* keep = (int32_t*)w;       w += align_up(q * sizeof(int32_t), 256);
    int32_t* n_cand = (int32_t*)w;     w += 256;
    int32_t* ftmp = (int32_t*)w;       w += align_up((q + q / 256 + 2) * sizeof(int32_t), 256);
    int32_t* back_idx = (int32_t*)w;   w += align_up(q * 2 * sizeof(int32_t), 256);
    double* back_dist = (double*)w;    w += align_up(q * 2 * sizeof(double), 256);
    double* rows = (double*)w;         w += align_up(q * (size_t)D * sizeof(double), 256);
    void* stmp = w;
    const size_t stmp_bytes = ws_bytes - (size_t)(w - (char*)ws);
    const bool fast = use_sad16(o.metric);

    int rc = fast ? run_sad16_top2(fS, Q, ldS, fM, M, ldM, D, idx, dist, stmp, stmp_bytes, st)
                  : run_score_top2(fS, Q, ldS, fM, M, ldM, D, o.metric, idx, dist, stmp, st);
    if (rc) return rc;
    double maxval = o.metric == PCREG_METRIC_SSD ? 4.0 : 2.0 * sqrt((double)D);   // percentToLevel
    double thr = (o.matchThreshold * 0.01) * maxval;
    rc = run_filter_top2<double>(idx, dist, Q, M, thr, o.maxRatio, cand_q, cand_m, n_cand, ftmp, st);
    if (rc) return rc;
    const int32_t* keep_ptr = nullptr;
    if (o.unique) {
        // Small problems (the reference's per-sphere sizes): the back-search runs on the CAPACITY (Q rows) and reads the
        // real candidate count on the device -- no host round trip, the kernels skip the rows past the count.  Large
        // ones keep the read-back: there 50 us of latency are nothing, and sizing the grid by the real count balances
        // the candidates kernel better (measured +3 % at 50 k x 50 k on the capacity).
        if (fast && (double)Q * (double)Q * (double)D <= 2.0e10) {
            hipLaunchKernelGGL(gather_rows_kernel, dim3(1024), dim3(256), 0, st, fM, ldM, D, cand_m, n_cand, Q, rows);
            PCREG_HIP(hipGetLastError());
            rc = run_sad16_top2(rows, Q, Q, fS, Q, ldS, D, back_idx, back_dist, stmp, stmp_bytes, st, n_cand);
            if (rc) return rc;
            hipLaunchKernelGGL(unique_flag_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, back_idx, cand_q, n_cand, keep);
        } else {
            // sized by the real candidate count (one small D2H read); also the exhaustive fp64 form (SSD, PCREG_MATCH_EXACT)
            int32_t P_host = 0;
            PCREG_HIP(hipMemcpyAsync(&P_host, n_cand, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            PCREG_HIP(hipStreamSynchronize(st));
            if (P_host > 0) {
                hipLaunchKernelGGL(gather_rows_kernel, dim3(1024), dim3(256), 0, st, fM, ldM, D, cand_m, n_cand, Q, rows);
                PCREG_HIP(hipGetLastError());
                rc = fast ? run_sad16_top2(rows, P_host, Q, fS, Q, ldS, D, back_idx, back_dist, stmp, stmp_bytes, st)
                          : run_score_top2(rows, P_host, Q, fS, Q, ldS, D, o.metric, back_idx, back_dist, stmp, st);
                if (rc) return rc;
                hipLaunchKernelGGL(unique_flag_kernel, dim3((P_host + 255) / 256), dim3(256), 0, st, back_idx, cand_q, n_cand, keep);
            }
        }
        keep_ptr = keep;
    }
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(1), dim3(256), 0, st, cand_q, cand_m, keep_ptr, n_cand, dist, pairs, metric, P_dev);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}


// pts1 = featSurface(matches(:,1),:), pts2 = featModel(matches(:,2),:)   (completeExperimentFast.m:205-206)
// feat* are row-major [.][3] (what the descriptor kernel emits); pts* are n x 3 column-major, ld = cap.
__global__ void gather_matched_rows_kernel(const uint32_t* __restrict__ pairs, const int32_t* __restrict__ n_pairs, int cap,
                                           const double* __restrict__ featS, const double* __restrict__ featM,
                                           double* __restrict__ pts1, double* __restrict__ pts2) {
    const int n = min(*n_pairs, cap);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const size_t a = (size_t)(pairs[(size_t)k * 2] - 1u), b = (size_t)(pairs[(size_t)k * 2 + 1] - 1u);
#pragma unroll
        for (int c = 0; c < 3; ++c) { pts1[k + (size_t)c * cap] = featS[a * 3 + c]; pts2[k + (size_t)c * cap] = featM[b * 3 + c]; }
    }
}
int launch_gather_matched_rows(const uint32_t* pairs, const int32_t* n_pairs, int cap, const double* featS, const double* featM,
                               double* pts1, double* pts2, hipStream_t st) {
    if (cap <= 0) return PCREG_OK;
    hipLaunchKernelGGL(gather_matched_rows_kernel, dim3(std::min(1024, (cap + 255) / 256)), dim3(256), 0, st, pairs, n_pairs, cap, featS, featM, pts1, pts2);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
