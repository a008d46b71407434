// pcreg_amd/csrc/descriptors.hip -- getSpacialHistogramDescriptors on gfx950, fp64.
//
//   getLocalPoints.m:8-35                 radius search (brute force O(S*P) in the reference,
//                                         called twice per keypoint)  -> uniform grid + 27 cells
//   getSpacialHistogramDescriptors.m:75-145  KNN-PCA local reference frame (the inlined twin of
//                                         AlignPoints_KNN: sort key = distance to the LOCAL
//                                         centroid, sign vote over the K kept rows)
//   :118-121                              eigenvalue-ratio rejection
//   :150-171 + histcn.m:94-131            spherical 10 x 7 x 14 count histogram (phi = atan2(y,y))
//   :177-179                              drop rejected keypoints, keep input order
//
// Pipeline (all resident):
//   grid_*       deterministic counting sort of the cloud into cells of edge >= R: per-tile
//                private count rows (no contended atomics), a (cell, tile) exclusive scan, and a
//                scatter whose in-tile rank comes from an LDS bitonic sort of (cell, index)
//                keys -> inside a cell points stay in ascending original index, every run the same.
//   desc_kernel  one workgroup per keypoint: gather the <= 27 cells (coalesced runs), keep
//                |p-c| < R in LDS (index + distance-to-centroid), bisection-select the K nearest to the
//                local centroid (ties by original index = stable sort), 3x3 covariance + Jacobi,
//                sign vote, rotate, bin into an LDS histogram (ds_add_u32), one coalesced row store.
//   compact      order-preserving compaction of the surviving rows into feat / desc (double).
// The dominant traffic is the 980-count row per keypoint: HBM-write-bound.
#include "common.hpp"
#include "wave_math.hpp"
#include "select.hpp"
#include "select_kth.hpp"
#include <cfloat>
#include <cmath>
#include <cstdlib>

namespace pcreg {
namespace {

#ifndef PCREG_DESC_DC
#define PCREG_DESC_DC 8
#endif
#ifndef PCREG_DESC_D64
#define PCREG_DESC_D64 2
#endif
#ifndef PCREG_DESC_D32
#define PCREG_DESC_D32 8
#endif
constexpr int kBlock = 256;
constexpr int kSortTile = 2048;            // points per counting-sort tile
constexpr int kMaxCells = 1 << 16;
constexpr int NR = 10, NT = 7, NP = 14, ND = NR * NT * NP;
static_assert(NT == 7, "bin_fast32 folds the seven theta bins about pi / 2");

struct Grid {
    double ox, oy, oz;                     // origin
    double invx, invy, invz;               // 1 / cell edge per axis: the edges are fractions of h (see grid_setup_kernel)
    double hy, hz;                         // cell edges along y and z
    int nx, ny, nz, ncells;
    int fx, fy, fz;                        // h / edge per axis: a sphere of radius R <= h reaches at most f cells to either side
    double L;                              // the cloud's largest extent: bounds |p - origin| (error bound of the fp32 copies)
};
// ct = cos(t); r2ge[k] = the smallest d2 with sqrt(d2) >= r[k], r2gt = the smallest d2 with sqrt(d2) > r[NR] (the radial
// bin of a point follows from its SQUARED distance, exactly); cts[k] = ct[k] |ct[k]| (theta's bin from z |z| vs cts d2)
struct Edges { double r[NR + 1], t[NT + 1], p[NP + 1], ct[NT + 1], r2ge[NR + 1], r2gt, cts[NT + 1]; };

__device__ __forceinline__ int cell_coord(double v, double o, double inv, int n) {
    int c = (int)floor((v - o) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
__device__ __forceinline__ int cell_of(const Grid& g, double x, double y, double z) {
    return (cell_coord(z, g.oz, g.invz, g.nz) * g.ny + cell_coord(y, g.oy, g.invy, g.ny)) * g.nx + cell_coord(x, g.ox, g.invx, g.nx);
}

// ---- grid: bounding box -> cell size ------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bbox_partial_d_kernel(const double* __restrict__ p, int P, int ld,
                                                                double* __restrict__ part) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < P; i += gridDim.x * kBlock) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { double v = p[i + (size_t)c * ld]; lo[c] = fmin(lo[c], v); hi[c] = fmax(hi[c], v); }
    }
    __shared__ double s[kBlock / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fmin(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmax(hi[c], __shfl_xor(hi[c], o)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = lo[c]; s[threadIdx.x >> 6][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = s[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? fmin(v, s[w][threadIdx.x]) : fmax(v, s[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void grid_setup_kernel(const double* __restrict__ part, int nparts, double R, Grid* __restrict__ g, int force_opt) {
    if (threadIdx.x != 0) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < nparts; ++b)
        for (int c = 0; c < 3; ++c) { lo[c] = fmin(lo[c], part[b * 6 + c]); hi[c] = fmax(hi[c], part[b * 6 + 3 + c]); }
    if (!(hi[0] >= lo[0])) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = 0; }
    double cell = R;                       // h >= R: a sphere reaches at most one h to either side along every axis
    for (;;) {
        double nx = floor((hi[0] - lo[0]) / cell) + 1, ny = floor((hi[1] - lo[1]) / cell) + 1, nz = floor((hi[2] - lo[2]) / cell) + 1;
        if (nx * ny * nz <= (double)kMaxCells) break;
        cell *= 1.26;                      // ~ halve the cell count
    }
    // The cells are a fraction of h wide: a keypoint's neighbourhood is then (2 fy + 1)(2 fz + 1) rows of x-contiguous
    // cells, each cut to the x interval the sphere can reach given the row's slab gaps.  Candidate volume per keypoint
    // (Monte Carlo over keypoint positions, in R^3; the sphere itself is 4.19): (1,1,1) 20.6, (8,1,1) 13.5, (4,2,1) 11.5,
    // (2,2,2) 10.5, (4,2,2) 9.2 -- the finest subdivision that fits the cell budget is taken.
    const int opts[7][3] = {{4, 2, 2}, {2, 2, 2}, {4, 2, 1}, {8, 1, 1}, {4, 1, 1}, {2, 1, 1}, {1, 1, 1}};
    int pick = 6;
    for (int k = force_opt >= 0 ? force_opt : 0; k < 7; ++k) {
        double cnt = 1.0;
        for (int c = 0; c < 3; ++c) cnt *= floor((hi[c] - lo[c]) * (opts[k][c] / cell)) + 1;     // the very expression that sizes the grid below
        if (cnt <= (double)kMaxCells) { pick = k; break; }
    }
    g->fx = opts[pick][0]; g->fy = opts[pick][1]; g->fz = opts[pick][2];
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2];
    g->invx = g->fx / cell; g->invy = g->fy / cell; g->invz = g->fz / cell;
    g->hy = cell / g->fy; g->hz = cell / g->fz;
    g->nx = (int)(floor((hi[0] - lo[0]) * g->invx) + 1); g->ny = (int)(floor((hi[1] - lo[1]) * g->invy) + 1); g->nz = (int)(floor((hi[2] - lo[2]) * g->invz) + 1);
    g->ncells = g->nx * g->ny * g->nz;
    g->L = fmax(hi[0] - lo[0], fmax(hi[1] - lo[1], hi[2] - lo[2]));
}

// ---- grid: deterministic counting sort ------------------------------------------------------
// counts[tile][cell] (row private to the tile's workgroup: atomics without cross-block races)
__global__ __launch_bounds__(kBlock) void grid_count_kernel(const double* __restrict__ p, int P, int ld,
                                                            const Grid* __restrict__ gp, int32_t* __restrict__ cell_id,
                                                            int32_t* __restrict__ counts) {
    const Grid g = *gp;
    const int t0 = blockIdx.x * kSortTile;
    int32_t* row = counts + (size_t)blockIdx.x * kMaxCells;
    for (int i = t0 + threadIdx.x; i < min(P, t0 + kSortTile); i += kBlock) {
        int c = cell_of(g, p[i], p[i + (size_t)ld], p[i + 2 * (size_t)ld]);
        cell_id[i] = c;
        atomicAdd(&row[c], 1);             // integer adds commute: the row is order-independent
    }
}
// per cell: prefix over tiles (in place) and the cell total
__global__ void grid_cell_prefix_kernel(int32_t* __restrict__ counts, int ntiles, const Grid* __restrict__ gp,
                                        int32_t* __restrict__ cell_total) {
    const int nc = gp->ncells;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) { if (c < kMaxCells + 1) cell_total[c] = 0; return; }
    int run = 0;
    for (int t = 0; t < ntiles; ++t) { int v = counts[(size_t)t * kMaxCells + c]; counts[(size_t)t * kMaxCells + c] = run; run += v; }
    cell_total[c] = run;
}
// exclusive scan of cell totals -> cell_start[0..kMaxCells] (one workgroup)
__global__ void grid_cell_scan_kernel(const int32_t* __restrict__ cell_total, int32_t* __restrict__ cell_start) {
    // thread t owns cells [t * 256, (t + 1) * 256): local sums, one 256-wide scan, local replay
    __shared__ int s[256];
    constexpr int kPer = kMaxCells / 256;
    const int base = threadIdx.x * kPer;
    int sum = 0;
    for (int k = 0; k < kPer; ++k) sum += cell_total[base + k];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    int run = s[threadIdx.x] - sum;
    for (int k = 0; k < kPer; ++k) { const int v = cell_total[base + k]; cell_start[base + k] = run; run += v; }
    if (threadIdx.x == 255) cell_start[kMaxCells] = run;
}
// scatter: in-tile rank from a bitonic sort of (cell << 32 | index) keys
__global__ __launch_bounds__(kBlock) void grid_scatter_kernel(const double* __restrict__ p, int P, int ld,
                                                              const int32_t* __restrict__ cell_id,
                                                              const int32_t* __restrict__ counts /*prefix over tiles*/,
                                                              const int32_t* __restrict__ cell_start,
                                                              int32_t* __restrict__ sorted_idx, double* __restrict__ sx,
                                                              double* __restrict__ sy, double* __restrict__ sz,
                                                              const Grid* __restrict__ gp, int single_mode, float4* __restrict__ f4) {
    __shared__ unsigned long long key[kSortTile];
    __shared__ int run_start[kSortTile];
    const int t0 = blockIdx.x * kSortTile;
    const int cnt = min(kSortTile, P - t0);
    for (int i = threadIdx.x; i < kSortTile; i += kBlock)
        key[i] = i < cnt ? (((unsigned long long)(unsigned)cell_id[t0 + i] << 32) | (unsigned)(t0 + i)) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= kSortTile; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < kSortTile; i += kBlock) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = key[i], b = key[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // start of each run of equal cells (max-scan of run heads)
    for (int i = threadIdx.x; i < kSortTile; i += kBlock)
        run_start[i] = (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)) ? i : 0;
    __syncthreads();
    for (int o = 1; o < kSortTile; o <<= 1) {
        int v[kSortTile / kBlock];
#pragma unroll
        for (int r = 0; r < kSortTile / kBlock; ++r) { int i = r * kBlock + threadIdx.x; v[r] = i >= o ? max(run_start[i], run_start[i - o]) : run_start[i]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSortTile / kBlock; ++r) run_start[r * kBlock + threadIdx.x] = v[r];
        __syncthreads();
    }
    const int32_t* row = counts + (size_t)blockIdx.x * kMaxCells;
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
        unsigned long long kk = key[i];
        int c = (int)(kk >> 32), idx = (int)(kk & 0xFFFFFFFFull);
        int dst = cell_start[c] + row[c] + (i - run_start[i]);
        sorted_idx[dst] = idx;
        const double X = p[idx], Y = p[idx + (size_t)ld], Z = p[idx + 2 * (size_t)ld];
        sx[dst] = X; sy[dst] = Y; sz[dst] = Z;
        // the fp32 copies desc_kernel streams: single data -> the single values themselves (MATLAB's arithmetic runs on them),
        // double data -> relative to the grid's origin (screening only: small magnitudes, whatever the georeference)
        // (one 16-byte element per point: a single load instruction and address per gather -- the kernel is bound by
        // instruction issue, not by bytes)
        if (single_mode) f4[dst] = make_float4((float)X, (float)Y, (float)Z, 0.0f);
        else f4[dst] = make_float4((float)(X - gp->ox), (float)(Y - gp->oy), (float)(Z - gp->oz), 0.0f);
    }
}

// ---- keypoints in cell order ---------------------------------------------------------------------
// Workgroup b describes keypoint perm[b]; perm lists the keypoints cell by cell, so that workgroups that
// run at the same time read the same few cells of the sorted cloud (L2 hits instead of Infinity-Cache
// round trips).  The order INSIDE a cell is whatever the atomics give: it only schedules work, every
// result is written to its keypoint's own row.
__global__ void kp_count_kernel(const double* __restrict__ kp, int S, int ldk, const Grid* __restrict__ gp,
                                int32_t* __restrict__ kcell, int32_t* __restrict__ kc_total) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int c = cell_of(*gp, kp[s], kp[s + (size_t)ldk], kp[s + 2 * (size_t)ldk]);
    kcell[s] = c;
    atomicAdd(&kc_total[c], 1);
}
__global__ void kp_scatter_kernel(const int32_t* __restrict__ kcell, int S, const int32_t* __restrict__ kc_start,
                                  int32_t* __restrict__ kc_fill, int32_t* __restrict__ perm) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int c = kcell[s];
    perm[kc_start[c] + atomicAdd(&kc_fill[c], 1)] = s;
}

// ---- per-keypoint descriptor ------------------------------------------------------------------
// block-wide sums of a 256-thread workgroup: DPP wave sums (wave_math.hpp: no LDS crossbar), then every thread adds the
// four wave partials in the same order, (w0 + w1) + (w2 + w3) (broadcast LDS reads), so all threads hold the same bits;
// one barrier pair per call
template <int N>
__device__ __forceinline__ void bsum_n(double (&v)[N], double* s_redn /*[4][N]*/) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum_dpp(v[k]);
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) s_redn[(threadIdx.x >> 6) * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = (s_redn[k] + s_redn[N + k]) + (s_redn[2 * N + k] + s_redn[3 * N + k]);
    __syncthreads();
}
__device__ __forceinline__ int bsum_i(int v, int* s_red) {
    v = wave_sum_dpp_i(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return t;
}
// histcounts bin (histcn.m:108): e[k-1] <= x < e[k], last bin closed, 0 = outside / NaN
template <int NE>
__device__ __forceinline__ int hist_loc(double x, const double (&e)[NE]) {
    if (!(x >= e[0]) || !(x <= e[NE - 1])) return 0;
    if (x == e[NE - 1]) return NE - 1;
    int k = 1;
#pragma unroll
    for (int j = 1; j < NE - 1; ++j) k += (x >= e[j]);
    return k;
}

// fp32 images of the bin edges (screening): the squared radial edges and cos |cos| of the theta edges
struct Edges32 { float r2ge[NR + 1], r2gt, cts[NT + 1]; };

// the smallest double t with sqrt(t) >= r under IEEE round-to-nearest (r > 0 finite): sqrt(d2) < r <=> d2 < t
__host__ __device__ inline double sqrt_threshold(double r) {
    if (!(r > 0.0) || !(r < INFINITY)) return r > 0.0 ? r : 0.0;       // r <= 0 or NaN: nothing is inside; +inf: everything finite
    double t = r * r;
    for (int it = 0; it < 64 && t > 0.0 && sqrt(t) >= r; ++it) t = nextafter(t, 0.0);
    for (int it = 0; it < 64 && sqrt(t) < r; ++it) t = nextafter(t, INFINITY);
    return t;
}

// SM = 0: double inputs, double arithmetic (fp32 only SCREENS decisions that fp64 re-takes when they are close).
// SM = 1: MATLAB's arithmetic for `single` data (either input single: getLocalPoints.m:8-31 then runs in single): the open
//         box test, pts_cube - c, sqrt(x^2 + y^2 + z^2) and dists < R are evaluated in fp32 exactly as written, which decides
//         WHICH keypoints survive and which points form a support; the support's coordinates are MATLAB's single pts_rel
//         values.  Everything after that (mean, pca, the histogram) stays in double on those values: the summation order of
//         MATLAB's single mean / pca is not knowable.
template <int SM>
__global__ __launch_bounds__(kBlock, 4) void desc_kernel(
    const double* __restrict__ sx, const double* __restrict__ sy, const double* __restrict__ sz,
    const float4* __restrict__ f4,
    const int32_t* __restrict__ sorted_idx, const int32_t* __restrict__ cell_start, const Grid* __restrict__ gp,
    const double* __restrict__ kp, const int32_t* __restrict__ perm, int S, int ldk, pcreg_desc_opts o, const Edges* __restrict__ edp, Edges32 e32, double R2T,
    int kp_single, int cap, int dbg_stop, int xcd_chunk, void* __restrict__ rows_out /*[S][ND] u32 or u16*/, int rows_u16,
    int32_t* __restrict__ valid, int32_t* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* lpos = reinterpret_cast<int*>(smem);                           // [cap] position in the sorted arrays
    __shared__ double s_redn[4 * 9];
    __shared__ int s_redi[4];
    __shared__ unsigned s_cnt[ND];
    __shared__ unsigned long long s_u64[8];
    constexpr int kSmall = 256;
    __shared__ unsigned long long s_small[kSmall];
    __shared__ int s_hist[256];
    __shared__ unsigned long long s_vk, s_tlo, s_thi;
    __shared__ int s_nsmall, s_bin, s_below;
    __shared__ int s_ok;
    __shared__ double s_m[9];
    __shared__ double s_V[9];
    constexpr int kDef = 128;
    __shared__ int s_def[kDef], s_vdef[kDef];
    __shared__ int s_ndef, s_nvdef;

    const Grid g = *gp;
    // the fp64 bin edges stay in device memory: the literal path indexes them dynamically (hist_loc), and a by-value struct
    // would be copied to scratch by every workgroup for it (496 B per lane at kernel entry: measured 1 ms per 100 k keypoints)
    const Edges& ed = *edp;
    // workgroups are dealt round-robin over the 8 XCDs: XCD x walks its own eighth of the cell-ordered keypoints, so its
    // L2 holds the few cells its ~128 concurrent keypoints share instead of a slice of everybody's (PCREG_DESC_XCD=0: off)
    int slot = blockIdx.x;
    if (xcd_chunk > 0) { slot = (blockIdx.x & 7) * xcd_chunk + (blockIdx.x >> 3); if ((int)(blockIdx.x >> 3) >= xcd_chunk || slot >= S) return; }
    const int s = perm[slot];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // an SGPR: the loops over a wave's chunks are scalar
    const double cx = kp[s], cy = kp[s + (size_t)ldk], cz = kp[s + 2 * (size_t)ldk];
    const double R = o.R;
    // the keypoint in the frame of the fp32 copies: SM 0: relative to the grid's origin (screening), SM 1: the single value itself
    const float cxf = SM ? (float)cx : (float)(cx - g.ox), cyf = SM ? (float)cy : (float)(cy - g.oy), czf = SM ? (float)cz : (float)(cz - g.oz);
    const float R32 = (float)R;
    // SM 1, getLocalPoints.m:8-13: xLim = c(1) + [-R, R] is single when c is single (R rounds to single first), double otherwise
    // -- and then rounds to single in the comparison with the single cloud
    const float xlo = kp_single ? cxf + (-R32) : (float)(cx - R), xhi = kp_single ? cxf + R32 : (float)(cx + R);
    const float ylo32 = kp_single ? cyf + (-R32) : (float)(cy - R), yhi32 = kp_single ? cyf + R32 : (float)(cy + R);
    const float zlo32 = kp_single ? czf + (-R32) : (float)(cz - R), zhi32 = kp_single ? czf + R32 : (float)(cz + R);
    // the fp32 copies' error per coordinate of (p - c), u = 2^-24: SM 0: copy and keypoint are rounded relative to the grid's
    // origin (u L each, L = the cloud's extent + R) and subtracted (one more rounding); SM 1: the copies ARE the data
    const float u32 = 5.9604644775390625e-08f;
    const float dlt = SM ? 0.0f : 2.0f * u32 * (float)(g.L + R) + 2.0f * u32 * (2.0f * R32);
    if (tid == 0) { valid[s] = 0; s_ndef = 0; s_nvdef = 0; }
    for (int i = tid; i < ND; i += kBlock) s_cnt[i] = 0u;
    __syncthreads();

    // ---- getLocalPoints: |p - c| < R (strict), from the cells around c ----
    // The grid's cells are a fraction of R wide (Grid::fx, fy, fz per axis).  A ROW = the cells of one (z, y) slab pair,
    // contiguous along x in the sorted arrays: at most (2 fy + 1)(2 fz + 1) <= kRowsMax rows can hold points of the sphere,
    // and each is cut to the x interval the sphere can reach given the row's slab gaps.  Rows are split into chunks of
    // 64 consecutive points; wave w takes the chunks w, w + 4, ... of the whole sequence (balanced whatever the row
    // lengths), the list is wave-major (all of wave 0's points, then wave 1's, ...): deterministic, and the later
    // passes do not care -- ties go by ORIGINAL index.
    constexpr int kDC = PCREG_DESC_DC, kD64 = PCREG_DESC_D64, kD32 = PCREG_DESC_D32;     // gathers in flight per lane and pass
    constexpr int kRowsMax = 25;
    __shared__ int s_rowb[kRowsMax], s_rowe[kRowsMax], s_cp[kRowsMax + 1], s_wtot[4];
    const int nyo = g.fy, nzo = g.fz, wy = 2 * nyo + 1, nrows = wy * (2 * nzo + 1);
    static_assert(kRowsMax < 64, "the rows of a keypoint are resolved by the lanes of wave 0");
    if (wave == 0) {
        int b = 0, e = 0;
        if (tid < nrows) {
        const int ky = cell_coord(cy, g.oy, g.invy, g.ny), kz = cell_coord(cz, g.oz, g.invz, g.nz);
        const int zz = kz + tid / wy - nzo, yy = ky + tid % wy - nyo;
        // a keypoint outside the cloud's box by more than R has no neighbours: all its rows are then farther than R
        if (zz >= 0 && zz < g.nz && yy >= 0 && yy < g.ny) {
            // the row's slab in y and z keeps every point at least (gy, gz) away from the keypoint: what is left of R^2
            // bounds |x - cx|.  The margins cover the rounding of the slab edges (oy + yy hy: about an ulp of the largest
            // term, which for georeferenced clouds is the coordinate itself, not the cell) and of w; cell_coord is
            // monotone in x.  SM 1 widens R by the single-precision slack of its own test.
            const double Rr = SM ? R * (1.0 + 1e-6) : R;
            const double ylo = g.oy + yy * g.hy, zlo = g.oz + zz * g.hz;
            const double my_ = 1e-9 * g.hy + 8.0 * DBL_EPSILON * (fabs(g.oy) + fabs(cy) + fabs(yy * g.hy)) + (SM ? 2e-7 * (fabs(cy) + R) : 0.0);
            const double mz_ = 1e-9 * g.hz + 8.0 * DBL_EPSILON * (fabs(g.oz) + fabs(cz) + fabs(zz * g.hz)) + (SM ? 2e-7 * (fabs(cz) + R) : 0.0);
            const double gy = fmax(0.0, fmax(ylo - cy, cy - (ylo + g.hy)) - my_), gz = fmax(0.0, fmax(zlo - cz, cz - (zlo + g.hz)) - mz_);
            const double w2 = Rr * Rr - gy * gy - gz * gz;
            if (w2 >= 0.0) {
                const double w = sqrt(w2) + 1e-9 * R + 8.0 * DBL_EPSILON * (fabs(g.ox) + fabs(cx) + R) + (SM ? 2e-7 * (fabs(cx) + R) : 0.0);
                const int x0 = cell_coord(cx - w, g.ox, g.invx, g.nx), x1 = cell_coord(cx + w, g.ox, g.invx, g.nx);
                b = cell_start[(zz * g.ny + yy) * g.nx + x0]; e = cell_start[(zz * g.ny + yy) * g.nx + x1 + 1];   // x-adjacent cells are contiguous
            }
        }
        }
        // chunk prefix over the rows: one DPP scan over wave 0's lanes (a serial loop by one thread cost ~50 dependent LDS round
        // trips at the start of every keypoint)
        const int cnt = (e - b + 63) >> 6;              // 0 for the lanes past nrows
        const int incl = wave_scan_incl_i(cnt);
        if (tid < nrows) { s_rowb[tid] = b; s_rowe[tid] = e; s_cp[tid] = incl - cnt; }
        if (tid == 63) s_cp[nrows] = incl;
    }
    __syncthreads();
    const int n_chunks = __builtin_amdgcn_readfirstlane(s_cp[nrows]);
    // pass 1 streams the candidates ONCE (four 64-point chunks = twelve loads in flight per wave) and keeps every chunk's
    // ballot; pass 2 replays the ballots, so it touches no global memory.  The kernel is bound by VALU issue
    // (profiles/r03_pmc_desc_kernel.json), so the loop is written for few vector instructions per chunk: chunk numbers and
    // ranges live in SGPRs, the ballots of a wave's chunks sit in the lanes of three register pairs (v_writelane / v_readlane:
    // no LDS traffic), the cloud is addressed as scalar base + 32-bit byte offset, the centroid's sums run under the ballot as
    // the exec mask (a branch, not six v_cndmask per chunk).
    constexpr int kMaskCap = 192;                   // chunks per wave whose ballot is kept (beyond: re-tested)
    __shared__ unsigned long long s_mask[4][kMaskCap];
    int blo = 0, bhi = 0;                           // lane l: the ballot of the wave's chunk 64 round + l (flushed to s_mask per round)
    auto keep_ballot = [&](int ch /* scalar */, unsigned long long bal) {
        // (value and lane select both in SGPRs exceed gfx9's one-scalar-operand limit: the lane select goes through m0)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // "m0 is reserved": it is, and this statement says that it overwrites it
        asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                     : "+v"(blo), "+v"(bhi) : "s"((int)(unsigned)bal), "s"((int)(unsigned)(bal >> 32)), "s"(ch & 63) : "m0");
#pragma clang diagnostic pop
    };
    auto flush_ballots = [&](int round /* scalar */) {
        if (round * 64 < kMaskCap) s_mask[wave][round * 64 + lane] = ((unsigned long long)(unsigned)bhi << 32) | (unsigned)blo;
    };
    // chunk c -> (first position, end of its row).  The lookup walks two LDS tables, and doing it inside the streaming loop
    // put ~4 dependent LDS round trips in front of every chunk's loads (the collection took 1.7 ms per 100 k keypoints for
    // 0.3 ms worth of loads and tests).  Every lane therefore resolves ONE chunk of its wave up front -- lane l owns the
    // wave's l-th chunk, chunks wave + 4 l -- and the loop reads the pair with v_readlane (beyond 64 chunks per wave a
    // second, third ... round of lookups follows).
    auto chunk_lookup = [&](int c, int& j0, int& end) {
        int rp = 0;
        while (c >= s_cp[rp + 1]) ++rp;
        j0 = s_rowb[rp] + ((c - s_cp[rp]) << 6); end = s_rowe[rp];
    };
    int my_j0 = 0, my_end = 0, my_round = -1;
    auto chunk_range = [&](int ord /* the wave's ord-th chunk: scalar */, int& j0, int& end) {
        const int round = ord >> 6;
        if (round != my_round) {                       // scalar
            my_round = round;
            const int c = wave + 4 * (round * 64 + lane);
            my_j0 = 0; my_end = 0;
            if (c < n_chunks) chunk_lookup(c, my_j0, my_end);
        }
        j0 = __builtin_amdgcn_readlane(my_j0, ord & 63); end = __builtin_amdgcn_readlane(my_end, ord & 63);
    };
    // the cloud through a scalar base and a 32-bit byte offset (global_load ... v_off, s[base]: no 64-bit address arithmetic
    // per load; the launcher refuses clouds of 2^28 points or more)
    auto ldd = [](const double* __restrict__ base, unsigned off) -> double { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off); };
    auto ldf4 = [&](unsigned off8 /* 8 j */) -> float4 { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(f4) + 2u * off8); };
    // is candidate j inside?  SM 1: (X, Y, Z) the single point, (x, y, z) = pts_cube - c in single (getLocalPoints.m:8-25 as
    // MATLAB runs it for single data); SM 0: the fp64 test on the double coordinates
    auto inside = [&](unsigned off, float X, float Y, float Z, float x, float y, float z) -> bool {
        if (SM) {
            const bool box = X > xlo && X < xhi && Y > ylo32 && Y < yhi32 && Z > zlo32 && Z < zhi32;
            return box && sqrtf((x * x + y * y) + z * z) < R32;
        }
        const double ex = ldd(sx, off) - cx, ey = ldd(sy, off) - cy, ez = ldd(sz, off) - cz;
        return ex * ex + ey * ey + ez * ez < R2T;                          // getLocalPoints.m:23-25 (see R2T)
    };
    double a3[3] = {0, 0, 0};
    {
        int cnt = 0;                                 // scalar: popcounts of ballots
        // SM 0 streams the DOUBLE coordinates once (test + centroid sums from the same registers).  A form that screened
        // the test on the fp32 copy and fetched the doubles of the inside lanes only was 1.6 ms per 100 k keypoints slower:
        // with 40 % of the candidates inside, every sector of the double arrays is touched anyway, on top of the copy.
        constexpr int kD = SM ? kDC : 4;             // chunks in flight per wave
        static_assert(64 % kD == 0, "a round of 64 chunks is a whole number of groups");
        for (int c0 = wave, ch0 = 0; c0 < n_chunks; c0 += 4 * kD, ch0 += kD) {
            if (ch0 != 0 && (ch0 & 63) == 0) flush_ballots((ch0 >> 6) - 1);      // scalar
            double X[SM ? 1 : kD], Y[SM ? 1 : kD], Z[SM ? 1 : kD]; float Xf[SM ? kD : 1], Yf[SM ? kD : 1], Zf[SM ? kD : 1];
            int J0[kD], E[kD];
#pragma unroll
            for (int u = 0; u < kD; ++u) {
                J0[u] = 0; E[u] = 0;
                if (c0 + 4 * u < n_chunks) chunk_range(ch0 + u, J0[u], E[u]);        // scalar
                const unsigned off = (unsigned)max(min(J0[u] + lane, E[u] - 1), 0) << 3;
                if constexpr (SM == 0) { X[u] = ldd(sx, off); Y[u] = ldd(sy, off); Z[u] = ldd(sz, off); }
                else { const float4 t4 = ldf4(off); Xf[u] = t4.x; Yf[u] = t4.y; Zf[u] = t4.z; }
            }
#pragma unroll
            for (int u = 0; u < kD; ++u) {
                if (c0 + 4 * u < n_chunks) {         // scalar
                    unsigned long long bal;
                    if constexpr (SM == 0) {
                        const double x = X[u] - cx, y = Y[u] - cy, z = Z[u] - cz;
                        // lanes past the end of the row tested a clamped (repeated) point: masked out in the scalar unit
                        const int live = E[u] - J0[u];
                        const unsigned long long lanes = live >= 64 ? ~0ull : (1ull << live) - 1ull;
                        bal = __builtin_amdgcn_ballot_w64(x * x + y * y + z * z < R2T) & lanes;   // getLocalPoints.m:23-25 (see R2T)
                        if (__builtin_amdgcn_inverse_ballot_w64(bal)) { asm volatile("" ::: "memory"); a3[0] += x; a3[1] += y; a3[2] += z; }   // the local centroid's sums ride along (:80)
                    } else {
                        const float x = Xf[u] - cxf, y = Yf[u] - cyf, z = Zf[u] - czf;
                        const int live = E[u] - J0[u];
                        const unsigned long long lanes = live >= 64 ? ~0ull : (1ull << live) - 1ull;
                        bal = __builtin_amdgcn_ballot_w64(inside(0u, Xf[u], Yf[u], Zf[u], x, y, z)) & lanes;
                        if (__builtin_amdgcn_inverse_ballot_w64(bal)) { asm volatile("" ::: "memory"); a3[0] += (double)x; a3[1] += (double)y; a3[2] += (double)z; }   // MATLAB's single pts_rel values (:80 sums them)
                    }
                    keep_ballot(ch0 + u, bal);
                    cnt += __builtin_popcountll(bal);
                }
            }
        }
        if (n_chunks > wave) flush_ballots(((n_chunks - wave + 3) / 4 - 1) >> 6);     // the last (partial) round
        if (lane == 0) s_wtot[wave] = cnt;
    }
    __syncthreads();
    int n_total = 0, my_base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int t = __builtin_amdgcn_readfirstlane(s_wtot[w]); if (w == wave) my_base = n_total; n_total += t; }
    const int n = n_total;
    if (n < 1 || n < o.min_pts || n > o.max_pts) return;                   // getLocalPoints.m:17,31
    if (n > cap) { if (tid == 0) atomicMax(err, n); return; }              // support larger than the LDS list
    {
        int pos = my_base;                           // scalar
        for (int c = wave, ch = 0; c < n_chunks; c += 4, ++ch) {
            int j0, end;
            chunk_range(ch, j0, end);
            unsigned long long bal;
            if ((ch & 63) == 0 && ch < kMaskCap) { const unsigned long long t = s_mask[wave][ch + lane]; blo = (int)(unsigned)t; bhi = (int)(unsigned)(t >> 32); }   // scalar
            if (ch < kMaskCap) bal = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(bhi, ch & 63) << 32) | (unsigned)__builtin_amdgcn_readlane(blo, ch & 63);
            else {
                const int j = j0 + lane; const unsigned off = (unsigned)max(min(j, end - 1), 0) << 3;
                float X = 0.0f, Y = 0.0f, Z = 0.0f;
                if (SM) { const float4 t4 = ldf4(off); X = t4.x; Y = t4.y; Z = t4.z; }
                bal = __builtin_amdgcn_ballot_w64(j < end && inside(off, X, Y, Z, X - cxf, Y - cyf, Z - czf));
            }
            // the list holds BYTE offsets into the double columns (8 x the sorted position)
            if (__builtin_amdgcn_inverse_ballot_w64(bal))
                lpos[pos + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (j0 + lane) << 3;
            pos += __builtin_popcountll(bal);
        }
    }
    __syncthreads();

    if (dbg_stop == 1) return;            // timing experiments only (PCREG_DESC_STOP)
    // A pass over the support: thread t visits its entries i = t + 256 r in ascending order; px/py/pz = the
    // point relative to the keypoint (SM 1: MATLAB's single pts_rel value, widened), psel = kept by the K-nearest selection
    // (bit r of selmask).  Four entries' gathers are issued before the first is used: the sorted cloud sits in the L2 /
    // Infinity Cache, so the passes are latency-bound and memory-level parallelism is what they need.
    unsigned selmask = 0xFFFFFFFFu;                  // cap <= 8191 -> r < 32
#define PCREG_LOAD_REL(j_ /* byte offset into the double columns */, X_, Y_, Z_)                      \
    if (SM) { const float4 t4_ = ldf4((unsigned)(j_)); X_ = (double)(t4_.x - cxf); Y_ = (double)(t4_.y - cyf); Z_ = (double)(t4_.z - czf); } \
    else { X_ = ldd(sx, (unsigned)(j_)) - cx; Y_ = ldd(sy, (unsigned)(j_)) - cy; Z_ = ldd(sz, (unsigned)(j_)) - cz; }
#define PCREG_MY_PTS(...)                                                                            \
    for (int i0_ = tid, r0_ = 0; i0_ < n; i0_ += kD64 * kBlock, r0_ += kD64) {                       \
        double X_[kD64], Y_[kD64], Z_[kD64];                                                         \
        _Pragma("unroll") for (int u_ = 0; u_ < kD64; ++u_) {                                        \
            const int j_ = lpos[min(i0_ + u_ * kBlock, n - 1)];                                      \
            PCREG_LOAD_REL(j_, X_[u_], Y_[u_], Z_[u_])                                               \
        }                                                                                            \
        _Pragma("unroll") for (int u_ = 0; u_ < kD64; ++u_) {                                        \
            const int i = i0_ + u_ * kBlock, pr = r0_ + u_;                                          \
            if (i < n) {                                                                             \
                const double px = X_[u_], py = Y_[u_], pz = Z_[u_];                                  \
                const bool psel = (selmask >> pr) & 1u; (void)psel; (void)i;                         \
                __VA_ARGS__                                                                          \
            }                                                                                        \
        }                                                                                            \
    }

    // ---- K nearest to the local centroid (:75-85) ----
    const bool all = o.k >= 1.0;
    const int K = all ? n : (int)floor(n * o.k + 0.5);
    if (K < 2) return;
    bsum_n<3>(a3, s_redn);                          // summed while the candidates streamed by (collection pass 1)
    const double gx = a3[0] / n, gy = a3[1] / n, gz = a3[2] / n;
    if (!all) {
        // The reference sorts dists = vecnorm(pts_local - centroid) (stable: ties keep their original order).  The sort key
        // here is the SQUARED distance d2 (the argument of that square root; no fp64 sqrt per point): sqrt is monotone, so
        // the K-th smallest distance is sK = sqrt(K-th smallest d2), and the points tied WITH it in the reference's order are
        // exactly those whose d2 lies in sqrt's preimage of sK, [t_lo, t_hi) -- found with a handful of square roots by one
        // thread.  Selection: d2 < t_lo, plus the lowest original indices of the tie group.
#define PCREG_KEY(px, py, pz) kth_key(((px) - gx) * ((px) - gx) + ((py) - gy) * ((py) - gy) + ((pz) - gz) * ((pz) - gz))
        // every thread keeps the keys of ITS entries in registers (one more gather pass, fully unrolled so
        // that the register index is a constant); the selection itself then never touches memory
        unsigned long long kreg[32];
#pragma unroll
        for (int t_ = 0; t_ < 32 / kD64; ++t_) {
            if (t_ * kD64 * kBlock >= n) continue;     // block-uniform: the unrolled loop is sized for 8191 entries, a typical support holds half (PCREG_MY_KEYS never reads the slots past n)
            double X_[kD64], Y_[kD64], Z_[kD64];
#pragma unroll
            for (int u_ = 0; u_ < kD64; ++u_) {
                const int j_ = lpos[min(tid + (t_ * kD64 + u_) * kBlock, n - 1)];
                PCREG_LOAD_REL(j_, X_[u_], Y_[u_], Z_[u_])
            }
#pragma unroll
            for (int u_ = 0; u_ < kD64; ++u_) kreg[t_ * kD64 + u_] = PCREG_KEY(X_[u_], Y_[u_], Z_[u_]);
        }
#define PCREG_MY_KEYS(...) _Pragma("unroll") for (int pg_ = 0; pg_ < 8; ++pg_) { if (pg_ * 4 * kBlock < n) { /* block-uniform */ \
        _Pragma("unroll") for (int pq_ = 0; pq_ < 4; ++pq_) { const int pr = pg_ * 4 + pq_; const int i = tid + pr * kBlock; if (i < n) { const unsigned long long k = kreg[pr]; (void)i; __VA_ARGS__ } } } }
        // K-th smallest by ONE 256-bin histogram over [min, max] (the bin index is monotone in the
        // key), then an exact rank inside the bin that holds it; bisection only if that bin is crowded
        // the keys are bit patterns of non-negative doubles: min / max as doubles (v_min_f64 / v_max_f64, DPP in wave_math.hpp)
        double lo_d = DBL_MAX, hi_d = 0.0;
        PCREG_MY_KEYS(const double kd = __longlong_as_double((long long)k); lo_d = fmin(lo_d, kd); hi_d = fmax(hi_d, kd);)
        unsigned long long lo = kth_key(wave_min_dpp(lo_d)), hi = kth_key(wave_max_dpp(hi_d));
        if (lane == 0) { s_u64[wave] = lo; s_u64[4 + wave] = hi; }
        for (int i = tid; i < 256; i += kBlock) s_hist[i] = 0;
        if (tid == 0) { s_nsmall = 0; s_vk = 0ull; }
        __syncthreads();
        lo = s_u64[0]; hi = s_u64[4];
#pragma unroll
        for (int w = 1; w < 4; ++w) { lo = s_u64[w] < lo ? s_u64[w] : lo; hi = s_u64[4 + w] > hi ? s_u64[4 + w] : hi; }
        const double dlo = __longlong_as_double((long long)lo), dhi = __longlong_as_double((long long)hi);
        const double scale = dhi > dlo ? 256.0 / (dhi - dlo) : 0.0;
        auto bin_of = [&](unsigned long long k) -> int {
            const int bb = (int)((__longlong_as_double((long long)k) - dlo) * scale);
            return bb > 255 ? 255 : bb;
        };
        if (dbg_stop == 4) return;
        unsigned kb[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};         // the keys' bins, four to a register (the second look costs a bit-field extract)
        PCREG_MY_KEYS(const int bb = bin_of(k); atomicAdd(&s_hist[bb], 1); kb[pr >> 2] |= (unsigned)bb << (8 * (pr & 3));)
        __syncthreads();
        if (dbg_stop == 5) return;
        if (wave == 0) {                             // the bin of the K-th and the number of entries below that bin
            int c4[4], run = 0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { c4[q4] = s_hist[lane * 4 + q4]; run += c4[q4]; }
            const int incl = wave_scan_incl_i(run);
            int before = incl - run;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                if (before < K && K <= before + c4[q4]) { s_bin = lane * 4 + q4; s_below = before; }
                before += c4[q4];
            }
        }
        __syncthreads();
        const int bstar = s_bin, below = s_below;
        PCREG_MY_KEYS(if ((int)((kb[pr >> 2] >> (8 * (pr & 3))) & 255u) == bstar) { const int q = atomicAdd(&s_nsmall, 1); if (q < kSmall) s_small[q] = k; })
        __syncthreads();
        const int m = s_nsmall, Kp = K - below;
        unsigned long long vK;
        if (m <= kSmall) {
            for (int t = tid; t < m; t += kBlock) {
                const unsigned long long x = s_small[t];
                int less = 0, eq = 0;
                for (int u = 0; u < m; ++u) { const unsigned long long yv = s_small[u]; less += yv < x; eq += yv == x; }
                if (less < Kp && Kp <= less + eq) s_vk = x;     // every writer writes the same
            }
            __syncthreads();
            vK = s_vk;
        } else {                                     // crowded bin (many equal distances): plain bisection on the keys
            unsigned long long blo = lo, bhi = hi;
            while (blo < bhi) {
                const unsigned long long mid = blo + ((bhi - blo) >> 1);
                int c = 0;
                PCREG_MY_KEYS(c += k <= mid;)
                c = bsum_i(c, s_redi);
                if (c >= K) bhi = mid; else blo = mid + 1;
            }
            vK = blo;
        }
        // sqrt's preimage of the K-th distance: every d2 in [t_lo, t_hi) has the reference's sort key sK
        if (tid == 0) {
            const double sK = sqrt(__longlong_as_double((long long)vK));
            s_tlo = kth_key(sqrt_threshold(sK));
            s_thi = kth_key(sqrt_threshold(nextafter(sK, INFINITY)));
        }
        __syncthreads();
        const unsigned long long t_lo = s_tlo, t_hi = s_thi;
        // n_less = #{d2 < t_lo}, n_eq = #{d2 in the tie group}.  The bin index is monotone in the key: when the whole tie group
        // lies in the K-th's bin (always, but for a group that straddles a bin border), the keys of the lower bins are all
        // below t_lo, those of the higher bins all >= t_hi, and the counts follow from the bin's <= 256 entries in s_small --
        // one entry per thread, two ballots -- instead of another pass over every key and two block reductions.
        int n_less, n_eq;
        if (m <= kSmall && bin_of(t_lo > lo ? t_lo : lo) == bstar && bin_of(t_hi - 1ull < hi ? t_hi - 1ull : hi) == bstar) {   // block-uniform
            static_assert(kSmall <= kBlock, "one s_small entry per thread");
            const unsigned long long ks = tid < m ? s_small[tid] : ~0ull;
            const unsigned long long b1 = __builtin_amdgcn_ballot_w64(ks < t_lo), b2 = __builtin_amdgcn_ballot_w64(ks >= t_lo && ks < t_hi);
            if (lane == 0) s_redi[wave] = __builtin_popcountll(b1) | (__builtin_popcountll(b2) << 16);
            __syncthreads();
            const int t = s_redi[0] + s_redi[1] + s_redi[2] + s_redi[3];
            __syncthreads();
            n_less = below + (t & 0xFFFF); n_eq = t >> 16;
        } else {
            int c1 = 0, c2 = 0;
            PCREG_MY_KEYS(c1 += k < t_lo; c2 += (k >= t_lo && k < t_hi);)
            n_less = bsum_i(c1, s_redi); n_eq = bsum_i(c2, s_redi);
        }
        if (dbg_stop == 6) return;
        const int take_eq = K - n_less;
        // ties at the K-th distance: the stable sort keeps the lowest ORIGINAL indices
        unsigned sm = 0u;
        if (n_eq == take_eq) {                       // the whole tie group is kept (nearly always a group of one): one comparison per key
            PCREG_MY_KEYS(sm |= (k < t_hi ? 1u : 0u) << pr;)
        } else {
        PCREG_MY_KEYS(
            const unsigned long long key = k;
            bool sel = key < t_lo;
            if (key >= t_lo && key < t_hi) {
                {
                    const int me = sorted_idx[lpos[i] >> 3];
                    int rank = 0;
                    for (int t = 0; t < n; ++t) {
                        const int jt = lpos[t];
                        double tx, ty, tz;
                        PCREG_LOAD_REL(jt, tx, ty, tz)
                        const unsigned long long kt = PCREG_KEY(tx, ty, tz);
                        if (kt >= t_lo && kt < t_hi && sorted_idx[jt >> 3] < me) ++rank;
                    }
                    sel = rank < take_eq;
                }
            }
            sm |= (sel ? 1u : 0u) << pr;)
        }
        selmask = sm;
#undef PCREG_KEY
#undef PCREG_MY_KEYS
    }

    if (dbg_stop == 2) return;
    // ---- pca(pts_k, 'eig') (:91) ----
    // ONE pass for the mean and the covariance of the K kept points: first and second moments of q = p - g about the
    // local centroid g (known since the selection; the kept points are the K nearest to it, so their mean is a small
    // fraction of their spread away and cov = (sum q q' - K d d') / (K - 1), d = sum q / K, loses nothing to cancellation)
    double mo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    PCREG_MY_PTS(
        if (psel) {
            const double x = px - gx, y = py - gy, z = pz - gz;
            mo[0] += x; mo[1] += y; mo[2] += z;
            mo[3] += x * x; mo[4] += x * y; mo[5] += x * z; mo[6] += y * y; mo[7] += y * z; mo[8] += z * z;
        })
    bsum_n<9>(mo, s_redn);
    if (dbg_stop == 7) return;
    const double dx = mo[0] / K, dy = mo[1] / K, dz = mo[2] / K;
    const double mx = gx + dx, my = gy + dy, mz = gz + dz;
    double cv[6] = {mo[3] - K * dx * dx, mo[4] - K * dx * dy, mo[5] - K * dx * dz, mo[6] - K * dy * dy, mo[7] - K * dy * dz, mo[8] - K * dz * dz};
#pragma unroll
    for (int k = 0; k < 6; ++k) cv[k] = cv[k] / (double)(K - 1);
    if (tid == 0) {
        double a[6] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5]};
        jacobi_sym3(a, s_V);
        double ev[3] = {a[0], a[3], a[5]};
        int od[3] = {0, 1, 2};
        if (ev[od[1]] > ev[od[0]]) { int t = od[0]; od[0] = od[1]; od[1] = t; }
        if (ev[od[2]] > ev[od[0]]) { int t = od[0]; od[0] = od[2]; od[2] = t; }
        if (ev[od[2]] > ev[od[1]]) { int t = od[1]; od[1] = od[2]; od[2] = t; }
        double v0 = ev[od[0]], v1 = ev[od[1]], v2 = ev[od[2]];
        s_ok = !((v0 / v1 < o.thVar[0]) || (v1 / v2 < o.thVar[1]));        // :118-121
        for (int col = 0; col < 3; ++col) {
            const double a0 = s_V[od[col]], a1 = s_V[3 + od[col]], a2 = s_V[6 + od[col]];
            double big = a0;
            if (fabs(a1) > fabs(big)) big = a1;
            if (fabs(a2) > fabs(big)) big = a2;
            double sg = big < 0 ? -1.0 : 1.0;
            s_m[0 * 3 + col] = sg * a0; s_m[1 * 3 + col] = sg * a1; s_m[2 * 3 + col] = sg * a2;
        }
    }
    __syncthreads();
    if (!s_ok || dbg_stop == 8) return;
    double cu[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cu[k] = s_m[k];
    // ---- sign vote (:128-145) and spherical histogram over ALL local points (:150-171, histcn.m:108-131) ----
    // Per point: r = |p|, theta = acos(z / r), phi = atan2(y, y) (sic, :152), then histcounts on each.  None of the three
    // needs its transcendental -- or even the square root and the division -- for the bin:
    //  * r's bin from d2 = x^2 + y^2 + z^2 against the exact squared images of the edges (Edges::r2ge);
    //  * phi only depends on the sign of y (two constants, binned once);
    //  * theta's bin from z |z| against cos(edge) |cos(edge)| d2, both monotone images of z / r and cos(edge).
    // With ALIGN_POINTS the vote and the histogram share ONE pass.  The vote only flips signs: column 0 by xs, column 2
    // by zs, and column 1 is scaled by ys = det(the flipped matrix) (:53) = xs zs det(the unflipped one) -- every term of
    // the determinant holds one entry of each column, so the flips factor out of the rounded expression exactly.  The
    // pass therefore bins every point in the frame (c0, c1 det0, c2), whose coordinates are the final ones up to the
    // signs (xs, xs zs, zs), bit for bit, and the row is written through the permutation the signs induce: theta's bins
    // mirror, phi's two populated bins swap.
    // Round 3: that pass runs in fp32 -- the point relative to the keypoint from the fp32 copies (4-byte gathers instead of
    // 8), rotation and bin tests at the full vector rate -- as a SCREEN: every comparison carries the bound of its fp32
    // error (e_rot per rotated coordinate: input error dlt per coordinate, 3 products and 2 sums at one ulp each on
    // |p| <= R, the matrix entries rounded to fp32), and a point whose comparisons do not all clear their bound is deferred
    // and binned literally in fp64 (sqrt, division, acos / atan2) in the final frame, as is any vote whose sign fp32 cannot
    // certify.  A cleared comparison has the sign of the exact one, so the counts are those of the all-fp64 form.
    const double ph_pos = atan2(1.0, 1.0), ph_neg = atan2(-1.0, -1.0);
    const int lp_pos = hist_loc<NP + 1>(ph_pos, ed.p), lp_neg = hist_loc<NP + 1>(ph_neg, ed.p);
    const float e_rot = 3.0f * dlt + 10.0f * u32 * R32;
    // Few instructions per point matter here (the kernel is bound by VALU issue: 87 % busy, profiles/r03_pmc_desc_kernel.json):
    //  * the radial bins are equal-volume, edge k = cbrt(k R^3 / NR): the bin is floor(t) + 1 with t = NR (r / R)^3, safe
    //    when t keeps its integer part under the error of d2 (the rounded edges sit 1e-16 off this formula: inside the margin);
    //  * the theta edges are symmetric about pi / 2 (cos(pi - x) = -cos x): three comparisons of z^2 with cos^2(edge) d2
    //    count the edges between the point and its pole, the sign of z picks the side (near z = 0 both sides give bin 4).
    //  * one v_rsq_f32 gives r = d2 / sqrt(d2) and every RELATIVE error term (e / r): no IEEE square root (15 instructions) and
    //    no IEEE division (11) per point; explicit fma chains (fewer roundings than the bounds assume).
    const float k3 = (float)((double)NR / (R * R * R));
    const int base_pos = lp_pos > 0 ? NR * NT * (lp_pos - 1) : -(1 << 20), base_neg = lp_neg > 0 ? NR * NT * (lp_neg - 1) : -(1 << 20);
    const float e2_rot = e_rot * e_rot, e_rot2 = 2.0f * e_rot;
    auto bin_fast32 = [&](float x, float y, float z, bool& safe) -> int {
        const float a = z * z;
        const float d2 = fmaf(x, x, fmaf(y, y, a));
        const float rinv = __builtin_amdgcn_rsqf(d2);                                      // 1 ulp; d2 = 0: inf -> NaN bounds -> deferred
        const float r = d2 * rinv;                                                          // sqrt(d2) to 3 u
        const float q = e_rot * rinv;
        // |d2 - exact| / d2 <= rel: 2 sqrt(3) r e + 3 e^2 from the inputs, 6 u d2 from the three products and two sums
        const float rel = fmaf(q, fmaf(q, 3.0f, 3.7f), 6.0f * u32);
        const float t = d2 * r * k3;                                                        // NR (r / R)^3
        const float e_t = fmaf(t, fmaf(rel, 1.5f, 12.0f * u32), 2e-6f);                     // d(r^3) / r^3 = 1.5 d(d2) / d2
        const float fr = __builtin_amdgcn_fractf(t);
        safe = fabsf(y) > e_rot && fr > e_t && fr + e_t < 1.0f && t + e_t < (float)NR && rel < 1.0f;
        // |(z^2 - c^2 d2) - exact| <= e_a, c^2 <= 1
        const float e_a = fmaf(d2, rel + 4.0f * u32, fmaf(r, e_rot2, e2_rot));
        int m = 0;
#pragma unroll
        for (int jj = 1; jj <= 3; ++jj) { const float dl = fmaf(-e32.cts[jj], d2, a); safe = safe && fabsf(dl) > e_a; m += dl < 0.0f; }
        const int ltm = z >= 0.0f ? m : 6 - m;                                              // lt - 1
        return (int)t + NR * ltm + (y > 0.0f ? base_pos : base_neg);                        // (lr - 1) + NR (lt - 1) + NR NT (lp - 1); lr <= NR when safe
    };
    // a deferred point is first binned with the fp64 images (no transcendental); only inside THEIR 1e-12 band -- where the
    // computed acos could land on the other side of an edge -- it takes the literal path
    auto bin_fast64 = [&](double x, double y, double z, bool& safe) -> int {
        const double xy2 = x * x + y * y; const double d2 = xy2 + z * z;
        const double zq = z * fabs(z);
        safe = xy2 > 1e-12 * d2 && y != 0.0;
        int lt = 1;
#pragma unroll
        for (int jj = 1; jj < NT; ++jj) { const double dl = zq - ed.cts[jj] * d2; safe = safe && fabs(dl) > 1e-12 * d2; lt += dl < 0.0; }
        int lr = 0;
        if (d2 < ed.r2gt) {
            lr = 1;
#pragma unroll
            for (int jj = 1; jj < NR; ++jj) lr += d2 >= ed.r2ge[jj];
            if (!(d2 >= ed.r2ge[0])) lr = 0;
        }
        const int lp = y > 0.0 ? lp_pos : lp_neg;
        return (lr > 0 && lp > 0) ? (lr - 1) + NR * (lt - 1) + NR * NT * (lp - 1) : -1;
    };
    auto bin_literal = [&](double x, double y, double z) -> int {
        const double d2 = x * x + y * y + z * z;
        const double r = sqrt(d2); const double u = z / r;
        const int lt = hist_loc<NT + 1>(acos(u), ed.t), lr = hist_loc<NR + 1>(r, ed.r), lp = hist_loc<NP + 1>(atan2(y, y), ed.p);
        return (lr > 0 && lt > 0 && lp > 0) ? (lr - 1) + NR * (lt - 1) + NR * NT * (lp - 1) : -1;
    };
    // the fp32 pass over the support: ax/ay/az = the point relative to the keypoint in fp32 (SM 1: exactly pts_rel)
#define PCREG_MY_PTS32(...)                                                                          \
    for (int i0_ = tid, r0_ = 0; i0_ < n; i0_ += kD32 * kBlock, r0_ += kD32) {                       \
        float X_[kD32], Y_[kD32], Z_[kD32];                                                          \
        _Pragma("unroll") for (int u_ = 0; u_ < kD32; ++u_) {                                        \
            const int j_ = lpos[min(i0_ + u_ * kBlock, n - 1)];                                      \
            { const float4 t4_ = ldf4((unsigned)j_); X_[u_] = t4_.x; Y_[u_] = t4_.y; Z_[u_] = t4_.z; }       \
        }                                                                                            \
        _Pragma("unroll") for (int u_ = 0; u_ < kD32; ++u_) {                                        \
            const int i = i0_ + u_ * kBlock, pr = r0_ + u_;                                          \
            if (i < n) {                                                                             \
                const float ax = X_[u_] - cxf, ay = Y_[u_] - cyf, az = Z_[u_] - czf;                 \
                const bool psel = (selmask >> pr) & 1u; (void)psel; (void)i;                         \
                __VA_ARGS__                                                                          \
            }                                                                                        \
        }                                                                                            \
    }
    bool permuted = false;
    double xs = 1.0, zs = 1.0;
    if (o.ALIGN_POINTS) {
        const double det0 = cu[0] * (cu[4] * cu[8] - cu[5] * cu[7]) - cu[1] * (cu[3] * cu[8] - cu[5] * cu[6]) + cu[2] * (cu[3] * cu[7] - cu[4] * cu[6]);
        const double c1x = cu[1] * det0, c1y = cu[4] * det0, c1z = cu[7] * det0;
        const float f0 = (float)cu[0], f3 = (float)cu[3], f6 = (float)cu[6], f2 = (float)cu[2], f5 = (float)cu[5], f8 = (float)cu[8];
        const float g1x = (float)c1x, g1y = (float)c1y, g1z = (float)c1z;
        // a vote is the sign of (p - mean) . column = p . column - mean . column: the first term is the rotated coordinate the
        // histogram needs anyway (error e_rot), the second a constant per keypoint (fp64, rounded once: u R), their
        // difference one more rounding on values up to 2 R
        const float m0f = (float)(mx * cu[0] + my * cu[3] + mz * cu[6]), m2f = (float)(mx * cu[2] + my * cu[5] + mz * cu[8]);
        const float e_vote = e_rot + 4.0f * u32 * R32;
        int votes = 0;                              // vx | vz << 16 (K <= n <= 8191)
        PCREG_MY_PTS32(
            const float x0 = fmaf(az, f6, fmaf(ay, f3, ax * f0)), y0 = fmaf(az, g1z, fmaf(ay, g1y, ax * g1x)), z0 = fmaf(az, f8, fmaf(ay, f5, ax * f2));
            bool need = false;
            if (psel) {
                const float vx = x0 - m0f, vz = z0 - m2f;
                if (fabsf(vx) > e_vote && fabsf(vz) > e_vote) votes += (vx > 0.0f ? 1 : 0) + (vz > 0.0f ? 1 << 16 : 0);
                else need = true;
            }
            if (need) { const int qv = atomicAdd(&s_nvdef, 1); if (qv < kDef) s_vdef[qv] = i; }
            bool safe; const int bb = bin_fast32(x0, y0, z0, safe);
            if (safe) { if (bb >= 0) atomicAdd(&s_cnt[bb], 1u); }
            else { const int q = atomicAdd(&s_ndef, 1); if (q < kDef) s_def[q] = i; })
        __syncthreads();                            // completes s_cnt, s_ndef, s_nvdef
        const int nvdef = s_nvdef;
        if (nvdef <= kDef) {                        // the votes fp32 could not certify: the fp64 expressions
            for (int qv = tid; qv < nvdef; qv += kBlock) {
                double px, py, pz;
                PCREG_LOAD_REL(lpos[s_vdef[qv]], px, py, pz)
                const double x = px - mx, y = py - my, z = pz - mz;
                votes += ((x * cu[0] + y * cu[3] + z * cu[6]) > 0 ? 1 : 0) + ((x * cu[2] + y * cu[5] + z * cu[8]) > 0 ? 1 << 16 : 0);
            }
        } else {                                    // (never seen: a support whose kept points all sit on the frame's planes) all votes in fp64
            votes = 0;
            PCREG_MY_PTS(
                if (psel) {
                    const double x = px - mx, y = py - my, z = pz - mz;
                    votes += ((x * cu[0] + y * cu[3] + z * cu[6]) > 0 ? 1 : 0) + ((x * cu[2] + y * cu[5] + z * cu[8]) > 0 ? 1 << 16 : 0);
                })
        }
        votes = bsum_i(votes, s_redi);
        const int vx = votes & 0xFFFF, vz = votes >> 16;
        xs = (2.0 * vx >= (double)K) ? 1.0 : -1.0; zs = (2.0 * vz >= (double)K) ? 1.0 : -1.0;
        double M[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) { M[r * 3] = cu[r * 3] * xs; M[r * 3 + 1] = cu[r * 3 + 1]; M[r * 3 + 2] = cu[r * 3 + 2] * zs; }
        double ys = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
#pragma unroll
        for (int r = 0; r < 3; ++r) { cu[r * 3] = cu[r * 3] * xs; cu[r * 3 + 1] = cu[r * 3 + 1] * ys; cu[r * 3 + 2] = cu[r * 3 + 2] * zs; }
        if (dbg_stop == 3) return;
        permuted = s_ndef <= kDef;
        if (!permuted) {                            // too many deferred points: bin everything again, literally, in the final frame
            __syncthreads();
            for (int i = tid; i < ND; i += kBlock) s_cnt[i] = 0u;
            __syncthreads();
        }
    } else {
        // no alignment: the points are binned as they are (identity frame); same screen, deferred points literally
        PCREG_MY_PTS32(
            bool safe; const int bb = bin_fast32(ax, ay, az, safe);
            if (safe) { if (bb >= 0) atomicAdd(&s_cnt[bb], 1u); }
            else { const int q = atomicAdd(&s_ndef, 1); if (q < kDef) s_def[q] = i; })
        __syncthreads();
        permuted = s_ndef <= kDef;                  // (identity permutation: xs = zs = 1)
        if (!permuted) {
            for (int i = tid; i < ND; i += kBlock) s_cnt[i] = 0u;
            __syncthreads();
        }
    }
    if (permuted) {
        // the deferred points: literally, in the final frame, into the bins of the UNPERMUTED histogram's image
        const int ndef = s_ndef;
        unsigned* s_fin = reinterpret_cast<unsigned*>(&s_mask[0][0]);     // the collection's ballots are dead: 6 KiB >= 980 words
        static_assert(sizeof(unsigned long long) * 4 * kMaskCap >= sizeof(unsigned) * ND, "s_fin aliases s_mask");
        if (tid < NT * NP) {                         // one (theta, phi) run of NR radial bins per thread: no division per bin
            const int lt1 = tid % NT, lp1 = tid / NT;
            const int st = zs < 0.0 ? NT - 1 - lt1 : lt1;
            int sp = lp1;
            if (xs * zs < 0.0) sp = lp1 == lp_pos - 1 ? lp_neg - 1 : (lp1 == lp_neg - 1 ? lp_pos - 1 : lp1);
#pragma unroll
            for (int lr1 = 0; lr1 < NR; ++lr1) s_fin[lr1 + NR * tid] = s_cnt[lr1 + NR * st + NR * NT * sp];
        }
        __syncthreads();
        for (int q = tid; q < ndef; q += kBlock) {
            double px, py, pz;
            PCREG_LOAD_REL(lpos[s_def[q]], px, py, pz)
            double x = px, y = py, z = pz;
            if (o.ALIGN_POINTS) { x = px * cu[0] + py * cu[3] + pz * cu[6]; y = px * cu[1] + py * cu[4] + pz * cu[7]; z = px * cu[2] + py * cu[5] + pz * cu[8]; }
            bool safe64; int bb = bin_fast64(x, y, z, safe64);
            if (!safe64) bb = bin_literal(x, y, z);
            if (bb >= 0) atomicAdd(&s_fin[bb], 1u);
        }
        __syncthreads();
        // ONE write of the finished row, in its final type (u16 rows: counts <= max_pts <= 65535)
        if (rows_u16) {                              // two counts per 4-byte store (a row is 1960 bytes: 4-byte aligned)
            static_assert(ND % 2 == 0, "u16 rows are written as pairs");
            uint32_t* row = reinterpret_cast<uint32_t*>((uint16_t*)rows_out + (size_t)s * ND);
            for (int i = tid; i < ND / 2; i += kBlock) row[i] = (s_fin[2 * i] & 0xFFFFu) | (s_fin[2 * i + 1] << 16);
        } else { uint32_t* row = (uint32_t*)rows_out + (size_t)s * ND; for (int i = tid; i < ND; i += kBlock) row[i] = s_fin[i]; }
    } else {
        PCREG_MY_PTS(
            double x = px; double y = py; double z = pz;
            if (o.ALIGN_POINTS) { x = px * cu[0] + py * cu[3] + pz * cu[6]; y = px * cu[1] + py * cu[4] + pz * cu[7]; z = px * cu[2] + py * cu[5] + pz * cu[8]; }
            bool safe64; int bb = bin_fast64(x, y, z, safe64);
            if (!safe64) bb = bin_literal(x, y, z);
            if (bb >= 0) atomicAdd(&s_cnt[bb], 1u);)
        __syncthreads();
        if (rows_u16) { uint16_t* row = (uint16_t*)rows_out + (size_t)s * ND; for (int i = tid; i < ND; i += kBlock) row[i] = (uint16_t)s_cnt[i]; }
        else { uint32_t* row = (uint32_t*)rows_out + (size_t)s * ND; for (int i = tid; i < ND; i += kBlock) row[i] = s_cnt[i]; }
    }
    if (tid == 0) valid[s] = 1;
#undef PCREG_MY_PTS
#undef PCREG_MY_PTS32
#undef PCREG_LOAD_REL
}

// the bin edges are computed on the HOST (the oracle's libm values) and parked in the workspace by one thread
__global__ void edges_store_kernel(Edges ed, Edges* __restrict__ out) { *out = ed; }

// ---- compaction of the surviving rows (:177-179) ----------------------------------------------
__global__ void desc_count_kernel(const int32_t* __restrict__ valid, int S, int32_t* __restrict__ block_cnt) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool kp = s < S && valid[s] != 0;
    __shared__ int s_c[4];
    unsigned long long b = __ballot(kp);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
__global__ void desc_slot_kernel(const int32_t* __restrict__ valid, int S, const int32_t* __restrict__ block_off,
                                 int32_t* __restrict__ slot) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool kp = s < S && valid[s] != 0;
    __shared__ int s_c[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long b = __ballot(kp);
    if (lane == 0) s_c[wave] = __popcll(b);
    __syncthreads();
    int base = block_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_c[w];
    if (s < S) slot[s] = kp ? base + __popcll(b & ((1ull << lane) - 1ull)) : -1;
}
// desc != nullptr: the MATLAB-shaped output -- row v of desc = the v-th surviving keypoint's counts as doubles (from the
// u16 staging rows; counts <= the support size <= 8191).  desc == nullptr: the rows stay where desc_kernel wrote them (row s =
// keypoint s) and row_index[v] = s lists the survivors.  feat rows are compact either way.
__global__ __launch_bounds__(kBlock) void desc_emit_kernel(const uint16_t* __restrict__ rows, const int32_t* __restrict__ slot,
                                                           const double* __restrict__ kp, int S, int ldk,
                                                           double* __restrict__ feat, double* __restrict__ desc,
                                                           int32_t* __restrict__ row_index) {
    const int s = blockIdx.x;
    const int v = slot[s];
    if (v < 0) return;
    if (desc) {
        const uint16_t* row = rows + (size_t)s * ND;
        double* out = desc + (size_t)v * ND;
        for (int i = threadIdx.x; i < ND; i += kBlock) out[i] = (double)row[i];
    }
    if (threadIdx.x < 3) feat[(size_t)v * 3 + threadIdx.x] = kp[s + (size_t)threadIdx.x * ldk];
    if (row_index && threadIdx.x == 0) row_index[v] = s;
}
// the index form needs no pass over the rows: one thread per keypoint
__global__ void desc_index_kernel(const int32_t* __restrict__ slot, const double* __restrict__ kp, int S, int ldk,
                                  double* __restrict__ feat, int32_t* __restrict__ row_index) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int v = slot[s];
    if (v < 0) return;
    for (int c = 0; c < 3; ++c) feat[(size_t)v * 3 + c] = kp[s + (size_t)c * ldk];
    row_index[v] = s;
}

}  // namespace

// workspace: Grid | Edges | bbox partials | cell_total | cell_start | cell_id [P] | counts [tiles][kMaxCells]
//            | sorted_idx [P] | sx sy sz [P] (f64) | f4 [P] (float4) | staging rows u16 [S][980] | valid [S] | slot [S]
//            | block counters | keypoint cell counts / starts / fill | keypoint cell [S] | perm [S]
size_t descriptors_workspace_bytes(int P, int S) {
    size_t p = (size_t)(P > 0 ? P : 1), s = (size_t)(S > 0 ? S : 1);
    size_t tiles = (p + kSortTile - 1) / kSortTile;
    return 256 + 1024 + align_up(512 * 6 * 8, 256) + 2 * align_up(((size_t)kMaxCells + 1) * 4, 256) + align_up(p * 4, 256) +
           align_up(tiles * kMaxCells * 4, 256) + align_up(p * 4, 256) + 3 * align_up(p * 8, 256) + align_up(p * 16, 256) +
           align_up(s * ND * 2, 256) + 2 * align_up(s * 4, 256) + align_up((s / 256 + 2) * 4, 256) +
           3 * align_up(((size_t)kMaxCells + 1) * 4, 256) + 2 * align_up(s * 4, 256);
}

// single_mode: 0 = double data; 1 = single data, keypoints single; 2 = single cloud, double keypoints (MATLAB: "single wins",
// but getLocalPoints.m:8-10's xLim = c + [-R, R] is then formed in double).  The inputs are double arrays either way
// (single data widened exactly).  Output forms: desc_f64 (compact doubles, MATLAB's shape) or rows_u16 + row_index (rows in
// keypoint order, written ONCE by desc_kernel, + the ascending list of the survivors).
int launch_descriptors(const double* pts, int P, int ld, const double* kp, int S, int ldk, const pcreg_desc_opts& o, int single_mode,
                       double* feat, double* desc_f64, uint16_t* rows_u16, int32_t* row_index, int32_t* V_dev, int32_t* err_dev,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    PCREG_ARG(P >= 0 && S >= 0 && o.R > 0 && o.k > 0 && o.min_pts >= 0 && single_mode >= 0 && single_mode <= 2);
    PCREG_ARG((desc_f64 != nullptr) != (rows_u16 != nullptr) && (rows_u16 == nullptr || row_index != nullptr));
    PCREG_HIP(hipMemsetAsync(V_dev, 0, sizeof(int32_t), st));
    PCREG_HIP(hipMemsetAsync(err_dev, 0, sizeof(int32_t), st));
    if (S == 0 || P == 0) return PCREG_OK;
    // desc_kernel addresses the sorted cloud as scalar base + 32-bit byte offset (16 P bytes of float4 copies)
    if (P >= (1 << 28)) { set_error("descriptors: clouds of 2^28 points or more are not supported (got %d)", P); return PCREG_E_ARG; }
    size_t need = descriptors_workspace_bytes(P, S);
    if (ws_bytes < need) { set_error("descriptor workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    size_t p = (size_t)P, s = (size_t)S;
    const int tiles = (P + kSortTile - 1) / kSortTile;
    char* w = (char*)ws;
    Grid* grid = (Grid*)w;                  w += 256;
    Edges* edges_dev = (Edges*)w;           w += 1024;
    static_assert(sizeof(Edges) <= 1024, "Edges fits its slot");
    double* bpart = (double*)w;             w += align_up(512 * 6 * 8, 256);
    int32_t* cell_total = (int32_t*)w;      w += align_up(((size_t)kMaxCells + 1) * 4, 256);
    int32_t* cell_start = (int32_t*)w;      w += align_up(((size_t)kMaxCells + 1) * 4, 256);
    int32_t* cell_id = (int32_t*)w;         w += align_up(p * 4, 256);
    int32_t* counts = (int32_t*)w;          w += align_up((size_t)tiles * kMaxCells * 4, 256);
    int32_t* sorted_idx = (int32_t*)w;      w += align_up(p * 4, 256);
    double* sx = (double*)w;                w += align_up(p * 8, 256);
    double* sy = (double*)w;                w += align_up(p * 8, 256);
    double* sz = (double*)w;                w += align_up(p * 8, 256);
    float4* f4 = (float4*)w;                w += align_up(p * 16, 256);
    uint16_t* stage = (uint16_t*)w;         w += align_up(s * ND * 2, 256);
    int32_t* valid = (int32_t*)w;           w += align_up(s * 4, 256);
    int32_t* slot = (int32_t*)w;            w += align_up(s * 4, 256);
    int32_t* bcnt = (int32_t*)w;            w += align_up((s / 256 + 2) * 4, 256);
    int32_t* kc_total = (int32_t*)w;        w += align_up(((size_t)kMaxCells + 1) * 4, 256);
    int32_t* kc_start = (int32_t*)w;        w += align_up(((size_t)kMaxCells + 1) * 4, 256);
    int32_t* kc_fill = (int32_t*)w;         w += align_up(((size_t)kMaxCells + 1) * 4, 256);
    int32_t* kcell = (int32_t*)w;           w += align_up(s * 4, 256);
    int32_t* perm = (int32_t*)w;

    int nb = (P + kBlock * 16 - 1) / (kBlock * 16); if (nb > 512) nb = 512; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(bbox_partial_d_kernel, dim3(nb), dim3(kBlock), 0, st, pts, P, ld, bpart);
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(64), 0, st, bpart, nb, o.R, grid, PCREG_EXP_ENV("PCREG_DESC_SUBDIV", -1));
    PCREG_HIP(hipMemsetAsync(counts, 0, (size_t)tiles * kMaxCells * 4, st));
    hipLaunchKernelGGL(grid_count_kernel, dim3(tiles), dim3(kBlock), 0, st, pts, P, ld, grid, cell_id, counts);
    hipLaunchKernelGGL(grid_cell_prefix_kernel, dim3((kMaxCells + 1 + 255) / 256), dim3(256), 0, st, counts, tiles, grid, cell_total);
    hipLaunchKernelGGL(grid_cell_scan_kernel, dim3(1), dim3(256), 0, st, cell_total, cell_start);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(tiles), dim3(kBlock), 0, st, pts, P, ld, cell_id, counts, cell_start,
                       sorted_idx, sx, sy, sz, (const Grid*)grid, single_mode != 0 ? 1 : 0, f4);
    // keypoints in cell order (kc_total | kc_start | kc_fill are contiguous: one memset)
    PCREG_HIP(hipMemsetAsync(kc_total, 0, 3 * align_up(((size_t)kMaxCells + 1) * 4, 256), st));
    hipLaunchKernelGGL(kp_count_kernel, dim3((S + 255) / 256), dim3(256), 0, st, kp, S, ldk, grid, kcell, kc_total);
    hipLaunchKernelGGL(grid_cell_scan_kernel, dim3(1), dim3(256), 0, st, kc_total, kc_start);
    hipLaunchKernelGGL(kp_scatter_kernel, dim3((S + 255) / 256), dim3(256), 0, st, kcell, S, kc_start, kc_fill, perm);
    PCREG_HIP(hipGetLastError());

    Edges ed; Edges32 e32;
    const double r3 = o.R * o.R * o.R, pi = 3.14159265358979323846;
    for (int k = 0; k <= NR; ++k) ed.r[k] = cbrt(k * (r3 / NR));                 // nthroot(0:R^3/10:R^3, 3)
    for (int k = 0; k <= NT; ++k) { ed.t[k] = k * (pi / NT); ed.ct[k] = cos(ed.t[k]); ed.cts[k] = ed.ct[k] * fabs(ed.ct[k]); }   // 0:pi/7:pi
    for (int k = 0; k <= NR; ++k) ed.r2ge[k] = sqrt_threshold(ed.r[k]);
    ed.r2gt = sqrt_threshold(std::nextafter(ed.r[NR], INFINITY));
    for (int k = 0; k <= NP; ++k) ed.p[k] = -pi + k * (2 * pi / NP);             // -pi:2*pi/14:pi
    for (int k = 0; k <= NR; ++k) e32.r2ge[k] = (float)ed.r2ge[k];
    e32.r2gt = (float)ed.r2gt;
    for (int k = 0; k <= NT; ++k) e32.cts[k] = (float)ed.cts[k];
    int cap = o.max_pts < 8190 ? o.max_pts + 1 : 8191;
    if (cap < 64) cap = 64;
    size_t lds = (size_t)cap * sizeof(int);
    const int xcd_chunk = PCREG_EXP_ENV("PCREG_DESC_XCD", 1) ? (S + 7) / 8 : 0;
    const double R2T = sqrt_threshold(o.R);
    hipLaunchKernelGGL(edges_store_kernel, dim3(1), dim3(1), 0, st, ed, edges_dev);
    void* rows_out = rows_u16 ? (void*)rows_u16 : (void*)stage;
    const int dbg = PCREG_EXP_ENV("PCREG_DESC_STOP", 0);
    if (single_mode) {
        PCREG_HIP(hipFuncSetAttribute((const void*)desc_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(desc_kernel<1>, dim3(xcd_chunk ? 8 * xcd_chunk : S), dim3(kBlock), lds, st, sx, sy, sz, (const float4*)f4, sorted_idx, cell_start, grid, kp, perm,
                           S, ldk, o, (const Edges*)edges_dev, e32, R2T, single_mode == 1 ? 1 : 0, cap, dbg, xcd_chunk, rows_out, 1, valid, err_dev);
    } else {
        PCREG_HIP(hipFuncSetAttribute((const void*)desc_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(desc_kernel<0>, dim3(xcd_chunk ? 8 * xcd_chunk : S), dim3(kBlock), lds, st, sx, sy, sz, (const float4*)f4, sorted_idx, cell_start, grid, kp, perm,
                           S, ldk, o, (const Edges*)edges_dev, e32, R2T, 0, cap, dbg, xcd_chunk, rows_out, 1, valid, err_dev);
    }
    PCREG_HIP(hipGetLastError());
    const int nbs = (S + 255) / 256;
    hipLaunchKernelGGL(desc_count_kernel, dim3(nbs), dim3(256), 0, st, valid, S, bcnt);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, st, bcnt, nbs, V_dev);
    hipLaunchKernelGGL(desc_slot_kernel, dim3(nbs), dim3(256), 0, st, valid, S, bcnt, slot);
    if (desc_f64) hipLaunchKernelGGL(desc_emit_kernel, dim3(S), dim3(kBlock), 0, st, (const uint16_t*)stage, slot, kp, S, ldk, feat, desc_f64, row_index);
    else hipLaunchKernelGGL(desc_index_kernel, dim3(nbs), dim3(256), 0, st, slot, kp, S, ldk, feat, row_index);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
