// pcreg_amd/csrc/sweep.hip -- the small device pieces of the sphere-sweep driver
// (completeExperimentFast.m:46-225) and of its final stage (:291, :356-394):
//   sphere_counts   #{ ||feat - c|| < R } for many centres        getLocalPoints.m:23-34 as used at :57-64
//   sphere_select   getDescriptorMask (:435-439) as an ordered index list
//   gather_rows     featCur / descCur = X(mask, :)                 :122-125
//   quick_tf        quickTF.m:5-7
// All of it is streaming fp64 work of a few MB: HBM/latency-bound, no tuning beyond coalescing.
#include "common.hpp"
#include "select.hpp"

namespace pcreg {
namespace {

// the smallest double t with sqrt(t) >= r under IEEE round-to-nearest (r > 0 finite): sqrt(d2) < r <=> d2 < t  (descriptors.hip has
// the same function for getLocalPoints)
__host__ __device__ inline double sphere_sqrt_threshold(double r) {
    if (!(r > 0.0) || !(r < INFINITY)) return r > 0.0 ? r : 0.0;       // r <= 0 or NaN: nothing is inside; +inf: everything finite
    double t = r * r;
    for (int it = 0; it < 64 && t > 0.0 && sqrt(t) >= r; ++it) t = nextafter(t, 0.0);
    for (int it = 0; it < 64 && sqrt(t) < r; ++it) t = nextafter(t, INFINITY);
    return t;
}
// dists = vecnorm(feat - c, 2, 2): sqrt((dx^2 + dy^2) + dz^2), compared with strict '<' -- as a comparison of the square root's
// ARGUMENT with R2T = sphere_sqrt_threshold(R): the same decision for every input (the rounded square root is monotone), without
// an fp64 square root per (sphere, keypoint)
__device__ __forceinline__ bool in_sphere(const double* __restrict__ feat, int i, double cx, double cy, double cz, double R2T) {
    const double dx = feat[(size_t)i * 3] - cx, dy = feat[(size_t)i * 3 + 1] - cy, dz = feat[(size_t)i * 3 + 2] - cz;
    return (dx * dx + dy * dy) + dz * dz < R2T;
}

// one workgroup per centre
__global__ __launch_bounds__(1024) void sphere_counts_kernel(const double* __restrict__ feat, int V, const double* __restrict__ centres,
                                                             int S, double R2T, int32_t* __restrict__ counts) {
    const int s = blockIdx.x;
    const double cx = centres[(size_t)s * 3], cy = centres[(size_t)s * 3 + 1], cz = centres[(size_t)s * 3 + 2];
    int c = 0;
#pragma unroll 4
    for (int i = threadIdx.x; i < V; i += 1024) c += in_sphere(feat, i, cx, cy, cz, R2T);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __shared__ int sc[16];
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += sc[w]; counts[s] = t; }
}

__global__ __launch_bounds__(256) void sphere_flag_kernel(const double* __restrict__ feat, int V, double cx, double cy, double cz, double R /* the threshold R2T */,
                                                          int32_t* __restrict__ flag, int32_t* __restrict__ block_cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in = i < V && in_sphere(feat, i, cx, cy, cz, R);
    if (i < V) flag[i] = in;
    __shared__ int sc[4];
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}
__global__ __launch_bounds__(256) void sphere_scatter_kernel(const int32_t* __restrict__ flag, int V, const int32_t* __restrict__ block_off,
                                                             int32_t* __restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in = i < V && flag[i] != 0;
    __shared__ int sc[4];
    const unsigned long long b = __ballot(in);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sc[wave] = __popcll(b);
    __syncthreads();
    int base = block_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += sc[w];
    if (in) idx[base + __popcll(b & ((1ull << lane) - 1ull))] = i;
}

// getDescriptorMask for ALL spheres in one launch: one workgroup per sphere walks the keypoints in order and writes the
// ascending 0-based row list of sphere s at idx[seg_off[s] ..] (seg_off from sphere_counts: the same predicate, so the
// lengths agree; a longer list would be cut) and, optionally, featCur = feat(mask, :) beside it.
__global__ __launch_bounds__(1024) void sphere_select_batched_kernel(const double* __restrict__ feat, int V, const double* __restrict__ centres,
                                                                     double R /* the threshold R2T */, const int32_t* __restrict__ seg_off, int32_t* __restrict__ idx,
                                                                     double* __restrict__ feat_out, int32_t* __restrict__ n_out) {
    __shared__ int sc[16];
    __shared__ int s_base;
    const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double cx = centres[(size_t)s * 3], cy = centres[(size_t)s * 3 + 1], cz = centres[(size_t)s * 3 + 2];
    const int off = seg_off[s], cap = seg_off[s + 1] - off;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < V; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const bool in = i < V && in_sphere(feat, i, cx, cy, cz, R);
        const unsigned long long b = __ballot(in);
        if (lane == 0) sc[wave] = __popcll(b);
        __syncthreads();
        int base = s_base, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) base += sc[w]; tot += sc[w]; }
        const int o = base + __popcll(b & ((1ull << lane) - 1ull));
        if (in && o < cap) {
            idx[off + o] = i;
            if (feat_out) {
                feat_out[(size_t)(off + o) * 3] = feat[(size_t)i * 3]; feat_out[(size_t)(off + o) * 3 + 1] = feat[(size_t)i * 3 + 1];
                feat_out[(size_t)(off + o) * 3 + 2] = feat[(size_t)i * 3 + 2];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && n_out) n_out[s] = s_base;
}

// ---- getLocalPoints.m:8-35 as a function of its own (the descriptor kernel has it folded in) ------------------------------
// mode 0: double arithmetic.  mode 1 / 2: MATLAB's single arithmetic when the keypoint (1) or only the cloud (2) is single
// (the values arrive exactly widened): the open box test against single limits (mode 1: c + [-R, R] in single; mode 2: formed in
// double, rounded for the comparison), pts_cube - c, vecnorm and dists < R in single, one rounding per operation.
struct LocalArgs { double cx, cy, cz, R; int mode; };
__device__ __forceinline__ void local_test(const double* __restrict__ pts, int i, int ld, const LocalArgs& a, bool& cube, bool& in, double (&rel)[3],
                                           double& dist) {
    const double x = pts[i], y = pts[i + (size_t)ld], z = pts[i + 2 * (size_t)ld];
    if (a.mode == 0) {
        cube = x > a.cx - a.R && x < a.cx + a.R && y > a.cy - a.R && y < a.cy + a.R && z > a.cz - a.R && z < a.cz + a.R;       // :8-13, open box
        rel[0] = x - a.cx; rel[1] = y - a.cy; rel[2] = z - a.cz;                                                                // :23
        dist = sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2]);                                                     // :24
        in = cube && dist < a.R;                                                                                                // :25
    } else {
        const float xf = (float)x, yf = (float)y, zf = (float)z, R32 = (float)a.R;
        const float c0 = (float)a.cx, c1 = (float)a.cy, c2 = (float)a.cz;
        float lo[3], hi[3];
        if (a.mode == 1) { lo[0] = c0 + (-R32); hi[0] = c0 + R32; lo[1] = c1 + (-R32); hi[1] = c1 + R32; lo[2] = c2 + (-R32); hi[2] = c2 + R32; }
        else { lo[0] = (float)(a.cx - a.R); hi[0] = (float)(a.cx + a.R); lo[1] = (float)(a.cy - a.R); hi[1] = (float)(a.cy + a.R);
               lo[2] = (float)(a.cz - a.R); hi[2] = (float)(a.cz + a.R); }
        cube = xf > lo[0] && xf < hi[0] && yf > lo[1] && yf < hi[1] && zf > lo[2] && zf < hi[2];
        const float r0 = xf - c0, r1 = yf - c1, r2 = zf - c2;
        const float d = sqrtf((r0 * r0 + r1 * r1) + r2 * r2);
        rel[0] = r0; rel[1] = r1; rel[2] = r2; dist = d;
        in = cube && d < R32;
    }
}
// per-workgroup counts of the box and of the sphere (cnt [nb][2])
__global__ __launch_bounds__(256) void local_count_kernel(const double* __restrict__ pts, int N, int ld, LocalArgs a, int32_t* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool cube = false, in = false; double rel[3], d;
    if (i < N) local_test(pts, i, ld, a, cube, in, rel, d);
    __shared__ int sc[4][2];
    const unsigned long long bc = __ballot(cube), bi = __ballot(in);
    if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6][0] = __popcll(bc); sc[threadIdx.x >> 6][1] = __popcll(bi); }
    __syncthreads();
    if (threadIdx.x < 2) cnt[(size_t)blockIdx.x * 2 + threadIdx.x] = sc[0][threadIdx.x] + sc[1][threadIdx.x] + sc[2][threadIdx.x] + sc[3][threadIdx.x];
}
// one workgroup: exclusive scan of the sphere counts in place, totals[0] = box count, totals[1] = sphere count
__global__ __launch_bounds__(256) void local_scan_kernel(int32_t* __restrict__ cnt, int nb, int32_t* __restrict__ totals) {
    __shared__ int s[256];
    __shared__ int s_cube[4];
    int carry = 0, cube = 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? cnt[(size_t)i * 2 + 1] : 0;
        cube += i < nb ? cnt[(size_t)i * 2] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int t = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) cnt[(size_t)i * 2 + 1] = carry + s[threadIdx.x] - v;
        const int tot = s[255];
        __syncthreads();
        carry += tot;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cube += __shfl_xor(cube, o);
    if ((threadIdx.x & 63) == 0) s_cube[threadIdx.x >> 6] = cube;
    __syncthreads();
    if (threadIdx.x == 0) { totals[0] = s_cube[0] + s_cube[1] + s_cube[2] + s_cube[3]; totals[1] = carry; }
}
// pts_sphere (n x 3 column-major, leading dimension ldo) and dists in the order of the cloud
__global__ __launch_bounds__(256) void local_scatter_kernel(const double* __restrict__ pts, int N, int ld, LocalArgs a, const int32_t* __restrict__ cnt,
                                                            double* __restrict__ out, int ldo, double* __restrict__ dists) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool cube = false, in = false; double rel[3], d = 0.0;
    if (i < N) local_test(pts, i, ld, a, cube, in, rel, d);
    __shared__ int sc[4];
    const unsigned long long b = __ballot(in);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sc[wave] = __popcll(b);
    __syncthreads();
    int base = cnt[(size_t)blockIdx.x * 2 + 1];
    for (int w = 0; w < wave; ++w) base += sc[w];
    if (in) {
        const int o = base + __popcll(b & ((1ull << lane) - 1ull));
        out[o] = rel[0]; out[o + (size_t)ldo] = rel[1]; out[o + 2 * (size_t)ldo] = rel[2];
        if (dists) dists[o] = d;
    }
}

// dst[k][:] = src[idx[k]][:] for k < *n (row-major, D doubles per row); one workgroup per row
__global__ __launch_bounds__(256) void gather_rows_f64_kernel(const double* __restrict__ src, int D, const int32_t* __restrict__ idx,
                                                              const int32_t* __restrict__ n, int cap, double* __restrict__ dst) {
    const int cnt = min(*n, cap);
    for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
        const double* s = src + (size_t)idx[k] * D;
        double* d = dst + (size_t)k * D;
        for (int e = threadIdx.x; e < D; e += 256) d[e] = s[e];
    }
}

// ---- the batched sweep (completeExperimentFast.m:166-224 for all spheres at once) -----------------------------------
// trial = find(num_putative > putative_thresh) (:175), in sphere order; offsets = where trial t's pairs start in the
// packed correspondence arrays.  One thread: S is a few hundred.  Entries past n_trials are empty registrations
// (offsets[t] == offsets[t+1]), so the batched ransac can be launched on the capacity S without a host round trip.
__global__ void sweep_plan_kernel(const int32_t* __restrict__ n_pairs, int S, int thresh, int32_t* __restrict__ trial_idx,
                                  int32_t* __restrict__ offsets /*S + 1*/, int32_t* __restrict__ n_trials) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int t = 0, off = 0;
    for (int i = 0; i < S; ++i) {
        if (n_pairs[i] > thresh) { trial_idx[t] = i; offsets[t] = off; off += n_pairs[i]; ++t; }
    }
    for (int k = t; k <= S; ++k) offsets[k] = off;
    for (int k = t; k < S; ++k) trial_idx[k] = -1;
    *n_trials = t;
}

// pts1 = featSurface(matches(:,1), :), pts2 = featuresM(matches(:,2), :) (:205-206) for every trial sphere, packed at
// offsets[t]: SoA columns with leading dimension ld.  pairs_all [S][VS][2] 1-based; featCur_all holds the spheres'
// gathered keypoints back to back, sphere i's rows starting at row_off[i].
__global__ __launch_bounds__(256) void sweep_gather_kernel(const uint32_t* __restrict__ pairs_all, int VS, const int32_t* __restrict__ n_pairs,
                                                           const int32_t* __restrict__ trial_idx, const int32_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ n_trials, const double* __restrict__ featS,
                                                           const double* __restrict__ featCur_all, const int64_t* __restrict__ row_off,
                                                           double* __restrict__ p1, double* __restrict__ p2, int ld) {
    const int t = blockIdx.y;
    if (t >= *n_trials) return;
    const int i = trial_idx[t], n = n_pairs[i], off = offsets[t];
    const uint32_t* pr = pairs_all + (size_t)i * VS * 2;
    const double* fc = featCur_all + (size_t)row_off[i] * 3;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const size_t a = (size_t)pr[2 * k] - 1, b = (size_t)pr[2 * k + 1] - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p1[(size_t)c * ld + off + k] = featS[a * 3 + c];
            p2[(size_t)c * ld + off + k] = fc[b * 3 + c];
        }
    }
}

struct TF16 { double t[16]; };
// pts_tf = [pts, 1] * TF (column-major 4x4), first three columns; pts n x 3 column-major
__global__ void quick_tf_kernel(const double* __restrict__ pts, int n, int ld, TF16 T, double* __restrict__ out, int ldo) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = pts[i], y = pts[i + (size_t)ld], z = pts[i + 2 * (size_t)ld];
#pragma unroll
        for (int j = 0; j < 3; ++j)      // the dot product of a row with column j, left to right as MATLAB's mtimes reference loop
            out[i + (size_t)j * ldo] = ((x * T.t[4 * j] + y * T.t[4 * j + 1]) + z * T.t[4 * j + 2]) + T.t[4 * j + 3];
    }
}

}  // namespace

int launch_sphere_counts(const double* feat, int V, const double* centres, int S, double R, int32_t* counts, hipStream_t st) {
    if (S <= 0) return PCREG_OK;
    hipLaunchKernelGGL(sphere_counts_kernel, dim3(S), dim3(1024), 0, st, feat, V, centres, S, sphere_sqrt_threshold(R), counts);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

size_t sphere_select_workspace_bytes(int V) {
    size_t v = (size_t)(V > 0 ? V : 1);
    return align_up(v * 4, 256) + align_up((v / 256 + 2) * 4, 256);
}
int launch_sphere_select(const double* feat, int V, const double c[3], double R, int32_t* idx, int32_t* n_out, void* ws, size_t ws_bytes,
                         hipStream_t st) {
    if (V <= 0) { PCREG_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), st)); return PCREG_OK; }
    if (ws_bytes < sphere_select_workspace_bytes(V)) { set_error("sphere_select workspace too small"); return PCREG_E_WORKSPACE; }
    int32_t* flag = (int32_t*)ws;
    int32_t* bc = (int32_t*)((char*)ws + align_up((size_t)V * 4, 256));
    const int nb = (V + 255) / 256;
    hipLaunchKernelGGL(sphere_flag_kernel, dim3(nb), dim3(256), 0, st, feat, V, c[0], c[1], c[2], sphere_sqrt_threshold(R), flag, bc);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, st, bc, nb, n_out);
    hipLaunchKernelGGL(sphere_scatter_kernel, dim3(nb), dim3(256), 0, st, flag, V, bc, idx);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_sphere_select_batched(const double* feat, int V, const double* centres, int S, double R, const int32_t* seg_off, int32_t* idx,
                                 double* feat_out, int32_t* n_out, hipStream_t st) {
    if (S <= 0) return PCREG_OK;
    hipLaunchKernelGGL(sphere_select_batched_kernel, dim3(S), dim3(1024), 0, st, feat, V, centres, sphere_sqrt_threshold(R), seg_off, idx, feat_out, n_out);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
size_t local_points_workspace_bytes(int N) { return align_up(((size_t)(N + 255) / 256 + 1) * 2 * sizeof(int32_t), 256) + 256; }
// counts -> totals[0] = points in the box, totals[1] = points in the sphere; pts_sphere / dists hold totals[1] rows
int launch_local_points(const double* pts, int N, int ld, double R, const double c[3], int mode, double* out, int ldo, double* dists,
                        int32_t* totals, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < local_points_workspace_bytes(N)) { set_error("getLocalPoints workspace too small"); return PCREG_E_WORKSPACE; }
    if (N <= 0) { PCREG_HIP(hipMemsetAsync(totals, 0, 2 * sizeof(int32_t), st)); return PCREG_OK; }
    const int nb = (N + 255) / 256;
    int32_t* cnt = (int32_t*)ws;
    const LocalArgs a{c[0], c[1], c[2], R, mode};
    hipLaunchKernelGGL(local_count_kernel, dim3(nb), dim3(256), 0, st, pts, N, ld, a, cnt);
    hipLaunchKernelGGL(local_scan_kernel, dim3(1), dim3(256), 0, st, cnt, nb, totals);
    hipLaunchKernelGGL(local_scatter_kernel, dim3(nb), dim3(256), 0, st, pts, N, ld, a, (const int32_t*)cnt, out, ldo, dists);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
int launch_gather_rows_f64(const double* src, int D, const int32_t* idx, const int32_t* n, int cap, double* dst, hipStream_t st) {
    if (cap <= 0 || D <= 0) return PCREG_OK;
    hipLaunchKernelGGL(gather_rows_f64_kernel, dim3(std::min(cap, 8192)), dim3(256), 0, st, src, D, idx, n, cap, dst);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_sweep_plan(const int32_t* n_pairs, int S, int thresh, int32_t* trial_idx, int32_t* offsets, int32_t* n_trials, hipStream_t st) {
    hipLaunchKernelGGL(sweep_plan_kernel, dim3(1), dim3(64), 0, st, n_pairs, S, thresh, trial_idx, offsets, n_trials);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
int launch_sweep_gather(const uint32_t* pairs_all, int VS, const int32_t* n_pairs, const int32_t* trial_idx, const int32_t* offsets,
                        const int32_t* n_trials, int S, const double* featS, const double* featCur_all, const int64_t* row_off,
                        double* p1, double* p2, int ld, hipStream_t st) {
    if (S <= 0 || VS <= 0) return PCREG_OK;
    hipLaunchKernelGGL(sweep_gather_kernel, dim3((VS + 255) / 256, S), dim3(256), 0, st, pairs_all, VS, n_pairs, trial_idx, offsets, n_trials,
                       featS, featCur_all, row_off, p1, p2, ld);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_quick_tf(const double* pts, int n, int ld, const double T[16], double* out, int ldo, hipStream_t st) {
    if (n <= 0) return PCREG_OK;
    TF16 t; for (int k = 0; k < 16; ++k) t.t[k] = T[k];
    hipLaunchKernelGGL(quick_tf_kernel, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, st, pts, n, ld, t, out, ldo);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
