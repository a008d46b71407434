// pcreg_amd/csrc/knn_points.hip -- all-pairs 3-D correspondence search on gfx950, fp32.
//
// This is the "KNN" half of BASELINE.json's metric: for every surface (query) point the
// two nearest model points, then matchFeatures' filter chain (threshold, ratio,
// Unique) -- the role getMatches.m:51-56 plays in completeExperimentFast.m:134-149,
// applied to raw 3-D points as in SURVEY.md section 8d (cfg 2/3).
//
// Kernel shape (CDNA4, wave64):
//   * every lane owns QPT queries in registers together with their running top-2
//     (distance + index), so the hot loop has no cross-lane traffic at all;
//   * the model shard is split in S chunks (grid.y) so that >= ~2k workgroups fill the
//     256 CUs; a chunk streams through LDS in tiles of float4 {x,y,z,-}; all lanes read
//     the same LDS address (broadcast ds_read_b128), 4 model points per batch;
//   * per batch and query: 6 VALU per pair (3 sub, 1 mul, 2 fma -- bit-identical to the
//     oracle's fmaf chain), one v_min3/v_min tree and ONE compare against the lane's
//     current 2nd-best; the index bookkeeping lives behind that rarely-taken branch;
//   * per-chunk partial top-2 lists are merged by a second kernel ordering (dist, idx),
//     the same kernel that merges the all-gathered per-GPU lists in the sharded setup.
// Ties resolve to the lowest model index (MATLAB min / partial-sort behaviour).
#include "common.hpp"
#include "select.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace pcreg {
namespace {

constexpr int kBlock = 256;
constexpr int QPT = 4;                       // queries per lane
constexpr int kQTile = kBlock * QPT;         // queries per workgroup
constexpr int kMTile = 1024;                 // model points per LDS tile (16 KiB)
constexpr int UB = 4;                        // model points per batch

struct Top2 { float d1, d2; int i1, i2; };

__device__ __forceinline__ void top2_insert(Top2& t, float d, int j) {
    // candidates arrive in ascending j inside a chunk, so strict '<' keeps the lowest index
    if (d < t.d2) {
        if (d < t.d1) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = j; }
        else { t.d2 = d; t.i2 = j; }
    }
}
__device__ __forceinline__ float sqd(float qx, float qy, float qz, const float4& m) {
    float dx = qx - m.x, dy = qy - m.y, dz = qz - m.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// grid = (query tiles, S model chunks).  part_* layout [S][Q][2].
template <int QPT_, int UB_>
__global__ __launch_bounds__(kBlock) void knn2_points_kernel(
    const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm,
    int chunk, int idx_base, int32_t* __restrict__ part_idx, float* __restrict__ part_dist,
    const int32_t* __restrict__ qlist, const int32_t* __restrict__ n_list, int min_active) {
    __shared__ float4 tile[kMTile];
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * (kBlock * QPT_);
    const int s = blockIdx.y;
    // optional query list (exact re-run of the queries the fast path could not certify):
    // slot k of this launch is query qlist[k]; partials stay indexed by slot
    int Qe = Q;
    if (qlist) { Qe = *n_list; if (Qe <= min_active || q0 >= Qe) return; }
    const int m_begin = s * chunk;
    const int m_end = min(M, m_begin + chunk);

    float qx[QPT_], qy[QPT_], qz[QPT_];
    Top2 best[QPT_];
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int slot = q0 + r * kBlock + tid;
        bool ok = slot < Qe;
        int qi = ok ? (qlist ? qlist[slot] : slot) : 0;
        qx[r] = ok ? q[qi] : 0.0f;
        qy[r] = ok ? q[qi + (size_t)ldq] : 0.0f;
        qz[r] = ok ? q[qi + 2 * (size_t)ldq] : 0.0f;
        best[r] = Top2{INFINITY, INFINITY, -1, -1};
    }

    for (int t0 = m_begin; t0 < m_end; t0 += kMTile) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMTile / kBlock; ++k) {
            int j = t0 + k * kBlock + tid;
            float4 v;
            if (j < m_end) { v.x = m[j]; v.y = m[j + (size_t)ldm]; v.z = m[j + 2 * (size_t)ldm]; v.w = 0.0f; }
            else { v.x = v.y = v.z = INFINITY; v.w = 0.0f; }     // padding never beats anything
            tile[k * kBlock + tid] = v;
        }
        __syncthreads();
        const int cnt = min(kMTile, m_end - t0);
        const int nb = (cnt + UB_ - 1) / UB_ * UB_;
        for (int jb = 0; jb < nb; jb += UB_) {
            float4 mp[UB_];
#pragma unroll
            for (int u = 0; u < UB_; ++u) mp[u] = tile[jb + u];
#pragma unroll
            for (int r = 0; r < QPT_; ++r) {
                float d[UB_];
#pragma unroll
                for (int u = 0; u < UB_; ++u) d[u] = sqd(qx[r], qy[r], qz[r], mp[u]);
                float mn = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
                if (UB_ == 8) mn = fminf(mn, fminf(fminf(d[4 % UB_], d[5 % UB_]), fminf(d[6 % UB_], d[7 % UB_])));
                if (mn < best[r].d2) {
                    const int j0 = idx_base + t0 + jb;
#pragma unroll
                    for (int u = 0; u < UB_; ++u) top2_insert(best[r], d[u], j0 + u);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int slot = q0 + r * kBlock + tid;
        if (slot < Qe) {
            size_t o = ((size_t)s * Q + slot) * 2;
            part_idx[o] = best[r].i1; part_idx[o + 1] = best[r].i2;
            part_dist[o] = best[r].d1; part_dist[o + 1] = best[r].d2;
        }
    }
}

// merge of the per-chunk partials of a query-list launch, scattered to the listed queries
__global__ void merge_top2_list_kernel(const int32_t* __restrict__ part_idx, const float* __restrict__ part_dist, int S,
                                       int Qcap, const int32_t* __restrict__ qlist, const int32_t* __restrict__ n_list,
                                       int min_active, int32_t* __restrict__ idx, float* __restrict__ dist) {
    const int n = *n_list;
    if (n <= min_active) return;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        Top2T<float> t{INFINITY, INFINITY, -1, -1};
        for (int r = 0; r < S; ++r) {
            size_t o = ((size_t)r * Qcap + slot) * 2;
            top2_insert_lex_t(t, part_dist[o], part_idx[o]);
            top2_insert_lex_t(t, part_dist[o + 1], part_idx[o + 1]);
        }
        int qi = qlist[slot];
        idx[(size_t)qi * 2] = t.i1; idx[(size_t)qi * 2 + 1] = t.i2;
        dist[(size_t)qi * 2] = t.d1; dist[(size_t)qi * 2 + 1] = t.d2;
    }
}

// Unique back-check: for candidate k (model row j in this shard) find the first-best
// query over all Q.  Same hot loop as the forward search with the roles swapped:
// "queries" are the matched model points (gathered on the fly), the surface streams
// through LDS.  Only the best index is needed.
__global__ __launch_bounds__(kBlock) void unique_points_kernel(
    const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm, int m_lo,
    const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m, const int32_t* __restrict__ n_cand,
    int chunk, int32_t* __restrict__ part_idx, float* __restrict__ part_dist) {
    __shared__ float4 tile[kMTile];
    const int P = *n_cand;
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * kQTile;
    if (k0 >= P) return;
    const int s = blockIdx.y;
    const int b_begin = s * chunk, b_end = min(Q, b_begin + chunk);
    float px[QPT], py[QPT], pz[QPT], bd[QPT]; int bi[QPT];
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        int k = k0 + r * kBlock + tid;
        int j = k < P ? cand_m[k] - m_lo : -1;
        bool ok = j >= 0 && j < M;
        px[r] = ok ? m[j] : 0.0f; py[r] = ok ? m[j + (size_t)ldm] : 0.0f; pz[r] = ok ? m[j + 2 * (size_t)ldm] : 0.0f;
        bd[r] = INFINITY; bi[r] = -1;
    }
    for (int t0 = b_begin; t0 < b_end; t0 += kMTile) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMTile / kBlock; ++k) {
            int i = t0 + k * kBlock + tid;
            float4 v;
            if (i < b_end) { v.x = q[i]; v.y = q[i + (size_t)ldq]; v.z = q[i + 2 * (size_t)ldq]; v.w = 0.0f; }
            else { v.x = v.y = v.z = INFINITY; v.w = 0.0f; }
            tile[k * kBlock + tid] = v;
        }
        __syncthreads();
        const int cnt = min(kMTile, b_end - t0);
        const int nb = (cnt + UB - 1) / UB * UB;
        for (int jb = 0; jb < nb; jb += UB) {
            float4 mp[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) mp[u] = tile[jb + u];
#pragma unroll
            for (int r = 0; r < QPT; ++r) {
                float d[UB];
                // the score of (query i, model j) must be the bits the forward pass saw:
                // dx = q - m, i.e. (surface - model)
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    float dx = mp[u].x - px[r], dy = mp[u].y - py[r], dz = mp[u].z - pz[r];
                    d[u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                }
                float mn = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
                if (mn < bd[r]) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) if (d[u] < bd[r]) { bd[r] = d[u]; bi[r] = t0 + jb + u; }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        int k = k0 + r * kBlock + tid;
        if (k < P) { size_t o = (size_t)s * Q + k; part_idx[o] = bi[r]; part_dist[o] = bd[r]; }
    }
}
__global__ void unique_reduce_kernel(const int32_t* __restrict__ part_idx, const float* __restrict__ part_dist, int S,
                                     int Q, int M, int m_lo, const int32_t* __restrict__ cand_q,
                                     const int32_t* __restrict__ cand_m, const int32_t* __restrict__ n_cand,
                                     int32_t* __restrict__ keep) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_cand) return;
    int j = cand_m[k] - m_lo;
    if (j < 0 || j >= M) return;                 // another shard's row
    float bd = INFINITY; int bi = -1;
    for (int s = 0; s < S; ++s) {                // chunks ascend in query index: strict '<' keeps the first
        float d = part_dist[(size_t)s * Q + k]; int i = part_idx[(size_t)s * Q + k];
        if (i >= 0 && d < bd) { bd = d; bi = i; }
    }
    keep[k] = (bi == cand_q[k]);
}

// Unique through the certified search (default): the matched model points become the queries of one
// more top-2 search over the surface; keep[k] = (nearest surface point of model row cand_m[k]) == cand_q[k].
// d(p, s) uses dx = p - s here and dx = s - p in the forward pass: the same bits (squares of negated values).
__global__ void unique_gather_kernel(const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm, int m_lo,
                                     const int32_t* __restrict__ cand_m, const int32_t* __restrict__ n_cand, float* __restrict__ pts /*[3][Q]*/) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Q) return;
    const int j = k < *n_cand ? cand_m[k] - m_lo : -1;
    const bool ok = j >= 0 && j < M;                 // rows of other shards / unused slots: any valid point (ignored later)
#pragma unroll
    for (int c = 0; c < 3; ++c) pts[k + (size_t)c * Q] = ok ? m[j + (size_t)c * ldm] : q[(size_t)c * ldq];
}
__global__ void unique_keep_kernel(const int32_t* __restrict__ idx2, int M, int m_lo, const int32_t* __restrict__ cand_q,
                                   const int32_t* __restrict__ cand_m, const int32_t* __restrict__ n_cand, int32_t* __restrict__ keep) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *n_cand) return;
    const int j = cand_m[k] - m_lo;
    if (j < 0 || j >= M) return;                     // another shard's row
    keep[k] = (idx2[(size_t)k * 2] == cand_q[k]);
}

// ---------------------------------------------------------------- Unique back-check on a grid of the queries
// keep (i, j) iff no other query i' beats i for model point j in the (distance, index) order of the search.
// That is a range-emptiness question with the candidate's OWN distance as radius (small: j is i's nearest
// model point), so a uniform grid over the queries answers it exactly with a handful of distance evaluations
// instead of a second all-pairs search.  Cells hold up to kUgSlots points; a candidate that meets a fuller
// cell, or whose ball covers too many cells, is re-done by ug_brute_kernel against every query.
constexpr int kUgSlots = 16;
constexpr int kUgMaxCells = 4 << 20;
constexpr int kUgMaxVisit = 343;
struct UgPrep { float x0, y0, z0, inv_c; int nx, ny, nz, pad; };
static size_t ug_cells_cap(int Q) { size_t c = 2 * (size_t)Q; if (c < 4096) c = 4096; if (c > (size_t)kUgMaxCells) c = kUgMaxCells; return c; }

__device__ __forceinline__ float ug_d2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;           // the search's exact formula (sign-symmetric)
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ int ug_cell1(float x, float x0, float inv_c, int n) {
    const int c = (int)floorf((x - x0) * inv_c);                     // monotone in x: the range test relies on it
    return min(max(c, 0), n - 1);
}

__global__ __launch_bounds__(1024) void ug_bbox_kernel(const float* __restrict__ q, int Q, int ldq, int cells_cap,
                                                       UgPrep* __restrict__ prep, int32_t* __restrict__ n_flag, int32_t* __restrict__ cnt) {
    if (blockIdx.x > 0) {                     // workgroups 1.. clear the grid's counters (saves a memset launch); 0 finds the box
        for (int i = (blockIdx.x - 1) * 1024 + threadIdx.x; i < cells_cap; i += (gridDim.x - 1) * 1024) cnt[i] = 0;
        return;
    }
    __shared__ float s_lo[3][16], s_hi[3][16];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i0 = threadIdx.x; i0 < Q; i0 += 4 * 1024) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = q[min(i0 + u * 1024, Q - 1) + (size_t)c * ldq];      // clamped repeats are harmless
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], v[u][c]); hi[c] = fmaxf(hi[c], v[u][c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
        if ((threadIdx.x & 63) == 0) { s_lo[c][threadIdx.x >> 6] = lo[c]; s_hi[c][threadIdx.x >> 6] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float e[3];
        for (int c = 0; c < 3; ++c) {
            for (int w = 1; w < 16; ++w) { lo[c] = fminf(lo[c], s_lo[c][w]); hi[c] = fmaxf(hi[c], s_hi[c][w]); }
            e[c] = fmaxf(hi[c] - lo[c], 0.0f);
        }
        // about two cells per query over the occupied extent; flat or degenerate axes get one layer
        const float emax = fmaxf(e[0], fmaxf(e[1], e[2]));
        float vol = 1.0f; int dims = 0;
        for (int c = 0; c < 3; ++c) if (e[c] > 1e-6f * emax) { vol *= e[c]; ++dims; }
        float cs = dims > 0 ? powf(vol / (2.0f * (float)Q), 1.0f / (float)dims) : 1.0f;
        if (!(cs > 0.0f) || !isfinite(cs)) cs = 1.0f;
        int nx, ny, nz;
        for (;;) {
            const float inv = 1.0f / cs;
            const float fx = floorf(e[0] * inv) + 1.0f, fy = floorf(e[1] * inv) + 1.0f, fz = floorf(e[2] * inv) + 1.0f;
            if (fx * fy * fz <= (float)cells_cap) { nx = (int)fx; ny = (int)fy; nz = (int)fz; break; }
            cs *= 1.08f;
        }
        prep->x0 = lo[0]; prep->y0 = lo[1]; prep->z0 = lo[2]; prep->inv_c = 1.0f / cs;
        prep->nx = nx; prep->ny = ny; prep->nz = nz; prep->pad = 0;
        *n_flag = 0;
    }
}

__global__ __launch_bounds__(256) void ug_fill_kernel(const float* __restrict__ q, int Q, int ldq, const UgPrep* __restrict__ prep,
                                                      int32_t* __restrict__ cnt, float4* __restrict__ slots) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Q) return;
    const UgPrep P = *prep;
    const float x = q[i], y = q[i + (size_t)ldq], z = q[i + 2 * (size_t)ldq];
    const int cell = (ug_cell1(z, P.z0, P.inv_c, P.nz) * P.ny + ug_cell1(y, P.y0, P.inv_c, P.ny)) * P.nx + ug_cell1(x, P.x0, P.inv_c, P.nx);
    const int s = atomicAdd(&cnt[cell], 1);
    if (s < kUgSlots) slots[(size_t)cell * kUgSlots + s] = make_float4(x, y, z, __int_as_float(i));
}

__global__ __launch_bounds__(64) void ug_check_kernel(const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm,
                                                     int m_lo, const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m,
                                                     const int32_t* __restrict__ n_cand, const UgPrep* __restrict__ prep,
                                                     const int32_t* __restrict__ cnt, const float4* __restrict__ slots,
                                                     int32_t* __restrict__ keep, int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= *n_cand) return;
    const int j = cand_m[k] - m_lo;
    if (j < 0 || j >= M) return;                                     // another shard's row
    const int i = cand_q[k];
    const UgPrep P = *prep;
    const float px = m[j], py = m[j + (size_t)ldm], pz = m[j + 2 * (size_t)ldm];
    const float di = ug_d2(q[i], q[i + (size_t)ldq], q[i + 2 * (size_t)ldq], px, py, pz);
    const float r = sqrtf(di) * 1.0001f + 1e-30f;                    // covers every point whose rounded distance is <= di
    const int x0 = ug_cell1(px - r, P.x0, P.inv_c, P.nx), x1 = ug_cell1(px + r, P.x0, P.inv_c, P.nx);
    const int y0 = ug_cell1(py - r, P.y0, P.inv_c, P.ny), y1 = ug_cell1(py + r, P.y0, P.inv_c, P.ny);
    const int z0 = ug_cell1(pz - r, P.z0, P.inv_c, P.nz), z1 = ug_cell1(pz + r, P.z0, P.inv_c, P.nz);
    bool brute = !(di == di) || (long long)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > kUgMaxVisit;
    bool kp = true;
    for (int cz = z0; cz <= z1 && !brute && kp; ++cz)
        for (int cy = y0; cy <= y1 && !brute && kp; ++cy)
            for (int cx = x0; cx <= x1 && kp; ++cx) {
                const int cell = (cz * P.ny + cy) * P.nx + cx;
                const int c = cnt[cell];
                if (c > kUgSlots) { brute = true; break; }
                for (int s = 0; s < c; ++s) {
                    const float4 t = slots[(size_t)cell * kUgSlots + s];
                    const int it = __float_as_int(t.w);
                    const float d = ug_d2(t.x, t.y, t.z, px, py, pz);
                    if (d < di || (d == di && it < i)) { kp = false; break; }
                }
            }
    if (brute && kp) flag_list[atomicAdd(n_flag, 1)] = k;            // undecided: the exhaustive scan settles it
    keep[k] = kp && !brute;
}

// one workgroup per undecided candidate against every query (four loads in flight per thread: the scan is
// latency-bound)
__global__ __launch_bounds__(1024) void ug_brute_kernel(const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int ldm, int m_lo,
                                                       const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m,
                                                       const int32_t* __restrict__ flag_list, const int32_t* __restrict__ n_flag,
                                                       int32_t* __restrict__ keep) {
    const int nf = *n_flag;
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {
        const int k = flag_list[f];
        const int j = cand_m[k] - m_lo, i = cand_q[k];
        const float px = m[j], py = m[j + (size_t)ldm], pz = m[j + 2 * (size_t)ldm];
        const float di = ug_d2(q[i], q[i + (size_t)ldq], q[i + 2 * (size_t)ldq], px, py, pz);
        bool b = false;
        for (int t0 = threadIdx.x; t0 < Q && !b; t0 += 4 * 1024) {
            float x[4], y[4], z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(t0 + u * 1024, Q - 1);
                x[u] = q[t]; y[u] = q[t + (size_t)ldq]; z[u] = q[t + 2 * (size_t)ldq];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * 1024;
                const float d = ug_d2(x[u], y[u], z[u], px, py, pz);
                b = b || (t < Q && (d < di || (d == di && t < i)));
            }
        }
        const int beaten = __syncthreads_or(b);
        if (threadIdx.x == 0) keep[k] = !beaten;
    }
}

// Multi-GPU: this rank's contribution to the [4][Q] candidate table (4-byte words): rows 0-2 = coordinate bits of
// the candidates whose model row lives in this shard, row 3 = their Unique verdict; zero everywhere else, so an
// integer SUM over the ranks assembles the table exactly.
__global__ void cand_table_kernel(const float* __restrict__ m, int M, int ldm, int m_lo, const int32_t* __restrict__ cand_m,
                                  const int32_t* __restrict__ keep, const int32_t* __restrict__ n_cand, int Q,
                                  int32_t* __restrict__ table) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Q) return;
    int w[4] = {0, 0, 0, 0};
    if (k < *n_cand) {
        const int j = cand_m[k] - m_lo;
        if (j >= 0 && j < M) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = __float_as_int(m[j + (size_t)c * ldm]);
            w[3] = keep ? (keep[k] != 0) : 1;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) table[k + (size_t)c * Q] = w[c];
}

// Ordered compaction of the kept candidates (three small launches: per-workgroup counts,
// exclusive scan, scatter) into 1-based pairs and, optionally, the matched coordinates.
__global__ void gather_count_kernel(const int32_t* __restrict__ keep, const int32_t* __restrict__ n_cand,
                                    int32_t* __restrict__ block_cnt) {
    const int P = *n_cand;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    bool kp = k < P && (keep == nullptr || keep[k] != 0);
    __shared__ int s_cnt[4];
    unsigned long long b = __ballot(kp);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
__global__ void gather_scatter_kernel(const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int ldm,
                                      const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m,
                                      const int32_t* __restrict__ keep, const int32_t* __restrict__ n_cand,
                                      const int32_t* __restrict__ block_off, int self_prefix, int32_t* __restrict__ n_pairs,
                                      uint32_t* __restrict__ pairs, double* __restrict__ pts1, double* __restrict__ pts2) {
    const int P = *n_cand;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    bool kp = k < P && (keep == nullptr || keep[k] != 0);
    __shared__ int s_cnt[4];
    __shared__ int s_pre[4];
    const int pre = self_prefix ? block_self_prefix_256(block_off, blockIdx.x, gridDim.x, s_pre, n_pairs) : block_off[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long b = __ballot(kp);
    if (lane == 0) s_cnt[wave] = __popcll(b);
    __syncthreads();
    int base = pre;
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    if (kp) {
        int o = base + __popcll(b & ((1ull << lane) - 1ull));
        int qi = cand_q[k], mj = cand_m[k];
        if (pairs) { pairs[(size_t)o * 2] = (uint32_t)qi + 1u; pairs[(size_t)o * 2 + 1] = (uint32_t)mj + 1u; }
        if (pts1) {
            pts1[o] = (double)q[qi]; pts1[o + (size_t)Q] = (double)q[qi + (size_t)ldq]; pts1[o + 2 * (size_t)Q] = (double)q[qi + 2 * (size_t)ldq];
            pts2[o] = (double)m[mj]; pts2[o + (size_t)Q] = (double)m[mj + (size_t)ldm]; pts2[o + 2 * (size_t)Q] = (double)m[mj + 2 * (size_t)ldm];
        }
    }
}

// the same in ONE launch for capacities up to kCompactMax (select.hpp): one 1024-thread workgroup, contiguous runs
__global__ __launch_bounds__(kCompactThreads) void gather_compact_kernel(const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int ldm,
                                                                         const int32_t* __restrict__ cand_q, const int32_t* __restrict__ cand_m,
                                                                         const int32_t* __restrict__ keep, const int32_t* __restrict__ n_cand,
                                                                         uint32_t* __restrict__ pairs, double* __restrict__ pts1, double* __restrict__ pts2,
                                                                         int32_t* __restrict__ n_pairs) {
    __shared__ int s_wave[kCompactThreads / 64];
    const int P = min(*n_cand, Q);
    const int per = (P + kCompactThreads - 1) / kCompactThreads;
    const int lo = min(P, (int)threadIdx.x * per), hi = min(P, lo + per);
    int cnt = 0;
    for (int k = lo; k < hi; ++k) cnt += (keep == nullptr || keep[k] != 0);
    int total;
    int o = block_exclusive_scan_1024(cnt, s_wave, &total);
    for (int k = lo; k < hi; ++k) {
        if (keep != nullptr && keep[k] == 0) continue;
        const int qi = cand_q[k], mj = cand_m[k];
        if (pairs) { pairs[(size_t)o * 2] = (uint32_t)qi + 1u; pairs[(size_t)o * 2 + 1] = (uint32_t)mj + 1u; }
        if (pts1) {
            pts1[o] = (double)q[qi]; pts1[o + (size_t)Q] = (double)q[qi + (size_t)ldq]; pts1[o + 2 * (size_t)Q] = (double)q[qi + 2 * (size_t)ldq];
            pts2[o] = (double)m[mj]; pts2[o + (size_t)Q] = (double)m[mj + (size_t)ldm]; pts2[o + 2 * (size_t)Q] = (double)m[mj + 2 * (size_t)ldm];
        }
        ++o;
    }
    if (threadIdx.x == 0) *n_pairs = total;
}

// number of model chunks so that the grid has ~>= 8 workgroups per CU
int pick_splits(int n_tiles, int M, int target = 2048) {
    int S = (target + n_tiles - 1) / n_tiles;
    int maxS = (M + kMTile - 1) / kMTile;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return S;
}
int chunk_of(int M, int S) {
    int c = (M + S - 1) / S;
    return (c + kMTile - 1) / kMTile * kMTile;       // whole LDS tiles
}

}  // namespace

size_t knn2_points_exact_workspace_bytes(int Q, int M) {
    // sized for the most-split configuration any tuning variant can pick
    int n_tiles = (Q + kBlock * 8 - 1) / (kBlock * 8); if (n_tiles < 1) n_tiles = 1;
    int S = pick_splits(n_tiles, M > 0 ? M : 1, 8192);
    return 2 * align_up((size_t)S * (size_t)(Q > 0 ? Q : 1) * 2 * sizeof(float), 256);
}

size_t knn2_points_exact_workspace_bytes(int Q, int M);
size_t knn2_points_fast_workspace_bytes(int Q, int M);
int launch_knn2_points_fast_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                                int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st);

static int knn2_exact_impl(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                           const int32_t* qlist, const int32_t* n_list, int min_active,
                           int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st) {
    PCREG_ARG(Q >= 0 && M >= 0 && ldq >= Q && ldm >= M);
    if (Q == 0) return PCREG_OK;
    size_t need = knn2_points_exact_workspace_bytes(Q, M);
    if (ws_bytes < need) { set_error("knn workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    const int variant = PCREG_EXP_ENV("PCREG_KNN_POINTS_VARIANT", 0);
    const int target = PCREG_EXP_ENV("PCREG_KNN_BLOCKS", 2048);
    const int qpt = (variant == 2) ? 8 : (variant == 3 ? 2 : 4);
    int n_tiles = (Q + kBlock * qpt - 1) / (kBlock * qpt);
    int S = pick_splits(n_tiles, M > 0 ? M : 1, target);
    int chunk = chunk_of(M > 0 ? M : 1, S);
    S = M > 0 ? (M + chunk - 1) / chunk : 1;
    int32_t* part_idx = (int32_t*)ws;
    float* part_dist = (float*)((char*)ws + align_up((size_t)S * Q * 2 * sizeof(float), 256));
    dim3 grid(n_tiles, S);
#define PCREG_KNN_LAUNCH(QP, UBV) hipLaunchKernelGGL((knn2_points_kernel<QP, UBV>), grid, dim3(kBlock), 0, st, q, Q, ldq, m, M, ldm, chunk, (int)idx_base, part_idx, part_dist, qlist, n_list, min_active)
    switch (variant) {
        case 1: PCREG_KNN_LAUNCH(4, 8); break;
        case 2: PCREG_KNN_LAUNCH(8, 4); break;
        case 3: PCREG_KNN_LAUNCH(2, 8); break;
        default: PCREG_KNN_LAUNCH(4, 4); break;
    }
#undef PCREG_KNN_LAUNCH
    PCREG_HIP(hipGetLastError());
    if (qlist)
        hipLaunchKernelGGL(merge_top2_list_kernel, dim3(64), dim3(256), 0, st, part_idx, part_dist, S, Q, qlist, n_list, min_active, idx, dist);
    else
        hipLaunchKernelGGL(merge_top2_kernel_t<float>, dim3((Q + 255) / 256), dim3(256), 0, st, part_idx, part_dist, S, Q, idx, dist, (size_t)0);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_knn2_points_exact_list(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                                  const int32_t* qlist, const int32_t* n_list, int min_active, int32_t* idx,
                                  float* dist, void* ws, size_t ws_bytes, hipStream_t st) {
    return knn2_exact_impl(q, Q, ldq, m, M, ldm, idx_base, qlist, n_list, min_active, idx, dist, ws, ws_bytes, st);
}

// PCREG_KNN_EXACT=1 forces the direct-form kernel (tuning / A-B runs); the default is the
// certified fast path of knn_fast.hip, which returns the same bits.
static bool use_exact_only() { return pcreg_env_int("PCREG_KNN_EXACT", 0) != 0; }

size_t knn2_points_workspace_bytes(int Q, int M) {
    size_t a = knn2_points_exact_workspace_bytes(Q, M), b = knn2_points_fast_workspace_bytes(Q, M);
    return a > b ? a : b;
}

int launch_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                           int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st) {
    if (use_exact_only())
        return knn2_exact_impl(q, Q, ldq, m, M, ldm, idx_base, nullptr, nullptr, 0, idx, dist, ws, ws_bytes, st);
    return launch_knn2_points_fast_f32(q, Q, ldq, m, M, ldm, idx_base, idx, dist, ws, ws_bytes, st);
}

int launch_merge_top2_f32(const int32_t* idx_in, const float* dist_in, int R, int Q, int32_t* idx, float* dist,
                          hipStream_t st, size_t rank_stride) {
    PCREG_ARG(R >= 1 && Q >= 0 && (rank_stride == 0 || rank_stride >= (size_t)Q * 2));
    if (Q == 0) return PCREG_OK;
    hipLaunchKernelGGL(merge_top2_kernel_t<float>, dim3((Q + 255) / 256), dim3(256), 0, st, idx_in, dist_in, R, Q, idx, dist, rank_stride);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_filter_top2_f32(const int32_t* idx, const float* dist, int Q, int M_total, float thr, float ratio,
                           int32_t* cand_q, int32_t* cand_m, int32_t* n_cand, hipStream_t st) {
    PCREG_ARG(Q >= 0);
    void* tmp = nullptr;   // flags [Q] + per-workgroup counters
    int rc = stream_scratch(st).get(20, ((size_t)Q + (Q + 255) / 256 + 1) * sizeof(int32_t), &tmp);
    if (rc) return rc;
    return run_filter_top2<float>(idx, dist, Q, M_total, thr, ratio, cand_q, cand_m, n_cand, (int32_t*)tmp, st);
}

static size_t unique_direct_bytes(int Q) {
    int n_tiles = (Q + kQTile - 1) / kQTile; if (n_tiles < 1) n_tiles = 1;
    int S = pick_splits(n_tiles, Q > 0 ? Q : 1);
    return 2 * align_up((size_t)S * (size_t)(Q > 0 ? Q : 1) * sizeof(float), 256);
}
// gathered points [3][Q] | idx2 [Q][2] | dist2 [Q][2] | search workspace
static size_t unique_grid_bytes(int Q) {   // prep | n_flag | cell counts | cell slots | undecided list
    return 256 + 256 + align_up(ug_cells_cap(Q) * 4, 256) + align_up(ug_cells_cap(Q) * kUgSlots * 16, 256) +
           align_up((size_t)(Q > 0 ? Q : 1) * 4, 256);
}
size_t unique_points_workspace_bytes(int Q) {
    size_t q = (size_t)(Q > 0 ? Q : 1);
    size_t fast = align_up(q * 3 * 4, 256) + 2 * align_up(q * 2 * 4, 256) + knn2_points_workspace_bytes(Q, Q);
    size_t direct = unique_direct_bytes(Q);
    size_t grid = unique_grid_bytes(Q);
    return std::max(std::max(fast, direct), grid);
}

int launch_unique_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t m_lo,
                             const int32_t* cand_q, const int32_t* cand_m, const int32_t* n_cand, int32_t* keep,
                             void* ws, size_t ws_bytes, hipStream_t st) {
    PCREG_ARG(Q >= 0 && M >= 0);
    if (Q == 0) return PCREG_OK;
    size_t need = unique_points_workspace_bytes(Q);
    if (ws_bytes < need) { set_error("unique workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    const int unique_mode = pcreg_env_int("PCREG_UNIQUE_MODE", 0);   // 1: second search
    if (!use_exact_only() && Q >= 4096 && unique_mode == 0) {
        // grid of the queries + exact range-emptiness test per candidate (see ug_* above)
        char* w = (char*)ws;
        UgPrep* prep = (UgPrep*)w;          w += 256;
        int32_t* n_flag = (int32_t*)w;      w += 256;
        const size_t cells = ug_cells_cap(Q);
        int32_t* cnt = (int32_t*)w;         w += align_up(cells * 4, 256);
        float4* slots = (float4*)w;         w += align_up(cells * kUgSlots * 16, 256);
        int32_t* flag_list = (int32_t*)w;
        hipLaunchKernelGGL(ug_bbox_kernel, dim3(1 + (unsigned)std::min<size_t>(64, (cells + 4095) / 4096)), dim3(1024), 0, st, q, Q, ldq, (int)cells, prep, n_flag, cnt);
        hipLaunchKernelGGL(ug_fill_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, q, Q, ldq, prep, cnt, slots);
        hipLaunchKernelGGL(ug_check_kernel, dim3((Q + 63) / 64), dim3(64), 0, st, q, Q, ldq, m, M, ldm, (int)m_lo, cand_q, cand_m, n_cand,
                           prep, cnt, slots, keep, flag_list, n_flag);
        hipLaunchKernelGGL(ug_brute_kernel, dim3(512), dim3(1024), 0, st, q, Q, ldq, m, ldm, (int)m_lo, cand_q, cand_m, flag_list, n_flag, keep);
        PCREG_HIP(hipGetLastError());
        return PCREG_OK;
    }
    if (!use_exact_only() && Q >= 4096) {
        size_t qq = (size_t)Q;
        char* w = (char*)ws;
        float* pts = (float*)w;             w += align_up(qq * 3 * 4, 256);
        int32_t* idx2 = (int32_t*)w;        w += align_up(qq * 2 * 4, 256);
        float* dist2 = (float*)w;           w += align_up(qq * 2 * 4, 256);
        hipLaunchKernelGGL(unique_gather_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, q, Q, ldq, m, M, ldm, (int)m_lo, cand_m, n_cand, pts);
        int rc = launch_knn2_points_f32(pts, Q, Q, q, Q, ldq, 0, idx2, dist2, w, ws_bytes - (size_t)(w - (char*)ws), st);
        if (rc) return rc;
        hipLaunchKernelGGL(unique_keep_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, idx2, M, (int)m_lo, cand_q, cand_m, n_cand, keep);
        PCREG_HIP(hipGetLastError());
        return PCREG_OK;
    }
    int n_tiles = (Q + kQTile - 1) / kQTile;         // capacity: every query matched
    int S = pick_splits(n_tiles, Q);
    int chunk = chunk_of(Q, S);
    S = (Q + chunk - 1) / chunk;
    int32_t* part_idx = (int32_t*)ws;
    float* part_dist = (float*)((char*)ws + align_up((size_t)S * Q * sizeof(float), 256));
    hipLaunchKernelGGL(unique_points_kernel, dim3(n_tiles, S), dim3(kBlock), 0, st, q, Q, ldq, m, M, ldm, (int)m_lo,
                       cand_q, cand_m, n_cand, chunk, part_idx, part_dist);
    hipLaunchKernelGGL(unique_reduce_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, part_idx, part_dist, S, Q, M,
                       (int)m_lo, cand_q, cand_m, n_cand, keep);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_cand_table_f32(const float* m, int M, int ldm, int32_t m_lo, const int32_t* cand_m, const int32_t* keep,
                          const int32_t* n_cand, int Q, int32_t* table, hipStream_t st) {
    PCREG_ARG(Q >= 0 && M >= 0);
    if (Q == 0) return PCREG_OK;
    hipLaunchKernelGGL(cand_table_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, m, M, ldm, (int)m_lo, cand_m, keep, n_cand, Q, table);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_gather_pairs_f32(const float* q, int Q, int ldq, const float* m, int ldm, const int32_t* cand_q,
                            const int32_t* cand_m, const int32_t* keep, const int32_t* n_cand, uint32_t* pairs,
                            double* pts1, double* pts2, int32_t* n_pairs, hipStream_t st) {
    PCREG_ARG((pts1 == nullptr) == (pts2 == nullptr));
    if (Q <= 0) { PCREG_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), st)); return PCREG_OK; }
    if (Q <= kCompactMax) {
        hipLaunchKernelGGL(gather_compact_kernel, dim3(1), dim3(kCompactThreads), 0, st, q, Q, ldq, m, ldm, cand_q, cand_m, keep, n_cand, pairs, pts1, pts2, n_pairs);
        PCREG_HIP(hipGetLastError());
        return PCREG_OK;
    }
    const int nb = (Q + 255) / 256;                   // capacity launch: the kernels read the real count
    void* tmp = nullptr;
    int rc = stream_scratch(st).get(21, ((size_t)nb + 1) * sizeof(int32_t), &tmp);
    if (rc) return rc;
    int32_t* bc = (int32_t*)tmp;
    hipLaunchKernelGGL(gather_count_kernel, dim3(nb), dim3(256), 0, st, keep, n_cand, bc);
    const int self_prefix = nb <= kSelfPrefixMax;
    if (!self_prefix) hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, st, bc, nb, n_pairs);
    hipLaunchKernelGGL(gather_scatter_kernel, dim3(nb), dim3(256), 0, st, q, Q, ldq, m, ldm, cand_q, cand_m, keep, n_cand,
                       bc, self_prefix, n_pairs, pairs, pts1, pts2);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
