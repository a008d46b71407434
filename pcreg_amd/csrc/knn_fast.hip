// pcreg_amd/csrc/knn_fast.hip -- certified fast path of the fp32 3-D point search.
//
// Same contract as knn2_points_kernel (knn_points.hip): for every query the two nearest
// model points under d = fmaf(dz,dz, fmaf(dy,dy, dx*dx)), dx = q - m, ties to the lowest
// index -- the bits the oracle produces.  The direct form costs 6 VALU per pair; this path
// gets the same answer from ~3.6 VALU per pair:
//
//  1. prep      model -> float4 {x-c, y-c, z-c, |m-c|^2} about the bounding-box centre c
//               (one streaming pass, 12 B in / 16 B out per point) and the radii R_m.
//  2. candidates  s(q,m) = |m~|^2 - 2 q~.m~  (= |q~-m~|^2 - |q~|^2): THREE fma per pair.
//               Every lane owns QPT queries with a sorted top-4 of s in registers; model
//               tiles stream through LDS (16-B loads, broadcast ds_read_b128); one
//               v_min tree + ONE compare per batch of 8 pairs guards the insertion code.
//               The model is split in chunks over grid.y; all chunks of a query share one
//               monotone threshold word in HBM (relaxed atomicMin of the ordered-uint image
//               of the lane's 4th-best): it only prunes, results never depend on who
//               published what first.
//  3. finalize  one wave per query: exact fmaf-chain distances for the few union candidates
//               that can still be in the top-2, wave-shuffle (dist,idx) top-2 reduction,
//               and the CERTIFICATE: every point not in a candidate list has s >= G (the
//               final threshold word), hence exact d >= G + |q~|^2 - E with the rounding
//               bound E below; if that is > the exact 2nd-best the answer is proven.
//  4. fallback  queries that fail the certificate (~0.2 % on the benchmark cloud, all of
//               them when coordinates are so large that E swamps the spacing) are redone
//               exactly: one workgroup per query when few, the tiled exact kernel when many.
//
// Rounding bound (u = 2^-24, R_m = max|m~|, r = |q~|), derived in DESIGN.md section 5:
//   E = u*(3 R_m^2 + 3.03 (R_m^2 + 2 r R_m) + 4.04 (r + R_m)^2) + 16 u (d2 + |G + r^2|)
#include "common.hpp"
#include "select.hpp"
#include "knn_fast_common.hpp"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <algorithm>

namespace pcreg {
namespace {

// ---- 1a. bounding box of model + queries (two-stage, deterministic) -----------------
__global__ __launch_bounds__(kBlock) void bbox_partial_kernel(const float* __restrict__ m, int M, int ldm,
                                                              const float* __restrict__ q, int Q, int ldq,
                                                              float* __restrict__ part /*[grid][12]: model lo/hi, query lo/hi*/,
                                                              int32_t* __restrict__ zero_me, int n_zero) {
    // the seeding grid's cell counters are cleared here (saves a memset launch; nothing reads them before seed_fill)
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_zero; i += gridDim.x * kBlock) zero_me[i] = 0;
    float lo[2][3], hi[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[k][c] = INFINITY; hi[k][c] = -INFINITY; }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < M + Q; i += gridDim.x * kBlock) {
        const bool is_m = i < M;
        const float* p = is_m ? m + i : q + (i - M);
        size_t ld = is_m ? (size_t)ldm : (size_t)ldq;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = p[c * ld];
            lo[0][c] = fminf(lo[0][c], is_m ? v : INFINITY); hi[0][c] = fmaxf(hi[0][c], is_m ? v : -INFINITY);
            lo[1][c] = fminf(lo[1][c], is_m ? INFINITY : v); hi[1][c] = fmaxf(hi[1][c], is_m ? -INFINITY : v);
        }
    }
    __shared__ float s[kBlock / 64][12];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[k][c] = fminf(lo[k][c], __shfl_xor(lo[k][c], o)); hi[k][c] = fmaxf(hi[k][c], __shfl_xor(hi[k][c], o)); }
        }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][k * 6 + c] = lo[k][c]; s[threadIdx.x >> 6][k * 6 + 3 + c] = hi[k][c]; }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const bool is_lo = (threadIdx.x % 6) < 3;
        float v = s[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) v = is_lo ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
        part[blockIdx.x * 12 + threadIdx.x] = v;
    }
}
__global__ void bbox_final_kernel(const float* __restrict__ part, int nparts, int M, int cell_cap, Prep* __restrict__ prep,
                                  unsigned* __restrict__ rm2_bits, int32_t* __restrict__ n_flag) {
    float lo2[2][3], hi2[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo2[k][c] = INFINITY; hi2[k][c] = -INFINITY; }
    for (int b = threadIdx.x; b < nparts; b += 64)              // launched with one wave
        for (int k = 0; k < 2; ++k)
            for (int c = 0; c < 3; ++c) { lo2[k][c] = fminf(lo2[k][c], part[b * 12 + k * 6 + c]); hi2[k][c] = fmaxf(hi2[k][c], part[b * 12 + k * 6 + 3 + c]); }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo2[k][c] = fminf(lo2[k][c], __shfl_xor(lo2[k][c], o)); hi2[k][c] = fmaxf(hi2[k][c], __shfl_xor(hi2[k][c], o)); }
        }
    if (threadIdx.x == 0) {
        float lo[3], hi[3];                                     // the joint box
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo2[0][c], lo2[1][c]); hi[c] = fmaxf(hi2[0][c], hi2[1][c]); }
        prep->cx = 0.5f * lo[0] + 0.5f * hi[0]; prep->cy = 0.5f * lo[1] + 0.5f * hi[1]; prep->cz = 0.5f * lo[2] + 0.5f * hi[2];
        {   // an upper bound of max |m~|^2 (and |q~|^2) from the box itself: the seeding margin uses it
            float ax = 0.5f * (hi[0] - lo[0]), ay = 0.5f * (hi[1] - lo[1]), az = 0.5f * (hi[2] - lo[2]);
            prep->rm2 = (ax * ax + ay * ay + az * az) * 1.0001f;
        }
        float H = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) * 0.5f;
        float sg = 1.0f;
        if (H > 0.0f && H < INFINITY) sg = ldexpf(1.0f, 5 - ilogbf(H));        // sigma * H in [32, 64)
        prep->sigma = sg; prep->inv_sigma2 = 1.0f / (sg * sg);                 // powers of two: exact
        prep->pad0 = prep->pad1 = 0.0f;
        // seeding grid: cells of about two model points (density of the MODEL box), laid over the QUERY box
        // plus one cell of margin -- the only cells a query ever looks at; at most cell_cap cells
        float mext[3], memax = 0.0f;
        for (int c = 0; c < 3; ++c) { mext[c] = hi2[0][c] - lo2[0][c]; memax = fmaxf(memax, mext[c]); }
        int n[3] = {1, 1, 1};
        float h = 1.0f, g0[3] = {lo[0], lo[1], lo[2]};
        const bool have_q = lo2[1][0] <= hi2[1][0];
        if (memax > 0.0f && memax < INFINITY && have_q) {
            for (int c = 0; c < 3; ++c) mext[c] = fmaxf(mext[c], memax * 1e-3f);
            h = cbrtf(mext[0] * mext[1] * mext[2] / fmaxf((float)M * 0.5f, 1.0f));
            for (int it = 0; it < 64; ++it) {
                long tot = 1;
                for (int c = 0; c < 3; ++c) {
                    // clip the query box to the model box grown by one cell: cells farther out hold no model point
                    const float a = fmaxf(lo2[1][c], lo2[0][c] - h), b = fminf(hi2[1][c], hi2[0][c] + h);
                    g0[c] = a - h;
                    n[c] = (int)fminf(ceilf(fmaxf(b - a, 0.0f) / h) + 2.0f, 2048.0f); if (n[c] < 1) n[c] = 1; tot *= n[c];
                }
                if (tot <= cell_cap) break;
                h *= 1.2f;
            }
        }
        prep->gx0 = g0[0]; prep->gy0 = g0[1]; prep->gz0 = g0[2]; prep->inv_h = 1.0f / h;
        prep->nx = n[0]; prep->ny = n[1]; prep->nz = n[2]; prep->ncell = n[0] * n[1] * n[2];
        *rm2_bits = 0u;
        *n_flag = 0;                                   // (saves a memset launch)
    }
}
// ---- 1b. model -> {m~, |m~|^2}; R_m^2 by atomicMax on the (non-negative) float bits ----
__global__ __launch_bounds__(kBlock) void prep_model_kernel(const float* __restrict__ m, int M, int ldm,
                                                            const Prep* __restrict__ prep, float4* __restrict__ out,
                                                            unsigned* __restrict__ rm2_bits) {
    const float cx = prep->cx, cy = prep->cy, cz = prep->cz;
    float mx = 0.0f;
    const int Mpad = (M + kMTile - 1) / kMTile * kMTile;   // whole LDS tiles; padding has w = +inf, never a candidate
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < Mpad; i += gridDim.x * kBlock) {
        if (i >= M) { out[i] = make_float4(0.0f, 0.0f, 0.0f, INFINITY); continue; }
        float x = m[i] - cx, y = m[i + (size_t)ldm] - cy, z = m[i + 2 * (size_t)ldm] - cz;
        float w = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        out[i] = make_float4(x, y, z, w);
        mx = fmaxf(mx, w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(rm2_bits, __float_as_uint(mx));     // max is order-independent
}

// ---- 1c. seeding: a first threshold per query from a coarse grid of the model ------------------
// The candidate kernels only touch their sorted lists when a score beats the query's threshold, and a
// wave pays that slow path whenever ANY of its lanes does.  Starting from +inf every lane does so
// O(log n) times; starting from "the 4th-nearest of a few model points around the query" almost never.
// The grid remembers up to kSeedSlots points per cell (whoever arrives first: the threshold is a hint,
// results never depend on it); a query looks at its 27 cells, takes the 4th-smallest EXACT distance d4
// and publishes s-space threshold d4 - |q~|^2 plus twice the score error bound, so that its true four
// nearest are still below it.  Every point that is later skipped was compared with a word >= the final
// word G, which is all the certificate needs.
__device__ __forceinline__ int seed_cell(float v, float lo, float inv_h, int n) {
    int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
__global__ __launch_bounds__(kBlock) void seed_fill_kernel(const float* __restrict__ m, int M, int ldm, const Prep* __restrict__ prep,
                                                           int32_t* __restrict__ cnt, float4* __restrict__ slots) {
    const float gx0 = prep->gx0, gy0 = prep->gy0, gz0 = prep->gz0, ih = prep->inv_h;
    const int nx = prep->nx, ny = prep->ny, nz = prep->nz;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < M; i += gridDim.x * kBlock) {
        const float x = m[i], y = m[i + (size_t)ldm], z = m[i + 2 * (size_t)ldm];
        // the grid covers the query box plus a margin: model points outside it are never looked at
        const float fx = floorf((x - gx0) * ih), fy = floorf((y - gy0) * ih), fz = floorf((z - gz0) * ih);
        if (!(fx >= 0.0f && fx < (float)nx && fy >= 0.0f && fy < (float)ny && fz >= 0.0f && fz < (float)nz)) continue;
        int cx = (int)fx, cy = (int)fy, cz = (int)fz;
        int cell = (cz * ny + cy) * nx + cx;
        int k = atomicAdd(&cnt[cell], 1);
        if (k < kSeedSlots) slots[(size_t)cell * kSeedSlots + k] = make_float4(x, y, z, 0.0f);   // the point itself: one 64-B line per cell
    }
}
// eight lanes per query: lane k of the group looks at cells k, k + 8, k + 16, k + 24 of the 27, keeps its
// own sorted four smallest distances, and three xor-shuffle rounds merge the eight lists
__device__ __forceinline__ void sort4(float (&d)[4]) {
#define PCREG_CS(a, b) { const float lo_ = fminf(d[a], d[b]), hi_ = fmaxf(d[a], d[b]); d[a] = lo_; d[b] = hi_; }
    PCREG_CS(0, 1) PCREG_CS(2, 3) PCREG_CS(0, 2) PCREG_CS(1, 3) PCREG_CS(1, 2)
#undef PCREG_CS
}
__global__ __launch_bounds__(kBlock) void seed_query_kernel(const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int ldm,
                                                            const Prep* __restrict__ prep,
                                                            const int32_t* __restrict__ cnt, const float4* __restrict__ slots,
                                                            int e_mode, unsigned* __restrict__ gthr, int rank) {
    const int qi = (blockIdx.x * kBlock + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool live = qi < Q;
    const int qq = live ? qi : 0;
    const float qx = q[qq], qy = q[qq + (size_t)ldq], qz = q[qq + 2 * (size_t)ldq];
    const int nx = prep->nx, ny = prep->ny, nz = prep->nz;
    const int cx = seed_cell(qx, prep->gx0, prep->inv_h, nx), cy = seed_cell(qy, prep->gy0, prep->inv_h, ny), cz = seed_cell(qz, prep->gz0, prep->inv_h, nz);
    float d[KC] = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c27 = sub + 8 * t;
        const int x = cx + c27 % 3 - 1, y = cy + (c27 / 3) % 3 - 1, z = cz + c27 / 9 - 1;
        if (c27 < 27 && x >= 0 && x < nx && y >= 0 && y < ny && z >= 0 && z < nz) {
            const int cell = (z * ny + y) * nx + x;
            const int n = min(cnt[cell], kSeedSlots);
            float4 pp[kSeedSlots];
#pragma unroll
            for (int k = 0; k < kSeedSlots; ++k) pp[k] = slots[(size_t)cell * kSeedSlots + k];
#pragma unroll
            for (int k = 0; k < kSeedSlots; ++k) {
                if (k < n) {
                    float ex = qx - pp[k].x, ey = qy - pp[k].y, ez = qz - pp[k].z;
                    float dd = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                    if (dd < d[3]) {
                        if (dd < d[1]) { d[3] = d[2]; d[2] = d[1]; if (dd < d[0]) { d[1] = d[0]; d[0] = dd; } else d[1] = dd; }
                        else { if (dd < d[2]) { d[3] = d[2]; d[2] = dd; } else d[3] = dd; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {               // the four smallest of two sorted fours: min(a_i, b_{3-i}), then sort
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = __shfl_xor(d[k], o);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = fminf(d[k], e[3 - k]);
        sort4(d);
    }
    if (!live || sub != 0) return;
    unsigned word = 0xFFFFFFFFu;                           // +inf: no hint
    // The `rank`-th smallest exact distance of the sample bounds the true rank-th nearest distance from above, and any
    // rank >= 2 keeps both true neighbours under the threshold.  Default 4; 2 makes the candidate kernel collect ~2.5
    // points per query instead of ~5.6 and was measured (EXPERIMENTS build, PCREG_KNN_SEED_RANK): the kernel's time does
    // not move (1.568-1.578 vs 1.576-1.581 ms), so the number of list updates is not what it waits for.
    const float dk = rank <= 2 ? d[1] : (rank == 3 ? d[2] : d[3]);
    if (dk < INFINITY) {
        const float tx = qx - prep->cx, ty = qy - prep->cy, tz = qz - prep->cz;
        const double r2 = (double)tx * tx + (double)ty * ty + (double)tz * tz;
        const double E = score_error_bound(e_mode, (double)prep->rm2, sqrt(r2));
        const double u = 5.9604644775390625e-08;
        const double t = (double)dk - r2 + 2.0 * E + 16.0 * u * ((double)dk + fabs((double)dk - r2));
        word = f2ord(nextafterf((float)t, INFINITY));
    }
    gthr[qi] = word;
}

template <int QPT_, int UB_, bool DRY = false>
__global__ __launch_bounds__(kBlock) void knn_candidates_kernel(
    const float* __restrict__ q, int Q, int ldq, const float4* __restrict__ mp, int M, int chunk, int chunk_stride,
    const Prep* __restrict__ prep, unsigned* __restrict__ gthr /*[Q] ordered-uint thresholds*/,
    int32_t* __restrict__ part_idx /*[S][Q][KC]*/, float* __restrict__ part_s) {
    __shared__ float4 tile[kMTile];
    // Query coefficients live in LDS and are re-read once per tile as 128-bit tuples {az, ax, ay, -}:
    // a ds_read_b128 result is an aligned VGPR quad, and with this component order no coefficient
    // shares a VGPR bank (register number mod 4) with the model component it multiplies -- an fma
    // whose src0 sits in the bank of another VGPR source issues at half rate on gfx950
    // (scripts/ubench/vgpr_bank.hip).
    __shared__ float4 qcoef[QPT_][kBlock];
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * (kBlock * QPT_);
    const int sidx = blockIdx.y;
    const int m_begin = min(M, sidx * chunk_stride), m_end = min(M, m_begin + chunk);
    const float cx = prep->cx, cy = prep->cy, cz = prep->cz;

    float thr[QPT_];
    unsigned gseen[QPT_];
    Cand cand[QPT_];
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int qi = q0 + r * kBlock + tid;
        bool ok = qi < Q;
        float x = ok ? q[qi] - cx : 0.0f, y = ok ? q[qi + (size_t)ldq] - cy : 0.0f, z = ok ? q[qi + 2 * (size_t)ldq] - cz : 0.0f;
        qcoef[r][tid] = make_float4(-2.0f * z, -2.0f * x, -2.0f * y, 0.0f);       // exact scaling
#pragma unroll
        for (int k = 0; k < KC; ++k) { cand[r].s[k] = INFINITY; cand[r].i[k] = -1; }
        thr[r] = INFINITY; gseen[r] = 0xFFFFFFFFu;
    }

    for (int t0 = m_begin; t0 < m_end; t0 += kMTile) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMTile / kBlock; ++k) {
            int j = t0 + k * kBlock + tid;
            tile[k * kBlock + tid] = j < m_end ? mp[j] : make_float4(0.0f, 0.0f, 0.0f, INFINITY);   // s = +inf: never a candidate
        }
        // pick up what the other chunks of these queries have already proven
#pragma unroll
        for (int r = 0; r < QPT_; ++r) {
            int qi = q0 + r * kBlock + tid;
            if (qi < Q) {
                unsigned g = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gseen[r] = g;
                thr[r] = fminf(cand[r].s[3], ord2f(g));
            }
        }
        __syncthreads();
        float4 qc[QPT_];
#pragma unroll
        for (int r = 0; r < QPT_; ++r) qc[r] = qcoef[r][tid];
        const int cnt = min(kMTile, m_end - t0);
        const int nb = (cnt + UB_ - 1) / UB_ * UB_;
        for (int jb = 0; jb < nb; jb += UB_) {
            float4 p[UB_];
#pragma unroll
            for (int u = 0; u < UB_; ++u) p[u] = tile[jb + u];
#pragma unroll
            for (int r = 0; r < QPT_; ++r) {
                float s[UB_];
#pragma unroll
                for (int u = 0; u < UB_; ++u)
                    s[u] = __builtin_fmaf(qc[r].y, p[u].x, __builtin_fmaf(qc[r].z, p[u].y, __builtin_fmaf(qc[r].x, p[u].z, p[u].w)));
                // min/max/compare issue at HALF the FMA rate on gfx950 (scripts/ubench/op_rates.hip):
                // fold three values per v_min3 -> 4 selection ops per 8 scores instead of 7
                float mn;
                if (UB_ == 8) {
                    float m1 = fminf(fminf(s[0], s[1]), s[2]);
                    float m2 = fminf(fminf(s[3 % UB_], s[4 % UB_]), s[5 % UB_]);
                    float m3 = fminf(fminf(s[6 % UB_], s[7 % UB_]), m1);
                    mn = fminf(m2, m3);
                } else {
                    mn = fminf(fminf(fminf(s[0], s[1]), s[2]), s[3]);
                }
                if (DRY) { asm volatile("" :: "v"(mn)); }   // timing-only build: hot loop without insertions
                else if (mn < thr[r]) {
                    const int j0 = t0 + jb;
#pragma unroll
                    for (int u = 0; u < UB_; ++u) if (s[u] < thr[r]) cand_insert(cand[r], s[u], j0 + u);
                    thr[r] = fminf(thr[r], cand[r].s[3]);
                }
            }
        }
        // publish a tighter bound (relaxed: a late or lost update only costs pruning)
#pragma unroll
        for (int r = 0; r < QPT_; ++r) {
            int qi = q0 + r * kBlock + tid;
            if (qi < Q && cand[r].s[3] < INFINITY) {
                unsigned k = f2ord(cand[r].s[3]);
                if (k < gseen[r]) { atomicMin(&gthr[qi], k); gseen[r] = k; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int qi = q0 + r * kBlock + tid;
        if (qi < Q && part_idx != nullptr) {          // the seeding pass only publishes thresholds
            size_t o = ((size_t)sidx * Q + qi) * KC;
            *reinterpret_cast<int4*>(part_idx + o) = make_int4(cand[r].i[0], cand[r].i[1], cand[r].i[2], cand[r].i[3]);
            *reinterpret_cast<float4*>(part_s + o) = make_float4(cand[r].s[0], cand[r].s[1], cand[r].s[2], cand[r].s[3]);
        }
    }
}

// ---- 2a'. the same kernel with LDS-DMA double buffering ------------------------------------
// Tile t+1 is copied global -> LDS by `global_load_lds_dwordx4` (no VGPR staging: each wave
// instruction lands 64 x 16 B contiguously) into the other buffer while tile t is scored;
// one barrier per tile instead of two and the L2 latency disappears behind the arithmetic.
// The prepared model is padded with w = +inf to a whole number of tiles, so no tail code.
template <int QPT_, int UB_>
__global__ __launch_bounds__(kBlock) void knn_candidates_dma_kernel(
    const float* __restrict__ q, int Q, int ldq, const float4* __restrict__ mp, int M, int chunk, int chunk_stride,
    const Prep* __restrict__ prep, unsigned* __restrict__ gthr, int32_t* __restrict__ part_idx, float* __restrict__ part_s) {
    __shared__ __attribute__((aligned(16))) float4 tile[2][kMTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * (kBlock * QPT_);
    const int sidx = blockIdx.y;
    const int m_begin = min(M, sidx * chunk_stride), m_end = min(M, m_begin + chunk);
    const float cx = prep->cx, cy = prep->cy, cz = prep->cz;

    float ax[QPT_], ay[QPT_], az[QPT_], thr[QPT_];
    unsigned gseen[QPT_];
    Cand cand[QPT_];
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int qi = q0 + r * kBlock + tid;
        bool ok = qi < Q;
        float x = ok ? q[qi] - cx : 0.0f, y = ok ? q[qi + (size_t)ldq] - cy : 0.0f, z = ok ? q[qi + 2 * (size_t)ldq] - cz : 0.0f;
        ax[r] = -2.0f * x; ay[r] = -2.0f * y; az[r] = -2.0f * z;
#pragma unroll
        for (int k = 0; k < KC; ++k) { cand[r].s[k] = INFINITY; cand[r].i[k] = -1; }
        thr[r] = INFINITY; gseen[r] = 0xFFFFFFFFu;
    }
    // wave w copies the 1-KiB segments w, w+4, w+8, w+12 of a 16-KiB tile
    auto dma_tile = [&](int t0, int buf) {
#pragma unroll
        for (int k = 0; k < kMTile / kBlock; ++k) {
            const int seg = k * (kBlock / 64) + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mp + t0 + seg * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(&tile[buf][seg * 64]), 16, 0, 0);
        }
    };
    const int ntile = (m_end - m_begin + kMTile - 1) / kMTile;
    if (ntile > 0) dma_tile(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int t0 = m_begin + t * kMTile;
        if (t + 1 < ntile) dma_tile(t0 + kMTile, (t + 1) & 1);
#pragma unroll
        for (int r = 0; r < QPT_; ++r) {
            int qi = q0 + r * kBlock + tid;
            if (qi < Q) {
                if (cand[r].s[3] < INFINITY) { unsigned k = f2ord(cand[r].s[3]); if (k < gseen[r]) atomicMin(&gthr[qi], k); }
                unsigned g = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gseen[r] = g;
                thr[r] = fminf(cand[r].s[3], ord2f(g));
            }
        }
        const float4* cur = tile[t & 1];
        const int cnt = min(kMTile, m_end - t0);
        const int nb = (cnt + UB_ - 1) / UB_ * UB_;
        for (int jb = 0; jb < nb; jb += UB_) {
            float4 p[UB_];
#pragma unroll
            for (int u = 0; u < UB_; ++u) p[u] = cur[jb + u];
#pragma unroll
            for (int r = 0; r < QPT_; ++r) {
                float s[UB_];
#pragma unroll
                for (int u = 0; u < UB_; ++u)
                    s[u] = __builtin_fmaf(ax[r], p[u].x, __builtin_fmaf(ay[r], p[u].y, __builtin_fmaf(az[r], p[u].z, p[u].w)));
                // min/max/compare issue at HALF the FMA rate on gfx950 (scripts/ubench/op_rates.hip):
                // fold three values per v_min3 -> 4 selection ops per 8 scores instead of 7
                float mn;
                if (UB_ == 8) {
                    float m1 = fminf(fminf(s[0], s[1]), s[2]);
                    float m2 = fminf(fminf(s[3 % UB_], s[4 % UB_]), s[5 % UB_]);
                    float m3 = fminf(fminf(s[6 % UB_], s[7 % UB_]), m1);
                    mn = fminf(m2, m3);
                } else {
                    mn = fminf(fminf(fminf(s[0], s[1]), s[2]), s[3]);
                }
                if (mn < thr[r]) {
                    const int j0 = t0 + jb;
#pragma unroll
                    for (int u = 0; u < UB_; ++u) if (s[u] < thr[r] && j0 + u < m_end) cand_insert(cand[r], s[u], j0 + u);
                    thr[r] = fminf(thr[r], cand[r].s[3]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile has landed
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int qi = q0 + r * kBlock + tid;
        if (qi < Q) {
            if (cand[r].s[3] < INFINITY) { unsigned k = f2ord(cand[r].s[3]); if (k < gseen[r]) atomicMin(&gthr[qi], k); }
            if (part_idx != nullptr) {
                size_t o = ((size_t)sidx * Q + qi) * KC;
                *reinterpret_cast<int4*>(part_idx + o) = make_int4(cand[r].i[0], cand[r].i[1], cand[r].i[2], cand[r].i[3]);
                *reinterpret_cast<float4*>(part_s + o) = make_float4(cand[r].s[0], cand[r].s[1], cand[r].s[2], cand[r].s[3]);
            }
        }
    }
}

// ---- 2c. candidate generation with the model in SCALAR registers ------------------------
// The model point of a batch is the same for every lane, so it does not need LDS (or VGPRs)
// at all: the prepared float4 stream is read with s_load_dwordx8 through the scalar cache and
// used as the SGPR operand of the FMAs (one SGPR per VOP3 on gfx9: |m~|^2 goes through one
// v_mov per point, shared by the lane's QPT queries).  No LDS, no barriers, ~35 fewer VGPRs
// than the LDS-tiled kernel -> more waves per SIMD.  The next batch is requested before the
// current one is consumed (scalar loads return out of order; one s_waitcnt per batch).
template <int QPT_>
__global__ __launch_bounds__(kBlock) void knn_candidates_sgpr_kernel(
    const float* __restrict__ q, int Q, int ldq, const float4* __restrict__ mp, int M, int chunk,
    const Prep* __restrict__ prep, unsigned* __restrict__ gthr, int32_t* __restrict__ part_idx, float* __restrict__ part_s) {
    constexpr int UB = 8;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * (kBlock * QPT_);
    const int sidx = blockIdx.y;
    const int m_begin = min(M, sidx * chunk), m_end = min(M, m_begin + chunk);
    const float cx = prep->cx, cy = prep->cy, cz = prep->cz;

    float ax[QPT_], ay[QPT_], az[QPT_], thr[QPT_];
    unsigned gseen[QPT_];
    Cand cand[QPT_];
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int qi = q0 + r * kBlock + tid;
        bool ok = qi < Q;
        float x = ok ? q[qi] - cx : 0.0f, y = ok ? q[qi + (size_t)ldq] - cy : 0.0f, z = ok ? q[qi + 2 * (size_t)ldq] - cz : 0.0f;
        ax[r] = -2.0f * x; ay[r] = -2.0f * y; az[r] = -2.0f * z;
#pragma unroll
        for (int k = 0; k < KC; ++k) { cand[r].s[k] = INFINITY; cand[r].i[k] = -1; }
        thr[r] = INFINITY; gseen[r] = 0xFFFFFFFFu;
    }
    // the prepared array is padded to a multiple of 16 points with w = +inf, so whole batches
    // can be read past m_end without a tail loop (a padded point is never a candidate)
    const int nb = (m_end - m_begin + UB - 1) / UB;
    float4 pn[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) pn[u] = mp[m_begin + u];
    for (int b = 0; b < nb; ++b) {
        const int j0 = m_begin + b * UB;
        float4 p[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) p[u] = pn[u];
        if (b + 1 < nb) {
#pragma unroll
            for (int u = 0; u < UB; ++u) pn[u] = mp[j0 + UB + u];
        }
        if ((b & 127) == 0) {      // every 1024 points: exchange thresholds with the other chunks
#pragma unroll
            for (int r = 0; r < QPT_; ++r) {
                int qi = q0 + r * kBlock + tid;
                if (qi < Q) {
                    if (cand[r].s[3] < INFINITY) { unsigned k = f2ord(cand[r].s[3]); if (k < gseen[r]) atomicMin(&gthr[qi], k); }
                    unsigned g = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gseen[r] = g;
                    thr[r] = fminf(cand[r].s[3], ord2f(g));
                }
            }
        }
        float w[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) w[u] = (j0 + u < m_end) ? p[u].w : INFINITY;
#pragma unroll
        for (int r = 0; r < QPT_; ++r) {
            float s[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u)
                s[u] = __builtin_fmaf(ax[r], p[u].x, __builtin_fmaf(ay[r], p[u].y, __builtin_fmaf(az[r], p[u].z, w[u])));
            float mn = fminf(fminf(fminf(s[0], s[1]), fminf(s[2], s[3])), fminf(fminf(s[4], s[5]), fminf(s[6], s[7])));
            if (mn < thr[r]) {
#pragma unroll
                for (int u = 0; u < UB; ++u) if (s[u] < thr[r]) cand_insert(cand[r], s[u], j0 + u);
                thr[r] = fminf(thr[r], cand[r].s[3]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QPT_; ++r) {
        int qi = q0 + r * kBlock + tid;
        if (qi < Q) {
            if (cand[r].s[3] < INFINITY) { unsigned k = f2ord(cand[r].s[3]); if (k < gseen[r]) atomicMin(&gthr[qi], k); }
            size_t o = ((size_t)sidx * Q + qi) * KC;
            *reinterpret_cast<int4*>(part_idx + o) = make_int4(cand[r].i[0], cand[r].i[1], cand[r].i[2], cand[r].i[3]);
            *reinterpret_cast<float4*>(part_s + o) = make_float4(cand[r].s[0], cand[r].s[1], cand[r].s[2], cand[r].s[3]);
        }
    }
}

}  // namespace
size_t knn_f16_prep_bytes(int M);
int launch_knn_candidates_f16(const float* q, int Q, int ldq, const float* m, int M, int ldm, const void* prep,
                              unsigned* rm2, void* mtiles, unsigned* gthr, void* cand_ent, int32_t* cand_cnt,
                              int target_blocks, int max_S, bool dry, int* S_out, int* group16_out, hipStream_t st);
namespace {
// ---- 3. exact re-rank + certificate: one wave per query ------------------------------------
__device__ __forceinline__ bool lex_lt_f(float da, int ia, float db, int ib) {
    return da < db || (da == db && (unsigned)ia < (unsigned)ib);
}
template <int LPQ>      // lanes per query: 64 for the dense lists (S * kc entries), 8 for the short per-query lists
__global__ __launch_bounds__(kBlock) void knn_finalize_kernel(
    const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm,
    const Prep* __restrict__ prep, const unsigned* __restrict__ rm2_bits, const unsigned* __restrict__ gthr,
    const int32_t* __restrict__ part_idx, const float* __restrict__ part_s, int S, int kc, int idx_base,
    int32_t* __restrict__ idx, float* __restrict__ dist, int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag, int e_mode,
    const int32_t* __restrict__ cand_cnt, int group16) {
    // group16 (sparse lists of the pipelined f16 kernel): an entry is (first row jb, minimum score) of the 16 model
    // points jb + 8 (r / 4) + r % 4, r = 0..15, that one lane of knn_candidates_f16_pipe_kernel scored together;
    // pass 2 expands every entry that can still matter.  Points past M (tile padding) are skipped.
    const int lane = threadIdx.x & (LPQ - 1);
    const int qi = blockIdx.x * (kBlock / LPQ) + (threadIdx.x / LPQ);
    if (qi >= Q) return;
    const float qx = q[qi], qy = q[qi + (size_t)ldq], qz = q[qi + 2 * (size_t)ldq];
    // cand_cnt == nullptr: dense lists, kc candidates per (chunk, query) at [chunk][query][kc] (VALU / fp32-MFMA paths).
    // cand_cnt != nullptr: one list per query, cand_cnt[qi] (index, score-bits) pairs at part_idx[(qi * S * kc + e) * 2].
    const bool sparse = cand_cnt != nullptr;
    const int total = sparse ? min(cand_cnt[qi], S * kc) : S * kc;
    const uint2* ent = reinterpret_cast<const uint2*>(part_idx) + (size_t)qi * S * kc;
    // pass 1: the two smallest approximate scores of the union (values only)
    float a1 = INFINITY, a2 = INFINITY;
    for (int e = lane; e < total; e += LPQ) {
        int j; float s;
        if (sparse) { const uint2 v = ent[e]; j = (int)v.x; s = __uint_as_float(v.y); }
        else { size_t o = ((size_t)(e / kc) * Q + qi) * kc + (e % kc); j = part_idx[o]; s = part_s[o]; }
        if (j >= 0) { if (s < a2) { if (s < a1) { a2 = a1; a1 = s; } else a2 = s; } }
    }
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) {
        float b1 = __shfl_xor(a1, o), b2 = __shfl_xor(a2, o);
        float n1 = fminf(a1, b1);
        float n2 = fminf(fmaxf(a1, b1), fminf(a2, b2));
        a1 = n1; a2 = n2;
    }
    // rounding bound (double arithmetic on the fp32 quantities the candidate kernel used)
    const double u = 5.9604644775390625e-08;                     // 2^-24
    const float cxf = prep->cx, cyf = prep->cy, czf = prep->cz;
    const float tx = qx - cxf, ty = qy - cyf, tz = qz - czf;       // the same q~ the candidate kernel formed
    const double r2 = (double)tx * tx + (double)ty * ty + (double)tz * tz;
    const double r = sqrt(r2);
    const double Rm2 = (double)__uint_as_float(*rm2_bits);
    // e_mode 0: scores from the fp32 fma chain.  e_mode 1: scores from the f16-split matrix-core product
    // (knn_mfma16.hip): 16 u r R_m for the two-term f16 representations of Q and m (2^-22 relative each) and
    // 32.1 u (R_m^2 + 2 r R_m) for sixteen fp32 accumulation steps of at most one ulp each.
    const double Eab = score_error_bound(e_mode, Rm2, r);
    // pass 2: exact distances of every candidate that can still reach the top-2
    const float cut = (float)((double)a2 + 2.0 * Eab + 16.0 * u * fabs((double)a2 + r2));
    const float cut_up = nextafterf(cut, INFINITY);              // float rounding of the cut must not exclude anything
    float d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
    for (int e = lane; e < total; e += LPQ) {
        int j; float sc;
        if (sparse) { const uint2 v = ent[e]; j = (int)v.x; sc = __uint_as_float(v.y); }
        else { size_t o = ((size_t)(e / kc) * Q + qi) * kc + (e % kc); j = part_idx[o]; sc = part_s[o]; }
        if (j >= 0 && (sc <= cut_up || !(a2 < INFINITY))) {
            const int nr = group16 ? 16 : 1;
            for (int r = 0; r < nr; ++r) {
                const int jj = group16 ? j + 8 * (r >> 2) + (r & 3) : j;
                if (jj >= M) continue;
                float dx = qx - m[jj], dy = qy - m[jj + (size_t)ldm], dz = qz - m[jj + 2 * (size_t)ldm];
                float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (lex_lt_f(d, jj, d2, i2)) {
                    if (lex_lt_f(d, jj, d1, i1)) { d2 = d1; i2 = i1; d1 = d; i1 = jj; } else { d2 = d; i2 = jj; }
                }
            }
        }
    }
    // wave-shuffle top-2 reduction ordered by (dist, idx); empty slots are (+inf, -1 -> max uint)
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) {
        float e1 = __shfl_xor(d1, o), e2 = __shfl_xor(d2, o);
        int j1 = __shfl_xor(i1, o), j2 = __shfl_xor(i2, o);
        // merge two sorted pairs
        bool first_mine = lex_lt_f(d1, i1, e1, j1);
        float w1 = first_mine ? d1 : e1; int k1 = first_mine ? i1 : j1;
        float x2 = first_mine ? d2 : d1; int y2 = first_mine ? i2 : i1;      // my next
        float x3 = first_mine ? e1 : e2; int y3 = first_mine ? j1 : j2;      // other's next
        bool sec_mine = lex_lt_f(x2, y2, x3, y3);
        d1 = w1; i1 = k1; d2 = sec_mine ? x2 : x3; i2 = sec_mine ? y2 : y3;
    }
    // certificate
    const unsigned gword = gthr[qi];
    const float G = ord2f(gword);
    bool ok;
    if (M <= 2) ok = (i1 >= 0) && (M < 2 || i2 >= 0);          // nothing outside the lists when M <= KC (handled below too)
    else if (gword == 0xFFFFFFFFu) ok = true;                    // never published, never seeded: nothing was skipped, every point is a candidate
    else if (!(G < INFINITY)) ok = false;                        // a published +inf (scores overflowed): no bound, redo exactly
    else {
        double lower = (double)G + r2;
        double E = Eab + 16.0 * u * ((double)d2 + fabs(lower));
        ok = (i2 >= 0) && (lower - E > (double)d2);
    }
    if (lane == 0) {
        if (ok) {
            idx[(size_t)qi * 2] = i1 >= 0 ? i1 + idx_base : -1; idx[(size_t)qi * 2 + 1] = i2 >= 0 ? i2 + idx_base : -1;
            dist[(size_t)qi * 2] = d1; dist[(size_t)qi * 2 + 1] = d2;
        } else {
            int slot = atomicAdd(n_flag, 1);
            flag_list[slot] = qi;
        }
    }
}

// ---- 4a. exact fallback, few queries: kFbSlices workgroups per flagged query + a merge ------------
constexpr int kFew = 1024, kFbSlices = 32;
__global__ __launch_bounds__(kBlock) void knn_fallback_slice_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ m, int M, int ldm,
    const int32_t* __restrict__ flag_list, const int32_t* __restrict__ n_flag,
    int32_t* __restrict__ fb_idx /*[kFew][kFbSlices][2]*/, float* __restrict__ fb_dist) {
    const int nf = *n_flag;
    if (nf > kFew) return;                                       // the tiled kernel handles big lists
    __shared__ float sd[kBlock / 64][2];
    __shared__ int si[kBlock / 64][2];
    const int len = (M + kFbSlices - 1) / kFbSlices;
    const int j0 = blockIdx.y * len, j1 = min(M, j0 + len);
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {           // a small grid: usually there is nothing to do
        const int qi = flag_list[f];
        const float qx = q[qi], qy = q[qi + (size_t)ldq], qz = q[qi + 2 * (size_t)ldq];
        float d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
        for (int j = j0 + threadIdx.x; j < j1; j += kBlock) {     // ascending j per thread: strict '<' keeps ties low
            float dx = qx - m[j], dy = qy - m[j + (size_t)ldm], dz = qz - m[j + 2 * (size_t)ldm];
            float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (d < d2) { if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j; } else { d2 = d; i2 = j; } }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float e1 = __shfl_xor(d1, o), e2 = __shfl_xor(d2, o);
            int j1s = __shfl_xor(i1, o), j2s = __shfl_xor(i2, o);
            bool first_mine = lex_lt_f(d1, i1, e1, j1s);
            float w1 = first_mine ? d1 : e1; int k1 = first_mine ? i1 : j1s;
            float x2 = first_mine ? d2 : d1; int y2 = first_mine ? i2 : i1;
            float x3 = first_mine ? e1 : e2; int y3 = first_mine ? j1s : j2s;
            bool sec_mine = lex_lt_f(x2, y2, x3, y3);
            d1 = w1; i1 = k1; d2 = sec_mine ? x2 : x3; i2 = sec_mine ? y2 : y3;
        }
        const int w = threadIdx.x >> 6;
        __syncthreads();                                          // the previous trip's readers are done with sd / si
        if ((threadIdx.x & 63) == 0) { sd[w][0] = d1; sd[w][1] = d2; si[w][0] = i1; si[w][1] = i2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            Top2T<float> t{INFINITY, INFINITY, -1, -1};
            for (int k = 0; k < kBlock / 64; ++k) { top2_insert_lex_t(t, sd[k][0], si[k][0]); top2_insert_lex_t(t, sd[k][1], si[k][1]); }
            const size_t o = ((size_t)f * kFbSlices + blockIdx.y) * 2;
            fb_idx[o] = t.i1; fb_idx[o + 1] = t.i2; fb_dist[o] = t.d1; fb_dist[o + 1] = t.d2;
        }
    }
}
__global__ void knn_fallback_merge_kernel(const int32_t* __restrict__ flag_list, const int32_t* __restrict__ n_flag, int idx_base,
                                          const int32_t* __restrict__ fb_idx, const float* __restrict__ fb_dist,
                                          int32_t* __restrict__ idx, float* __restrict__ dist) {
    const int nf = *n_flag, f = blockIdx.x * blockDim.x + threadIdx.x;
    if (nf > kFew || f >= nf) return;
    Top2T<float> t{INFINITY, INFINITY, -1, -1};
    for (int sl = 0; sl < kFbSlices; ++sl) {
        const size_t o = ((size_t)f * kFbSlices + sl) * 2;
        top2_insert_lex_t(t, fb_dist[o], fb_idx[o]); top2_insert_lex_t(t, fb_dist[o + 1], fb_idx[o + 1]);
    }
    const int qi = flag_list[f];
    idx[(size_t)qi * 2] = t.i1 >= 0 ? t.i1 + idx_base : -1; idx[(size_t)qi * 2 + 1] = t.i2 >= 0 ? t.i2 + idx_base : -1;
    dist[(size_t)qi * 2] = t.d1; dist[(size_t)qi * 2 + 1] = t.d2;
}

int pick_splits_fast(int n_tiles, int M, int target) {
    int S = (target + n_tiles - 1) / n_tiles;
    int maxS = (M + kMTile - 1) / kMTile;
    if (S > maxS) S = maxS;
    return S < 1 ? 1 : S;
}

}  // namespace

// MFMA formulation of the candidate kernel (knn_mfma.hip, built with its own flags)
int launch_knn_candidates_mfma(const float* q, int Q, int ldq, const float* m, int M, int ldm, const void* prep,
                               unsigned* rm2, float* mtiles, int n_tiles, int tiles_per_chunk, int q_blocks, int S,
                               unsigned* gthr, int32_t* part_idx, float* part_s, bool dry, hipStream_t st);

// tiled exact kernel on a query list (implemented in knn_points.hip)
int launch_knn2_points_exact_list(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                                  const int32_t* qlist, const int32_t* n_list, int min_active, int32_t* idx,
                                  float* dist, void* ws, size_t ws_bytes, hipStream_t st);
size_t knn2_points_exact_workspace_bytes(int Q, int M);

// workspace layout: Prep | rm2 bits | n_flag | bbox partials | gthr [Q] | flag_list [Q]
//                   | prepared model (16 B/point, padded to whole 16-point tiles)
//                   | part_idx [S][Q][kc] | part_s | exact-kernel workspace (fallback)
static constexpr int kPartCap = 40;           // upper bound of S * kc / 16 any variant may use (x16 entries per query)
static constexpr int kSeedMinM = 16 * 1024;      // below this the lists settle within the first tiles anyway
// the grid is sized on the device (about M/2 cells); this is the capacity it may use
static size_t seed_cell_cap(int M) { return std::min<size_t>((size_t)kSeedMaxCells, std::max<size_t>(4096, (size_t)M)); }
static size_t seed_bytes(int M) {
    if (M < kSeedMinM) return 0;
    const size_t cells = seed_cell_cap(M);
    return align_up(cells * 4, 256) + align_up(cells * kSeedSlots * 16, 256);
}
static size_t fast_fixed_bytes(int Q, int M) {
    size_t q = (size_t)(Q > 0 ? Q : 1), mm = (size_t)(M > 0 ? M : 1) + kMTile + 16;
    return 256 + 256 + 256 + align_up(512 * 12 * sizeof(float), 256) + align_up(q * 4, 256) + align_up(q * 4, 256) +
           align_up(std::max(mm * 16, knn_f16_prep_bytes(M)), 256) + 2 * align_up((size_t)kPartCap * 16 * q * 4, 256) +
           seed_bytes(M) + 2 * align_up((size_t)1024 * 32 * 2 * 4, 256) + align_up(q * 4, 256);
}

size_t knn2_points_fast_workspace_bytes(int Q, int M) {
    return fast_fixed_bytes(Q, M) + knn2_points_exact_workspace_bytes(Q, M);
}

// PCREG_KNN_VARIANT: 40 (default) f16-split matrix-core candidates (knn_mfma16.hip), 41 its timing-only form;
//                    0 VALU candidates QPT4/UB8 (the previous default); 11 QPT4/UB4; 12 QPT8/UB8; 13 QPT2/UB8; 19 VALU
//                    timing-only (no insertions); 5 MFMA candidates; 9 MFMA timing-only.
//                    The fp32 MFMA shares the SIMD's fp32 datapath with the VALU (measured: 32.5 -> 49 cycles per
//                    MFMA once three VALU ops sit between issues, scripts/ubench/mfma_f32.hip), so the MFMA
//                    formulation is slower than the 3-FMA VALU form here and is kept for reference only.
//                    PCREG_KNN_BLOCKS: grid target.
int launch_knn2_points_fast_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                                int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st) {
    PCREG_ARG(Q >= 0 && M >= 0 && ldq >= Q && ldm >= M);
    if (Q == 0) return PCREG_OK;
    size_t need = knn2_points_fast_workspace_bytes(Q, M);
    if (ws_bytes < need) { set_error("knn (fast) workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    const int target_env = PCREG_EXP_ENV("PCREG_KNN_BLOCKS", 0);
    const int variant = PCREG_EXP_ENV("PCREG_KNN_VARIANT", 40);
    const bool use_mfma = variant >= 5 && variant < 10;
    size_t qq = (size_t)Q, mm = (size_t)(M > 0 ? M : 1) + kMTile + 16;
    char* w = (char*)ws;
    Prep* prep = (Prep*)w;                 w += 256;
    unsigned* rm2 = (unsigned*)w;          w += 256;
    int32_t* n_flag = (int32_t*)w;         w += 256;
    float* bpart = (float*)w;              w += align_up(512 * 12 * sizeof(float), 256);
    unsigned* gthr = (unsigned*)w;         w += align_up(qq * 4, 256);
    int32_t* flag_list = (int32_t*)w;      w += align_up(qq * 4, 256);
    void* mprep = w;                       w += align_up(std::max(mm * 16, knn_f16_prep_bytes(M)), 256);
    int32_t* part_idx = (int32_t*)w;       w += align_up((size_t)kPartCap * 16 * qq * 4, 256);
    float* part_s = (float*)w;             w += align_up((size_t)kPartCap * 16 * qq * 4, 256);
    const size_t seed_cells = seed_cell_cap(M);
    int32_t* seed_cnt = (int32_t*)w;       w += M >= kSeedMinM ? align_up(seed_cells * 4, 256) : 0;
    float4* seed_slots = (float4*)w;       w += M >= kSeedMinM ? align_up(seed_cells * kSeedSlots * 16, 256) : 0;
    int32_t* fb_idx = (int32_t*)w;         w += align_up((size_t)1024 * 32 * 2 * 4, 256);
    float* fb_dist = (float*)w;            w += align_up((size_t)1024 * 32 * 2 * 4, 256);
    int32_t* cand_cnt = (int32_t*)w;       w += align_up(qq * 4, 256);
    void* ews = w;
    size_t ews_bytes = ws_bytes - (size_t)(w - (char*)ws);

    int nb = (M + Q + kBlock * 16 - 1) / (kBlock * 16); if (nb > 512) nb = 512; if (nb < 1) nb = 1;
    const bool no_seed = PCREG_EXP_ENV("PCREG_KNN_NOSEED", 0) != 0;
    const bool seeded = M >= kSeedMinM && !no_seed;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(kBlock), 0, st, m, M, ldm, q, Q, ldq, bpart, seed_cnt, seeded ? (int)seed_cells : 0);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, st, bpart, nb, M, (int)seed_cell_cap(M), prep, rm2, n_flag);
    int S = 1, kc = KC, e_mode = (variant == 40 || variant == 41) ? 1 : 0, group16 = 0;
    bool sparse_lists = false;
    if (seeded) {                               // first thresholds from the grid (stage 1c)
        int fb = (M + kBlock - 1) / kBlock; if (fb > 16384) fb = 16384;     // one point per thread: the atomics want parallelism
        hipLaunchKernelGGL(seed_fill_kernel, dim3(fb), dim3(kBlock), 0, st, m, M, ldm, prep, seed_cnt, seed_slots);
        hipLaunchKernelGGL(seed_query_kernel, dim3((Q * 8 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, q, Q, ldq, m, ldm, prep,
                           seed_cnt, seed_slots, e_mode, gthr, PCREG_EXP_ENV("PCREG_KNN_SEED_RANK", 4));
    } else {
        PCREG_HIP(hipMemsetAsync(gthr, 0xFF, qq * 4, st));        // +inf in the ordered-uint image
    }
    if (variant == 40 || variant == 41) {        // f16-split matrix-core candidates (41: timing only)
        kc = KC;
        sparse_lists = true;             // entries of both 4-byte arrays' space: [Q][S * KC] (index, score) pairs
        int rc = launch_knn_candidates_f16(q, Q, ldq, m, M, ldm, prep, rm2, mprep, gthr, part_idx, cand_cnt,
                                           target_env > 0 ? target_env : 4096, kPartCap * 2, variant == 41, &S, &group16, st);
        if (rc) return rc;
    } else if (use_mfma) {
        constexpr int NQ = 8;                    // must match knn_mfma.hip
        const int n_tiles = (M + 15) / 16;
        const int q_blocks = (Q + (kBlock / 64) * NQ * 16 - 1) / ((kBlock / 64) * NQ * 16);
        // whole rounds of resident workgroups (4 per CU at <= 128 VGPRs): avoid a ragged last round
        const int target = target_env > 0 ? target_env : 2048;
        S = target / q_blocks; if (S < 1) S = 1;
        if (S > kPartCap) S = kPartCap;
        if (S > n_tiles) S = n_tiles > 0 ? n_tiles : 1;
        int tiles_per_chunk = n_tiles > 0 ? (n_tiles + S - 1) / S : 1;
        S = n_tiles > 0 ? (n_tiles + tiles_per_chunk - 1) / tiles_per_chunk : 1;
        kc = 16;
        PCREG_HIP(hipMemsetAsync(part_idx, 0xFF, (size_t)S * qq * 16 * 4, st));     // -1: empty slots (M == 0, padding)
        if (M > 0) {
            int rc = launch_knn_candidates_mfma(q, Q, ldq, m, M, ldm, prep, rm2, (float*)mprep, n_tiles, tiles_per_chunk,
                                                q_blocks, S, gthr, part_idx, part_s, variant == 9, st);
            if (rc) return rc;
        }
    } else {
        const int qpt = (variant == 12 || variant == 21) ? 8 : ((variant == 13 || variant == 22) ? 2 : ((variant == 15 || variant == 16 || variant == 32) ? 3 : 4));
        int n_qt = (Q + kBlock * qpt - 1) / (kBlock * qpt);
        S = pick_splits_fast(n_qt, M > 0 ? M : 1, target_env > 0 ? target_env : 4096);
        if (S > kPartCap * 4) S = kPartCap * 4;
        int chunk = (((M > 0 ? M : 1) + S - 1) / S + kMTile - 1) / kMTile * kMTile;
        S = M > 0 ? (M + chunk - 1) / chunk : 1;
        if (M > 0) {
            int pb = (M + kMTile + kBlock * 4 - 1) / (kBlock * 4); if (pb > 2048) pb = 2048;
            hipLaunchKernelGGL(prep_model_kernel, dim3(pb), dim3(kBlock), 0, st, m, M, ldm, prep, (float4*)mprep, rm2);
        }
        dim3 grid(n_qt, S);
#define PCREG_CAND_LAUNCH(QP, UBV) hipLaunchKernelGGL((knn_candidates_kernel<QP, UBV>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, chunk, prep, gthr, part_idx, part_s)
        switch (variant) {
            case 20: hipLaunchKernelGGL((knn_candidates_sgpr_kernel<4>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, prep, gthr, part_idx, part_s); break;
            case 21: hipLaunchKernelGGL((knn_candidates_sgpr_kernel<8>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, prep, gthr, part_idx, part_s); break;
            case 22: hipLaunchKernelGGL((knn_candidates_sgpr_kernel<2>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, prep, gthr, part_idx, part_s); break;
            case 30: hipLaunchKernelGGL((knn_candidates_dma_kernel<4, 8>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, chunk, prep, gthr, part_idx, part_s); break;
            case 31: hipLaunchKernelGGL((knn_candidates_dma_kernel<4, 4>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, chunk, prep, gthr, part_idx, part_s); break;
            case 32: hipLaunchKernelGGL((knn_candidates_dma_kernel<3, 4>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, chunk, prep, gthr, part_idx, part_s); break;
            case 15: PCREG_CAND_LAUNCH(3, 4); break;
            case 16: PCREG_CAND_LAUNCH(3, 8); break;
            case 11: PCREG_CAND_LAUNCH(4, 4); break;
            case 12: PCREG_CAND_LAUNCH(8, 8); break;
            case 13: PCREG_CAND_LAUNCH(2, 8); break;
            case 19: hipLaunchKernelGGL((knn_candidates_kernel<4, 8, true>), grid, dim3(kBlock), 0, st, q, Q, ldq, (const float4*)mprep, M, chunk, chunk, prep, gthr, part_idx, part_s); break;
            default: PCREG_CAND_LAUNCH(4, 8); break;
        }
#undef PCREG_CAND_LAUNCH
    }
    PCREG_HIP(hipGetLastError());
    if (sparse_lists)
        hipLaunchKernelGGL(knn_finalize_kernel<8>, dim3((Q + kBlock / 8 - 1) / (kBlock / 8)), dim3(kBlock), 0, st, q, Q, ldq, m, M, ldm, prep, rm2, gthr,
                           part_idx, part_s, S, kc, (int)idx_base, idx, dist, flag_list, n_flag, e_mode, (const int32_t*)cand_cnt, group16);
    else
        hipLaunchKernelGGL(knn_finalize_kernel<64>, dim3((Q + 3) / 4), dim3(kBlock), 0, st, q, Q, ldq, m, M, ldm, prep, rm2, gthr,
                           part_idx, part_s, S, kc, (int)idx_base, idx, dist, flag_list, n_flag, e_mode, (const int32_t*)nullptr, 0);
    PCREG_HIP(hipGetLastError());
    if (PCREG_EXP_ENV("PCREG_KNN_DEBUG", 0)) {
        int32_t nf = 0;
        PCREG_HIP(hipMemcpyAsync(&nf, n_flag, 4, hipMemcpyDeviceToHost, st)); PCREG_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "[pcreg] knn fast: Q=%d M=%d variant=%d S=%d kc=%d unproven=%d\n", Q, M, variant, S, kc, nf);
    }
    // fallbacks (both launched; each decides from the device-side count which one works)
    hipLaunchKernelGGL(knn_fallback_slice_kernel, dim3(64, kFbSlices), dim3(kBlock), 0, st, q, ldq, m, M, ldm, flag_list, n_flag, fb_idx, fb_dist);
    hipLaunchKernelGGL(knn_fallback_merge_kernel, dim3(kFew / 256), dim3(256), 0, st, flag_list, n_flag, (int)idx_base, fb_idx, fb_dist, idx, dist);
    PCREG_HIP(hipGetLastError());
    return launch_knn2_points_exact_list(q, Q, ldq, m, M, ldm, idx_base, flag_list, n_flag, kFew, idx, dist, ews, ews_bytes, st);
}

}  // namespace pcreg
