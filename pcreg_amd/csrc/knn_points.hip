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
#include "knn_fast_common.hpp"               // kBlock, kMTile, the query grid (UgPrep ...), SearchCounters
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace pcreg {
namespace {


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

// ---------------------------------------------------------------- Unique back-check on a grid of the queries
// keep (i, j) iff no other query i' beats i for model point j in the (distance, index) order of the search.
// That is a range-emptiness question with the candidate's OWN distance as radius (small: j is i's nearest
// model point), so a uniform grid over the queries answers it exactly with a handful of distance evaluations
// instead of a second all-pairs search.  The grid is a by-product of the search call (knn_fast.hip: boxes in
// seed_query_kernel, geometry by a surplus workgroup of the candidate kernel, cells in knn_finalize_kernel).  Cells
// hold up to kUgSlots points; a candidate that meets a fuller cell, or whose ball covers too many cells, is
// checked against every query by its wave (`unsure` in match_finish_kernel).  The cell index is a monotone function of the
// coordinate, so the cell range provably contains every query whose ROUNDED distance is <= d.

// ---------------------------------------------------------------- the match stage in ONE launch
// threshold + ratio test (matchFeatures' removeWeakMatches / removeAmbiguousMatches), the Unique back-check on the
// query grid the search call left in its workspace, and the ordered compaction into pairs + matched coordinates
// (completeExperimentFast.m:205-206) -- round 2 ran eight launches for this.
//   kMatchSingle     one rank: everything.
//   kMatchTable      multi-GPU, before the exchange: this rank's contribution to the [4][Q] table of 4-byte words,
//                    column = QUERY: rows 0-2 = coordinate bits of the query's nearest model point and row 3 = its
//                    Unique verdict (1 if Unique is off) when that point's row lives in this shard and the query is
//                    a candidate; zero otherwise -- an integer SUM over the ranks assembles the table exactly.
//   kMatchFromTable  multi-GPU, after the exchange: candidates again (deterministic), verdicts and coordinates from
//                    the table, ordered compaction.
// Ordered compaction across workgroups without a second launch: workgroups take TICKETS (so workgroup b started
// after every workgroup before it), publish their kept count with a ready bit and add up the counts of their
// predecessors (decoupled look-back) -- a predecessor publishes before it waits, so nobody can wait for ever.
enum { kMatchSingle = 0, kMatchTable = 1, kMatchFromTable = 2 };
// EIGHT lanes per query (like seed_query_kernel): the cells a candidate's ball touches are dealt to the eight lanes, so
// the walk is one dependent load chain deep instead of one per cell (a lane-per-candidate form of this kernel took 49 us
// for the benchmark's 50 k queries).
constexpr int kMB = 1024;                            // threads per workgroup: few workgroups = few tickets, few status words
constexpr int kMLanes = 8, kMQ = kMB / kMLanes;      // 128 queries per chunk of a workgroup
constexpr int kMatchCpbMax = 16;                     // chunks per workgroup: Q <= 2048 * 128 * 16 = 4 Mi

template <int MODE>
__global__ __launch_bounds__(kMB) void match_finish_kernel(
    const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm, int m_lo, int M_total,
    const int32_t* __restrict__ idx, const float* __restrict__ dist, float thr, float ratio, int unique,
    const UgPrep* __restrict__ ug_prep, const int32_t* __restrict__ ug_cnt, const float4* __restrict__ ug_slots,
    int32_t* __restrict__ table, SearchCounters* __restrict__ ctr, int cpb,
    uint32_t* __restrict__ pairs, double* __restrict__ pts1, double* __restrict__ pts2, int32_t* __restrict__ n_pairs) {
    __shared__ int s_ticket;
    __shared__ int s_cnt[kMatchCpbMax][kMB / 64];
    __shared__ int s_red[kMB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & (kMLanes - 1);
    const unsigned long long heads = 0x0101010101010101ull;          // the lanes with sub == 0: one per query
    int b = blockIdx.x;
    if (MODE != kMatchTable) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(&ctr->ticket, 1);
        __syncthreads();
        b = s_ticket;
    }
    const int base_q = b * cpb * kMQ;
    unsigned long long keep_bits = 0ull;
    for (int c = 0; c < cpb; ++c) {
        const int qi = base_q + c * kMQ + (threadIdx.x >> 3);
        const bool cand = qi < Q && filter_keep<float>(idx, dist, qi, M_total, thr, ratio);     // the same in all eight lanes
        const int j = cand ? idx[(size_t)qi * 2] : -1;             // global model row
        bool kp = cand;
        if (MODE == kMatchFromTable) {
            kp = cand && table[qi + 3 * (size_t)Q] != 0;
        } else {
            const int jl = j - m_lo;
            const bool mine = cand && jl >= 0 && jl < M;
            float px = 0.0f, py = 0.0f, pz = 0.0f, di = 0.0f;
            bool und = false;
            if (mine) { px = m[jl]; py = m[jl + (size_t)ldm]; pz = m[jl + 2 * (size_t)ldm]; }
            if (unique) {                      // (the shuffles below want every lane here)
                bool beaten = false, unsure = false;
                if (mine) {
                    const UgPrep P = *ug_prep;
                    di = ug_d2(q[qi], q[qi + (size_t)ldq], q[qi + 2 * (size_t)ldq], px, py, pz);
                    const float r = sqrtf(di) * 1.0001f + 1e-30f;                // covers every point whose rounded distance is <= di
                    const int x0 = ug_cell1(px - r, P.x0, P.inv_c, P.nx), x1 = ug_cell1(px + r, P.x0, P.inv_c, P.nx);
                    const int y0 = ug_cell1(py - r, P.y0, P.inv_c, P.ny), y1 = ug_cell1(py + r, P.y0, P.inv_c, P.ny);
                    const int z0 = ug_cell1(pz - r, P.z0, P.inv_c, P.nz), z1 = ug_cell1(pz + r, P.z0, P.inv_c, P.nz);
                    const int ex = x1 - x0 + 1, ey = y1 - y0 + 1;
                    const long long ncl = (long long)ex * ey * (z1 - z0 + 1);
                    unsure = !(di == di) || ncl > kUgMaxVisit;
                    const int ncell = unsure ? 0 : (int)ncl;
                    for (int k = sub; k < ncell && !beaten; k += kMLanes) {        // this lane's cells of the ball
                        const int cx = x0 + k % ex, cy = y0 + (k / ex) % ey, cz = z0 + k / (ex * ey);
                        const int cell = (cz * P.ny + cy) * P.nx + cx;
                        const int cn = ug_cnt[cell];
                        if (cn > kUgSlots) { unsure = true; continue; }             // an overflowing cell: the scan below decides
                        for (int sl = 0; sl < cn; ++sl) {
                            const float4 t = ug_slots[(size_t)cell * kUgSlots + sl];
                            const int it = __float_as_int(t.w);
                            const float d = ug_d2(t.x, t.y, t.z, px, py, pz);
                            if (d < di || (d == di && it < qi)) { beaten = true; break; }
                        }
                    }
                }
#pragma unroll
                for (int o = 1; o < kMLanes; o <<= 1) {        // every lane takes part in every shuffle: no short-circuit `||` here
                    const int ob = __shfl_xor((int)beaten, o), ou = __shfl_xor((int)unsure, o);
                    beaten = beaten | (ob != 0);
                    unsure = unsure | (ou != 0);
                }
                kp = mine && !beaten;
                und = mine && !beaten && unsure;
            } else {
                kp = mine;
            }
            // a candidate the grid could not decide (an overflowing cell, a ball over too many cells): the wave scans
            // every query for it, together
            unsigned long long um = __ballot(und) & heads;
            while (um != 0ull) {
                const int l = __builtin_ctzll(um); um &= um - 1ull;
                const float bx = __shfl(px, l), by = __shfl(py, l), bz = __shfl(pz, l), bd = __shfl(di, l);
                const int bi = __shfl(qi, l);
                bool beaten = false;
                for (int t0 = 0; t0 < Q && !beaten; t0 += 4 * 64) {
                    bool bb = false;
                    float x[4], y[4], z[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = min(t0 + u * 64 + lane, Q - 1);
                        x[u] = q[t]; y[u] = q[t + (size_t)ldq]; z[u] = q[t + 2 * (size_t)ldq];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + u * 64 + lane;
                        const float d = ug_d2(x[u], y[u], z[u], bx, by, bz);
                        bb = bb || (t < Q && (d < bd || (d == bd && t < bi)));
                    }
                    beaten = __any(bb);
                }
                if ((lane & ~(kMLanes - 1)) == l) kp = !beaten;             // the query's eight lanes
            }
            if (MODE == kMatchTable) {
                if (qi < Q && sub == 0) {
                    table[qi] = mine ? __float_as_int(px) : 0;
                    table[qi + (size_t)Q] = mine ? __float_as_int(py) : 0;
                    table[qi + 2 * (size_t)Q] = mine ? __float_as_int(pz) : 0;
                    table[qi + 3 * (size_t)Q] = (mine && kp) ? 1 : 0;
                }
                continue;
            }
        }
        const unsigned long long bal = __ballot(kp) & heads;
        if (lane == 0) s_cnt[c][wave] = __popcll(bal);
        keep_bits |= kp ? (1ull << c) : 0ull;
    }
    if (MODE == kMatchTable) return;
    __syncthreads();
    int mine_total = 0;
    for (int c = 0; c < cpb; ++c)
#pragma unroll
        for (int w = 0; w < kMB / 64; ++w) mine_total += s_cnt[c][w];
    // The kept counts of the workgroups before this one (in ticket order): publish the own count with a ready bit, then
    // all 1024 threads read the predecessors' words in ONE parallel round (every workgroup of this launch starts at about
    // the same time, so a decoupled look-back finds only aggregates and degenerates into a serial walk: 24 windows of 64
    // for the last workgroup, measured 20 us slower).  A status word is a self-contained flag, so every access is a RELAXED
    // agent-scope atomic: an acquire or a release at agent scope invalidates / writes back the XCD's L2 on this chip (with
    // acquire polls this loop took 670 us).
    if (threadIdx.x == 0) __hip_atomic_store(&ctr->status[b], (int)(0x80000000u | (unsigned)mine_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int v = 0;
    for (int p = threadIdx.x; p < b; p += kMB) {
        int sv;
        do { sv = __hip_atomic_load(&ctr->status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (sv >= 0);
        v += sv & 0x7FFFFFFF;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < kMB / 64; ++w) base += s_red[w];
    if (b == (int)gridDim.x - 1 && threadIdx.x == 0) *n_pairs = base + mine_total;
    for (int c = 0; c < cpb; ++c) {
        const bool kp = (keep_bits >> c) & 1ull;
        const unsigned long long bal = __ballot(kp) & heads;
        int o = base;
        for (int w = 0; w < wave; ++w) o += s_cnt[c][w];
        if (kp) {                          // all eight lanes of a kept query: they share the writes
            o += __popcll(bal & ((1ull << (lane & ~(kMLanes - 1))) - 1ull));
            const int qi = base_q + c * kMQ + (threadIdx.x >> 3);
            const int j = idx[(size_t)qi * 2];
            if (pairs && sub == 6) pairs[(size_t)o * 2] = (uint32_t)qi + 1u;
            if (pairs && sub == 7) pairs[(size_t)o * 2 + 1] = (uint32_t)j + 1u;
            if (pts1 && sub < 3) pts1[o + (size_t)sub * Q] = (double)q[qi + (size_t)sub * ldq];
            if (pts1 && sub >= 3 && sub < 6) {
                const int cc = sub - 3;
                if (MODE == kMatchFromTable) pts2[o + (size_t)cc * Q] = (double)__int_as_float(table[qi + (size_t)cc * Q]);
                else pts2[o + (size_t)cc * Q] = (double)m[(j - m_lo) + (size_t)cc * ldm];
            }
        }
#pragma unroll
        for (int w = 0; w < kMB / 64; ++w) base += s_cnt[c][w];
    }
    // the last workgroup through leaves the counters as it found them, so that the match stage may run again on the same
    // search (every other workgroup has read the status words it needed before it counted itself in)
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ctr->finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket == (int)gridDim.x - 1) {
        for (int p = threadIdx.x; p < (int)gridDim.x; p += kMB) ctr->status[p] = 0;
        if (threadIdx.x == 0) { ctr->ticket = 0; ctr->finished = 0; }
    }
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
                                int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st, bool timed);

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

// pcreg_debug_set("knn_exact", 1) forces the direct-form kernel (tuning / A-B runs); the default is the
// certified fast path of knn_fast.hip, which returns the same bits.
static bool use_exact_only() { return debug_flag(kDbgKnnExact) != 0; }

size_t knn2_points_workspace_bytes(int Q, int M) {
    size_t a = knn2_points_exact_workspace_bytes(Q, M), b = knn2_points_fast_workspace_bytes(Q, M);
    return a > b ? a : b;
}

int launch_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                           int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st, bool timed) {
    if (use_exact_only())
        return knn2_exact_impl(q, Q, ldq, m, M, ldm, idx_base, nullptr, nullptr, 0, idx, dist, ws, ws_bytes, st);
    return launch_knn2_points_fast_f32(q, Q, ldq, m, M, ldm, idx_base, idx, dist, ws, ws_bytes, st, timed);
}

int launch_merge_top2_f32(const int32_t* idx_in, const float* dist_in, int R, int Q, int32_t* idx, float* dist,
                          hipStream_t st, size_t rank_stride) {
    PCREG_ARG(R >= 1 && Q >= 0 && (rank_stride == 0 || rank_stride >= (size_t)Q * 2));
    if (Q == 0) return PCREG_OK;
    hipLaunchKernelGGL(merge_top2_kernel_t<float>, dim3((Q + 255) / 256), dim3(256), 0, st, idx_in, dist_in, R, Q, idx, dist, rank_stride);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

// ---- the match stage on a finished search: launchers ----------------------------------------------------------------
static int match_shape(int Q, int* blocks, int* cpb) {
    int c = 1;
    while ((long long)kMatchMaxBlocks * kMQ * c < Q) ++c;
    if (c > kMatchCpbMax) { set_error("match: %d queries exceed the %d a call handles", Q, kMatchMaxBlocks * kMQ * kMatchCpbMax); return PCREG_E_ARG; }
    *cpb = c; *blocks = (Q + kMQ * c - 1) / (kMQ * c);
    return PCREG_OK;
}
int launch_match_finish(const ModelView& v, const float* q, int Q, int ldq, const int32_t* idx, const float* dist, float thr,
                        float ratio, int unique, void* ws, size_t ws_bytes, uint32_t* pairs, double* pts1, double* pts2,
                        int32_t* n_pairs, hipStream_t st) {
    PCREG_ARG(Q >= 0 && ldq >= Q && (pts1 == nullptr) == (pts2 == nullptr));
    if (Q == 0 || v.M == 0) { PCREG_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), st)); return PCREG_OK; }
    size_t need; SearchWs s = search_ws_layout(Q, v.M, ws, &need);
    if (ws_bytes < need) { set_error("match workspace too small: %zu < %zu (pass the search call's workspace)", ws_bytes, need); return PCREG_E_WORKSPACE; }
    int blocks, cpb; { int rc = match_shape(Q, &blocks, &cpb); if (rc) return rc; }
    hipLaunchKernelGGL(match_finish_kernel<kMatchSingle>, dim3(blocks), dim3(kMB), 0, st, q, Q, ldq, v.m, v.M, v.ldm, 0, v.M, idx, dist, thr, ratio, unique,
                       (const UgPrep*)s.ug_prep, (const int32_t*)s.ug_cnt, (const float4*)s.ug_slots, (int32_t*)nullptr, (SearchCounters*)s.ctr, cpb,
                       pairs, pts1, pts2, n_pairs);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
int launch_match_table(const ModelView& v, int32_t m_lo, int M_total, const float* q, int Q, int ldq, const int32_t* idx,
                       const float* dist, float thr, float ratio, int unique, void* ws, size_t ws_bytes, int32_t* table,
                       hipStream_t st) {
    PCREG_ARG(Q >= 0 && ldq >= Q && m_lo >= 0 && M_total >= 0);
    if (Q == 0) return PCREG_OK;
    if (v.M == 0) { PCREG_HIP(hipMemsetAsync(table, 0, sizeof(int32_t) * 4 * (size_t)Q, st)); return PCREG_OK; }
    size_t need; SearchWs s = search_ws_layout(Q, v.M, ws, &need);
    if (ws_bytes < need) { set_error("match workspace too small: %zu < %zu (pass the search call's workspace)", ws_bytes, need); return PCREG_E_WORKSPACE; }
    int blocks, cpb; { int rc = match_shape(Q, &blocks, &cpb); if (rc) return rc; }
    hipLaunchKernelGGL(match_finish_kernel<kMatchTable>, dim3(blocks), dim3(kMB), 0, st, q, Q, ldq, v.m, v.M, v.ldm, (int)m_lo, M_total, idx, dist, thr, ratio,
                       unique, (const UgPrep*)s.ug_prep, (const int32_t*)s.ug_cnt, (const float4*)s.ug_slots, table, (SearchCounters*)s.ctr, cpb,
                       (uint32_t*)nullptr, (double*)nullptr, (double*)nullptr, (int32_t*)nullptr);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}
int launch_match_from_table(const float* q, int Q, int ldq, int M_total, const int32_t* idx, const float* dist, float thr,
                            float ratio, const int32_t* table, void* ws, size_t ws_bytes, uint32_t* pairs, double* pts1,
                            double* pts2, int32_t* n_pairs, hipStream_t st) {
    PCREG_ARG(Q >= 0 && ldq >= Q && (pts1 == nullptr) == (pts2 == nullptr));
    if (Q == 0 || M_total == 0) { PCREG_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), st)); return PCREG_OK; }
    PCREG_ARG(ws != nullptr && ws_bytes >= align_up(sizeof(SearchCounters), 256));       // the counters at the head of the search workspace
    int blocks, cpb; { int rc = match_shape(Q, &blocks, &cpb); if (rc) return rc; }
    hipLaunchKernelGGL(match_finish_kernel<kMatchFromTable>, dim3(blocks), dim3(kMB), 0, st, q, Q, ldq, (const float*)nullptr, 0, 0, 0, M_total, idx, dist, thr,
                       ratio, 1, (const UgPrep*)nullptr, (const int32_t*)nullptr, (const float4*)nullptr, const_cast<int32_t*>(table), (SearchCounters*)ws, cpb,
                       pairs, pts1, pts2, n_pairs);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
