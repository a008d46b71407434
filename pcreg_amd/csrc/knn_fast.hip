// pcreg_amd/csrc/knn_fast.hip -- certified fast path of the fp32 3-D point search against a PREPARED model.
//
// Same contract as knn2_points_kernel (knn_points.hip): for every query the two nearest model points under
// d = fmaf(dz,dz, fmaf(dy,dy, dx*dx)), dx = q - m, ties to the lowest index -- the bits the oracle produces.
//
// The reference matches MANY surfaces against ONE model (completeExperimentFast.m:131-149,201-216), so everything that
// depends on the model alone is done once (model_prepare):
//   P1  model_bbox_partial/final_kernel   box of the model -> centre c, power-of-two scale sigma, seeding-grid geometry
//   P2  prep_model_f16_kernel             model -> tiles of f16 matrix-core operands, R_m^2, and the seeding grid's cells
// and a search is FOUR launches (round 2: eleven):
//   S1  seed_query_kernel    a first threshold per query from the model-wide seeding grid; clears the call's counters
//                            and the candidate lists; per-workgroup boxes of the queries (query grid, below)
//   S2  knn_candidates_f16_pipe_kernel (knn_mfma16.hip)   all Q x M scores on the matrix cores + selection
//   S3  knn_finalize_kernel  one 8-lane group per query: exact fmaf-chain distances of the listed candidates,
//                            (distance, index) top-2, and the CERTIFICATE: every point outside the lists has
//                            s >= G (the final threshold word), hence exact d >= G + |q~|^2 - E; proven answers are
//                            written, the others are listed; fills the query grid
//   S4  knn_tail_kernel      the listed queries again, exactly: slices of the model per query when few, the tiled
//                            all-pairs form when many; per-query / per-tile arrival counters let the last workgroup
//                            merge, so there is no second launch.  Idle (one read) when the list is empty.
// By-product: a uniform grid over the QUERIES (boxes in S1, geometry by a surplus workgroup of S2, cells in S3), which
// the Unique back-check of the match stage walks (knn_points.hip: match_finish_kernel) -- no launches of its own.
//
// Rounding bound (u = 2^-24, R_m = max|m~|, r = |q~|), derived in DESIGN.md section 4.1:
//   E = u*(3 R_m^2 + 16 r R_m + 32.1 (R_m^2 + 2 r R_m) + 4.04 (r + R_m)^2) + 16 u (d2 + |G + r^2|)
#include "common.hpp"
#include "select.hpp"
#include "knn_fast_common.hpp"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <algorithm>

namespace pcreg {

size_t knn_f16_prep_bytes(int M);
void knn_f16_shape(int Q, int M, int target_blocks, int* q_blocks, int* S, int* tiles_per_chunk);
int launch_prep_model_f16(const float* m, int M, int ldm, const void* prep, unsigned* rm2, void* mtiles, int32_t* seed_cnt,
                          void* seed_slots, hipStream_t st);
int launch_knn_candidates_f16(const float* q, int Q, int ldq, int M, const void* prep, const void* mtiles, unsigned* gthr,
                              void* cand_ent, int32_t* cand_cnt, int target_blocks, bool dry, bool timed, const float* ug_part,
                              int ug_nparts, int ug_cells, void* ug_prep, int* S_out, hipStream_t st);

namespace {

// ---- P1. bounding box of the model (two-stage, deterministic) ----------------------------------------------------
__global__ __launch_bounds__(kBlock) void model_bbox_partial_kernel(const float* __restrict__ m, int M, int ldm,
                                                                    float* __restrict__ part /*[grid][6]: lo, hi*/,
                                                                    int32_t* __restrict__ zero_me, int n_zero) {
    // the seeding grid's cell counters are cleared here (saves a memset; nothing reads them before the fill)
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_zero; i += gridDim.x * kBlock) zero_me[i] = 0;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < M; i += gridDim.x * kBlock) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float v = m[i + (size_t)c * ldm]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
    }
    __shared__ float s[kBlock / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = lo[c]; s[threadIdx.x >> 6][3 + c] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void model_bbox_final_kernel(const float* __restrict__ part, int nparts, int M, int cell_cap, Prep* __restrict__ prep,
                                        unsigned* __restrict__ rm2_bits) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < nparts; b += 64)              // launched with one wave
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], part[b * 6 + c]); hi[c] = fmaxf(hi[c], part[b * 6 + 3 + c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    }
    if (threadIdx.x == 0) {
        prep->cx = 0.5f * lo[0] + 0.5f * hi[0]; prep->cy = 0.5f * lo[1] + 0.5f * hi[1]; prep->cz = 0.5f * lo[2] + 0.5f * hi[2];
        {   // an upper bound of max |m~|^2 from the box itself: the seeding margin uses it
            float ax = 0.5f * (hi[0] - lo[0]), ay = 0.5f * (hi[1] - lo[1]), az = 0.5f * (hi[2] - lo[2]);
            prep->rm2 = (ax * ax + ay * ay + az * az) * 1.0001f;
        }
        float H = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) * 0.5f;
        float sg = 1.0f;
        if (H > 0.0f && H < INFINITY) sg = ldexpf(1.0f, 5 - ilogbf(H));        // sigma * H in [32, 64)
        prep->sigma = sg; prep->inv_sigma2 = 1.0f / (sg * sg);                 // powers of two: exact
        prep->pad0 = prep->pad1 = 0.0f;
        // seeding grid: cells of about two model points over the model's box plus one cell of margin on every side (a
        // query outside looks at the border cells); at most cell_cap cells
        float ext[3], emax = 0.0f;
        for (int c = 0; c < 3; ++c) { ext[c] = hi[c] - lo[c]; emax = fmaxf(emax, ext[c]); }
        int n[3] = {1, 1, 1};
        float h = 1.0f, g0[3] = {lo[0], lo[1], lo[2]};
        if (emax > 0.0f && emax < INFINITY) {
            float e2[3];
            for (int c = 0; c < 3; ++c) e2[c] = fmaxf(ext[c], emax * 1e-3f);
            h = cbrtf(e2[0] * e2[1] * e2[2] / fmaxf((float)M * 0.5f, 1.0f));
            for (int it = 0; it < 64; ++it) {
                long tot = 1;
                for (int c = 0; c < 3; ++c) {
                    g0[c] = lo[c] - h;
                    n[c] = (int)fminf(ceilf(ext[c] / h) + 2.0f, 2048.0f); if (n[c] < 1) n[c] = 1; tot *= n[c];
                }
                if (tot <= cell_cap) break;
                h *= 1.2f;
            }
            if ((long)n[0] * n[1] * n[2] > cell_cap) { n[0] = n[1] = n[2] = 1; g0[0] = lo[0]; g0[1] = lo[1]; g0[2] = lo[2]; h = emax * 2.0f; }
        }
        prep->gx0 = g0[0]; prep->gy0 = g0[1]; prep->gz0 = g0[2]; prep->inv_h = 1.0f / h;
        prep->nx = n[0]; prep->ny = n[1]; prep->nz = n[2]; prep->ncell = n[0] * n[1] * n[2];
        *rm2_bits = 0u;
    }
}

// ---- S1. seeding: a first threshold per query from the model-wide grid ------------------------------------------
// The candidate kernel only touches its sorted lists when a score beats the query's threshold, and a wave pays that
// slow path whenever ANY of its lanes does.  Starting from +inf every lane does so O(log n) times; starting from "the
// k-th nearest of a few model points around the query" (k = kSeedRank = 2) almost never.  The grid remembers up to kSeedSlots points per
// cell (whoever arrived first: the threshold is a hint, results never depend on it); a query looks at the 27 cells
// around its own (clamped into the grid: any model point's exact distance is a valid upper bound), takes the
// k-th smallest EXACT distance dk and publishes the s-space threshold dk - |q~|^2 plus twice the score error bound, so
// that its true k nearest are still below it.  Every point that is later skipped was compared with a word >= the
// final word G, which is all the certificate needs.
__device__ __forceinline__ int seed_cell(float v, float lo, float inv_h, int n) {
    int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
// eight lanes per query: lane k of the group looks at cells k, k + 8, k + 16, k + 24 of the 27, keeps its
// own sorted four smallest distances, and three xor-shuffle rounds merge the eight lists
__device__ __forceinline__ void sort4(float (&d)[4]) {
#define PCREG_CS(a, b) { const float lo_ = fminf(d[a], d[b]), hi_ = fmaxf(d[a], d[b]); d[a] = lo_; d[b] = hi_; }
    PCREG_CS(0, 1) PCREG_CS(2, 3) PCREG_CS(0, 2) PCREG_CS(1, 3) PCREG_CS(1, 2)
#undef PCREG_CS
}
__global__ __launch_bounds__(kBlock) void seed_query_kernel(const float* __restrict__ q, int Q, int ldq,
                                                            const Prep* __restrict__ prep, const int32_t* __restrict__ cnt,
                                                            const float4* __restrict__ slots, int seeded, unsigned* __restrict__ gthr,
                                                            int32_t* __restrict__ cand_cnt, SearchCounters* __restrict__ ctr,
                                                            float* __restrict__ ug_part, int32_t* __restrict__ ug_cnt, int ug_cells) {
    // housekeeping for the launches that follow
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < (int)(sizeof(SearchCounters) / 4); i += gridDim.x * kBlock) ((int32_t*)ctr)[i] = 0;
    if (ug_cnt) for (int i = blockIdx.x * kBlock + threadIdx.x; i < ug_cells; i += gridDim.x * kBlock) ug_cnt[i] = 0;
    const int qi = (blockIdx.x * kBlock + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool live = qi < Q;
    const int qq = live ? qi : 0;
    const float qx = q[qq], qy = q[qq + (size_t)ldq], qz = q[qq + 2 * (size_t)ldq];
    if (ug_part) {                  // box of this workgroup's queries (the query grid's geometry follows from all of them)
        __shared__ float s_box[kBlock / 64][6];
        float v[6] = {live ? qx : INFINITY, live ? qy : INFINITY, live ? qz : INFINITY, live ? qx : -INFINITY, live ? qy : -INFINITY, live ? qz : -INFINITY};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { v[c] = fminf(v[c], __shfl_xor(v[c], o)); v[3 + c] = fmaxf(v[3 + c], __shfl_xor(v[3 + c], o)); }
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) s_box[threadIdx.x >> 6][c] = v[c];
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            float r = s_box[0][threadIdx.x];
            for (int w = 1; w < kBlock / 64; ++w) r = threadIdx.x < 3 ? fminf(r, s_box[w][threadIdx.x]) : fmaxf(r, s_box[w][threadIdx.x]);
            ug_part[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
        }
    }
    if (live && sub == 0) cand_cnt[qi] = 0;                   // the query's candidate list starts empty
    if (!seeded) { if (live && sub == 0) gthr[qi] = 0xFFFFFFFFu; return; }       // +inf: no hint
    const int nx = prep->nx, ny = prep->ny, nz = prep->nz;
    const int cx = seed_cell(qx, prep->gx0, prep->inv_h, nx), cy = seed_cell(qy, prep->gy0, prep->inv_h, ny), cz = seed_cell(qz, prep->gz0, prep->inv_h, nz);
    float d[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
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
    // The k-th smallest exact distance of the sample bounds the true k-th nearest distance from above; any rank >= 2 keeps
    // both true neighbours under the threshold (kSeedRank = 2 since round 3).
    const float dk = d[kSeedRank - 1];
    if (dk < INFINITY) {
        const float tx = qx - prep->cx, ty = qy - prep->cy, tz = qz - prep->cz;
        const double r2 = (double)tx * tx + (double)ty * ty + (double)tz * tz;
        const double E = score_error_bound(1, (double)prep->rm2, sqrt(r2));
        const double u = 5.9604644775390625e-08;
        const double t = (double)dk - r2 + 2.0 * E + 16.0 * u * ((double)dk + fabs((double)dk - r2));
        word = f2ord(nextafterf((float)t, INFINITY));
    }
    gthr[qi] = word;
}

// ---- S3. exact re-rank + certificate: one 8-lane group per query --------------------------------------------------
__device__ __forceinline__ bool lex_lt_f(float da, int ia, float db, int ib) {
    return da < db || (da == db && (unsigned)ia < (unsigned)ib);
}
// A list entry is (first row jb, minimum score) of the 16 model points jb + 8 (r / 4) + r % 4, r = 0..15, that one lane
// of knn_candidates_f16_pipe_kernel scored together; pass 2 expands every entry that can still matter.  Points past M
// (tile padding) are skipped.  A query's list: cand_cnt[qi] entries at ent[(size_t)qi * cap + e].
constexpr int LPQ = 8;
__global__ __launch_bounds__(kBlock) void knn_finalize_kernel(
    const float* __restrict__ q, int Q, int ldq, const float* __restrict__ m, int M, int ldm,
    const Prep* __restrict__ prep, const unsigned* __restrict__ rm2_bits, const unsigned* __restrict__ gthr,
    const uint2* __restrict__ ent_all, const int32_t* __restrict__ cand_cnt, int cap, int idx_base,
    int32_t* __restrict__ idx, float* __restrict__ dist, int32_t* __restrict__ flag_list, int32_t* __restrict__ n_flag,
    const UgPrep* __restrict__ ug_prep, int32_t* __restrict__ ug_cnt, float4* __restrict__ ug_slots) {
    const int lane = threadIdx.x & (LPQ - 1);
    const int qi = blockIdx.x * (kBlock / LPQ) + (threadIdx.x / LPQ);
    if (qi >= Q) return;
    const float qx = q[qi], qy = q[qi + (size_t)ldq], qz = q[qi + 2 * (size_t)ldq];
    if (ug_prep && lane == 0) {                // the query grid of the Unique back-check (knn_points.hip): this query's cell
        const UgPrep P = *ug_prep;
        const int cell = (ug_cell1(qz, P.z0, P.inv_c, P.nz) * P.ny + ug_cell1(qy, P.y0, P.inv_c, P.ny)) * P.nx + ug_cell1(qx, P.x0, P.inv_c, P.nx);
        const int s = atomicAdd(&ug_cnt[cell], 1);
        if (s < kUgSlots) ug_slots[(size_t)cell * kUgSlots + s] = make_float4(qx, qy, qz, __int_as_float(qi));
    }
    const int total = min(cand_cnt[qi], cap);
    const uint2* ent = ent_all + (size_t)qi * cap;
    // pass 1: the two smallest approximate scores of the union (values only)
    float a1 = INFINITY, a2 = INFINITY;
    for (int e = lane; e < total; e += LPQ) {
        const uint2 v = ent[e];
        const float s = __uint_as_float(v.y);
        if ((int)v.x >= 0) { if (s < a2) { if (s < a1) { a2 = a1; a1 = s; } else a2 = s; } }
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
    const float cxf = prep->cx, cyf = prep->cy, czf = prep->cz, sg = prep->sigma;
    const float tx = qx - cxf, ty = qy - cyf, tz = qz - czf;       // the same q~ the candidate kernel formed
    const bool scored = fabsf(sg * tx) <= kQueryScaledMax && fabsf(sg * ty) <= kQueryScaledMax && fabsf(sg * tz) <= kQueryScaledMax;
    const double r2 = (double)tx * tx + (double)ty * ty + (double)tz * tz;
    const double r = sqrt(r2);
    const double Rm2 = (double)__uint_as_float(*rm2_bits);
    // scores from the f16-split matrix-core product (knn_mfma16.hip): 16 u r R_m for the two-term f16 representations
    // of Q and m (2^-22 relative each) and 32.1 u (R_m^2 + 2 r R_m) for sixteen fp32 accumulation steps of at most
    // one ulp each.
    const double Eab = score_error_bound(1, Rm2, r);
    // pass 2: exact distances of every candidate that can still reach the top-2
    const float cut = (float)((double)a2 + 2.0 * Eab + 16.0 * u * fabs((double)a2 + r2));
    const float cut_up = nextafterf(cut, INFINITY);              // float rounding of the cut must not exclude anything
    float d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
    for (int e = lane; e < total; e += LPQ) {
        const uint2 v = ent[e];
        const int j = (int)v.x; const float sc = __uint_as_float(v.y);
        if (j >= 0 && (sc <= cut_up || !(a2 < INFINITY))) {
            for (int rr = 0; rr < 16; ++rr) {
                const int jj = j + 8 * (rr >> 2) + (rr & 3);
                if (jj >= M) continue;
                float dx = qx - m[jj], dy = qy - m[jj + (size_t)ldm], dz = qz - m[jj + 2 * (size_t)ldm];
                float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (lex_lt_f(d, jj, d2, i2)) {
                    if (lex_lt_f(d, jj, d1, i1)) { d2 = d1; i2 = i1; d1 = d; i1 = jj; } else { d2 = d; i2 = jj; }
                }
            }
        }
    }
    // group-shuffle top-2 reduction ordered by (dist, idx); empty slots are (+inf, -1 -> max uint)
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
    if (!scored) ok = false;                                     // never scored on the matrix cores: redo exactly
    else if (M <= 2) ok = (i1 >= 0) && (M < 2 || i2 >= 0);       // nothing outside the lists when M <= KC
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

// ---- S4. the unproven queries again, exactly: ONE launch --------------------------------------------------------
// few (<= kFew): work item = (listed query, slice of the model); all threads of a workgroup stride over the slice.
// many: work item = (tile of kTailQ listed queries, chunk of the model); a lane owns four queries, the chunk streams
// through LDS -- knn2_points_kernel's loop (6 VALU per pair, the oracle's bits).
// Either way the workgroup that delivers the LAST partial of a query / tile (an arrival counter, cleared by S1) merges
// the partials by (distance, index) and writes the result.  Partials cross workgroups through device-coherent accesses
// (relaxed agent-scope atomic stores / loads) and a relaxed counter: no agent-scope fence (it would write back / invalidate
// the XCD's L2).
constexpr int kFew = 1024, kFbSlices = 32;
constexpr int kTailGrid = 2048, kTailQ = 4 * kBlock;              // tiled form: 1024 queries per tile
struct Top2 { float d1, d2; int i1, i2; };
__device__ __forceinline__ void top2_insert(Top2& t, float d, int j) {
    // candidates arrive in ascending j, so strict '<' keeps the lowest index
    if (d < t.d2) {
        if (d < t.d1) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = j; }
        else { t.d2 = d; t.i2 = j; }
    }
}
__device__ __forceinline__ void top2_wave_merge(float& d1, float& d2, int& i1, int& i2) {
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
}
__global__ __launch_bounds__(kBlock) void knn_tail_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ m, int M, int ldm, int idx_base,
    const int32_t* __restrict__ flag_list, SearchCounters* __restrict__ ctr,
    int32_t* __restrict__ part_idx, float* __restrict__ part_dist /* few: [kFew][kFbSlices][2]; many: [S][tiles * kTailQ][2] */,
    int32_t* __restrict__ idx, float* __restrict__ dist) {
    const int nf = ctr->n_flag;
    if (nf <= 0) return;
    __shared__ float sd[kBlock / 64][2];
    __shared__ int si[kBlock / 64][2];
    __shared__ int s_last;
    __shared__ float4 tile[kMTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (nf <= kFew) {
        const int len = (M + kFbSlices - 1) / kFbSlices;
        for (int w = blockIdx.x; w < nf * kFbSlices; w += gridDim.x) {
            const int f = w / kFbSlices, sl = w % kFbSlices;
            const int j0 = sl * len, j1 = min(M, j0 + len);
            const int qi = flag_list[f];
            const float qx = q[qi], qy = q[qi + (size_t)ldq], qz = q[qi + 2 * (size_t)ldq];
            float d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
            for (int j = j0 + threadIdx.x; j < j1; j += kBlock) {     // ascending j per thread: strict '<' keeps ties low
                float dx = qx - m[j], dy = qy - m[j + (size_t)ldm], dz = qz - m[j + 2 * (size_t)ldm];
                float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (d < d2) { if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j; } else { d2 = d; i2 = j; } }
            }
            top2_wave_merge(d1, d2, i1, i2);
            __syncthreads();                                          // the previous trip's readers are done with sd / si
            if (lane == 0) { sd[wave][0] = d1; sd[wave][1] = d2; si[wave][0] = i1; si[wave][1] = i2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                Top2T<float> t{INFINITY, INFINITY, -1, -1};
                for (int k = 0; k < kBlock / 64; ++k) { top2_insert_lex_t(t, sd[k][0], si[k][0]); top2_insert_lex_t(t, sd[k][1], si[k][1]); }
                const size_t o = ((size_t)f * kFbSlices + sl) * 2;
                __hip_atomic_store(&part_idx[o], t.i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part_idx[o + 1], t.i2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part_dist[o], t.d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part_dist[o + 1], t.d2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the four stores above are complete (and device-coherent)
                const int old = __hip_atomic_fetch_add(&ctr->done[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == kFbSlices - 1) {                            // every slice of this query is in: merge
                    Top2T<float> r{INFINITY, INFINITY, -1, -1};
                    for (int s2 = 0; s2 < kFbSlices; ++s2) {
                        const size_t p = ((size_t)f * kFbSlices + s2) * 2;
                        const int a = __hip_atomic_load(&part_idx[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int b = __hip_atomic_load(&part_idx[p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float da = __hip_atomic_load(&part_dist[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float db = __hip_atomic_load(&part_dist[p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        top2_insert_lex_t(r, da, a); top2_insert_lex_t(r, db, b);
                    }
                    idx[(size_t)qi * 2] = r.i1 >= 0 ? r.i1 + idx_base : -1; idx[(size_t)qi * 2 + 1] = r.i2 >= 0 ? r.i2 + idx_base : -1;
                    dist[(size_t)qi * 2] = r.d1; dist[(size_t)qi * 2 + 1] = r.d2;
                }
            }
        }
        return;
    }
    // many: tiles of kTailQ listed queries x S chunks of the model
    const int n_qt = (nf + kTailQ - 1) / kTailQ;
    const int max_S = (M + kMTile - 1) / kMTile;
    int S = kTailGrid / n_qt; if (S < 1) S = 1; if (S > max_S) S = max_S; if (S < 1) S = 1;
    const int chunk = ((M + S - 1) / S + kMTile - 1) / kMTile * kMTile;
    S = M > 0 ? (M + chunk - 1) / chunk : 1;
    const size_t slots_cap = (size_t)n_qt * kTailQ;
    for (int w = blockIdx.x; w < n_qt * S; w += gridDim.x) {
        const int qt = w / S, s = w % S;
        const int m_begin = s * chunk, m_end = min(M, m_begin + chunk);
        float qx[4], qy[4], qz[4]; Top2 best[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int slot = qt * kTailQ + r * kBlock + threadIdx.x;
            const bool ok = slot < nf;
            const int qi = ok ? flag_list[slot] : 0;
            qx[r] = ok ? q[qi] : 0.0f; qy[r] = ok ? q[qi + (size_t)ldq] : 0.0f; qz[r] = ok ? q[qi + 2 * (size_t)ldq] : 0.0f;
            best[r] = Top2{INFINITY, INFINITY, -1, -1};
        }
        for (int t0 = m_begin; t0 < m_end; t0 += kMTile) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kMTile / kBlock; ++k) {
                const int j = t0 + k * kBlock + threadIdx.x;
                float4 v;
                if (j < m_end) { v.x = m[j]; v.y = m[j + (size_t)ldm]; v.z = m[j + 2 * (size_t)ldm]; v.w = 0.0f; }
                else { v.x = v.y = v.z = INFINITY; v.w = 0.0f; }     // padding never beats anything
                tile[k * kBlock + threadIdx.x] = v;
            }
            __syncthreads();
            const int cnt = min(kMTile, m_end - t0);
            const int nb = (cnt + 3) / 4 * 4;
            for (int jb = 0; jb < nb; jb += 4) {
                float4 mp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) mp[u] = tile[jb + u];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float d[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float dx = qx[r] - mp[u].x, dy = qy[r] - mp[u].y, dz = qz[r] - mp[u].z;
                        d[u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    }
                    const float mn = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
                    if (mn < best[r].d2) {
                        const int j0 = t0 + jb;
#pragma unroll
                        for (int u = 0; u < 4; ++u) top2_insert(best[r], d[u], j0 + u);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int slot = qt * kTailQ + r * kBlock + threadIdx.x;
            if (slot < nf) {
                const size_t o = ((size_t)s * slots_cap + slot) * 2;
                __hip_atomic_store(&part_idx[o], best[r].i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part_idx[o + 1], best[r].i2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part_dist[o], best[r].d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part_dist[o + 1], best[r].d2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&ctr->done[qt], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
        __syncthreads();
        if (s_last) {                         // every chunk of this tile is in: a wave per listed query merges its S partials
            const int s_hi = min(nf, (qt + 1) * kTailQ);
            for (int slot = qt * kTailQ + wave; slot < s_hi; slot += kBlock / 64) {
                float d1 = INFINITY, d2 = INFINITY; int i1 = -1, i2 = -1;
                for (int s2 = lane; s2 < S; s2 += 64) {           // chunks ascend in model index; (distance, index) order throughout
                    const size_t p = ((size_t)s2 * slots_cap + slot) * 2;
                    const int a = __hip_atomic_load(&part_idx[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int b = __hip_atomic_load(&part_idx[p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float da = __hip_atomic_load(&part_dist[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float db = __hip_atomic_load(&part_dist[p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a >= 0 && lex_lt_f(da, a, d2, i2)) { if (lex_lt_f(da, a, d1, i1)) { d2 = d1; i2 = i1; d1 = da; i1 = a; } else { d2 = da; i2 = a; } }
                    if (b >= 0 && lex_lt_f(db, b, d2, i2)) { if (lex_lt_f(db, b, d1, i1)) { d2 = d1; i2 = i1; d1 = db; i1 = b; } else { d2 = db; i2 = b; } }
                }
                top2_wave_merge(d1, d2, i1, i2);
                if (lane == 0) {
                    const int qi = flag_list[slot];
                    idx[(size_t)qi * 2] = i1 >= 0 ? i1 + idx_base : -1; idx[(size_t)qi * 2 + 1] = i2 >= 0 ? i2 + idx_base : -1;
                    dist[(size_t)qi * 2] = d1; dist[(size_t)qi * 2 + 1] = d2;
                }
            }
        }
    }
}

constexpr int kSeedMinM = 16 * 1024;             // below this the lists settle within the first tiles anyway
// the grid is sized on the device (about M/2 cells plus the margin layer); this is the capacity it may use
size_t seed_cell_cap(int M) { return std::min<size_t>((size_t)kSeedMaxCells, std::max<size_t>(4096, (size_t)M)); }

}  // namespace

// ---- a prepared model in device memory ---------------------------------------------------------------------------
// layout of the prepared block: Prep | R_m^2 bits | box partials | f16 tiles | seeding-grid counters | seeding-grid slots
size_t model_prep_bytes(int M) {
    size_t b = 256 + 256 + align_up(512 * 6 * sizeof(float), 256) + align_up(knn_f16_prep_bytes(M), 256);
    if (M >= kSeedMinM) b += align_up(seed_cell_cap(M) * 4, 256) + align_up(seed_cell_cap(M) * kSeedSlots * 16, 256);
    return b;
}
ModelView model_view(const float* m, int M, int ldm, void* block) {
    ModelView v{};
    char* w = (char*)block;
    v.m = m; v.M = M; v.ldm = ldm;
    v.prep = w; w += 256;
    v.rm2 = (unsigned*)w; w += 256;
    v.box_part = (float*)w; w += align_up(512 * 6 * sizeof(float), 256);
    v.tiles = w; w += align_up(knn_f16_prep_bytes(M), 256);
    v.seeded = M >= kSeedMinM && PCREG_EXP_ENV("PCREG_KNN_NOSEED", 0) == 0;
    if (M >= kSeedMinM) {
        v.seed_cnt = (int32_t*)w; w += align_up(seed_cell_cap(M) * 4, 256);
        v.seed_slots = w;
    }
    return v;
}
// enqueue the two preparation passes (P1, P2) on `st`
int launch_model_prepare(const ModelView& v, hipStream_t st) {
    PCREG_ARG(v.M >= 0 && v.ldm >= v.M);
    if (v.M == 0) return PCREG_OK;
    int nb = (v.M + kBlock * 16 - 1) / (kBlock * 16); if (nb > 512) nb = 512; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(model_bbox_partial_kernel, dim3(nb), dim3(kBlock), 0, st, v.m, v.M, v.ldm, v.box_part, v.seed_cnt,
                       v.seeded ? (int)seed_cell_cap(v.M) : 0);
    hipLaunchKernelGGL(model_bbox_final_kernel, dim3(1), dim3(64), 0, st, v.box_part, nb, v.M, (int)seed_cell_cap(v.M), (Prep*)v.prep, v.rm2);
    PCREG_HIP(hipGetLastError());
    return launch_prep_model_f16(v.m, v.M, v.ldm, v.prep, v.rm2, v.tiles, v.seeded ? v.seed_cnt : nullptr, v.seed_slots, st);
}

// ---- the per-call workspace of a search (and of the match stage that follows it) ----------------------------------
static constexpr int kTargetBlocks = 4096;
SearchWs search_ws_layout(int Q, int M, void* base, size_t* bytes) {
    SearchWs s{};
    const size_t qq = (size_t)(Q > 0 ? Q : 1);
    int q_blocks, S, tpc;
    knn_f16_shape(Q > 0 ? Q : 1, M > 0 ? M : 1, kTargetBlocks, &q_blocks, &S, &tpc);
    s.cap = S * KC;
    char* w = (char*)base;
    s.ctr = w; w += align_up(sizeof(SearchCounters), 256);
    s.gthr = (unsigned*)w; w += align_up(qq * 4, 256);
    s.flag_list = (int32_t*)w; w += align_up(qq * 4, 256);
    s.cand_cnt = (int32_t*)w; w += align_up(qq * 4, 256);
    s.cand_ent = w; w += align_up(qq * (size_t)(kF16MaxS * KC) * 8, 256);      // the call may see a smaller M than the sizing did
    const size_t few = (size_t)kFew * kFbSlices * 2, many = (size_t)(kTailGrid + (qq + kTailQ - 1) / kTailQ) * kTailQ * 2;
    s.tail_idx = (int32_t*)w; w += align_up(std::max(few, many) * 4, 256);
    s.tail_dist = (float*)w; w += align_up(std::max(few, many) * 4, 256);
    s.ug_cells = (int)ug_cells_cap(Q);
    s.ug_nparts = (int)((qq * 8 + kBlock - 1) / kBlock);
    s.ug_prep = w; w += 256;
    s.ug_part = (float*)w; w += align_up((size_t)s.ug_nparts * 6 * 4, 256);
    s.ug_cnt = (int32_t*)w; w += align_up((size_t)s.ug_cells * 4, 256);
    s.ug_slots = w; w += align_up((size_t)s.ug_cells * kUgSlots * 16, 256);
    *bytes = (size_t)(w - (char*)base);
    return s;
}
size_t search_ws_bytes(int Q, int M) { size_t b; (void)search_ws_layout(Q, M, nullptr, &b); return b; }

// S1..S4 against a prepared model.  with_grid: also build the query grid (the match stage with Unique needs it).
int launch_model_search(const ModelView& v, const float* q, int Q, int ldq, int32_t idx_base, int32_t* idx, float* dist,
                        void* ws, size_t ws_bytes, bool with_grid, bool timed, hipStream_t st) {
    PCREG_ARG(Q >= 0 && ldq >= Q && Q <= kMaxQTiles * 1024);
    if (Q == 0) return PCREG_OK;
    size_t need;
    SearchWs s = search_ws_layout(Q, v.M, ws, &need);
    if (ws_bytes < need) { set_error("search workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    SearchCounters* ctr = (SearchCounters*)s.ctr;
    const bool grid = with_grid && v.M > 0;          // (an empty model matches nothing: the match stage never looks at the grid)
    hipLaunchKernelGGL(seed_query_kernel, dim3(s.ug_nparts), dim3(kBlock), 0, st, q, Q, ldq, (const Prep*)v.prep, (const int32_t*)v.seed_cnt,
                       (const float4*)v.seed_slots, (v.seeded && v.M > 0) ? 1 : 0, s.gthr, s.cand_cnt, ctr, grid ? s.ug_part : nullptr,
                       grid ? s.ug_cnt : nullptr, s.ug_cells);
    int S = 1;
    const int variant = PCREG_EXP_ENV("PCREG_KNN_VARIANT", 40);      // 41: timing-only form of the candidate kernel (EXPERIMENTS builds)
    const int target_env = PCREG_EXP_ENV("PCREG_KNN_BLOCKS", 0);      // (any shape fits: the lists are sized for kF16MaxS chunks)
    int rc = launch_knn_candidates_f16(q, Q, ldq, v.M, v.prep, v.tiles, s.gthr, s.cand_ent, s.cand_cnt, target_env > 0 ? target_env : kTargetBlocks,
                                       variant == 41, timed, grid ? s.ug_part : nullptr, s.ug_nparts, s.ug_cells, grid ? s.ug_prep : nullptr, &S, st);
    if (rc) return rc;
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((Q + kBlock / LPQ - 1) / (kBlock / LPQ)), dim3(kBlock), 0, st, q, Q, ldq, v.m, v.M, v.ldm,
                       (const Prep*)v.prep, (const unsigned*)v.rm2, (const unsigned*)s.gthr, (const uint2*)s.cand_ent, (const int32_t*)s.cand_cnt,
                       S * KC, (int)idx_base, idx, dist, s.flag_list, &ctr->n_flag, grid ? (const UgPrep*)s.ug_prep : nullptr,
                       s.ug_cnt, (float4*)s.ug_slots);
    if (PCREG_EXP_ENV("PCREG_KNN_DEBUG", 0)) {
        int32_t nf = 0;
        PCREG_HIP(hipMemcpyAsync(&nf, &ctr->n_flag, 4, hipMemcpyDeviceToHost, st)); PCREG_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "[pcreg] knn fast: Q=%d M=%d S=%d unproven=%d\n", Q, v.M, S, nf);
    }
    hipLaunchKernelGGL(knn_tail_kernel, dim3(kTailGrid), dim3(kBlock), 0, st, q, ldq, v.m, v.M, v.ldm, (int)idx_base, (const int32_t*)s.flag_list, ctr,
                       s.tail_idx, s.tail_dist, idx, dist);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

// ---- the search without a handle: prepare into the caller's workspace, then search ---------------------------------
size_t knn2_points_fast_workspace_bytes(int Q, int M) { return search_ws_bytes(Q, M) + model_prep_bytes(M); }

int launch_knn2_points_fast_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm, int32_t idx_base,
                                int32_t* idx, float* dist, void* ws, size_t ws_bytes, hipStream_t st, bool timed) {
    PCREG_ARG(Q >= 0 && M >= 0 && ldq >= Q && ldm >= M);
    if (Q == 0) return PCREG_OK;
    const size_t need = knn2_points_fast_workspace_bytes(Q, M);
    if (ws_bytes < need) { set_error("knn (fast) workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    const size_t sb = search_ws_bytes(Q, M);
    const ModelView v = model_view(m, M, ldm, (char*)ws + sb);
    int rc = launch_model_prepare(v, st);
    if (rc) return rc;
    return launch_model_search(v, q, Q, ldq, idx_base, idx, dist, ws, sb, false, timed, st);
}

}  // namespace pcreg
