// pcreg_amd/csrc/knn_fast_common.hpp -- device helpers shared by knn_fast.hip, knn_mfma16.hip and knn_points.hip
#pragma once
#include "common.hpp"
#include <cmath>

namespace pcreg {
namespace {

constexpr int kBlock = 256;
#ifndef PCREG_KC
#define PCREG_KC 4
#endif
constexpr int KC = PCREG_KC;                 // group entries kept per lane, query and chunk (>= 2: the two nearest points sit in the two best groups)
#ifndef PCREG_SEED_RANK
#define PCREG_SEED_RANK 2
#endif
constexpr int kSeedRank = PCREG_SEED_RANK;   // the first threshold: the kSeedRank-th smallest exact distance among the seeding grid's points (>= 2;
                                             // 2 is the tightest valid hint: -1.3 % on the candidates kernel against 4, interleaved A/B in round 3)
constexpr int kMTile = 1024;                 // model points per LDS tile (16 KiB)

// What a PREPARED MODEL carries besides its f16 tiles (model-only quantities: one model, many query sets --
// completeExperimentFast.m:131-149): the centre of the model's bounding box, the power-of-two scale that brings its
// half extent into [32, 64) (and 1/scale^2), a bound of max |m~|^2, and the geometry of the model-wide seeding grid.
struct Prep {
    float cx, cy, cz, rm2, sigma, inv_sigma2, pad0, pad1;
    float gx0, gy0, gz0, inv_h;                  // seeding grid over the model's bounding box plus one cell of margin
    int nx, ny, nz, ncell;
};
// A query whose scaled coordinate leaves this range is not scored on the matrix cores (its f16 operands would
// overflow or lose the error bound's assumptions): it goes to the exact fallback.
constexpr float kQueryScaledMax = 16384.0f;
constexpr int kF16MaxS = 80;                      // most model chunks the f16 candidate kernel is split in
constexpr int kSeedSlots = 4;                    // model points remembered per grid cell
constexpr int kSeedMaxCells = 1 << 21;

// rounding bound of the candidate scores, without the d2-dependent part
__device__ __forceinline__ double score_error_bound(int e_mode, double Rm2, double r) {
    const double u = 5.9604644775390625e-08, Rm = sqrt(Rm2);
    return e_mode == 0 ? u * (3.0 * Rm2 + 3.03 * (Rm2 + 2.0 * r * Rm) + 4.04 * (r + Rm) * (r + Rm))
                       : u * (3.0 * Rm2 + 16.0 * r * Rm + 32.1 * (Rm2 + 2.0 * r * Rm) + 4.04 * (r + Rm) * (r + Rm));
}

__device__ __forceinline__ unsigned f2ord(float f) {       // order-preserving float -> uint
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ---- uniform grid over the QUERIES (the Unique back-check, knn_points.hip): built as a by-product of the search call
constexpr int kUgSlots = 16;
constexpr int kUgMaxCells = 4 << 20;
constexpr int kUgMaxVisit = 343;
struct UgPrep { float x0, y0, z0, inv_c; int nx, ny, nz, pad; };
static inline size_t ug_cells_cap(int Q) { size_t c = 2 * (size_t)(Q > 0 ? Q : 1); if (c < 4096) c = 4096; if (c > (size_t)kUgMaxCells) c = kUgMaxCells; return c; }
__device__ __forceinline__ int ug_cell1(float x, float x0, float inv_c, int n) {
    const int c = (int)floorf((x - x0) * inv_c);                     // monotone in x: the range test relies on it
    return min(max(c, 0), n - 1);
}
__device__ __forceinline__ float ug_d2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;           // the search's exact formula (sign-symmetric)
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
// about two cells per query over the occupied extent; flat or degenerate axes get one layer
__device__ __forceinline__ void ug_make_prep(const float lo[3], const float hi[3], int Q, int cells_cap, UgPrep* prep) {
    float e[3];
    for (int c = 0; c < 3; ++c) e[c] = fmaxf(hi[c] - lo[c], 0.0f);
    const float emax = fmaxf(e[0], fmaxf(e[1], e[2]));
    float vol = 1.0f; int dims = 0;
    for (int c = 0; c < 3; ++c) if (e[c] > 1e-6f * emax) { vol *= e[c]; ++dims; }
    float cs = dims > 0 ? powf(vol / (2.0f * (float)Q), 1.0f / (float)dims) : 1.0f;
    if (!(cs > 0.0f) || !isfinite(cs)) cs = 1.0f;
    int nx = 1, ny = 1, nz = 1;
    for (int it = 0; it < 4096; ++it) {
        const float inv = 1.0f / cs;
        const float fx = floorf(e[0] * inv) + 1.0f, fy = floorf(e[1] * inv) + 1.0f, fz = floorf(e[2] * inv) + 1.0f;
        if (fx * fy * fz <= (float)cells_cap) { nx = (int)fx; ny = (int)fy; nz = (int)fz; break; }
        cs *= 1.08f;
    }
    prep->x0 = lo[0]; prep->y0 = lo[1]; prep->z0 = lo[2]; prep->inv_c = 1.0f / cs;
    prep->nx = nx; prep->ny = ny; prep->nz = nz; prep->pad = 0;
}

// the small counters of one search call (+ the match stage that follows it), cleared by seed_query_kernel
constexpr int kMaxQTiles = 4096;                 // tiles of 1024 queries: Q <= 4 Mi per call
constexpr int kMatchMaxBlocks = 2048;
struct SearchCounters {
    int32_t n_flag;                  // unproven queries (knn_finalize_kernel -> knn_tail_kernel)
    int32_t ticket;                  // match_finish_kernel's workgroup tickets
    int32_t finished;                // ... and how many of its workgroups are through (the last one clears ticket / status again)
    int32_t pad[61];
    int32_t done[kMaxQTiles];        // tail: arrivals per listed query (few) or per tile of listed queries (many)
    int32_t status[kMatchMaxBlocks]; // match_finish_kernel: per-workgroup kept counts (ready bit 31)
};

// ---- 2. candidate generation -------------------------------------------------------------
struct Cand { float s[KC]; int i[KC]; };

// Sorted insertion without a branch: a KC-stage compare-exchange chain that carries (score, index) down the sorted
// list; `pred` false (or s >= c.s[3]) leaves the list as it is.  Straight-line code for the rare list update of the
// pipelined f16 kernel: the nested branches of cand_insert cost a lone wave ~3 scalar branch latencies per level.
__device__ __forceinline__ void cand_insert_branchless(Cand& c, float s, int j, bool pred) {
    float x = pred ? s : INFINITY; int xi = j;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const bool lt = x < c.s[k];                 // strict: an earlier equal entry keeps its place
        const float lo = lt ? x : c.s[k], hi = lt ? c.s[k] : x;
        const int loi = lt ? xi : c.i[k], hii = lt ? c.i[k] : xi;
        c.s[k] = lo; c.i[k] = loi; x = hi; xi = hii;
    }
}

}  // namespace
}  // namespace pcreg
