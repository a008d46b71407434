// pcreg_amd/csrc/knn_fast_common.hpp -- device helpers shared by knn_fast.hip and knn_mfma.hip
#pragma once
#include "common.hpp"
#include <cmath>

namespace pcreg {
namespace {

constexpr int kBlock = 256;
constexpr int KC = 4;                        // candidates kept per query and chunk
constexpr int kMTile = 1024;                 // model points per LDS tile (16 KiB)

// centre (and, for the f16 matrix-core path, the power-of-two scale that brings the half extent of the
// joint bounding box into [32, 64) and 1/scale^2), produced on device
struct Prep {
    float cx, cy, cz, rm2, sigma, inv_sigma2, pad0, pad1;
    float gx0, gy0, gz0, inv_h;                  // seeding grid over the joint bounding box (knn_fast.hip, stage 1c)
    int nx, ny, nz, ncell;
};
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

// ---- 2. candidate generation -------------------------------------------------------------
struct Cand { float s[KC]; int i[KC]; };

__device__ __forceinline__ void cand_insert(Cand& c, float s, int j) {
    // keep c.s ascending; strict '<' so that, within a lane, earlier (lower) indices win ties
    if (s < c.s[3]) {
        if (s < c.s[1]) {
            c.s[3] = c.s[2]; c.i[3] = c.i[2];
            c.s[2] = c.s[1]; c.i[2] = c.i[1];
            if (s < c.s[0]) { c.s[1] = c.s[0]; c.i[1] = c.i[0]; c.s[0] = s; c.i[0] = j; }
            else { c.s[1] = s; c.i[1] = j; }
        } else {
            if (s < c.s[2]) { c.s[3] = c.s[2]; c.i[3] = c.i[2]; c.s[2] = s; c.i[2] = j; }
            else { c.s[3] = s; c.i[3] = j; }
        }
    }
}


// The same insertion without a branch: a four-stage compare-exchange chain that carries (score, index) down the sorted
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
