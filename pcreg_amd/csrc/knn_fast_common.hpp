// pcreg_amd/csrc/knn_fast_common.hpp -- device helpers shared by knn_fast.hip and knn_mfma.hip
#pragma once
#include "common.hpp"
#include <cmath>

namespace pcreg {
namespace {

constexpr int kBlock = 256;
constexpr int KC = 4;                        // candidates kept per query and chunk
constexpr int kMTile = 1024;                 // model points per LDS tile (16 KiB)

struct Prep { float cx, cy, cz, rm2; };      // centre and max |m~|^2, produced on device

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


}  // namespace
}  // namespace pcreg
