// pcreg_amd/csrc/wave_math.hpp -- wave-level fp64 helpers shared by the support kernels (align.hip, descriptors.hip).
//
// wave_sum_dpp: sum of a double over the 64 lanes WITHOUT the LDS crossbar.  `__shfl_xor` on a double is two
// ds_bpermute_b32 + a wait per step; six dependent steps cost ~2-3 k cycles for a handful of values, and the support
// kernels are chains of exactly such steps.  Here the four in-row steps are DPP moves (xor 1, xor 2, half-mirror,
// mirror: after each step the partners hold the same bits, fp addition being commutative) and the four row totals
// are read with v_readlane and added in a fixed order, so every lane returns the same bits.
//
// jacobi_sym3: cyclic Jacobi on the six unique entries of a symmetric 3 x 3 matrix.  One thread runs it while its
// workgroup waits, so its LATENCY is what matters: per rotation one division, one square root and one reciprocal
// square root (t = sgn * |h| / (|d| + sqrt(d^2 + h^2)) with d = a_qq - a_pp, h = 2 a_pq is the textbook
// t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = d / h multiplied through by |h|), and the closed-form
// update of the symmetric matrix (a_pp -= t a_pq, a_qq += t a_pq, two entries rotated) instead of two 3 x 3 products.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>

namespace pcreg {

template <int CTRL>
__device__ __forceinline__ double dpp_partner_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane of these permutations has a source lane: no `old` operand, so no register copy in front of the v_mov_b32_dpp
    const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_partner_f64<0xB1>(v);        // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_partner_f64<0x4E>(v);        // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_partner_f64<0x141>(v);       // row_half_mirror: the other quad of the 8
    v += dpp_partner_f64<0x140>(v);       // row_mirror: the other half of the 16
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return ((r0 + r1) + r2) + r3;
}

// min / max of NON-NEGATIVE doubles (their order is that of their bit patterns); same structure
__device__ __forceinline__ double wave_min_dpp(double v) {
    v = fmin(v, dpp_partner_f64<0xB1>(v)); v = fmin(v, dpp_partner_f64<0x4E>(v));
    v = fmin(v, dpp_partner_f64<0x141>(v)); v = fmin(v, dpp_partner_f64<0x140>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return fmin(fmin(r0, r1), fmin(r2, r3));
}
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_partner_f64<0xB1>(v)); v = fmax(v, dpp_partner_f64<0x4E>(v));
    v = fmax(v, dpp_partner_f64<0x141>(v)); v = fmax(v, dpp_partner_f64<0x140>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

__device__ __forceinline__ int wave_sum_dpp_i(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// inclusive prefix sum over the 64 lanes: four row_shr steps inside each row of 16, then row_bcast:15 / row_bcast:31 carry the
// row totals (lanes without a source lane add the `old` operand, 0) -- six DPP adds instead of six ds_bpermute round trips
__device__ __forceinline__ int wave_scan_incl_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return v;
}

// a = {a00, a01, a02, a11, a12, a22} (in/out: the diagonal ends up in a[0], a[3], a[5]); V row-major 3 x 3, columns =
// eigenvectors.  V may live in LDS: its addresses are compile-time constants here.
__device__ __forceinline__ void jacobi_sym3(double (&a)[6], double* V) {
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = (i % 4) == 0 ? 1.0 : 0.0;
#pragma nounroll
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(a[1]) + fabs(a[2]) + fabs(a[4]);
        const double dia = fabs(a[0]) + fabs(a[3]) + fabs(a[5]);
        if (off <= 1e-300 || off <= DBL_EPSILON * 1e-3 * dia) break;
        // rotation in the (P, Q) plane; R is the third index.  APP/AQQ/APQ/ARP/ARQ name the entries of `a`.
#define PCREG_SYM_ROT(APP, AQQ, APQ, ARP, ARQ, P, Q)                                               \
        if (a[APQ] != 0.0) {                                                                       \
            const double h = 2.0 * a[APQ], d = a[AQQ] - a[APP];                                    \
            const double t = copysign(fabs(h), h * d >= 0.0 ? 1.0 : -1.0) / (fabs(d) + sqrt(d * d + h * h)); \
            const double c = rsqrt(t * t + 1.0), s = c * t;                                        \
            const double tp = t * a[APQ];                                                          \
            a[APP] -= tp; a[AQQ] += tp; a[APQ] = 0.0;                                              \
            const double rp = a[ARP], rq = a[ARQ];                                                 \
            a[ARP] = c * rp - s * rq; a[ARQ] = s * rp + c * rq;                                    \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                        \
                const double vp = V[k * 3 + P], vq = V[k * 3 + Q];                                 \
                V[k * 3 + P] = c * vp - s * vq; V[k * 3 + Q] = s * vp + c * vq;                    \
            }                                                                                      \
        }
        PCREG_SYM_ROT(0, 3, 1, 2, 4, 0, 1)      // (0,1): third index 2 -> a02, a12
        PCREG_SYM_ROT(0, 5, 2, 1, 4, 0, 2)      // (0,2): third index 1 -> a01, a12
        PCREG_SYM_ROT(3, 5, 4, 1, 2, 1, 2)      // (1,2): third index 0 -> a01, a02
#undef PCREG_SYM_ROT
    }
}

}  // namespace pcreg
