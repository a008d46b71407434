// pcreg_amd/csrc/ransac.hip -- RANSAC rigid alignment on gfx950 (MI355X), fp64.
//
// Replaces, behind the same semantics, the interpreted loop of the reference:
//   ransac.m:40-66            hypothesis loop   -> ransac_hyp_kernel
//   ransac.m:69-98            winner / outputs  -> ransac_select_kernel
//   estimateTransform.m:8-71  SVD Procrustes    -> fit_3pt / fit_moments (per lane)
//   getInliersRANSAC.m:46-54  calcDists         -> score_pass (canonical FMA order)
//
// Mapping to the hardware (CDNA4, wave64):
//   * one wave owns `hpw` hypotheses.  The minimal-sample fits run one hypothesis PER
//     LANE (the 3x3 Jacobi SVD is lane-parallel, no cross-lane traffic); scoring then
//     walks the wave's hypotheses one pair at a time with ALL 64 lanes striding over
//     the correspondences, the transform broadcast with v_readlane, inliers counted
//     with v_cmp -> s_bcnt1 (ballot + popcount), i.e. "one hypothesis per wavefront".
//   * correspondences (n x 6 doubles) are staged once per workgroup into LDS as six
//     SoA columns (conflict-free ds_read_b64) when they fit, otherwise streamed from
//     L2; two hypotheses share every point load.
//   * the refit (estimateTransform on the inlier set) is a wave-reduced set of 27
//     moments per passing hypothesis followed by a second lane-parallel SVD.
// No data leaves the chip between the sample fit and the final inlier list.
#include "common.hpp"
#include <cfloat>
#include <cstdlib>

namespace pcreg {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int HB = 4;   // hypotheses sharing one pass over the points (2 until round 3: 4 halves the LDS reads per score, +4 % on batched cfg 1)

struct RansacArgs {
    const double* p1; const double* p2; int ld;
    const int32_t* offsets;     // B+1 (device) or null
    const int32_t* n_dev;       // device n for the single-registration resident path, or null
    int n_cap; int iters; int m;
    double thDist; double ratio; int refine; unsigned long long seed;
    const int32_t* sample_idx;  // [B*iters][m] 1-based or null
    int hpw;                    // hypotheses per wave (<= 64)
    int hyp0g;                  // global index of this launch's first hypothesis (hypotheses split over ranks)
    double* TF;                 // [B*iters][12]
    int32_t* cnt1; int32_t* cnt2; unsigned char* has;
    int n_hi;                   // ransac_hyp_kernel: registrations of up to n_hi correspondences fit the launch's dynamic LDS
    int n_lo;                   // this launch serves registrations of n_lo < n (<= n_hi from LDS) correspondences
    double* msc;                // ransac_hyp32_kernel: [B*iters][16] the fifteen refit sums of a hypothesis
};

// ---------------------------------------------------------------- lane utilities
__device__ __forceinline__ double rdlane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// wave_sum of 27 values at once, the same bits as 27 calls.  wave_sum's butterfly adds lane l and lane l ^ o for o = 32, 16, ..., 1;
// after every step the two partners hold the same bits (fp addition commutes), so only ONE of them needs to go on with a given
// value: at distance 32 the lower half-wave keeps values 0-13 and the upper one 14-26, each sending the other its remaining
// half, and so on down -- 14 + 7 + 4 + 2 + 1 + 1 = 29 exchanged doubles instead of 27 x 6 = 162.  Value k ends in the lanes whose
// bits 5..1 spell its path (k = 14 b5 + 7 b4 + 4 b3 + 2 b2 + b1) and is broadcast from there with v_readlane.
__device__ __forceinline__ void wave_sum27(double (&v)[27]) {
    const int lane = threadIdx.x & 63;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    double k1[14], k2[8], k3[4], k4[2];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const double hi = 14 + j < 27 ? v[14 + j < 27 ? 14 + j : 26] : 0.0;
        k1[j] = (b5 ? hi : v[j]) + __shfl_xor(b5 ? v[j] : hi, 32);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) k2[j] = (b4 ? k1[7 + j] : k1[j]) + __shfl_xor(b4 ? k1[j] : k1[7 + j], 16);
    k2[7] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) k3[j] = (b3 ? k2[4 + j] : k2[j]) + __shfl_xor(b3 ? k2[j] : k2[4 + j], 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) k4[j] = (b2 ? k3[2 + j] : k3[j]) + __shfl_xor(b2 ? k3[j] : k3[2 + j], 4);
    double k5 = (b1 ? k4[1] : k4[0]) + __shfl_xor(b1 ? k4[0] : k4[1], 2);
    k5 += __shfl_xor(k5, 1);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int c5 = k >= 14, r5 = k - 14 * c5, c4 = r5 >= 7, r4 = r5 - 7 * c4, c3 = r4 >= 4, r3 = r4 - 4 * c3, c2 = r3 >= 2, c1 = r3 - 2 * c2;
        v[k] = rdlane(k5, 32 * c5 + 16 * c4 + 8 * c3 + 4 * c2 + 2 * c1);
    }
}
__device__ __forceinline__ double ulp_at(double x) {   // MATLAB eps(x)
    x = fabs(x);
    return __longlong_as_double(__double_as_longlong(x) + 1) - x;
}
__device__ __forceinline__ int matlab_round_i(double x) {   // ransac.m:28
    return (int)(x >= 0 ? floor(x + 0.5) : -floor(-x + 0.5));
}
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------- 3x3 kernels (per lane)
// One Hestenes rotation orthogonalising columns P,Q of W (stored W[col][row]).
template <int P, int Q, bool WITH_V>
__device__ __forceinline__ bool hrot(double (&W)[3][3], double (&V)[3][3]) {
    double al = fma(W[P][2], W[P][2], fma(W[P][1], W[P][1], W[P][0] * W[P][0]));
    double be = fma(W[Q][2], W[Q][2], fma(W[Q][1], W[Q][1], W[Q][0] * W[Q][0]));
    double ga = fma(W[P][2], W[Q][2], fma(W[P][1], W[Q][1], W[P][0] * W[Q][0]));
    bool doit = (ga != 0.0) && (fabs(ga) > DBL_EPSILON * sqrt(al * be));
    if (doit) {
        double zeta = (be - al) / (2.0 * ga);
        double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
        double c = 1.0 / sqrt(fma(t, t, 1.0)), s = c * t;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double a = W[P][r], b = W[Q][r];
            W[P][r] = c * a - s * b; W[Q][r] = s * a + c * b;
        }
        if (WITH_V) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double a = V[r][P], b = V[r][Q];
                V[r][P] = c * a - s * b; V[r][Q] = s * a + c * b;
            }
        }
    }
    return doit;
}
template <bool WITH_V>
__device__ __forceinline__ void hestenes3(double (&W)[3][3], double (&V)[3][3]) {
    if (WITH_V) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool r0 = hrot<0, 1, WITH_V>(W, V);
        bool r1 = hrot<0, 2, WITH_V>(W, V);
        bool r2 = hrot<1, 2, WITH_V>(W, V);
        if (!(r0 | r1 | r2)) break;
    }
}

// rank(A) >= need for a raw 3x3 point matrix A[pt][coord] (estimateTransform.m:11).
// Cheap certificates first (|det| and 2x2 minors bound the small singular values from
// below); the exact MATLAB rule -- sigma > 3*eps(sigma_max) -- only when they fail.
__device__ bool rank3x3_at_least(const double (&A)[3][3], int need) {
    double fro2 = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) fro2 = fma(A[r][c], A[r][c], fro2);
    double c0x = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    double c0y = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    double c0z = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    if (need >= 3) {
        double det = fma(A[0][0], c0x, fma(A[0][1], c0y, A[0][2] * c0z));
        double fro = sqrt(fro2);
        if (fabs(det) > 1e-12 * fro2 * fro) return true;       // sigma3 >= 2|det|/fro^2 >> tol
    } else {
        double c1x = A[0][1] * A[2][2] - A[0][2] * A[2][1];
        double c1y = A[0][2] * A[2][0] - A[0][0] * A[2][2];
        double c1z = A[0][0] * A[2][1] - A[0][1] * A[2][0];
        double c2x = A[0][1] * A[1][2] - A[0][2] * A[1][1];
        double c2y = A[0][2] * A[1][0] - A[0][0] * A[1][2];
        double c2z = A[0][0] * A[1][1] - A[0][1] * A[1][0];
        double m2 = c0x*c0x + c0y*c0y + c0z*c0z + c1x*c1x + c1y*c1y + c1z*c1z + c2x*c2x + c2y*c2y + c2z*c2z;
        if (m2 > 1e-24 * fro2 * fro2) return true;             // sigma2 >= sqrt(m2/3)/fro >> tol
    }
    double W[3][3], V[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) W[c][r] = A[r][c];
    hestenes3<false>(W, V);
    double s[3], smax = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s[c] = sqrt(fma(W[c][2], W[c][2], fma(W[c][1], W[c][1], W[c][0] * W[c][0])));
        smax = fmax(smax, s[c]);
    }
    double tol = 3.0 * ulp_at(smax);
    int rk = (s[0] > tol) + (s[1] > tol) + (s[2] > tol);
    return rk >= need;
}

// rank from a raw 3x3 Gram matrix G = A'A of an N x 3 point matrix (refit path).
// g = {xx, xy, xz, yy, yz, zz}.
__device__ bool rank_gram_at_least(const double (&g)[6], int N, int need) {
    double tr = g[0] + g[3] + g[5];
    if (!(tr > 0.0)) return false;
    double m00 = g[3] * g[5] - g[4] * g[4], m11 = g[0] * g[5] - g[2] * g[2], m22 = g[0] * g[3] - g[1] * g[1];
    if (need >= 3) {
        double det = g[0] * m00 - g[1] * (g[1] * g[5] - g[4] * g[2]) + g[2] * (g[1] * g[4] - g[3] * g[2]);
        if (det > 1e-10 * tr * tr * tr) return true;
    } else {
        if (m00 + m11 + m22 > 1e-10 * tr * tr) return true;
    }
    // exact-ish: Jacobi eigenvalues of G, sigma = sqrt(lambda)
    double A[3][3] = {{g[0], g[1], g[2]}, {g[1], g[3], g[4]}, {g[2], g[4], g[5]}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= 1e-300 || off <= 1e-19 * tr) break;
#define PCREG_JROT(P, Q)                                                                    \
        if (A[P][Q] != 0.0) {                                                               \
            double th = (A[Q][Q] - A[P][P]) / (2.0 * A[P][Q]);                              \
            double t = copysign(1.0, th) / (fabs(th) + sqrt(fma(th, th, 1.0)));             \
            double c = 1.0 / sqrt(fma(t, t, 1.0)), s = c * t;                               \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = A[k][P], b = A[k][Q]; A[k][P] = c*a - s*b; A[k][Q] = s*a + c*b; } \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = A[P][k], b = A[Q][k]; A[P][k] = c*a - s*b; A[Q][k] = s*a + c*b; } \
        }
        PCREG_JROT(0, 1) PCREG_JROT(0, 2) PCREG_JROT(1, 2)
#undef PCREG_JROT
    }
    double s0 = sqrt(fmax(A[0][0], 0.0)), s1 = sqrt(fmax(A[1][1], 0.0)), s2 = sqrt(fmax(A[2][2], 0.0));
    double smax = fmax(s0, fmax(s1, s2));
    double tol = (double)(N > 3 ? N : 3) * ulp_at(smax);
    return ((s0 > tol) + (s1 > tol) + (s2 > tol)) >= need;
}

// R = V U' (estimateTransform.m:60-62) is the ORTHOGONAL POLAR FACTOR of H' = V S U' (SURVEY 8a row 5), reflections included
// (its determinant has the sign of det H: the reference applies no fix either).  For a well-conditioned H it comes from the
// scaled Newton iteration X <- (g X + e cof(X)) / 2, e ~ 1 / (g det X), which keeps the singular VECTORS and drives every
// singular value to 1: seven 3 x 3 cofactor evaluations instead of ~6 one-sided Jacobi sweeps x 3 rotations, each of which is a
// chain of fp64 divisions and square roots -- 3.6 x fewer dependent instructions per lane-parallel fit.  The scale factors of the
// first five steps only steer the convergence (any g > 0, e det X > 0 preserves the singular vectors), so they are formed in
// fp32 with single-instruction rcp / sqrt; the last two steps are the plain iteration with an fp64 division, which take an
// X within 1e-4 of orthogonal to rounding (defect 4e-16; |R - V U'| <= 1e-14 over 3 x 10^4 sample fits incl. cond 10^6:
// the conditioning of the problem, the same for the SVD).  H with sigma_3 / sigma_1 below ~2e-7 -- rank-2 H of a planar
// set among them -- returns false and takes the SVD path below, which completes the missing singular vector.
__device__ __forceinline__ bool polar_newton(const double (&H)[3][3], double (&X)[3][3]) {
    double f2 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) f2 = fma(H[i][j], H[i][j], f2);
    if (!(f2 > 1e-290) || !(f2 < 1e290)) return false;
    const double inv = 1.0 / sqrt(f2);
    {   // the decision from H itself (nothing else is live yet): det(H' / |H|_F) = det(H) / |H|_F^3
        const double dh = fma(H[0][0], H[1][1] * H[2][2] - H[1][2] * H[2][1],
                              fma(H[0][1], H[1][2] * H[2][0] - H[1][0] * H[2][2], H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0])));
        if (!(fabs(dh) * (inv * inv * inv) >= 1e-7)) return false;          // |X|_F = 1: sigma_3 >= 2 |det X|
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) X[i][j] = H[j][i] * inv;
#define PCREG_COF(C, X)                                                                                   \
    C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0]; \
    C[1][0] = X[2][1] * X[0][2] - X[2][2] * X[0][1]; C[1][1] = X[2][2] * X[0][0] - X[2][0] * X[0][2]; C[1][2] = X[2][0] * X[0][1] - X[2][1] * X[0][0]; \
    C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
    double C[3][3];
    double det;
#pragma unroll 1
    for (int it = 0; it < 7; ++it) {
        PCREG_COF(C, X)
        det = fma(X[0][0], C[0][0], fma(X[0][1], C[0][1], X[0][2] * C[0][2]));
        double g = 0.5, e;
        if (it < 5) {
            float sc = 0.0f, sx = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { const float c = (float)C[i][j], x = (float)X[i][j]; sc = __builtin_fmaf(c, c, sc); sx = __builtin_fmaf(x, x, sx); }
            const float d32 = (float)det;
            const float q = sc * __builtin_amdgcn_rcpf(d32 * d32 * sx);                  // (|cof| / (|det| |X|))^2 = (|X^-1| / |X|)^2
            const float gf = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(q));
            const float ef = __builtin_amdgcn_rcpf(gf * d32);
            // a non-finite or non-positive factor (overflow in fp32) would not be a scaling: take the plain step instead
            const bool okf = gf > 0.0f && gf < 1e30f && ef * d32 > 0.0f && fabsf(ef) < 1e30f;
            g = okf ? 0.5 * (double)gf : 0.5;
            e = okf ? 0.5 * (double)ef : 0.5 / det;
        } else {
            e = 0.5 / det;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) X[i][j] = fma(g, X[i][j], e * C[i][j]);
    }
#undef PCREG_COF
    return true;          // (a non-finite X -- impossible from a finite H past the determinant test -- is caught by the caller's finiteness test of T)
}

// R = V*U' of H = U*S*V' (estimateTransform.m:60-62; no reflection fix), then
// t = cd - R*cm (:63).  T12[j*4+k] = R(j,k), T12[j*4+3] = t(j).  Returns false when the
// rotation is undefined (rank(H) <= 1) or not finite.
__device__ bool polar_to_T(const double (&H)[3][3], const double (&cd)[3], const double (&cm)[3],
                           double (&T)[12]) {
    {
        double Rn[3][3];
        if (polar_newton(H, Rn)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double t = cd[i] - fma(Rn[i][2], cm[2], fma(Rn[i][1], cm[1], Rn[i][0] * cm[0]));
                T[i * 4 + 0] = Rn[i][0]; T[i * 4 + 1] = Rn[i][1]; T[i * 4 + 2] = Rn[i][2]; T[i * 4 + 3] = t;
            }
            bool fin = true;
#pragma unroll
            for (int k = 0; k < 12; ++k) fin = fin && isfinite(T[k]);
            return fin;
        }
    }
    double W[3][3], V[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) W[c][r] = H[r][c];
    hestenes3<true>(W, V);
    double S[3], smax = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S[c] = sqrt(fma(W[c][2], W[c][2], fma(W[c][1], W[c][1], W[c][0] * W[c][0])));
        smax = fmax(smax, S[c]);
    }
    bool ok[3]; int nok = 0;
    double U[3][3];   // U[row][col]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ok[c] = (S[c] > 1e-300) && (S[c] >= smax * 1e-12);
        nok += ok[c];
        double inv = 1.0 / S[c];
#pragma unroll
        for (int r = 0; r < 3; ++r) U[r][c] = W[c][r] * inv;
    }
    if (nok < 2 || !(smax > 0.0)) return false;
    if (nok == 2) {   // complete the missing column so that (u_m, u_a, u_b) is right-handed
#define PCREG_CROSS(M, A_, B_)                                                   \
        U[0][M] = U[1][A_] * U[2][B_] - U[2][A_] * U[1][B_];                     \
        U[1][M] = U[2][A_] * U[0][B_] - U[0][A_] * U[2][B_];                     \
        U[2][M] = U[0][A_] * U[1][B_] - U[1][A_] * U[0][B_];
        if (!ok[0]) { PCREG_CROSS(0, 1, 2) } else if (!ok[1]) { PCREG_CROSS(1, 2, 0) } else { PCREG_CROSS(2, 0, 1) }
#undef PCREG_CROSS
    }
    bool fin = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double r0 = fma(V[i][2], U[0][2], fma(V[i][1], U[0][1], V[i][0] * U[0][0]));
        double r1 = fma(V[i][2], U[1][2], fma(V[i][1], U[1][1], V[i][0] * U[1][0]));
        double r2 = fma(V[i][2], U[2][2], fma(V[i][1], U[2][1], V[i][0] * U[2][0]));
        double t = cd[i] - fma(r2, cm[2], fma(r1, cm[1], r0 * cm[0]));
        T[i * 4 + 0] = r0; T[i * 4 + 1] = r1; T[i * 4 + 2] = r2; T[i * 4 + 3] = t;
        fin = fin && isfinite(r0) && isfinite(r1) && isfinite(r2) && isfinite(t);
    }
    return fin;
}

// estimateTransform for exactly three correspondences (estimateTransform.m:18-37 adds
// the synthetic 4th point, then the common path :41-71).  A[pt][coord].
__device__ bool fit_3pt(const double (&A1)[3][3], const double (&A2)[3][3], double (&T)[12]) {
    if (!rank3x3_at_least(A1, 3) || !rank3x3_at_least(A2, 2)) return false;   // :11-14
    double d[4][3], m[4][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double (&P)[3][3] = s == 0 ? A1 : A2;
        double (&O)[4][3] = s == 0 ? d : m;
        double cen[3], a[3], b[3], e[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cen[c] = (P[0][c] + P[1][c] + P[2][c]) / 3.0;                      // :20-21
            a[c] = P[2][c] - P[1][c]; b[c] = P[2][c] - P[0][c];
        }
        double nx = a[1] * b[2] - a[2] * b[1], ny = a[2] * b[0] - a[0] * b[2], nz = a[0] * b[1] - a[1] * b[0]; // :24-25
#pragma unroll
        for (int i = 0; i < 3; ++i) {                                          // :28-29
            const int j = (i + 2) % 3;
            double dx = P[i][0] - P[j][0], dy = P[i][1] - P[j][1], dz = P[i][2] - P[j][2];
            e[i] = sqrt(dx * dx + dy * dy + dz * dz);
        }
        double l = fmax(fmin(e[0], e[1]), fmin(fmax(e[0], e[1]), e[2]));     // median
        double nn = sqrt(nx * nx + ny * ny + nz * nz);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) O[i][c] = P[i][c];
        O[3][0] = cen[0] + (nx / nn) * l; O[3][1] = cen[1] + (ny / nn) * l; O[3][2] = cen[2] + (nz / nn) * l;   // :32-36
    }
    double cd[3], cm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                              // :46-47
        cd[c] = (((d[0][c] + d[1][c]) + d[2][c]) + d[3][c]) / 4.0;
        cm[c] = (((m[0][c] + m[1][c]) + m[2][c]) + m[3][c]) / 4.0;
    }
    double H[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {                                          // :55-58
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(m[k][i] - cm[i], d[k][j] - cd[j], acc);
            H[i][j] = acc;
        }
    return polar_to_T(H, cd, cm, T);
}

// Moments of a correspondence set, taken about a fixed origin (o1,o2) so that the
// centring of estimateTransform.m:55-58 does not cancel digits:
//   mom[0..2]  = sum(d')      mom[3..5] = sum(m')            (d' = p1-o1, m' = p2-o2)
//   mom[6..14] = sum(m'_i d'_j), row-major i,j
//   mom[15..20]= raw Gram of p1 (xx,xy,xz,yy,yz,zz), mom[21..26] = raw Gram of p2.
__device__ __forceinline__ void mom_core(double (&mom)[27], const double (&p)[6], const double (&o)[6]) {
    double d0 = p[0] - o[0], d1 = p[1] - o[1], d2 = p[2] - o[2];
    double m0 = p[3] - o[3], m1 = p[4] - o[4], m2 = p[5] - o[5];
    mom[0] += d0; mom[1] += d1; mom[2] += d2; mom[3] += m0; mom[4] += m1; mom[5] += m2;
    mom[6]  = fma(m0, d0, mom[6]);  mom[7]  = fma(m0, d1, mom[7]);  mom[8]  = fma(m0, d2, mom[8]);
    mom[9]  = fma(m1, d0, mom[9]);  mom[10] = fma(m1, d1, mom[10]); mom[11] = fma(m1, d2, mom[11]);
    mom[12] = fma(m2, d0, mom[12]); mom[13] = fma(m2, d1, mom[13]); mom[14] = fma(m2, d2, mom[14]);
}
// the raw Grams only feed the rank test of estimateTransform.m:11-14
__device__ __forceinline__ void mom_gram(double (&mom)[27], const double (&p)[6]) {
    mom[15] = fma(p[0], p[0], mom[15]); mom[16] = fma(p[0], p[1], mom[16]); mom[17] = fma(p[0], p[2], mom[17]);
    mom[18] = fma(p[1], p[1], mom[18]); mom[19] = fma(p[1], p[2], mom[19]); mom[20] = fma(p[2], p[2], mom[20]);
    mom[21] = fma(p[3], p[3], mom[21]); mom[22] = fma(p[3], p[4], mom[22]); mom[23] = fma(p[3], p[5], mom[23]);
    mom[24] = fma(p[4], p[4], mom[24]); mom[25] = fma(p[4], p[5], mom[25]); mom[26] = fma(p[5], p[5], mom[26]);
}
__device__ __forceinline__ void mom_accumulate(double (&mom)[27], const double (&p)[6],
                                               const double (&o)[6]) {
    mom_core(mom, p, o);
    mom_gram(mom, p);
}

// The same decision with the eigenvectors of the two smaller singular values and a flag: a singular value below ~1e-7
// sigma_max lies under the noise floor of the Gram matrix (forming A'A squares the condition number), MATLAB's tolerance
// N eps(sigma_max) is ~1e-12 sigma_max, so in between the Gram cannot tell.  The caller that still has the points settles
// it with one more pass: the 2 x 2 Gram of the points projected on span(vmid, vmin) holds sigma_2^2 and sigma_3^2 free of
// the cancellation against sigma_1^2 (the span is accurate to ~eps even when the split inside it is noise); see
// rank_from_projection, estimate_transform_kernel (estimateTransform.m:11).
__device__ bool rank_gram_probe(const double (&g)[6], int N, int need, double (&vmid)[3], double (&vmin)[3], double& tol,
                                bool& ambiguous) {
    ambiguous = false; tol = 0.0;
    vmid[0] = vmid[1] = vmid[2] = vmin[0] = vmin[1] = vmin[2] = 0.0;
    double tr = g[0] + g[3] + g[5];
    if (!(tr > 0.0)) return false;
    double A[3][3] = {{g[0], g[1], g[2]}, {g[1], g[3], g[4]}, {g[2], g[4], g[5]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= 1e-300 || off <= 1e-19 * tr) break;
#define PCREG_JROT(P, Q)                                                                    \
        if (A[P][Q] != 0.0) {                                                               \
            double th = (A[Q][Q] - A[P][P]) / (2.0 * A[P][Q]);                              \
            double t = copysign(1.0, th) / (fabs(th) + sqrt(fma(th, th, 1.0)));             \
            double c = 1.0 / sqrt(fma(t, t, 1.0)), s = c * t;                               \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = A[k][P], b = A[k][Q]; A[k][P] = c*a - s*b; A[k][Q] = s*a + c*b; } \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = A[P][k], b = A[Q][k]; A[P][k] = c*a - s*b; A[Q][k] = s*a + c*b; } \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) { double a = V[k][P], b = V[k][Q]; V[k][P] = c*a - s*b; V[k][Q] = s*a + c*b; } \
        }
        PCREG_JROT(0, 1) PCREG_JROT(0, 2) PCREG_JROT(1, 2)
#undef PCREG_JROT
    }
    double sv[3] = {sqrt(fmax(A[0][0], 0.0)), sqrt(fmax(A[1][1], 0.0)), sqrt(fmax(A[2][2], 0.0))};
    int od[3] = {0, 1, 2};                                   // descending
    if (sv[od[1]] > sv[od[0]]) { int t = od[0]; od[0] = od[1]; od[1] = t; }
    if (sv[od[2]] > sv[od[0]]) { int t = od[0]; od[0] = od[2]; od[2] = t; }
    if (sv[od[2]] > sv[od[1]]) { int t = od[1]; od[1] = od[2]; od[2] = t; }
    const double smax = sv[od[0]];
    tol = (double)(N > 3 ? N : 3) * ulp_at(smax);
    const int kd = need >= 3 ? od[2] : od[1];                // the singular value that decides rank >= need
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { if (c == od[1]) a = V[k][c]; if (c == od[2]) b = V[k][c]; }
        vmid[k] = a; vmin[k] = b;
    }
    ambiguous = sv[kd] <= 1e-7 * smax;
    return ((sv[0] > tol) + (sv[1] > tol) + (sv[2] > tol)) >= need;
}
// sigma_2 (need = 2) or sigma_3 (need = 3) from the projected 2 x 2 Gram [aa ab; ab bb], compared with MATLAB's tolerance
__device__ bool rank_from_projection(double aa, double ab, double bb, int need, double tol) {
    const double h = 0.5 * (aa - bb), lmax = 0.5 * (aa + bb) + sqrt(fma(h, h, ab * ab));
    if (!(lmax > 0.0)) return false;
    if (need < 3) return sqrt(lmax) > tol;
    const double w = ab * ab, det = fma(aa, bb, -w) + fma(-ab, ab, w);      // aa bb - ab^2 with the product's rounding undone
    return sqrt(fmax(det, 0.0) / lmax) > tol;
}

// estimateTransform for N > 3 correspondences given their moments.
__device__ bool fit_moments(int N, const double (&mom)[27], const double (&o)[6], double (&T)[12], bool rank_certified = false) {
    if (N < 4) return false;
    double g1[6] = {mom[15], mom[16], mom[17], mom[18], mom[19], mom[20]};
    double g2[6] = {mom[21], mom[22], mom[23], mom[24], mom[25], mom[26]};
    if (!rank_certified && (!rank_gram_at_least(g1, N, 3) || !rank_gram_at_least(g2, N, 2))) return false;   // :11-14
    double inv = 1.0 / (double)N;
    double cdp[3] = {mom[0] * inv, mom[1] * inv, mom[2] * inv};
    double cmp_[3] = {mom[3] * inv, mom[4] * inv, mom[5] * inv};
    double H[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) H[i][j] = mom[6 + i * 3 + j] - (double)N * cmp_[i] * cdp[j];
    double cd[3] = {cdp[0] + o[0], cdp[1] + o[1], cdp[2] + o[2]};
    double cm[3] = {cmp_[0] + o[3], cmp_[1] + o[4], cmp_[2] + o[5]};
    return polar_to_T(H, cd, cm, T);
}

// ---------------------------------------------------------------- point access
template <bool LDS_PTS>
struct Pts {
    const double* g1; const double* g2; int ld; const double* s; int n;
    __device__ __forceinline__ void load(int i, double (&p)[6]) const {
        if (LDS_PTS) {
#pragma unroll
            for (int c = 0; c < 6; ++c) p[c] = s[c * n + i];
        } else {
            p[0] = g1[i]; p[1] = g1[i + (size_t)ld]; p[2] = g1[i + 2 * (size_t)ld];
            p[3] = g2[i]; p[4] = g2[i + (size_t)ld]; p[5] = g2[i + 2 * (size_t)ld];
        }
    }
};

// calcDists (getInliersRANSAC.m:50-53) in the canonical FMA order shared with the oracle.
__device__ __forceinline__ double sqdist(const double (&p)[6], const double (&T)[12]) {
    double tx = fma(p[3], T[0], fma(p[4], T[1], fma(p[5], T[2],  T[3])));
    double ty = fma(p[3], T[4], fma(p[4], T[5], fma(p[5], T[6],  T[7])));
    double tz = fma(p[3], T[8], fma(p[4], T[9], fma(p[5], T[10], T[11])));
    double dx = p[0] - tx, dy = p[1] - ty, dz = p[2] - tz;
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

template <bool LDS_PTS>
__device__ __forceinline__ void score_pass(const Pts<LDS_PTS>& P, int n, int lane,
                                           const double (&T)[HB][12], double th, int (&cnt)[HB]) {
#pragma unroll
    for (int b = 0; b < HB; ++b) cnt[b] = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        int i = i0 + lane;
        bool act = i < n;
        double p[6];
        P.load(act ? i : n - 1, p);
#pragma unroll
        for (int b = 0; b < HB; ++b) {
            double d = sqdist(p, T[b]);
            cnt[b] += __popcll(__ballot(act && d < th));
        }
    }
}

__device__ __forceinline__ void bcast_T(const double (&Tl)[12], int h, double (&T)[12]) {
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = rdlane(Tl[k], h);
}
// A VOP3 may read one SGPR: with the whole transform in SGPRs every fma(x, R, t) needs a
// v_mov_b64 for t first.  Keeping the three translations in VGPRs removes 3 moves per point.
__device__ __forceinline__ void pin_translation_vgpr(double (&T)[12]) {
    asm volatile("" : "+v"(T[3]), "+v"(T[7]), "+v"(T[11]));
}

// sample indices of hypothesis p (0-based out).  Built-in sampler = oracle's sample_table.
__device__ __forceinline__ void sample3(const RansacArgs& a, int b, int p, int n, int (&s)[3]) {
    if (a.sample_idx) {
        const int32_t* t = a.sample_idx + ((size_t)b * a.iters + p) * 3;
        s[0] = t[0] - 1; s[1] = t[1] - 1; s[2] = t[2] - 1;
    } else {
        unsigned long long seed = a.seed + (unsigned long long)b;
        unsigned r[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            unsigned long long h = splitmix64(seed ^ splitmix64((unsigned long long)(p + a.hyp0g) * 16ull + j));
            r[j] = (unsigned)(((h >> 32) * (unsigned long long)(unsigned)(n - j)) >> 32);
        }
        unsigned i0 = r[0], i1 = r[1];
        if (i1 >= i0) ++i1;
        unsigned lo = min(i0, i1), hi = max(i0, i1), i2 = r[2];
        if (i2 >= lo) ++i2;
        if (i2 >= hi) ++i2;
        s[0] = (int)i0; s[1] = (int)i1; s[2] = (int)i2;
    }
    // out-of-range tables must not fault the GPU: clamp (documented in pcreg.h)
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] = min(max(s[j], 0), n - 1);
}

// ---------------------------------------------------------------- hypothesis kernel
template <bool LDS_PTS, int NW = kWavesPerBlock>
__device__ __forceinline__ void ransac_hyp_body(const RansacArgs& a, double* __restrict__ sp, const int b, const int off, const int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t hyp0 = (size_t)b * a.iters;
    const int wbase = (blockIdx.x * NW + wave) * a.hpw;   // first hypothesis of this wave
    const double* g1 = a.p1 + off; const double* g2 = a.p2 + off;

    if (LDS_PTS) {
        for (int i = threadIdx.x; i < n; i += NW * 64) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { sp[c * n + i] = g1[i + (size_t)c * a.ld]; sp[(3 + c) * n + i] = g2[i + (size_t)c * a.ld]; }
        }
        __syncthreads();
    }
    if (wbase >= a.iters) return;
    const int nh = min(a.hpw, a.iters - wbase);
    const int p = wbase + lane;
    const bool mine = lane < nh;
    Pts<LDS_PTS> P{g1, g2, a.ld, sp, n};
    const int thInlr = matlab_round_i(a.ratio * (double)n);                   // ransac.m:28

    if (n < a.m || n < 3) {   // randperm(ptNum)(1:minPtNum) would throw; report nothing found
        if (mine) { a.cnt1[hyp0 + p] = 0; a.cnt2[hyp0 + p] = 0; a.has[hyp0 + p] = 0; }
        return;
    }
    double o[6];
    P.load(0, o);   // fixed origin for the refit moments
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = rdlane(o[c], 0);     // the same six numbers in every lane: SGPRs, not 12 of the kernel's 177 VGPRs

    // ---- phase 0: minimal-sample fit, one hypothesis per lane (ransac.m:42-45)
    double T1[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T1[k] = 0.0;
    bool v1 = false;
    if (mine) {
        if (a.m == 3) {
            int s[3]; sample3(a, b, p, n, s);
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double q[6]; P.load(s[j], q);
                A1[j][0] = q[0]; A1[j][1] = q[1]; A1[j][2] = q[2];
                A2[j][0] = q[3]; A2[j][1] = q[4]; A2[j][2] = q[5];
            }
            v1 = fit_3pt(A1, A2, T1);
        } else {   // minPtNum > 3: general estimateTransform path on the sample
            double mom[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) mom[k] = 0.0;
            const int32_t* t = a.sample_idx + ((size_t)hyp0 + p) * a.m;
            double os[6];
            for (int j = 0; j < a.m; ++j) {
                int idx = min(max(t[j] - 1, 0), n - 1);
                double q[6]; P.load(idx, q);
                if (j == 0) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) os[c] = q[c];
                }
                mom_accumulate(mom, q, os);
            }
            v1 = fit_moments(a.m, mom, os, T1);
        }
    }

    // ---- phase 1: score every sample fit (ransac.m:48-50)
    int c1 = 0, c2 = 0;
    for (int h = 0; h < nh; h += HB) {
        double T[HB][12]; int cnt[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) bcast_T(T1, min(h + k, nh - 1), T[k]);
        score_pass<LDS_PTS>(P, n, lane, T, a.thDist, cnt);
#pragma unroll
        for (int k = 0; k < HB; ++k) if (lane == h + k) c1 = cnt[k];
    }
    if (!v1) c1 = 0;   // empty transform: scores 0 (deviation documented in DESIGN.md)
    const bool pass1 = mine && v1 && c1 >= thInlr;                              // ransac.m:53

    if (!a.refine) {                                                            // ransac.m:62-64
        if (mine) {
            a.cnt1[hyp0 + p] = c1; a.cnt2[hyp0 + p] = 0; a.has[hyp0 + p] = pass1;
            if (pass1) {
#pragma unroll
                for (int k = 0; k < 12; ++k) a.TF[(hyp0 + p) * 12 + k] = T1[k];
            }
        }
        return;
    }

    // ---- phase 2: moments of each passing hypothesis' inlier set (ransac.m:55)
    double mom[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) mom[k] = 0.0;
    unsigned long long mask = __ballot(pass1);
    while (mask) {
        const int h = __builtin_ctzll(mask);
        mask &= mask - 1;
        double T[12]; bcast_T(T1, h, T);
        const int ch = __builtin_amdgcn_readlane(c1, h);
        if (ch == 3) {
            // exactly three inliers: the refit is estimateTransform's N == 3 branch
            // (estimateTransform.m:18-37); hand the three points to lane h in mom[0..17]
            int k = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                int i = i0 + lane;
                bool act = i < n;
                double q[6]; P.load(act ? i : n - 1, q);
                unsigned long long bal = __ballot(act && sqdist(q, T) < a.thDist);
                while (bal) {
                    const int L = __builtin_ctzll(bal);
                    bal &= bal - 1;
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = rdlane(q[c], L);
                        if (lane == h) {
                            if (k == 0) mom[c] = v; else if (k == 1) mom[6 + c] = v; else if (k == 2) mom[12 + c] = v;
                        }
                    }
                    ++k;
                }
            }
            continue;
        }
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            int i = i0 + lane;
            bool act = i < n;
            double q[6]; P.load(act ? i : n - 1, q);
            if (act && sqdist(q, T) < a.thDist) mom_accumulate(acc, q, o);
        }
        wave_sum27(acc);
#pragma unroll
        for (int k = 0; k < 27; ++k) if (lane == h) mom[k] = acc[k];
    }

    // ---- phase 3: refit, one hypothesis per lane (estimateTransform on the inliers)
    double T2[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T2[k] = 0.0;
    bool v2 = false;
    if (pass1) {
        if (c1 == 3) {
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) { A1[j][c] = mom[j * 6 + c]; A2[j][c] = mom[j * 6 + 3 + c]; }
            v2 = fit_3pt(A1, A2, T2);
        } else {
            v2 = fit_moments(c1, mom, o, T2);
        }
    }

    // ---- phase 4: rescore the refined transforms (ransac.m:56-58)
    mask = __ballot(v2);
    while (mask) {
        int hs[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            hs[k] = mask ? __builtin_ctzll(mask) : hs[k > 0 ? k - 1 : 0];
            if (mask) mask &= mask - 1;
        }
        double T[HB][12]; int cnt[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) bcast_T(T2, hs[k], T[k]);
        score_pass<LDS_PTS>(P, n, lane, T, a.thDist, cnt);
#pragma unroll
        for (int k = 0; k < HB; ++k) if (lane == hs[k]) c2 = cnt[k];
    }
    if (mine) {
        const bool keep = v2 && c2 >= thInlr;                                   // ransac.m:59-61
        a.cnt1[hyp0 + p] = c1; a.cnt2[hyp0 + p] = v2 ? c2 : 0; a.has[hyp0 + p] = keep;
        if (keep) {
#pragma unroll
            for (int k = 0; k < 12; ++k) a.TF[(hyp0 + p) * 12 + k] = T2[k];
        }
    }
}

// The correspondences of a registration sit in LDS when they fit the launch's dynamic LDS (a.n_hi of them); a registration of a
// batch that does not (the batch's LDS is sized for the common case, not for its capacity: a workgroup that RESERVES 96 KB is
// alone on its CU) reads them from L2 instead -- the same code on the other point source, chosen per workgroup.
__global__ __launch_bounds__(kBlock) void ransac_hyp_kernel(RansacArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sp[];
    const int b = blockIdx.y;
    const int off = a.offsets ? a.offsets[b] : 0;
    int n = a.offsets ? (a.offsets[b + 1] - off) : (a.n_dev ? *a.n_dev : a.n_cap);
    n = min(n, a.n_cap);
    if (n <= a.n_lo) return;                                    // ransac_hyp32_kernel serves it
    if (n <= a.n_hi) ransac_hyp_body<true>(a, sp, b, off, n);
    else ransac_hyp_body<false>(a, sp, b, off, n);
}

// ---------------------------------------------------------------- hypothesis kernel, large n
// When the correspondences do not fit LDS, a per-wave sweep streams n x 48 B from L2 for
// every hypothesis pair and the kernel is L2-bandwidth bound (measured: 31 GB per
// registration at n = 32.5k).  Here an 8-wave workgroup marches over the points TOGETHER:
// a tile of kTile points is staged once in LDS (six SoA fp64 columns) and every wave scores
// its own hypothesis pair against it, so L2 traffic drops 8x and the sweeps run at LDS
// speed.  All waves execute the same sequence of sweeps (score, refit moments, rescore);
// block-wide votes (__syncthreads_or) decide which sweeps are needed at all.
constexpr int kTW = 8;                  // waves per workgroup (2 per SIMD: 256 VGPRs each)
constexpr int kTBlock = kTW * 64;
constexpr int kPPT = 2;                 // points per thread and tile
constexpr int kTile = kTBlock * kPPT;   // points per LDS tile (2 x 48 KiB, double-buffered)

// One pass of the whole workgroup over the correspondences through LDS tiles, software
// pipelined: the loads of tile t+1 are issued before tile t is scored and land in the other
// LDS buffer afterwards (one barrier per tile, L2 latency hidden behind the arithmetic).
// A macro pair rather than a lambda-taking template: captured-by-reference accumulators
// ended up in scratch.
#define PCREG_TILE_SWEEP_BEGIN                                                              \
    {                                                                                       \
        const int nt_ = (n + kTile - 1) / kTile;                                            \
        double pf_[kPPT][6];                                                                \
        _Pragma("unroll") for (int v_ = 0; v_ < kPPT; ++v_) {                               \
            const int gi_ = v_ * kTBlock + threadIdx.x; const bool ok_ = gi_ < n;           \
            _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                              \
                pf_[v_][c_] = ok_ ? g1[gi_ + (size_t)c_ * a.ld] : 0.0;                      \
                pf_[v_][3 + c_] = ok_ ? g2[gi_ + (size_t)c_ * a.ld] : 0.0;                  \
            }                                                                               \
            _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) sp[c_ * kTile + gi_] = pf_[v_][c_]; \
        }                                                                                   \
        __syncthreads();                                                                    \
        for (int t_ = 0; t_ < nt_; ++t_) {                                                  \
            const bool more_ = t_ + 1 < nt_;                                                \
            if (more_) {                                                                    \
                _Pragma("unroll") for (int v_ = 0; v_ < kPPT; ++v_) {                       \
                    const int gi_ = (t_ + 1) * kTile + v_ * kTBlock + threadIdx.x; const bool ok_ = gi_ < n; \
                    _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                      \
                        pf_[v_][c_] = ok_ ? g1[gi_ + (size_t)c_ * a.ld] : 0.0;              \
                        pf_[v_][3 + c_] = ok_ ? g2[gi_ + (size_t)c_ * a.ld] : 0.0;          \
                    }                                                                       \
                }                                                                           \
            }                                                                               \
            const double* buf_ = sp + (t_ & 1) * 6 * kTile;                                 \
            const int cnt_ = min(kTile, n - t_ * kTile);                                    \
            for (int i0_ = 0; i0_ < cnt_; i0_ += 128) {   /* two 64-point slots per trip: 12 LDS reads in flight */ \
                double qq_[2][6];                                                           \
                _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_)                            \
                    _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) qq_[u_][c_] = buf_[c_ * kTile + i0_ + u_ * 64 + lane]; \
                _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                          \
                    const bool act = i0_ + u_ * 64 + lane < cnt_;                           \
                    double q[6];                                                            \
                    _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) q[c_] = qq_[u_][c_];
#define PCREG_TILE_SWEEP_END                                                                \
                }                                                                           \
            }                                                                               \
            if (more_) {                                                                    \
                double* nb_ = sp + ((t_ + 1) & 1) * 6 * kTile;                              \
                _Pragma("unroll") for (int v_ = 0; v_ < kPPT; ++v_)                         \
                    _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) nb_[c_ * kTile + v_ * kTBlock + threadIdx.x] = pf_[v_][c_]; \
            }                                                                               \
            __syncthreads();                                                                \
        }                                                                                   \
    }

__global__ __launch_bounds__(kTBlock) void ransac_hyp_tiled_kernel(RansacArgs a) {
    __shared__ __attribute__((aligned(16))) double sp[2 * 6 * kTile];
    __shared__ double s_mom[kTW][HB][27];    // refit moments (or the 3 inlier points) of the wave's current hypotheses
    const int b = blockIdx.y;
    const int off = a.offsets ? a.offsets[b] : 0;
    int n = a.offsets ? (a.offsets[b + 1] - off) : (a.n_dev ? *a.n_dev : a.n_cap);
    n = min(n, a.n_cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t hyp0 = (size_t)b * a.iters;
    const int wbase = (blockIdx.x * kTW + wave) * a.hpw;
    const double* g1 = a.p1 + off; const double* g2 = a.p2 + off;
    const int nh = max(0, min(a.hpw, a.iters - wbase));      // 0: this wave only helps with the tiles
    const int p = wbase + lane;
    const bool mine = lane < nh;
    Pts<false> P{g1, g2, a.ld, nullptr, n};
    const int thInlr = matlab_round_i(a.ratio * (double)n);

    if (n < a.m || n < 3) {
        if (mine) { a.cnt1[hyp0 + p] = 0; a.cnt2[hyp0 + p] = 0; a.has[hyp0 + p] = 0; }
        return;
    }
    double o[6];
    P.load(0, o);

    // ---- phase 0: sample fits, one hypothesis per lane
    double T1[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T1[k] = 0.0;
    bool v1 = false;
    if (mine) {
        if (a.m == 3) {
            int s[3]; sample3(a, b, p, n, s);
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double q[6]; P.load(s[j], q);
                A1[j][0] = q[0]; A1[j][1] = q[1]; A1[j][2] = q[2];
                A2[j][0] = q[3]; A2[j][1] = q[4]; A2[j][2] = q[5];
            }
            v1 = fit_3pt(A1, A2, T1);
        } else {
            double mom[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) mom[k] = 0.0;
            const int32_t* t = a.sample_idx + ((size_t)hyp0 + p) * a.m;
            double os[6];
            for (int j = 0; j < a.m; ++j) {
                int idx = min(max(t[j] - 1, 0), n - 1);
                double q[6]; P.load(idx, q);
                if (j == 0) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) os[c] = q[c];
                }
                mom_accumulate(mom, q, os);
            }
            v1 = fit_moments(a.m, mom, os, T1);
        }
    }

    int c1 = 0, c2 = 0;
    double T2[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T2[k] = 0.0;
    bool v2 = false, pass1 = false;
    const double th = a.thDist;

    for (int h0 = 0; h0 < a.hpw; h0 += HB) {            // rounds: every wave, same trip count
        // ---- phase 1: score the sample fits of this round
        {
            double T[HB][12]; int cnt[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k) { bcast_T(T1, min(h0 + k, 63), T[k]); pin_translation_vgpr(T[k]); cnt[k] = 0; }
            PCREG_TILE_SWEEP_BEGIN
#pragma unroll
                for (int k = 0; k < HB; ++k) if (h0 + k < a.hpw) cnt[k] += __popcll(__ballot((sqdist(q, T[k]) < th) & act));
            PCREG_TILE_SWEEP_END
#pragma unroll
            for (int k = 0; k < HB; ++k) if (lane == h0 + k) c1 = cnt[k];
        }
        if (!v1) c1 = 0;
        {
            const bool in_round = mine && lane >= h0 && lane < h0 + HB;
            if (in_round) pass1 = v1 && c1 >= thInlr;
        }
        if (!a.refine) continue;

        // ---- phase 2+3: refit moments, one passing hypothesis of every wave per sweep;
        //      the 27 sums (or the three inlier points) go through s_mom to the owning lane
#pragma unroll 1
        for (int k = 0; k < HB; ++k) {
            const int h = h0 + k;
            const bool want = __builtin_amdgcn_readlane((int)(pass1 && (lane == h)), min(h, 63)) != 0 && h < nh;
            if (!__syncthreads_or(want)) continue;          // nobody in the workgroup needs this sweep
            double T[12]; bcast_T(T1, min(h, 63), T); pin_translation_vgpr(T);
            const int ch = __builtin_amdgcn_readlane(c1, min(h, 63));
            double acc[27];
#pragma unroll
            for (int e = 0; e < 27; ++e) acc[e] = 0.0;
            int k3 = 0;
            PCREG_TILE_SWEEP_BEGIN
                const bool in = (sqdist(q, T) < th) & act & want;
                if (in) mom_accumulate(acc, q, o);
                if (ch == 3) {                  // estimateTransform's N == 3 branch needs the points themselves
                    unsigned long long bal = __ballot(in);
                    const int r3 = k3 + __popcll(bal & ((1ull << lane) - 1ull));
                    if (in && r3 < 3) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) s_mom[wave][k][r3 * 6 + c] = q[c];
                    }
                    k3 += __popcll(bal);
                }
            PCREG_TILE_SWEEP_END
            if (want && ch != 3) {
                wave_sum27(acc);
#pragma unroll
                for (int e = 0; e < 27; ++e) if (lane == 0) s_mom[wave][k][e] = acc[e];
            }
        }
        // s_mom was written by this wave only (after the sweep's last barrier)
        __builtin_amdgcn_wave_barrier();
        {
            const bool in_round = mine && lane >= h0 && lane < h0 + HB;
            if (in_round && pass1) {         // both hypotheses of the round refit side by side
                double mom[27];
#pragma unroll
                for (int e = 0; e < 27; ++e) mom[e] = s_mom[wave][lane - h0][e];
                if (c1 == 3) {
                    double A1[3][3], A2[3][3];
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int c = 0; c < 3; ++c) { A1[j][c] = mom[j * 6 + c]; A2[j][c] = mom[j * 6 + 3 + c]; }
                    v2 = fit_3pt(A1, A2, T2);
                } else {
                    v2 = fit_moments(c1, mom, o, T2);
                }
            }
        }
        // ---- phase 4: rescore the refined transforms of this round
        {
            bool any = false;
#pragma unroll
            for (int k = 0; k < HB; ++k) any |= (__builtin_amdgcn_readlane((int)v2, min(h0 + k, 63)) != 0) && (h0 + k < nh);
            if (__syncthreads_or(any)) {
                double T[HB][12]; int cnt[HB];
#pragma unroll
                for (int k = 0; k < HB; ++k) { bcast_T(T2, min(h0 + k, 63), T[k]); pin_translation_vgpr(T[k]); cnt[k] = 0; }
                PCREG_TILE_SWEEP_BEGIN
#pragma unroll
                    for (int k = 0; k < HB; ++k) if (h0 + k < a.hpw) cnt[k] += __popcll(__ballot((sqdist(q, T[k]) < th) & act));
                PCREG_TILE_SWEEP_END
#pragma unroll
                for (int k = 0; k < HB; ++k) if (lane == h0 + k) c2 = cnt[k];
            }
        }
    }
    if (mine) {
        if (!a.refine) {
            a.cnt1[hyp0 + p] = c1; a.cnt2[hyp0 + p] = 0; a.has[hyp0 + p] = pass1;
            if (pass1) {
#pragma unroll
                for (int k = 0; k < 12; ++k) a.TF[(hyp0 + p) * 12 + k] = T1[k];
            }
        } else {
            const bool keep = v2 && c2 >= thInlr;
            a.cnt1[hyp0 + p] = c1; a.cnt2[hyp0 + p] = v2 ? c2 : 0; a.has[hyp0 + p] = keep;
            if (keep) {
#pragma unroll
                for (int k = 0; k < 12; ++k) a.TF[(hyp0 + p) * 12 + k] = T2[k];
            }
        }
    }
}


// ================================================================ staged pipeline (large sets)
// The fused tiled kernel above holds the lane-parallel SVD fits, the scoring sweeps and the
// moment sweeps in one 219-VGPR body (2 waves per SIMD).  For one large registration
// (B == 1, n > 1365) the same arithmetic runs as a chain of lean kernels instead:
//   rs_fit1   one hypothesis per lane: sample + estimateTransform            -> T1, v1
//   rs_score  TRANSPOSED scoring: a wave keeps 256 correspondences in registers and walks the
//             hypotheses (their transforms arrive through the scalar cache); no LDS tiles, no
//             barriers, 17 fp64 ops per pair and nothing else                -> partial counts
//   rs_moments the refit sweeps of the tiled kernel, two hypotheses per pass -> 27 moments
//   rs_fit2   one hypothesis per lane: estimateTransform from the moments   -> T2, v2
//   rs_score  on T2                                                          -> partial counts
//   rs_finish counts, `has` flags; then the unchanged ransac_select_kernel.
// Counts are integers (order-free); the moments keep the tiled kernel's summation order
// (lane-strided partial sums over all points, butterfly over the wave), so every number is
// the one the fused kernels produce.
constexpr int kSW = 4, kSS = 8;                 // scoring: waves per workgroup, 64-point slots per wave
constexpr int kSPts = kSW * kSS * 64;           // 2048 correspondences per workgroup
constexpr int kSMaxPB = 64;                     // point blocks with a partial-count row each
#ifndef PCREG_MOM_SLOTS
#define PCREG_MOM_SLOTS 44
#endif
constexpr int kStagedMinN = 2049;               // one registration of at least this many correspondences MAY run staged (see staged_pays)
constexpr int kMomSlots = PCREG_MOM_SLOTS;                   // lane-per-hypothesis refit: 64-correspondence slots per chunk
constexpr int kRec = 16;                        // doubles per correspondence record (15 used)
#ifndef PCREG_SCHUNK
#define PCREG_SCHUNK 32
#endif
constexpr int kSChunk = PCREG_SCHUNK;                     // hypotheses per workgroup

struct StagedArgs {
    RansacArgs a;
    double* T1; unsigned char* v1; unsigned char* pass1; unsigned char* v2;
    unsigned char* cert;         // rank of the inlier set certified from the sample alone (rs_fit1 + rs_pass1)
    double* certq;               // [iters][2]: the sample's singular-value bounds / tolerance factor (0: no certificate)
    double* bounds;              // [4]: max |pts1 row|^2, max |pts2 row|^2, the same of the rows relative to correspondence 0
    void* sel_ctr;               // SelCtr of ransac_select_multi_kernel (cleared by rs_stage1_kernel)
    double* bpart;               // [record workgroups][4]: their maxima (rs_stage1_kernel -> rs_stage2_kernel; no atomics, no clearing)
    int n_rec_blocks;
    double* mom;                 // [iters][27]
    int32_t* part;               // [kSMaxPB][iters]
    int pb;                      // point blocks in use
    // mask-driven refit (rs_moments_mfma_kernel): the first scoring pass keeps every hypothesis' inlier
    // mask, the moments are then sums of per-correspondence records under that mask
    unsigned long long* masks;   // [iters][nslots_cap]: bit (i & 63) of word i / 64 = correspondence i is an inlier
    int nslots_cap;              // ceil(n_cap / kSPts) * kSPts / 64
    double* rec;                 // [nslots_cap * 64][kRec]: d(3) m(3) m (x) d (9) relative to correspondence 0
    double* mpart;               // [chunks][iters][15] partial moments
    uint4* dig;                  // [nslots_cap * 2 k-steps][4 tiles][64 lanes] x 16 int8: the records' base-128 digits (rs_digits_kernel)
    int32_t* pass_list;          // hypotheses on this path, arrival order
    int32_t* n_pass;
    unsigned char* dense;        // [iters] refit needs rs_moments_kernel (rank not certified, or the N == 3 branch)
    int use_lane;
    // fp32-screened scoring (rs_score32_kernel): centred single-precision copies of the correspondences and of
    // the transforms; a hypothesis with any distance within +-E of thDist is re-scored in fp64 by the same wave
    float* c32;                  // [6][n32]: p1 - o1 (x,y,z), p2 - o2 (x,y,z) rounded to fp32, o = correspondence 0
    int n32;                     // row length of c32
    float* T32a; float* T32b;    // [iters][16]: R (9, rows), t' (3), thlo, thhi, 2 pad -- sample fits / refits
    int use_f32;
};

__device__ __forceinline__ int staged_n(const RansacArgs& a) { return min(a.n_dev ? *a.n_dev : a.n_cap, a.n_cap); }

// per-correspondence records of the moment sums (mom_core's fifteen terms), relative to correspondence 0
__device__ __forceinline__ void rs_records_body(const StagedArgs& sa, int block) {
    const RansacArgs& a = sa.a;
    const int n = staged_n(a);
    const int i = block * 256 + threadIdx.x;
    double m1 = 0.0, m2 = 0.0, c1m = 0.0, c2m = 0.0;
    if (i >= n && sa.use_lane && i < sa.nslots_cap * 64) {      // padding records must be finite: the sums multiply them by 0
        double* r = sa.rec + (size_t)i * kRec;
#pragma unroll
        for (int e = 0; e < kRec; ++e) r[e] = 0.0;
    }
    if (i < n) {
        Pts<false> P{a.p1, a.p2, a.ld, nullptr, n};
        double o[6], q[6];
        P.load(0, o); P.load(i, q);
        m1 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
        m2 = q[3] * q[3] + q[4] * q[4] + q[5] * q[5];
        if (sa.use_f32 || sa.use_lane) {                // bounds of the rows relative to correspondence 0: the fp32 screen's and the digit grid's
            double v[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) { v[c] = q[c] - o[c]; if (sa.use_f32) sa.c32[(size_t)c * sa.n32 + i] = (float)v[c]; }
            c1m = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            c2m = v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
        }
        if (sa.use_lane) {
            const double d0 = q[0] - o[0], d1 = q[1] - o[1], d2 = q[2] - o[2];
            const double m0 = q[3] - o[3], m1_ = q[4] - o[4], m2_ = q[5] - o[5];
            double* r = sa.rec + (size_t)i * kRec;
            r[0] = d0; r[1] = d1; r[2] = d2; r[3] = m0; r[4] = m1_; r[5] = m2_;
            r[6] = m0 * d0; r[7] = m0 * d1; r[8] = m0 * d2;
            r[9] = m1_ * d0; r[10] = m1_ * d1; r[11] = m1_ * d2;
            r[12] = m2_ * d0; r[13] = m2_ * d1; r[14] = m2_ * d2; r[15] = 0.0;
        }
    }
    // max squared row norms of both point sets (the only global quantity the rank certificate needs):
    // non-negative doubles order like their bit patterns, the maximum is order-free
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) {
        m1 = fmax(m1, __shfl_xor(m1, o_)); m2 = fmax(m2, __shfl_xor(m2, o_));
        c1m = fmax(c1m, __shfl_xor(c1m, o_)); c2m = fmax(c2m, __shfl_xor(c2m, o_));
    }
    // this workgroup's four maxima (zero when it holds no correspondence): rs_stage2_kernel folds the workgroups'
    __shared__ double s_mx[4][4];
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_mx[w][0] = m1; s_mx[w][1] = m2; s_mx[w][2] = c1m; s_mx[w][3] = c2m; }
    __syncthreads();
    if (threadIdx.x < 4) sa.bpart[(size_t)block * 4 + threadIdx.x] = fmax(fmax(s_mx[0][threadIdx.x], s_mx[1][threadIdx.x]), fmax(s_mx[2][threadIdx.x], s_mx[3][threadIdx.x]));
}

__device__ __forceinline__ void rs_fit1_body(const StagedArgs& sa, int block) {
    const RansacArgs& a = sa.a;
    const int n = staged_n(a);
    const int p = block * 256 + threadIdx.x;
    if (p >= a.iters) return;
    double T1[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T1[k] = 0.0;
    bool v1 = false;
    double cq1 = 0.0, cq2 = 0.0;
    if (n >= a.m && n >= 3) {
        Pts<false> P{a.p1, a.p2, a.ld, nullptr, n};
        if (a.m == 3) {
            int s[3]; sample3(a, 0, p, n, s);
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double q[6]; P.load(s[j], q);
                A1[j][0] = q[0]; A1[j][1] = q[1]; A1[j][2] = q[2];
                A2[j][0] = q[3]; A2[j][1] = q[4]; A2[j][2] = q[5];
            }
            v1 = fit_3pt(A1, A2, T1);
            // Rank certificate for the refit (estimateTransform.m:11-14 on the INLIER rows): the three sample
            // points are inliers of their own fit, and deleting rows cannot raise a singular value, so
            // sigma_3(pts1(inliers)) >= sigma_3(sample) >= 2 |det| / |A|_F^2 and sigma_2(pts2(inliers)) >=
            // (largest 2x2 minor) / |A|_F.  MATLAB's tolerance is N eps(sigma_1) <= N^1.5 2^-52 max|row|.
            // With a factor 4 to spare the refit may skip the twelve Gram sums and the rank test.
            if (v1 && a.refine) {
                bool all_in = true;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double q[6] = {A1[j][0], A1[j][1], A1[j][2], A2[j][0], A2[j][1], A2[j][2]};
                    all_in = all_in && sqdist(q, T1) < a.thDist;
                }
                const double nn = (double)n, tolf = 4.0 * nn * sqrt(nn) * 2.220446049250313e-16;
                double f1 = 0.0, f2 = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { f1 = fma(A1[j][c], A1[j][c], f1); f2 = fma(A2[j][c], A2[j][c], f2); }
                const double det1 = A1[0][0] * (A1[1][1] * A1[2][2] - A1[1][2] * A1[2][1]) - A1[0][1] * (A1[1][0] * A1[2][2] - A1[1][2] * A1[2][0])
                                  + A1[0][2] * (A1[1][0] * A1[2][1] - A1[1][1] * A1[2][0]);
                double mm = 0.0;
#pragma unroll
                for (int r0 = 0; r0 < 3; ++r0)
#pragma unroll
                    for (int c0 = 0; c0 < 3; ++c0) {
                        const int r1 = (r0 + 1) % 3, c1 = (c0 + 1) % 3;
                        mm = fmax(mm, fabs(A2[r0][c0] * A2[r1][c1] - A2[r0][c1] * A2[r1][c0]));
                    }
                // compared with tolf * max|row| in rs_pass1_body, once the records' maxima are known
                if (all_in && f1 > 0.0 && f2 > 0.0) { cq1 = 2.0 * fabs(det1) / (f1 * tolf); cq2 = mm / (sqrt(f2) * tolf); }
            }
        } else {
            double mom[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) mom[k] = 0.0;
            const int32_t* t = a.sample_idx + (size_t)p * a.m;
            double os[6];
            for (int j = 0; j < a.m; ++j) {
                int idx = min(max(t[j] - 1, 0), n - 1);
                double q[6]; P.load(idx, q);
                if (j == 0) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) os[c] = q[c];
                }
                mom_accumulate(mom, q, os);
            }
            v1 = fit_moments(a.m, mom, os, T1);
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) sa.T1[(size_t)p * 12 + k] = T1[k];
    sa.v1[p] = v1;
    sa.certq[2 * (size_t)p] = cq1; sa.certq[2 * (size_t)p + 1] = cq2;
}

// stage 1 of the staged chain in ONE launch: workgroups [0, n_fit) fit the samples (one hypothesis per lane), the rest
// write the correspondence records; workgroup 0 also clears the refit list's counter (nothing reads it in this launch)
__global__ __launch_bounds__(256) void rs_stage1_kernel(StagedArgs sa, int n_fit) {
    if ((int)blockIdx.x < n_fit) {
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) *sa.n_pass = 0;
            if (threadIdx.x < 64 + 2) ((int32_t*)sa.sel_ctr)[threadIdx.x] = 0;
        }
        rs_fit1_body(sa, blockIdx.x);
    } else {
        rs_records_body(sa, blockIdx.x - n_fit);
    }
}

// ---- after each scoring pass: counts from the point blocks' rows of partial counts --------------------------------------
// after the first pass the thInlr test, the rank certificate and the list of refits (rs_pass1), after the second the
// refined counts and the `has` flags (rs_finish)
__device__ __forceinline__ int rs_sum_parts(const StagedArgs& sa, int h) {
    int c = 0;
#pragma unroll 8
    for (int pb = 0; pb < sa.pb; ++pb) c += sa.part[(size_t)pb * sa.a.iters + h];
    return c;
}
__device__ __forceinline__ void rs_pass1_body(const StagedArgs& sa, int h) {
    const RansacArgs& a = sa.a;
    const int n = staged_n(a);
    const int thInlr = matlab_round_i(a.ratio * (double)n);
    int c = rs_sum_parts(sa, h);
    const bool v = sa.v1[h] != 0;
    if (!v) c = 0;
    const bool pass = v && c >= thInlr;
    a.cnt1[h] = c; a.cnt2[h] = 0;
    sa.pass1[h] = pass;
    // refit path: rank certified from the sample and more than three inliers -> masked record sums
    const bool cert = a.refine && sa.certq[2 * (size_t)h] > sqrt(sa.bounds[0]) * (1.0 + 1e-12) &&
                      sa.certq[2 * (size_t)h + 1] > sqrt(sa.bounds[1]) * (1.0 + 1e-12);
    sa.cert[h] = cert;
    const bool lane_path = sa.use_lane && a.refine && pass && cert && c >= 4;
    sa.dense[h] = pass && !lane_path;
    if (lane_path) sa.pass_list[atomicAdd(sa.n_pass, 1)] = h;
    if (!a.refine) {
        a.has[h] = pass;
        if (pass) {
#pragma unroll
            for (int k = 0; k < 12; ++k) a.TF[(size_t)h * 12 + k] = sa.T1[(size_t)h * 12 + k];
        }
    }
}
__device__ __forceinline__ void rs_finish_body(const StagedArgs& sa, int h) {
    const RansacArgs& a = sa.a;
    const int n = staged_n(a);
    const int thInlr = matlab_round_i(a.ratio * (double)n);
    const int c = rs_sum_parts(sa, h);
    const bool v = sa.v2[h] != 0;
    a.cnt2[h] = v ? c : 0;
    a.has[h] = v && c >= thInlr;
}
// Fusing these two into the scoring launches (the last point block of a chunk running them as an epilogue) was built and
// measured: the arrival counter needs the block's rows of partial counts to be complete first, i.e. a wait for ALL of
// its stores, the 40 MB of inlier masks included, which costs the first pass 25 us where the launch costs 6; and an
// agent-scope fence (__threadfence) writes back / invalidates the XCD's whole L2 on this chip: 5 x slower passes.
__global__ void rs_pass1_kernel(StagedArgs sa) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < sa.a.iters) rs_pass1_body(sa, h);
}
__global__ void rs_finish_kernel(StagedArgs sa) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < sa.a.iters) rs_finish_body(sa, h);
}

// grid (point blocks, hypothesis chunks).  part[pb][h] = inliers of hypothesis h among this block's points.
// EMIT: keep the inlier masks (sa.masks) for the lane-per-hypothesis refit.
template <bool EMIT>
__global__ __launch_bounds__(kSW * 64) void rs_score_kernel(StagedArgs sa, const double* __restrict__ TT,
                                                            const unsigned char* __restrict__ valid) {
    const RansacArgs& a = sa.a;
    __shared__ int s_cnt[kSW][kSChunk];
    const int n = staged_n(a);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h0 = blockIdx.y * kSChunk, h1 = min(a.iters, h0 + kSChunk);
    const double th = a.thDist;
    for (int hl = threadIdx.x; hl < kSW * kSChunk; hl += kSW * 64) (&s_cnt[0][0])[hl] = 0;
    // validity of the chunk's hypotheses as three wave-uniform bit masks (no per-hypothesis load in the loop)
    unsigned long long vmask[(kSChunk + 63) / 64];
#pragma unroll
    for (int k = 0; k < (kSChunk + 63) / 64; ++k) {
        const int h = h0 + k * 64 + lane;
        vmask[k] = __ballot(h < h1 && k * 64 + lane < kSChunk && valid[h] != 0);
    }
    __syncthreads();
    for (int pb = blockIdx.x; pb * kSPts < n; pb += gridDim.x) {       // normally one trip
        double q[kSS][6]; bool act[kSS];
#pragma unroll
        for (int s = 0; s < kSS; ++s) {
            const int i = pb * kSPts + (wave * kSS + s) * 64 + lane;
            act[s] = i < n;
            const int ii = act[s] ? i : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) { q[s][c] = a.p1[ii + (size_t)c * a.ld]; q[s][3 + c] = a.p2[ii + (size_t)c * a.ld]; }
        }
        for (int h = h0; h < h1; ++h) {
            if (!((vmask[(h - h0) >> 6] >> ((h - h0) & 63)) & 1ull)) continue;   // wave-uniform
            double T[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] = TT[(size_t)h * 12 + k];   // uniform address: scalar loads
            pin_translation_vgpr(T);
            int cnt = 0;
            int mine_lo = 0, mine_hi = 0;
#pragma unroll
            for (int s = 0; s < kSS; ++s) {
                const unsigned long long b = __ballot((sqdist(q[s], T) < th) & act[s]);
                cnt += __popcll(b);
                if (EMIT) {                 // ballot s into lane s
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mine_lo) : "s"((int)(unsigned)b), "n"(s));
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mine_hi) : "s"((int)(unsigned)(b >> 32)), "n"(s));
                }
            }
            if (lane == 0) s_cnt[wave][h - h0] += cnt;
            if (EMIT && lane < kSS)         // one 64-byte row segment per (wave, hypothesis)
                sa.masks[(size_t)h * sa.nslots_cap + (size_t)pb * (kSPts / 64) + wave * kSS + lane] =
                    ((unsigned long long)(unsigned)mine_hi << 32) | (unsigned)mine_lo;
        }
    }
    __syncthreads();
    for (int hl = threadIdx.x; hl < h1 - h0; hl += kSW * 64) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < kSW; ++w) c += s_cnt[w][hl];
        sa.part[(size_t)blockIdx.x * a.iters + h0 + hl] = c;
    }
}

// ---- fp32-screened scoring ---------------------------------------------------------------------------------
// calcDists in single precision on coordinates relative to correspondence 0: with a = p1 - o1, b = p2 - o2 and
// t' = t + R o2 - o1 the residual is a - (R b + t').  Rounding analysis (u = 2^-24, A = max|a|, B = max|b|,
// rho = largest row norm of R, tau = max|t'_c|): every component of R b + t' carries at most 5.1 u (B rho + tau)
// (two input roundings per product, one for t', three FMA roundings), so each residual component is off by at
// most e1 + u |r_c| with e1 = u (A + 5.1 (B rho + tau)), and for every true d <= 4 thDist
//     |d32 - d| <= 2 sqrt(3) sqrt(4 thDist) e1 + 3 e1^2 + 5.2 u 4 thDist  =: E0.
// E = 1.5 E0 (the slack also covers the ~1e-16-level difference between the centred and the raw fp64 forms).
// d32 < thDist - E proves an inlier, d32 > thDist + E proves an outlier (for d > 4 thDist the error grows slower
// than d itself once E <= thDist / 2).  A wave that meets ANY distance in between re-scores that hypothesis with
// the fp64 sqdist on the raw coordinates -- so every count and every mask is the fp64 one.  E > thDist / 2 or a
// non-finite E sends everything to fp64.
__device__ __forceinline__ void make_t32(const double (&T)[12], const double (&o)[6], double A, double B, double th,
                                         float* __restrict__ out /*16*/, const bool resident = false /* sqdist32r's operation order */) {
    const double u = 5.9604644775390625e-08;
    double rho = 0.0, tau = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double tp = T[r * 4 + 3] + (T[r * 4] * o[3] + T[r * 4 + 1] * o[4] + T[r * 4 + 2] * o[5]) - o[r];
        rho = fmax(rho, sqrt(T[r * 4] * T[r * 4] + T[r * 4 + 1] * T[r * 4 + 1] + T[r * 4 + 2] * T[r * 4 + 2]));
        tau = fmax(tau, fabs(tp));
        out[r * 3] = (float)T[r * 4]; out[r * 3 + 1] = (float)T[r * 4 + 1]; out[r * 3 + 2] = (float)T[r * 4 + 2];
        out[9 + r] = (float)tp;
    }
    const double e1 = resident ? 6.1 * u * (A + B * rho + tau) : u * (A + 5.1 * (B * rho + tau));
    const double E = 1.5 * (2.0 * 1.7320508075688774 * sqrt(4.0 * th) * e1 + 3.0 * e1 * e1 + 5.2 * u * 4.0 * th);
    float lo = -INFINITY, hi = INFINITY;
    if (E == E && E <= 0.5 * th && th > 0.0 && tau < 1e30 && A < 1e15 && B < 1e15) {
        lo = (float)(th - E); if ((double)lo > th - E) lo = nextafterf(lo, -INFINITY);
        hi = (float)(th + E); if ((double)hi < th + E) hi = nextafterf(hi, INFINITY);
    }
    out[12] = lo; out[13] = hi; out[14] = 0.0f; out[15] = 0.0f;
}

// stage 2: every workgroup folds the record workgroups' maxima (a few hundred doubles: cheaper than a launch of its own),
// workgroup 0 leaves them in sa.bounds for the later stages, and each thread writes the fp32 row of its sample fit
__device__ __forceinline__ void rs_fold_maxima(const StagedArgs& sa, double (&mx)[4]) {
    __shared__ double s_mx[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mx[k] = 0.0;
    for (int b = threadIdx.x; b < sa.n_rec_blocks; b += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmax(mx[k], sa.bpart[(size_t)b * 4 + k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mx[k] = fmax(mx[k], __shfl_xor(mx[k], o_));
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_mx[threadIdx.x >> 6][k] = mx[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) mx[k] = fmax(fmax(s_mx[0][k], s_mx[1][k]), fmax(s_mx[2][k], s_mx[3][k]));
}
__device__ __forceinline__ void rs_digits_body(const StagedArgs& sa, uint4* __restrict__ dig, int block, const double (&bounds)[4]);
// workgroups [0, n_t32): the fp32 rows of the sample fits; the rest: the records' int8 digits (round 4: one launch for both -- the
// digits only need the folded maxima, which every workgroup of this launch computes for itself)
__global__ __launch_bounds__(256) void rs_stage2_kernel(StagedArgs sa, const double* __restrict__ TT, float* __restrict__ out, int n_t32,
                                                        uint4* __restrict__ dig) {
    const RansacArgs& a = sa.a;
    double mx[4];
    rs_fold_maxima(sa, mx);
    if ((int)blockIdx.x >= n_t32) { rs_digits_body(sa, dig, blockIdx.x - n_t32, mx); return; }
    if (blockIdx.x == 0 && threadIdx.x < 4) sa.bounds[threadIdx.x] = mx[threadIdx.x];
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= a.iters || !sa.use_f32) return;
    const int n = staged_n(a);
    if (n < 1) return;
    Pts<false> P{a.p1, a.p2, a.ld, nullptr, n};
    double o[6]; P.load(0, o);
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = TT[(size_t)h * 12 + k];
    make_t32(T, o, sqrt(mx[2]) * (1.0 + 1e-12), sqrt(mx[3]) * (1.0 + 1e-12), a.thDist, out + (size_t)h * 16);
}

__device__ __forceinline__ float sqdist32(const float (&p)[6], const float (&T)[12]) {      // T: R rows (9), t' (3)
    const float tx = __builtin_fmaf(p[3], T[0], __builtin_fmaf(p[4], T[1], __builtin_fmaf(p[5], T[2], T[9])));
    const float ty = __builtin_fmaf(p[3], T[3], __builtin_fmaf(p[4], T[4], __builtin_fmaf(p[5], T[5], T[10])));
    const float tz = __builtin_fmaf(p[3], T[6], __builtin_fmaf(p[4], T[7], __builtin_fmaf(p[5], T[8], T[11])));
    const float dx = p[0] - tx, dy = p[1] - ty, dz = p[2] - tz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// rs_score_kernel with the fp32 screen in front: same grid, same partial counts, same masks.  No branch per slot:
// the eight slots of a hypothesis are screened straight through, the "somebody is in the band" masks are OR-ed,
// and only then does the wave decide whether to redo the hypothesis in fp64.
typedef int rs_i32x16 __attribute__((ext_vector_type(16)));
template <bool EMIT>
__global__ __launch_bounds__(kSW * 64) void rs_score32_kernel(StagedArgs sa, const double* __restrict__ TT,
                                                              const float* __restrict__ T32,
                                                              const unsigned char* __restrict__ valid,
                                                              int32_t* __restrict__ acc /* null: this block's row of partial counts; else counts are ADDED to acc[h] (integers: order-free) */) {
    const RansacArgs& a = sa.a;
    __shared__ int s_cnt[kSW][kSChunk];
    const int n = staged_n(a);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h0 = blockIdx.y * kSChunk, h1 = min(a.iters, h0 + kSChunk);
    const double th = a.thDist;
    for (int hl = threadIdx.x; hl < kSW * kSChunk; hl += kSW * 64) (&s_cnt[0][0])[hl] = 0;
    unsigned long long vmask[(kSChunk + 63) / 64];
#pragma unroll
    for (int k = 0; k < (kSChunk + 63) / 64; ++k) {
        const int h = h0 + k * 64 + lane;
        vmask[k] = __ballot(h < h1 && k * 64 + lane < kSChunk && valid[h] != 0);
    }
    __syncthreads();
    for (int pb = blockIdx.x; pb * kSPts < n; pb += gridDim.x) {       // normally one trip
        float q[kSS][6];
        const int ibase = pb * kSPts + wave * kSS * 64 + lane;
#pragma unroll
        for (int s = 0; s < kSS; ++s) {
            const int i = ibase + s * 64;
            const bool act = i < n;
            const int ii = act ? i : 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) q[s][c] = sa.c32[(size_t)c * sa.n32 + ii];
            // a lane past the end scores NaN: neither screen test holds for it, so the hot loop carries no activity masks
            if (!act) q[s][0] = __builtin_nanf("");
        }
        // One hypothesis = one 64-byte row of T32 (R, t', thlo, thhi), wave-uniform: a scalar load -- and a COLD one (every
        // row is read once per workgroup), ~2000 cycles that five or six waves per SIMD do not cover.  The rows are
        // therefore fetched one hypothesis ahead, into two alternating sets of 16 SGPRs (written as asm: the compiler
        // waits for a scalar load where it issues it).
        rs_i32x16 RA, RB;
#define PCREG_ROW_LOAD(R, H) { const float* rp_ = T32 + (size_t)min((H), h1 - 1) * 16; asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(R) : "s"(rp_)); }
#define PCREG_ROW_WAIT(R) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R))
#define PCREG_SCORE_HYP(H, R)                                                                                      \
        if ((vmask[((H) - h0) >> 6] >> (((H) - h0) & 63)) & 1ull) {               /* wave-uniform */                 \
            float T[12];                                                                                           \
            _Pragma("unroll") for (int k = 0; k < 12; ++k) { const int w_ = R[k]; T[k] = __int_as_float(w_); }                 \
            const int wlo_ = R[12], whi_ = R[13]; const float thlo = __int_as_float(wlo_), thhi = __int_as_float(whi_);          \
            unsigned long long b[kSS], band = 0ull;                                                                \
            _Pragma("unroll") for (int s = 0; s < kSS; ++s) {                                                      \
                const float d = sqdist32(q[s], T);                                                                 \
                b[s] = __ballot(d < thlo);                                                                         \
                band |= __ballot(d <= thhi) ^ b[s];     /* nested sets: they differ iff somebody is in the band */  \
            }                                                                                                      \
            if (band != 0ull) {      /* 5 % of the (hypothesis, 512 correspondences) units on the benchmark: the SLOTS with somebody in */ \
                double T64[12];      /* the band (one in 180) again, fp64 on the raw coordinates -- round 2 redid all eight.  The loop stays  */ \
                _Pragma("unroll") for (int k = 0; k < 12; ++k) T64[k] = TT[(size_t)(H) * 12 + k];   /* rolled (unrolled, its eight branches */ \
                _Pragma("unroll 1") for (int s = 0; s < kSS; ++s) {      /* pushed the kernel from 75 to 171 VGPRs), so a slot's fp32 values are */ \
                    const int i = ibase + s * 64;                        /* read again rather than indexed out of q[][] with a runtime s         */ \
                    const bool act = i < n;                                                                        \
                    const int ii = act ? i : 0;                                                                    \
                    float qq[6];                                                                                   \
                    _Pragma("unroll") for (int c = 0; c < 6; ++c) qq[c] = sa.c32[(size_t)c * sa.n32 + ii];         \
                    if (!act) qq[0] = __builtin_nanf("");                                                          \
                    const float d = sqdist32(qq, T);                                                               \
                    if ((__ballot(d <= thhi) ^ __ballot(d < thlo)) != 0ull) {                /* wave-uniform */     \
                        double p[6];                                                                               \
                        _Pragma("unroll") for (int c = 0; c < 3; ++c) { p[c] = a.p1[ii + (size_t)c * a.ld]; p[3 + c] = a.p2[ii + (size_t)c * a.ld]; } \
                        const unsigned long long bb = __ballot((sqdist(p, T64) < th) & act);                       \
                        _Pragma("unroll") for (int s2 = 0; s2 < kSS; ++s2) if (s2 == s) b[s2] = bb;                \
                    }                                                                                              \
                }                                                                                                  \
            }                                                                                                      \
            int cnt = 0;                                                                                           \
            int mine_lo = 0, mine_hi = 0;                                                                          \
            _Pragma("unroll") for (int s = 0; s < kSS; ++s) {                                                      \
                cnt += __popcll(b[s]);                                                                             \
                if (EMIT) {                 /* ballot s into lane s */                                              \
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mine_lo) : "s"((int)(unsigned)b[s]), "n"(s)); \
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mine_hi) : "s"((int)(unsigned)(b[s] >> 32)), "n"(s)); \
                }                                                                                                  \
            }                                                                                                      \
            if (lane == 0) s_cnt[wave][(H) - h0] += cnt;                                                           \
            if (EMIT && lane < kSS)         /* one 64-byte row segment per (wave, hypothesis) */                   \
                sa.masks[(size_t)(H) * sa.nslots_cap + (size_t)pb * (kSPts / 64) + wave * kSS + lane] =            \
                    ((unsigned long long)(unsigned)mine_hi << 32) | (unsigned)mine_lo;                             \
        }
        // Rule for the inline-asm row loads (the compiler believes an asm output is ready at once): a load's destination is
        // never in flight across the loop's back edge -- RA is waited for at the END of the iteration that issued it, so
        // only waited values are loop-carried (tests/test_isa_lint.py walks the emitted ISA from every such load to its
        // wait, along every path, for an instruction that touches the destination SGPRs).
        PCREG_ROW_LOAD(RA, h0)
        PCREG_ROW_WAIT(RA);
        for (int h = h0; h < h1; h += 2) {
            PCREG_ROW_LOAD(RB, h + 1)
            PCREG_SCORE_HYP(h, RA)
            PCREG_ROW_WAIT(RB); PCREG_ROW_LOAD(RA, h + 2)
            if (h + 1 < h1) { PCREG_SCORE_HYP(h + 1, RB) }
            PCREG_ROW_WAIT(RA);
        }
#undef PCREG_SCORE_HYP
#undef PCREG_ROW_WAIT
#undef PCREG_ROW_LOAD
    }
    __syncthreads();
    for (int hl = threadIdx.x; hl < h1 - h0; hl += kSW * 64) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < kSW; ++w) c += s_cnt[w][hl];
        if (acc) { if (c) atomicAdd(&acc[h0 + hl], c); } else sa.part[(size_t)blockIdx.x * a.iters + h0 + hl] = c;
    }
}

// ================================================================ hypothesis kernel, fp32-screened (LDS-resident)
// ransac_hyp_kernel's arithmetic with rs_score32_kernel's screen in front of both scoring passes (round 4): the registrations
// the reference really runs (n = 170 - 2000 putative matches, completeExperimentFast.m:166,205-206) stay in LDS and were scored
// in fp64 only.  Per workgroup the correspondences are staged ONCE, relative to correspondence 0 (a = p1 - o1, b = p2 - o2):
//   c64[np / 64][6][64]      the fp64 differences (exactly mom_core's d', m': the refit sums read them back),
//   c32[np / 128][6][2][64]  their fp32 roundings (the screen's operands), a pad lane holds NaN and scores nothing,
// np = n rounded up to 128.  A wave owns up to 64 hypotheses (sample fits one per lane as before) and walks them four at a
// time: the four fp32 transforms sit in SGPRs, every lane screens two 64-point slots per trip (15 packed-fp32 operations per
// pair and two compares), the "proven inlier" ballots are counted AND kept -- ballot of slot s in lane s of a VGPR pair per
// hypothesis -- so that the refit pass does not evaluate a single distance: it runs under those ballots as EXEC masks
// (inverse_ballot -> s_and_saveexec) over the fp64 differences, two hypotheses per read of the points, fifteen sums each (the
// rank of the inlier set is certified from the sample as in rs_fit1_body; an uncertified or three-inlier refit takes the old
// 27-sum path on the raw coordinates from L2).  A hypothesis that meets ANY distance inside the band thDist +- E is re-scored
// slot by slot and the slots concerned decided by the fp64 sqdist on the raw coordinates, so counts and masks are the fp64 ones.
// The fifteen wave sums of a refit leave the lanes through two levels of v_permlane32/16_swap and four DPP levels (65 vector
// instructions where 15 shuffle butterflies took 270 and the LDS pipe) and land, one value per lane, in a 128-byte row of
// scratch per hypothesis; the lane-parallel refit reads its row back.  Sample fits and refits are parked in their output
// rows (a.TF) instead of 24 VGPRs.
constexpr int kHB32 = 4;                       // hypotheses per pass over the points
typedef unsigned rs_u32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// (A, B) -> A + B after v_permlane32_swap: lanes 0-31 hold A(l) + A(l + 32), lanes 32-63 hold B(l - 32) + B(l)
__device__ __forceinline__ double swap32_add(double A, double B) {
    const rs_u32x2 r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(A), (unsigned)__double2loint(B), false, false);
    const rs_u32x2 r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(A), (unsigned)__double2hiint(B), false, false);
    return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}
// the same one level down: even 16-lane rows end with A(l) + A(l + 16), odd rows with B(l - 16) + B(l)
__device__ __forceinline__ double swap16_add(double A, double B) {
    const rs_u32x2 r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(A), (unsigned)__double2loint(B), false, false);
    const rs_u32x2 r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(A), (unsigned)__double2hiint(B), false, false);
    return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}
// Wave sums of fifteen values, scattered: lane l returns the sum over the wave of v[k(l)], k(l) = 8 b5 + 4 b4 + 2 b3 + b2 of l
// (k = 15: zero; bits 1 and 0 of the lane do not matter).  The additions of one value form a fixed tree over the lanes, so the
// result depends on the hypothesis alone, not on the wave or rank that owns it.
__device__ __forceinline__ double wave_sum15_scatter(const double (&v)[15]) {
    const int lane = threadIdx.x & 63;
    double r1[8], r2[4], r3[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) r1[j] = swap32_add(v[j], j < 7 ? v[j < 7 ? 8 + j : 14] : 0.0);
#pragma unroll
    for (int j = 0; j < 4; ++j) r2[j] = swap16_add(r1[j], r1[4 + j]);
    const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) r3[j] = (b3 ? r2[2 + j] : r2[j]) + dpp_f64<0x128>(b3 ? r2[j] : r2[2 + j]);             // row_ror:8 = lane ^ 8
    double r = (b2 ? r3[1] : r3[0]) + dpp_f64<0x1B>(dpp_f64<0x141>(b2 ? r3[0] : r3[1]));    // row_half_mirror then quad reverse = lane ^ 4
    r += dpp_f64<0x4E>(r);                                                                    // quad_perm [2,3,0,1] = lane ^ 2
    r += dpp_f64<0xB1>(r);                                                                    // quad_perm [1,0,3,2] = lane ^ 1
    return r;
}
__device__ __forceinline__ void mom15_add(double (&m)[15], const double (&q)[6]) {      // mom_core on stored differences
    m[0] += q[0]; m[1] += q[1]; m[2] += q[2]; m[3] += q[3]; m[4] += q[4]; m[5] += q[5];
    m[6]  = fma(q[3], q[0], m[6]);  m[7]  = fma(q[3], q[1], m[7]);  m[8]  = fma(q[3], q[2], m[8]);
    m[9]  = fma(q[4], q[0], m[9]);  m[10] = fma(q[4], q[1], m[10]); m[11] = fma(q[4], q[2], m[11]);
    m[12] = fma(q[5], q[0], m[12]); m[13] = fma(q[5], q[1], m[13]); m[14] = fma(q[5], q[2], m[14]);
}
// estimateTransform for N > 3 correspondences of certified rank, from the fifteen sums about (o1, o2)
__device__ bool fit_moments15(int N, const double (&mom)[15], const double (&o)[6], double (&T)[12]) {
    if (N < 4) return false;
    const double inv = 1.0 / (double)N;
    const double cdp[3] = {mom[0] * inv, mom[1] * inv, mom[2] * inv};
    const double cmp_[3] = {mom[3] * inv, mom[4] * inv, mom[5] * inv};
    double H[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) H[i][j] = mom[6 + i * 3 + j] - (double)N * cmp_[i] * cdp[j];
    const double cd[3] = {cdp[0] + o[0], cdp[1] + o[1], cdp[2] + o[2]};
    const double cm[3] = {cmp_[0] + o[3], cmp_[1] + o[4], cmp_[2] + o[5]};
    return polar_to_T(H, cd, cm, T);
}
// a value another lane of this wave stored (after a wavefront-scope release): read at the coherent level, not from the CU's L1
__device__ __forceinline__ double ld_coherent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Hyp32Lds { const double* c64; const float* c32; int np; int ns; };
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));

// The screen's arithmetic, written so that every operation reads ONE scalar operand (a VALU instruction of this chip reads one SGPR):
//     u_c = a_c - t'_c;   u_c = fma(b_z, -R_c2, u_c);   u_c = fma(b_y, -R_c1, u_c);   u_c = fma(b_x, -R_c0, u_c);   c = x, y, z
//     d   = fma(u_z, u_z, fma(u_y, u_y, u_x * u_x))
// Rounding: the inputs carry u |a|, u |t'| and 2.01 u |R_c| |b| <= 2.01 u rho B, the subtraction and the three FMAs four roundings of
// a running value bounded by A + tau + rho B, so every u_c is within e1 = 6.1 u (A + rho B + tau) of the exact residual component,
// and |d32 - d| <= 2 sqrt(3) sqrt(d) e1 + 3 e1^2 + 3.1 u d -- make_t32's E with this e1 (its `resident` flag).
__device__ __forceinline__ float sqdist32r(const float (&q)[6], const float (&T)[12]) {      // T: R rows (9), t' (3)
    float ux = q[0] - T[9], uy = q[1] - T[10], uz = q[2] - T[11];
    ux = __builtin_fmaf(q[5], -T[2], ux); uy = __builtin_fmaf(q[5], -T[5], uy); uz = __builtin_fmaf(q[5], -T[8], uz);
    ux = __builtin_fmaf(q[4], -T[1], ux); uy = __builtin_fmaf(q[4], -T[4], uy); uz = __builtin_fmaf(q[4], -T[7], uz);
    ux = __builtin_fmaf(q[3], -T[0], ux); uy = __builtin_fmaf(q[3], -T[3], uy); uz = __builtin_fmaf(q[3], -T[6], uz);
    return __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
}
// sqdist32r of the correspondences of TWO slots (the halves of q[c]) under one hypothesis: fifteen packed-fp32 instructions, the
// transform in six SGPR pairs P = (t'x, t'y) (t'z, R02) (R12, R22) (R01, R11) (R21, R00) (R10, R20), each operation broadcasting one
// half of a pair to both lanes of the packed instruction (op_sel / op_sel_hi) -- the compiler's own packing duplicated every
// coefficient into a pair of its own (96 SGPRs for four hypotheses: spilled, and re-read inside the loop).
__device__ __forceinline__ rs_f32x2 sqdist32r_x2(const rs_f32x2 (&q)[6], const rs_f32x2 (&P)[6]) {
    rs_f32x2 ux, uy, uz, d;
    asm("v_pk_add_f32 %0, %4, %10 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %5, %10 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %6, %11 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %9, %11, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %9, %12, %1 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %9, %12, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %8, %13, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %8, %13, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %8, %14, %2 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %7, %14, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %7, %15, %1 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %7, %15, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
        "v_pk_mul_f32 %3, %0, %0\n\t"
        "v_pk_fma_f32 %3, %1, %1, %3\n\t"
        "v_pk_fma_f32 %3, %2, %2, %3"
        : "=&v"(ux), "=&v"(uy), "=&v"(uz), "=&v"(d)
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "s"(P[0]), "s"(P[1]), "s"(P[2]), "s"(P[3]), "s"(P[4]), "s"(P[5]));
    return d;
}
// own := 2 own + (this lane's bit of the ballot): v_addc_co_u32 shifts and inserts in one instruction
__device__ __forceinline__ void push_bit(unsigned& own, unsigned long long ballot) {
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(own), "=s"(carry_out) : "s"(ballot));
}
// ballot of the lanes' top bits, and w := 2 w: v_add_co_u32's carry-out is the predicate, as a wave mask
__device__ __forceinline__ unsigned long long pop_bit(unsigned& w) {
    unsigned long long carry_out;
    asm("v_add_co_u32 %0, %1, %0, %0" : "+v"(w), "=s"(carry_out));
    return carry_out;
}
__device__ __forceinline__ float rdlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// One screened pass of up to kHB32 hypotheses (lanes hs[k] of T32) over the staged correspondences.  cnt[k]: inliers.
// EMIT: own[k] = this LANE's inlier bits, slot s at bit ns - 1 - s (ns <= 32).  TFwave: the wave's rows of fp64 transforms.
template <bool EMIT>
__device__ __forceinline__ void score32_group(const RansacArgs& a, const Hyp32Lds& L, const double* g1, const double* g2, const int n,
                                              const int lane, const float (&T32)[14], const double* __restrict__ TFwave,
                                              const int (&hs)[kHB32], int (&cnt)[kHB32], unsigned (&own)[kHB32]) {
    rs_f32x2 P[kHB32][6];
    float thlo[kHB32], thhi[kHB32];             // wave-uniform, but kept in VGPRs: the 48 SGPRs of the transforms leave no room for them
    unsigned bandbits = 0u;                     // bit k: hypothesis k met a distance inside its band
#pragma unroll
    for (int k = 0; k < kHB32; ++k) {
        const int h = hs[k];
        P[k][0] = rs_f32x2{rdlane_f(T32[9], h), rdlane_f(T32[10], h)}; P[k][1] = rs_f32x2{rdlane_f(T32[11], h), rdlane_f(T32[2], h)};
        P[k][2] = rs_f32x2{rdlane_f(T32[5], h), rdlane_f(T32[8], h)};  P[k][3] = rs_f32x2{rdlane_f(T32[1], h), rdlane_f(T32[4], h)};
        P[k][4] = rs_f32x2{rdlane_f(T32[7], h), rdlane_f(T32[0], h)};  P[k][5] = rs_f32x2{rdlane_f(T32[3], h), rdlane_f(T32[6], h)};
        thlo[k] = rdlane_f(T32[12], h); thhi[k] = rdlane_f(T32[13], h);
        asm volatile("" : "+v"(thlo[k]), "+v"(thhi[k]));
        cnt[k] = 0; own[k] = 0u;
    }
    for (int s = 0; s < L.ns; s += 2) {
        rs_f32x2 q[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) q[c] = rs_f32x2{L.c32[s * 384 + c * 128 + lane], L.c32[s * 384 + c * 128 + 64 + lane]};      // one address, constant offsets
#pragma unroll
        for (int k = 0; k < kHB32; ++k) {
            const rs_f32x2 d = sqdist32r_x2(q, P[k]);
            const unsigned long long b0 = __ballot(d.x < thlo[k]), b1 = __ballot(d.y < thlo[k]);
            if (((__ballot(d.x <= thhi[k]) ^ b0) | (__ballot(d.y <= thhi[k]) ^ b1)) != 0ull) bandbits |= 1u << k;     // nested sets: they differ iff somebody is in the band
            cnt[k] += __popcll(b0) + __popcll(b1);
            if (EMIT) { push_bit(own[k], b0); push_bit(own[k], b1); }
        }
    }
    // somebody inside the band: that hypothesis again, slot by slot, the slots concerned in fp64 on the raw coordinates
#pragma unroll 1
    for (int k = 0; k < kHB32; ++k) {
        if (!((bandbits >> k) & 1u)) continue;                                    // wave-uniform
        int hk = hs[0];
#pragma unroll
        for (int kk = 1; kk < kHB32; ++kk) hk = kk == k ? hs[kk] : hk;
        float Tk[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) Tk[j] = rdlane_f(T32[j], hk);
        const float lo = rdlane_f(T32[12], hk), hi = rdlane_f(T32[13], hk);
        double T64[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) T64[j] = ld_coherent(TFwave + (size_t)hk * 12 + j);
        int nc = 0; unsigned nown = 0u;
#pragma unroll 1
        for (int s = 0; s < L.ns; ++s) {
            float q[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) q[c] = L.c32[(s >> 1) * 768 + c * 128 + (s & 1) * 64 + lane];
            const float d = sqdist32r(q, Tk);
            unsigned long long bb = __ballot(d < lo);
            if ((__ballot(d <= hi) ^ bb) != 0ull) {                               // wave-uniform
                const int i = s * 64 + lane;
                const bool act = i < n;
                const int ii = act ? i : 0;
                double p[6];
#pragma unroll
                for (int c = 0; c < 3; ++c) { p[c] = g1[ii + (size_t)c * a.ld]; p[3 + c] = g2[ii + (size_t)c * a.ld]; }
                bb = __ballot((sqdist(p, T64) < a.thDist) & act);
            }
            nc += __popcll(bb);
            if (EMIT) push_bit(nown, bb);
        }
#pragma unroll
        for (int kk = 0; kk < kHB32; ++kk) {
            if (kk == k) { cnt[kk] = nc; if (EMIT) own[kk] = nown; }
        }
    }
}

// The refits the fifteen masked sums cannot serve -- the rank of the inlier set is not certified from the sample, or there are
// exactly three inliers (estimateTransform's N == 3 branch needs the points themselves) -- one hypothesis at a time on the raw
// coordinates (L2): rare by construction (a sample whose points are not inliers of their own fit seldom reaches thInlr).  Only
// what is DIFFERENT runs here: the 27-sum pass and the two rank tests of estimateTransform.m:11-14 on lane h, or the search for
// the three inliers.  The fits themselves stay lane-parallel in phase 3: a hypothesis whose rank test passes leaves its fifteen
// sums in its scratch row like a certified one, a three-inlier hypothesis leaves the indices of its points (i3).
// Returns (wave-uniform) 1: the row holds fifteen sums; 2: i3 holds three indices; 0: estimateTransform returns [].
__device__ __forceinline__ int dense_refit_prepare(const RansacArgs& a, const Pts<false>& P, const int n, const int lane, const int h, const int ch,
                                                   const double (&T)[12], const double (&o)[6], double* __restrict__ row, int (&i3)[3]) {
    if (ch == 3) {
        int k = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < n;
            double q[6]; P.load(act ? i : n - 1, q);
            unsigned long long bal = __ballot(act && sqdist(q, T) < a.thDist);
            while (bal) {
                const int Ln = __builtin_ctzll(bal);
                bal &= bal - 1;
                if (k < 3) i3[k] = i0 + Ln;
                ++k;
            }
        }
        return k == 3 ? 2 : 0;
    }
    if (ch < 4) return 0;                               // estimateTransform on fewer than three correspondences: []
    double mom[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) mom[k] = 0.0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool act = i < n;
        double q[6]; P.load(act ? i : n - 1, q);
        if (act && sqdist(q, T) < a.thDist) mom_accumulate(mom, q, o);
    }
    wave_sum27(mom);
    int ok = 0;
    if (lane == h) {
        const double g1[6] = {mom[15], mom[16], mom[17], mom[18], mom[19], mom[20]};
        const double g2[6] = {mom[21], mom[22], mom[23], mom[24], mom[25], mom[26]};
        ok = rank_gram_at_least(g1, ch, 3) && rank_gram_at_least(g2, ch, 2);          // estimateTransform.m:11-14
    }
    ok = __builtin_amdgcn_readlane(ok, h);
    if (ok) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) v = lane == k ? mom[k] : v;
        if (lane < 15) row[lane] = v;
    }
    return ok;
}

template <int NW, bool REFINE>
__device__ __forceinline__ void ransac_hyp32_body(const RansacArgs& a, char* __restrict__ smem, const int b, const int off, const int n) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t hyp0 = (size_t)b * a.iters;
    const int wbase = (blockIdx.x * NW + wave) * a.hpw;
    const double* g1 = a.p1 + off; const double* g2 = a.p2 + off;
    const int np = (n + 127) / 128 * 128;
    double* c64 = (double*)smem;
    float* c32 = (float*)(c64 + (size_t)6 * np);
    double* s_mx = (double*)(c32 + (size_t)6 * np);          // [NW][4]
    if (n < a.m || n < 3) {   // randperm(ptNum)(1:minPtNum) would throw; report nothing found
        const int p = wbase + lane;
        if (wbase < a.iters && lane < min(a.hpw, a.iters - wbase)) { a.cnt1[hyp0 + p] = 0; a.cnt2[hyp0 + p] = 0; a.has[hyp0 + p] = 0; }
        return;
    }
    Pts<false> P{g1, g2, a.ld, nullptr, n};
    // ---- stage the differences, their fp32 roundings and the four maxima the screen and the rank certificate need
    {
        double o[6];
        P.load(0, o);
        double mx[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = threadIdx.x; i < np; i += NW * 64) {
            double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (i < n) {
                double q[6]; P.load(i, q);
#pragma unroll
                for (int c = 0; c < 6; ++c) v[c] = q[c] - o[c];
                mx[0] = fmax(mx[0], q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
                mx[1] = fmax(mx[1], q[3] * q[3] + q[4] * q[4] + q[5] * q[5]);
                mx[2] = fmax(mx[2], v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                mx[3] = fmax(mx[3], v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
            }
            // slot-major layouts: c64[slot][6][64], c32[slot pair][6][2][64] -- a pass over the points moves ONE address, the six
            // coordinates (and the pair's second slot) sit at constant offsets of it
            const int sl = i >> 6, ln = i & 63;
#pragma unroll
            for (int c = 0; c < 6; ++c) { c64[sl * 384 + c * 64 + ln] = v[c]; c32[(sl >> 1) * 768 + c * 128 + (sl & 1) * 64 + ln] = (float)v[c]; }
            if (i >= n) c32[(sl >> 1) * 768 + (sl & 1) * 64 + ln] = __builtin_nanf("");          // a pad scores NaN: neither screen test holds for it
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1) mx[k] = fmax(mx[k], __shfl_xor(mx[k], o_));
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_mx[wave * 4 + k] = mx[k];
        }
    }
    __syncthreads();
    if (wbase >= a.iters) return;
    // the workgroup's maxima: sqrt(max |p1 row|^2), sqrt(max |p2 row|^2) (rank certificate), the same of the differences (screen).
    // Folded where they are used (three places) rather than kept in eight SGPRs across the scoring loops; likewise correspondence 0.
    auto bound = [&](int k) {
        double m = s_mx[k];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmax(m, s_mx[w * 4 + k]);
        return sqrt(m) * (1.0 + 1e-12);
    };
    const Hyp32Lds L{c64, c32, np, np / 64};
    const int nh = min(a.hpw, a.iters - wbase);
    const int p = wbase + lane;
    const bool mine = lane < nh;
    const int thInlr = matlab_round_i(a.ratio * (double)n);                   // ransac.m:28
    double* const TFwave = a.TF + (hyp0 + wbase) * 12;                         // this wave's output rows: lane h's transform in row h
    double* const MSwave = a.msc + (hyp0 + wbase) * 16;                        // ... and its refit sums

    // ---- phase 0: minimal-sample fit, one hypothesis per lane (ransac.m:42-45); the fit is parked in its output row
    float T32[14];
    bool v1 = false, cert = false;
    {
        double T1[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T1[k] = 0.0;
        if (mine) {
            if (a.m == 3) {
                int s[3]; sample3(a, b, p, n, s);
                double A1[3][3], A2[3][3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double q[6]; P.load(s[j], q);
                    A1[j][0] = q[0]; A1[j][1] = q[1]; A1[j][2] = q[2];
                    A2[j][0] = q[3]; A2[j][1] = q[4]; A2[j][2] = q[5];
                }
                // the rank certificate of rs_fit1_body / rs_pass1_body; its two quantities are taken BEFORE the fit so that the
                // eighteen coordinates do not stay live across it (they are read again for the "sample points are inliers" test)
                bool cq = false;
                if (REFINE) {
                    const double nn = (double)n, tolf = 4.0 * nn * sqrt(nn) * 2.220446049250313e-16;
                    double f1 = 0.0, f2 = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int c = 0; c < 3; ++c) { f1 = fma(A1[j][c], A1[j][c], f1); f2 = fma(A2[j][c], A2[j][c], f2); }
                    const double det1 = A1[0][0] * (A1[1][1] * A1[2][2] - A1[1][2] * A1[2][1]) - A1[0][1] * (A1[1][0] * A1[2][2] - A1[1][2] * A1[2][0])
                                      + A1[0][2] * (A1[1][0] * A1[2][1] - A1[1][1] * A1[2][0]);
                    double mm = 0.0;
#pragma unroll
                    for (int r0 = 0; r0 < 3; ++r0)
#pragma unroll
                        for (int c0 = 0; c0 < 3; ++c0) {
                            const int r1 = (r0 + 1) % 3, c1_ = (c0 + 1) % 3;
                            mm = fmax(mm, fabs(A2[r0][c0] * A2[r1][c1_] - A2[r0][c1_] * A2[r1][c0]));
                        }
                    cq = f1 > 0.0 && f2 > 0.0 && 2.0 * fabs(det1) / (f1 * tolf) > bound(0) && mm / (sqrt(f2) * tolf) > bound(1);
                }
                v1 = fit_3pt(A1, A2, T1);
                if (v1 && cq) {
                    bool all_in = true;
#pragma unroll 1
                    for (int j = 0; j < 3; ++j) {
                        double q[6]; P.load(s[j], q);
                        all_in = all_in && sqdist(q, T1) < a.thDist;
                    }
                    cert = all_in;
                }
            } else {   // minPtNum > 3: general estimateTransform path on the sample
                double mom[27];
#pragma unroll
                for (int k = 0; k < 27; ++k) mom[k] = 0.0;
                const int32_t* t = a.sample_idx + ((size_t)hyp0 + p) * a.m;
                double os[6];
                for (int j = 0; j < a.m; ++j) {
                    int idx = min(max(t[j] - 1, 0), n - 1);
                    double q[6]; P.load(idx, q);
                    if (j == 0) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) os[c] = q[c];
                    }
                    mom_accumulate(mom, q, os);
                }
                v1 = fit_moments(a.m, mom, os, T1);
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) TFwave[(size_t)lane * 12 + k] = T1[k];
        }
        float t16[16];
        double o[6];
        P.load(0, o);
        make_t32(T1, o, bound(2), bound(3), a.thDist, t16, true);
#pragma unroll
        for (int k = 0; k < 14; ++k) T32[k] = t16[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

    // ---- phases 1 + 2, four hypotheses at a time: screened scores (ransac.m:48-50), then the refit sums of those that pass (:53-55)
    int c1 = 0;
    bool onlane = false;                       // this lane's hypothesis has its fifteen sums in MSwave
    int same_as = -1;                          // ... which are those of this earlier hypothesis of the wave (the same inlier set): its refit is this one's
    const int v1c = (v1 ? 1 : 0) | (cert ? 2 : 0);
    for (int h = 0; h < nh; h += kHB32) {
        int hs[kHB32], cnt[kHB32];
        unsigned own[kHB32];
#pragma unroll
        for (int k = 0; k < kHB32; ++k) hs[k] = min(h + k, nh - 1);
        score32_group<REFINE>(a, L, g1, g2, n, lane, T32, TFwave, hs, cnt, own);
        bool lp[kHB32];
#pragma unroll
        for (int k = 0; k < kHB32; ++k) {
            if (lane == h + k) c1 = cnt[k];
            lp[k] = REFINE && h + k < nh && __builtin_amdgcn_readlane(v1c, hs[k]) == 3 && cnt[k] >= thInlr && cnt[k] >= 4;
            if (lp[k] && lane == hs[k]) onlane = true;
        }
        if (!REFINE) continue;
#ifndef NO_MOM
        // Hypotheses of the group that selected the SAME inliers have the same fifteen sums (in a sphere that holds the surface most
        // good samples do: half of this launch's time on the sweep went into summing them again and again): rep[k] = the first
        // hypothesis of the group with k's inlier set; only representatives are summed, the others receive the representative's sums.
        int rep[kHB32];
#pragma unroll
        for (int k = 0; k < kHB32; ++k) {
            rep[k] = k;
#pragma unroll
            for (int j = k - 1; j >= 0; --j)
                if (lp[k] && lp[j] && __builtin_amdgcn_ballot_w64(own[k] != own[j]) == 0ull) rep[k] = j;          // wave-uniform
        }
#pragma unroll
        for (int k = 0; k < kHB32; ++k) if (rep[k] != k) { if (lane == hs[k]) same_as = hs[rep[k]]; lp[k] = false; }          // summed through its representative
#pragma unroll
        for (int kp = 0; kp < kHB32; kp += 2) {
            if (!(lp[kp] || lp[kp + 1])) continue;                               // wave-uniform
            double accA[15], accB[15];
#pragma unroll
            for (int e = 0; e < 15; ++e) { accA[e] = 0.0; accB[e] = 0.0; }
            // this lane's inlier bits, slot 0 at the top: every v_add_co_u32 hands the next slot's predicate out as a wave mask
            unsigned wA = lp[kp] ? own[kp] << (32 - L.ns) : 0u, wB = lp[kp + 1] ? own[kp + 1] << (32 - L.ns) : 0u;
            for (int s = 0; s < L.ns; ++s) {
                double q[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) q[c] = L.c64[s * 384 + c * 64 + lane];
                const unsigned long long wa = pop_bit(wA), wb = pop_bit(wB);
                if (__builtin_amdgcn_inverse_ballot_w64(wa)) mom15_add(accA, q);
                if (__builtin_amdgcn_inverse_ballot_w64(wb)) mom15_add(accB, q);
            }
            const int kq = 8 * ((lane >> 5) & 1) + 4 * ((lane >> 4) & 1) + 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1);
            if (lp[kp]) {
                const double r = wave_sum15_scatter(accA);
                if ((lane & 3) == 0 && kq < 15) {
#pragma unroll
                    for (int k = kp; k < kHB32; ++k) if (rep[k] == kp) MSwave[(size_t)hs[k] * 16 + kq] = r;          // itself and its copies
                }
            }
            if (lp[kp + 1]) {
                const double r = wave_sum15_scatter(accB);
                if ((lane & 3) == 0 && kq < 15) {
#pragma unroll
                    for (int k = kp + 1; k < kHB32; ++k) if (rep[k] == kp + 1) MSwave[(size_t)hs[k] * 16 + kq] = r;
                }
            }
        }
#endif
    }
    if (!v1) c1 = 0;   // empty transform: scores 0 (deviation documented in DESIGN.md)
    const bool pass1 = mine && v1 && c1 >= thInlr;                              // ransac.m:53
    if (!REFINE) {                                                              // ransac.m:62-64: the sample fit is already in its row
        if (mine) { a.cnt1[hyp0 + p] = c1; a.cnt2[hyp0 + p] = 0; a.has[hyp0 + p] = pass1; }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

    // ---- phase 3: refit, one hypothesis per lane (estimateTransform on the inliers)
    bool v2 = false;
    {
        double o[6];
        P.load(0, o);
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] = rdlane(o[c], 0);
#ifdef NO_DENSE
        unsigned long long dense = 0;
#else
        unsigned long long dense = __ballot(pass1 && !onlane);
#endif
        int i3[3] = {-1, -1, -1};         // this lane's three-inlier refit: the indices of its points
        while (dense) {                   // rank not certified from the sample, or the N == 3 branch
            const int h = __builtin_ctzll(dense);
            dense &= dense - 1;
            double T[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] = rdlane(ld_coherent(TFwave + (size_t)h * 12 + k), 0);
            int j3[3] = {-1, -1, -1};
            const int kind = dense_refit_prepare(a, P, n, lane, h, __builtin_amdgcn_readlane(c1, h), T, o, MSwave + (size_t)h * 16, j3);
            if (lane == h) {
                if (kind == 1) onlane = true;
                if (kind == 2) { i3[0] = j3[0]; i3[1] = j3[1]; i3[2] = j3[2]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        double T2[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T2[k] = 0.0;
        if (onlane) {
            double mom[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) mom[k] = ld_coherent(MSwave + (size_t)lane * 16 + k);
            v2 = fit_moments15(c1, mom, o, T2);
        } else if (i3[0] >= 0) {
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double q[6]; P.load(i3[j], q);
                A1[j][0] = q[0]; A1[j][1] = q[1]; A1[j][2] = q[2];
                A2[j][0] = q[3]; A2[j][1] = q[4]; A2[j][2] = q[5];
            }
            v2 = fit_3pt(A1, A2, T2);
        }
        if (v2) {
#pragma unroll
            for (int k = 0; k < 12; ++k) TFwave[(size_t)lane * 12 + k] = T2[k];
        }
        float t16[16];
        make_t32(T2, o, bound(2), bound(3), a.thDist, t16, true);
#pragma unroll
        for (int k = 0; k < 14; ++k) T32[k] = t16[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

    // ---- phase 4: rescore the refined transforms (ransac.m:56-58)
    int c2 = 0;
    unsigned long long mask = __ballot(v2 && same_as < 0);          // a copy has its representative's sums, hence its transform and its count
    while (mask) {
        int hs[kHB32], cnt[kHB32];
        unsigned own[kHB32];
#pragma unroll
        for (int k = 0; k < kHB32; ++k) {
            hs[k] = mask ? __builtin_ctzll(mask) : hs[k > 0 ? k - 1 : 0];
            if (mask) mask &= mask - 1;
        }
        score32_group<false>(a, L, g1, g2, n, lane, T32, TFwave, hs, cnt, own);
#pragma unroll
        for (int k = 0; k < kHB32; ++k) if (lane == hs[k]) c2 = cnt[k];
    }
    {
        const int c2r = __shfl(c2, max(same_as, 0));
        if (same_as >= 0) c2 = c2r;
    }
    if (mine) {
        const bool keep = v2 && c2 >= thInlr;                                   // ransac.m:59-61
        a.cnt1[hyp0 + p] = c1; a.cnt2[hyp0 + p] = v2 ? c2 : 0; a.has[hyp0 + p] = keep;
    }
}

template <int NW, bool REFINE>
__global__ __launch_bounds__(NW * 64) void ransac_hyp32_kernel(RansacArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem32[];
    const int b = blockIdx.y;
    const int off = a.offsets ? a.offsets[b] : 0;
    int n = a.offsets ? (a.offsets[b + 1] - off) : (a.n_dev ? *a.n_dev : a.n_cap);
    n = min(n, a.n_cap);
    if (n <= a.n_lo) return;                                    // a smaller launch class serves it
    if (n <= a.n_hi) ransac_hyp32_body<NW, REFINE>(a, smem32, b, off, n);   // above every class: ransac_hyp_kernel, fp64 from L2 (a launch of its own)
}

// Refit moments: the tiled kernel's phase 2, two passing hypotheses of every wave per sweep.
__global__ __launch_bounds__(kTBlock) void rs_moments_kernel(StagedArgs sa) {
    const RansacArgs& a = sa.a;
    __shared__ __attribute__((aligned(16))) double sp[2 * 6 * kTile];
    const int n = staged_n(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wbase = (blockIdx.x * kTW + wave) * a.hpw;
    const double* g1 = a.p1; const double* g2 = a.p2;
    const int nh = max(0, min(a.hpw, a.iters - wbase));
    if (n < a.m || n < 3) return;                                   // block-uniform
    Pts<false> P{g1, g2, a.ld, nullptr, n};
    double o[6];
    P.load(0, o);
    const double th = a.thDist;
    // lane l < nh owns hypothesis wbase + l: its transform and inlier count
    double T1[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T1[k] = 0.0;
    int c1 = 0; bool pass = false; int certified = 0;
    if (lane < nh) {
        const int h = wbase + lane;
        pass = sa.dense[h] != 0; c1 = a.cnt1[h]; certified = sa.cert[h];
#pragma unroll
        for (int k = 0; k < 12; ++k) T1[k] = sa.T1[(size_t)h * 12 + k];
    }
    unsigned long long todo = __ballot(pass);
    while (__syncthreads_or(todo != 0)) {                           // every wave of the workgroup sweeps together
        int hs[2]; bool want[2], gram[2]; int ch[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            want[k] = todo != 0;
            hs[k] = want[k] ? __builtin_ctzll(todo) : 0;
            if (want[k]) todo &= todo - 1;
            ch[k] = __builtin_amdgcn_readlane(c1, hs[k]);
            gram[k] = __builtin_amdgcn_readlane(certified, hs[k]) == 0;      // wave-uniform: the Gram sums only feed the rank test
        }
        double T[2][12];
#pragma unroll
        for (int k = 0; k < 2; ++k) { bcast_T(T1, hs[k], T[k]); pin_translation_vgpr(T[k]); }
        double acc[2][27];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 27; ++e) acc[k][e] = 0.0;
        int k3[2] = {0, 0};
        const bool any3 = (want[0] && ch[0] == 3) || (want[1] && ch[1] == 3);
        PCREG_TILE_SWEEP_BEGIN
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool in = (sqdist(q, T[k]) < th) & act & want[k];
                if (in) { mom_core(acc[k], q, o); if (gram[k]) mom_gram(acc[k], q); }
                if (any3 && ch[k] == 3 && want[k]) {     // estimateTransform's N == 3 branch needs the points themselves
                    unsigned long long bal = __ballot(in);
                    const int r3 = k3[k] + __popcll(bal & ((1ull << lane) - 1ull));
                    if (in && r3 < 3) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) sa.mom[(size_t)(wbase + hs[k]) * 27 + r3 * 6 + c] = q[c];
                    }
                    k3[k] += __popcll(bal);
                }
            }
        PCREG_TILE_SWEEP_END
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (want[k] && ch[k] != 3) {
                wave_sum27(acc[k]);
#pragma unroll
                for (int e = 0; e < 27; ++e) if (lane == 0) sa.mom[(size_t)(wbase + hs[k]) * 27 + e] = acc[k][e];
            }
        }
    }
}

typedef int rs_i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double rs_pair(int lo, int hi) { return __builtin_bit_cast(double, rs_i32x2{lo, hi}); }

// ---- refit moments on the matrix cores -------------------------------------------------------------------------------
// A hypothesis' fifteen moments are sums of per-correspondence records under its inlier mask: [hypotheses x points] 0/1 times
// [points x 15] -- a matrix product whose left factor is EXACT in any number format.  The records are put on a common
// fixed-point grid per column (2^E_c above the column's bound) and cut into eight balanced base-128 digits (int8); the
// masks expand to 0/1 bytes; v_mfma_i32_32x32x32_i8 adds digit columns in int32 without any rounding; the eight digit sums
// of a column are recombined in fp64 (two exact 4-digit halves, ONE rounding).  The result is the correctly rounded
// masked sum of records truncated at 2^-57 of their column's bound: closer to the true sum than fp64 adds in any order
// (whose error grows with the 32 k terms), at the int8 matrix rate instead of 15 fp64 FMAs per pair -- the lane-per-
// hypothesis fp64 kernel this replaces took 247 us of the step's 2.2 ms.
// eight digits per record column; digit p of column c sits in tile p % 4, column 2 c + p / 4 of the B operand
constexpr int kMmRows = 64;                   // hypotheses per wave (two 32-row tiles)
constexpr int kMmWaves = 4;                   // waves per workgroup: 256 hypotheses share every tile of digits
typedef int rs_v4i __attribute__((ext_vector_type(4)));
typedef int rs_v16i __attribute__((ext_vector_type(16)));

// exponent of the fixed-point grid of record column c: |record| < 2^(E - 1)  (bounds[2], [3] = max |p1 - o1|^2, |p2 - o2|^2)
__device__ __forceinline__ int rs_rec_exp(const double* __restrict__ bounds, int c) {
    const double D = sqrt(bounds[2]), M = sqrt(bounds[3]);
    double b = c < 3 ? D : (c < 6 ? M : D * M);
    b *= 1.0000001;                           // the records' own rounding
    return (b > 0.0 && b < INFINITY) ? ilogb(b) + 2 : 0;
}
// dig [(kstep * 4 + tile) * 64 + lane] (16 int8): lane = (half << 5) | col; byte b = correspondence kstep * 32 + half * 16 + b
__device__ __forceinline__ void rs_digits_body(const StagedArgs& sa, uint4* __restrict__ dig, int block, const double (&bounds)[4]) {
    const int g = block * 256 + threadIdx.x;                    // (kstep, tile, lane)
    const int lane = g & 63, tile = (g >> 6) & 3, kstep = g >> 8;
    if (kstep >= sa.nslots_cap * 2) return;
    const int col = lane & 31, half = lane >> 5;
    const int c = col >> 1, pd = tile + 4 * (col & 1);
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (c < 15) {
        const int E = rs_rec_exp(bounds, c);
        const double* r = sa.rec + (size_t)(kstep * 32 + half * 16) * kRec + c;
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) {
            double y = ldexp(r[(size_t)bb * kRec], -E);         // exact; |y| < 1/2
            double d = 0.0;
            for (int q = 0; q <= pd; ++q) { y *= 128.0; d = rint(y); y -= d; }       // every step exact: |d| <= 64
            w[bb >> 2] |= ((unsigned)(int)d & 0xFFu) << (8 * (bb & 3));
        }
    }
    dig[g] = make_uint4(w[0], w[1], w[2], w[3]);
}

// grid (groups of 256 listed hypotheses, chunks of kMomSlots * 64 correspondences); mpart [chunk][iters][15]
__global__ __launch_bounds__(kMmWaves * 64, 2) void rs_moments_mfma_kernel(StagedArgs sa, const uint4* __restrict__ dig,
                                                                            const unsigned long long* __restrict__ masks) {
    __shared__ __attribute__((aligned(16))) uint4 s_b[2][16 * 64];               // two stages of 4 k-steps x 4 tiles x 64 lanes
    const RansacArgs& a = sa.a;
    const int n = staged_n(a), np = *sa.n_pass;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x * kMmWaves * kMmRows >= np) return;
    const int nslots = (n + 63) >> 6;
    const int s0 = (int)blockIdx.y * kMomSlots;
    if (s0 >= nslots) return;
    const int s1 = min(nslots, s0 + kMomSlots);
    const int hrow = lane & 31, half = lane >> 5;
    int h[2]; bool live[2]; const unsigned long long* row[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int li = (blockIdx.x * kMmWaves + wave) * kMmRows + r * 32 + hrow;
        live[r] = li < np;
        h[r] = live[r] ? sa.pass_list[li] : sa.pass_list[0];
        row[r] = masks + (size_t)h[r] * sa.nslots_cap;
    }
    rs_v16i acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][t][v] = 0;
    const int nst = (s1 - s0 + 1) >> 1;                          // stages of two slots = four k-steps (kMomSlots is even)
    const uint4* dsrc = dig + (size_t)s0 * 2 * 4 * 64 + tid;    // a stage = 4 k-steps x 256 uint4; thread tid moves element k * 256 + tid
    uint4 pre0 = dsrc[0], pre1 = dsrc[256], pre2 = dsrc[512], pre3 = dsrc[768];
    s_b[0][tid] = pre0; s_b[0][256 + tid] = pre1; s_b[0][512 + tid] = pre2; s_b[0][768 + tid] = pre3;
    __syncthreads();
    // a stage's mask bits: two 64-bit words per row = four k-steps of 32 correspondences (lane half takes 16 of each); they come
    // from HBM (40 MB per registration) one row per lane, so the next stage's are requested before this stage's products
    uint4 mw[2], mwn[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) { mw[r] = *(const uint4*)(row[r] + s0); mwn[r] = mw[r]; }
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) {
            const uint4* nx = dsrc + (size_t)(st + 1) * 1024;
            pre0 = nx[0]; pre1 = nx[256]; pre2 = nx[512]; pre3 = nx[768];
#pragma unroll
            for (int r = 0; r < 2; ++r) mwn[r] = *(const uint4*)(row[r] + s0 + 2 * (st + 1));
        }
        const uint4* B = s_b[st & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rs_v4i A[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned wj = j == 0 ? mw[r].x : (j == 1 ? mw[r].y : (j == 2 ? mw[r].z : mw[r].w));
                const unsigned b16 = (wj >> (16 * half)) & 0xFFFFu;
#pragma unroll
                for (int q = 0; q < 4; ++q) A[r][q] = (int)((((b16 >> (4 * q)) & 0xFu) * 0x00204081u) & 0x01010101u);   // 4 bits -> 4 bytes of 0 / 1
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4 bv = B[(j * 4 + t) * 64 + lane];
                const rs_v4i Bv = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r], Bv, acc[r][t], 0, 0, 0);
            }
        }
        if (st + 1 < nst) {
            uint4* nb = s_b[(st + 1) & 1];
            nb[tid] = pre0; nb[256 + tid] = pre1; nb[512 + tid] = pre2; nb[768 + tid] = pre3;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) mw[r] = mwn[r];
        __syncthreads();
    }
    // digits -> doubles.  A lane holds column `hrow` of the four tiles for 16 rows: digits p = 0..3 (even column) or 4..7 (odd)
    // of record column hrow / 2; the odd neighbour's half is 128^-4 times smaller.  The sums go through LDS (the digit stages are
    // done with) so that a hypothesis' fifteen doubles leave as one 120-byte run.
    const int c = hrow >> 1;
    const int E = c < 15 ? rs_rec_exp(sa.bounds, c) : 0;
    double* s_out = (double*)&s_b[0][0] + (size_t)wave * (kMmRows * 15);             // 4 waves x 64 rows x 15 doubles = 30 KB of the 32
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            // exact: four integers below 2^22 on a 2^-7 ladder span 43 bits
            double part = (((double)acc[r][3][v] * 0.0078125 + (double)acc[r][2][v]) * 0.0078125 + (double)acc[r][1][v]) * 0.0078125 + (double)acc[r][0][v];
            const double other = __shfl_xor(part, 1);
            const int rr = (v >> 2) * 8 + half * 4 + (v & 3);                     // row of this accumulator element
            if ((hrow & 1) == 0 && c < 15)
                s_out[(r * 32 + rr) * 15 + c] = ldexp(part + other * 3.7252902984619140625e-09, E - 7);   // (hi + lo * 128^-4) * 2^E / 128
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < kMmRows * 15; e += 64) {
        const int rw = e / 15, cc = e - rw * 15;
        const int li = (blockIdx.x * kMmWaves + wave) * kMmRows + rw;
        if (li < np) sa.mpart[((size_t)blockIdx.y * a.iters + sa.pass_list[li]) * 15 + cc] = s_out[e];
    }
}


__global__ __launch_bounds__(64) void rs_fit2_kernel(StagedArgs sa) {
    const RansacArgs& a = sa.a;
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= a.iters) return;
    bool v2 = false;
    double T2[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T2[k] = 0.0;
    if (sa.pass1[h]) {
        const int n = staged_n(a);
        Pts<false> P{a.p1, a.p2, a.ld, nullptr, n};
        double o[6]; P.load(0, o);
        double mom[27];
        const int c1 = a.cnt1[h];
        if (!sa.dense[h]) {                         // the chunk partials of rs_moments_mfma_kernel, in chunk order
#pragma unroll
            for (int e = 0; e < 27; ++e) mom[e] = 0.0;
            const int nch = (((n + 63) >> 6) + kMomSlots - 1) / kMomSlots;
#pragma unroll 8
            for (int ch = 0; ch < nch; ++ch) {
                const double* pp = sa.mpart + ((size_t)ch * a.iters + h) * 15;
#pragma unroll
                for (int e = 0; e < 15; ++e) mom[e] += pp[e];
            }
            v2 = fit_moments(c1, mom, o, T2, true);
        } else if (c1 == 3) {
#pragma unroll
            for (int e = 0; e < 27; ++e) mom[e] = sa.mom[(size_t)h * 27 + e];
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) { A1[j][c] = mom[j * 6 + c]; A2[j][c] = mom[j * 6 + 3 + c]; }
            v2 = fit_3pt(A1, A2, T2);
        } else {
#pragma unroll
            for (int e = 0; e < 27; ++e) mom[e] = sa.mom[(size_t)h * 27 + e];
            v2 = fit_moments(c1, mom, o, T2, sa.cert[h] != 0);
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) a.TF[(size_t)h * 12 + k] = T2[k];
    sa.v2[h] = v2;
    if (sa.use_f32 && v2) {
        const int n32_ = staged_n(a);
        Pts<false> P32{a.p1, a.p2, a.ld, nullptr, n32_};
        double o32[6]; P32.load(0, o32);
        make_t32(T2, o32, sqrt(sa.bounds[2]) * (1.0 + 1e-12), sqrt(sa.bounds[3]) * (1.0 + 1e-12), a.thDist, sa.T32b + (size_t)h * 16);
    }
}

// ---------------------------------------------------------------- winner + outputs
// ransac.m:69-98.  One workgroup per registration.
// result struct + inlierIdx = find(dist < thDist), ascending, 1-based (ransac.m:75-98), by one workgroup
template <int NTHR>
__device__ void ransac_emit_result(const RansacArgs& a, int n, int off, pcreg_dev_ransac_result* r, int32_t* inlier_idx,
                                   bool failed, const double* s_T /*LDS, 12*/, int ns, int maxInl, int winner,
                                   int* s_base, int* s_wcnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) *s_base = 0;
    __syncthreads();
    if (threadIdx.x < 16) {   // column-major 4x4: T(k,j) = R(j,k), T(4,j) = t(j)
        int k = threadIdx.x & 3, j = threadIdx.x >> 2;
        double v = 0.0;
        if (!failed) v = (j < 3) ? s_T[j * 4 + k] : (k == 3 ? 1.0 : 0.0);
        r->T[k + 4 * j] = v;
    }
    if (threadIdx.x == 0) {
        r->failed = failed; r->num_success = failed ? 0 : ns; r->max_inliers = failed ? 0 : maxInl;
        r->n = n; r->winner = winner;
    }
    if (failed) { if (threadIdx.x == 0) r->n_inliers = 0; return; }
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = s_T[k];
    Pts<false> P{a.p1 + off, a.p2 + off, a.ld, nullptr, n};
    int32_t* dst = inlier_idx + off;
    // Ordered compaction in rounds of 32 sweeps: sweep it of the round tests points i0 + it * NTHR + tid (coalesced),
    // every wave parks its ballot; one scan over the (sweep, wave) counts; then the lanes write.  Two barriers
    // per round instead of three per sweep.
    constexpr int NW = NTHR / 64;
    __shared__ unsigned long long s_bal[32][NW];
    __shared__ int s_pre[32 * NW];
    for (int i0 = 0; i0 < n; i0 += 32 * NTHR) {
        unsigned mine = 0u;
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int i = i0 + it * NTHR + threadIdx.x;
            const bool act = i < n;
            double q[6]; P.load(act ? i : n - 1, q);
            const bool in = act && sqdist(q, T) < a.thDist;
            const unsigned long long bal = __ballot(in);
            if (lane == 0) s_bal[it][wave] = bal;
            mine |= in ? (1u << it) : 0u;
        }
        __syncthreads();
        // exclusive scan of the 32 * NW counts in (sweep, wave) order: thread t owns entries [t * E, t * E + E)
        constexpr int E = (32 * NW + NTHR - 1) / NTHR;           // 1 for NTHR >= 512... entries per thread
        int cnt[E], tot = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = threadIdx.x * E + e;
            cnt[e] = k < 32 * NW ? __popcll(s_bal[k / NW][k % NW]) : 0;
            tot += cnt[e];
        }
        int incl = tot;                                          // inclusive scan across the workgroup
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) s_wcnt[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wcnt[w];
        int run = *s_base + wbase + incl - tot;
#pragma unroll
        for (int e = 0; e < E; ++e) { const int k = threadIdx.x * E + e; if (k < 32 * NW) s_pre[k] = run; run += cnt[e]; }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            if (mine & (1u << it))
                dst[s_pre[it * NW + wave] + __popcll(s_bal[it][wave] & ((1ull << lane) - 1ull))] = i0 + it * NTHR + threadIdx.x + 1;
        }
        __syncthreads();
        if (threadIdx.x == NTHR - 1) *s_base = run;              // the last thread's running total covers the round
        __syncthreads();
    }
    if (threadIdx.x == 0) r->n_inliers = *s_base;
}

// part != nullptr: hypotheses split over ranks -- leave this share's best (global-index key, success
// count, winner's transform) for the cross-rank combine instead of a result.
template <int NTHR>
__global__ __launch_bounds__(NTHR) void ransac_select_kernel(RansacArgs a, pcreg_dev_ransac_result* out,
                                                               int32_t* inlier_idx, pcreg_dev_ransac_part* part) {
    __shared__ unsigned long long s_key[(NTHR / 64)];
    __shared__ int s_cnt[(NTHR / 64)];
    __shared__ double s_T[12];
    __shared__ int s_base;
    __shared__ int s_wcnt[(NTHR / 64)];
    const int b = blockIdx.x;
    const int off = a.offsets ? a.offsets[b] : 0;
    int n = a.offsets ? (a.offsets[b + 1] - off) : (a.n_dev ? *a.n_dev : a.n_cap);
    n = min(n, a.n_cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t hyp0 = (size_t)b * a.iters;
    const int32_t* cc = a.refine ? a.cnt2 + hyp0 : a.cnt1 + hyp0;               // :69-73
    const int thInlr = matlab_round_i(a.ratio * (double)n);
    // first index of the maximum: max over (count << 32 | ~index), index = GLOBAL hypothesis number
    unsigned long long key = 0; int ns = 0;
    for (int p = threadIdx.x; p < a.iters; p += NTHR) {
        unsigned long long k = ((unsigned long long)(unsigned)cc[p] << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(p + a.hyp0g));
        key = k > key ? k : key;
        ns += cc[p] >= thInlr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
        ns += __shfl_xor(ns, o);
    }
    if (lane == 0) { s_key[wave] = key; s_cnt[wave] = ns; }
    __syncthreads();
    key = s_key[0]; ns = s_cnt[0];
#pragma unroll
    for (int w = 1; w < (NTHR / 64); ++w) { key = s_key[w] > key ? s_key[w] : key; ns += s_cnt[w]; }
    const int winner_g = a.iters > 0 ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0;
    const int winner = winner_g - a.hyp0g;
    const int maxInl = (int)(key >> 32);
    const bool failed = !(a.iters > 0 && a.has[hyp0 + winner]);                 // :75-89
    if (part) {
        if (threadIdx.x < 12) part->T[threadIdx.x] = failed ? 0.0 : a.TF[(hyp0 + winner) * 12 + threadIdx.x];
        if (threadIdx.x == 0) { part->key = a.iters > 0 ? key : 0ull; part->num_success = ns; part->has = failed ? 0 : 1; }
        return;
    }
    if (threadIdx.x < 12) s_T[threadIdx.x] = failed ? 0.0 : a.TF[(hyp0 + winner) * 12 + threadIdx.x];
    ransac_emit_result<NTHR>(a, n, off, out + b, inlier_idx, failed, s_T, ns, maxInl, winner_g, &s_base, s_wcnt);
}

// ---- the same over SEVERAL workgroups (one large registration: the staged chain) --------------------------------------
// One workgroup spends most of ransac_select_kernel's 33 us reading the correspondences (1.5 MB through one CU) for the
// ordered inlier list.  Here every workgroup finds the winner itself (10 k counts: nothing), tests ITS slice of the
// correspondences, publishes its kept count with a ready bit, adds up the counts of the workgroups before it (ticket order,
// relaxed agent-scope accesses: see match_finish_kernel) and writes its piece of the list; the last ticket also writes the
// result struct.  The counters must be zero at entry (rs_stage1_kernel clears them); the last workgroup through leaves
// them zero again.
constexpr int kSelBlocks = 64, kSelThreads = 1024, kSelPerThread = 32;
struct SelCtr { int32_t ticket, finished; int32_t status[kSelBlocks]; };
// The staged chain's last epilogue (rs_finish: refined counts from the point blocks' partial rows, `has` flags) rides along:
// every workgroup needs all counts for the maximum anyway, so each sums the partial rows itself (pb <= 64 coalesced rows of
// `iters` words from L2) and the first one stores cnt2 / has for the callers that read them -- one launch fewer per step.
struct SelFinish { const int32_t* part; const unsigned char* v2; int pb; };      // v2 == null: cnt2 / has are already final; part == null: cnt2 holds the summed counts
__global__ __launch_bounds__(kSelThreads) void ransac_select_multi_kernel(RansacArgs a, pcreg_dev_ransac_result* out, int32_t* inlier_idx,
                                                                           SelCtr* __restrict__ ctr, int per_block, SelFinish fin) {
    constexpr int NW = kSelThreads / 64;
    __shared__ unsigned long long s_key[NW];
    __shared__ int s_cnt[NW];
    __shared__ double s_T[12];
    __shared__ int s_ticket;
    __shared__ int s_wcnt[kSelPerThread][NW];
    __shared__ int s_red[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = min(a.n_dev ? *a.n_dev : a.n_cap, a.n_cap);
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ctr->ticket, 1);
    // the winner (ransac.m:69-73): first index of the maximum = max over (count << 32 | ~global index)
    const int32_t* cc = a.refine ? a.cnt2 : a.cnt1;
    const int thInlr = matlab_round_i(a.ratio * (double)n);
    unsigned long long key = 0; int ns = 0;
    for (int p = threadIdx.x; p < a.iters; p += kSelThreads) {
        int c;
        if (fin.v2) {                                                        // rs_finish_body, per workgroup
            c = 0;
            if (fin.v2[p]) {
                if (fin.part) {
#pragma unroll 8
                    for (int pbk = 0; pbk < fin.pb; ++pbk) c += fin.part[(size_t)pbk * a.iters + p];
                } else c = a.cnt2[p];
            }
            if (blockIdx.x == 0) { if (fin.part) a.cnt2[p] = c; a.has[p] = fin.v2[p] && c >= thInlr; }
        } else c = cc[p];
        unsigned long long k = ((unsigned long long)(unsigned)c << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(p + a.hyp0g));
        key = k > key ? k : key;
        ns += c >= thInlr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
        ns += __shfl_xor(ns, o);
    }
    if (lane == 0) { s_key[wave] = key; s_cnt[wave] = ns; }
    __syncthreads();
    key = s_key[0]; ns = s_cnt[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { key = s_key[w] > key ? s_key[w] : key; ns += s_cnt[w]; }
    const int winner_g = a.iters > 0 ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0;
    const int winner = winner_g - a.hyp0g;
    const int maxInl = (int)(key >> 32);
    // ransac.m:75-89.  (With the finish folded in, has[winner] = v2[winner] && its count >= thInlr: no read of another workgroup's store.)
    const bool failed = !(a.iters > 0 && (fin.v2 ? (fin.v2[winner] != 0 && maxInl >= thInlr) : a.has[winner] != 0));
    if (threadIdx.x < 12) s_T[threadIdx.x] = failed ? 0.0 : a.TF[(size_t)winner * 12 + threadIdx.x];
    __syncthreads();
    const int b = s_ticket;
    // this workgroup's slice: points [b * per_block, ...), thread t tests i0 + it * 1024 + t (coalesced), it < per_block / 1024
    const int i0 = b * per_block, i1 = min(n, i0 + per_block);
    unsigned mine = 0u; int my_total = 0;
    if (!failed) {
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = s_T[k];
        Pts<false> P{a.p1, a.p2, a.ld, nullptr, n};
        const int rounds = per_block / kSelThreads;
        for (int it = 0; it < rounds; ++it) {
            const int i = i0 + it * kSelThreads + threadIdx.x;
            const bool act = i < i1;
            double q[6]; P.load(act ? i : max(n - 1, 0), q);
            const bool in = act && n > 0 && sqdist(q, T) < a.thDist;
            const unsigned long long bal = __ballot(in);
            if (lane == 0) s_wcnt[it][wave] = __popcll(bal);
            mine |= in ? (1u << it) : 0u;
        }
        __syncthreads();
        for (int it = 0; it < rounds; ++it)
#pragma unroll
            for (int w = 0; w < NW; ++w) my_total += s_wcnt[it][w];
    }
    if (threadIdx.x == 0) __hip_atomic_store(&ctr->status[b], (int)(0x80000000u | (unsigned)my_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int v = 0;
    if ((int)threadIdx.x < b) {
        int sv;
        do { sv = __hip_atomic_load(&ctr->status[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (sv >= 0);
        v = sv & 0x7FFFFFFF;
    }
    if (wave == 0) {                                       // b <= 64 predecessors: one wave holds them all
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[0] = v;
    }
    __syncthreads();
    int base = s_red[0];
    if (b == (int)gridDim.x - 1) {                         // the last slice knows the total: the result struct (ransac.m:75-98)
        pcreg_dev_ransac_result* r = out;
        if (threadIdx.x < 16) {   // column-major 4x4: T(k,j) = R(j,k), T(4,j) = t(j)
            const int k = threadIdx.x & 3, j = threadIdx.x >> 2;
            double tv = 0.0;
            if (!failed) tv = (j < 3) ? s_T[j * 4 + k] : (k == 3 ? 1.0 : 0.0);
            r->T[k + 4 * j] = tv;
        }
        if (threadIdx.x == 0) {
            r->failed = failed; r->num_success = failed ? 0 : ns; r->max_inliers = failed ? 0 : maxInl;
            r->n = n; r->winner = winner_g; r->n_inliers = failed ? 0 : base + my_total;
        }
    }
    if (!failed) {
        const int rounds = per_block / kSelThreads;
        for (int it = 0; it < rounds; ++it) {
            const bool in = (mine >> it) & 1u;
            const unsigned long long bal = __ballot(in);
            int o = base;
            for (int w = 0; w < wave; ++w) o += s_wcnt[it][w];
            if (in) inlier_idx[o + __popcll(bal & ((1ull << lane) - 1ull))] = i0 + it * kSelThreads + threadIdx.x + 1;
#pragma unroll
            for (int w = 0; w < NW; ++w) base += s_wcnt[it][w];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ctr->finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket == (int)gridDim.x - 1) {
        if ((int)threadIdx.x < kSelBlocks) ctr->status[threadIdx.x] = 0;
        if (threadIdx.x == 0) { ctr->ticket = 0; ctr->finished = 0; }
    }
}

// the shares' parts (key by MAX, num_success by SUM, has/T from the share whose key won; n_parts == 1: already
// combined) -> result + inlier list
template <int NTHR>
__global__ __launch_bounds__(NTHR) void ransac_finish_kernel(RansacArgs a, const pcreg_dev_ransac_part* parts, int n_parts,
                                                               pcreg_dev_ransac_result* out, int32_t* inlier_idx) {
    __shared__ double s_T[12];
    __shared__ int s_base;
    __shared__ int s_wcnt[(NTHR / 64)];
    __shared__ unsigned long long s_key;
    __shared__ int s_ns, s_win;
    int n = a.n_dev ? *a.n_dev : a.n_cap;
    n = min(n, a.n_cap);
    if (threadIdx.x == 0) {
        unsigned long long key = 0ull; int ns = 0, win = 0;
        for (int r = 0; r < n_parts; ++r) {              // keys of different shares differ (global hypothesis index) unless both are 0
            ns += parts[r].num_success;
            if (parts[r].key > key) { key = parts[r].key; win = r; }
        }
        s_key = key; s_ns = ns; s_win = win;
    }
    __syncthreads();
    const pcreg_dev_ransac_part* part = parts + s_win;
    const unsigned long long key = s_key;
    const bool failed = key == 0ull || part->has == 0;
    if (threadIdx.x < 12) s_T[threadIdx.x] = failed ? 0.0 : part->T[threadIdx.x];
    ransac_emit_result<NTHR>(a, n, 0, out, inlier_idx, failed, s_T, s_ns, (int)(key >> 32),
                             (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)), &s_base, s_wcnt);
}

// ---------------------------------------------------------------- single-fit kernels
// estimateTransform.m as a stand-alone call (host API / tests): one wave.
__global__ __launch_bounds__(64) void estimate_transform_kernel(const double* p1, const double* p2, int n,
                                                                int ld, double* T16, int32_t* empty) {
    const int lane = threadIdx.x;
    Pts<false> P{p1, p2, ld, nullptr, n};
    double T[12]; bool ok = false;
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.0;
    if (n == 3) {
        double A1[3][3], A2[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double q[6]; P.load(j, q);
            A1[j][0] = q[0]; A1[j][1] = q[1]; A1[j][2] = q[2];
            A2[j][0] = q[3]; A2[j][1] = q[4]; A2[j][2] = q[5];
        }
        ok = fit_3pt(A1, A2, T);
    } else if (n > 3) {
        double o[6]; P.load(0, o);
        double acc[27], mom[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.0;
        for (int i = lane; i < n; i += 64) { double q[6]; P.load(i, q); mom_accumulate(acc, q, o); }
        wave_sum27(acc);
#pragma unroll
        for (int k = 0; k < 27; ++k) mom[k] = acc[k];
        // rank(pts1) >= 3 and rank(pts2) >= 2 (estimateTransform.m:11): from the Grams where they can tell, from the points
        // themselves (one more pass along the deciding eigenvector) where the Gram's noise floor hides the answer
        const double g1[6] = {mom[15], mom[16], mom[17], mom[18], mom[19], mom[20]};
        const double g2[6] = {mom[21], mom[22], mom[23], mom[24], mom[25], mom[26]};
        double u1[3], w1[3], u2[3], w2[3], tol1, tol2; bool amb1, amb2;
        bool r1 = rank_gram_probe(g1, n, 3, u1, w1, tol1, amb1), r2 = rank_gram_probe(g2, n, 2, u2, w2, tol2, amb2);
        if (amb1 || amb2) {
            double e[6] = {0, 0, 0, 0, 0, 0};
            for (int i = lane; i < n; i += 64) {
                double q[6]; P.load(i, q);
                const double a1 = fma(q[2], u1[2], fma(q[1], u1[1], q[0] * u1[0])), b1 = fma(q[2], w1[2], fma(q[1], w1[1], q[0] * w1[0]));
                const double a2 = fma(q[5], u2[2], fma(q[4], u2[1], q[3] * u2[0])), b2 = fma(q[5], w2[2], fma(q[4], w2[1], q[3] * w2[0]));
                e[0] = fma(a1, a1, e[0]); e[1] = fma(a1, b1, e[1]); e[2] = fma(b1, b1, e[2]);
                e[3] = fma(a2, a2, e[3]); e[4] = fma(a2, b2, e[4]); e[5] = fma(b2, b2, e[5]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) e[k] = wave_sum(e[k]);
            if (amb1) r1 = rank_from_projection(e[0], e[1], e[2], 3, tol1);
            if (amb2) r2 = rank_from_projection(e[3], e[4], e[5], 2, tol2);
        }
        ok = r1 && r2 && fit_moments(n, mom, o, T, true);
    }
    if (lane < 16) {
        int k = lane & 3, j = lane >> 2;
        double v = 0.0;
        if (ok) {
            double tv = 0.0;
#pragma unroll
            for (int e = 0; e < 12; ++e) if (e == j * 4 + k) tv = T[e];
            v = (j < 3) ? tv : (k == 3 ? 1.0 : 0.0);
        }
        T16[k + 4 * j] = v;
    }
    if (lane == 0) *empty = ok ? 0 : 1;
}

// completeExperimentFast.m:368-391: inliers = find(vecnorm(pts1 - pts2, 2, 2) < maxDist), then
// T_refine = estimateTransform(pts1(inliers,:), pts2(inliers,:)).  One wave; n read from the device.
__global__ __launch_bounds__(64) void refine_by_distance_kernel(const double* p1, const double* p2, const int32_t* n_dev, int cap,
                                                                int ld, double maxDist, double* T16, int32_t* info /*[2]: inliers, empty*/) {
    __shared__ double s3[18];
    const int lane = threadIdx.x;
    const int n = min(*n_dev, cap);
    Pts<false> P{p1, p2, ld, nullptr, n};
    double T[12]; bool ok = false;
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.0;
    int cnt = 0;
    if (n > 0) {
        double o[6]; P.load(0, o);
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < n;
            double q[6]; P.load(act ? i : 0, q);
            const double dx = q[0] - q[3], dy = q[1] - q[4], dz = q[2] - q[5];
            const bool in = act && sqrt((dx * dx + dy * dy) + dz * dz) < maxDist;
            const unsigned long long bal = __ballot(in);
            if (in) mom_accumulate(acc, q, o);
            if (cnt < 3) {                                   // the N == 3 branch needs the points themselves
                const int r3 = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                if (in && r3 < 3) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) s3[r3 * 6 + c] = q[c];
                }
            }
            cnt += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (cnt == 3) {
            double A1[3][3], A2[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) { A1[j][c] = s3[j * 6 + c]; A2[j][c] = s3[j * 6 + 3 + c]; }
            ok = fit_3pt(A1, A2, T);
        } else if (cnt > 3) {
            double mom[27];
            wave_sum27(acc);
#pragma unroll
            for (int k = 0; k < 27; ++k) mom[k] = acc[k];
            ok = fit_moments(cnt, mom, o, T);
        }
    }
    if (lane < 16) {
        int k = lane & 3, j = lane >> 2;
        double v = 0.0;
        if (ok) {
            double tv = 0.0;
#pragma unroll
            for (int e = 0; e < 12; ++e) if (e == j * 4 + k) tv = T[e];
            v = (j < 3) ? tv : (k == 3 ? 1.0 : 0.0);
        }
        T16[k + 4 * j] = v;
    }
    if (lane == 0) { info[0] = cnt; info[1] = ok ? 0 : 1; }
}

__global__ void calc_dists_kernel(const double* T16, const double* p1, const double* p2, int n, int ld, double* d) {
    double T[12];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) T[j * 4 + k] = T16[k + 4 * j];
    Pts<false> P{p1, p2, ld, nullptr, n};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double q[6]; P.load(i, q);
        d[i] = sqdist(q, T);
    }
}

}  // namespace

// ---------------------------------------------------------------- launchers
// The staged chain carries ~0.1 ms of launch-sized kernels; the tiled kernel (fp64 only) has none but scores at half the rate.
// Measured (round 4, same box, REFINE): n = 2100 / 3000 / 4000 with 10^4 hypotheses 0.206 / 0.244 / 0.267 ms tiled against
// 0.149 / 0.158 / 0.169 staged; (4000, 10^3) 0.099 against 0.119, (2500, 2 x 10^3) 0.079 against 0.114.
static bool staged_pays(int n_cap, int iters) { return n_cap >= 4096 || (n_cap >= kStagedMinN && (long long)n_cap * iters >= 20000000LL); }
static size_t staged_slots_cap(int n_cap) { return (size_t)((n_cap + kSPts - 1) / kSPts) * (kSPts / 64); }
static size_t staged_chunks_cap(int n_cap) { return (staged_slots_cap(n_cap) + kMomSlots - 1) / kMomSlots; }
static size_t staged_extra_bytes(size_t h, int n_cap) {    // T1 | mom | part | v1 | pass1 | v2 | cert | dense | lane-path buffers
    size_t b = align_up(h * 12 * sizeof(double), 256) + align_up(h * 27 * sizeof(double), 256) +
               align_up(h * kSMaxPB * sizeof(int32_t), 256) + 5 * align_up(h, 256) + 256 + align_up(h * 2 * sizeof(double), 256) +
               align_up((staged_slots_cap(n_cap) * 64 + 255) / 256 * 4 * sizeof(double), 256) + 512;
    if (n_cap >= kStagedMinN)
        b += align_up(staged_slots_cap(n_cap) * 64 * 6 * sizeof(float), 256) + 2 * align_up(h * 16 * sizeof(float), 256) +
             align_up(h * staged_slots_cap(n_cap) * 8, 256) + align_up(staged_slots_cap(n_cap) * 64 * kRec * sizeof(double) + 256, 256) +
             align_up(staged_chunks_cap(n_cap) * h * 15 * sizeof(double), 256) + align_up(h * sizeof(int32_t), 256) + 256 +
             align_up(staged_slots_cap(n_cap) * 2 * 256 * sizeof(uint4), 256);
    return b;
}
size_t ransac_workspace_bytes(int iters, int B, int n_cap) {
    size_t h = (size_t)iters * (size_t)B;
    return align_up(h * 12 * sizeof(double), 256) + 2 * align_up(h * sizeof(int32_t), 256) + align_up(h, 256) +
           align_up(h * 16 * sizeof(double), 256) /* ransac_hyp32_kernel's refit sums */ + staged_extra_bytes(h, B == 1 ? n_cap : 0);
}

// ransac_hyp32_kernel: launch classes by correspondences per registration (LDS = 72 B per correspondence, rounded up to 128 of
// them).  The kernel holds 126 VGPRs, four waves per SIMD, so a CU takes sixteen waves: two 8-wave workgroups of up to 1024
// correspondences (74 KB each), or one 16-wave workgroup of up to 2048 (148 KB of the CU's 160).
constexpr int kHyp32NClasses = 2;
constexpr int kHyp32Classes[kHyp32NClasses] = {1024, 2048};
constexpr int kHyp32Waves[kHyp32NClasses] = {8, 16};
constexpr int kHyp32BatchCap = 3242;            // batches of a larger capacity run on the tiled kernel, as before round 4
static size_t hyp32_lds_bytes(int n, int nw) { return (size_t)((n + 127) / 128 * 128) * 72 + (size_t)nw * 4 * sizeof(double); }
template <int NW>
static int launch_hyp32(const RansacArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        PCREG_HIP(hipFuncSetAttribute((const void*)ransac_hyp32_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PCREG_HIP(hipFuncSetAttribute((const void*)ransac_hyp32_kernel<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (a.refine) hipLaunchKernelGGL((ransac_hyp32_kernel<NW, true>), grid, dim3(NW * 64), lds, st, a);
    else hipLaunchKernelGGL((ransac_hyp32_kernel<NW, false>), grid, dim3(NW * 64), lds, st, a);
    return PCREG_OK;
}
static int launch_ransac_impl(const double* p1, const double* p2, int ld, const int32_t* offsets, const int32_t* n_dev,
                  int n_cap, int B, pcreg_ransac_opts o, const int32_t* sample_idx_dev,
                  pcreg_dev_ransac_result* out, int32_t* inlier_idx, int32_t* iter_inl, int32_t* iter_inl_ref,
                  void* ws, size_t ws_bytes, hipStream_t st, int hyp_begin, pcreg_dev_ransac_part* part) {
    PCREG_ARG(o.iterNum >= 1 && o.minPtNum >= 3 && B >= 1 && n_cap >= 0);
    PCREG_ARG(o.minPtNum == 3 || sample_idx_dev != nullptr);   // built-in sampler draws triples
    size_t need = ransac_workspace_bytes(o.iterNum, B, n_cap);
    if (ws_bytes < need) { set_error("ransac workspace too small: %zu < %zu", ws_bytes, need); return PCREG_E_WORKSPACE; }
    size_t h = (size_t)o.iterNum * (size_t)B;
    char* w = (char*)ws;
    RansacArgs a{};
    a.p1 = p1; a.p2 = p2; a.ld = ld; a.offsets = offsets; a.n_dev = n_dev; a.n_cap = n_cap;
    a.iters = o.iterNum; a.m = o.minPtNum; a.thDist = o.thDist; a.ratio = o.thInlrRatio;
    a.refine = o.REFINE != 0; a.seed = o.seed; a.sample_idx = sample_idx_dev; a.hyp0g = hyp_begin;
    a.n_hi = 0x7FFFFFFF; a.n_lo = -1;
    a.TF = (double*)w; w += align_up(h * 12 * sizeof(double), 256);
    a.cnt1 = (int32_t*)w; w += align_up(h * sizeof(int32_t), 256);
    a.cnt2 = (int32_t*)w; w += align_up(h * sizeof(int32_t), 256);
    a.has = (unsigned char*)w; w += align_up(h, 256);
    long long total = (long long)o.iterNum * B;
    void* sel_ctr = nullptr;              // set by the staged chain: its selection runs on several workgroups
    SelFinish sel_fin{nullptr, nullptr, 0};
    bool fold_finish = false;
    if (debug_flag(kDbgRansacResidentF64) && n_cap > 0 && (size_t)n_cap * 48 <= (offsets ? (size_t)152 * 1024 : (size_t)64 * 1024)) {
        // A/B and parity switch: the round-3 kernel, fp64 scoring on the raw coordinates in LDS (up to 32 KB per registration of a
        // batch of large capacity, the rest from L2)
        size_t lds = (size_t)n_cap * 48;
        int hpw = 64;
        while (hpw > 8 && total / hpw < 256LL * 4 * 2) hpw >>= 1;
        a.hpw = hpw;
        if (offsets && lds > 64 * 1024) { lds = 32 * 1024; a.n_hi = (int)(lds / 48); }
        const int pb4 = hpw * kWavesPerBlock;
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((o.iterNum + pb4 - 1) / pb4, B), dim3(kBlock), lds, st, a);
    } else if (n_cap > 0 && n_cap <= (offsets ? kHyp32BatchCap : kHyp32Classes[kHyp32NClasses - 1])) {
        // correspondences resident in LDS, scored through the fp32 screen (ransac_hyp32_kernel); registrations of a batch above
        // 2048 correspondences take the fp64 kernel on the raw coordinates from L2 (a second launch whose other workgroups return
        // at once: n is read from the offsets on the device).  (Two launch classes by LDS size, <= 1024 and <= 2048, were measured
        // on the sweep -- capacity 2000, most trials ~250 pairs, the right spheres ~2000: 1.13 + 0.61 ms one after the other.)
        // Hypotheses per wave: fill the chip first (>= ~2 waves per SIMD), then grow towards 64 so the lane-parallel fits run full.
        int hpw = 64;
        while (hpw > 8 && total / hpw < 256LL * 4 * 2) hpw >>= 1;
        a.hpw = hpw;
        a.msc = (double*)w; w += align_up(h * 16 * sizeof(double), 256);
        // one launch: capacities up to 1024 as 8-wave workgroups with the LDS of the capacity (two to a CU), larger ones as 16-wave
        // workgroups with the LDS of min(capacity, 2048) correspondences (one to a CU: the same sixteen waves)
        const int c = n_cap <= kHyp32Classes[0] ? 0 : 1;
        a.n_lo = -1; a.n_hi = std::min(n_cap, kHyp32Classes[c]);
        const int nw = kHyp32Waves[c], per_block = hpw * nw;
        const dim3 grid((o.iterNum + per_block - 1) / per_block, B);
        const int rc = nw == 8 ? launch_hyp32<8>(a, grid, hyp32_lds_bytes(a.n_hi, nw), st) : launch_hyp32<16>(a, grid, hyp32_lds_bytes(a.n_hi, nw), st);
        if (rc != PCREG_OK) return rc;
        const int lo = a.n_hi;
        if (n_cap > lo) {     // registrations of a batch above every class: the fp64 kernel on the raw coordinates from L2
            a.n_lo = lo; a.n_hi = -1;
            const int pb4 = hpw * kWavesPerBlock;
            hipLaunchKernelGGL(ransac_hyp_kernel, dim3((o.iterNum + pb4 - 1) / pb4, B), dim3(kBlock), 0, st, a);
        }
    } else if (B == 1 && !offsets && staged_pays(n_cap, o.iterNum) && !debug_flag(kDbgRansacFused)) {
        // one large registration: the staged chain of lean kernels (see rs_* above)
        StagedArgs sa{};
        sa.T1 = (double*)w; w += align_up(h * 12 * sizeof(double), 256);
        sa.mom = (double*)w; w += align_up(h * 27 * sizeof(double), 256);
        sa.part = (int32_t*)w; w += align_up(h * kSMaxPB * sizeof(int32_t), 256);
        sa.v1 = (unsigned char*)w; w += align_up(h, 256);
        sa.pass1 = (unsigned char*)w; w += align_up(h, 256);
        sa.v2 = (unsigned char*)w; w += align_up(h, 256);
        sa.cert = (unsigned char*)w; w += align_up(h, 256);
        sa.dense = (unsigned char*)w; w += align_up(h, 256);
        sa.bounds = (double*)w; w += 256;
        sa.n_rec_blocks = (int)((staged_slots_cap(n_cap) * 64 + 255) / 256);
        sa.bpart = (double*)w; w += align_up((size_t)sa.n_rec_blocks * 4 * sizeof(double), 256);
        sa.sel_ctr = w; w += 512;
        sel_ctr = sa.sel_ctr;
        sa.certq = (double*)w; w += align_up(h * 2 * sizeof(double), 256);
        sa.nslots_cap = (int)staged_slots_cap(n_cap);
        sa.masks = (unsigned long long*)w; w += align_up(h * staged_slots_cap(n_cap) * 8, 256);
        sa.rec = (double*)w; w += align_up(staged_slots_cap(n_cap) * 64 * kRec * sizeof(double) + 256, 256);
        sa.mpart = (double*)w; w += align_up(staged_chunks_cap(n_cap) * h * 15 * sizeof(double), 256);
        sa.pass_list = (int32_t*)w; w += align_up(h * sizeof(int32_t), 256);
        sa.dig = (uint4*)w; w += align_up(staged_slots_cap(n_cap) * 2 * 256 * sizeof(uint4), 256);
        sa.n_pass = (int32_t*)w; w += 256;
        sa.n32 = (int)staged_slots_cap(n_cap) * 64;
        sa.c32 = (float*)w; w += align_up((size_t)sa.n32 * 6 * sizeof(float), 256);
        sa.T32a = (float*)w; w += align_up(h * 16 * sizeof(float), 256);
        sa.T32b = (float*)w; w += align_up(h * 16 * sizeof(float), 256);
        sa.use_lane = a.refine && !debug_flag(kDbgRansacNoLane);
        sa.use_f32 = !debug_flag(kDbgRansacF64Score);
        int pb = (n_cap + kSPts - 1) / kSPts; if (pb > kSMaxPB) pb = kSMaxPB; if (pb < 1) pb = 1;
        sa.pb = pb;
        int hpw = (int)((total + 250LL * kTW - 1) / (250LL * kTW));
        if (hpw < 1) hpw = 1;
        if (hpw > 16) hpw = 16;
        a.hpw = hpw;
        sa.a = a;
        const int it = o.iterNum;
        const dim3 sgrid(pb, (it + kSChunk - 1) / kSChunk);
        // 9 launches (round 3: 11): stage1 (sample fits + records), stage2 (maxima + fp32 rows + the records' digits), scoring pass 1,
        // pass1, refit sums on the matrix cores, dense refit sums (normally idle), refits, scoring pass 2; then select, which also
        // does what rs_finish did
        const int n_fit = (it + 255) / 256;
        hipLaunchKernelGGL(rs_stage1_kernel, dim3((unsigned)(n_fit + sa.n_rec_blocks)), dim3(256), 0, st, sa, n_fit);
        const int n_t32 = (it + 255) / 256, n_dig = sa.use_lane ? (int)staged_slots_cap(n_cap) * 2 : 0;
        hipLaunchKernelGGL(rs_stage2_kernel, dim3((unsigned)(n_t32 + n_dig)), dim3(256), 0, st, sa, (const double*)sa.T1, sa.T32a, n_t32, sa.dig);
        if (sa.use_f32) {
            if (sa.use_lane)
                hipLaunchKernelGGL(rs_score32_kernel<true>, sgrid, dim3(kSW * 64), 0, st, sa, (const double*)sa.T1, (const float*)sa.T32a, (const unsigned char*)sa.v1, (int32_t*)nullptr);
            else
                hipLaunchKernelGGL(rs_score32_kernel<false>, sgrid, dim3(kSW * 64), 0, st, sa, (const double*)sa.T1, (const float*)sa.T32a, (const unsigned char*)sa.v1, (int32_t*)nullptr);
        } else if (sa.use_lane) {
            hipLaunchKernelGGL(rs_score_kernel<true>, sgrid, dim3(kSW * 64), 0, st, sa, (const double*)sa.T1, (const unsigned char*)sa.v1);
        } else {
            hipLaunchKernelGGL(rs_score_kernel<false>, sgrid, dim3(kSW * 64), 0, st, sa, (const double*)sa.T1, (const unsigned char*)sa.v1);
        }
        hipLaunchKernelGGL(rs_pass1_kernel, dim3((it + 255) / 256), dim3(256), 0, st, sa);
        if (a.refine) {
            if (sa.use_lane)
                hipLaunchKernelGGL(rs_moments_mfma_kernel, dim3((it + kMmWaves * kMmRows - 1) / (kMmWaves * kMmRows), (unsigned)staged_chunks_cap(n_cap)), dim3(kMmWaves * 64), 0, st, sa,
                                   (const uint4*)sa.dig, (const unsigned long long*)sa.masks);
            hipLaunchKernelGGL(rs_moments_kernel, dim3((it + hpw * kTW - 1) / (hpw * kTW)), dim3(kTBlock), 0, st, sa);
            hipLaunchKernelGGL(rs_fit2_kernel, dim3((it + 63) / 64), dim3(64), 0, st, sa);
            fold_finish = !part && n_cap <= kSelBlocks * kSelPerThread * kSelThreads;        // ransac_select_multi_kernel does rs_finish's job
            // with the finish folded into the selection the second pass ADDS its point blocks' counts into cnt2 (zeroed by rs_pass1;
            // integer sums are order-free) instead of leaving pb partial rows that every selecting workgroup would have to sum
            if (sa.use_f32)
                hipLaunchKernelGGL(rs_score32_kernel<false>, sgrid, dim3(kSW * 64), 0, st, sa, (const double*)a.TF, (const float*)sa.T32b, (const unsigned char*)sa.v2,
                                   fold_finish ? a.cnt2 : (int32_t*)nullptr);
            else
                hipLaunchKernelGGL(rs_score_kernel<false>, sgrid, dim3(kSW * 64), 0, st, sa, (const double*)a.TF, (const unsigned char*)sa.v2);
            if (fold_finish) { sel_fin.part = sa.use_f32 ? nullptr : sa.part; sel_fin.v2 = sa.v2; sel_fin.pb = sa.pb; }
            else hipLaunchKernelGGL(rs_finish_kernel, dim3((it + 255) / 256), dim3(256), 0, st, sa);
        }
    } else {
        // several large sets: 8-wave workgroups share LDS tiles of the correspondences
        // one workgroup per CU is resident (<= 256 VGPRs x 8 waves); give every wave just enough
        // hypotheses that ~250 workgroups cover the job in a single round (a second, ragged
        // round of workgroups costs a full sweep sequence for a fraction of the CUs)
        int hpw = (int)((total + 250LL * kTW - 1) / (250LL * kTW));
        if (hpw < 1) hpw = 1;
        if (hpw > 16) hpw = 16;
        a.hpw = hpw;
        int per_block = hpw * kTW;
        dim3 grid((o.iterNum + per_block - 1) / per_block, B);
        hipLaunchKernelGGL(ransac_hyp_tiled_kernel, grid, dim3(kTBlock), 0, st, a);
    }
    PCREG_HIP(hipGetLastError());
    if (sel_ctr && !part && n_cap <= kSelBlocks * kSelPerThread * kSelThreads) {
        int per_block = 4 * kSelThreads;
        if ((n_cap + per_block - 1) / per_block > kSelBlocks) per_block = ((n_cap + kSelBlocks - 1) / kSelBlocks + kSelThreads - 1) / kSelThreads * kSelThreads;
        const int nb = std::max(1, (n_cap + per_block - 1) / per_block);
        hipLaunchKernelGGL(ransac_select_multi_kernel, dim3(nb), dim3(kSelThreads), 0, st, a, out, inlier_idx, (SelCtr*)sel_ctr, per_block, sel_fin);
    }
    else if (n_cap >= 8192) hipLaunchKernelGGL(ransac_select_kernel<1024>, dim3(B), dim3(1024), 0, st, a, out, inlier_idx, part);   // long inlier lists
    else hipLaunchKernelGGL(ransac_select_kernel<kBlock>, dim3(B), dim3(kBlock), 0, st, a, out, inlier_idx, part);
    PCREG_HIP(hipGetLastError());
    if (iter_inl) PCREG_HIP(hipMemcpyAsync(iter_inl, a.cnt1, h * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (iter_inl_ref) PCREG_HIP(hipMemcpyAsync(iter_inl_ref, a.cnt2, h * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return PCREG_OK;
}

int launch_ransac(const double* p1, const double* p2, int ld, const int32_t* offsets, const int32_t* n_dev,
                  int n_cap, int B, const pcreg_ransac_opts& o, const int32_t* sample_idx_dev,
                  pcreg_dev_ransac_result* out, int32_t* inlier_idx, int32_t* iter_inl, int32_t* iter_inl_ref,
                  void* ws, size_t ws_bytes, hipStream_t st) {
    return launch_ransac_impl(p1, p2, ld, offsets, n_dev, n_cap, B, o, sample_idx_dev, out, inlier_idx, iter_inl, iter_inl_ref,
                              ws, ws_bytes, st, 0, nullptr);
}

// hypotheses [hyp_begin, hyp_begin + hyp_count) of one registration of o.iterNum; sample_idx_dev (if any) holds
// the rows of THIS share
int launch_ransac_partial(const double* p1, const double* p2, int ld, const int32_t* n_dev, int n_cap,
                          const pcreg_ransac_opts& o, const int32_t* sample_idx_dev, int hyp_begin, int hyp_count,
                          pcreg_dev_ransac_part* part, void* ws, size_t ws_bytes, hipStream_t st) {
    PCREG_ARG(hyp_begin >= 0 && hyp_count >= 0 && hyp_begin + hyp_count <= o.iterNum);
    if (hyp_count == 0) { PCREG_HIP(hipMemsetAsync(part, 0, sizeof(pcreg_dev_ransac_part), st)); return PCREG_OK; }
    pcreg_ransac_opts ol = o;
    ol.iterNum = hyp_count;
    return launch_ransac_impl(p1, p2, ld, nullptr, n_dev, n_cap, 1, ol, sample_idx_dev, nullptr, nullptr, nullptr, nullptr,
                              ws, ws_bytes, st, hyp_begin, part);
}

int launch_ransac_finish(const double* p1, const double* p2, int ld, const int32_t* n_dev, int n_cap,
                         const pcreg_ransac_opts& o, const pcreg_dev_ransac_part* combined,
                         pcreg_dev_ransac_result* out, int32_t* inlier_idx, hipStream_t st, int n_parts) {
    PCREG_ARG(n_parts >= 1);
    RansacArgs a{};
    a.p1 = p1; a.p2 = p2; a.ld = ld; a.n_dev = n_dev; a.n_cap = n_cap; a.thDist = o.thDist; a.ratio = o.thInlrRatio;
    if (n_cap >= 8192) hipLaunchKernelGGL(ransac_finish_kernel<1024>, dim3(1), dim3(1024), 0, st, a, combined, n_parts, out, inlier_idx);
    else hipLaunchKernelGGL(ransac_finish_kernel<kBlock>, dim3(1), dim3(kBlock), 0, st, a, combined, n_parts, out, inlier_idx);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_estimate_transform(const double* p1, const double* p2, int n, int ld, double* T16_dev,
                              int32_t* empty_dev, hipStream_t st) {
    hipLaunchKernelGGL(estimate_transform_kernel, dim3(1), dim3(64), 0, st, p1, p2, n, ld, T16_dev, empty_dev);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_refine_by_distance(const double* p1, const double* p2, const int32_t* n_dev, int cap, int ld, double maxDist,
                              double* T16_dev, int32_t* info_dev, hipStream_t st) {
    hipLaunchKernelGGL(refine_by_distance_kernel, dim3(1), dim3(64), 0, st, p1, p2, n_dev, cap, ld, maxDist, T16_dev, info_dev);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

int launch_calc_dists(const double* T16_dev, const double* p1, const double* p2, int n, int ld, double* d,
                      hipStream_t st) {
    if (n <= 0) return PCREG_OK;
    int blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(calc_dists_kernel, dim3(blocks), dim3(256), 0, st, T16_dev, p1, p2, n, ld, d);
    PCREG_HIP(hipGetLastError());
    return PCREG_OK;
}

}  // namespace pcreg
