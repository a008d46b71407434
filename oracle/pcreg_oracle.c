/* oracle/pcreg_oracle.c -- plain-C CPU restatement of the PCReg hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker + reported CPU baseline).  The product
 * (pcreg_amd/, libpcreg_hip.so) never links, loads or calls this file.
 *
 * Reference lines restated (paths relative to the LCJebe/PCReg checkout):
 *   orc_estimate_transform   estimateTransform.m:8-71
 *   orc_calc_dists           getInliersRANSAC.m:46-54 (same body in 6 other scripts)
 *   orc_ransac               ransac.m:21-116
 *   orc_preprocess           getMatches.m:22-41
 *   orc_match_features       getMatches.m:51-56 -> MathWorks matchFeatures (documented
 *                            semantics; closed source, un-vendored: PARITY UNPINNED)
 *   orc_align_points_knn     AlignPoints_KNN.m:8-59 -> MathWorks pca (PARITY UNPINNED)
 *   orc_spatial_histogram_descriptors  getSpacialHistogramDescriptors.m:18-179,
 *                            getLocalPoints.m:8-35, histcn.m:94-131
 *
 * Pinning: MATLAB cannot run here.  This file is pinned (tests/test_oracle_kat.py)
 * against the known answers in the reference's own test scripts
 * (testTransformEstimation.m:2-14, testRANSAC.m:13-42, debugRANSAC.m:2-54,
 * invertTF/quickTF round trip) and against the independent numpy/LAPACK
 * restatement oracle/pcreg_oracle.py.
 *
 * Rounding-level conventions this oracle fixes (unknowable without MATLAB; the HIP
 * path follows the same ones so that index sets can be compared exactly):
 *   - calcDists: tx = fma(x,T11, fma(y,T21, fma(z,T31, T41))), d = fma(dz,dz,
 *     fma(dy,dy, dx*dx)).
 *   - descriptor scores accumulate over the feature index in ascending order
 *     (SAD: s += |a-b|; SSD: s = fma(a-b, a-b, s)).
 *   - fp32 point search: d = fmaf(dz,dz, fmaf(dy,dy, dx*dx)), dx = q - m.
 *   - ties resolve to the lowest index (MATLAB min / stable sort behaviour).
 *
 * Deliberate deviation: a rank-deficient sample makes
 * estimateTransform return [] and the reference then throws out of ransac.m:48;
 * here the hypothesis scores 0 inliers.
 */
#include "pcreg_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small helpers */
int orc_matlab_round(double x) { /* MATLAB round: half away from zero (ransac.m:28) */
    return (int)(x >= 0 ? floor(x + 0.5) : -floor(-x + 0.5));
}
static double eps_at(double x) { x = fabs(x); return nextafter(x, INFINITY) - x; }

/* One-sided (Hestenes) Jacobi on the 3 columns of an n x 3 matrix W (column c at
 * W + c*n).  On return the columns are mutually orthogonal; V (row-major 3x3, may be
 * NULL) accumulates the right rotations: W_out = W_in * V. */
static void hestenes_cols(double* W, int n, double* V) {
    if (V) { memset(V, 0, 9 * sizeof(double)); V[0] = V[4] = V[8] = 1.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double* wp = W + (size_t)p * n; double* wq = W + (size_t)q * n;
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < n; ++i) { al += wp[i] * wp[i]; be += wq[i] * wq[i]; ga += wp[i] * wq[i]; }
                if (ga == 0.0 || fabs(ga) <= DBL_EPSILON * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < n; ++i) {
                    double a = wp[i], b = wq[i];
                    wp[i] = c * a - s * b; wq[i] = s * a + c * b;
                }
                if (V) for (int i = 0; i < 3; ++i) {
                    double a = V[i * 3 + p], b = V[i * 3 + q];
                    V[i * 3 + p] = c * a - s * b; V[i * 3 + q] = s * a + c * b;
                }
            }
        if (!rotated) break;
    }
}

/* A = U diag(S) V^T for a row-major 3x3.  S is not sorted.  Columns of U that belong
 * to (numerically) zero singular values are completed to a right-handed frame. */
void orc_svd3(const double A[9], double U[9], double S[3], double V[9]) {
    double W[9]; /* column c of A at W + 3c */
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) W[c * 3 + r] = A[r * 3 + c];
    hestenes_cols(W, 3, V);
    double smax = 0; int ok[3];
    for (int c = 0; c < 3; ++c) {
        S[c] = sqrt(W[c*3]*W[c*3] + W[c*3+1]*W[c*3+1] + W[c*3+2]*W[c*3+2]);
        if (S[c] > smax) smax = S[c];
    }
    int nok = 0;
    for (int c = 0; c < 3; ++c) {
        ok[c] = (S[c] > 0.0 && S[c] > 1e-300 && S[c] >= smax * 1e-18);
        if (ok[c]) { ++nok; for (int r = 0; r < 3; ++r) U[r * 3 + c] = W[c * 3 + r] / S[c]; }
    }
    if (nok == 3) return;
    if (nok == 2) {
        int m = !ok[0] ? 0 : (!ok[1] ? 1 : 2), a = (m + 1) % 3, b = (m + 2) % 3;
        /* u_m = u_a x u_b keeps (u_m,u_a,u_b) cyclic => det U = +1 */
        U[0*3+m] = U[1*3+a]*U[2*3+b] - U[2*3+a]*U[1*3+b];
        U[1*3+m] = U[2*3+a]*U[0*3+b] - U[0*3+a]*U[2*3+b];
        U[2*3+m] = U[0*3+a]*U[1*3+b] - U[1*3+a]*U[0*3+b];
        return;
    }
    /* rank <= 1: any orthonormal completion */
    double e[3][3] = {{1,0,0},{0,1,0},{0,0,1}};
    int have = -1; for (int c = 0; c < 3; ++c) if (ok[c]) have = c;
    double u0[3] = {1, 0, 0};
    if (have >= 0) for (int r = 0; r < 3; ++r) u0[r] = U[r * 3 + have];
    int k = 0; for (int r = 1; r < 3; ++r) if (fabs(u0[r]) < fabs(u0[k])) k = r;
    double v1[3], dot = u0[k];
    for (int r = 0; r < 3; ++r) v1[r] = e[k][r] - dot * u0[r];
    double nv = sqrt(v1[0]*v1[0] + v1[1]*v1[1] + v1[2]*v1[2]);
    for (int r = 0; r < 3; ++r) v1[r] /= nv;
    double v2[3] = { u0[1]*v1[2]-u0[2]*v1[1], u0[2]*v1[0]-u0[0]*v1[2], u0[0]*v1[1]-u0[1]*v1[0] };
    int c0 = have >= 0 ? have : 0, c1 = (c0 + 1) % 3, c2 = (c0 + 2) % 3;
    for (int r = 0; r < 3; ++r) { U[r*3+c0] = u0[r]; U[r*3+c1] = v1[r]; U[r*3+c2] = v2[r]; }
}

/* MATLAB rank() of an n x 3 point matrix (estimateTransform.m:11):
 * #singular values > max(n,3) * eps(largest singular value). */
int orc_rank_nx3(const double* p, int n, int ld) {
    if (n <= 0) return 0;
    double* W = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    for (int c = 0; c < 3; ++c) for (int i = 0; i < n; ++i) W[(size_t)c * n + i] = p[i + (size_t)c * ld];
    hestenes_cols(W, n, NULL);
    double s[3], smax = 0;
    for (int c = 0; c < 3; ++c) {
        double a = 0; for (int i = 0; i < n; ++i) a += W[(size_t)c*n+i] * W[(size_t)c*n+i];
        s[c] = sqrt(a); if (s[c] > smax) smax = s[c];
    }
    free(W);
    double tol = (double)(n > 3 ? n : 3) * eps_at(smax);
    int r = 0; for (int c = 0; c < 3; ++c) if (s[c] > tol) ++r;
    return r;
}

/* ---------------------------------------------------------------- estimateTransform */
int orc_estimate_transform(const double* p1, const double* p2, int n, int ld,
                           double T[16], int* empty) {
    *empty = 1;
    if (n < 1) return 0;
    if (orc_rank_nx3(p1, n, ld) < 3 || orc_rank_nx3(p2, n, ld) < 2) return 0;   /* :11-14 */
    int N = n;
    double *d, *m;   /* N x 3, column c at +c*N (the .m file's d', m') */
    if (n == 3) {                                                             /* :18 */
        N = 4;
        d = (double*)malloc(sizeof(double) * 12); m = (double*)malloc(sizeof(double) * 12);
        const double* src[2] = { p1, p2 }; double* dst[2] = { d, m };
        for (int s = 0; s < 2; ++s) {
            double P[3][3];
            for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) P[i][c] = src[s][i + (size_t)c * ld];
            double cen[3], a[3], b[3], nrm[3], e[3];
            for (int c = 0; c < 3; ++c) cen[c] = (P[0][c] + P[1][c] + P[2][c]) / 3.0;   /* :20-21 */
            for (int c = 0; c < 3; ++c) { a[c] = P[2][c] - P[1][c]; b[c] = P[2][c] - P[0][c]; }
            nrm[0] = a[1]*b[2] - a[2]*b[1]; nrm[1] = a[2]*b[0] - a[0]*b[2]; nrm[2] = a[0]*b[1] - a[1]*b[0]; /* :24-25 */
            for (int i = 0; i < 3; ++i) {                                     /* :28-29 */
                int j = (i + 2) % 3;   /* circshift(pts,1,1): row i pairs with row i-1 */
                double dx = P[i][0]-P[j][0], dy = P[i][1]-P[j][1], dz = P[i][2]-P[j][2];
                e[i] = sqrt(dx*dx + dy*dy + dz*dz);
            }
            double l = fmax(fmin(e[0], e[1]), fmin(fmax(e[0], e[1]), e[2]));   /* median of three */
            double nn = sqrt(nrm[0]*nrm[0] + nrm[1]*nrm[1] + nrm[2]*nrm[2]);
            for (int c = 0; c < 3; ++c) {
                for (int i = 0; i < 3; ++i) dst[s][c * 4 + i] = P[i][c];
                dst[s][c * 4 + 3] = cen[c] + (nrm[c] / nn) * l;               /* :32-36 */
            }
        }
    } else {
        d = (double*)malloc(sizeof(double) * 3 * (size_t)N); m = (double*)malloc(sizeof(double) * 3 * (size_t)N);
        for (int c = 0; c < 3; ++c) for (int i = 0; i < N; ++i) {
            d[(size_t)c*N+i] = p1[i + (size_t)c*ld]; m[(size_t)c*N+i] = p2[i + (size_t)c*ld];
        }
    }
    double cd[3], cm[3];                                                      /* :46-47 */
    for (int c = 0; c < 3; ++c) {
        double sd = 0, sm = 0;
        for (int i = 0; i < N; ++i) { sd += d[(size_t)c*N+i]; sm += m[(size_t)c*N+i]; }
        cd[c] = sd / N; cm[c] = sm / N;
    }
    double H[9];                                                              /* :55-58 */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < N; ++k) a += (m[(size_t)i*N+k] - cm[i]) * (d[(size_t)j*N+k] - cd[j]);
        H[i * 3 + j] = a;
    }
    free(d); free(m);
    double U[9], S[3], V[9], R[9], t[3];
    orc_svd3(H, U, S, V);                                                     /* :60 */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)                   /* :62 R = V*U' */
        R[i*3+j] = V[i*3+0]*U[j*3+0] + V[i*3+1]*U[j*3+1] + V[i*3+2]*U[j*3+2];
    for (int i = 0; i < 3; ++i) t[i] = cd[i] - (R[i*3]*cm[0] + R[i*3+1]*cm[1] + R[i*3+2]*cm[2]); /* :63 */
    for (int k = 0; k < 16; ++k) T[k] = 0.0;                                  /* :66-71 T = TF' */
    for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) T[k + 4*j] = R[j*3 + k];
    for (int j = 0; j < 3; ++j) T[3 + 4*j] = t[j];
    T[15] = 1.0;
    *empty = 0;
    return 0;
}

/* ------------------------------------------------------------------------ calcDists */
void orc_calc_dists(const double T[16], const double* p1, const double* p2, int n, int ld, double* d) {
    for (int i = 0; i < n; ++i) {                                             /* getInliersRANSAC.m:50-53 */
        double x = p2[i], y = p2[i + (size_t)ld], z = p2[i + 2*(size_t)ld];
        double tx = fma(x, T[0], fma(y, T[1], fma(z, T[2],  T[3])));
        double ty = fma(x, T[4], fma(y, T[5], fma(z, T[6],  T[7])));
        double tz = fma(x, T[8], fma(y, T[9], fma(z, T[10], T[11])));
        double dx = p1[i] - tx, dy = p1[i + (size_t)ld] - ty, dz = p1[i + 2*(size_t)ld] - tz;
        d[i] = fma(dz, dz, fma(dy, dy, dx * dx));
    }
}

/* --------------------------------------------------------------------------- sampler */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* The build's own "device RNG" sampler (NOT in the reference, which uses randperm,
 * ransac.m:42-43).  See oracle/pcreg_oracle.py:sample_table for the definition. */
void orc_sample_table(int n, int iterNum, int minPtNum, uint64_t seed, int32_t* out) {
    for (int p = 0; p < iterNum; ++p) {
        int32_t chosen[64];
        for (int j = 0; j < minPtNum; ++j) {
            uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)p * 16u + (uint64_t)j));
            uint32_t r = (uint32_t)(((h >> 32) * (uint64_t)(uint32_t)(n - j)) >> 32);
            /* insert into the ascending list of chosen indices */
            int k = 0;
            for (; k < j; ++k) { if (r >= (uint32_t)chosen[k]) ++r; else break; }
            for (int q = j; q > k; --q) chosen[q] = chosen[q - 1];
            chosen[k] = (int32_t)r;
            out[(size_t)p * minPtNum + j] = (int32_t)r + 1;
        }
    }
}

/* ----------------------------------------------------------------------------- ransac */
int orc_ransac(const double* p1, const double* p2, int n, int ld, const orc_ransac_opts* o,
               const int32_t* sample_idx, double Tout[16], int32_t* inlier_idx, int* n_inliers,
               int* num_success, int* max_inliers, int* failed,
               int32_t* iter_inl, int32_t* iter_inl_ref) {
    const int m = o->minPtNum, iters = o->iterNum;
    const int thInlr = orc_matlab_round(o->thInlrRatio * n);                  /* :28 */
    int32_t* cnt  = (int32_t*)calloc((size_t)iters, sizeof(int32_t));          /* :36 */
    int32_t* cntr = (int32_t*)calloc((size_t)iters, sizeof(int32_t));          /* :37 */
    double* TF = (double*)malloc(sizeof(double) * 16 * (size_t)iters);        /* :38 */
    char* has = (char*)calloc((size_t)iters, 1);
    double* dist = (double*)malloc(sizeof(double) * (size_t)n);
    double* s1 = (double*)malloc(sizeof(double) * 3 * (size_t)(n > m ? n : m));
    double* s2 = (double*)malloc(sizeof(double) * 3 * (size_t)(n > m ? n : m));
    for (int p = 0; p < iters; ++p) {                                         /* :40 */
        for (int j = 0; j < m; ++j) {                                         /* :42-45 */
            int i = sample_idx[(size_t)p * m + j] - 1;
            for (int c = 0; c < 3; ++c) { s1[j + c*m] = p1[i + (size_t)c*ld]; s2[j + c*m] = p2[i + (size_t)c*ld]; }
        }
        double f1[16]; int empty;
        orc_estimate_transform(s1, s2, m, m, f1, &empty);
        if (empty) continue;                       /* deviation: scores 0 */
        orc_calc_dists(f1, p1, p2, n, ld, dist);                              /* :48 */
        int c1 = 0;
        for (int i = 0; i < n; ++i) if (dist[i] < o->thDist) {                /* :49 */
            for (int c = 0; c < 3; ++c) { s1[c1 + (size_t)c*n] = p1[i + (size_t)c*ld]; s2[c1 + (size_t)c*n] = p2[i + (size_t)c*ld]; }
            ++c1;
        }
        cnt[p] = c1;                                                          /* :50 */
        if (c1 >= thInlr) {                                                   /* :53 */
            if (o->REFINE) {
                double f2[16];
                /* inlier rows were gathered with stride n: compact to stride c1 */
                for (int c = 1; c < 3; ++c) { memmove(s1 + (size_t)c*c1, s1 + (size_t)c*n, sizeof(double)*c1); memmove(s2 + (size_t)c*c1, s2 + (size_t)c*n, sizeof(double)*c1); }
                orc_estimate_transform(s1, s2, c1, c1, f2, &empty);           /* :55 */
                if (empty) continue;
                orc_calc_dists(f2, p1, p2, n, ld, dist);                      /* :56 */
                int c2 = 0; for (int i = 0; i < n; ++i) c2 += dist[i] < o->thDist;
                cntr[p] = c2;                                                 /* :58 */
                if (c2 >= thInlr) { memcpy(TF + 16*(size_t)p, f2, sizeof f2); has[p] = 1; }  /* :59-61 */
            } else { memcpy(TF + 16*(size_t)p, f1, sizeof f1); has[p] = 1; }  /* :63 */
        }
    }
    const int32_t* cc = o->REFINE ? cntr : cnt;                               /* :69-73 */
    int idx = 0; for (int p = 1; p < iters; ++p) if (cc[p] > cc[idx]) idx = p;
    *failed = !(iters > 0 && has[idx]);                                       /* :75-89 */
    if (*failed) {
        *n_inliers = 0; *num_success = 0; *max_inliers = 0;
        for (int k = 0; k < 16; ++k) Tout[k] = 0;
    } else {
        memcpy(Tout, TF + 16*(size_t)idx, sizeof(double) * 16);
        orc_calc_dists(Tout, p1, p2, n, ld, dist);                            /* :78 */
        int k = 0; for (int i = 0; i < n; ++i) if (dist[i] < o->thDist) inlier_idx[k++] = i + 1;  /* :92 */
        *n_inliers = k;
        int ns = 0; for (int p = 0; p < iters; ++p) ns += cc[p] >= thInlr;    /* :94-98 */
        *num_success = ns; *max_inliers = cc[idx];
    }
    if (iter_inl) memcpy(iter_inl, cnt, sizeof(int32_t) * (size_t)iters);
    if (iter_inl_ref) memcpy(iter_inl_ref, cntr, sizeof(int32_t) * (size_t)iters);
    free(cnt); free(cntr); free(TF); free(has); free(dist); free(s1); free(s2);
    return 0;
}

/* ------------------------------------------------------------- fp32 point KNN (D=3) */
#define KBLK 128
static void knn2_one(const float qx, const float qy, const float qz,
                     const float* mx, const float* my, const float* mz, int M,
                     int32_t idx[2], float dist[2]) {
    float d1 = INFINITY, d2 = INFINITY; int32_t i1 = -1, i2 = -1;
    float buf[KBLK];
    for (int j0 = 0; j0 < M; j0 += KBLK) {
        int nb = M - j0 < KBLK ? M - j0 : KBLK;
        for (int j = 0; j < nb; ++j) {
            float dx = qx - mx[j0 + j], dy = qy - my[j0 + j], dz = qz - mz[j0 + j];
            buf[j] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        }
        for (int j = 0; j < nb; ++j) {
            float d = buf[j];
            if (d < d2) {
                if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j0 + j; }
                else { d2 = d; i2 = j0 + j; }
            }
        }
    }
    idx[0] = i1; idx[1] = i2; dist[0] = d1; dist[1] = d2;
}

void orc_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                         int32_t* idx, float* dist, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < Q; ++i)
        knn2_one(q[i], q[i + (size_t)ldq], q[i + 2*(size_t)ldq], m, m + ldm, m + 2*(size_t)ldm, M,
                 idx + 2*(size_t)i, dist + 2*(size_t)i);
}

int orc_match_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                         float thr_abs, float max_ratio, int unique, uint32_t* pairs, int nthreads) {
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)Q);
    float* dist = (float*)malloc(sizeof(float) * 2 * (size_t)Q);
    orc_knn2_points_f32(q, Q, ldq, m, M, ldm, idx, dist, nthreads);
    int P = 0;
    int32_t* ci = (int32_t*)malloc(sizeof(int32_t) * (size_t)Q);
    int32_t* cj = (int32_t*)malloc(sizeof(int32_t) * (size_t)Q);
    for (int i = 0; i < Q; ++i) {
        float d1 = dist[2*(size_t)i], d2 = dist[2*(size_t)i + 1];
        if (!(d1 <= thr_abs)) continue;
        if (M > 1) { float t1 = d1, t2 = d2; if (t2 < 1e-6f) { t1 = 1.0f; t2 = 1.0f; } if (!(t1 / t2 <= max_ratio)) continue; }
        ci[P] = i; cj[P] = idx[2*(size_t)i]; ++P;
    }
    int out = 0;
    if (unique && P > 0) {
        float* mq = (float*)malloc(sizeof(float) * 3 * (size_t)P);
        for (int k = 0; k < P; ++k) for (int c = 0; c < 3; ++c) mq[k + (size_t)c*P] = m[cj[k] + (size_t)c*ldm];
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)P);
        float* bd = (float*)malloc(sizeof(float) * 2 * (size_t)P);
        orc_knn2_points_f32(mq, P, P, q, Q, ldq, bi, bd, nthreads);
        for (int k = 0; k < P; ++k) if (bi[2*(size_t)k] == ci[k]) { pairs[2*(size_t)out] = ci[k] + 1; pairs[2*(size_t)out+1] = cj[k] + 1; ++out; }
        free(mq); free(bi); free(bd);
    } else {
        for (int k = 0; k < P; ++k) { pairs[2*(size_t)k] = ci[k] + 1; pairs[2*(size_t)k+1] = cj[k] + 1; }
        out = P;
    }
    free(idx); free(dist); free(ci); free(cj);
    return out;
}

/* ------------------------------------------------- generic-D descriptor matching */
void orc_preprocess(const double* dS, int Q, int ldS, const double* dM, int M, int ldM, int D,
                    const orc_match_opts* o, double* outS, double* outM, int* Dp_out) {
    int Dp = D + (o->unnormalize ? 1 : 0);
    double col = 0.0;
    if (o->unnormalize) {                                                     /* getMatches.m:22-26 */
        double tot = 0.0;
        for (int i = 0; i < Q; ++i) { double s = 0; for (int d = 0; d < D; ++d) s += fabs(dS[i + (size_t)d*ldS]); tot += s; }
        for (int i = 0; i < M; ++i) { double s = 0; for (int d = 0; d < D; ++d) s += fabs(dM[i + (size_t)d*ldM]); tot += s; }
        col = o->norm_factor * (tot / (double)(Q + M));
    }
    for (int d = 0; d < Dp; ++d) {
        for (int i = 0; i < Q; ++i) { double v = d < D ? dS[i + (size_t)d*ldS] : col; outS[i + (size_t)d*Q] = o->change_metric ? pow(v, o->metric_factor) : v; }  /* :35-37 */
        for (int i = 0; i < M; ++i) { double v = d < D ? dM[i + (size_t)d*ldM] : col; outM[i + (size_t)d*M] = o->change_metric ? pow(v, o->metric_factor) : v; }
    }
    *Dp_out = Dp;
}

/* column-major n x D (ld) -> row-major n x D, optionally L2-normalised rows */
static double* to_rows(const double* f, int n, int ld, int D, int normalize) {
    double* r = (double*)malloc(sizeof(double) * (size_t)n * D);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int d = 0; d < D; ++d) { double v = f[i + (size_t)d*ld]; r[(size_t)i*D + d] = v; s = fma(v, v, s); }
        if (normalize) {
            double nrm = sqrt(s);
            if (nrm <= (double)FLT_EPSILON) for (int d = 0; d < D; ++d) r[(size_t)i*D + d] = 0.0;
            else for (int d = 0; d < D; ++d) r[(size_t)i*D + d] /= nrm;
        }
    }
    return r;
}
static inline double score_rows(const double* a, const double* b, int D, int metric) {
    double s = 0;
    if (metric == 0) for (int d = 0; d < D; ++d) s += fabs(a[d] - b[d]);
    else for (int d = 0; d < D; ++d) { double t = a[d] - b[d]; s = fma(t, t, s); }
    return s;
}

int orc_match_features(const double* fS, int Q, int ldS, const double* fM, int M, int ldM, int D,
                       const orc_match_opts* o, uint32_t* pairs, double* metric, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    if (Q <= 0 || M <= 0) return 0;
    double* A = to_rows(fS, Q, ldS, D, !o->prenormalized);
    double* B = to_rows(fM, M, ldM, D, !o->prenormalized);
    int32_t* i1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)Q);
    double* d1 = (double*)malloc(sizeof(double) * (size_t)Q);
    double* d2 = (double*)malloc(sizeof(double) * (size_t)Q);
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < Q; ++i) {
        double b1 = INFINITY, b2 = INFINITY; int32_t k1 = -1;
        for (int j = 0; j < M; ++j) {
            double s = score_rows(A + (size_t)i*D, B + (size_t)j*D, D, o->metric);
            if (s < b2) { if (s < b1) { b2 = b1; b1 = s; k1 = j; } else b2 = s; }
        }
        i1[i] = k1; d1[i] = b1; d2[i] = b2;
    }
    double maxval = o->metric == 1 ? 4.0 : 2.0 * sqrt((double)D);
    double thr = (o->matchThreshold * 0.01) * maxval;
    int32_t* ci = (int32_t*)malloc(sizeof(int32_t) * (size_t)Q); int P = 0;
    for (int i = 0; i < Q; ++i) {
        if (!(d1[i] <= thr)) continue;
        if (M > 1) { double t1 = d1[i], t2 = d2[i]; if (t2 < 1e-6) { t1 = 1.0; t2 = 1.0; } if (!(t1 / t2 <= o->maxRatio)) continue; }
        ci[P++] = i;
    }
    int out = 0;
    if (o->unique) {
        char* keep = (char*)calloc((size_t)(P > 0 ? P : 1), 1);
        #pragma omp parallel for schedule(static)
        for (int k = 0; k < P; ++k) {
            int j = i1[ci[k]]; double b = INFINITY; int bi = -1;
            for (int i = 0; i < Q; ++i) { double s = score_rows(A + (size_t)i*D, B + (size_t)j*D, D, o->metric); if (s < b) { b = s; bi = i; } }
            keep[k] = (bi == ci[k]);
        }
        for (int k = 0; k < P; ++k) if (keep[k]) { pairs[2*(size_t)out] = ci[k] + 1; pairs[2*(size_t)out+1] = i1[ci[k]] + 1; if (metric) metric[out] = d1[ci[k]]; ++out; }
        free(keep);
    } else {
        for (int k = 0; k < P; ++k) { pairs[2*(size_t)k] = ci[k] + 1; pairs[2*(size_t)k+1] = i1[ci[k]] + 1; if (metric) metric[k] = d1[ci[k]]; }
        out = P;
    }
    free(A); free(B); free(i1); free(d1); free(d2); free(ci);
    return out;
}

int orc_get_matches(const double* dS, int Q, int ldS, const double* dM, int M, int ldM, int D,
                    const orc_match_opts* o, uint32_t* pairs, double* metric, int nthreads) {
    int Dp = D + (o->unnormalize ? 1 : 0);
    double* s = (double*)malloc(sizeof(double) * (size_t)Q * Dp);
    double* m = (double*)malloc(sizeof(double) * (size_t)M * Dp);
    orc_preprocess(dS, Q, ldS, dM, M, ldM, D, o, s, m, &Dp);
    int P = orc_match_features(s, Q, Q, m, M, M, Dp, o, pairs, metric, nthreads);
    free(s); free(m);
    return P;
}

/* ----------------------------------------------------------------- AlignPoints_KNN */
/* cyclic Jacobi eigen-decomposition of a symmetric 3x3 (row-major); V columns = vectors */
static void jacobi_eig3(double A[9], double V[9]) {
    memset(V, 0, 9 * sizeof(double)); V[0] = V[4] = V[8] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        double dia = fabs(A[0]) + fabs(A[4]) + fabs(A[8]);
        if (off <= 1e-300 || off <= DBL_EPSILON * 1e-3 * dia) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            double apq = A[p*3+q];
            if (apq == 0.0) continue;
            double theta = (A[q*3+q] - A[p*3+p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
            double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
            for (int k = 0; k < 3; ++k) { double a = A[k*3+p], b = A[k*3+q]; A[k*3+p] = c*a - s*b; A[k*3+q] = s*a + c*b; }
            for (int k = 0; k < 3; ++k) { double a = A[p*3+k], b = A[q*3+k]; A[p*3+k] = c*a - s*b; A[q*3+k] = s*a + c*b; }
            for (int k = 0; k < 3; ++k) { double a = V[k*3+p], b = V[k*3+q]; V[k*3+p] = c*a - s*b; V[k*3+q] = s*a + c*b; }
        }
    }
}
typedef struct { double d; int i; } di_t;
static int cmp_di(const void* a, const void* b) {
    const di_t* x = (const di_t*)a; const di_t* y = (const di_t*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);      /* stable: equal keys keep index order */
}

int orc_align_points_knn(const double* pts, int n, int ld, int C1, int C2,
                         double* aligned, double coeff_out[9], double c[3]) {
    if (n < 2) return -1;
    for (int k = 0; k < 3; ++k) { double s = 0; for (int i = 0; i < n; ++i) s += pts[i + (size_t)k*ld]; c[k] = s / n; }  /* :17 */
    int K = orc_matlab_round(n * 0.85);                                       /* :20-21 */
    di_t* o = (di_t*)malloc(sizeof(di_t) * (size_t)n);
    for (int i = 0; i < n; ++i) {                                             /* :22-23 */
        double x = pts[i] - c[0], y = pts[i + (size_t)ld] - c[1], z = pts[i + 2*(size_t)ld] - c[2];
        o[i].d = sqrt(x*x + y*y + z*z); o[i].i = i;
    }
    qsort(o, (size_t)n, sizeof(di_t), cmp_di);                                /* :24 */
    double* pk = (double*)malloc(sizeof(double) * 3 * (size_t)K);             /* :25-26 */
    for (int r = 0; r < K; ++r) for (int k = 0; k < 3; ++k) pk[r + (size_t)k*K] = pts[o[r].i + (size_t)k*ld] - c[k];
    free(o);
    double mu[3] = {0, 0, 0};                                                 /* :30-34 pca */
    if (!C1) for (int k = 0; k < 3; ++k) { double s = 0; for (int r = 0; r < K; ++r) s += pk[r + (size_t)k*K]; mu[k] = s / K; }
    double C[9];
    double dof = C1 ? (double)K : (double)(K - 1); if (dof < 1) dof = 1;
    for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) {
        double s = 0; for (int r = 0; r < K; ++r) s += (pk[r + (size_t)a*K] - mu[a]) * (pk[r + (size_t)b*K] - mu[b]);
        C[a*3+b] = C[b*3+a] = s / dof;
    }
    double V[9]; jacobi_eig3(C, V);
    int ord[3] = {0, 1, 2};                                                   /* descending eigenvalue */
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (C[ord[b]*3+ord[b]] > C[ord[a]*3+ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double coeff[9];   /* row-major here: coeff[r*3+col] */
    for (int col = 0; col < 3; ++col) {
        int src = ord[col], mi = 0;
        for (int r = 1; r < 3; ++r) if (fabs(V[r*3+src]) > fabs(V[mi*3+src])) mi = r;
        double sg = V[mi*3+src] < 0 ? -1.0 : 1.0;            /* largest-|.| entry positive */
        for (int r = 0; r < 3; ++r) coeff[r*3+col] = sg * V[r*3+src];
    }
    /* sign disambiguation, :37-56 */
    int posx = 0, posz = 0;
    if (C2) {
        for (int i = 0; i < n; ++i) {
            double x = pts[i], y = pts[i + (size_t)ld], z = pts[i + 2*(size_t)ld];
            posx += (x*coeff[0] + y*coeff[3] + z*coeff[6]) > 0;
            posz += (x*coeff[2] + y*coeff[5] + z*coeff[8]) > 0;
        }
    } else {
        for (int r = 0; r < K; ++r) {
            double x = pk[r] - mu[0], y = pk[r + (size_t)K] - mu[1], z = pk[r + 2*(size_t)K] - mu[2];
            posx += (x*coeff[0] + y*coeff[3] + z*coeff[6]) > 0;
            posz += (x*coeff[2] + y*coeff[5] + z*coeff[8]) > 0;
        }
    }
    free(pk);
    double xs = (2.0 * posx >= (double)n) ? 1.0 : -1.0;                       /* :45,49 (k = N) */
    double zs = (2.0 * posz >= (double)n) ? 1.0 : -1.0;                       /* :46,50 */
    double Mx[9]; for (int r = 0; r < 3; ++r) { Mx[r*3] = coeff[r*3]*xs; Mx[r*3+1] = coeff[r*3+1]; Mx[r*3+2] = coeff[r*3+2]*zs; }
    double ys = Mx[0]*(Mx[4]*Mx[8]-Mx[5]*Mx[7]) - Mx[1]*(Mx[3]*Mx[8]-Mx[5]*Mx[6]) + Mx[2]*(Mx[3]*Mx[7]-Mx[4]*Mx[6]);  /* :53 */
    double cu[9]; for (int r = 0; r < 3; ++r) { cu[r*3] = coeff[r*3]*xs; cu[r*3+1] = coeff[r*3+1]*ys; cu[r*3+2] = coeff[r*3+2]*zs; }    /* :56 */
    for (int i = 0; i < n; ++i) {                                             /* :59 */
        double x = pts[i], y = pts[i + (size_t)ld], z = pts[i + 2*(size_t)ld];
        for (int col = 0; col < 3; ++col) aligned[i + (size_t)col*n] = x*cu[col] + y*cu[3+col] + z*cu[6+col];
    }
    for (int r = 0; r < 3; ++r) for (int col = 0; col < 3; ++col) coeff_out[r + 3*col] = cu[r*3+col];
    return 0;
}

/* ------------------------------------------- getSpacialHistogramDescriptors (cfg 4) */
#define ORC_NR 10
#define ORC_NT 7
#define ORC_NP 14
static int hist_loc(double x, const double* e, int ne) {   /* histcounts bin, 1-based, 0 = outside / NaN */
    if (!(x >= e[0]) || !(x <= e[ne - 1])) return 0;
    if (x == e[ne - 1]) return ne - 1;
    int k = 1; while (k < ne && x >= e[k]) ++k;            /* e[k-1] <= x < e[k] */
    return k;
}
static int desc_one(const double* pts, int P, int ld, const double c[3], const orc_desc_opts* o, int single_mode,
                    const double* rb, const double* tb, const double* pb, double* out /*980*/) {
    const double R = o->R;
    /* getLocalPoints.m:8-35: open box, then dists < R, min <= n <= max; points relative to c */
    int cap = 1024, n = 0;
    double* L = (double*)malloc(sizeof(double) * 3 * cap);
    if (single_mode) {
        /* MATLAB's arithmetic when either input is single: every binary operation on a single and a double operand runs in
         * single, so :8-25 are element-wise float operations, each rounded once (this file is compiled with
         * -ffp-contract=off: no fused multiply-add, no excess precision on x86-64 SSE).  1: c single -> xLim single (:8-10);
         * 2: cloud single, c double -> xLim formed in double, converted in the comparison (:11-13). */
        const float R32 = (float)R, c32[3] = { (float)c[0], (float)c[1], (float)c[2] };
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            if (single_mode == 1) { lo[a] = c32[a] + (-R32); hi[a] = c32[a] + R32; }
            else { lo[a] = (float)(c[a] - R); hi[a] = (float)(c[a] + R); }
        }
        for (int i = 0; i < P; ++i) {
            const float x = (float)pts[i], y = (float)pts[i + (size_t)ld], z = (float)pts[i + 2*(size_t)ld];
            if (!(x > lo[0] && x < hi[0] && y > lo[1] && y < hi[1] && z > lo[2] && z < hi[2])) continue;      /* :11-13 */
            const float dx = x - c32[0], dy = y - c32[1], dz = z - c32[2];                                     /* :23 */
            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            const float s2 = xx + yy; const float s3 = s2 + zz;
            if (!(sqrtf(s3) < R32)) continue;                                                                 /* :24-25 */
            if (n == cap) { cap *= 2; L = (double*)realloc(L, sizeof(double) * 3 * cap); }
            L[3*n] = (double)dx; L[3*n+1] = (double)dy; L[3*n+2] = (double)dz; ++n;
        }
    } else
    for (int i = 0; i < P; ++i) {
        double x = pts[i], y = pts[i + (size_t)ld], z = pts[i + 2*(size_t)ld];
        if (!(x > c[0] - R && x < c[0] + R && y > c[1] - R && y < c[1] + R && z > c[2] - R && z < c[2] + R)) continue;
        double dx = x - c[0], dy = y - c[1], dz = z - c[2];
        if (!(sqrt(dx*dx + dy*dy + dz*dz) < R)) continue;
        if (n == cap) { cap *= 2; L = (double*)realloc(L, sizeof(double) * 3 * cap); }
        L[3*n] = dx; L[3*n+1] = dy; L[3*n+2] = dz; ++n;
    }
    int ok = 0;
    if (n >= 1 && n >= o->min_pts && n <= o->max_pts) {
        int k = n;
        if (!(o->k >= 1.0)) {                                                 /* :75-84 */
            k = orc_matlab_round(n * o->k);
            double cen[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) { double s = 0; for (int i = 0; i < n; ++i) s += L[3*i+a]; cen[a] = s / n; }
            di_t* ord = (di_t*)malloc(sizeof(di_t) * (size_t)n);
            for (int i = 0; i < n; ++i) { double x = L[3*i]-cen[0], y = L[3*i+1]-cen[1], z = L[3*i+2]-cen[2]; ord[i].d = sqrt(x*x+y*y+z*z); ord[i].i = i; }
            qsort(ord, (size_t)n, sizeof(di_t), cmp_di);
            double* S = (double*)malloc(sizeof(double) * 3 * (size_t)n);
            for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) S[3*i+a] = L[3*ord[i].i+a];
            memcpy(L, S, sizeof(double) * 3 * (size_t)n); free(S); free(ord);
        }
        if (k >= 2) {
            double mu[3], C[9], V[9];                                         /* pca(pts_k,'eig'), :91 */
            for (int a = 0; a < 3; ++a) { double s = 0; for (int i = 0; i < k; ++i) s += L[3*i+a]; mu[a] = s / k; }
            for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) {
                double s = 0; for (int i = 0; i < k; ++i) s += (L[3*i+a]-mu[a]) * (L[3*i+b]-mu[b]);
                C[a*3+b] = C[b*3+a] = s / (double)(k - 1);
            }
            jacobi_eig3(C, V);
            int od[3] = {0, 1, 2};
            for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (C[od[b]*3+od[b]] > C[od[a]*3+od[a]]) { int t = od[a]; od[a] = od[b]; od[b] = t; }
            double var[3] = { C[od[0]*3+od[0]], C[od[1]*3+od[1]], C[od[2]*3+od[2]] };
            if (!(var[0] / var[1] < o->thVar[0]) && !(var[1] / var[2] < o->thVar[1])) {   /* :118-121 */
                double co[9];
                for (int col = 0; col < 3; ++col) {
                    int src = od[col], mi = 0;
                    for (int r = 1; r < 3; ++r) if (fabs(V[r*3+src]) > fabs(V[mi*3+src])) mi = r;
                    double sg = V[mi*3+src] < 0 ? -1.0 : 1.0;
                    for (int r = 0; r < 3; ++r) co[r*3+col] = sg * V[r*3+src];
                }
                double cu[9] = {1,0,0, 0,1,0, 0,0,1};
                if (o->ALIGN_POINTS) {                                        /* :128-145, vote over k rows */
                    int px = 0, pz = 0;
                    for (int i = 0; i < k; ++i) {
                        double x = L[3*i]-mu[0], y = L[3*i+1]-mu[1], z = L[3*i+2]-mu[2];
                        px += (x*co[0] + y*co[3] + z*co[6]) > 0; pz += (x*co[2] + y*co[5] + z*co[8]) > 0;
                    }
                    double xs = (2.0*px >= (double)k) ? 1.0 : -1.0, zs = (2.0*pz >= (double)k) ? 1.0 : -1.0;
                    double M[9]; for (int r = 0; r < 3; ++r) { M[r*3] = co[r*3]*xs; M[r*3+1] = co[r*3+1]; M[r*3+2] = co[r*3+2]*zs; }
                    double ys = M[0]*(M[4]*M[8]-M[5]*M[7]) - M[1]*(M[3]*M[8]-M[5]*M[6]) + M[2]*(M[3]*M[7]-M[4]*M[6]);
                    for (int r = 0; r < 3; ++r) { cu[r*3] = co[r*3]*xs; cu[r*3+1] = co[r*3+1]*ys; cu[r*3+2] = co[r*3+2]*zs; }
                }
                for (int b = 0; b < ORC_NR*ORC_NT*ORC_NP; ++b) out[b] = 0.0;
                for (int i = 0; i < n; ++i) {                                 /* :150-171 */
                    double x0 = L[3*i], y0 = L[3*i+1], z0 = L[3*i+2], x = x0, y = y0, z = z0;
                    if (o->ALIGN_POINTS) { x = x0*cu[0] + y0*cu[3] + z0*cu[6]; y = x0*cu[1] + y0*cu[4] + z0*cu[7]; z = x0*cu[2] + y0*cu[5] + z0*cu[8]; }
                    double r = sqrt(x*x + y*y + z*z), th = acos(z / r), ph = atan2(y, y);
                    int lr = hist_loc(r, rb, ORC_NR+1), lt = hist_loc(th, tb, ORC_NT+1), lp = hist_loc(ph, pb, ORC_NP+1);
                    if (lr > 0 && lt > 0 && lp > 0) out[(lr-1) + ORC_NR*(lt-1) + ORC_NR*ORC_NT*(lp-1)] += 1.0;
                }
                ok = 1;
            }
        }
    }
    free(L);
    return ok;
}

int orc_spatial_histogram_descriptors(const double* pts, int P, int ld, const double* kp, int S, int ldk,
                                      const orc_desc_opts* o, double* feat, double* desc, int nthreads) {
    return orc_spatial_histogram_descriptors_sm(pts, P, ld, kp, S, ldk, o, 0, feat, desc, nthreads);
}
/* single_mode: 0 double data; 1 / 2 MATLAB's single arithmetic in getLocalPoints (see desc_one); feat / desc are double
 * either way (getSpacialHistogramDescriptors.m:61-62) */
int orc_spatial_histogram_descriptors_sm(const double* pts, int P, int ld, const double* kp, int S, int ldk,
                                         const orc_desc_opts* o, int single_mode, double* feat, double* desc, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const int ND = ORC_NR*ORC_NT*ORC_NP;
    double rb[ORC_NR+1], tb[ORC_NT+1], pb[ORC_NP+1];
    const double r3 = o->R * o->R * o->R, pi = 3.14159265358979323846;
    for (int k = 0; k <= ORC_NR; ++k) rb[k] = cbrt(k * (r3 / ORC_NR));
    for (int k = 0; k <= ORC_NT; ++k) tb[k] = k * (pi / ORC_NT);
    for (int k = 0; k <= ORC_NP; ++k) pb[k] = -pi + k * (2 * pi / ORC_NP);
    char* valid = (char*)calloc((size_t)(S > 0 ? S : 1), 1);
    double* tmp = (double*)malloc(sizeof(double) * (size_t)ND * (size_t)(S > 0 ? S : 1));
    #pragma omp parallel for schedule(dynamic, 4)
    for (int s = 0; s < S; ++s) {
        double c[3] = { kp[s], kp[s + (size_t)ldk], kp[s + 2*(size_t)ldk] };
        valid[s] = (char)desc_one(pts, P, ld, c, o, single_mode, rb, tb, pb, tmp + (size_t)s * ND);
    }
    int V = 0;
    for (int s = 0; s < S; ++s) if (valid[s]) {                               /* :177-179 */
        for (int a = 0; a < 3; ++a) feat[3*(size_t)V + a] = kp[s + (size_t)a*ldk];
        memcpy(desc + (size_t)V * ND, tmp + (size_t)s * ND, sizeof(double) * ND);
        ++V;
    }
    free(valid); free(tmp);
    return V;
}
