/* oracle/pcreg_oracle.h -- plain-C restatement of the PCReg hot path (CPU, IEEE double).
 *
 * TEST INFRASTRUCTURE ONLY: the checker and the reported CPU baseline.  Nothing in
 * pcreg_amd/ links, loads or calls this.  See pcreg_oracle.c for the reference
 * file:line each function follows and for the pinning status.
 *
 * Conventions (MATLAB's): matrices are column-major with an explicit leading
 * dimension; point sets are n x 3 (x = p[i], y = p[i+ld], z = p[i+2*ld]); T is a
 * column-major 4x4 with [p 1]*T row-vector semantics; indices are 1-based where
 * they cross the boundary.
 */
#ifndef PCREG_ORACLE_H
#define PCREG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t minPtNum;
    int32_t iterNum;
    double  thDist;
    double  thInlrRatio;
    int32_t REFINE;
} orc_ransac_opts;

typedef struct {
    int32_t metric;          /* 0 = SAD, 1 = SSD                                   */
    double  matchThreshold;  /* percent (0..100] of the max unit-vector distance   */
    double  maxRatio;
    int32_t unique;
    int32_t prenormalized;
    int32_t unnormalize;     /* getMatches par.UNNORMALIZE                          */
    double  norm_factor;
    int32_t change_metric;   /* getMatches par.CHANGE_METRIC                        */
    double  metric_factor;
} orc_match_opts;

int  orc_matlab_round(double x);
void orc_svd3(const double A[9], double U[9], double S[3], double V[9]);  /* row-major 3x3 */
int  orc_rank_nx3(const double* p, int n, int ld);
int  orc_estimate_transform(const double* p1, const double* p2, int n, int ld,
                            double T[16], int* empty);
void orc_calc_dists(const double T[16], const double* p1, const double* p2, int n, int ld,
                    double* d);
void orc_sample_table(int n, int iterNum, int minPtNum, uint64_t seed, int32_t* out);
int  orc_ransac(const double* p1, const double* p2, int n, int ld, const orc_ransac_opts* o,
                const int32_t* sample_idx /* [iterNum][minPtNum], 1-based */,
                double T[16], int32_t* inlier_idx, int* n_inliers, int* num_success,
                int* max_inliers, int* failed,
                int32_t* iter_inl /* optional [iterNum] */, int32_t* iter_inl_ref /* optional */);

/* fp32 3-D point search: two nearest model points per query, d = fmaf chain. */
void orc_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                         int32_t* idx /* [Q][2] 0-based, -1 if absent */, float* dist /* [Q][2] */,
                         int nthreads);
int  orc_match_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                          float thr_abs, float max_ratio, int unique,
                          uint32_t* pairs /* [cap Q][2] row-major, 1-based */, int nthreads);

/* generic-D descriptor matching (getMatches + documented matchFeatures semantics) */
void orc_preprocess(const double* dS, int Q, int ldS, const double* dM, int M, int ldM, int D,
                    const orc_match_opts* o, double* outS /* Q x Dp, ld Q */,
                    double* outM /* M x Dp, ld M */, int* Dp);
int  orc_match_features(const double* fS, int Q, int ldS, const double* fM, int M, int ldM, int D,
                        const orc_match_opts* o, uint32_t* pairs /* [cap Q][2] */,
                        double* metric /* optional [cap Q] */, int nthreads);
int  orc_get_matches(const double* dS, int Q, int ldS, const double* dM, int M, int ldM, int D,
                     const orc_match_opts* o, uint32_t* pairs, double* metric, int nthreads);

typedef struct {
    int32_t min_pts;      /* options.min_pts                                             */
    int32_t max_pts;      /* options.max_pts (INT32_MAX for inf)                         */
    double  R;            /* options.R                                                   */
    double  thVar[2];     /* options.thVar                                               */
    double  k;            /* options.k: fraction in (0,1); 1 (or 'all' -> 1) = all points */
    int32_t ALIGN_POINTS; /* options.ALIGN_POINTS                                        */
} orc_desc_opts;
/* getSpacialHistogramDescriptors: feat [V][3] and desc [V][980] ROW-major, returns V. */
int  orc_spatial_histogram_descriptors(const double* pts, int P, int ld, const double* kp, int S, int ldk,
                                       const orc_desc_opts* o, double* feat, double* desc, int nthreads);
int  orc_spatial_histogram_descriptors_sm(const double* pts, int P, int ld, const double* kp, int S, int ldk,
                                          const orc_desc_opts* o, int single_mode, double* feat, double* desc, int nthreads);

int  orc_align_points_knn(const double* pts, int n, int ld, int C1, int C2,
                          double* aligned /* n x 3, ld n */, double coeff[9] /* col-major */,
                          double c[3]);
#ifdef __cplusplus
}
#endif
#endif
