/* include/pcreg.h -- C ABI of libpcreg_hip.so: the MI355X (gfx950) implementation of
 * the PCReg correspondence-search + RANSAC rigid-alignment hot path.
 *
 * The reference (LCJebe/PCReg) is pure MATLAB and has no FFI layer; the boundary it
 * exposes for this path is the set of MATLAB function signatures below.  Each entry
 * point names the reference function it replaces (file:line in the reference tree);
 * INTEGRATION.md shows the MEX binding a maintainer adds on the MATLAB side.
 *
 * Conventions (MATLAB's, so a MEX shim passes mxGetPr() pointers straight through):
 *   - matrices are column-major with an explicit leading dimension `ld` (>= rows);
 *     a point set is n x 3: x = p[i], y = p[i+ld], z = p[i+2*ld];
 *   - T is a column-major 4x4 used as [p 1]*T (quickTF.m:5-7): rotation in
 *     T(1:3,1:3), translation in T(4,1:3);
 *   - indices that cross the boundary are 1-based, pairs are uint32 like matchFeatures';
 *   - every function returns 0 on success or a PCREG_E_* code; pcreg_last_error()
 *     gives the text.  "RANSAC found nothing" is NOT an error: it is reported through
 *     *failed = 1 with T zeroed, mirroring ransac.m:77-89 (empty T, zeros).
 *   - there is no CPU fallback: without a usable HIP device every compute entry
 *     point returns PCREG_E_NODEVICE.
 *
 * Two tiers:
 *   pcreg_*      host tier  -- pointers are HOST memory (what MEX hands over); the call
 *                              stages to HBM, runs the kernels, copies results back.
 *   pcreg_dev_*  device tier -- pointers are DEVICE memory and work is enqueued on the
 *                              caller's HIP stream (passed as void*); nothing
 *                              synchronises.  Used by resident pipelines, the
 *                              multi-GPU host code and bench.py.
 */
#ifndef PCREG_H
#define PCREG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCREG_OK            0
#define PCREG_E_ARG         1   /* bad argument (null pointer, negative size, ld < n ...) */
#define PCREG_E_HIP         2   /* a HIP runtime call failed                              */
#define PCREG_E_NODEVICE    3   /* no usable gfx950 device                                */
#define PCREG_E_WORKSPACE   4   /* caller's workspace too small (device tier)             */

#define PCREG_METRIC_SAD 0
#define PCREG_METRIC_SSD 1

/* ransacCoef of ransac.m:7-12,23-34 (+ the build's sampler seed). */
typedef struct pcreg_ransac_opts {
    int32_t  minPtNum;     /* ransac.m:23; estimateTransform needs >= 3              */
    int32_t  iterNum;      /* ransac.m:24                                            */
    double   thDist;       /* ransac.m:26; compared with the SQUARED distance         */
    double   thInlrRatio;  /* ransac.m:25; thInlr = round(thInlrRatio*ptNum), :28     */
    int32_t  REFINE;       /* ransac.m:29                                            */
    int32_t  VERBOSE;      /* ransac.m:30-34; printing is done by the host wrappers   */
    uint64_t seed;         /* used only when sample_idx == NULL (built-in sampler)    */
} pcreg_ransac_opts;

/* `par` of getMatches.m:5-9,22,35,51-56 + matchFeatures' Prenormalized. */
typedef struct pcreg_match_opts {
    int32_t metric;          /* PCREG_METRIC_SAD | PCREG_METRIC_SSD   (par.Metric)     */
    double  matchThreshold;  /* percent, par.MatchThreshold                            */
    double  maxRatio;        /* par.MaxRatio                                           */
    int32_t unique;          /* par.Unique                                             */
    int32_t prenormalized;   /* matchFeatures 'Prenormalized' (getMatches passes 0)    */
    int32_t unnormalize;     /* par.UNNORMALIZE, getMatches.m:22-26                    */
    double  norm_factor;     /* par.norm_factor                                        */
    int32_t change_metric;   /* par.CHANGE_METRIC, getMatches.m:35-37                  */
    double  metric_factor;   /* par.metric_factor                                      */
} pcreg_match_opts;

/* ---- library ------------------------------------------------------------------- */
const char* pcreg_last_error(void);
const char* pcreg_version(void);
int  pcreg_device_count(int* count);
int  pcreg_set_device(int ordinal);          /* one process per GPU: call once per rank */
int  pcreg_device_name(char* buf, int cap);  /* e.g. "gfx950:..."                       */
/* Test hook, not part of the reference's interface: selects the OTHER side of a certified fast path (process-wide), so that
 * the parity tests can run both sides inside one process.  Every setting returns the same indices and counts.  Keys:
 * "knn_exact", "match_exact", "match_force_fallback" (1, 2), "ransac_fused", "ransac_nolane", "ransac_f64score",
 * "ransac_resident_f64", "align_times", "align_shape", "seg_debug", "seg_batched", "seg_wave_finalize", "match_stats"; value 0 restores the default.  The library reads NO
 * environment variable (tests/test_abi.py greps the binary).  PCREG_E_ARG for an unknown key. */
int  pcreg_debug_set(const char* key, int value);
/* With pcreg_debug_set("match_stats", 1): the counters of the certified SAD matcher summed over the calls since the last
 * reset -- out[0] queries finalised, [1] candidates re-scored exactly (fp64), [2] queries the certificate left unproven,
 * [3] queries handed to the exhaustive exact-rows kernel, [4] Unique back-check items (segmented form), [5] back-check items
 * handed to the exhaustive kernel, [6] matcher calls, [7] segments.  Synchronises the device.  bench.py reports them so that
 * a throughput figure says how much of it the certificates carried. */
int  pcreg_debug_match_stats(long long out[8], int reset);

/* ---- host tier ------------------------------------------------------------------ */

/* estimateTransform.m:2  T = estimateTransform(pts1, pts2), [pts2,1]*T = [pts1,1].
 * *empty = 1 reproduces the `T = []` return of estimateTransform.m:11-14. */
int pcreg_estimate_transform(const double* pts1, const double* pts2, int n, int ld,
                             double T[16], int* empty);

/* getInliersRANSAC.m:46  d = calcDists(T, pts1, pts2): squared distances, length n. */
int pcreg_calc_dists(const double T[16], const double* pts1, const double* pts2, int n, int ld,
                     double* d);

/* ransac.m:1  [T, inlierIdx, numSuccess, maxInliers, ratio] = ransac(pts1, pts2,
 * ransacCoef, @estimateTransform, @calcDists).
 * sample_idx: [iterNum][minPtNum] 1-based indices, hypothesis-major (pass the
 * transpose of a MATLAB iterNum x minPtNum table, i.e. minPtNum x iterNum
 * column-major), replacing `randperm(ptNum)(1:minPtNum)` of ransac.m:42-43; NULL
 * selects the built-in counter-based sampler seeded by opts->seed.
 * inlier_idx has capacity n (1-based, ascending); iter_inl / iter_inl_ref (optional,
 * may be NULL, length iterNum) receive inlrNum / inlrNum_refined of ransac.m:36-37. */
int pcreg_ransac(const double* pts1, const double* pts2, int n, int ld,
                 const pcreg_ransac_opts* opts, const int32_t* sample_idx,
                 double T[16], int32_t* inlier_idx, int* n_inliers, int* num_success,
                 int* max_inliers, int* failed, int32_t* iter_inl, int32_t* iter_inl_ref);

/* The same for B independent registrations in one launch (the parfor of
 * completeExperimentFast.m:201-225).  Registration b owns rows
 * [offsets[b], offsets[b+1]) of the concatenated pts1/pts2 (total = offsets[B]) and
 * the sample table rows [b*iterNum, (b+1)*iterNum) (or seed + b).  Outputs are
 * arrays of length B (T: 16*B); inlier_idx is concatenated with the same offsets. */
int pcreg_ransac_batched(const double* pts1, const double* pts2, int total, int ld,
                         const int32_t* offsets, int B, const pcreg_ransac_opts* opts,
                         const int32_t* sample_idx, double* T, int32_t* inlier_idx,
                         int32_t* n_inliers, int32_t* num_success, int32_t* max_inliers,
                         int32_t* failed);

/* fp32 3-D point search (the "KNN" of BASELINE.json's metric): for each query the two
 * nearest model points, squared distance fmaf(dz,dz,fmaf(dy,dy,dx*dx)), ties to the
 * lowest index.  idx [Q][2] 0-based (-1 when M < 2), dist [Q][2]. */
int pcreg_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                          int32_t* idx, float* dist);

/* matchFeatures' filter chain on raw fp32 points (Prenormalized, SSD, absolute
 * threshold thr_abs on the squared distance, ratio test, optional Unique back-check).
 * pairs: capacity Q x 2, ROW-major [k][0]=query, [k][1]=model, 1-based, ascending
 * query.  *P receives the number of pairs. */
int pcreg_match_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                           float thr_abs, float max_ratio, int unique,
                           uint32_t* pairs, int* P);

/* The same against a model that is uploaded and prepared ONCE (host tier of the handle below): MATLAB keeps the
 * handle as a uint64 and matches any number of surfaces against it (completeExperimentFast.m:131-149). */
typedef struct pcreg_model pcreg_model;
int pcreg_model_create(const float* m, int M, int ldm, pcreg_model** model);
int pcreg_model_destroy(pcreg_model* model);
int pcreg_model_match_points_f32(pcreg_model* model, const float* q, int Q, int ldq, float thr_abs, float max_ratio,
                                 int unique, uint32_t* pairs, int* P);

/* getLocalPoints.m:8-35  [pts_sphere, dists] = getLocalPoints(pts, R, c, min_points, max_points): the points of the cloud strictly
 * inside the open box AND the open ball of radius R around c, RELATIVE to c, in the cloud's order; [] when the box holds fewer
 * than min_points (:17) or the ball fewer than min_points / more than max_points (:31).  pts: N x 3 column-major (ld >= N).
 * pts_sphere: capacity N x 3, written as n_out x 3 column-major with leading dimension *n_out; dists (or NULL): capacity N;
 * *n_out = rows returned, 0 for MATLAB's [].  single_mode as in pcreg_spatial_histogram_descriptors_mixed (0 double; 1 keypoint
 * single, 2 only the cloud single: MATLAB's element-wise single arithmetic, values passed exactly widened, outputs are those single
 * values widened). */
int pcreg_get_local_points(const double* pts, int N, int ld, double R, const double c[3], double min_points, double max_points,
                           int single_mode, double* pts_sphere, double* dists, int* n_out);

/* getMatches for S row subsets of ONE model descriptor set in one call: segment s = getMatches(descSurface,
 * descModel(rows_s + 1, :), par) with rows_s = seg_rows[seg_off[s] .. seg_off[s+1]) (0-based, ascending, HOST arrays) -- the
 * per-sphere calls that completeExperimentFast.m:131-149 runs under parfor.  Descriptors as MATLAB holds them (n x D
 * column-major doubles, ld >= n).  pairs_all [S][Q][2]: segment s's pairs start at pairs_all + s*Q*2, n_pairs[s] of them,
 * 1-based, the model index counting within the segment like the per-sphere call's.  Same pairs as S calls of
 * pcreg_get_matches (the powered columns and the approximate scores are computed once for all segments).
 * Metric SAD; S <= 65535. */
int pcreg_get_matches_segmented(const double* descSurface, int Q, int ldS, const double* descModel, int VM, int ldM, int D,
                                const int32_t* seg_rows, const int32_t* seg_off, int S, const pcreg_match_opts* par,
                                uint32_t* pairs_all, int32_t* n_pairs);

/* getMatches.m:51  matchFeatures(features1, features2, 'Method',..,'MatchThreshold',..,
 * 'MaxRatio',..,'Metric',..,'Unique',..) on Q x D / M x D double features (exact
 * search; 'Approximate' is answered exactly).  Only the matchFeatures fields of opts
 * are used.  pairs as above; metric (optional) receives matchMetric. */
int pcreg_match_features(const double* f1, int Q, int ld1, const double* f2, int M, int ld2, int D,
                         const pcreg_match_opts* opts, uint32_t* pairs, double* metric, int* P);

/* getMatches.m:1  matches = getMatches(descSurface, descModel, par): append the
 * constant column (:22-26), element-wise power (:35-37), then matchFeatures (:51-56). */
int pcreg_get_matches(const double* descSurface, int Q, int ldS, const double* descModel, int M,
                      int ldM, int D, const pcreg_match_opts* par, uint32_t* pairs, double* metric,
                      int* P);

/* One surface set against MANY row subsets of one model set -- the loop of completeExperimentFast.m:101-150
 *     descCur = descModel(mask, :);  matches = getMatches(descSurface, descCur, par);        (:121-149, hundreds of spheres)
 * with both sets uploaded ONCE.  pcreg_desc_set_create copies an n x D column-major double matrix (ld >= n) to the device;
 * pcreg_get_matches_on_sets(surface, model, rows, n_rows, ...) is pcreg_get_matches(descSurface, descModel(rows + 1, :), ...)
 * -- the same kernels on the same values, the same pairs bit for bit -- with rows = the 0-based ascending row numbers of the
 * subset (NULL: the whole model set).  A set belongs to the device that was current when it was created. */
typedef struct pcreg_desc_set pcreg_desc_set;
int pcreg_desc_set_create(const double* desc, int n, int ld, int D, pcreg_desc_set** set);
int pcreg_desc_set_destroy(pcreg_desc_set* set);
int pcreg_desc_set_size(const pcreg_desc_set* set, int* n, int* D);
int pcreg_get_matches_on_sets(const pcreg_desc_set* surface, const pcreg_desc_set* model, const int32_t* model_rows, int n_rows,
                              const pcreg_match_opts* par, uint32_t* pairs, double* metric, int* P);
/* pcreg_get_matches_segmented on uploaded sets: ALL spheres of the loop above in one call, nothing but the row lists going up and
 * the pairs coming down (the 470 MB of a 60 000 x 980 model set take longer to upload than the whole sweep takes to match).
 * Arguments and results as pcreg_get_matches_segmented.  From their first segmented call on the sets keep a row-major copy, and
 * the model set its powered rows for the options of the last call (so a set then holds up to three times its n x D doubles). */
int pcreg_get_matches_segmented_on_sets(const pcreg_desc_set* surface, const pcreg_desc_set* model, const int32_t* seg_rows,
                                        const int32_t* seg_off, int S, const pcreg_match_opts* par, uint32_t* pairs_all, int32_t* n_pairs);

/* The sphere sweep of completeExperimentFast.m:46-224 at the host tier, in two calls (the first fixes the sizes of the second).
 *
 * pcreg_sphere_counts: counts[i] = #{ rows of featModel with norm(row - centres(i, :)) < R }  (:52-64; featModel VM x 3, centres
 * S x 3, both column-major doubles).  The caller keeps the spheres with min_pts <= counts <= max_pts (:61-64).
 *
 * pcreg_sphere_sweep, for the S spheres kept (centres S x 3, num_desc[i] = their counts): per sphere
 *     mask = getDescriptorMask(featModel, c_i, R_desc);  matches = getMatches(descSurface, descModel(mask, :), par)      (:109-149)
 * on resident descriptor sets (pcreg_desc_set_create), then for every sphere with more than putative_thresh matches (:166-184)
 *     [T, ~, numSuccess, maxInliers] = ransac(featSurface(matches(:,1), :), featCur(matches(:,2), :), coef, ...)          (:201-216)
 * with the built-in sampler seeded coef->seed + t for the t-th such sphere -- one launch chain, two synchronisations.
 * Outputs: model_rows: the spheres' 0-based ascending row lists back to back (sum of num_desc entries); pairs_all [S][Q][2]
 * (Q = rows of the surface set) and n_pairs[S] as pcreg_get_matches_segmented; trial[t], t < *n_trials: the 0-based spheres
 * that were registered, in ascending order; T (16 per trial, column-major 4 x 4, zeros where failed[t]), num_success, max_inliers,
 * failed: capacity S each.  Metric SAD; coef->minPtNum = 3. */
int pcreg_sphere_counts(const double* featModel, int VM, int ldM, const double* centres, int S, int ldC, double R, int32_t* counts);
int pcreg_sphere_sweep(const pcreg_desc_set* surface, const pcreg_desc_set* model, const double* featSurface, int ldS, const double* featModel, int ldM,
                       const double* centres, int S, int ldC, const int32_t* num_desc, double R_desc, const pcreg_match_opts* par, int putative_thresh,
                       const pcreg_ransac_opts* coef, int32_t* model_rows, uint32_t* pairs_all, int32_t* n_pairs, int32_t* trial, int* n_trials,
                       double* T, int32_t* num_success, int32_t* max_inliers, int32_t* failed);

/* The same with ONE model and many surfaces (completeExperimentFast.m runs once per surface crop): what the sweep makes of the model
 * alone -- the spheres' row lists and gathered keypoints, the model set restricted to the union of those rows, its powered rows per
 * set of getMatches options -- lives in a handle.  pcreg_sphere_model_create takes pcreg_sphere_sweep's model-side arguments and
 * returns the row lists (model_rows: sum of num_desc entries, 0-based); pcreg_sphere_sweep_on_model takes the surface-side ones and
 * returns pcreg_sphere_sweep's other outputs (the same values).  The handle holds copies: the model set may be destroyed first. */
typedef struct pcreg_sphere_model pcreg_sphere_model;
int pcreg_sphere_model_create(const pcreg_desc_set* model, const double* featModel, int ldM, const double* centres, int S, int ldC,
                              const int32_t* num_desc, double R_desc, int32_t* model_rows, pcreg_sphere_model** out);
int pcreg_sphere_model_destroy(pcreg_sphere_model* m);
int pcreg_sphere_sweep_on_model(pcreg_sphere_model* m, const pcreg_desc_set* surface, const double* featSurface, int ldS, const pcreg_match_opts* par,
                                int putative_thresh, const pcreg_ransac_opts* coef, uint32_t* pairs_all, int32_t* n_pairs, int32_t* trial, int* n_trials,
                                double* T, int32_t* num_success, int32_t* max_inliers, int32_t* failed);

/* AlignPoints_KNN.m:1  [pts_aligned, coeff_unambig, c] = AlignPoints_KNN(pts, C1, C2).
 * aligned: n x 3 (ld n); coeff: column-major 3x3; c: 3. */
int pcreg_align_points_knn(const double* pts, int n, int ld, int C1, int C2,
                           double* aligned, double coeff[9], double c[3]);

/* B supports in one launch (the per-keypoint loop of
 * getSpacialHistogramDescriptors.m:64-145 calls the same LRF per support):
 * support b owns rows [offsets[b], offsets[b+1]) of pts / aligned; coeff 9*B, c 3*B;
 * status[b] = 0 ok, 1 = support too small (n < 2).  A support may hold at most 8192 points (it is kept in the
 * registers of one workgroup); larger -> PCREG_E_ARG.  The reference's supports hold 500 ... 6000
 * (completeExperimentFast.m:300-302). */
int pcreg_align_points_knn_batched(const double* pts, int total, int ld, const int32_t* offsets,
                                   int B, int C1, int C2, double* aligned, double* coeff,
                                   double* c, int32_t* status);

/* The same for a `single` cloud: the outputs keep the class (AlignPoints_KNN.m:17,59: everything derives from pts).  Inputs
 * are widened to double exactly, the double kernel runs, outputs are rounded once to float.  MATLAB's own single
 * arithmetic may decide a borderline K-th-nearest / sign vote differently: documented in INTEGRATION.md, not reproduced. */
int pcreg_align_points_knn_f32(const float* pts, int n, int ld, int C1, int C2,
                               float* aligned, float coeff[9], float c[3]);

/* `options` of getSpacialHistogramDescriptors.m:18-27 (completeExperimentFast.m:299-304). */
typedef struct pcreg_desc_opts {
    int32_t min_pts;       /* options.min_pts                                            */
    int32_t max_pts;       /* options.max_pts (INT32_MAX for inf)                        */
    double  R;             /* options.R: support radius                                  */
    double  thVar[2];      /* options.thVar: eigenvalue-ratio rejection thresholds       */
    double  k;             /* options.k: fraction of nearest points used for the LRF;
                              >= 1 means 'all'                                           */
    int32_t ALIGN_POINTS;  /* options.ALIGN_POINTS                                       */
} pcreg_desc_opts;
#define PCREG_DESC_LEN 980   /* NUM_R*NUM_THETA*NUM_PHI = 10*7*14, getSpacialHistogramDescriptors.m:38-40 */

/* getSpacialHistogramDescriptors.m:2  [feat, desc] = getSpacialHistogramDescriptors(pts,
 * sample_pts, options) (with getLocalPoints.m and histcn.m folded in).  pts: P x 3,
 * sample_pts: S x 3 (column-major).  Outputs are ROW-major with capacity S rows:
 * feat [V][3], desc [V][980] (counts, r fastest / then theta / then phi, i.e. MATLAB's
 * reshape(counts,[],1)); *V = number of surviving keypoints, in input order.
 * The O(S*P) brute-force radius search of the reference is replaced by a uniform grid.
 * Limits: a support of more than 8191 points (possible only with options.max_pts > 8190;
 * it lives in LDS) and a cloud of 2^28 points or more (the kernel addresses the sorted
 * cloud with 32-bit byte offsets) are refused with PCREG_E_ARG. */
int pcreg_spatial_histogram_descriptors(const double* pts, int P, int ld, const double* sample_pts, int S, int lds,
                                        const pcreg_desc_opts* options, double* feat, double* desc, int* V);

/* The same for `single` data (clouds from pcread are single: upsampleMesh.m:21, GetPointcloudFromModel.m:269, fed on by
 * completeExperimentFast.m:291,309).  feat / desc are DOUBLE whatever the input classes: the reference preallocates them
 * with nan(...) and assigns into them (getSpacialHistogramDescriptors.m:61-62).  When either input is single, MATLAB runs
 * getLocalPoints.m:8-31 in single; what is element-wise there is reproduced in fp32 exactly as written -- the open box test,
 * pts_cube - c, sqrt(x^2 + y^2 + z^2), dists < R -- i.e. WHICH keypoints survive and which points form a support, and the
 * support's coordinates are MATLAB's single pts_rel values.  mean / pca / the histogram run in double on those values
 * (INTEGRATION.md says what is not knowable).  _mixed takes each input in its own class: *_is_single != 0 -> const float*,
 * else const double*. */
int pcreg_spatial_histogram_descriptors_f32(const float* pts, int P, int ld, const float* sample_pts, int S, int lds,
                                            const pcreg_desc_opts* options, double* feat, double* desc, int* V);
int pcreg_spatial_histogram_descriptors_mixed(const void* pts, int pts_is_single, int P, int ld, const void* sample_pts,
                                              int sample_is_single, int S, int lds, const pcreg_desc_opts* options,
                                              double* feat, double* desc, int* V);

/* ---- device tier ------------------------------------------------------------------
 * All pointers are device memory on the current device; `stream` is a hipStream_t.
 * Workspaces are caller-owned device buffers; query the size first. */

/* Two nearest model points per query for one model shard.  idx_base is added to the
 * reported indices (global row of the shard's first point).  idx [Q][2] int32, dist
 * [Q][2] float. */
size_t pcreg_dev_knn2_points_f32_workspace(int Q, int M);
int pcreg_dev_knn2_points_f32(const float* q, int Q, int ldq, const float* m, int M, int ldm,
                              int32_t idx_base, int32_t* idx, float* dist,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- a PREPARED MODEL: one model, many surfaces (completeExperimentFast.m:131-149 matches every candidate sphere,
 * :201-216 registers every promising one, against the same model; BASELINE cfg 5: 64 crops vs one CT model) ----------
 * Everything that depends on the model alone -- its bounding box, the tiles of matrix-core operands, the model-wide
 * seeding grid -- is computed once; a search against the handle is four launches, the match stage one.
 * pcreg_dev_model_create enqueues the preparation on `stream` and returns at once; the model's points (`m`, n x 3
 * column-major, ldm) are NOT copied and must stay valid and unchanged while the handle lives.  A handle may be used
 * from several streams at once as long as each call has its own workspace.  Reported rows are idx_base + local row. */
typedef struct pcreg_dev_model pcreg_dev_model;
int pcreg_dev_model_create(const float* m, int M, int ldm, void* stream, pcreg_dev_model** model);
int pcreg_dev_model_destroy(pcreg_dev_model* model);
/* Top-2 of every query over the prepared model (pcreg_dev_knn2_points_f32's contract and bits).  The workspace also
 * receives a uniform grid over the queries, which pcreg_dev_model_match_f32 / _match_table_f32 need for Unique: pass
 * the SAME workspace to them, with no other search on it in between. */
size_t pcreg_dev_model_search_workspace(int Q, int M);
int pcreg_dev_model_search_f32(const pcreg_dev_model* model, const float* q, int Q, int ldq, int32_t idx_base,
                               int32_t* idx, float* dist, void* workspace, size_t workspace_bytes, void* stream);
/* matchFeatures' filter chain on that top-2 in ONE launch: threshold, ratio test, Unique back-check, ordered compaction
 * into 1-based pairs [k][2] and the matched coordinates pts1 / pts2 (n x 3 column-major doubles, ld = Q; both NULL to
 * skip) -- completeExperimentFast.m:205-206.  The handle holds the WHOLE model (one rank). */
int pcreg_dev_model_match_f32(const pcreg_dev_model* model, const float* q, int Q, int ldq, const int32_t* idx,
                              const float* dist, float thr_abs, float max_ratio, int unique, void* workspace,
                              size_t workspace_bytes, uint32_t* pairs, double* pts1, double* pts2, int32_t* n_pairs,
                              void* stream);
/* The same split around the multi-GPU exchange (idx / dist are the MERGED global top-2):
 * _match_table_f32: this rank's contribution to a [4][Q] table of 4-byte words, column = query: rows 0-2 the coordinate
 *   bits of the query's nearest model point, row 3 its Unique verdict (1 when Unique is off), written by the rank whose
 *   shard [m_lo, m_lo + M) holds that point and only for queries that pass the filters; zero elsewhere.  An integer
 *   all_reduce(SUM) over the ranks assembles the table exactly (one contributor per column).
 * pcreg_dev_match_from_table_f32: filters again (deterministic), verdicts and coordinates from the summed table, ordered
 *   compaction.  `workspace`: any search workspace of this Q (only its counters are used). */
int pcreg_dev_model_match_table_f32(const pcreg_dev_model* model, int32_t m_lo, int M_total, const float* q, int Q, int ldq,
                                    const int32_t* idx, const float* dist, float thr_abs, float max_ratio, int unique,
                                    void* workspace, size_t workspace_bytes, int32_t* table, void* stream);
int pcreg_dev_match_from_table_f32(const float* q, int Q, int ldq, int M_total, const int32_t* idx, const float* dist,
                                   float thr_abs, float max_ratio, const int32_t* table, void* workspace,
                                   size_t workspace_bytes, uint32_t* pairs, double* pts1, double* pts2, int32_t* n_pairs,
                                   void* stream);

/* Merge R candidate lists (e.g. the all-gathered per-shard results, laid out
 * [R][Q][2]) into one top-2 per query, ordering by (dist, idx). */
int pcreg_dev_merge_top2_f32(const int32_t* idx_in, const float* dist_in, int R, int Q,
                             int32_t* idx, float* dist, void* stream);
/* The same with list r starting r * rank_stride ELEMENTS into idx_in / dist_in (>= 2 Q): lets one all_gather
 * carry a rank's indices and distances in a single buffer. */
int pcreg_dev_merge_top2_strided_f32(const int32_t* idx_in, const float* dist_in, int R, int Q, size_t rank_stride,
                                     int32_t* idx, float* dist, void* stream);

/* Device-resident ransac: n is read from device memory (*n_dev <= n_cap), so the
 * match stage can feed it without a host round trip.  Results land in `out`
 * (pcreg_dev_ransac_result) and inlier_idx (capacity n_cap). */
typedef struct pcreg_dev_ransac_result {
    double  T[16];
    int32_t n_inliers, num_success, max_inliers, failed, n, winner;
} pcreg_dev_ransac_result;
size_t pcreg_dev_ransac_workspace(int n_cap, int iterNum);
int pcreg_dev_ransac(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                     const pcreg_ransac_opts* opts, const int32_t* sample_idx,
                     pcreg_dev_ransac_result* out, int32_t* inlier_idx,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid: with timing enabled, the candidates kernel of every point search the CALLER asks for (any size) is
 * bracketed by two HIP events on its launch stream; pcreg_dev_search_kernel_ms returns the mean
 * duration over the launches since the previous call (it waits for them) and their number. */
int pcreg_dev_search_kernel_timing(int enable);
int pcreg_dev_search_kernel_ms(float* mean_ms, int* launches);

/* One registration's hypotheses split over ranks (optional mode): every rank runs the
 * hypotheses [hyp_begin, hyp_begin + hyp_count) of a job of opts->iterNum -- the built-in sampler and the
 * first-maximum tie-break (ransac.m:70-72) use the GLOBAL hypothesis index, so the union over ranks is
 * exactly the single-rank run -- and leaves its share's best in `part` (device).  Combine on the ranks:
 * key by MAX, num_success by SUM, has / T from the rank whose key equals the maximum; then
 * pcreg_dev_ransac_finish builds the result and the inlier list from the combined part.  sample_idx,
 * if given, holds the hyp_count rows of THIS share. */
typedef struct pcreg_dev_ransac_part {
    unsigned long long key;          /* (inlier count << 32) | ~global hypothesis index; 0 for an empty share */
    int32_t num_success, has;        /* hypotheses of the share with count >= thInlr; 1 if the winner holds a transform */
    double  T[12];                   /* the winner's transform, rows of [R t] (as stored internally) */
} pcreg_dev_ransac_part;
int pcreg_dev_ransac_partial(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                             const pcreg_ransac_opts* opts, const int32_t* sample_idx, int hyp_begin, int hyp_count,
                             pcreg_dev_ransac_part* part, void* workspace, size_t workspace_bytes, void* stream);
int pcreg_dev_ransac_finish(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                            const pcreg_ransac_opts* opts, const pcreg_dev_ransac_part* combined,
                            pcreg_dev_ransac_result* out, int32_t* inlier_idx, void* stream);
/* The same from the n_parts UNCOMBINED parts of all shares (e.g. one all_gather of the 112-byte structs): the
 * kernel takes the maximum key, sums num_success and uses the winner's transform itself. */
int pcreg_dev_ransac_finish_parts(const double* pts1, const double* pts2, const int32_t* n_dev, int n_cap, int ld,
                                  const pcreg_ransac_opts* opts, const pcreg_dev_ransac_part* parts, int n_parts,
                                  pcreg_dev_ransac_result* out, int32_t* inlier_idx, void* stream);

/* ---- descriptor stage, resident: speedyDescriptors.m:59 -> getMatches.m -> ransac.m --------
 * (completeExperimentFast.m:131-213 per sphere position) without a host copy in between. */

/* getSpacialHistogramDescriptors.m:25-258 on device buffers.  pts / sample_pts are n x 3
 * column-major (ld / lds); feat [S][3] and desc [S][980] are ROW-major with capacity S
 * (row v < V is the v-th surviving keypoint, in sample order); counters[0] = V,
 * counters[1] = 0 or the size of a support that exceeded the LDS capacity (then the
 * outputs are invalid: lower max_pts). */
size_t pcreg_dev_spatial_histogram_descriptors_workspace(int P, int S);
int pcreg_dev_spatial_histogram_descriptors(const double* pts, int P, int ld, const double* sample_pts, int S, int lds,
                                            const pcreg_desc_opts* options, double* feat, double* desc,
                                            int32_t* counters, void* workspace, size_t workspace_bytes, void* stream);
/* What a resident pipeline keeps: the 980 counts of a row as uint16 (integer counts <= the support size <= 8191: 1.96 KB
 * per keypoint instead of 7.84 KB), written ONCE, straight from the histogram in LDS: rows [S][980] in KEYPOINT order (row s is
 * meaningful only for a surviving keypoint s), row_index [S] = the ascending list of the V survivors, feat [V][3] compact.
 * pcreg_dev_get_matches_rows_u16 takes rows + index as they are.  single_mode: 0 double data; 1 `single` arithmetic for the
 * support (see pcreg_spatial_histogram_descriptors_f32; data widened exactly to double by the caller), keypoints single;
 * 2 the same with a single cloud and double keypoints.  Same workspace. */
int pcreg_dev_spatial_histogram_descriptors_rows_u16(const double* pts, int P, int ld, const double* sample_pts, int S, int lds,
                                                     const pcreg_desc_opts* options, int single_mode, double* feat, uint16_t* rows,
                                                     int32_t* row_index, int32_t* counters, void* workspace, size_t workspace_bytes,
                                                     void* stream);

/* getMatches.m:21-59 on device buffers.  layout: PCREG_LAYOUT_FEATURE_MAJOR = MATLAB's
 * column-major n x D (ld >= n); PCREG_LAYOUT_ROW_MAJOR = dense [n][D] (ld == D), what the
 * descriptor entry point above emits.  The inputs are not modified.  pairs [Q][2] uint32,
 * 1-based, ascending surface row; metric [Q] or NULL; *n_pairs device int32.  Q and M are
 * host integers (grid sizes); with Unique and Q*Q*D > 2e10 the call synchronises the stream ONCE (the
 * candidate count that sizes the back-search); otherwise it does not synchronise. */
#define PCREG_LAYOUT_FEATURE_MAJOR 0
#define PCREG_LAYOUT_ROW_MAJOR     1
size_t pcreg_dev_get_matches_workspace(int Q, int M, int D);
int pcreg_dev_get_matches(const double* descSurface, int Q, int ldS, const double* descModel, int M, int ldM, int D,
                          int layout, const pcreg_match_opts* par, uint32_t* pairs, double* metric,
                          int32_t* n_pairs, void* workspace, size_t workspace_bytes, void* stream);
/* The same on uint16 rows (pcreg_dev_spatial_histogram_descriptors_rows_u16's output): row i of a descriptor set is
 * rows[index[i]] (index NULL: rows[i]); the counts are widened to double, exactly, on the way into getMatches.m:24-37's
 * private copies -- pairs (numbered by i) and metric are those of the double entry.  Same workspace. */
int pcreg_dev_get_matches_rows_u16(const uint16_t* rowsSurface, const int32_t* indexSurface, int Q, const uint16_t* rowsModel,
                                   const int32_t* indexModel, int M, int D, const pcreg_match_opts* par, uint32_t* pairs,
                                   double* metric, int32_t* n_pairs, void* workspace, size_t workspace_bytes, void* stream);

/* completeExperimentFast.m:205-206: pts1 = featSurface(matches(:,1),:), pts2 =
 * featModel(matches(:,2),:) as n x 3 column-major with ld = cap -- the input of
 * pcreg_dev_ransac (n = *n_pairs stays on the device).  feat* are row-major [.][3]. */
int pcreg_dev_gather_matched_rows(const uint32_t* pairs, const int32_t* n_pairs, int cap, const double* featSurface,
                                  const double* featModel, double* pts1, double* pts2, void* stream);

/* ---- sphere-sweep driver pieces (completeExperimentFast.m:46-225) and its final stage ----
 * feat is row-major [V][3] (what the descriptor entry point emits). */

/* counts[s] = #{ i : vecnorm(feat(i,:) - centres(s,:)) < R }: the sphere validity test of
 * completeExperimentFast.m:57-64 (getLocalPoints.m:23-34 reduced to its count). centres [S][3]. */
int pcreg_dev_sphere_counts(const double* feat, int V, const double* centres, int S, double R,
                            int32_t* counts, void* stream);

/* getDescriptorMask (completeExperimentFast.m:435-439, margin folded into R) as the ascending
 * 0-based index list of the rows inside the sphere; *n_out = its length.  `centre` is HOST memory. */
size_t pcreg_dev_sphere_select_workspace(int V);
int pcreg_dev_sphere_select(const double* feat, int V, const double centre[3], double R, int32_t* idx,
                            int32_t* n_out, void* workspace, size_t workspace_bytes, void* stream);

/* getDescriptorMask for S spheres in one launch (completeExperimentFast.m:109-125 for every sphere of the sweep): the row
 * list of sphere s is written at idx[seg_off[s] ..], ascending and 0-based; seg_off [S + 1] (device) are the running sums of
 * pcreg_dev_sphere_counts' counts.  feat_out (or NULL): featCur of every sphere back to back ([seg_off[S]][3] row-major);
 * n_out (or NULL) [S]: the lengths found (== the counts).  centres [S][3] on the device. */
int pcreg_dev_sphere_select_batched(const double* feat, int V, const double* centres, int S, double R, const int32_t* seg_off,
                                    int32_t* idx, double* feat_out, int32_t* n_out, void* stream);

/* getMatches.m:21-59 for S segments in ONE chain of launches: getMatches(descSurface, descModel(rows_s, :), par) for every
 * segment s, rows_s = seg_rows[seg_off[s] .. seg_off[s+1]) (0-based, ascending) -- the per-sphere calls of the sweep,
 * completeExperimentFast.m:131-149, which the reference runs under parfor.  descSurface [Q][D], descModel [VM][D] dense
 * row-major doubles (what the descriptor entry point emits).  total_rows = seg_off[S] and max_rows = the longest segment
 * are known to the host from the counts.  pairs_all [S][Q][2]: segment s's pairs, 1-based, the model index counting WITHIN
 * the segment like the per-sphere call's; n_pairs [S]; metric_all [S][Q] or NULL.  The pairs are those of one
 * pcreg_dev_get_matches call per segment (same arithmetic on the same operands: the appended constant and the row norms
 * differ per segment, the powered columns do not and are computed once).  Metric SAD only.
 * Memory: the one-chain form takes  4 VM' Q'  (the shared score matrix, VM' / Q' = VM / Q rounded up to 128)  +  8 D (VM + Q)  (the
 * powered rows)  +  ~S Q (32 splits + 500) bytes (the per-segment lists and re-rank work items);  up to 4 GB of that it runs as ONE chain and nothing
 * synchronises.  Above, the call runs in batches of consecutive segments on gathered sub-models (the union of the rows a batch
 * names), inside a workspace of 4 GB + 12 VM + 4 total_rows + 4 S bytes: same pairs, one host synchronisation at entry and one or
 * two per batch.  A single segment that does not fit the bound is refused with PCREG_E_WORKSPACE. */
size_t pcreg_dev_get_matches_segmented_workspace(int Q, int VM, int D, int S, int total_rows, int max_rows);
int pcreg_dev_get_matches_segmented(const double* descSurface, int Q, const double* descModel, int VM, int D, const int32_t* seg_rows,
                                    const int32_t* seg_off, int S, int total_rows, int max_rows, const pcreg_match_opts* par,
                                    uint32_t* pairs_all, double* metric_all, int32_t* n_pairs, void* workspace, size_t workspace_bytes,
                                    void* stream);
/* One model, many surfaces: the part of the call above that depends on the model set and the options only -- its powered rows and
 * six scalars per row -- prepared once into a caller-owned buffer of pcreg_dev_segmented_model_bytes(VM, D) bytes, and the call that
 * uses it (prepared_change_metric / prepared_metric_factor: the options it was made with; a call whose par differs recomputes them
 * itself).  The caller must not change descModel while the preparation is in use.  Same pairs as pcreg_dev_get_matches_segmented. */
size_t pcreg_dev_segmented_model_bytes(int VM, int D);
int pcreg_dev_segmented_model_prepare(const double* descModel, int VM, int D, const pcreg_match_opts* par, void* prepared, size_t prepared_bytes,
                                      void* stream);
int pcreg_dev_get_matches_segmented_prepared(const double* descSurface, int Q, const double* descModel, int VM, int D, const void* prepared,
                                             int prepared_change_metric, double prepared_metric_factor, const int32_t* seg_rows,
                                             const int32_t* seg_off, int S, int total_rows, int max_rows, const pcreg_match_opts* par,
                                             uint32_t* pairs_all, double* metric_all, int32_t* n_pairs, void* workspace, size_t workspace_bytes,
                                             void* stream);

/* dst(k,:) = src(idx(k),:), k < min(*n, cap): featCur / descCur of completeExperimentFast.m:122-125
 * (row-major, D doubles per row). */
int pcreg_dev_gather_rows_f64(const double* src, int D, const int32_t* idx, const int32_t* n, int cap,
                              double* dst, void* stream);

/* The batched sphere sweep (completeExperimentFast.m:166-224 for ALL spheres in one enqueue chain, no host round trip
 * between the per-sphere getMatches calls and the per-trial ransac calls -- the reference runs both under parfor).
 * pcreg_dev_sweep_plan:   trial = find(num_putative > putative_thresh) (:175) in sphere order -> trial_idx [S] (-1 past
 *                         n_trials), offsets [S+1] of each trial's pairs in the packed arrays, *n_trials.
 * pcreg_dev_sweep_gather: pts1 = featSurface(matches(:,1),:), pts2 = featuresM(matches(:,2),:) (:205-206) for every trial,
 *                         packed at offsets[t] (n x 3 column-major, leading dimension ld).  pairs_all [S][VS][2] are the
 *                         1-based pairs of pcreg_dev_get_matches per sphere, featCur_all the spheres' gathered keypoints
 *                         ([rows][3]) back to back, sphere i starting at row row_off[i].
 * pcreg_dev_ransac_batched: pcreg_ransac_batched on device buffers: registration b owns rows [offsets[b], offsets[b+1])
 *                         (offsets on the DEVICE: sizes need not be known on the host), built-in sampler seeded seed + b;
 *                         empty registrations report failed = 1.  out [B], inlier_idx has the rows' layout. */
int pcreg_dev_sweep_plan(const int32_t* n_pairs, int S, int putative_thresh, int32_t* trial_idx, int32_t* offsets,
                         int32_t* n_trials, void* stream);
int pcreg_dev_sweep_gather(const uint32_t* pairs_all, int VS, const int32_t* n_pairs, const int32_t* trial_idx,
                           const int32_t* offsets, const int32_t* n_trials, int S, const double* featSurface,
                           const double* featCur_all, const int64_t* row_off, double* pts1, double* pts2, int ld,
                           void* stream);
size_t pcreg_dev_ransac_batched_workspace(int n_cap, int iterNum, int B);
int pcreg_dev_ransac_batched(const double* pts1, const double* pts2, int ld, const int32_t* offsets, int B, int n_cap,
                             const pcreg_ransac_opts* opts, pcreg_dev_ransac_result* out, int32_t* inlier_idx,
                             void* workspace, size_t workspace_bytes, void* stream);

/* AlignPoints_KNN.m:17-59 for B supports resident in HBM (the device-tier form of
 * pcreg_align_points_knn_batched; same layouts, every pointer is device memory): support b owns rows
 * [offsets[b], offsets[b+1]) of pts / aligned (n x 3 column-major, ld >= total); max_n = the largest support;
 * coeff 9*B, c 3*B, status B (0 ok, 1 = fewer than 2 points).  Enqueued on `stream`, nothing synchronises. */
int pcreg_dev_align_points_knn_batched(const double* pts, int total, int ld, const int32_t* offsets, int B, int max_n,
                                       int C1, int C2, double* aligned, double* coeff, double* c, int32_t* status,
                                       void* stream);

/* quickTF.m:5-7: out = [pts, 1] * T (first three columns).  pts/out n x 3 column-major on the
 * device, T 4x4 column-major in HOST memory (invertTF is a 16-number host operation). */
int pcreg_dev_quick_tf(const double* pts, int n, int ld, const double T[16], double* out, int ldo, void* stream);

/* completeExperimentFast.m:368-391: inliers = vecnorm(pts1 - pts2) < maxDist, T_refine =
 * estimateTransform over them.  n = *n_dev <= cap; T16 (device, column-major 4x4, zeros when
 * empty), info[0] = number of inliers, info[1] = 1 if the transform is empty. */
int pcreg_dev_refine_by_distance(const double* pts1, const double* pts2, const int32_t* n_dev, int cap, int ld,
                                 double maxDist, double* T16, int32_t* info, void* stream);

/* ---- multi-GPU behind the C ABI (RCCL over xGMI inside the library; comm.hip) -----------------------------
 * One process per GPU -- a MATLAB parfor / spmd worker each, the reference's own unit of parallelism
 * (completeExperimentFast.m:134,201).  Worker 0 calls pcreg_comm_get_unique_id; the 128 bytes reach the other
 * workers by the host language's own means (labBroadcast, a file); every worker calls pcreg_set_device and then
 * pcreg_comm_init(rank, world, id).  The sharded calls are collective: every rank must make them, in the same
 * order, and every rank gets the same result -- bit for bit the single-GPU one.  RCCL is dlopen'ed at
 * pcreg_comm_init (PCREG_E_HIP with a message if librccl.so.1 is missing). */
typedef struct pcreg_comm_id { char bytes[128]; } pcreg_comm_id;      /* an ncclUniqueId */
int pcreg_comm_get_unique_id(pcreg_comm_id* id);
int pcreg_comm_init(int rank, int world, const pcreg_comm_id* id);
/* The same communicator over HOST-STAGED exchanges instead of RCCL: the ranks meet in the POSIX shared-memory segment
 * `name` ("/something", unique per group; every rank passes the same).  For workers that share one GPU (RCCL refuses two
 * ranks on a device) and for boxes without RCCL; the messages of this path are latency-sized (<= 16 Q bytes). */
int pcreg_comm_init_host_staged(int rank, int world, const char* name);
int pcreg_comm_rank(int* rank, int* world);
int pcreg_comm_destroy(void);

/* pcreg_match_points_f32 with the MODEL rows split over the ranks: this rank passes rows
 * [m_lo, m_lo + M_local) of the M_total-row model and the whole (replicated) surface.  One all_gather of the
 * per-rank top-2 lists + merge, filters on every rank, Unique by the owner of the model row, one integer
 * all_reduce of the candidate table.  pairs: 1-based [surface row, GLOBAL model row], capacity Q. */
int pcreg_match_points_sharded_f32(const float* q, int Q, int ldq, const float* m_local, int M_local, int ldm,
                                   int m_lo, int M_total, float thr_abs, float max_ratio, int unique,
                                   uint32_t* pairs, int* P);

/* pcreg_ransac (built-in sampler, minPtNum 3) with the hypotheses of ONE registration split over the ranks:
 * every rank passes the same pts1 / pts2, scores iterNum / world hypotheses, one all_gather of the 112-byte
 * partial results agrees the first-maximum winner (ransac.m:70-72 on the GLOBAL hypothesis index). */
int pcreg_ransac_sharded(const double* pts1, const double* pts2, int n, int ld, const pcreg_ransac_opts* opts,
                         double T[16], int32_t* inlier_idx, int* n_inliers, int* num_success, int* max_inliers,
                         int* failed);

/* ---- on-disk formats of the drivers (host code, no device needed) --------------------------
 * .pcd clouds (pcread / pcwrite, completeExperimentFast.m:12-13,30,403): ascii, binary and
 * binary_compressed (LZF) files are read; x/y/z may be float or double, rgb/rgba is returned as
 * the packed 0x00RRGGBB word.  xyz is n x 3 column-major (ld >= n), like pointCloud.Location. */
int pcreg_pcd_info(const char* path, int* n_points, int* has_rgb);
int pcreg_pcd_read(const char* path, float* xyz, int ld, uint32_t* rgb /* n or NULL */, int n);
int pcreg_pcd_write(const char* path, const float* xyz, int n, int ld, const uint32_t* rgb /* or NULL */,
                    int binary);

/* .mat descriptor caches (load, completeExperimentFast.m:21-24,312-313): one real numeric
 * variable of a Level-5 MAT-file (save -v6 / -v7, zlib-compressed elements included; v7.3 = HDF5
 * is not supported) as column-major doubles.  name NULL or "" = the first numeric array.  Call
 * with out == NULL for the shape (dimensions beyond the second are folded into cols). */
int pcreg_mat_read_double(const char* path, const char* name, double* out, int* rows, int* cols);

#ifdef __cplusplus
}
#endif
#endif /* PCREG_H */
