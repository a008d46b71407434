// tests/mexstub/mex_driver.cpp -- TEST INFRASTRUCTURE.  Plays MATLAB for mex/pcreg_mex.cpp: builds the
// mxArrays the matlab/*.m wrappers would pass (column-major doubles, the coef / par / options structs,
// the int32 sample table), calls mexFunction, and hands the outputs back through a plain C interface
// that tests/test_mex_shim.py drives with ctypes.  Returns 0, or 1 with the raised id:message in err.
#include "mex.h"
#include <algorithm>

int g_mex_live_arrays = 0;

static mxArray* dmat(const double* p, size_t m, size_t n) { mxArray* a = mxCreateDoubleMatrix(m, n, mxREAL); if (m * n > 0) memcpy(mxGetPr(a), p, m * n * 8); return a; }
static void put(mxArray* s, const char* k, double v) { mxSetField(s, 0, k, mxCreateDoubleScalar(v)); }

static int call(int nlhs, mxArray** plhs, std::vector<mxArray*>& rhs, char* err, int errlen) {
    int rc = 0;
    try { mexFunction(nlhs, plhs, (int)rhs.size(), const_cast<const mxArray**>(rhs.data())); }
    catch (const MexError& e) { snprintf(err, errlen, "%s: %s", e.id.c_str(), e.msg.c_str()); rc = 1; }
    for (mxArray* a : rhs) mxDestroyArray(a);
    return rc;
}

extern "C" {

int drv_live_arrays() { return g_mex_live_arrays; }

int drv_estimate_transform(const double* p1, const double* p2, int n, double* T16, int* empty, char* err, int errlen) {
    std::vector<mxArray*> rhs{mxCreateString("estimateTransform"), dmat(p1, n, 3), dmat(p2, n, 3)};
    mxArray* lhs[1] = {nullptr};
    if (call(1, lhs, rhs, err, errlen)) return 1;
    *empty = mxIsEmpty(lhs[0]);
    if (!*empty) memcpy(T16, mxGetPr(lhs[0]), 128);
    mxDestroyArray(lhs[0]);
    return 0;
}

int drv_calc_dists(const double* T16, const double* p1, const double* p2, int n, double* d, char* err, int errlen) {
    std::vector<mxArray*> rhs{mxCreateString("calcDists"), dmat(T16, 4, 4), dmat(p1, n, 3), dmat(p2, n, 3)};
    mxArray* lhs[1] = {nullptr};
    if (call(1, lhs, rhs, err, errlen)) return 1;
    memcpy(d, mxGetPr(lhs[0]), (size_t)n * 8);
    mxDestroyArray(lhs[0]);
    return 0;
}

// coef = {minPtNum, iterNum, thDist, thInlrRatio, REFINE}; sample: minPtNum x iterNum int32 (column per hypothesis) or NULL
int drv_ransac(const double* p1, const double* p2, int n, const double* coef5, const int32_t* sample, double seed,
               double* T16, double* inlier_idx, int* n_inl, int* num_success, int* max_inl, int* failed, char* err, int errlen) {
    mxArray* c = mxCreateStructMatrix(1, 1, 0, nullptr);
    put(c, "minPtNum", coef5[0]); put(c, "iterNum", coef5[1]); put(c, "thDist", coef5[2]); put(c, "thInlrRatio", coef5[3]);
    put(c, "REFINE", coef5[4]); put(c, "VERBOSE", 0);
    mxArray* si;
    if (sample) { si = mxCreateNumericMatrix((size_t)coef5[0], (size_t)coef5[1], mxINT32_CLASS, mxREAL); memcpy(mxGetData(si), sample, (size_t)coef5[0] * (size_t)coef5[1] * 4); }
    else si = mxCreateDoubleMatrix(0, 0, mxREAL);
    std::vector<mxArray*> rhs{mxCreateString("ransac"), dmat(p1, n, 3), dmat(p2, n, 3), c, si, mxCreateDoubleScalar(seed)};
    mxArray* lhs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (call(5, lhs, rhs, err, errlen)) return 1;
    *failed = (int)mxGetScalar(lhs[4]);
    if (!*failed) memcpy(T16, mxGetPr(lhs[0]), 128);
    *n_inl = (int)(mxGetM(lhs[1]) * mxGetN(lhs[1]));
    if (*n_inl) memcpy(inlier_idx, mxGetPr(lhs[1]), (size_t)*n_inl * 8);
    *num_success = (int)mxGetScalar(lhs[2]); *max_inl = (int)mxGetScalar(lhs[3]);
    for (mxArray* a : lhs) mxDestroyArray(a);
    return 0;
}

// B registrations through ONE 'ransacBatched' call: p1 / p2 total x 3 column-major, off B + 1 offsets, sample (or null): 3 x (iterNum B)
int drv_ransac_batched(const double* p1, const double* p2, int total, const int32_t* off, int B, const double* coef5, const int32_t* sample, double seed,
                       double* T16 /* B x 16 */, double* inlier_idx /* cap total */, int* n_inl_total, double* n_inl, double* num_success, double* max_inl,
                       double* failed, char* err, int errlen) {
    mxArray* c = mxCreateStructMatrix(1, 1, 0, nullptr);
    put(c, "minPtNum", coef5[0]); put(c, "iterNum", coef5[1]); put(c, "thDist", coef5[2]); put(c, "thInlrRatio", coef5[3]);
    put(c, "REFINE", coef5[4]); put(c, "VERBOSE", 0);
    mxArray* o = mxCreateNumericMatrix(B + 1, 1, mxINT32_CLASS, mxREAL);
    memcpy(mxGetData(o), off, (size_t)(B + 1) * 4);
    mxArray* si;
    if (sample) { const size_t ne = (size_t)coef5[0] * (size_t)coef5[1] * (size_t)B; si = mxCreateNumericMatrix((size_t)coef5[0], ne / (size_t)coef5[0], mxINT32_CLASS, mxREAL); memcpy(mxGetData(si), sample, ne * 4); }
    else si = mxCreateDoubleMatrix(0, 0, mxREAL);
    std::vector<mxArray*> rhs{mxCreateString("ransacBatched"), dmat(p1, total, 3), dmat(p2, total, 3), o, c, si, mxCreateDoubleScalar(seed)};
    mxArray* lhs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (call(6, lhs, rhs, err, errlen)) return 1;
    if (mxGetM(lhs[0]) != 4 || mxGetN(lhs[0]) != (size_t)4 * B) { snprintf(err, errlen, "T is not 4 x 4 x B"); return 1; }
    memcpy(T16, mxGetPr(lhs[0]), (size_t)B * 128);
    *n_inl_total = (int)(mxGetM(lhs[1]) * mxGetN(lhs[1]));
    if (*n_inl_total) memcpy(inlier_idx, mxGetPr(lhs[1]), (size_t)*n_inl_total * 8);
    memcpy(n_inl, mxGetPr(lhs[2]), (size_t)B * 8); memcpy(num_success, mxGetPr(lhs[3]), (size_t)B * 8);
    memcpy(max_inl, mxGetPr(lhs[4]), (size_t)B * 8); memcpy(failed, mxGetPr(lhs[5]), (size_t)B * 8);
    for (mxArray* a : lhs) mxDestroyArray(a);
    return 0;
}

// par = {MatchThreshold, MaxRatio, Unique, UNNORMALIZE, norm_factor, CHANGE_METRIC, metric_factor}; metric "SAD" | "SSD"
int drv_get_matches(const double* dS, int Q, const double* dM, int M, int D, const char* metric, const double* par7,
                    uint32_t* pairs_colmajor /* cap Q x 2 */, int* P, char* err, int errlen) {
    mxArray* p = mxCreateStructMatrix(1, 1, 0, nullptr);
    mxSetField(p, 0, "Metric", mxCreateString(metric)); mxSetField(p, 0, "Method", mxCreateString("Approximate"));
    put(p, "MatchThreshold", par7[0]); put(p, "MaxRatio", par7[1]); put(p, "Unique", par7[2]); put(p, "UNNORMALIZE", par7[3]);
    put(p, "norm_factor", par7[4]); put(p, "CHANGE_METRIC", par7[5]); put(p, "metric_factor", par7[6]); put(p, "VERBOSE", 0);
    std::vector<mxArray*> rhs{mxCreateString("getMatches"), dmat(dS, Q, D), dmat(dM, M, D), p};
    mxArray* lhs[1] = {nullptr};
    if (call(1, lhs, rhs, err, errlen)) return 1;
    *P = (int)mxGetM(lhs[0]);
    if (mxGetN(lhs[0]) != 2 && *P) { snprintf(err, errlen, "getMatches returned %zu columns", mxGetN(lhs[0])); mxDestroyArray(lhs[0]); return 1; }
    memcpy(pairs_colmajor, mxGetData(lhs[0]), (size_t)*P * 2 * 4);
    mxDestroyArray(lhs[0]);
    return 0;
}

// rows1: 1-based rows of all segments back to back; off: S + 1 offsets; pairs_colmajor cap S * Q x 2; n_pairs S
int drv_get_matches_segmented(const double* dS, int Q, const double* dM, int VM, int D, const double* par7, const int32_t* rows1, int tot,
                              const int32_t* off, int S, uint32_t* pairs_colmajor, int* P_total, int32_t* n_pairs, char* err, int errlen) {
    mxArray* p = mxCreateStructMatrix(1, 1, 0, nullptr);
    mxSetField(p, 0, "Metric", mxCreateString("SAD")); mxSetField(p, 0, "Method", mxCreateString("Approximate"));
    put(p, "MatchThreshold", par7[0]); put(p, "MaxRatio", par7[1]); put(p, "Unique", par7[2]); put(p, "UNNORMALIZE", par7[3]);
    put(p, "norm_factor", par7[4]); put(p, "CHANGE_METRIC", par7[5]); put(p, "metric_factor", par7[6]); put(p, "VERBOSE", 0);
    mxArray* r = mxCreateNumericMatrix(tot, 1, mxINT32_CLASS, mxREAL); memcpy(mxGetData(r), rows1, (size_t)tot * 4);
    mxArray* o = mxCreateNumericMatrix(S + 1, 1, mxINT32_CLASS, mxREAL); memcpy(mxGetData(o), off, (size_t)(S + 1) * 4);
    std::vector<mxArray*> rhs{mxCreateString("getMatchesSegmented"), dmat(dS, Q, D), dmat(dM, VM, D), r, o, p};
    mxArray* lhs[2] = {nullptr, nullptr};
    if (call(2, lhs, rhs, err, errlen)) return 1;
    *P_total = (int)mxGetM(lhs[0]);
    memcpy(pairs_colmajor, mxGetData(lhs[0]), (size_t)*P_total * 2 * 4);
    memcpy(n_pairs, mxGetData(lhs[1]), (size_t)S * 4);
    mxDestroyArray(lhs[0]); mxDestroyArray(lhs[1]);
    return 0;
}

// pts / c in their own classes (single or double); out / dists: capacity n (as doubles); *cls_single = class of the results
int drv_get_local_points(const void* pts, int pts_single, int n, double R, const void* c, int c_single, double minp, double maxp,
                         double* out_colmajor, double* dists, int* n_out, int* cls_single, char* err, int errlen) {
    mxArray* p = mxCreateNumericMatrix(n, 3, pts_single ? mxSINGLE_CLASS : mxDOUBLE_CLASS, mxREAL);
    memcpy(mxGetData(p), pts, (size_t)n * 3 * (pts_single ? 4 : 8));
    mxArray* cc = mxCreateNumericMatrix(1, 3, c_single ? mxSINGLE_CLASS : mxDOUBLE_CLASS, mxREAL);
    memcpy(mxGetData(cc), c, 3 * (c_single ? 4 : 8));
    std::vector<mxArray*> rhs{mxCreateString("getLocalPoints"), p, mxCreateDoubleScalar(R), cc, mxCreateDoubleScalar(minp), mxCreateDoubleScalar(maxp)};
    mxArray* lhs[2] = {nullptr, nullptr};
    if (call(2, lhs, rhs, err, errlen)) return 1;
    *n_out = (int)mxGetM(lhs[0]);
    *cls_single = mxIsSingle(lhs[0]) ? 1 : 0;
    for (size_t k = 0; k < (size_t)*n_out * 3; ++k) out_colmajor[k] = *cls_single ? (double)((const float*)mxGetData(lhs[0]))[k] : mxGetPr(lhs[0])[k];
    for (int k = 0; k < *n_out; ++k) dists[k] = *cls_single ? (double)((const float*)mxGetData(lhs[1]))[k] : mxGetPr(lhs[1])[k];
    mxDestroyArray(lhs[0]); mxDestroyArray(lhs[1]);
    return 0;
}

int drv_align_points_knn(const double* pts, int n, int C1, int C2, double* aligned, double* coeff9, double* c3, char* err, int errlen) {
    std::vector<mxArray*> rhs{mxCreateString("AlignPoints_KNN"), dmat(pts, n, 3), mxCreateDoubleScalar(C1), mxCreateDoubleScalar(C2)};
    mxArray* lhs[3] = {nullptr, nullptr, nullptr};
    if (call(3, lhs, rhs, err, errlen)) return 1;
    memcpy(aligned, mxGetPr(lhs[0]), (size_t)n * 3 * 8); memcpy(coeff9, mxGetPr(lhs[1]), 72); memcpy(c3, mxGetPr(lhs[2]), 24);
    for (mxArray* a : lhs) mxDestroyArray(a);
    return 0;
}

// opts = {min_pts, max_pts, R, thVar1, thVar2, ALIGN_POINTS}; k is passed as the string 'all' like completeExperimentFast.m:303
int drv_descriptors(const double* pts, int Pn, const double* kp, int S, const double* opts6, double* feat /* cap S x 3 col-major */,
                    double* desc /* cap S x 980 col-major */, int* V, char* err, int errlen) {
    mxArray* o = mxCreateStructMatrix(1, 1, 0, nullptr);
    put(o, "min_pts", opts6[0]); put(o, "max_pts", opts6[1]); put(o, "R", opts6[2]);
    const double tv[2] = {opts6[3], opts6[4]};
    mxSetField(o, 0, "thVar", dmat(tv, 1, 2)); put(o, "ALIGN_POINTS", opts6[5]); mxSetField(o, 0, "k", mxCreateString("all")); put(o, "VERBOSE", 0);
    std::vector<mxArray*> rhs{mxCreateString("getSpacialHistogramDescriptors"), dmat(pts, Pn, 3), dmat(kp, S, 3), o};
    mxArray* lhs[2] = {nullptr, nullptr};
    if (call(2, lhs, rhs, err, errlen)) return 1;
    *V = (int)mxGetM(lhs[0]);
    memcpy(feat, mxGetPr(lhs[0]), (size_t)*V * 3 * 8);
    memcpy(desc, mxGetPr(lhs[1]), (size_t)*V * mxGetN(lhs[1]) * 8);
    for (mxArray* a : lhs) mxDestroyArray(a);
    return 0;
}

// `single` inputs: what matlab/AlignPoints_KNN.m and getSpacialHistogramDescriptors.m pass for pcread clouds
static mxArray* fmat(const float* p, size_t m, size_t n) { mxArray* a = mxCreateNumericMatrix(m, n, mxSINGLE_CLASS, mxREAL); if (m * n > 0) memcpy(mxGetData(a), p, m * n * 4); return a; }

int drv_align_points_knn_f32(const float* pts, int n, float* aligned, float* coeff9, float* c3, char* err, int errlen) {
    std::vector<mxArray*> rhs{mxCreateString("AlignPoints_KNN"), fmat(pts, n, 3), mxCreateDoubleScalar(0), mxCreateDoubleScalar(0)};
    mxArray* lhs[3] = {nullptr, nullptr, nullptr};
    if (call(3, lhs, rhs, err, errlen)) return 1;
    if (!mxIsSingle(lhs[0]) || !mxIsSingle(lhs[1]) || !mxIsSingle(lhs[2])) { snprintf(err, errlen, "outputs are not single"); return 1; }
    memcpy(aligned, mxGetData(lhs[0]), (size_t)n * 3 * 4); memcpy(coeff9, mxGetData(lhs[1]), 36); memcpy(c3, mxGetData(lhs[2]), 12);
    for (mxArray* a : lhs) mxDestroyArray(a);
    return 0;
}

// each input in its own class (single or double); the outputs must be DOUBLE whatever the inputs (getSpacialHistogramDescriptors.m:61-62)
int drv_descriptors_classes(const void* pts, int pts_single, int Pn, const void* kp, int kp_single, int S, const double* opts6, double* feat,
                            double* desc, int* V, char* err, int errlen) {
    mxArray* o = mxCreateStructMatrix(1, 1, 0, nullptr);
    put(o, "min_pts", opts6[0]); put(o, "max_pts", opts6[1]); put(o, "R", opts6[2]);
    const double tv[2] = {opts6[3], opts6[4]};
    mxSetField(o, 0, "thVar", dmat(tv, 1, 2)); put(o, "ALIGN_POINTS", opts6[5]); mxSetField(o, 0, "k", mxCreateString("all")); put(o, "VERBOSE", 0);
    std::vector<mxArray*> rhs{mxCreateString("getSpacialHistogramDescriptors"),
                              pts_single ? fmat((const float*)pts, Pn, 3) : dmat((const double*)pts, Pn, 3),
                              kp_single ? fmat((const float*)kp, S, 3) : dmat((const double*)kp, S, 3), o};
    mxArray* lhs[2] = {nullptr, nullptr};
    if (call(2, lhs, rhs, err, errlen)) return 1;
    if (!mxIsDouble(lhs[0]) || !mxIsDouble(lhs[1])) { snprintf(err, errlen, "feat / desc must be double whatever the input classes"); return 1; }
    *V = (int)mxGetM(lhs[0]);
    memcpy(feat, mxGetPr(lhs[0]), (size_t)*V * 3 * 8);
    memcpy(desc, mxGetPr(lhs[1]), (size_t)*V * mxGetN(lhs[1]) * 8);
    for (mxArray* a : lhs) mxDestroyArray(a);
    return 0;
}

// the prepared-model commands: modelCreate -> handle, modelMatchPoints x n_surfaces, modelDestroy
int drv_model_round_trip(const float* model, int M, const float* surfaces /* n_surf x (Q x 3 col-major) */, int Q, int n_surf, float thr, float ratio,
                         int unique, uint32_t* pairs_colmajor /* n_surf x (Q x 2) */, int* P /* n_surf */, char* err, int errlen) {
    mxArray* lhs[1] = {nullptr};
    { std::vector<mxArray*> rhs{mxCreateString("modelCreate"), fmat(model, M, 3)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* h = lhs[0]; lhs[0] = nullptr;
    if (!mxIsUint64(h)) { snprintf(err, errlen, "the handle is not uint64"); return 1; }
    int rc = 0;
    for (int k = 0; k < n_surf && !rc; ++k) {
        std::vector<mxArray*> rhs{mxCreateString("modelMatchPoints"), mxDuplicateArray(h), fmat(surfaces + (size_t)k * Q * 3, Q, 3), mxCreateDoubleScalar(thr),
                                  mxCreateDoubleScalar(ratio), mxCreateDoubleScalar(unique)};
        rc = call(1, lhs, rhs, err, errlen);
        if (!rc) {
            P[k] = (int)mxGetM(lhs[0]);
            memcpy(pairs_colmajor + (size_t)k * Q * 2, mxGetData(lhs[0]), (size_t)P[k] * 2 * 4);
            mxDestroyArray(lhs[0]); lhs[0] = nullptr;
        }
    }
    { std::vector<mxArray*> rhs{mxCreateString("modelDestroy"), h}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    return rc;
}

// descCreate x 2, one getMatchesOnSet per row subset (rows1: 1-based, subsets back to back, off: n_sub + 1 offsets; an EMPTY subset
// stands for "the whole model set", passed as []), descDestroy x 2
int drv_desc_set_round_trip(const double* dS, int Q, const double* dM, int VM, int D, const double* par7, const int32_t* rows1, const int32_t* off,
                            int n_sub, uint32_t* pairs_colmajor /* n_sub x (Q x 2) */, int* P /* n_sub */, char* err, int errlen) {
    mxArray* lhs[1] = {nullptr};
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dS, Q, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hS = lhs[0]; lhs[0] = nullptr;
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dM, VM, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hM = lhs[0]; lhs[0] = nullptr;
    if (!mxIsUint64(hS) || !mxIsUint64(hM)) { snprintf(err, errlen, "a handle is not uint64"); return 1; }
    int rc = 0;
    for (int k = 0; k < n_sub && !rc; ++k) {
        mxArray* p = mxCreateStructMatrix(1, 1, 0, nullptr);
        mxSetField(p, 0, "Metric", mxCreateString("SAD")); mxSetField(p, 0, "Method", mxCreateString("Approximate"));
        put(p, "MatchThreshold", par7[0]); put(p, "MaxRatio", par7[1]); put(p, "Unique", par7[2]); put(p, "UNNORMALIZE", par7[3]);
        put(p, "norm_factor", par7[4]); put(p, "CHANGE_METRIC", par7[5]); put(p, "metric_factor", par7[6]); put(p, "VERBOSE", 0);
        const int n = off[k + 1] - off[k];
        mxArray* r = n > 0 ? mxCreateNumericMatrix(n, 1, mxINT32_CLASS, mxREAL) : mxCreateDoubleMatrix(0, 0, mxREAL);
        if (n > 0) memcpy(mxGetData(r), rows1 + off[k], (size_t)n * 4);
        std::vector<mxArray*> rhs{mxCreateString("getMatchesOnSet"), mxDuplicateArray(hS), mxDuplicateArray(hM), r, p};
        rc = call(1, lhs, rhs, err, errlen);
        if (!rc) {
            P[k] = (int)mxGetM(lhs[0]);
            memcpy(pairs_colmajor + (size_t)k * Q * 2, mxGetData(lhs[0]), (size_t)P[k] * 2 * 4);
            mxDestroyArray(lhs[0]); lhs[0] = nullptr;
        }
    }
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hS}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hM}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    return rc;
}

// descCreate x 2, ONE getMatchesSegmentedOnSet for all row subsets (rows1: 1-based, subsets back to back, off: n_sub + 1 offsets),
// descDestroy x 2.  pairs_colmajor: sum(P) x 2 as the gateway returns it; P: the S counts
int drv_desc_set_segmented(const double* dS, int Q, const double* dM, int VM, int D, const double* par7, const int32_t* rows1, int n_rows,
                           const int32_t* off, int n_sub, uint32_t* pairs_colmajor, int* P, int* P_total, char* err, int errlen) {
    mxArray* lhs[2] = {nullptr, nullptr};
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dS, Q, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hS = lhs[0]; lhs[0] = nullptr;
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dM, VM, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hM = lhs[0]; lhs[0] = nullptr;
    mxArray* p = mxCreateStructMatrix(1, 1, 0, nullptr);
    mxSetField(p, 0, "Metric", mxCreateString("SAD")); mxSetField(p, 0, "Method", mxCreateString("Approximate"));
    put(p, "MatchThreshold", par7[0]); put(p, "MaxRatio", par7[1]); put(p, "Unique", par7[2]); put(p, "UNNORMALIZE", par7[3]);
    put(p, "norm_factor", par7[4]); put(p, "CHANGE_METRIC", par7[5]); put(p, "metric_factor", par7[6]); put(p, "VERBOSE", 0);
    mxArray* r = mxCreateNumericMatrix(n_rows > 0 ? n_rows : 0, 1, mxINT32_CLASS, mxREAL);
    if (n_rows > 0) memcpy(mxGetData(r), rows1, (size_t)n_rows * 4);
    mxArray* o = mxCreateNumericMatrix(n_sub + 1, 1, mxINT32_CLASS, mxREAL);
    memcpy(mxGetData(o), off, (size_t)(n_sub + 1) * 4);
    int rc = 0;
    {
        std::vector<mxArray*> rhs{mxCreateString("getMatchesSegmentedOnSet"), mxDuplicateArray(hS), mxDuplicateArray(hM), r, o, p};
        rc = call(2, lhs, rhs, err, errlen);
    }
    if (!rc) {
        *P_total = (int)mxGetM(lhs[0]);
        memcpy(pairs_colmajor, mxGetData(lhs[0]), (size_t)*P_total * 2 * 4);
        memcpy(P, mxGetData(lhs[1]), (size_t)n_sub * 4);
        mxDestroyArray(lhs[0]); mxDestroyArray(lhs[1]); lhs[0] = lhs[1] = nullptr;
    }
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hS}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hM}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    return rc;
}

// sphereCounts, descCreate x 2, sphereSweep, descDestroy x 2 through the gateway.  Every output flattened as the gateway returns it
// (rows1 / trial1 are 1-based); capacities: rows1 >= sum(num_desc of the kept spheres), pairs_colmajor >= S x Q x 2, the per-trial arrays >= S.
int drv_sphere_sweep(const double* dS, int Q, const double* dM, int VM, int D, const double* fS, const double* fM, const double* centres, int n_c, double R,
                     double min_pts, const double* par7, double thresh, const double* coef5, double seed, double* counts, double* rows1, int* n_rows,
                     uint32_t* pairs_colmajor, int* P_total, double* n_pairs, int* S_out, double* trial1, int* n_trials, double* T16, double* num_success,
                     double* max_inl, double* failed, char* err, int errlen) {
    mxArray* lhs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    { std::vector<mxArray*> rhs{mxCreateString("sphereCounts"), dmat(fM, VM, 3), dmat(centres, n_c, 3), mxCreateDoubleScalar(R)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    memcpy(counts, mxGetPr(lhs[0]), (size_t)n_c * 8);
    mxDestroyArray(lhs[0]); lhs[0] = nullptr;
    std::vector<double> kept; std::vector<int32_t> nd;
    for (int i = 0; i < n_c; ++i) if (counts[i] >= min_pts) nd.push_back((int32_t)counts[i]);
    const int S = (int)nd.size();
    kept.resize((size_t)S * 3);
    for (int i = 0, k = 0; i < n_c; ++i) if (counts[i] >= min_pts) { for (int c = 0; c < 3; ++c) kept[k + (size_t)c * S] = centres[i + (size_t)c * n_c]; ++k; }
    *S_out = S;
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dS, Q, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hS = lhs[0]; lhs[0] = nullptr;
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dM, VM, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hM = lhs[0]; lhs[0] = nullptr;
    mxArray* p = mxCreateStructMatrix(1, 1, 0, nullptr);
    mxSetField(p, 0, "Metric", mxCreateString("SAD")); mxSetField(p, 0, "Method", mxCreateString("Approximate"));
    put(p, "MatchThreshold", par7[0]); put(p, "MaxRatio", par7[1]); put(p, "Unique", par7[2]); put(p, "UNNORMALIZE", par7[3]);
    put(p, "norm_factor", par7[4]); put(p, "CHANGE_METRIC", par7[5]); put(p, "metric_factor", par7[6]); put(p, "VERBOSE", 0);
    mxArray* c = mxCreateStructMatrix(1, 1, 0, nullptr);
    put(c, "minPtNum", coef5[0]); put(c, "iterNum", coef5[1]); put(c, "thDist", coef5[2]); put(c, "thInlrRatio", coef5[3]);
    put(c, "REFINE", coef5[4]); put(c, "VERBOSE", 0);
    mxArray* ndm = mxCreateNumericMatrix(S, 1, mxINT32_CLASS, mxREAL);
    if (S) memcpy(mxGetData(ndm), nd.data(), (size_t)S * 4);
    int rc = 0;
    {
        std::vector<mxArray*> rhs{mxCreateString("sphereSweep"), mxDuplicateArray(hS), mxDuplicateArray(hM), dmat(fS, Q, 3), dmat(fM, VM, 3), dmat(kept.data(), S, 3), ndm,
                                  mxCreateDoubleScalar(R), p, mxCreateDoubleScalar(thresh), c, mxCreateDoubleScalar(seed)};
        rc = call(8, lhs, rhs, err, errlen);
    }
    if (!rc) {
        *n_rows = (int)(mxGetM(lhs[0]) * mxGetN(lhs[0])); if (*n_rows) memcpy(rows1, mxGetPr(lhs[0]), (size_t)*n_rows * 8);
        *P_total = (int)mxGetM(lhs[1]); if (*P_total) memcpy(pairs_colmajor, mxGetData(lhs[1]), (size_t)*P_total * 2 * 4);
        if (S) memcpy(n_pairs, mxGetPr(lhs[2]), (size_t)S * 8);
        *n_trials = (int)(mxGetM(lhs[3]) * mxGetN(lhs[3]));
        if (*n_trials) {
            memcpy(trial1, mxGetPr(lhs[3]), (size_t)*n_trials * 8); memcpy(T16, mxGetPr(lhs[4]), (size_t)*n_trials * 128);
            memcpy(num_success, mxGetPr(lhs[5]), (size_t)*n_trials * 8); memcpy(max_inl, mxGetPr(lhs[6]), (size_t)*n_trials * 8);
            memcpy(failed, mxGetPr(lhs[7]), (size_t)*n_trials * 8);
        }
        for (mxArray*& a : lhs) { mxDestroyArray(a); a = nullptr; }
    }
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hS}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hM}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    return rc;
}

// matlab/sphereSweepModel.m + sphereSweepOn.m: the model handle made once (and the model set destroyed right after), then n_surf
// surfaces (rows of dS / fS back to back, Q rows each) swept against it; outputs of surface u at offset u of every buffer
int drv_sphere_sweep_model(const double* dS, const double* fS, int n_surf, int Q, const double* dM, int VM, int D, const double* fM, const double* kept, int S,
                           const int32_t* nd, double R, const double* par7, double thresh, const double* coef5, double seed, double* rows1, int* n_rows,
                           uint32_t* pairs_colmajor, int* P_total, double* n_pairs, double* trial1, int* n_trials, double* T16, double* num_success,
                           double* max_inl, double* failed, char* err, int errlen) {
    mxArray* lhs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dM, VM, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* hM = lhs[0]; lhs[0] = nullptr;
    mxArray* ndm = mxCreateNumericMatrix(S, 1, mxINT32_CLASS, mxREAL);
    if (S) memcpy(mxGetData(ndm), nd, (size_t)S * 4);
    { std::vector<mxArray*> rhs{mxCreateString("sphereModelCreate"), mxDuplicateArray(hM), dmat(fM, VM, 3), dmat(kept, S, 3), ndm, mxCreateDoubleScalar(R)};
      if (call(2, lhs, rhs, err, errlen)) return 1; }
    mxArray* sm = lhs[0]; lhs[0] = nullptr;
    *n_rows = (int)(mxGetM(lhs[1]) * mxGetN(lhs[1])); if (*n_rows) memcpy(rows1, mxGetPr(lhs[1]), (size_t)*n_rows * 8);
    mxDestroyArray(lhs[1]); lhs[1] = nullptr;
    { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hM}; if (call(0, lhs, rhs, err, errlen)) return 1; }      // the handle owns what it needs
    int rc = 0;
    for (int u = 0; u < n_surf && !rc; ++u) {
        { std::vector<mxArray*> rhs{mxCreateString("descCreate"), dmat(dS + (size_t)u * Q * D, Q, D)}; if (call(1, lhs, rhs, err, errlen)) return 1; }
        mxArray* hS = lhs[0]; lhs[0] = nullptr;
        mxArray* p = mxCreateStructMatrix(1, 1, 0, nullptr);
        mxSetField(p, 0, "Metric", mxCreateString("SAD")); mxSetField(p, 0, "Method", mxCreateString("Approximate"));
        put(p, "MatchThreshold", par7[0]); put(p, "MaxRatio", par7[1]); put(p, "Unique", par7[2]); put(p, "UNNORMALIZE", par7[3]);
        put(p, "norm_factor", par7[4]); put(p, "CHANGE_METRIC", par7[5]); put(p, "metric_factor", par7[6]); put(p, "VERBOSE", 0);
        mxArray* c = mxCreateStructMatrix(1, 1, 0, nullptr);
        put(c, "minPtNum", coef5[0]); put(c, "iterNum", coef5[1]); put(c, "thDist", coef5[2]); put(c, "thInlrRatio", coef5[3]);
        put(c, "REFINE", coef5[4]); put(c, "VERBOSE", 0);
        {
            std::vector<mxArray*> rhs{mxCreateString("sphereSweepOnModel"), mxDuplicateArray(sm), mxDuplicateArray(hS), dmat(fS + (size_t)u * Q * 3, Q, 3),
                                      mxCreateDoubleScalar(S), p, mxCreateDoubleScalar(thresh), c, mxCreateDoubleScalar(seed)};
            rc = call(7, lhs, rhs, err, errlen);
        }
        if (!rc) {
            const size_t us = (size_t)u * S;
            P_total[u] = (int)mxGetM(lhs[0]); if (P_total[u]) memcpy(pairs_colmajor + (size_t)u * S * Q * 2, mxGetData(lhs[0]), (size_t)P_total[u] * 2 * 4);
            if (S) memcpy(n_pairs + us, mxGetPr(lhs[1]), (size_t)S * 8);
            n_trials[u] = (int)(mxGetM(lhs[2]) * mxGetN(lhs[2]));
            if (n_trials[u]) {
                memcpy(trial1 + us, mxGetPr(lhs[2]), (size_t)n_trials[u] * 8); memcpy(T16 + us * 16, mxGetPr(lhs[3]), (size_t)n_trials[u] * 128);
                memcpy(num_success + us, mxGetPr(lhs[4]), (size_t)n_trials[u] * 8); memcpy(max_inl + us, mxGetPr(lhs[5]), (size_t)n_trials[u] * 8);
                memcpy(failed + us, mxGetPr(lhs[6]), (size_t)n_trials[u] * 8);
            }
            for (mxArray*& a : lhs) { mxDestroyArray(a); a = nullptr; }
        }
        { std::vector<mxArray*> rhs{mxCreateString("descDestroy"), hS}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    }
    { std::vector<mxArray*> rhs{mxCreateString("sphereModelDestroy"), sm}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    return rc;
}

// one-worker rehearsal of the spmd block of INTEGRATION.md section 3: setDevice, commId, commInit, matchPointsSharded,
// ransacSharded, commDestroy -- all through the gateway
int drv_comm_round_trip(const float* surf, int Q, const float* model, int M, float thr, float ratio, uint32_t* pairs_colmajor, int* P,
                        const double* coef5, double seed, double* T16, double* inlier_idx, int* n_inl, int* num_success, int* max_inl, int* failed,
                        const double* p1, const double* p2, int n, char* err, int errlen) {
    mxArray* lhs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    { std::vector<mxArray*> rhs{mxCreateString("setDevice"), mxCreateDoubleScalar(0)}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    { std::vector<mxArray*> rhs{mxCreateString("commId")}; if (call(1, lhs, rhs, err, errlen)) return 1; }
    mxArray* id = lhs[0]; lhs[0] = nullptr;
    if (!mxIsUint8(id) || mxGetN(id) != 128) { snprintf(err, errlen, "commId is not 1 x 128 uint8"); return 1; }
    { std::vector<mxArray*> rhs{mxCreateString("commInit"), mxCreateDoubleScalar(0), mxCreateDoubleScalar(1), id}; if (call(0, lhs, rhs, err, errlen)) return 1; }
    int rc = 0;
    {
        std::vector<mxArray*> rhs{mxCreateString("matchPointsSharded"), fmat(surf, Q, 3), fmat(model, M, 3), mxCreateDoubleScalar(0), mxCreateDoubleScalar(M),
                                  mxCreateDoubleScalar(thr), mxCreateDoubleScalar(ratio), mxCreateDoubleScalar(1)};
        rc = call(1, lhs, rhs, err, errlen);
        if (!rc) { *P = (int)mxGetM(lhs[0]); memcpy(pairs_colmajor, mxGetData(lhs[0]), (size_t)*P * 2 * 4); mxDestroyArray(lhs[0]); lhs[0] = nullptr; }
    }
    if (!rc) {
        mxArray* c = mxCreateStructMatrix(1, 1, 0, nullptr);
        put(c, "minPtNum", coef5[0]); put(c, "iterNum", coef5[1]); put(c, "thDist", coef5[2]); put(c, "thInlrRatio", coef5[3]); put(c, "REFINE", coef5[4]);
        std::vector<mxArray*> rhs{mxCreateString("ransacSharded"), dmat(p1, n, 3), dmat(p2, n, 3), c, mxCreateDoubleScalar(seed)};
        rc = call(5, lhs, rhs, err, errlen);
        if (!rc) {
            *failed = (int)mxGetScalar(lhs[4]);
            if (!*failed) memcpy(T16, mxGetPr(lhs[0]), 128);
            *n_inl = (int)(mxGetM(lhs[1]) * mxGetN(lhs[1]));
            if (*n_inl) memcpy(inlier_idx, mxGetPr(lhs[1]), (size_t)*n_inl * 8);
            *num_success = (int)mxGetScalar(lhs[2]); *max_inl = (int)mxGetScalar(lhs[3]);
            for (mxArray*& a : lhs) { mxDestroyArray(a); a = nullptr; }
        }
    }
    { char e2[256]; std::vector<mxArray*> rhs{mxCreateString("commDestroy")}; call(0, lhs, rhs, e2, sizeof e2); }
    return rc;
}

int drv_bad_command(char* err, int errlen) {
    std::vector<mxArray*> rhs{mxCreateString("noSuchCommand")};
    mxArray* lhs[1] = {nullptr};
    return call(1, lhs, rhs, err, errlen);
}

}  // extern "C"
