// mex/pcreg_mex.cpp -- the MATLAB binding of libpcreg_hip.so (include/pcreg.h).
//
// One gateway, string-dispatched:   out = pcreg_mex('<command>', args...)
//   'estimateTransform', pts1, pts2                      -> T (4x4 or [])
//   'calcDists', T, pts1, pts2                           -> d (n x 1)
//   'ransac', pts1, pts2, coef, sample_idx|[], seed      -> T, inlierIdx, numSuccess, maxInliers, failed
//   'getMatches', descSurface, descModel, par            -> matches (P x 2 uint32)
//   'AlignPoints_KNN', pts, C1, C2                       -> pts_aligned, coeff_unambig, c
//   'getSpacialHistogramDescriptors', pts, sample_pts, options -> feat (V x 3), desc (V x 980)
//   'modelCreate', model (single M x 3) -> handle (uint64) | 'modelMatchPoints', handle, surface (single Q x 3), thrAbs, maxRatio, unique
//                                          -> pairs (P x 2 uint32) | 'modelDestroy', handle      (one model, many surfaces)
//   'descCreate', desc (double n x D) -> handle (uint64) | 'getMatchesOnSet', hSurface, hModel, int32 rows | [], par -> matches
//                                          | 'descDestroy', handle        (one surface set, many row subsets of one model set)
//   'getMatchesSegmented', descSurface, descModel, int32 rows, int32 segOff, par | 'getMatchesSegmentedOnSet', hSurface, hModel, ...
//                                          -> pairs, nPairs                (every sphere of the sweep in one call)
//   'ransacBatched', pts1, pts2, int32 offsets, coef, sample_idx|[], seed -> T (4x4xB), inlierIdx, nInliers, numSuccess, maxInliers, failed
//   'sphereCounts', featModel, centres, R -> counts | 'sphereSweep', hSurface, hModel, featSurface, featModel, centres, int32 numDesc,
//       R_desc, par, putativeThresh, coef, seed -> modelRows, pairs, nPairs, trial, T, numSuccess, maxInliers, failed   (completeExperimentFast.m:52-224)
//   'sphereModelCreate', hModel, featModel, centres, int32 numDesc, R_desc -> handle, modelRows | 'sphereSweepOnModel', hSphereModel, hSurface,
//       featSurface, S, par, putativeThresh, coef, seed -> pairs, nPairs, trial, T, numSuccess, maxInliers, failed | 'sphereModelDestroy', handle
//   'getLocalPoints', pts, R, c, min_points, max_points   -> pts_sphere, dists
//   'setDevice', ordinal | 'commId' -> id | 'commInit', rank, world, id | 'commDestroy'     (one worker per GPU)
//   'matchPointsSharded', surface, modelRows, m_lo, M_total, thrAbs, maxRatio, unique     -> pairs (P x 2 uint32, global model rows)
//   'ransacSharded', pts1, pts2, coef, seed                -> T, inlierIdx, numSuccess, maxInliers, failed
// The shim only unpacks mxArrays: MATLAB's column-major doubles go straight through
// (ld = number of rows).  It never throws with C++ objects alive:
// errors are collected as codes and raised by one mexErrMsgIdAndTxt at the very end.
//
// Build (on a machine with MATLAB; neither the build container nor the GPU box has one):
//   mex -I../include mex/pcreg_mex.cpp -L../pcreg_amd -lpcreg_hip -output matlab/pcreg_mex
// This file is compile-gated on mex.h and is NOT part of libpcreg_hip.so.
#if __has_include("mex.h")
#include "mex.h"
#include <cstdint>
#include <cstring>
#include <string>
#include "../include/pcreg.h"

static double field(const mxArray* s, const char* name, double dflt, bool* missing = nullptr) {
    const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, name) : nullptr;
    if (!f) { if (missing) *missing = true; return dflt; }
    if (mxIsChar(f)) return dflt;
    return mxGetScalar(f);
}
static bool field_is(const mxArray* s, const char* name, const char* value) {
    const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, name) : nullptr;
    if (!f || !mxIsChar(f)) return false;
    char buf[64]; mxGetString(f, buf, sizeof buf);
    return strcmp(buf, value) == 0;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) { mexErrMsgIdAndTxt("pcreg:usage", "pcreg_mex('<command>', ...)"); return; }
    char cmd[64]; mxGetString(prhs[0], cmd, sizeof cmd);
    int rc = PCREG_OK;
    const char* usage = nullptr;

    if (!strcmp(cmd, "estimateTransform")) {
        if (nrhs != 3) usage = "estimateTransform: pts1, pts2";
        else {
            int n = (int)mxGetM(prhs[1]);
            double T[16]; int empty = 1;
            rc = pcreg_estimate_transform(mxGetPr(prhs[1]), mxGetPr(prhs[2]), n, n, T, &empty);
            if (rc == PCREG_OK) {
                plhs[0] = empty ? mxCreateDoubleMatrix(0, 0, mxREAL) : mxCreateDoubleMatrix(4, 4, mxREAL);
                if (!empty) memcpy(mxGetPr(plhs[0]), T, sizeof T);
            }
        }
    } else if (!strcmp(cmd, "calcDists")) {
        if (nrhs != 4) usage = "calcDists: T, pts1, pts2";
        else {
            int n = (int)mxGetM(prhs[2]);
            plhs[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
            rc = pcreg_calc_dists(mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), n, n, mxGetPr(plhs[0]));
        }
    } else if (!strcmp(cmd, "ransac")) {
        if (nrhs < 4) usage = "ransac: pts1, pts2, coef[, sample_idx, seed]";
        else {
            const mxArray* c = prhs[3];
            pcreg_ransac_opts o;
            o.minPtNum = (int)field(c, "minPtNum", 3); o.iterNum = (int)field(c, "iterNum", 1000);
            o.thDist = field(c, "thDist", 0.5); o.thInlrRatio = field(c, "thInlrRatio", 0.1);
            o.REFINE = (int)field(c, "REFINE", 1); o.VERBOSE = (int)field(c, "VERBOSE", 1);
            o.seed = nrhs > 5 ? (uint64_t)mxGetScalar(prhs[5]) : 0;
            int n = (int)mxGetM(prhs[1]);
            // sample_idx: int32 minPtNum x iterNum (each COLUMN one hypothesis) or []
            const int32_t* si = (nrhs > 4 && !mxIsEmpty(prhs[4]) && mxIsInt32(prhs[4])) ? (const int32_t*)mxGetData(prhs[4]) : nullptr;
            mxArray* inl = mxCreateNumericMatrix(n > 0 ? n : 1, 1, mxINT32_CLASS, mxREAL);
            double T[16]; int ni = 0, ns = 0, mi = 0, fl = 1;
            rc = pcreg_ransac(mxGetPr(prhs[1]), mxGetPr(prhs[2]), n, n, &o, si, T, (int32_t*)mxGetData(inl), &ni, &ns, &mi, &fl, nullptr, nullptr);
            if (rc == PCREG_OK) {
                plhs[0] = fl ? mxCreateDoubleMatrix(0, 0, mxREAL) : mxCreateDoubleMatrix(4, 4, mxREAL);
                if (!fl) memcpy(mxGetPr(plhs[0]), T, sizeof T);
                if (nlhs > 1) {                         // inlierIdx as a double column, like find()
                    plhs[1] = mxCreateDoubleMatrix(fl ? 0 : ni, fl ? 0 : 1, mxREAL);
                    const int32_t* src = (const int32_t*)mxGetData(inl);
                    for (int k = 0; k < (fl ? 0 : ni); ++k) mxGetPr(plhs[1])[k] = (double)src[k];
                }
                if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(ns);
                if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(mi);
                if (nlhs > 4) plhs[4] = mxCreateDoubleScalar(fl);
            }
            mxDestroyArray(inl);
        }
    } else if (!strcmp(cmd, "ransacBatched")) {                // [T, inlierIdx, nInliers, numSuccess, maxInliers, failed] =
        // pcreg_mex('ransacBatched', pts1, pts2, int32(offsets), coef[, sampleIdx, seed]): B registrations in one launch (the parfor of
        // completeExperimentFast.m:201-216).  pts1 / pts2: the registrations' rows back to back (total x 3); offsets: B + 1 running row
        // offsets (0-based); sampleIdx: int32 minPtNum x (iterNum B), registration after registration, or []; seed: registration b
        // samples with seed + b.  T: 4 x 4 x B (zeros where failed); inlierIdx: the lists back to back (1-based inside a registration),
        // nInliers / numSuccess / maxInliers / failed: B x 1.  matlab/ransacBatched.m
        if (nrhs < 5 || !mxIsInt32(prhs[3])) usage = "ransacBatched: pts1, pts2, int32 offsets, coef[, sample_idx, seed]";
        else {
            const mxArray* c = prhs[4];
            pcreg_ransac_opts o;
            o.minPtNum = (int)field(c, "minPtNum", 3); o.iterNum = (int)field(c, "iterNum", 1000);
            o.thDist = field(c, "thDist", 0.5); o.thInlrRatio = field(c, "thInlrRatio", 0.1);
            o.REFINE = (int)field(c, "REFINE", 1); o.VERBOSE = (int)field(c, "VERBOSE", 1);
            o.seed = nrhs > 6 ? (uint64_t)mxGetScalar(prhs[6]) : 0;
            const int total = (int)mxGetM(prhs[1]);
            const int B = (int)(mxGetM(prhs[3]) * mxGetN(prhs[3])) - 1;
            const int32_t* off = (const int32_t*)mxGetData(prhs[3]);
            const int32_t* si = (nrhs > 5 && !mxIsEmpty(prhs[5]) && mxIsInt32(prhs[5])) ? (const int32_t*)mxGetData(prhs[5]) : nullptr;
            if (B < 0 || off[0] != 0 || off[B] != total || (int)mxGetM(prhs[2]) != total) usage = "ransacBatched: offsets must run from 0 to size(pts1, 1) = size(pts2, 1)";
            else if (si && mxGetM(prhs[5]) * mxGetN(prhs[5]) != (size_t)o.minPtNum * (size_t)o.iterNum * (size_t)B) usage = "ransacBatched: sampleIdx must be minPtNum x (iterNum * B)";
            else {
                const size_t b1 = (size_t)(B > 0 ? B : 1);
                mxArray* inl = mxCreateNumericMatrix(total > 0 ? total : 1, 1, mxINT32_CLASS, mxREAL);
                mxArray* ni = mxCreateNumericMatrix(b1, 1, mxINT32_CLASS, mxREAL); mxArray* ns = mxCreateNumericMatrix(b1, 1, mxINT32_CLASS, mxREAL);
                mxArray* mi = mxCreateNumericMatrix(b1, 1, mxINT32_CLASS, mxREAL); mxArray* fl = mxCreateNumericMatrix(b1, 1, mxINT32_CLASS, mxREAL);
                mwSize dims[3] = {4, 4, (mwSize)B};
                plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
                rc = B == 0 ? PCREG_OK : pcreg_ransac_batched(mxGetPr(prhs[1]), mxGetPr(prhs[2]), total, total > 0 ? total : 1, off, B, &o, si, mxGetPr(plhs[0]),
                                                               (int32_t*)mxGetData(inl), (int32_t*)mxGetData(ni), (int32_t*)mxGetData(ns), (int32_t*)mxGetData(mi),
                                                               (int32_t*)mxGetData(fl));
                if (rc == PCREG_OK) {
                    const int32_t *n_ = (const int32_t*)mxGetData(ni), *f_ = (const int32_t*)mxGetData(fl), *src = (const int32_t*)mxGetData(inl);
                    size_t tot_inl = 0;
                    for (int b = 0; b < B; ++b) {
                        if (f_[b]) memset(mxGetPr(plhs[0]) + (size_t)b * 16, 0, 128);
                        else tot_inl += (size_t)n_[b];
                    }
                    if (nlhs > 1) {                     // the inlier lists back to back, as a double column like find()
                        plhs[1] = mxCreateDoubleMatrix(tot_inl, tot_inl ? 1 : 0, mxREAL);
                        size_t k = 0;
                        for (int b = 0; b < B; ++b) if (!f_[b]) for (int e = 0; e < n_[b]; ++e) mxGetPr(plhs[1])[k++] = (double)src[off[b] + e];
                    }
                    auto col = [&](const mxArray* a, bool zero_failed) {
                        mxArray* out = mxCreateDoubleMatrix(B, B ? 1 : 0, mxREAL);
                        const int32_t* v = (const int32_t*)mxGetData(a);
                        for (int b = 0; b < B; ++b) mxGetPr(out)[b] = (zero_failed && f_[b]) ? 0.0 : (double)v[b];
                        return out;
                    };
                    if (nlhs > 2) plhs[2] = col(ni, true);
                    if (nlhs > 3) plhs[3] = col(ns, false);
                    if (nlhs > 4) plhs[4] = col(mi, false);
                    if (nlhs > 5) plhs[5] = col(fl, false);
                }
                mxDestroyArray(inl); mxDestroyArray(ni); mxDestroyArray(ns); mxDestroyArray(mi); mxDestroyArray(fl);
            }
        }
    } else if (!strcmp(cmd, "getMatches")) {
        if (nrhs != 4) usage = "getMatches: descSurface, descModel, par";
        else {
            const mxArray* p = prhs[3];
            pcreg_match_opts o;
            o.metric = field_is(p, "Metric", "SAD") ? PCREG_METRIC_SAD : PCREG_METRIC_SSD;
            o.matchThreshold = field(p, "MatchThreshold", 1.0); o.maxRatio = field(p, "MaxRatio", 0.6);
            o.unique = (int)field(p, "Unique", 0); o.prenormalized = 0;
            o.unnormalize = (int)field(p, "UNNORMALIZE", 0); o.norm_factor = field(p, "norm_factor", 0.0);
            o.change_metric = (int)field(p, "CHANGE_METRIC", 0); o.metric_factor = field(p, "metric_factor", 1.0);
            int Q = (int)mxGetM(prhs[1]), M = (int)mxGetM(prhs[2]), D = (int)mxGetN(prhs[1]);
            mxArray* buf = mxCreateNumericMatrix(2, Q > 0 ? Q : 1, mxUINT32_CLASS, mxREAL);   // row-major pairs = 2 x Q column-major
            int P = 0;
            rc = pcreg_get_matches(mxGetPr(prhs[1]), Q, Q, mxGetPr(prhs[2]), M, M, D, &o, (uint32_t*)mxGetData(buf), nullptr, &P);
            if (rc == PCREG_OK) {
                plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                for (int k = 0; k < P; ++k) { dst[k] = src[2 * k]; dst[k + P] = src[2 * k + 1]; }
            }
            mxDestroyArray(buf);
        }
    } else if (!strcmp(cmd, "getMatchesSegmented")) {         // [pairs, nPairs] = pcreg_mex('getMatchesSegmented', descSurface, descModel, int32(rows), int32(segOff), par)
        // rows: the model rows of all segments back to back (1-based, ascending inside a segment); segOff: S + 1 running offsets
        // (0-based); pairs: sum(nPairs) x 2 uint32, segment after segment; nPairs: S x 1 int32.  matlab/getMatchesSegmented.m
        if (nrhs != 6 || !mxIsInt32(prhs[3]) || !mxIsInt32(prhs[4])) usage = "getMatchesSegmented: descSurface, descModel, int32 rows, int32 segOff, par";
        else {
            const mxArray* p = prhs[5];
            pcreg_match_opts o;
            o.metric = field_is(p, "Metric", "SAD") ? PCREG_METRIC_SAD : PCREG_METRIC_SSD;
            o.matchThreshold = field(p, "MatchThreshold", 1.0); o.maxRatio = field(p, "MaxRatio", 0.6);
            o.unique = (int)field(p, "Unique", 0); o.prenormalized = 0;
            o.unnormalize = (int)field(p, "UNNORMALIZE", 0); o.norm_factor = field(p, "norm_factor", 0.0);
            o.change_metric = (int)field(p, "CHANGE_METRIC", 0); o.metric_factor = field(p, "metric_factor", 1.0);
            const int Q = (int)mxGetM(prhs[1]), VM = (int)mxGetM(prhs[2]), D = (int)mxGetN(prhs[1]);
            const int S = (int)(mxGetM(prhs[4]) * mxGetN(prhs[4])) - 1;
            const size_t tot = mxGetM(prhs[3]) * mxGetN(prhs[3]);
            const int32_t* off = (const int32_t*)mxGetData(prhs[4]);
            if (S < 0 || off[0] != 0 || (size_t)off[S] != tot) usage = "getMatchesSegmented: segOff must run from 0 to numel(rows)";
            else {
                mxArray* r0 = mxCreateNumericMatrix(tot > 0 ? tot : 1, 1, mxINT32_CLASS, mxREAL);
                mxArray* buf = mxCreateNumericMatrix(2, (size_t)(S > 0 ? S : 1) * (Q > 0 ? Q : 1), mxUINT32_CLASS, mxREAL);
                mxArray* np = mxCreateNumericMatrix(S > 0 ? S : 1, 1, mxINT32_CLASS, mxREAL);
                int32_t* rz = (int32_t*)mxGetData(r0); const int32_t* r1 = (const int32_t*)mxGetData(prhs[3]);
                for (size_t k = 0; k < tot; ++k) rz[k] = r1[k] - 1;
                rc = pcreg_get_matches_segmented(mxGetPr(prhs[1]), Q, Q, mxGetPr(prhs[2]), VM, VM, D, rz, off, S, &o, (uint32_t*)mxGetData(buf),
                                                 (int32_t*)mxGetData(np));
                if (rc == PCREG_OK) {
                    const int32_t* n = (const int32_t*)mxGetData(np);
                    size_t P = 0; for (int z = 0; z < S; ++z) P += (size_t)n[z];
                    plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                    const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                    size_t k = 0;
                    for (int z = 0; z < S; ++z)
                        for (int e = 0; e < n[z]; ++e, ++k) { dst[k] = src[((size_t)z * Q + e) * 2]; dst[k + P] = src[((size_t)z * Q + e) * 2 + 1]; }
                    if (nlhs > 1) { plhs[1] = mxCreateNumericMatrix(S, 1, mxINT32_CLASS, mxREAL); memcpy(mxGetData(plhs[1]), n, (size_t)S * 4); }
                }
                mxDestroyArray(r0); mxDestroyArray(buf); mxDestroyArray(np);
            }
        }
    } else if (!strcmp(cmd, "AlignPoints_KNN")) {
        if (nrhs != 4) usage = "AlignPoints_KNN: pts, C1, C2";
        else {
            int n = (int)mxGetM(prhs[1]);
            const bool sgl = mxIsSingle(prhs[1]);                // single in -> single out, like MATLAB's own function
            const mxClassID cls = sgl ? mxSINGLE_CLASS : mxDOUBLE_CLASS;
            plhs[0] = mxCreateNumericMatrix(n, 3, cls, mxREAL);
            mxArray* co = mxCreateNumericMatrix(3, 3, cls, mxREAL); mxArray* c = mxCreateNumericMatrix(1, 3, cls, mxREAL);
            if (sgl) rc = pcreg_align_points_knn_f32((const float*)mxGetData(prhs[1]), n, n, (int)mxGetScalar(prhs[2]), (int)mxGetScalar(prhs[3]),
                                                     (float*)mxGetData(plhs[0]), (float*)mxGetData(co), (float*)mxGetData(c));
            else rc = pcreg_align_points_knn(mxGetPr(prhs[1]), n, n, (int)mxGetScalar(prhs[2]), (int)mxGetScalar(prhs[3]),
                                             mxGetPr(plhs[0]), mxGetPr(co), mxGetPr(c));
            if (nlhs > 1) plhs[1] = co; else mxDestroyArray(co);
            if (nlhs > 2) plhs[2] = c; else mxDestroyArray(c);
        }
    } else if (!strcmp(cmd, "getSpacialHistogramDescriptors")) {
        if (nrhs != 4) usage = "getSpacialHistogramDescriptors: pts, sample_pts, options";
        else {
            const mxArray* p = prhs[3];
            pcreg_desc_opts o;
            double mx = field(p, "max_pts", 6000);
            o.min_pts = (int)field(p, "min_pts", 500); o.max_pts = mx > 2147483647.0 ? 2147483647 : (int)mx;
            o.R = field(p, "R", 3.5); o.ALIGN_POINTS = (int)field(p, "ALIGN_POINTS", 1);
            const mxArray* tv = mxGetField(p, 0, "thVar");
            o.thVar[0] = tv ? mxGetPr(tv)[0] : 3.0; o.thVar[1] = tv ? mxGetPr(tv)[1] : 1.5;
            const mxArray* kk = mxGetField(p, 0, "k");
            o.k = (!kk || mxIsChar(kk)) ? 1.0 : mxGetScalar(kk);           // 'all' -> 1
            int P = (int)mxGetM(prhs[1]), S = (int)mxGetM(prhs[2]);
            // each input in its own class (single or double); feat / desc are DOUBLE whatever the inputs: the reference
            // preallocates them with nan(...) and assigns into them (getSpacialHistogramDescriptors.m:61-62)
            const int ps = mxIsSingle(prhs[1]) ? 1 : 0, ss = mxIsSingle(prhs[2]) ? 1 : 0;
            mxArray* f = mxCreateNumericMatrix(3, S > 0 ? S : 1, mxDOUBLE_CLASS, mxREAL);     // row-major V x 3 == 3 x V column-major
            mxArray* d = mxCreateNumericMatrix(PCREG_DESC_LEN, S > 0 ? S : 1, mxDOUBLE_CLASS, mxREAL);
            int V = 0;
            rc = pcreg_spatial_histogram_descriptors_mixed(mxGetData(prhs[1]), ps, P, P, mxGetData(prhs[2]), ss, S, S, &o, mxGetPr(f), mxGetPr(d), &V);
            if (rc == PCREG_OK) {      // transpose into MATLAB's V x 3 / V x 980
                plhs[0] = mxCreateNumericMatrix(V, 3, mxDOUBLE_CLASS, mxREAL);
                if (nlhs > 1) plhs[1] = mxCreateNumericMatrix(V, PCREG_DESC_LEN, mxDOUBLE_CLASS, mxREAL);
                for (int v = 0; v < V; ++v) for (int c = 0; c < 3; ++c) mxGetPr(plhs[0])[v + (size_t)c * V] = mxGetPr(f)[c + 3 * (size_t)v];
                if (nlhs > 1) for (int v = 0; v < V; ++v) for (int c = 0; c < PCREG_DESC_LEN; ++c) mxGetPr(plhs[1])[v + (size_t)c * V] = mxGetPr(d)[c + PCREG_DESC_LEN * (size_t)v];
            }
            mxDestroyArray(f); mxDestroyArray(d);
        }
    } else if (!strcmp(cmd, "getLocalPoints")) {              // [pts_sphere, dists] = pcreg_mex('getLocalPoints', pts, R, c, min_points, max_points)
        if (nrhs != 6 || mxGetN(prhs[1]) != 3) usage = "getLocalPoints: pts (N x 3), R, c (1 x 3), min_points, max_points";
        else {
            // a single cloud or centre: MATLAB's element-wise single arithmetic (getLocalPoints.m:8-25) and single results
            const bool ps = mxIsSingle(prhs[1]), cs = mxIsSingle(prhs[3]);
            const int mode = cs ? 1 : (ps ? 2 : 0);
            const int N = (int)mxGetM(prhs[1]);
            mxArray* wide = mxCreateNumericMatrix(N > 0 ? N : 1, 3, mxDOUBLE_CLASS, mxREAL);
            if (ps) { const float* f = (const float*)mxGetData(prhs[1]); for (size_t k = 0; k < (size_t)N * 3; ++k) mxGetPr(wide)[k] = (double)f[k]; }
            else if (N > 0) memcpy(mxGetPr(wide), mxGetPr(prhs[1]), sizeof(double) * 3 * (size_t)N);
            double c[3];
            for (int k = 0; k < 3; ++k) c[k] = cs ? (double)((const float*)mxGetData(prhs[3]))[k] : mxGetPr(prhs[3])[k];
            mxArray* out = mxCreateNumericMatrix(N > 0 ? N : 1, 3, mxDOUBLE_CLASS, mxREAL);
            mxArray* dd = mxCreateNumericMatrix(N > 0 ? N : 1, 1, mxDOUBLE_CLASS, mxREAL);
            int n = 0;
            rc = pcreg_get_local_points(mxGetPr(wide), N, N > 0 ? N : 1, mxGetScalar(prhs[2]), c, mxGetScalar(prhs[4]), mxGetScalar(prhs[5]), mode,
                                        mxGetPr(out), mxGetPr(dd), &n);
            if (rc == PCREG_OK) {
                const mxClassID cls = mode ? mxSINGLE_CLASS : mxDOUBLE_CLASS;
                plhs[0] = n ? mxCreateNumericMatrix(n, 3, cls, mxREAL) : mxCreateDoubleMatrix(0, 0, mxREAL);          // [] as in the reference
                if (nlhs > 1) plhs[1] = n ? mxCreateNumericMatrix(n, 1, cls, mxREAL) : mxCreateDoubleMatrix(0, 0, mxREAL);
                if (n && mode) {
                    float* o = (float*)mxGetData(plhs[0]); for (size_t k = 0; k < (size_t)n * 3; ++k) o[k] = (float)mxGetPr(out)[k];
                    if (nlhs > 1) { float* q = (float*)mxGetData(plhs[1]); for (int k = 0; k < n; ++k) q[k] = (float)mxGetPr(dd)[k]; }
                } else if (n) {
                    memcpy(mxGetPr(plhs[0]), mxGetPr(out), sizeof(double) * 3 * (size_t)n);
                    if (nlhs > 1) memcpy(mxGetPr(plhs[1]), mxGetPr(dd), sizeof(double) * (size_t)n);
                }
            }
            mxDestroyArray(wide); mxDestroyArray(out); mxDestroyArray(dd);
        }
    } else if (!strcmp(cmd, "modelCreate")) {                 // h = pcreg_mex('modelCreate', single(model)): uploaded and prepared ONCE
        if (nrhs != 2 || !mxIsSingle(prhs[1]) || mxGetN(prhs[1]) != 3) usage = "modelCreate: model (single M x 3)";
        else {
            int M = (int)mxGetM(prhs[1]);
            pcreg_model* h = nullptr;
            rc = pcreg_model_create((const float*)mxGetData(prhs[1]), M, M > 0 ? M : 1, &h);
            if (rc == PCREG_OK) { plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)h; }
        }
    } else if (!strcmp(cmd, "modelMatchPoints")) {            // pairs = pcreg_mex('modelMatchPoints', h, single(surface), thrAbs, maxRatio, unique)
        if (nrhs != 6 || !mxIsUint64(prhs[1]) || !mxIsSingle(prhs[2]) || mxGetN(prhs[2]) != 3) usage = "modelMatchPoints: handle (uint64), surface (single Q x 3), thrAbs, maxRatio, unique";
        else {
            pcreg_model* h = (pcreg_model*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]);
            int Q = (int)mxGetM(prhs[2]);
            mxArray* buf = mxCreateNumericMatrix(2, Q > 0 ? Q : 1, mxUINT32_CLASS, mxREAL);
            int P = 0;
            rc = pcreg_model_match_points_f32(h, (const float*)mxGetData(prhs[2]), Q, Q > 0 ? Q : 1, (float)mxGetScalar(prhs[3]), (float)mxGetScalar(prhs[4]),
                                              (int)mxGetScalar(prhs[5]), (uint32_t*)mxGetData(buf), &P);
            if (rc == PCREG_OK) {
                plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                for (int k = 0; k < P; ++k) { dst[k] = src[2 * k]; dst[k + P] = src[2 * k + 1]; }
            }
            mxDestroyArray(buf);
        }
    } else if (!strcmp(cmd, "descCreate")) {                  // h = pcreg_mex('descCreate', desc): an n x D double descriptor set, uploaded ONCE
        if (nrhs != 2 || !mxIsDouble(prhs[1])) usage = "descCreate: desc (double n x D)";
        else {
            const int n = (int)mxGetM(prhs[1]), D = (int)mxGetN(prhs[1]);
            pcreg_desc_set* h = nullptr;
            rc = pcreg_desc_set_create(mxGetPr(prhs[1]), n, n > 0 ? n : 1, D > 0 ? D : 1, &h);
            if (rc == PCREG_OK) { plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)h; }
        }
    } else if (!strcmp(cmd, "getMatchesOnSet")) {             // matches = pcreg_mex('getMatchesOnSet', hSurface, hModel, int32(rows) | [], par)
        // = getMatches(descSurface, descModel(rows, :), par) on the resident sets (rows 1-based, ascending; []: the whole model set)
        if (nrhs != 5 || !mxIsUint64(prhs[1]) || !mxIsUint64(prhs[2]) || !(mxIsInt32(prhs[3]) || mxGetM(prhs[3]) * mxGetN(prhs[3]) == 0))
            usage = "getMatchesOnSet: hSurface (uint64), hModel (uint64), int32 rows or [], par";
        else {
            const mxArray* p = prhs[4];
            pcreg_match_opts o;
            o.metric = field_is(p, "Metric", "SAD") ? PCREG_METRIC_SAD : PCREG_METRIC_SSD;
            o.matchThreshold = field(p, "MatchThreshold", 1.0); o.maxRatio = field(p, "MaxRatio", 0.6);
            o.unique = (int)field(p, "Unique", 0); o.prenormalized = 0;
            o.unnormalize = (int)field(p, "UNNORMALIZE", 0); o.norm_factor = field(p, "norm_factor", 0.0);
            o.change_metric = (int)field(p, "CHANGE_METRIC", 0); o.metric_factor = field(p, "metric_factor", 1.0);
            pcreg_desc_set* hS = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]);
            pcreg_desc_set* hM = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[2]);
            int Q = 0, D = 0;
            rc = pcreg_desc_set_size(hS, &Q, &D);
            const size_t nr = mxGetM(prhs[3]) * mxGetN(prhs[3]);
            mxArray* r0 = mxCreateNumericMatrix(nr > 0 ? nr : 1, 1, mxINT32_CLASS, mxREAL);          // 0-based copy of the row list
            if (nr > 0) { const int32_t* r1 = (const int32_t*)mxGetData(prhs[3]); int32_t* z = (int32_t*)mxGetData(r0); for (size_t k = 0; k < nr; ++k) z[k] = r1[k] - 1; }
            mxArray* buf = mxCreateNumericMatrix(2, Q > 0 ? Q : 1, mxUINT32_CLASS, mxREAL);
            int P = 0;
            if (rc == PCREG_OK) rc = pcreg_get_matches_on_sets(hS, hM, nr > 0 ? (const int32_t*)mxGetData(r0) : nullptr, (int)nr, &o, (uint32_t*)mxGetData(buf), nullptr, &P);
            if (rc == PCREG_OK) {
                plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                for (int k = 0; k < P; ++k) { dst[k] = src[2 * k]; dst[k + P] = src[2 * k + 1]; }
            }
            mxDestroyArray(buf); mxDestroyArray(r0);
        }
    } else if (!strcmp(cmd, "getMatchesSegmentedOnSet")) {    // [pairs, nPairs] = pcreg_mex('getMatchesSegmentedOnSet', hSurface, hModel, int32(rows), int32(segOff), par)
        // getMatchesSegmented on the resident sets: rows / segOff / outputs as there.  matlab/getMatchesSegmentedOnSet.m
        if (nrhs != 6 || !mxIsUint64(prhs[1]) || !mxIsUint64(prhs[2]) || !mxIsInt32(prhs[3]) || !mxIsInt32(prhs[4]))
            usage = "getMatchesSegmentedOnSet: hSurface (uint64), hModel (uint64), int32 rows, int32 segOff, par";
        else {
            const mxArray* p = prhs[5];
            pcreg_match_opts o;
            o.metric = field_is(p, "Metric", "SAD") ? PCREG_METRIC_SAD : PCREG_METRIC_SSD;
            o.matchThreshold = field(p, "MatchThreshold", 1.0); o.maxRatio = field(p, "MaxRatio", 0.6);
            o.unique = (int)field(p, "Unique", 0); o.prenormalized = 0;
            o.unnormalize = (int)field(p, "UNNORMALIZE", 0); o.norm_factor = field(p, "norm_factor", 0.0);
            o.change_metric = (int)field(p, "CHANGE_METRIC", 0); o.metric_factor = field(p, "metric_factor", 1.0);
            pcreg_desc_set* hS = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]);
            pcreg_desc_set* hM = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[2]);
            int Q = 0, D = 0;
            rc = pcreg_desc_set_size(hS, &Q, &D);
            const int S = (int)(mxGetM(prhs[4]) * mxGetN(prhs[4])) - 1;
            const size_t tot = mxGetM(prhs[3]) * mxGetN(prhs[3]);
            const int32_t* off = (const int32_t*)mxGetData(prhs[4]);
            if (S < 0 || off[0] != 0 || (size_t)off[S] != tot) usage = "getMatchesSegmentedOnSet: segOff must run from 0 to numel(rows)";
            else if (rc == PCREG_OK) {
                mxArray* r0 = mxCreateNumericMatrix(tot > 0 ? tot : 1, 1, mxINT32_CLASS, mxREAL);
                mxArray* buf = mxCreateNumericMatrix(2, (size_t)(S > 0 ? S : 1) * (Q > 0 ? Q : 1), mxUINT32_CLASS, mxREAL);
                mxArray* np = mxCreateNumericMatrix(S > 0 ? S : 1, 1, mxINT32_CLASS, mxREAL);
                int32_t* rz = (int32_t*)mxGetData(r0); const int32_t* r1 = (const int32_t*)mxGetData(prhs[3]);
                for (size_t k = 0; k < tot; ++k) rz[k] = r1[k] - 1;
                rc = pcreg_get_matches_segmented_on_sets(hS, hM, rz, off, S, &o, (uint32_t*)mxGetData(buf), (int32_t*)mxGetData(np));
                if (rc == PCREG_OK) {
                    const int32_t* n = (const int32_t*)mxGetData(np);
                    size_t P = 0; for (int z = 0; z < S; ++z) P += (size_t)n[z];
                    plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                    const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                    size_t k = 0;
                    for (int z = 0; z < S; ++z)
                        for (int e = 0; e < n[z]; ++e, ++k) { dst[k] = src[((size_t)z * Q + e) * 2]; dst[k + P] = src[((size_t)z * Q + e) * 2 + 1]; }
                    if (nlhs > 1) { plhs[1] = mxCreateNumericMatrix(S, 1, mxINT32_CLASS, mxREAL); memcpy(mxGetData(plhs[1]), n, (size_t)S * 4); }
                }
                mxDestroyArray(r0); mxDestroyArray(buf); mxDestroyArray(np);
            }
        }
    } else if (!strcmp(cmd, "sphereCounts")) {                // counts = pcreg_mex('sphereCounts', featModel, centres, R): completeExperimentFast.m:52-64
        if (nrhs != 4 || !mxIsDouble(prhs[1]) || !mxIsDouble(prhs[2]) || mxGetN(prhs[1]) != 3 || mxGetN(prhs[2]) != 3) usage = "sphereCounts: featModel (n x 3 double), centres (S x 3 double), R";
        else {
            const int VM = (int)mxGetM(prhs[1]), S = (int)mxGetM(prhs[2]);
            mxArray* cnt = mxCreateNumericMatrix(S > 0 ? S : 1, 1, mxINT32_CLASS, mxREAL);
            rc = pcreg_sphere_counts(mxGetPr(prhs[1]), VM, VM > 0 ? VM : 1, mxGetPr(prhs[2]), S, S > 0 ? S : 1, mxGetScalar(prhs[3]), (int32_t*)mxGetData(cnt));
            if (rc == PCREG_OK) {
                plhs[0] = mxCreateDoubleMatrix(S, S ? 1 : 0, mxREAL);
                const int32_t* c = (const int32_t*)mxGetData(cnt);
                for (int i = 0; i < S; ++i) mxGetPr(plhs[0])[i] = (double)c[i];
            }
            mxDestroyArray(cnt);
        }
    } else if (!strcmp(cmd, "sphereSweep")) {
        // [modelRows, pairs, nPairs, trial, T, numSuccess, maxInliers, failed] = pcreg_mex('sphereSweep', hSurface, hModel, featSurface, featModel,
        //     centres, int32(numDesc), R_desc, par, putativeThresh, ransacCoef, seed)
        // completeExperimentFast.m:101-224 for the spheres kept after sphereCounts.  modelRows: the spheres' row lists back to back (1-based);
        // pairs: sum(nPairs) x 2 uint32, sphere after sphere; trial: the registered spheres (1-based); T: 4 x 4 x numel(trial).  matlab/sphereSweep.m
        if (nrhs != 12 || !mxIsUint64(prhs[1]) || !mxIsUint64(prhs[2]) || !mxIsDouble(prhs[3]) || !mxIsDouble(prhs[4]) || !mxIsDouble(prhs[5]) || !mxIsInt32(prhs[6]))
            usage = "sphereSweep: hSurface, hModel (uint64), featSurface, featModel, centres (double n x 3), int32 numDesc, R_desc, par, putativeThresh, ransacCoef, seed";
        else {
            const mxArray* p = prhs[8];
            pcreg_match_opts o;
            o.metric = field_is(p, "Metric", "SAD") ? PCREG_METRIC_SAD : PCREG_METRIC_SSD;
            o.matchThreshold = field(p, "MatchThreshold", 1.0); o.maxRatio = field(p, "MaxRatio", 0.6);
            o.unique = (int)field(p, "Unique", 0); o.prenormalized = 0;
            o.unnormalize = (int)field(p, "UNNORMALIZE", 0); o.norm_factor = field(p, "norm_factor", 0.0);
            o.change_metric = (int)field(p, "CHANGE_METRIC", 0); o.metric_factor = field(p, "metric_factor", 1.0);
            const mxArray* c = prhs[10];
            pcreg_ransac_opts ro;
            ro.minPtNum = (int)field(c, "minPtNum", 3); ro.iterNum = (int)field(c, "iterNum", 1000);
            ro.thDist = field(c, "thDist", 0.5); ro.thInlrRatio = field(c, "thInlrRatio", 0.1);
            ro.REFINE = (int)field(c, "REFINE", 1); ro.VERBOSE = 0;
            ro.seed = (uint64_t)mxGetScalar(prhs[11]);
            pcreg_desc_set* hS = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]);
            pcreg_desc_set* hM = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[2]);
            int Q = 0, VM = 0, D = 0, D2 = 0;
            rc = pcreg_desc_set_size(hS, &Q, &D);
            if (rc == PCREG_OK) rc = pcreg_desc_set_size(hM, &VM, &D2);
            const int S = (int)mxGetM(prhs[5]);
            const int32_t* nd = (const int32_t*)mxGetData(prhs[6]);
            if (rc == PCREG_OK && ((int)mxGetM(prhs[3]) != Q || (int)mxGetM(prhs[4]) != VM || mxGetN(prhs[3]) != 3 || mxGetN(prhs[4]) != 3 || (S > 0 && mxGetN(prhs[5]) != 3) ||
                                   (int)(mxGetM(prhs[6]) * mxGetN(prhs[6])) != S))
                usage = "sphereSweep: featSurface / featModel must have the rows of their descriptor sets, centres S x 3, numDesc S entries";
            else if (rc == PCREG_OK) {
                size_t tot = 0; for (int i = 0; i < S; ++i) tot += (size_t)(nd[i] > 0 ? nd[i] : 0);
                const size_t s1 = (size_t)(S > 0 ? S : 1);
                mxArray* rows = mxCreateNumericMatrix(tot > 0 ? tot : 1, 1, mxINT32_CLASS, mxREAL);
                mxArray* buf = mxCreateNumericMatrix(2, s1 * (Q > 0 ? Q : 1), mxUINT32_CLASS, mxREAL);
                mxArray* np = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL); mxArray* tr = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL);
                mxArray* Tb = mxCreateDoubleMatrix(16, s1, mxREAL);
                mxArray* ns = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL); mxArray* mi = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL);
                mxArray* fl = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL);
                int nt = 0;
                rc = pcreg_sphere_sweep(hS, hM, mxGetPr(prhs[3]), Q > 0 ? Q : 1, mxGetPr(prhs[4]), VM > 0 ? VM : 1, mxGetPr(prhs[5]), S, S > 0 ? S : 1, nd, mxGetScalar(prhs[7]),
                                        &o, (int)mxGetScalar(prhs[9]), &ro, (int32_t*)mxGetData(rows), (uint32_t*)mxGetData(buf), (int32_t*)mxGetData(np), (int32_t*)mxGetData(tr),
                                        &nt, mxGetPr(Tb), (int32_t*)mxGetData(ns), (int32_t*)mxGetData(mi), (int32_t*)mxGetData(fl));
                if (rc == PCREG_OK) {
                    const int32_t* n = (const int32_t*)mxGetData(np);
                    size_t P = 0; for (int z = 0; z < S; ++z) P += (size_t)n[z];
                    plhs[0] = mxCreateDoubleMatrix(tot, tot ? 1 : 0, mxREAL);
                    { const int32_t* r = (const int32_t*)mxGetData(rows); for (size_t k = 0; k < tot; ++k) mxGetPr(plhs[0])[k] = (double)r[k] + 1.0; }
                    auto out = [&](int k, mxArray* a) { if (nlhs > k) plhs[k] = a; else mxDestroyArray(a); };
                    {
                        mxArray* pr = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                        const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(pr);
                        size_t k = 0;
                        for (int z = 0; z < S; ++z)
                            for (int e = 0; e < n[z]; ++e, ++k) { dst[k] = src[((size_t)z * Q + e) * 2]; dst[k + P] = src[((size_t)z * Q + e) * 2 + 1]; }
                        out(1, pr);
                    }
                    auto col = [&](const int32_t* v, int len, double add) { mxArray* a = mxCreateDoubleMatrix(len, len ? 1 : 0, mxREAL); for (int i = 0; i < len; ++i) mxGetPr(a)[i] = (double)v[i] + add; return a; };
                    out(2, col(n, S, 0.0));
                    out(3, col((const int32_t*)mxGetData(tr), nt, 1.0));
                    { mwSize dims[3] = {4, 4, (mwSize)nt}; mxArray* T3 = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL); if (nt) memcpy(mxGetPr(T3), mxGetPr(Tb), (size_t)nt * 128); out(4, T3); }
                    out(5, col((const int32_t*)mxGetData(ns), nt, 0.0));
                    out(6, col((const int32_t*)mxGetData(mi), nt, 0.0));
                    out(7, col((const int32_t*)mxGetData(fl), nt, 0.0));
                }
                mxDestroyArray(rows); mxDestroyArray(buf); mxDestroyArray(np); mxDestroyArray(tr); mxDestroyArray(Tb); mxDestroyArray(ns); mxDestroyArray(mi); mxDestroyArray(fl);
            }
        }
    } else if (!strcmp(cmd, "sphereModelCreate")) {           // [h, modelRows] = pcreg_mex('sphereModelCreate', hModel, featModel, centres, int32(numDesc), R_desc)
        // the model side of sphereSweep in a handle (one model, many surfaces); modelRows: the spheres' row lists back to back (1-based)
        if (nrhs != 6 || !mxIsUint64(prhs[1]) || !mxIsDouble(prhs[2]) || !mxIsDouble(prhs[3]) || !mxIsInt32(prhs[4]) || mxGetN(prhs[2]) != 3)
            usage = "sphereModelCreate: hModel (uint64), featModel (n x 3 double), centres (S x 3 double), int32 numDesc, R_desc";
        else {
            pcreg_desc_set* hM = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]);
            int VM = 0, D = 0;
            rc = pcreg_desc_set_size(hM, &VM, &D);
            const int S = (int)mxGetM(prhs[3]);
            const int32_t* nd = (const int32_t*)mxGetData(prhs[4]);
            if (rc == PCREG_OK && ((int)mxGetM(prhs[2]) != VM || (S > 0 && mxGetN(prhs[3]) != 3) || (int)(mxGetM(prhs[4]) * mxGetN(prhs[4])) != S))
                usage = "sphereModelCreate: featModel must have the rows of the model set, centres S x 3, numDesc S entries";
            else if (rc == PCREG_OK) {
                size_t tot = 0; for (int i = 0; i < S; ++i) tot += (size_t)(nd[i] > 0 ? nd[i] : 0);
                mxArray* rows = mxCreateNumericMatrix(tot > 0 ? tot : 1, 1, mxINT32_CLASS, mxREAL);
                pcreg_sphere_model* h = nullptr;
                rc = pcreg_sphere_model_create(hM, mxGetPr(prhs[2]), VM > 0 ? VM : 1, mxGetPr(prhs[3]), S, S > 0 ? S : 1, nd, mxGetScalar(prhs[5]), (int32_t*)mxGetData(rows), &h);
                if (rc == PCREG_OK) {
                    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)h;
                    if (nlhs > 1) {
                        plhs[1] = mxCreateDoubleMatrix(tot, tot ? 1 : 0, mxREAL);
                        const int32_t* r = (const int32_t*)mxGetData(rows);
                        for (size_t k = 0; k < tot; ++k) mxGetPr(plhs[1])[k] = (double)r[k] + 1.0;
                    }
                }
                mxDestroyArray(rows);
            }
        }
    } else if (!strcmp(cmd, "sphereSweepOnModel")) {
        // [pairs, nPairs, trial, T, numSuccess, maxInliers, failed] = pcreg_mex('sphereSweepOnModel', hSphereModel, hSurface, featSurface, S, par,
        //     putativeThresh, ransacCoef, seed)            (S = the number of spheres of the model handle; outputs as 'sphereSweep' without the rows)
        if (nrhs != 9 || !mxIsUint64(prhs[1]) || !mxIsUint64(prhs[2]) || !mxIsDouble(prhs[3]) || mxGetN(prhs[3]) != 3)
            usage = "sphereSweepOnModel: hSphereModel, hSurface (uint64), featSurface (n x 3 double), S, par, putativeThresh, ransacCoef, seed";
        else {
            const mxArray* p = prhs[5];
            pcreg_match_opts o;
            o.metric = field_is(p, "Metric", "SAD") ? PCREG_METRIC_SAD : PCREG_METRIC_SSD;
            o.matchThreshold = field(p, "MatchThreshold", 1.0); o.maxRatio = field(p, "MaxRatio", 0.6);
            o.unique = (int)field(p, "Unique", 0); o.prenormalized = 0;
            o.unnormalize = (int)field(p, "UNNORMALIZE", 0); o.norm_factor = field(p, "norm_factor", 0.0);
            o.change_metric = (int)field(p, "CHANGE_METRIC", 0); o.metric_factor = field(p, "metric_factor", 1.0);
            const mxArray* c = prhs[7];
            pcreg_ransac_opts ro;
            ro.minPtNum = (int)field(c, "minPtNum", 3); ro.iterNum = (int)field(c, "iterNum", 1000);
            ro.thDist = field(c, "thDist", 0.5); ro.thInlrRatio = field(c, "thInlrRatio", 0.1);
            ro.REFINE = (int)field(c, "REFINE", 1); ro.VERBOSE = 0;
            ro.seed = (uint64_t)mxGetScalar(prhs[8]);
            pcreg_sphere_model* sm = (pcreg_sphere_model*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]);
            pcreg_desc_set* hS = (pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[2]);
            int Q = 0, D = 0;
            rc = pcreg_desc_set_size(hS, &Q, &D);
            const int S = (int)mxGetScalar(prhs[4]);
            if (rc == PCREG_OK && ((int)mxGetM(prhs[3]) != Q || S < 0)) usage = "sphereSweepOnModel: featSurface must have the rows of the surface set";
            else if (rc == PCREG_OK) {
                const size_t s1 = (size_t)(S > 0 ? S : 1);
                mxArray* buf = mxCreateNumericMatrix(2, s1 * (Q > 0 ? Q : 1), mxUINT32_CLASS, mxREAL);
                mxArray* np = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL); mxArray* tr = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL);
                mxArray* Tb = mxCreateDoubleMatrix(16, s1, mxREAL);
                mxArray* ns = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL); mxArray* mi = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL);
                mxArray* fl = mxCreateNumericMatrix(s1, 1, mxINT32_CLASS, mxREAL);
                int nt = 0;
                rc = pcreg_sphere_sweep_on_model(sm, hS, mxGetPr(prhs[3]), Q > 0 ? Q : 1, &o, (int)mxGetScalar(prhs[6]), &ro, (uint32_t*)mxGetData(buf), (int32_t*)mxGetData(np),
                                                 (int32_t*)mxGetData(tr), &nt, mxGetPr(Tb), (int32_t*)mxGetData(ns), (int32_t*)mxGetData(mi), (int32_t*)mxGetData(fl));
                if (rc == PCREG_OK) {
                    const int32_t* n = (const int32_t*)mxGetData(np);
                    size_t P = 0; for (int z = 0; z < S; ++z) P += (size_t)n[z];
                    plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                    {
                        const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                        size_t k = 0;
                        for (int z = 0; z < S; ++z)
                            for (int e = 0; e < n[z]; ++e, ++k) { dst[k] = src[((size_t)z * Q + e) * 2]; dst[k + P] = src[((size_t)z * Q + e) * 2 + 1]; }
                    }
                    auto out = [&](int k, mxArray* a) { if (nlhs > k) plhs[k] = a; else mxDestroyArray(a); };
                    auto col = [&](const int32_t* v, int len, double add) { mxArray* a = mxCreateDoubleMatrix(len, len ? 1 : 0, mxREAL); for (int i = 0; i < len; ++i) mxGetPr(a)[i] = (double)v[i] + add; return a; };
                    out(1, col(n, S, 0.0));
                    out(2, col((const int32_t*)mxGetData(tr), nt, 1.0));
                    { mwSize dims[3] = {4, 4, (mwSize)nt}; mxArray* T3 = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL); if (nt) memcpy(mxGetPr(T3), mxGetPr(Tb), (size_t)nt * 128); out(3, T3); }
                    out(4, col((const int32_t*)mxGetData(ns), nt, 0.0));
                    out(5, col((const int32_t*)mxGetData(mi), nt, 0.0));
                    out(6, col((const int32_t*)mxGetData(fl), nt, 0.0));
                }
                mxDestroyArray(buf); mxDestroyArray(np); mxDestroyArray(tr); mxDestroyArray(Tb); mxDestroyArray(ns); mxDestroyArray(mi); mxDestroyArray(fl);
            }
        }
    } else if (!strcmp(cmd, "sphereModelDestroy")) {
        if (nrhs != 2 || !mxIsUint64(prhs[1])) usage = "sphereModelDestroy: handle (uint64)";
        else rc = pcreg_sphere_model_destroy((pcreg_sphere_model*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]));
    } else if (!strcmp(cmd, "descDestroy")) {
        if (nrhs != 2 || !mxIsUint64(prhs[1])) usage = "descDestroy: handle (uint64)";
        else rc = pcreg_desc_set_destroy((pcreg_desc_set*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]));
    } else if (!strcmp(cmd, "modelDestroy")) {
        if (nrhs != 2 || !mxIsUint64(prhs[1])) usage = "modelDestroy: handle (uint64)";
        else rc = pcreg_model_destroy((pcreg_model*)(uintptr_t)*(const uint64_t*)mxGetData(prhs[1]));
    } else if (!strcmp(cmd, "setDevice")) {                   // one GPU per parfor / spmd worker: pcreg_mex('setDevice', labindex - 1)
        if (nrhs != 2) usage = "setDevice: ordinal";
        else rc = pcreg_set_device((int)mxGetScalar(prhs[1]));
    } else if (!strcmp(cmd, "commId")) {                      // id = pcreg_mex('commId') on worker 1; labBroadcast it
        if (nrhs != 1) usage = "commId: no arguments";
        else {
            pcreg_comm_id id;
            rc = pcreg_comm_get_unique_id(&id);
            if (rc == PCREG_OK) { plhs[0] = mxCreateNumericMatrix(1, sizeof id, mxUINT8_CLASS, mxREAL); memcpy(mxGetData(plhs[0]), &id, sizeof id); }
        }
    } else if (!strcmp(cmd, "commInit")) {                    // pcreg_mex('commInit', rank, world, id)
        if (nrhs != 4 || mxGetM(prhs[3]) * mxGetN(prhs[3]) != sizeof(pcreg_comm_id) || !mxIsUint8(prhs[3])) usage = "commInit: rank, world, id (1 x 128 uint8)";
        else { pcreg_comm_id id; memcpy(&id, mxGetData(prhs[3]), sizeof id); rc = pcreg_comm_init((int)mxGetScalar(prhs[1]), (int)mxGetScalar(prhs[2]), &id); }
    } else if (!strcmp(cmd, "commDestroy")) {
        rc = pcreg_comm_destroy();
    } else if (!strcmp(cmd, "matchPointsSharded")) {          // pairs = pcreg_mex('matchPointsSharded', surface, modelRows, m_lo, M_total, thrAbs, maxRatio, unique)
        if (nrhs != 8 || !mxIsSingle(prhs[1]) || !mxIsSingle(prhs[2])) usage = "matchPointsSharded: surface (single Q x 3), model rows (single M_local x 3), m_lo, M_total, thrAbs, maxRatio, unique";
        else {
            int Q = (int)mxGetM(prhs[1]), Ml = (int)mxGetM(prhs[2]);
            mxArray* buf = mxCreateNumericMatrix(2, Q > 0 ? Q : 1, mxUINT32_CLASS, mxREAL);
            int P = 0;
            rc = pcreg_match_points_sharded_f32((const float*)mxGetData(prhs[1]), Q, Q, (const float*)mxGetData(prhs[2]), Ml, Ml > 0 ? Ml : 1,
                                                (int)mxGetScalar(prhs[3]), (int)mxGetScalar(prhs[4]), (float)mxGetScalar(prhs[5]), (float)mxGetScalar(prhs[6]),
                                                (int)mxGetScalar(prhs[7]), (uint32_t*)mxGetData(buf), &P);
            if (rc == PCREG_OK) {
                plhs[0] = mxCreateNumericMatrix(P, 2, mxUINT32_CLASS, mxREAL);
                const uint32_t* src = (const uint32_t*)mxGetData(buf); uint32_t* dst = (uint32_t*)mxGetData(plhs[0]);
                for (int k = 0; k < P; ++k) { dst[k] = src[2 * k]; dst[k + P] = src[2 * k + 1]; }
            }
            mxDestroyArray(buf);
        }
    } else if (!strcmp(cmd, "ransacSharded")) {               // [T, inlierIdx, numSuccess, maxInliers, failed] = pcreg_mex('ransacSharded', pts1, pts2, coef, seed)
        if (nrhs != 5) usage = "ransacSharded: pts1, pts2, coef, seed";
        else {
            const mxArray* c = prhs[3];
            pcreg_ransac_opts o;
            o.minPtNum = (int)field(c, "minPtNum", 3); o.iterNum = (int)field(c, "iterNum", 1000);
            o.thDist = field(c, "thDist", 0.5); o.thInlrRatio = field(c, "thInlrRatio", 0.1);
            o.REFINE = (int)field(c, "REFINE", 1); o.VERBOSE = 0; o.seed = (uint64_t)mxGetScalar(prhs[4]);
            int n = (int)mxGetM(prhs[1]);
            mxArray* inl = mxCreateNumericMatrix(n > 0 ? n : 1, 1, mxINT32_CLASS, mxREAL);
            double T[16]; int ni = 0, ns = 0, mi = 0, fl = 1;
            rc = pcreg_ransac_sharded(mxGetPr(prhs[1]), mxGetPr(prhs[2]), n, n, &o, T, (int32_t*)mxGetData(inl), &ni, &ns, &mi, &fl);
            if (rc == PCREG_OK) {
                plhs[0] = fl ? mxCreateDoubleMatrix(0, 0, mxREAL) : mxCreateDoubleMatrix(4, 4, mxREAL);
                if (!fl) memcpy(mxGetPr(plhs[0]), T, sizeof T);
                if (nlhs > 1) {
                    plhs[1] = mxCreateDoubleMatrix(fl ? 0 : ni, fl ? 0 : 1, mxREAL);
                    const int32_t* src = (const int32_t*)mxGetData(inl);
                    for (int k = 0; k < (fl ? 0 : ni); ++k) mxGetPr(plhs[1])[k] = (double)src[k];
                }
                if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(ns);
                if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(mi);
                if (nlhs > 4) plhs[4] = mxCreateDoubleScalar(fl);
            }
            mxDestroyArray(inl);
        }
    } else {
        usage = "unknown command";
    }
    if (usage) mexErrMsgIdAndTxt("pcreg:usage", "%s", usage);
    if (rc != PCREG_OK) mexErrMsgIdAndTxt("pcreg:hip", "%s", pcreg_last_error());
}
#endif  // __has_include("mex.h")
