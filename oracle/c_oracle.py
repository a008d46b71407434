"""ctypes binding of oracle/libpcreg_oracle.so (the plain-C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by pcreg_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpcreg_oracle.so")


class RansacOpts(C.Structure):
    _fields_ = [("minPtNum", C.c_int32), ("iterNum", C.c_int32), ("thDist", C.c_double),
                ("thInlrRatio", C.c_double), ("REFINE", C.c_int32)]


class MatchOpts(C.Structure):
    _fields_ = [("metric", C.c_int32), ("matchThreshold", C.c_double), ("maxRatio", C.c_double),
                ("unique", C.c_int32), ("prenormalized", C.c_int32), ("unnormalize", C.c_int32),
                ("norm_factor", C.c_double), ("change_metric", C.c_int32), ("metric_factor", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pcreg_oracle.c")
    stale = (not os.path.exists(_SO)) or (os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_rank_nx3.restype = C.c_int
        _lib.orc_match_points_f32.restype = C.c_int
        _lib.orc_match_features.restype = C.c_int
        _lib.orc_get_matches.restype = C.c_int
    return _lib


def _f(a):  # column-major double copy
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def estimateTransform(pts1, pts2):
    p1, p2 = _f(pts1), _f(pts2)
    T = np.zeros(16)
    empty = C.c_int(0)
    lib().orc_estimate_transform(_p(p1), _p(p2), C.c_int(p1.shape[0]), C.c_int(p1.shape[0]), _p(T), C.byref(empty))
    return None if empty.value else T.reshape(4, 4, order="F").copy()


def rank_nx3(pts):
    p = _f(pts)
    return lib().orc_rank_nx3(_p(p), C.c_int(p.shape[0]), C.c_int(p.shape[0]))


def calcDists(T, pts1, pts2):
    p1, p2 = _f(pts1), _f(pts2)
    Tf = np.asarray(T, dtype=np.float64).reshape(-1, order="F").copy()
    d = np.zeros(p1.shape[0])
    lib().orc_calc_dists(_p(Tf), _p(p1), _p(p2), C.c_int(p1.shape[0]), C.c_int(p1.shape[0]), _p(d))
    return d


def sample_table(n, iterNum, minPtNum, seed):
    out = np.zeros((iterNum, minPtNum), dtype=np.int32)
    lib().orc_sample_table(C.c_int(n), C.c_int(iterNum), C.c_int(minPtNum), C.c_uint64(seed), _p(out, C.c_int32))
    return out


def ransac(pts1, pts2, coef: dict, sample_idx=None, seed=0) -> dict:
    p1, p2 = _f(pts1), _f(pts2)
    n = p1.shape[0]
    o = RansacOpts(int(coef["minPtNum"]), int(coef["iterNum"]), float(coef["thDist"]),
                   float(coef["thInlrRatio"]), int(bool(coef["REFINE"])))
    if sample_idx is None:
        sample_idx = sample_table(n, o.iterNum, o.minPtNum, seed)
    si = np.ascontiguousarray(sample_idx, dtype=np.int32)
    T = np.zeros(16)
    inl = np.zeros(max(n, 1), dtype=np.int32)
    ni, ns, mi, fl = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    it1 = np.zeros(o.iterNum, dtype=np.int32)
    it2 = np.zeros(o.iterNum, dtype=np.int32)
    lib().orc_ransac(_p(p1), _p(p2), C.c_int(n), C.c_int(n), C.byref(o), _p(si, C.c_int32), _p(T),
                     _p(inl, C.c_int32), C.byref(ni), C.byref(ns), C.byref(mi), C.byref(fl),
                     _p(it1, C.c_int32), _p(it2, C.c_int32))
    return dict(T=None if fl.value else T.reshape(4, 4, order="F").copy(),
                inlierIdx=inl[:ni.value].astype(np.int64), numSuccess=ns.value, maxInliers=mi.value,
                ratio=(100.0 * mi.value / n) if n else 0.0, failed=bool(fl.value),
                inlrNum=it1.astype(np.int64), inlrNum_refined=it2.astype(np.int64))


def knn2_points_f32(q, m, nthreads=0):
    qf = np.asfortranarray(np.asarray(q, dtype=np.float32))
    mf = np.asfortranarray(np.asarray(m, dtype=np.float32))
    Q, M = qf.shape[0], mf.shape[0]
    idx = np.zeros((Q, 2), dtype=np.int32)
    dist = np.zeros((Q, 2), dtype=np.float32)
    lib().orc_knn2_points_f32(_p(qf, C.c_float), C.c_int(Q), C.c_int(Q), _p(mf, C.c_float), C.c_int(M),
                              C.c_int(M), _p(idx, C.c_int32), _p(dist, C.c_float), C.c_int(nthreads))
    return idx, dist


def match_points_f32(q, m, thr_abs, max_ratio, unique=True, nthreads=0):
    qf = np.asfortranarray(np.asarray(q, dtype=np.float32))
    mf = np.asfortranarray(np.asarray(m, dtype=np.float32))
    Q, M = qf.shape[0], mf.shape[0]
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    P = lib().orc_match_points_f32(_p(qf, C.c_float), C.c_int(Q), C.c_int(Q), _p(mf, C.c_float), C.c_int(M),
                                   C.c_int(M), C.c_float(thr_abs), C.c_float(max_ratio), C.c_int(int(unique)),
                                   _p(pairs, C.c_uint32), C.c_int(nthreads))
    return pairs[:P].copy()


def _mopts(par: dict) -> MatchOpts:
    metric = str(par.get("Metric", "SSD")).upper()
    return MatchOpts(0 if metric == "SAD" else 1, float(par.get("MatchThreshold", 10.0)),
                     float(par.get("MaxRatio", 0.6)), int(bool(par.get("Unique", False))),
                     int(bool(par.get("Prenormalized", False))), int(bool(par.get("UNNORMALIZE", False))),
                     float(par.get("norm_factor", 0.0)), int(bool(par.get("CHANGE_METRIC", False))),
                     float(par.get("metric_factor", 1.0)))


def matchFeatures(f1, f2, par: dict, nthreads=0):
    a, b = _f(f1), _f(f2)
    Q, D = a.shape
    M = b.shape[0]
    o = _mopts(par)
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    met = np.zeros(max(Q, 1))
    P = lib().orc_match_features(_p(a), C.c_int(Q), C.c_int(Q), _p(b), C.c_int(M), C.c_int(M), C.c_int(D),
                                 C.byref(o), _p(pairs, C.c_uint32), _p(met), C.c_int(nthreads))
    return pairs[:P].copy(), met[:P].copy()


def getMatches(descSurface, descModel, par: dict, nthreads=0):
    a, b = _f(descSurface), _f(descModel)
    Q, D = a.shape
    M = b.shape[0]
    o = _mopts(par)
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    met = np.zeros(max(Q, 1))
    P = lib().orc_get_matches(_p(a), C.c_int(Q), C.c_int(Q), _p(b), C.c_int(M), C.c_int(M), C.c_int(D),
                              C.byref(o), _p(pairs, C.c_uint32), _p(met), C.c_int(nthreads))
    return pairs[:P].copy()


class DescOpts(C.Structure):
    _fields_ = [("min_pts", C.c_int32), ("max_pts", C.c_int32), ("R", C.c_double), ("thVar", C.c_double * 2),
                ("k", C.c_double), ("ALIGN_POINTS", C.c_int32)]


def desc_opts(options: dict) -> DescOpts:
    k = options["k"]
    kf = 1.0 if (k == "all" or k == 1) else float(k)
    mx = options["max_pts"]
    mx = 2**31 - 1 if (mx == float("inf") or mx > 2**31 - 1) else int(mx)
    return DescOpts(int(options["min_pts"]), mx, float(options["R"]), (C.c_double * 2)(*[float(v) for v in options["thVar"]]),
                    kf, int(bool(options["ALIGN_POINTS"])))


def getSpacialHistogramDescriptors(pts, sample_pts, options: dict, nthreads=0, single_mode=0):
    """single_mode: 0 double data; 1 = `single` arithmetic in getLocalPoints, keypoints single; 2 = cloud single, keypoints
    double (oracle/pcreg_oracle.py::getLocalPoints).  The arrays are passed as doubles (single data widened exactly)."""
    p, k = _f(pts), _f(sample_pts)
    P, S = p.shape[0], k.shape[0]
    feat = np.zeros((max(S, 1), 3)); desc = np.zeros((max(S, 1), 980))
    o = desc_opts(options)
    lib().orc_spatial_histogram_descriptors_sm.restype = C.c_int
    V = lib().orc_spatial_histogram_descriptors_sm(_p(p), C.c_int(P), C.c_int(P), _p(k), C.c_int(S), C.c_int(S), C.byref(o), C.c_int(single_mode),
                                                   _p(feat), _p(desc), C.c_int(nthreads))
    return feat[:V].copy(), desc[:V].copy()


def AlignPoints_KNN(pts, C1=False, C2=False):
    p = _f(pts)
    n = p.shape[0]
    al = np.zeros((n, 3), order="F")
    coeff = np.zeros(9)
    c = np.zeros(3)
    lib().orc_align_points_knn(_p(p), C.c_int(n), C.c_int(n), C.c_int(int(C1)), C.c_int(int(C2)), _p(al), _p(coeff), _p(c))
    return np.ascontiguousarray(al), coeff.reshape(3, 3, order="F").copy(), c
