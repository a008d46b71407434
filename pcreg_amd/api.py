"""Host-side mirror of the reference's MATLAB interface for the hot path.

Same function names, argument meaning and failure behaviour as the reference .m files
(LCJebe/PCReg); the arithmetic happens in libpcreg_hip.so on an MI355X through the
C ABI of include/pcreg.h (host tier).  Differences forced by the host language only:
MATLAB structs are dicts, ``[]`` is an empty (0, 0) array, indices stay 1-based.

    estimateTransform(pts1, pts2)                          estimateTransform.m:2
    calcDists(T, pts1, pts2)                               getInliersRANSAC.m:46
    ransac(pts1, pts2, ransacCoef, funcFindTransf, funcDist)   ransac.m:1
    getInliersRANSAC(loc1M, loc1S)                         getInliersRANSAC.m:12-42
    getMatches(descSurface, descModel, par)                getMatches.m:1
    AlignPoints_KNN(pts, C1, C2)                           AlignPoints_KNN.m:1
    quickTF(pts, TF) / invertTF(TF)                        quickTF.m:1 / invertTF.m:1
"""
from __future__ import annotations

import ctypes as C

import math

import numpy as np

from . import _lib
from ._lib import DescOpts, MatchOpts, RansacOpts, check, lib

EMPTY = np.zeros((0, 0))
"""MATLAB's ``[]`` (what estimateTransform / ransac return on failure)."""


def _fcol(a, dtype=np.float64) -> np.ndarray:
    a = np.asarray(a, dtype=dtype)
    if a.ndim != 2:
        raise ValueError("expected a 2-D array")
    return np.asfortranarray(a)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def _pts3(a, name: str) -> np.ndarray:
    a = _fcol(a)
    if a.shape[1] != 3:
        raise ValueError(f"{name} must be N x 3")
    return a


# ------------------------------------------------------------------ estimateTransform
def estimateTransform(pts1, pts2) -> np.ndarray:
    """T = estimateTransform(pts1, pts2) with [pts2, 1] * T = [pts1, 1]
    (estimateTransform.m:2-71).  Returns ``EMPTY`` where the reference returns []."""
    p1, p2 = _pts3(pts1, "pts1"), _pts3(pts2, "pts2")
    if p1.shape != p2.shape:
        raise ValueError("pts1 and pts2 must have the same size")
    n = p1.shape[0]
    T = np.zeros(16)
    empty = C.c_int(1)
    check(lib().pcreg_estimate_transform(_ptr(p1, C.c_double), _ptr(p2, C.c_double), C.c_int(n), C.c_int(n),
                                         _ptr(T, C.c_double), C.byref(empty)))
    return EMPTY.copy() if empty.value else T.reshape(4, 4, order="F").copy()


def calcDists(T, pts1, pts2) -> np.ndarray:
    """d = calcDists(T, pts1, pts2): squared distance of pts1 to [pts2,1]*T
    (getInliersRANSAC.m:46-54).  Like the reference it raises on an empty T."""
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("calcDists: T must be 4 x 4 (the reference errors on [] as well, ransac.m:77-79)")
    p1, p2 = _pts3(pts1, "pts1"), _pts3(pts2, "pts2")
    n = p1.shape[0]
    d = np.zeros(n)
    Tf = T.reshape(-1, order="F").copy()
    check(lib().pcreg_calc_dists(_ptr(Tf, C.c_double), _ptr(p1, C.c_double), _ptr(p2, C.c_double), C.c_int(n),
                                 C.c_int(n), _ptr(d, C.c_double)))
    return d


# ------------------------------------------------------------------------------ ransac
def _ransac_opts(coef: dict, seed: int = 0) -> RansacOpts:
    for k in ("minPtNum", "iterNum", "thInlrRatio", "thDist", "REFINE"):     # ransac.m:23-29
        if k not in coef:
            raise KeyError(f"ransacCoef.{k} is required (ransac.m:23-29)")
    return RansacOpts(int(coef["minPtNum"]), int(coef["iterNum"]), float(coef["thDist"]),
                      float(coef["thInlrRatio"]), int(bool(coef["REFINE"])),
                      int(bool(coef.get("VERBOSE", 1))), int(seed) & ((1 << 64) - 1))


def _matlab_round(x: float) -> int:
    return int(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def ransac(pts1, pts2, ransacCoef: dict, funcFindTransf=None, funcDist=None, *, sample_idx=None,
           seed: int = 0, return_iter_counts: bool = False):
    """[T, inlierIdx, numSuccess, maxInliers, ratio] = ransac(pts1, pts2, ransacCoef,
    @estimateTransform, @calcDists)   (ransac.m:1-118).

    With the handles every caller of the reference passes (``estimateTransform`` and
    ``calcDists`` of this module, or None) the whole loop runs on the GPU.  Any other
    pair of handles runs the reference's generic loop on the host, calling the handles.

    sample_idx (iterNum x minPtNum, 1-based) replaces ``randperm(ptNum)(1:minPtNum)``
    (ransac.m:42-43); None selects the built-in counter-based sampler seeded by `seed`.
    Failure mirrors ransac.m:77-89: T = EMPTY, inlierIdx empty, numSuccess = maxInliers = 0.
    """
    fit_is_ours = funcFindTransf is None or funcFindTransf is estimateTransform
    dist_is_ours = funcDist is None or funcDist is calcDists
    if not (fit_is_ours and dist_is_ours):
        return _ransac_generic(pts1, pts2, ransacCoef, funcFindTransf or estimateTransform,
                               funcDist or calcDists, sample_idx, seed)
    p1, p2 = _pts3(pts1, "pts1"), _pts3(pts2, "pts2")
    if p1.shape != p2.shape:
        raise ValueError("pts1 and pts2 must have the same size")
    n = p1.shape[0]
    o = _ransac_opts(ransacCoef, seed)
    si_ptr = None
    if sample_idx is not None:
        si = np.ascontiguousarray(sample_idx, dtype=np.int32)
        if si.shape != (o.iterNum, o.minPtNum):
            raise ValueError("sample_idx must be iterNum x minPtNum")
        if n and (si.min() < 1 or si.max() > n):
            raise ValueError("sample_idx entries must lie in 1..ptNum")
        si_ptr = _ptr(si, C.c_int32)
    T = np.zeros(16)
    inl = np.zeros(max(n, 1), dtype=np.int32)
    ni, ns, mi, fl = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(1)
    it1 = np.zeros(o.iterNum, dtype=np.int32) if return_iter_counts else None
    it2 = np.zeros(o.iterNum, dtype=np.int32) if return_iter_counts else None
    check(lib().pcreg_ransac(_ptr(p1, C.c_double), _ptr(p2, C.c_double), C.c_int(n), C.c_int(n), C.byref(o), si_ptr,
                             _ptr(T, C.c_double), _ptr(inl, C.c_int32), C.byref(ni), C.byref(ns), C.byref(mi),
                             C.byref(fl), _ptr(it1, C.c_int32) if it1 is not None else None,
                             _ptr(it2, C.c_int32) if it2 is not None else None))
    if fl.value:
        if o.VERBOSE:
            print("RANSAC could not find an appropriate transformation")                 # ransac.m:82
        out = (EMPTY.copy(), np.zeros(0), 0, 0, 0.0)
    else:
        if o.VERBOSE:                                                                     # ransac.m:100
            print("RANSAC succeeded %d times with a maximum of %d Inliers (%0.2f %%)"
                  % (ns.value, mi.value, 100.0 * mi.value / n))
        out = (T.reshape(4, 4, order="F").copy(), inl[:ni.value].astype(np.float64), ns.value, mi.value,
               100.0 * mi.value / n)
    if return_iter_counts:
        return out + (it1, it2)
    return out


def _ransac_generic(pts1, pts2, coef, fit, dist, sample_idx, seed):
    """ransac.m:21-116 for arbitrary function handles (host loop; the handles do the math)."""
    pts1 = np.asarray(pts1, dtype=np.float64)
    pts2 = np.asarray(pts2, dtype=np.float64)
    minPtNum, iterNum = int(coef["minPtNum"]), int(coef["iterNum"])
    thDist, ptNum = float(coef["thDist"]), pts1.shape[0]
    thInlr = _matlab_round(float(coef["thInlrRatio"]) * ptNum)
    REFINE, VERBOSE = bool(coef["REFINE"]), bool(coef.get("VERBOSE", 1))
    rng = np.random.default_rng(seed)
    inlrNum = np.zeros(iterNum, dtype=np.int64)
    inlrNum_refined = np.zeros(iterNum, dtype=np.int64)
    TForms = [None] * iterNum

    def _empty(f):
        return f is None or (isinstance(f, np.ndarray) and f.size == 0)

    for p in range(iterNum):
        s = (np.asarray(sample_idx[p]) - 1) if sample_idx is not None else rng.permutation(ptNum)[:minPtNum]
        f1 = fit(pts1[s], pts2[s])
        d = dist(f1, pts1, pts2)        # like ransac.m:48 this raises if f1 is empty
        inl = np.nonzero(np.asarray(d) < thDist)[0]
        inlrNum[p] = inl.size
        if inl.size >= thInlr:
            if REFINE:
                f2 = fit(pts1[inl], pts2[inl])
                d = dist(f2, pts1, pts2)
                inlrNum_refined[p] = int(np.sum(np.asarray(d) < thDist))
                if inlrNum_refined[p] >= thInlr:
                    TForms[p] = f2
            else:
                TForms[p] = f1
    counts = inlrNum_refined if REFINE else inlrNum
    idx = int(np.argmax(counts)) if iterNum else 0
    T = TForms[idx] if iterNum else None
    try:
        if _empty(T):
            raise ValueError("empty transform")
        d = dist(T, pts1, pts2)
    except Exception:
        if VERBOSE:
            print("RANSAC could not find an appropriate transformation")
        return EMPTY.copy(), np.zeros(0), 0, 0, 0.0
    inlierIdx = (np.nonzero(np.asarray(d) < thDist)[0] + 1).astype(np.float64)
    numSuccess, maxInliers = int(np.sum(counts >= thInlr)), int(counts[idx])
    if VERBOSE:
        print("RANSAC succeeded %d times with a maximum of %d Inliers (%0.2f %%)"
              % (numSuccess, maxInliers, 100.0 * maxInliers / ptNum))
    return T, inlierIdx, numSuccess, maxInliers, 100.0 * maxInliers / ptNum


def ransac_batched(pts1_list, pts2_list, ransacCoef: dict, *, sample_idx=None, seed: int = 0):
    """B independent registrations in one launch (the parfor of
    completeExperimentFast.m:201-225).  Returns a list of ransac() 5-tuples."""
    B = len(pts1_list)
    if B == 0:
        return []
    sizes = [np.asarray(p).shape[0] for p in pts1_list]
    offsets = np.zeros(B + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(sizes)
    total = int(offsets[-1])
    p1 = _pts3(np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 3) for p in pts1_list], axis=0), "pts1")
    p2 = _pts3(np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 3) for p in pts2_list], axis=0), "pts2")
    o = _ransac_opts(ransacCoef, seed)
    si_ptr = None
    if sample_idx is not None:
        si = np.ascontiguousarray(sample_idx, dtype=np.int32)
        if si.shape != (B, o.iterNum, o.minPtNum):
            raise ValueError("sample_idx must be B x iterNum x minPtNum")
        si_ptr = _ptr(si, C.c_int32)
    T = np.zeros(16 * B)
    inl = np.zeros(max(total, 1), dtype=np.int32)
    ni = np.zeros(B, dtype=np.int32); ns = np.zeros(B, dtype=np.int32)
    mi = np.zeros(B, dtype=np.int32); fl = np.zeros(B, dtype=np.int32)
    check(lib().pcreg_ransac_batched(_ptr(p1, C.c_double), _ptr(p2, C.c_double), C.c_int(total), C.c_int(total),
                                     _ptr(offsets, C.c_int32), C.c_int(B), C.byref(o), si_ptr, _ptr(T, C.c_double),
                                     _ptr(inl, C.c_int32), _ptr(ni, C.c_int32), _ptr(ns, C.c_int32),
                                     _ptr(mi, C.c_int32), _ptr(fl, C.c_int32)))
    out = []
    for b in range(B):
        if fl[b]:
            out.append((EMPTY.copy(), np.zeros(0), 0, 0, 0.0))
        else:
            a = int(offsets[b])
            out.append((T[16 * b:16 * b + 16].reshape(4, 4, order="F").copy(),
                        inl[a:a + int(ni[b])].astype(np.float64), int(ns[b]), int(mi[b]),
                        100.0 * int(mi[b]) / max(sizes[b], 1)))
    return out


GETINLIERS_COEFF = dict(minPtNum=3, iterNum=int(2e4), thDist=0.5, thInlrRatio=0.1, REFINE=True)
"""coeff of getInliersRANSAC.m:17-31."""


def getInliersRANSAC(loc1M, loc1S, *, sample_idx=None, seed: int = 0, VERBOSE: int = 1) -> dict:
    """The script getInliersRANSAC.m as a function of its two workspace inputs; returns the
    variables it leaves in the workspace (T, inlierPtIdx, pts1_aligned)."""
    pts1 = np.asarray(loc1M, dtype=np.float64)                                  # :12
    pts2 = np.asarray(loc1S, dtype=np.float64)                                  # :13
    coeff = dict(GETINLIERS_COEFF, VERBOSE=VERBOSE)                             # :17-31
    T, inlierPtIdx, *_ = ransac(pts1, pts2, coeff, estimateTransform, calcDists,
                                sample_idx=sample_idx, seed=seed)               # :34
    ws = dict(T=T, inlierPtIdx=inlierPtIdx, coeff=coeff, pts1=pts1, pts2=pts2)
    if T.size:                                                                  # :39
        ws["pts1_aligned"] = quickTF(pts1, T)                                   # :40-41
    return ws


# -------------------------------------------------------------------------- getMatches
def _match_opts(par: dict) -> MatchOpts:
    metric = str(par.get("Metric", "SSD")).upper()
    if metric not in ("SAD", "SSD"):
        raise ValueError("par.Metric must be 'SAD' or 'SSD'")
    method = str(par.get("Method", "Exhaustive"))
    if method not in ("Exhaustive", "Approximate"):
        raise ValueError("par.Method must be 'Exhaustive' or 'Approximate'")
    # 'Approximate' (randomised kd-trees in MATLAB) is answered with the exact search
    return MatchOpts(_lib.METRIC_SAD if metric == "SAD" else _lib.METRIC_SSD,
                     float(par.get("MatchThreshold", 10.0 if metric == "SAD" else 1.0)),
                     float(par.get("MaxRatio", 0.6)), int(bool(par.get("Unique", False))),
                     int(bool(par.get("Prenormalized", False))), int(bool(par.get("UNNORMALIZE", False))),
                     float(par.get("norm_factor", 0.0)), int(bool(par.get("CHANGE_METRIC", False))),
                     float(par.get("metric_factor", 1.0)))


def getLocalPoints(pts, R, c, min_points, max_points):
    """[pts_sphere, dists] = getLocalPoints(pts, R, c, min_points, max_points)  (getLocalPoints.m:1-35): the points within R of c,
    RELATIVE to c, or two empty arrays where MATLAB returns [].  A float32 cloud or centre selects MATLAB's single arithmetic
    (the result is float32 then, as in MATLAB)."""
    p_single = isinstance(pts, np.ndarray) and pts.dtype == np.float32
    c_single = isinstance(c, np.ndarray) and c.dtype == np.float32
    mode = 1 if c_single else (2 if p_single else 0)
    P = _fcol(pts)
    if P.shape[1] != 3:
        raise ValueError("pts must be N x 3")
    N = P.shape[0]
    cc = (C.c_double * 3)(*[float(v) for v in np.asarray(c, dtype=np.float64).ravel()[:3]])
    out = np.zeros((max(N, 1), 3), dtype=np.float64, order="F")
    dists = np.zeros(max(N, 1), dtype=np.float64)
    n = C.c_int(0)
    check(lib().pcreg_get_local_points(_ptr(P, C.c_double), N, max(N, 1), C.c_double(float(R)), cc, C.c_double(float(min_points)),
                                       C.c_double(float(max_points)), mode, _ptr(out, C.c_double), _ptr(dists, C.c_double), C.byref(n)))
    k = n.value
    res = np.ascontiguousarray(out.ravel(order="F")[:3 * k].reshape(k, 3, order="F")), dists[:k].copy()
    if mode:
        return res[0].astype(np.float32), res[1].astype(np.float32)
    return res


class DescSet:
    """A descriptor set resident on the GPU (pcreg_desc_set_create): uploaded once, matched many times -- the surface set and
    the model set of completeExperimentFast.m:101-150.  `with DescSet(desc) as h:` or call close()."""

    def __init__(self, desc):
        d = _fcol(desc)
        self.n, self.D = d.shape
        self._h = C.c_void_p()
        check(lib().pcreg_desc_set_create(_ptr(d, C.c_double), self.n, max(self.n, 1), self.D, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().pcreg_desc_set_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def getMatchesOnSet(hSurface: DescSet, hModel: DescSet, rows, par: dict) -> np.ndarray:
    """getMatches(descSurface, descModel[rows], par) on resident sets (pcreg_get_matches_on_sets): the same pairs, bit for bit,
    without uploading either matrix again.  rows: 0-based ascending row numbers of the model set, or None for all of it."""
    o = _match_opts(par)
    Q = hSurface.n
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    P = C.c_int(0)
    if rows is None:
        rp, nr = None, 0
    else:
        r = np.ascontiguousarray(np.asarray(rows, dtype=np.int32).ravel())
        if r.size == 0:
            return np.zeros((0, 2), dtype=np.uint32)
        rp, nr = _ptr(r, C.c_int32), int(r.size)
    check(lib().pcreg_get_matches_on_sets(hSurface._h, hModel._h, rp, nr, C.byref(o), _ptr(pairs, C.c_uint32), None, C.byref(P)))
    return pairs[:P.value].copy()


def getMatchesSegmented(descSurface, descModel, rows_list, par: dict) -> list:
    """[getMatches(descSurface, descModel[rows], par) for rows in rows_list] in ONE library call (pcreg_get_matches_segmented):
    the per-sphere calls of completeExperimentFast.m:131-149.  rows: 0-based ascending row numbers of descModel."""
    dS, dM = _fcol(descSurface), _fcol(descModel)
    if dS.shape[1] != dM.shape[1]:
        raise ValueError("descriptor lengths differ")
    Q, D = dS.shape
    VM = dM.shape[0]
    S = len(rows_list)
    off = np.zeros(S + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(r) for r in rows_list])
    rows = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.int32).ravel() for r in rows_list] + [np.zeros(0, np.int32)]))
    o = _match_opts(par)
    pairs = np.zeros((max(S, 1), max(Q, 1), 2), dtype=np.uint32)
    n_pairs = np.zeros(max(S, 1), dtype=np.int32)
    check(lib().pcreg_get_matches_segmented(_ptr(dS, C.c_double), Q, max(Q, 1), _ptr(dM, C.c_double), VM, max(VM, 1), D,
                                            _ptr(rows, C.c_int32) if rows.size else None, _ptr(off, C.c_int32), S, C.byref(o),
                                            _ptr(pairs, C.c_uint32), _ptr(n_pairs, C.c_int32)))
    return [pairs[z, :n_pairs[z]].copy() for z in range(S)]


def getMatchesSegmentedOnSet(hSurface: DescSet, hModel: DescSet, rows_list, par: dict) -> list:
    """getMatchesSegmented on resident sets (pcreg_get_matches_segmented_on_sets): every sphere of completeExperimentFast.m:131-149
    in one call with only the row lists going up and the pairs coming down."""
    if hSurface.D != hModel.D:
        raise ValueError("descriptor lengths differ")
    Q, S = hSurface.n, len(rows_list)
    off = np.zeros(S + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(r) for r in rows_list])
    rows = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.int32).ravel() for r in rows_list] + [np.zeros(0, np.int32)]))
    o = _match_opts(par)
    pairs = np.zeros((max(S, 1), max(Q, 1), 2), dtype=np.uint32)
    n_pairs = np.zeros(max(S, 1), dtype=np.int32)
    check(lib().pcreg_get_matches_segmented_on_sets(hSurface._h, hModel._h, _ptr(rows, C.c_int32) if rows.size else None, _ptr(off, C.c_int32), S,
                                                    C.byref(o), _ptr(pairs, C.c_uint32), _ptr(n_pairs, C.c_int32)))
    return [pairs[z, :n_pairs[z]].copy() for z in range(S)]


def sphereCounts(featModel, centres, R: float) -> np.ndarray:
    """counts(i) = nnz(vecnorm(featModel - centres(i, :), 2, 2) < R)  (completeExperimentFast.m:52-64, pcreg_sphere_counts)."""
    f, c = _fcol(featModel), _fcol(centres)
    VM, S = f.shape[0], c.shape[0]
    counts = np.zeros(max(S, 1), dtype=np.int32)
    check(lib().pcreg_sphere_counts(_ptr(f, C.c_double), VM, max(VM, 1), _ptr(c, C.c_double), S, max(S, 1), C.c_double(R), _ptr(counts, C.c_int32)))
    return counts[:S].astype(np.int64)


def sphereSweep(hSurface: DescSet, hModel: DescSet, featSurface, featModel, centres, num_desc, R_desc: float, par: dict, putative_thresh: int,
                ransacCoef: dict, seed: int = 0) -> dict:
    """completeExperimentFast.m:101-224 for the spheres `centres` (already filtered by their counts `num_desc` = sphereCounts) in one
    library call on resident descriptor sets (pcreg_sphere_sweep): per-sphere row lists and matches, the spheres above the putative
    threshold and their ransac results.  Keys as pcreg_amd.sweep.SphereSweep.run's."""
    fS, fM, c = _fcol(featSurface), _fcol(featModel), _fcol(centres)
    S = c.shape[0]
    nd = np.ascontiguousarray(num_desc, dtype=np.int32)
    Q = hSurface.n
    tot = int(nd.sum())
    rows = np.zeros(max(tot, 1), dtype=np.int32)
    pairs = np.zeros((max(S, 1), max(Q, 1), 2), dtype=np.uint32)
    n_pairs = np.zeros(max(S, 1), dtype=np.int32)
    trial = np.zeros(max(S, 1), dtype=np.int32)
    nt = C.c_int(0)
    T = np.zeros((max(S, 1), 16)); ns = np.zeros(max(S, 1), dtype=np.int32); mi = np.zeros(max(S, 1), dtype=np.int32); fl = np.zeros(max(S, 1), dtype=np.int32)
    o, rc = _match_opts(par), _ransac_opts(ransacCoef, seed)
    check(lib().pcreg_sphere_sweep(hSurface._h, hModel._h, _ptr(fS, C.c_double), max(fS.shape[0], 1), _ptr(fM, C.c_double), max(fM.shape[0], 1),
                                   _ptr(c, C.c_double), S, max(S, 1), _ptr(nd, C.c_int32), C.c_double(R_desc), C.byref(o), int(putative_thresh), C.byref(rc),
                                   _ptr(rows, C.c_int32), _ptr(pairs, C.c_uint32), _ptr(n_pairs, C.c_int32), _ptr(trial, C.c_int32), C.byref(nt),
                                   _ptr(T, C.c_double), _ptr(ns, C.c_int32), _ptr(mi, C.c_int32), _ptr(fl, C.c_int32)))
    n = nt.value
    off = np.zeros(S + 1, dtype=np.int64); off[1:] = np.cumsum(nd)
    npr = n_pairs[:S].astype(np.int64)
    tr = trial[:n].astype(np.int64)
    return dict(centres=np.asarray(centres, dtype=np.float64), num_desc=nd.astype(np.int64), num_putative=npr,
                matches=[pairs[i, :npr[i]].copy() for i in range(S)], model_rows=[rows[off[i]:off[i + 1]].astype(np.int64) for i in range(S)], trial=tr,
                statsPutative=npr[tr], statsSuccess=ns[:n].astype(np.int64), statsInliers=mi[:n].astype(np.int64),
                statsRatio=np.array([100.0 * mi[t] / npr[tr[t]] if not fl[t] else 0.0 for t in range(n)], dtype=np.float64),
                transforms=[None if fl[t] else T[t].reshape(4, 4, order="F").copy() for t in range(n)])


class SphereModel:
    """What the sphere sweep makes of the MODEL alone (pcreg_sphere_model_create): the kept spheres' row lists and keypoints, the model
    descriptor set restricted to the union of those rows, its powered rows per getMatches options -- one model, many surfaces.
    centres / num_desc: the spheres kept after sphereCounts.  `with SphereModel(...) as sm:` or call close()."""

    def __init__(self, hModel: DescSet, featModel, centres, num_desc, R_desc: float):
        fM, c = _fcol(featModel), _fcol(centres)
        self.S = c.shape[0]
        self.centres = np.asarray(centres, dtype=np.float64).reshape(self.S, 3)
        self.num_desc = np.ascontiguousarray(num_desc, dtype=np.int32)
        tot = int(self.num_desc.sum())
        rows = np.zeros(max(tot, 1), dtype=np.int32)
        self._h = C.c_void_p()
        check(lib().pcreg_sphere_model_create(hModel._h, _ptr(fM, C.c_double), max(fM.shape[0], 1), _ptr(c, C.c_double), self.S, max(self.S, 1),
                                              _ptr(self.num_desc, C.c_int32), C.c_double(R_desc), _ptr(rows, C.c_int32), C.byref(self._h)))
        off = np.zeros(self.S + 1, dtype=np.int64); off[1:] = np.cumsum(self.num_desc)
        self.model_rows = [rows[off[i]:off[i + 1]].astype(np.int64) for i in range(self.S)]

    def close(self) -> None:
        if self._h:
            lib().pcreg_sphere_model_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sphereSweepOnModel(sm: SphereModel, hSurface: DescSet, featSurface, par: dict, putative_thresh: int, ransacCoef: dict, seed: int = 0) -> dict:
    """sphereSweep for one more surface against a prepared SphereModel (pcreg_sphere_sweep_on_model): the same results, without redoing
    what belongs to the model."""
    fS = _fcol(featSurface)
    S, Q = sm.S, hSurface.n
    pairs = np.zeros((max(S, 1), max(Q, 1), 2), dtype=np.uint32)
    n_pairs = np.zeros(max(S, 1), dtype=np.int32)
    trial = np.zeros(max(S, 1), dtype=np.int32)
    nt = C.c_int(0)
    T = np.zeros((max(S, 1), 16)); ns = np.zeros(max(S, 1), dtype=np.int32); mi = np.zeros(max(S, 1), dtype=np.int32); fl = np.zeros(max(S, 1), dtype=np.int32)
    o, rc = _match_opts(par), _ransac_opts(ransacCoef, seed)
    check(lib().pcreg_sphere_sweep_on_model(sm._h, hSurface._h, _ptr(fS, C.c_double), max(fS.shape[0], 1), C.byref(o), int(putative_thresh), C.byref(rc),
                                            _ptr(pairs, C.c_uint32), _ptr(n_pairs, C.c_int32), _ptr(trial, C.c_int32), C.byref(nt), _ptr(T, C.c_double),
                                            _ptr(ns, C.c_int32), _ptr(mi, C.c_int32), _ptr(fl, C.c_int32)))
    n = nt.value
    npr = n_pairs[:S].astype(np.int64)
    tr = trial[:n].astype(np.int64)
    return dict(centres=sm.centres, num_desc=sm.num_desc.astype(np.int64), num_putative=npr,
                matches=[pairs[i, :npr[i]].copy() for i in range(S)], model_rows=sm.model_rows, trial=tr,
                statsPutative=npr[tr], statsSuccess=ns[:n].astype(np.int64), statsInliers=mi[:n].astype(np.int64),
                statsRatio=np.array([100.0 * mi[t] / npr[tr[t]] if not fl[t] else 0.0 for t in range(n)], dtype=np.float64),
                transforms=[None if fl[t] else T[t].reshape(4, 4, order="F").copy() for t in range(n)])


def getMatches(descSurface, descModel, par: dict) -> np.ndarray:
    """matches = getMatches(descSurface, descModel, par)  (getMatches.m:1-59):
    P x 2 uint32, 1-based [surfaceIdx, modelIdx], ascending in the first column."""
    for k in ("UNNORMALIZE", "CHANGE_METRIC", "Method", "MatchThreshold", "MaxRatio", "Metric", "Unique"):
        if k not in par:
            raise KeyError(f"par.{k} is required (getMatches.m:22-56)")
    import time
    t0 = time.time()                                                            # :12 tic
    dS, dM = _fcol(descSurface), _fcol(descModel)
    if dS.shape[1] != dM.shape[1]:
        raise ValueError("descriptor lengths differ")
    Q, D = dS.shape
    M = dM.shape[0]
    o = _match_opts(par)
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    P = C.c_int(0)
    check(lib().pcreg_get_matches(_ptr(dS, C.c_double), C.c_int(Q), C.c_int(Q), _ptr(dM, C.c_double), C.c_int(M),
                                  C.c_int(M), C.c_int(D), C.byref(o), _ptr(pairs, C.c_uint32), None, C.byref(P)))
    if par.get("VERBOSE", 1):
        print("Calculated matches in %0.1f seconds..." % (time.time() - t0))    # :58
    return pairs[:P.value].copy()


def matchFeatures(features1, features2, **kw):
    """The subset of MathWorks matchFeatures that getMatches uses (getMatches.m:51-56):
    returns (indexPairs P x 2 uint32, matchMetric P)."""
    par = dict(Method=kw.get("Method", "Exhaustive"), MatchThreshold=kw.get("MatchThreshold", 10.0),
               MaxRatio=kw.get("MaxRatio", 0.6), Metric=kw.get("Metric", "SSD"), Unique=kw.get("Unique", False),
               Prenormalized=kw.get("Prenormalized", False))
    f1, f2 = _fcol(features1), _fcol(features2)
    Q, D = f1.shape
    M = f2.shape[0]
    o = _match_opts(par)
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    met = np.zeros(max(Q, 1))
    P = C.c_int(0)
    check(lib().pcreg_match_features(_ptr(f1, C.c_double), C.c_int(Q), C.c_int(Q), _ptr(f2, C.c_double), C.c_int(M),
                                     C.c_int(M), C.c_int(D), C.byref(o), _ptr(pairs, C.c_uint32),
                                     _ptr(met, C.c_double), C.byref(P)))
    return pairs[:P.value].copy(), met[:P.value].copy()


# ------------------------------------------------------------------------ point search
def knn2_points(query, model):
    """Two nearest model points per query (fp32): (idx [Q,2] 0-based, dist [Q,2] squared)."""
    q, m = _fcol(query, np.float32), _fcol(model, np.float32)
    Q, M = q.shape[0], m.shape[0]
    idx = np.zeros((max(Q, 1), 2), dtype=np.int32)
    dist = np.zeros((max(Q, 1), 2), dtype=np.float32)
    check(lib().pcreg_knn2_points_f32(_ptr(q, C.c_float), C.c_int(Q), C.c_int(Q), _ptr(m, C.c_float), C.c_int(M),
                                      C.c_int(M), _ptr(idx, C.c_int32), _ptr(dist, C.c_float)))
    return idx[:Q], dist[:Q]


class Model:
    """A model cloud uploaded and prepared ONCE (pcreg_model_create), matched against any number of surfaces: the host-tier
    handle a MATLAB caller keeps across the sphere loop of completeExperimentFast.m:131-149.  Use as a context manager or
    call close()."""

    def __init__(self, model):
        m = _fcol(model, np.float32)
        self.M = m.shape[0]
        self._h = C.c_void_p()
        check(lib().pcreg_model_create(_ptr(m, C.c_float), C.c_int(self.M), C.c_int(max(self.M, 1)), C.byref(self._h)))

    def match_points(self, query, thr_abs: float, max_ratio: float, unique: bool = True) -> np.ndarray:
        if not self._h.value:
            raise ValueError("the model handle is closed")
        q = _fcol(query, np.float32)
        Q = q.shape[0]
        pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
        P = C.c_int(0)
        check(lib().pcreg_model_match_points_f32(self._h, _ptr(q, C.c_float), C.c_int(Q), C.c_int(max(Q, 1)), C.c_float(thr_abs),
                                                 C.c_float(max_ratio), C.c_int(int(unique)), _ptr(pairs, C.c_uint32), C.byref(P)))
        return pairs[:P.value].copy()

    def close(self):
        if self._h.value:
            lib().pcreg_model_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match_points(query, model, thr_abs: float, max_ratio: float, unique: bool = True) -> np.ndarray:
    """matchFeatures' filter chain on raw 3-D fp32 points -> P x 2 uint32 (1-based)."""
    q, m = _fcol(query, np.float32), _fcol(model, np.float32)
    Q, M = q.shape[0], m.shape[0]
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    P = C.c_int(0)
    check(lib().pcreg_match_points_f32(_ptr(q, C.c_float), C.c_int(Q), C.c_int(Q), _ptr(m, C.c_float), C.c_int(M),
                                       C.c_int(M), C.c_float(thr_abs), C.c_float(max_ratio), C.c_int(int(unique)),
                                       _ptr(pairs, C.c_uint32), C.byref(P)))
    return pairs[:P.value].copy()


# --------------------------------------------------------------------- AlignPoints_KNN
def AlignPoints_KNN(pts, *varargin):
    """[pts_aligned, coeff_unambig, c] = AlignPoints_KNN(pts[, C1, C2])  (AlignPoints_KNN.m:1-60).
    Like the reference, C1/C2 are honoured only when BOTH are given (:8-14)."""
    C1, C2 = (bool(varargin[0]), bool(varargin[1])) if len(varargin) == 2 else (False, False)
    if getattr(pts, "dtype", None) == np.float32:          # class-preserving, like MATLAB: single in -> single out
        p = _fcol(np.asarray(pts).reshape(-1, 3), np.float32)
        n = p.shape[0]
        aligned = np.zeros((n, 3), dtype=np.float32, order="F"); coeff = np.zeros(9, np.float32); c = np.zeros(3, np.float32)
        check(lib().pcreg_align_points_knn_f32(_ptr(p, C.c_float), C.c_int(n), C.c_int(n), C.c_int(int(C1)), C.c_int(int(C2)),
                                               _ptr(aligned, C.c_float), _ptr(coeff, C.c_float), _ptr(c, C.c_float)))
        return np.ascontiguousarray(aligned), coeff.reshape(3, 3, order="F").copy(), c.reshape(1, 3)
    p = _pts3(pts, "pts")
    n = p.shape[0]
    aligned = np.zeros((n, 3), order="F")
    coeff = np.zeros(9)
    c = np.zeros(3)
    check(lib().pcreg_align_points_knn(_ptr(p, C.c_double), C.c_int(n), C.c_int(n), C.c_int(int(C1)), C.c_int(int(C2)),
                                       _ptr(aligned, C.c_double), _ptr(coeff, C.c_double), _ptr(c, C.c_double)))
    return np.ascontiguousarray(aligned), coeff.reshape(3, 3, order="F").copy(), c.reshape(1, 3)


def AlignPoints_KNN_batched(pts_list, C1: bool = False, C2: bool = False):
    """The same LRF for a list of supports in one launch; returns (aligned_list, coeff [B,3,3], c [B,3], status [B])."""
    B = len(pts_list)
    sizes = [np.asarray(p).shape[0] for p in pts_list]
    offsets = np.zeros(B + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(sizes)
    total = int(offsets[-1])
    p = _pts3(np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1, 3) for x in pts_list], axis=0), "pts")
    aligned = np.zeros((max(total, 1), 3), order="F")
    coeff = np.zeros(9 * B); c = np.zeros(3 * B); status = np.zeros(B, dtype=np.int32)
    check(lib().pcreg_align_points_knn_batched(_ptr(p, C.c_double), C.c_int(total), C.c_int(total), _ptr(offsets, C.c_int32),
                                               C.c_int(B), C.c_int(int(C1)), C.c_int(int(C2)), _ptr(aligned, C.c_double),
                                               _ptr(coeff, C.c_double), _ptr(c, C.c_double), _ptr(status, C.c_int32)))
    al = [np.ascontiguousarray(aligned[offsets[b]:offsets[b + 1]]) for b in range(B)]
    return al, coeff.reshape(B, 3, 3).transpose(0, 2, 1).copy(), c.reshape(B, 3), status


# ------------------------------------------------------- getSpacialHistogramDescriptors
def _desc_opts(options: dict) -> DescOpts:
    kk = options["k"]
    kf = 1.0 if (isinstance(kk, str) and kk == "all") or kk == 1 else float(kk)      # :75
    mx = options["max_pts"]
    mx = 2**31 - 1 if (mx == float("inf") or mx > 2**31 - 1) else int(mx)
    return DescOpts(int(options["min_pts"]), mx, float(options["R"]), (C.c_double * 2)(*[float(v) for v in options["thVar"]]),
                    kf, int(bool(options["ALIGN_POINTS"])))


def speedyDescriptors(pts, sample_opts: dict, options: dict, rng=None):
    """[feat, desc] = speedyDescriptors(pts, sample_opts, options)  (speedyDescriptors.m:2-82).

    The reference cuts the model into cuboid regions so that its brute-force getLocalPoints scans stay short, draws the
    keypoints of every region on the host (:55, :86-101) and calls getSpacialHistogramDescriptors once per region on the
    region's crop (+ a margin of R).  Every keypoint lies at least R inside its crop's bounding box, so its support is the same
    in the crop and in the whole cloud: here the keypoints are drawn exactly as the reference draws them (same regions, same
    order, same counts; `rng` -- a numpy Generator, default_rng(0) if None -- plays MATLAB's rand) and ALL of them go through
    ONE device call on the whole cloud (the uniform grid replaces the tiling): the same rows in the same order.
    -> (feat V x 3, desc V x 980, the sampled keypoints in region order)."""
    for k in ("max_region_size", "R"):
        if k not in options:
            raise KeyError(f"options.{k} is required (speedyDescriptors.m:10,33)")
    rng = rng or np.random.default_rng(0)
    p = np.asarray(pts)
    d, R = float(sample_opts["d"]), float(options["R"])
    lo, hi = p.min(axis=0).astype(np.float64), p.max(axis=0).astype(np.float64)          # :18-19 (pointCloud limits)
    ext = hi - lo
    nreg = np.ceil(ext / float(options["max_region_size"])).astype(np.int64)              # :20
    step = ext / nreg                                                                     # :21
    edges = [lo[a] + step[a] * np.arange(int(np.floor(ext[a] / step[a] + 1e-10)) + 1) for a in range(3)]    # :24-26
    if [len(e) - 1 for e in edges] != nreg.tolist():                                      # :27
        raise AssertionError("region bounds do not tile the cloud")
    if options.get("VERBOSE", 0):
        print(f"Divided model into {int(nreg.prod())} regions")                           # :35-37
    draws = []
    for ix in range(nreg[0]):                                                             # :44-46
        inx = (p[:, 0] > edges[0][ix] - R) & (p[:, 0] < edges[0][ix + 1] + R)
        for iy in range(nreg[1]):
            inxy = inx & (p[:, 1] > edges[1][iy] - R) & (p[:, 1] < edges[1][iy + 1] + R)
            for iz in range(nreg[2]):
                crop = p[inxy & (p[:, 2] > edges[2][iz] - R) & (p[:, 2] < edges[2][iz + 1] + R)]        # :48-52
                if crop.shape[0] <= 500:                                                  # :87 / :99
                    continue
                clo, chi = crop.min(axis=0).astype(np.float64), crop.max(axis=0).astype(np.float64)
                span = (chi - clo) - 2.0 * R                                              # :89-91, margin = -R
                num = float(span[0] * span[1] * span[2]) / d ** 3
                num = int(math.floor(num + 0.5)) if num >= 0 else -int(math.floor(-num + 0.5))          # :92 round
                draws.append(rng.random((max(num, 0), 3)) * span + clo + R)              # :95-101
    kp = np.vstack(draws) if draws else np.zeros((0, 3))
    if kp.shape[0] == 0:
        return np.zeros((0, 3)), np.zeros((0, 980)), kp
    opt = dict(options); opt["VERBOSE"] = 0                                               # :12
    feat, desc = getSpacialHistogramDescriptors(p, kp, opt)                               # ONE call instead of :58-59 per region
    return feat, desc, kp


def getSpacialHistogramDescriptors(pts, sample_pts, options: dict):
    """[feat, desc] = getSpacialHistogramDescriptors(pts, sample_pts, options)
    (getSpacialHistogramDescriptors.m:2-183): feat V x 3 keypoint locations, desc V x 980
    spherical count histograms of the surviving keypoints, in input order."""
    for k in ("min_pts", "max_pts", "R", "thVar", "k", "ALIGN_POINTS"):        # :18-23
        if k not in options:
            raise KeyError(f"options.{k} is required (getSpacialHistogramDescriptors.m:18-23)")
    import time
    t0 = time.time()
    ps, ss = getattr(pts, "dtype", None) == np.float32, getattr(sample_pts, "dtype", None) == np.float32
    if ps or ss:
        # `single` data (pcread; completeExperimentFast.m:309).  feat / desc are DOUBLE (the reference preallocates them with
        # nan(...), getSpacialHistogramDescriptors.m:61-62); getLocalPoints' element-wise single arithmetic -- which keypoints
        # survive, which points form a support -- is reproduced (include/pcreg.h)
        p = _fcol(np.asarray(pts).reshape(-1, 3), np.float32 if ps else np.float64)
        s = _fcol(np.asarray(sample_pts).reshape(-1, 3), np.float32 if ss else np.float64)
        P, S = p.shape[0], s.shape[0]
        o = _desc_opts(options)
        feat = np.zeros((max(S, 1), 3)); desc = np.zeros((max(S, 1), 980))
        V = C.c_int(0)
        check(lib().pcreg_spatial_histogram_descriptors_mixed(p.ctypes.data_as(C.c_void_p), C.c_int(int(ps)), C.c_int(P), C.c_int(P),
                                                              s.ctypes.data_as(C.c_void_p), C.c_int(int(ss)), C.c_int(S), C.c_int(S), C.byref(o),
                                                              _ptr(feat, C.c_double), _ptr(desc, C.c_double), C.byref(V)))
        if options.get("VERBOSE", 1):
            print("Calculated descriptors in %0.1f seconds..." % (time.time() - t0))
        return feat[:V.value].copy(), desc[:V.value].copy()
    p, s = _pts3(pts, "pts"), _pts3(sample_pts, "sample_pts")
    P, S = p.shape[0], s.shape[0]
    o = _desc_opts(options)
    feat = np.zeros((max(S, 1), 3))
    desc = np.zeros((max(S, 1), 980))
    V = C.c_int(0)
    check(lib().pcreg_spatial_histogram_descriptors(_ptr(p, C.c_double), C.c_int(P), C.c_int(P), _ptr(s, C.c_double),
                                                    C.c_int(S), C.c_int(S), C.byref(o), _ptr(feat, C.c_double),
                                                    _ptr(desc, C.c_double), C.byref(V)))
    if options.get("VERBOSE", 1):
        print("Calculated descriptors in %0.1f seconds..." % (time.time() - t0))       # :181
    return feat[:V.value].copy(), desc[:V.value].copy()


# ----------------------------------------------------------------- transform helpers
def quickTF(pts, TF) -> np.ndarray:
    """quickTF.m:5-7 -- [pts 1] * TF (host arithmetic: 12 flops per point, not a hot path)."""
    pts = np.asarray(pts, dtype=np.float64)
    return (np.hstack([pts, np.ones((pts.shape[0], 1))]) @ np.asarray(TF, dtype=np.float64))[:, :3]


def invertTF(TF) -> np.ndarray:
    """invertTF.m:5-7."""
    TF = np.asarray(TF, dtype=np.float64)
    out = np.eye(4)
    out[:3, :3] = TF[:3, :3].T
    out[3, :3] = -TF[3, :3] @ TF[:3, :3].T
    return out
