"""oracle/pcreg_oracle.py -- CPU (numpy, IEEE double) restatement of the PCReg hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pcreg_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker.

Every function cites the reference lines it restates (paths relative to the
reference checkout, LCJebe/PCReg):

    estimateTransform   estimateTransform.m:8-71
    calcDists           getInliersRANSAC.m:46-54
    ransac              ransac.m:21-116
    getInliersRANSAC    getInliersRANSAC.m:12-42
    getMatches          getMatches.m:22-56
    AlignPoints_KNN     AlignPoints_KNN.m:8-59
    quickTF / invertTF  quickTF.m:5-7 / invertTF.m:5-7
    getLocalPoints      getLocalPoints.m:8-35
    sphere_sweep        completeExperimentFast.m:46-224
    final_stage         completeExperimentFast.m:280-394

Pinning status
--------------
* The reference is MATLAB and cannot run here (no MATLAB/Octave).  The restatement is
  pinned against every known answer the reference's own test scripts hold
  (testTransformEstimation.m:2-14, testRANSAC.m:13-42, debugRANSAC.m:2-54, the
  invariants of ransac.m, invertTF/quickTF round trip) -- see tests/test_oracle_kat.py.
* ``matchFeatures`` (Computer Vision Toolbox) and ``pca`` (Statistics Toolbox) are
  closed-source MathWorks code that is not in the reference tree and that no reference
  test pins.  Their *documented* semantics are restated below; for those two
  boundaries this oracle is "parity unpinned" (SURVEY.md section 8c).
* Rounding-level choices MATLAB makes inside BLAS/LAPACK (summation order, FMA use,
  SVD sign conventions) are unknowable here.  This file uses numpy/LAPACK (the same
  family MATLAB calls); oracle/pcreg_oracle.c is a second, independent plain-C
  restatement; tests require the two to agree to ~1e-12 and the HIP path to agree
  with them on index sets exactly.

Deliberate deviation: a rank-deficient sample makes
estimateTransform return [] (estimateTransform.m:11-14) and the reference then throws
out of ransac at ransac.m:48.  Here such a hypothesis scores 0 inliers instead.
"""
from __future__ import annotations

import math
import numpy as np

__all__ = [
    "matlab_round", "matlab_rank", "eul2rotm", "estimateTransform", "calcDists",
    "ransac", "getInliersRANSAC", "quickTF", "invertTF", "sample_table",
    "matchFeatures", "getMatches", "preprocess_descriptors", "pca_eig",
    "AlignPoints_KNN", "getLocalPoints", "knn2_points_f32", "match_points_f32",
    "histogram_edges", "histcounts_loc", "getSpacialHistogramDescriptors",
]


# --------------------------------------------------------------------------- helpers
def matlab_round(x: float) -> int:
    """MATLAB ``round``: half away from zero (ransac.m:28, AlignPoints_KNN.m:21)."""
    return int(math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def matlab_rank(A: np.ndarray) -> int:
    """MATLAB ``rank``: #singular values > max(size(A)) * eps(norm(A)).

    eps(x) is the spacing of doubles at x (``np.spacing``).  Used at
    estimateTransform.m:11.
    """
    A = np.asarray(A, dtype=np.float64)
    if A.size == 0:
        return 0
    s = np.linalg.svd(A, compute_uv=False)
    tol = max(A.shape) * np.spacing(s.max())
    return int(np.sum(s > tol))


def eul2rotm(eul, seq: str = "ZYX") -> np.ndarray:
    """Robotics Toolbox ``eul2rotm`` (test-only; testRANSAC.m:17, debugRANSAC.m:11).

    'ZYX' (default): R = Rz(e1) Ry(e2) Rx(e3);  'XYZ': R = Rx(e1) Ry(e2) Rz(e3).
    """
    a, b, c = (float(v) for v in eul)

    def rx(t):
        return np.array([[1, 0, 0], [0, math.cos(t), -math.sin(t)], [0, math.sin(t), math.cos(t)]])

    def ry(t):
        return np.array([[math.cos(t), 0, math.sin(t)], [0, 1, 0], [-math.sin(t), 0, math.cos(t)]])

    def rz(t):
        return np.array([[math.cos(t), -math.sin(t), 0], [math.sin(t), math.cos(t), 0], [0, 0, 1]])

    if seq.upper() == "ZYX":
        return rz(a) @ ry(b) @ rx(c)
    if seq.upper() == "XYZ":
        return rx(a) @ ry(b) @ rz(c)
    raise ValueError(seq)


def quickTF(pts: np.ndarray, TF: np.ndarray) -> np.ndarray:
    """quickTF.m:5-7 -- [pts 1] * TF, first three columns."""
    pts = np.asarray(pts, dtype=np.float64)
    h = np.hstack([pts, np.ones((pts.shape[0], 1))])
    return (h @ TF)[:, :3]


def invertTF(TF: np.ndarray) -> np.ndarray:
    """invertTF.m:5-7 -- closed-form inverse of a row-vector rigid transform."""
    TFinv = np.eye(4)
    TFinv[:3, :3] = TF[:3, :3].T
    TFinv[3, :3] = -TF[3, :3] @ TF[:3, :3].T
    return TFinv


# ------------------------------------------------------------------ estimateTransform
def estimateTransform(pts1: np.ndarray, pts2: np.ndarray):
    """estimateTransform.m:8-71.  Returns 4x4 T with [pts2,1]*T = [pts1,1], or None for [].

    (The header comment of the .m file states the opposite direction; :65 and calcDists
    agree with the one implemented here -- SURVEY.md section 8a row 5.)
    """
    pts1 = np.asarray(pts1, dtype=np.float64)
    pts2 = np.asarray(pts2, dtype=np.float64)
    num_points = pts1.shape[0]                                  # :8
    if matlab_rank(pts1) < 3 or matlab_rank(pts2) < 2:          # :11
        return None                                             # :12-13
    if num_points == 3:                                         # :18
        c1 = pts1.sum(axis=0) / 3.0                             # :20 mean
        c2 = pts2.sum(axis=0) / 3.0                             # :21
        n1 = np.cross(pts1[2] - pts1[1], pts1[2] - pts1[0])     # :24
        n2 = np.cross(pts2[2] - pts2[1], pts2[2] - pts2[0])     # :25
        # :28-29  circshift(pts,1,1) moves the last row to the top
        l1 = float(np.median(np.linalg.norm(pts1 - np.roll(pts1, 1, axis=0), axis=1)))
        l2 = float(np.median(np.linalg.norm(pts2 - np.roll(pts2, 1, axis=0), axis=1)))
        p1 = c1 + (n1 / np.linalg.norm(n1)) * l1                # :32
        p2 = c2 + (n2 / np.linalg.norm(n2)) * l2                # :33
        pts1 = np.vstack([pts1, p1])                            # :35
        pts2 = np.vstack([pts2, p2])                            # :36
    d = pts1.T                                                  # :41
    m = pts2.T                                                  # :42
    cd = d.mean(axis=1, keepdims=True)                          # :46
    cm = m.mean(axis=1, keepdims=True)                          # :47
    d_c = d - cd                                                # :55
    m_c = m - cm                                                # :56
    H = m_c @ d_c.T                                             # :58
    U, _S, Vt = np.linalg.svd(H)                                # :60  H = U*S*V'
    R = Vt.T @ U.T                                              # :62  no reflection fix
    t = cd - R @ cm                                             # :63
    TF = np.eye(4)                                              # :66
    TF[:3, :3] = R                                              # :67
    TF[:3, 3] = t[:, 0]                                         # :68
    return TF.T                                                 # :71


def calcDists(T: np.ndarray, pts1: np.ndarray, pts2: np.ndarray) -> np.ndarray:
    """getInliersRANSAC.m:46-54 -- SQUARED distance of pts1 to [pts2,1]*T."""
    pts1 = np.asarray(pts1, dtype=np.float64)
    pts2 = np.asarray(pts2, dtype=np.float64)
    h = np.hstack([pts2, np.ones((pts2.shape[0], 1))])          # :50
    tr = (h @ T)[:, :3]                                         # :51-52
    return np.sum((pts1 - tr) ** 2, axis=1)                     # :53


# --------------------------------------------------------------------------- sampling
_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def sample_table(n: int, iter_num: int, min_pt_num: int, seed: int) -> np.ndarray:
    """The build's own counter-based sampler ("device RNG" mode of pcreg_ransac_*).

    NOT part of the reference (which draws ``randperm(ptNum)`` from MATLAB's global
    Mersenne-Twister stream, ransac.m:42-43, irreproducible here).  Draw j of
    hypothesis p (0-based) uses h = splitmix64(seed ^ splitmix64(p*16 + j)) and picks
    r = ((h >> 32) * (n - j)) >> 32 among the not-yet-chosen indices in ascending
    order (partial Fisher-Yates without replacement).  Returns 1-based int32
    [iter_num, min_pt_num].
    """
    out = np.empty((iter_num, min_pt_num), dtype=np.int32)
    for p in range(iter_num):
        chosen: list[int] = []
        for j in range(min_pt_num):
            h = _splitmix64((seed & _M64) ^ _splitmix64(p * 16 + j))
            r = ((h >> 32) * (n - j)) >> 32
            for c in sorted(chosen):
                if r >= c:
                    r += 1
            chosen.append(r)
            out[p, j] = r + 1
    return out


# ----------------------------------------------------------------------------- ransac
def ransac(pts1, pts2, ransacCoef: dict, sample_idx: np.ndarray | None = None,
           seed: int = 0, funcFindTransf=estimateTransform, funcDist=calcDists) -> dict:
    """ransac.m:21-116 with the sample indices as an input (1-based, iterNum x minPtNum).

    Returns a dict with the five outputs of the reference (T, inlierIdx, numSuccess,
    maxInliers, ratio) plus the per-iteration arrays for parity checking.
    """
    pts1 = np.asarray(pts1, dtype=np.float64)
    pts2 = np.asarray(pts2, dtype=np.float64)
    minPtNum = int(ransacCoef["minPtNum"])                      # :23
    iterNum = int(ransacCoef["iterNum"])                        # :24
    thInlrRatio = float(ransacCoef["thInlrRatio"])              # :25
    thDist = float(ransacCoef["thDist"])                        # :26
    ptNum = pts1.shape[0]                                       # :27
    thInlr = matlab_round(thInlrRatio * ptNum)                  # :28
    REFINE = bool(ransacCoef["REFINE"])                         # :29
    if sample_idx is None:
        sample_idx = sample_table(ptNum, iterNum, minPtNum, seed)
    sample_idx = np.asarray(sample_idx)
    assert sample_idx.shape == (iterNum, minPtNum)

    inlrNum = np.zeros(iterNum, dtype=np.int64)                 # :36
    inlrNum_refined = np.zeros(iterNum, dtype=np.int64)         # :37
    TForms: list = [None] * iterNum                             # :38
    for p in range(iterNum):                                    # :40
        s = sample_idx[p] - 1                                   # :42-43
        f1 = funcFindTransf(pts1[s], pts2[s])                   # :45
        if f1 is None:
            continue                                            # deviation: 0 inliers
        dist = funcDist(f1, pts1, pts2)                         # :48
        inlier1 = np.nonzero(dist < thDist)[0]                  # :49
        inlrNum[p] = inlier1.size                               # :50
        if inlier1.size >= thInlr:                              # :53
            if REFINE:                                          # :54
                f1_ref = funcFindTransf(pts1[inlier1], pts2[inlier1])   # :55
                if f1_ref is None:
                    continue
                dist = funcDist(f1_ref, pts1, pts2)             # :56
                inlrNum_refined[p] = int(np.sum(dist < thDist))  # :57-58
                if inlrNum_refined[p] >= thInlr:                # :59
                    TForms[p] = f1_ref                          # :60
            else:
                TForms[p] = f1                                  # :63
    counts = inlrNum_refined if REFINE else inlrNum             # :69-73
    idx = int(np.argmax(counts))                                # first max
    maxInliers = int(counts[idx])
    T = TForms[idx]                                             # :75
    if T is None:                                               # :77-89 FAILED
        return dict(T=None, inlierIdx=np.zeros(0, dtype=np.int64), numSuccess=0,
                    maxInliers=0, ratio=0.0, failed=True, inlrNum=inlrNum,
                    inlrNum_refined=inlrNum_refined, winner=idx, thInlr=thInlr)
    dist = funcDist(T, pts1, pts2)                              # :78
    inlierIdx = np.nonzero(dist < thDist)[0] + 1                # :92 (1-based)
    numSuccess = int(np.sum(counts >= thInlr))                  # :94-98
    return dict(T=T, inlierIdx=inlierIdx, numSuccess=numSuccess, maxInliers=maxInliers,
                ratio=100.0 * maxInliers / ptNum, failed=False, inlrNum=inlrNum,
                inlrNum_refined=inlrNum_refined, winner=idx, thInlr=thInlr)


GETINLIERS_COEFF = dict(minPtNum=3, iterNum=20000, thDist=0.5, thInlrRatio=0.1, REFINE=True)
"""getInliersRANSAC.m:17-31."""


def getInliersRANSAC(loc1M, loc1S, sample_idx=None, seed=0, coeff=None) -> dict:
    """getInliersRANSAC.m:12-42 as a function of the two workspace variables."""
    pts1 = np.asarray(loc1M, dtype=np.float64)                  # :12
    pts2 = np.asarray(loc1S, dtype=np.float64)                  # :13
    coeff = dict(GETINLIERS_COEFF if coeff is None else coeff)
    res = ransac(pts1, pts2, coeff, sample_idx=sample_idx, seed=seed)   # :34
    if res["T"] is not None:                                    # :39
        res["pts1_aligned"] = quickTF(pts1, res["T"])           # :40-41
    return res


# ----------------------------------------------------------------------- matchFeatures
_EPS_SINGLE = float(np.finfo(np.float32).eps)


def _normalize_rows(X: np.ndarray) -> np.ndarray:
    """matchFeatures' normalizeX: unit L2 rows; effectively-zero rows become 0."""
    nrm = np.sqrt(np.sum(X * X, axis=1))
    out = X / np.where(nrm > 0, nrm, 1.0)[:, None]
    out[nrm <= _EPS_SINGLE] = 0.0
    return out


def match_threshold(pct: float, D: int, metric: str) -> float:
    """percentToLevel: percent of the largest distance between two unit vectors."""
    max_val = 4.0 if metric.upper() == "SSD" else 2.0 * math.sqrt(D)
    return (pct * 0.01) * max_val


def _scores_block(A: np.ndarray, B: np.ndarray, metric: str) -> np.ndarray:
    if metric.upper() == "SAD":
        return np.abs(A[:, None, :] - B[None, :, :]).sum(axis=2)
    return ((A[:, None, :] - B[None, :, :]) ** 2).sum(axis=2)


def _top2_rows(F1, F2, metric, qblock=64, mblock=4096):
    Q, M = F1.shape[0], F2.shape[0]
    i1 = np.zeros(Q, dtype=np.int64)
    d1 = np.full(Q, np.inf)
    i2 = np.full(Q, -1, dtype=np.int64)
    d2 = np.full(Q, np.inf)
    for q0 in range(0, Q, qblock):
        A = F1[q0:q0 + qblock]
        S = np.concatenate([_scores_block(A, F2[m0:m0 + mblock], metric)
                            for m0 in range(0, M, mblock)], axis=1)
        a1 = np.argmin(S, axis=1)                       # first occurrence on ties
        r = np.arange(S.shape[0])
        i1[q0:q0 + qblock] = a1
        d1[q0:q0 + qblock] = S[r, a1]
        if M > 1:
            S2 = S.copy()
            S2[r, a1] = np.inf
            a2 = np.argmin(S2, axis=1)
            i2[q0:q0 + qblock] = a2
            d2[q0:q0 + qblock] = S[r, a2]
    return i1, d1, i2, d2


def _colbest(F1, F2cols, metric, mblock=64, qblock=4096):
    """For each selected model row, the first query row with the smallest score."""
    P, Q = F2cols.shape[0], F1.shape[0]
    best = np.zeros(P, dtype=np.int64)
    for p0 in range(0, P, mblock):
        B = F2cols[p0:p0 + mblock]
        S = np.concatenate([_scores_block(B, F1[q0:q0 + qblock], metric)
                            for q0 in range(0, Q, qblock)], axis=1)
        best[p0:p0 + mblock] = np.argmin(S, axis=1)
    return best


def matchFeatures(features1, features2, Method="Exhaustive", MatchThreshold=10.0,
                  MaxRatio=0.6, Metric="SSD", Unique=False, Prenormalized=False,
                  return_all=False):
    """Documented semantics of MathWorks ``matchFeatures`` (called at getMatches.m:51-56).

    PARITY UNPINNED: closed source, absent from the reference tree, no reference test
    pins it.  Encoded here: (1) rows L2-normalised unless Prenormalized; (2) SAD =
    sum|a-b|, SSD = sum (a-b)^2; (3) per query the two smallest scores (first index on
    ties); (4) keep if best <= MatchThreshold% of the max distance of unit vectors
    (SSD 4, SAD 2*sqrt(D)); (5) ratio test best/second <= MaxRatio, with second < 1e-6
    forcing both to 1; skipped when there is a single model row; (6) Unique: keep (i,j)
    only if i is the first-best query of model row j over ALL queries; (7) pairs in
    ascending query order, 1-based uint32.  'Approximate' is answered with the exact
    search.
    """
    F1 = np.asarray(features1, dtype=np.float64)
    F2 = np.asarray(features2, dtype=np.float64)
    if not Prenormalized:
        F1 = _normalize_rows(F1)
        F2 = _normalize_rows(F2)
    Q, D = F1.shape
    M = F2.shape[0]
    if Q == 0 or M == 0:
        z = np.zeros((0, 2), dtype=np.uint32)
        return (z, np.zeros(0)) if not return_all else (z, np.zeros(0), None)
    i1, d1, i2, d2 = _top2_rows(F1, F2, Metric)
    keep = d1 <= match_threshold(MatchThreshold, D, Metric)
    if M > 1:
        t1 = d1.copy()
        t2 = d2.copy()
        z = t2 < 1e-6
        t1[z] = 1.0
        t2[z] = 1.0
        keep &= (t1 / t2) <= MaxRatio
    qi = np.nonzero(keep)[0]
    mj = i1[qi]
    if Unique and qi.size:
        cb = _colbest(F1, F2[mj], Metric)
        u = cb == qi
        qi, mj = qi[u], mj[u]
    pairs = np.stack([qi + 1, mj + 1], axis=1).astype(np.uint32)
    if return_all:
        return pairs, d1[qi], dict(i1=i1, d1=d1, i2=i2, d2=d2)
    return pairs, d1[qi]


def preprocess_descriptors(descSurface, descModel, par: dict):
    """getMatches.m:22-41 -- the two element-wise steps before matchFeatures."""
    dS = np.asarray(descSurface, dtype=np.float64)
    dM = np.asarray(descModel, dtype=np.float64)
    if par.get("UNNORMALIZE", False):                           # :22
        avg = np.mean(np.sum(np.abs(np.vstack([dS, dM])), axis=1))   # :24
        col = float(par["norm_factor"]) * avg
        dS = np.hstack([dS, np.full((dS.shape[0], 1), col)])    # :25
        dM = np.hstack([dM, np.full((dM.shape[0], 1), col)])    # :26
    if par.get("CHANGE_METRIC", False):                         # :35
        dS = dS ** float(par["metric_factor"])                  # :36
        dM = dM ** float(par["metric_factor"])                  # :37
    return dS, dM


def getMatches(descSurface, descModel, par: dict) -> np.ndarray:
    """getMatches.m:1-59 -> P x 2 uint32, 1-based [surfaceIdx, modelIdx]."""
    dS, dM = preprocess_descriptors(descSurface, descModel, par)
    pairs, _ = matchFeatures(dS, dM, Method=par.get("Method", "Exhaustive"),
                             MatchThreshold=par["MatchThreshold"], MaxRatio=par["MaxRatio"],
                             Metric=par["Metric"], Unique=par["Unique"])   # :51-56
    return pairs


# ------------------------------------------------------------------------ AlignPoints
def pca_eig(X: np.ndarray, centered: bool = True):
    """Documented semantics of MathWorks ``pca(X,'Algorithm','eig'[,'Centered','off'])``.

    PARITY UNPINNED (closed source; AlignPoints_KNN.m:31,33).  coeff = eigenvectors of
    the covariance by descending eigenvalue, each column flipped so its
    largest-magnitude entry is positive; score = (X - mu) * coeff; latent = eigenvalues.
    """
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    mu = X.mean(axis=0) if centered else np.zeros(X.shape[1])
    Xc = X - mu
    dof = (n - 1) if centered else n
    C = (Xc.T @ Xc) / max(dof, 1)
    w, V = np.linalg.eigh(C)
    order = np.argsort(-w, kind="stable")
    w, V = w[order], V[:, order]
    mi = np.argmax(np.abs(V), axis=0)
    sg = np.sign(V[mi, np.arange(V.shape[1])])
    sg[sg == 0] = 1.0
    V = V * sg
    return V, Xc @ V, w


def AlignPoints_KNN(pts, C1: bool = False, C2: bool = False):
    """AlignPoints_KNN.m:8-59 -> (pts_aligned, coeff_unambig, c)."""
    pts = np.asarray(pts, dtype=np.float64)
    N = pts.shape[0]
    c = pts.mean(axis=0)                                        # :17
    K = matlab_round(N * 0.85)                                  # :20-21
    pts_rel = pts - c                                           # :22
    dists = np.sqrt(np.sum(pts_rel * pts_rel, axis=1))          # :23
    I = np.argsort(dists, kind="stable")                        # :24
    pts_k = pts_rel[I][:K]                                      # :25-26
    coeff, pts_lrf, _ = pca_eig(pts_k, centered=not C1)         # :30-34
    k = N                                                       # :37
    if C2:
        pts_lrf = pts @ coeff                                   # :39-41
    x_sign = 1.0 if np.sum(pts_lrf[:, 0] > 0) >= k / 2 else -1.0   # :45,49
    z_sign = 1.0 if np.sum(pts_lrf[:, 2] > 0) >= k / 2 else -1.0   # :46,50
    y_sign = float(np.linalg.det(coeff * np.array([x_sign, 1.0, z_sign])))   # :53
    coeff_unambig = coeff * np.array([x_sign, y_sign, z_sign])  # :56
    return pts @ coeff_unambig, coeff_unambig, c                # :59


def getLocalPoints(pts, R, c, min_points, max_points, single_mode: int = 0):
    """getLocalPoints.m:8-35 -> (pts_sphere relative to c, dists) or (None, None).

    single_mode != 0: MATLAB's arithmetic when either input is `single` (clouds from pcread are: upsampleMesh.m:21,
    GetPointcloudFromModel.m:269; completeExperimentFast.m:291,309 feeds them on).  A binary operation on a single and a
    double operand is carried out in single (the double is converted first), so every step below is element-wise float32,
    each operation rounded once -- reproducible, unlike the summation order of mean / pca that follows:
      1 = keypoint c single: xLim = c(1) + [-R, R] is single (:8-10, R converted to single first);
      2 = cloud single, c double: xLim is formed in double and converted in the comparison with the single cloud (:11-13).
    pts_rel = pts_cube - c (:23), vecnorm = sqrt(x^2 + y^2 + z^2) (:24) and dists < R (:25) are single either way; the
    returned pts_sphere are those single values (widened)."""
    if single_mode:
        f32 = np.float32
        p = np.asarray(pts).astype(f32)                              # a double cloud meets single limits: converted in the comparison
        c64 = np.asarray(c, dtype=np.float64)
        c32 = c64.astype(f32)
        R32 = f32(R)
        if single_mode == 1:
            lo, hi = c32 + f32(-R32), c32 + R32                        # :8-10 in single
        else:
            lo, hi = (c64 - R).astype(f32), (c64 + R).astype(f32)      # :8-10 in double, converted at :11-13
        m = np.all((p > lo) & (p < hi), axis=1)                        # :11-13 open box
        cube = p[m]
        if cube.shape[0] < min_points:                                 # :17
            return None, None
        rel = cube - c32                                               # :23 (single)
        d = np.sqrt((rel[:, 0] * rel[:, 0] + rel[:, 1] * rel[:, 1]) + rel[:, 2] * rel[:, 2])   # :24, each step a float32 op
        k = d < R32                                                    # :25
        if k.sum() < min_points or k.sum() > max_points:               # :31
            return None, None
        return rel[k].astype(np.float64), d[k].astype(np.float64)
    pts = np.asarray(pts, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    m = np.all((pts > c - R) & (pts < c + R), axis=1)           # :8-13 open box
    cube = pts[m]
    if cube.shape[0] < min_points:                              # :17
        return None, None
    rel = cube - c                                              # :23
    d = np.sqrt(np.sum(rel * rel, axis=1))                      # :24
    k = d < R                                                   # :25
    if k.sum() < min_points or k.sum() > max_points:            # :31
        return None, None
    return rel[k], d[k]


# ------------------------------------------------- getSpacialHistogramDescriptors (cfg 4)
NUM_R, NUM_THETA, NUM_PHI = 10, 7, 14          # getSpacialHistogramDescriptors.m:38-40


def histogram_edges(R: float):
    """Bin edges of getSpacialHistogramDescriptors.m:155-158 as a + k*step (the colon
    operator's rounding of the interior edges is not reproducible here; only points
    lying exactly on an edge could tell the difference)."""
    r3 = float(R) ** 3
    r_bins = np.cbrt(np.array([k * (r3 / NUM_R) for k in range(NUM_R + 1)]))            # nthroot(0:R^3/10:R^3, 3)
    theta_bins = np.array([k * (math.pi / NUM_THETA) for k in range(NUM_THETA + 1)])     # 0:pi/7:pi
    phi_bins = np.array([-math.pi + k * (2 * math.pi / NUM_PHI) for k in range(NUM_PHI + 1)])   # -pi:2pi/14:pi
    return r_bins, theta_bins, phi_bins


def histcounts_loc(x: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """Bin index (1-based, 0 = outside/NaN) as MATLAB histcounts returns it to histcn.m:108:
    edges(k) <= x < edges(k+1), the last bin also contains x == edges(end)."""
    x = np.asarray(x, dtype=np.float64)
    loc = np.searchsorted(edges, x, side="right")
    loc = np.where(x == edges[-1], len(edges) - 1, loc)
    bad = np.isnan(x) | (x < edges[0]) | (x > edges[-1])
    return np.where(bad, 0, loc).astype(np.int64)


def getSpacialHistogramDescriptors(pts, sample_pts, options: dict, single_mode: int = 0):
    """getSpacialHistogramDescriptors.m:18-179 (+ histcn.m:94-131) -> (feat V x 3, desc V x 980), both double whatever
    the input classes (the reference preallocates them with nan(...), :61-62).

    single_mode (see getLocalPoints): which keypoints survive and which points form a support follow MATLAB's single
    arithmetic, and the support's coordinates are the single pts_rel values; everything after getLocalPoints is evaluated in
    double on those values -- MATLAB would continue in single, with a summation order (mean, pca) that is not knowable.

    Restated quirks: the LRF sign vote uses k = K rows (:125,129-130); phi = atan2(y, y)
    (:152); NORMALIZE and LOCAL_PCA are hard-wired off (:31-32); points with r == 0 give
    NaN angles and are dropped by histcn.  Deviation: a support whose K nearest points are
    fewer than 2 is skipped (the reference would index an empty `variances`)."""
    pts = np.asarray(pts, dtype=np.float64)
    sample_pts = np.asarray(sample_pts, dtype=np.float64)
    if single_mode:
        pts = pts.astype(np.float32).astype(np.float64)                          # (already single values, or rounded in the comparison)
        if single_mode == 1:
            sample_pts = sample_pts.astype(np.float32).astype(np.float64)
    min_pts, max_pts = options["min_pts"], options["max_pts"]                   # :18-19
    R, thVar, K, ALIGN = float(options["R"]), options["thVar"], options["k"], bool(options["ALIGN_POINTS"])
    r_bins, theta_bins, phi_bins = histogram_edges(R)
    feat, desc = [], []
    for c in sample_pts:                                                        # :48-54 and :64-174 fused
        pts_local, _ = getLocalPoints(pts, R, c, min_pts, max_pts, single_mode) # :50, :68
        if pts_local is None or pts_local.shape[0] == 0:
            continue
        n = pts_local.shape[0]                                                  # :71
        if K == "all" or K == 1:                                                # :75-76
            k = n
        else:
            k = matlab_round(n * K)                                             # :78
            centroid = pts_local.mean(axis=0)                                   # :80
            d = np.sqrt(np.sum((pts_local - centroid) ** 2, axis=1))            # :81
            pts_local = pts_local[np.argsort(d, kind="stable")]                 # :82-83
        if k < 2:
            continue
        pts_k = pts_local[:k]                                                   # :85
        coeff, pts_lrf, variances = pca_eig(pts_k)                              # :91
        if variances[0] / variances[1] < thVar[0] or variances[1] / variances[2] < thVar[1]:   # :118-121
            continue
        kk = pts_lrf.shape[0]                                                   # :125
        if ALIGN:                                                               # :128-145
            x_sign = 1.0 if np.sum(pts_lrf[:, 0] > 0) >= kk / 2 else -1.0
            z_sign = 1.0 if np.sum(pts_lrf[:, 2] > 0) >= kk / 2 else -1.0
            y_sign = float(np.linalg.det(coeff * np.array([x_sign, 1.0, z_sign])))
            pts_local = pts_local @ (coeff * np.array([x_sign, y_sign, z_sign]))
        with np.errstate(invalid="ignore", divide="ignore"):
            r = np.sqrt(np.sum(pts_local * pts_local, axis=1))                  # :150
            theta = np.arccos(pts_local[:, 2] / r)                              # :151
            phi = np.arctan2(pts_local[:, 1], pts_local[:, 1])                  # :152 (sic)
        lr, lt, lp = histcounts_loc(r, r_bins), histcounts_loc(theta, theta_bins), histcounts_loc(phi, phi_bins)
        ok = (lr > 0) & (lt > 0) & (lp > 0)                                     # histcn.m:126
        counts = np.zeros(NUM_R * NUM_THETA * NUM_PHI)
        flat = (lr[ok] - 1) + NUM_R * (lt[ok] - 1) + NUM_R * NUM_THETA * (lp[ok] - 1)   # column-major reshape (:164)
        np.add.at(counts, flat, 1.0)
        desc.append(counts)                                                     # :171
        feat.append(c)                                                          # :172
    if not feat:
        return np.zeros((0, 3)), np.zeros((0, NUM_R * NUM_THETA * NUM_PHI))
    return np.array(feat), np.array(desc)


# --------------------------------------------- fp32 point KNN (the bench's D=3 search)
def knn2_points_f32(q: np.ndarray, m: np.ndarray, block: int = 2048):
    """Two nearest model points per query, float32 arithmetic, squared distance
    d = fma(dz,dz, fma(dy,dy, dx*dx)) with dx = q.x - m.x (each step rounded to
    float32; the fma steps are emulated in float64, which is exact for float32
    operands: a 24x24-bit product plus a float32 addend fits in 53 bits only
    approximately, so the result is rounded twice -- see oracle C for the
    bit-exact fmaf version used by the parity tests).

    This numpy version is a slow cross-check of the C oracle, used on small inputs.
    Ties resolve to the lowest model index.  Returns (idx [Q,2] int64 0-based,
    dist [Q,2] float32); second column is (-1, inf) when M == 1.
    """
    q = np.asarray(q, dtype=np.float32)
    m = np.asarray(m, dtype=np.float32)
    Q, M = q.shape[0], m.shape[0]
    idx = np.full((Q, 2), -1, dtype=np.int64)
    dist = np.full((Q, 2), np.inf, dtype=np.float32)
    for q0 in range(0, Q, block):
        A = q[q0:q0 + block]
        dx = (A[:, None, 0] - m[None, :, 0]).astype(np.float32)
        dy = (A[:, None, 1] - m[None, :, 1]).astype(np.float32)
        dz = (A[:, None, 2] - m[None, :, 2]).astype(np.float32)
        s = (dx * dx).astype(np.float32)
        s = (dy.astype(np.float64) * dy.astype(np.float64) + s.astype(np.float64)).astype(np.float32)
        s = (dz.astype(np.float64) * dz.astype(np.float64) + s.astype(np.float64)).astype(np.float32)
        r = np.arange(A.shape[0])
        a1 = np.argmin(s, axis=1)
        idx[q0:q0 + block, 0] = a1
        dist[q0:q0 + block, 0] = s[r, a1]
        if M > 1:
            s2 = s.copy()
            s2[r, a1] = np.inf
            a2 = np.argmin(s2, axis=1)
            idx[q0:q0 + block, 1] = a2
            dist[q0:q0 + block, 1] = s[r, a2]
    return idx, dist


def match_points_f32(q, m, match_threshold_abs, max_ratio, unique=True):
    """matchFeatures filter chain on raw 3-D points (Prenormalized, SSD, absolute
    threshold): the bench's cfg-2 pipeline.  1-based pairs."""
    idx, dist = knn2_points_f32(q, m)
    d1 = dist[:, 0].astype(np.float32)
    d2 = dist[:, 1].astype(np.float32)
    keep = d1 <= np.float32(match_threshold_abs)
    if m.shape[0] > 1:
        z = d2 < np.float32(1e-6)
        t1 = np.where(z, np.float32(1), d1)
        t2 = np.where(z, np.float32(1), d2)
        keep &= (t1 / t2).astype(np.float32) <= np.float32(max_ratio)
    qi = np.nonzero(keep)[0]
    mj = idx[qi, 0]
    if unique and qi.size:
        back, _ = knn2_points_f32(np.asarray(m, dtype=np.float32)[mj], q)
        u = back[:, 0] == qi
        qi, mj = qi[u], mj[u]
    return np.stack([qi + 1, mj + 1], axis=1).astype(np.uint32)


# ------------------------------------------------------------------ sphere-sweep driver
def pcUniformSamples(pts: np.ndarray, d: float) -> np.ndarray:
    """completeExperimentFast.m:406-414 -- meshgrid over the cloud's X/Y/ZLimits with pitch d;
    MATLAB's meshgrid + column-major reshape put y fastest, then x, then z."""
    pts = np.asarray(pts, dtype=np.float64)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    ax = [lo[k] + d * np.arange(int(np.floor((hi[k] - lo[k]) / d + 1e-12)) + 1) for k in range(3)]   # a:d:b
    X, Y, Z = np.meshgrid(ax[0], ax[1], ax[2])                  # shape (ny, nx, nz), like MATLAB
    return np.column_stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")])


def getDescriptorMask(featModel: np.ndarray, current_center, R_desc: float, margin: float = 0.0) -> np.ndarray:
    """completeExperimentFast.m:435-439."""
    rel = np.asarray(featModel, dtype=np.float64) - np.asarray(current_center, dtype=np.float64)
    dists = np.sqrt((rel[:, 0] * rel[:, 0] + rel[:, 1] * rel[:, 1]) + rel[:, 2] * rel[:, 2])
    return dists < (R_desc + margin)


def sphere_sweep(featModel, descModel, featSurface, descSurface, par: dict, options: dict, R_desc: float,
                 d_spheres: float = 5.0, min_pts: int = 1400, putative_thresh: int = 170, seed: int = 0,
                 get_matches=None, run_ransac=None) -> dict:
    """completeExperimentFast.m:46-224 on the CPU: sphere centres, validity by descriptor count,
    getMatches per sphere, putative threshold, ransac per promising sphere, the stats arrays.
    `seed + trial index` seeds the built-in sampler of each ransac call."""
    get_matches = get_matches or getMatches
    run_ransac = run_ransac or ransac
    featModel = np.asarray(featModel, dtype=np.float64); featSurface = np.asarray(featSurface, dtype=np.float64)
    centres = pcUniformSamples(featModel, d_spheres)                                     # :49-50
    num_desc = np.array([int(getDescriptorMask(featModel, c, R_desc).sum()) for c in centres], dtype=np.int64)
    valid = num_desc >= min_pts                                                          # :57-64 (max_pts = inf)
    centres = centres[valid]
    S = len(centres)
    num_putative = np.zeros(S, dtype=np.int64); num_desc_all = np.zeros(S, dtype=np.int64)
    matches_cells, idx_cells = [], []
    for i in range(S):                                                                   # :109-149
        mask = getDescriptorMask(featModel, centres[i], R_desc, 0.0)
        idx = np.nonzero(mask)[0]
        m = get_matches(descSurface, np.asarray(descModel)[idx], dict(par, VERBOSE=0))
        num_putative[i] = len(m); num_desc_all[i] = len(idx)
        matches_cells.append(m); idx_cells.append(idx)
    trial = np.nonzero(num_putative > putative_thresh)[0]                                # :175
    stats = dict(putative=[], success=[], inliers=[], ratio=[], T=[])
    for t, i in enumerate(trial):                                                        # :200-224
        m = matches_cells[i]
        pts1 = featSurface[m[:, 0].astype(np.int64) - 1]
        pts2 = featModel[idx_cells[i]][m[:, 1].astype(np.int64) - 1]
        r = run_ransac(pts1, pts2, dict(options, VERBOSE=0), seed=seed + t)
        stats["putative"].append(len(m)); stats["success"].append(r["numSuccess"]); stats["inliers"].append(r["maxInliers"])
        stats["ratio"].append(r["ratio"]); stats["T"].append(r["T"])
    return dict(centres=centres, num_desc=num_desc_all, num_putative=num_putative, matches=matches_cells,
                model_rows=idx_cells, trial=trial,
                statsPutative=np.array(stats["putative"], dtype=np.int64), statsSuccess=np.array(stats["success"], dtype=np.int64),
                statsInliers=np.array(stats["inliers"], dtype=np.int64), statsRatio=np.array(stats["ratio"], dtype=np.float64),
                transforms=stats["T"])


def refine_by_distance(pts1, pts2, maxDist: float):
    """completeExperimentFast.m:383-391: inliers by vecnorm(pts1 - pts2) < maxDist, estimateTransform on them.
    Returns (T or None, inlier_idx 0-based)."""
    pts1 = np.asarray(pts1, dtype=np.float64); pts2 = np.asarray(pts2, dtype=np.float64)
    d = pts1 - pts2
    d1 = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    inl = np.nonzero(d1 < maxDist)[0]
    T = estimateTransform(pts1[inl], pts2[inl]) if len(inl) >= 3 else None
    return T, inl


def pcRandomUniformSamples(pts, d: float, margin: float, rng=None) -> np.ndarray:
    """completeExperimentFast.m:416-429: round(volume of the margin-padded bounding box / d^3) uniform random keypoints
    in that box.  MATLAB's rand stream is unknowable here: `rng` (numpy Generator) stands in, so only the COUNT and the
    BOX are pinned; the final stage below takes the keypoints as an input for that reason."""
    pts = np.asarray(pts, dtype=np.float64)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    rng_xyz = (hi - lo) + 2.0 * margin
    num_pts = matlab_round(float(rng_xyz[0] * rng_xyz[1] * rng_xyz[2]) / (d ** 3))
    rng = rng or np.random.default_rng(0)
    return rng.random((num_pts, 3)) * rng_xyz + lo - margin


def final_stage(ptsSurface, clusters, sample_pts, featModel_noLRF, descModel_noLRF, R_desc: float, descOpt: dict, par: dict,
                maxDist: float = 1.5, get_descriptors=None, get_matches=None) -> dict:
    """completeExperimentFast.m:280-394 on the CPU.  clusters = [(locCur, transCur)]: the middle of every cluster of promising
    spheres and the RANSAC transform of the sphere closest to it (:283-291; clusterPoints itself is host bookkeeping over a
    few dozen points and not restated).  sample_pts[i]: the keypoints the script draws at random for cluster i (:297).

    Per cluster (:293-353): quickTF(surface, invertTF(transCur)); descriptors WITHOUT local alignment (ALIGN_POINTS = false);
    the model descriptors inside the sphere of radius R_desc around locCur; getMatches.  Then (:357-378) the share of
    matches closer than maxDist, (:381) the cluster with the largest share (MATLAB's max: first maximum, NaN skipped),
    (:383-394) T_refine = estimateTransform over its close matches and the final surface quickTF(pts_tform, invertTF(T_refine))."""
    get_descriptors = get_descriptors or getSpacialHistogramDescriptors
    get_matches = get_matches or getMatches
    ptsSurface = np.asarray(ptsSurface, dtype=np.float64)
    featModel_noLRF = np.asarray(featModel_noLRF, dtype=np.float64); descModel_noLRF = np.asarray(descModel_noLRF, dtype=np.float64)
    opt = dict(descOpt, ALIGN_POINTS=False, VERBOSE=0)                                       # :300 "this false is the key"
    per = []
    for (locCur, transCur), kp in zip(clusters, sample_pts):
        pts_tform = quickTF(ptsSurface, invertTF(np.asarray(transCur, dtype=np.float64)))    # :291
        feat, desc = get_descriptors(pts_tform, np.asarray(kp, dtype=np.float64), opt)       # :309-310
        mask = getDescriptorMask(featModel_noLRF, np.asarray(locCur, dtype=np.float64), R_desc, 0.0)   # :319
        featCur, descCur = featModel_noLRF[mask], descModel_noLRF[mask]
        if len(feat) and len(featCur):
            m = get_matches(desc, descCur, dict(par, VERBOSE=0))                             # :344
        else:
            m = np.zeros((0, 2), dtype=np.uint32)
        pts1 = feat[m[:, 0].astype(np.int64) - 1]; pts2 = featCur[m[:, 1].astype(np.int64) - 1]   # :366-367
        dd = pts1 - pts2
        d1 = np.sqrt((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2])
        inl = np.nonzero(d1 < maxDist)[0]                                                    # :369
        prec = (len(inl) / len(d1) * 100.0) if len(d1) else float("nan")                     # :373 (0/0 = NaN)
        per.append(dict(pts_tform=pts_tform, feat=feat, featCur=featCur, matches=m, inliers=inl, precision=prec, pts1=pts1, pts2=pts2))
    precisions = np.array([c["precision"] for c in per], dtype=np.float64)
    best = 0 if np.all(np.isnan(precisions)) else int(np.nanargmax(precisions))             # :381
    b = per[best]
    T_refine = estimateTransform(b["pts1"][b["inliers"]], b["pts2"][b["inliers"]]) if len(b["inliers"]) >= 3 else None   # :391
    pts_final = quickTF(b["pts_tform"], invertTF(T_refine)) if T_refine is not None else b["pts_tform"]                 # :394
    return dict(per_cluster=per, precisions=precisions, best=best, T_refine=T_refine, pts_final=pts_final)


def speedy_regions(pts, max_region_size: float):
    """speedyDescriptors.m:17-27: the cloud's bounding box cut into ceil(range / max_region_size) equal cuboids per axis.
    -> (bounds per axis: lo : step : hi as MATLAB's colon builds it, number of regions per axis)."""
    pts = np.asarray(pts)
    lo, hi = pts.min(axis=0).astype(np.float64), pts.max(axis=0).astype(np.float64)
    rng_xyz = hi - lo                                                        # :19
    n = np.ceil(rng_xyz / max_region_size).astype(np.int64)                  # :20
    step = rng_xyz / n                                                       # :21
    # :24-26  a : s : b  has floor((b - a) / s + tol) + 1 elements a + k s (MATLAB's colon tolerates rounding of (b - a) / s)
    bounds = [lo[k] + step[k] * np.arange(int(np.floor((hi[k] - lo[k]) / step[k] + 1e-10)) + 1) for k in range(3)]
    assert [len(b) - 1 for b in bounds] == n.tolist()                        # :27
    return bounds, n


def speedyDescriptors(pts, sample_opts: dict, options: dict, rng=None, get_descriptors=None):
    """speedyDescriptors.m:10-82 as the reference runs it: region by region (x outermost, z innermost, :44-46), the crop with a
    margin of R (open box, :48-52), round(volume of the crop's R-shrunk bounding box / d^3) uniform keypoints when the crop
    holds more than 500 points (:55, :86-101), ONE getSpacialHistogramDescriptors call per region on the CROP (:58-59), results
    stacked (:62-63).  MATLAB's rand stream is unknowable here: `rng` (numpy Generator; rand(n, 3) -> rng.random((n, 3))) stands
    in.  -> (feat, desc, all sampled keypoints in region order)."""
    get_descriptors = get_descriptors or getSpacialHistogramDescriptors
    rng = rng or np.random.default_rng(0)
    pts = np.asarray(pts)
    d, R = float(sample_opts["d"]), float(options["R"])
    bounds, n = speedy_regions(pts, float(options["max_region_size"]))
    feats, descs, kps = [], [], []
    for ix in range(n[0]):
        for iy in range(n[1]):
            for iz in range(n[2]):
                b = (bounds[0][ix], bounds[0][ix + 1], bounds[1][iy], bounds[1][iy + 1], bounds[2][iz], bounds[2][iz + 1])
                mask = ((pts[:, 0] > b[0] - R) & (pts[:, 0] < b[1] + R) & (pts[:, 1] > b[2] - R) & (pts[:, 1] < b[3] + R) &
                        (pts[:, 2] > b[4] - R) & (pts[:, 2] < b[5] + R))                                          # :48-50
                crop = pts[mask]
                if crop.shape[0] <= 500:                                                                           # :87, else [] (:99)
                    continue
                spts = pcRandomUniformSamples(crop, d, -R, rng)                                                     # :55
                kps.append(spts)
                if spts.shape[0] == 0:
                    continue
                f, dsc = get_descriptors(crop, spts, options)                                                      # :58-59
                feats.append(f); descs.append(dsc)
    feat = np.vstack(feats) if feats else np.zeros((0, 3))
    desc = np.vstack(descs) if descs else np.zeros((0, 980))
    return feat, desc, (np.vstack(kps) if kps else np.zeros((0, 3)))
