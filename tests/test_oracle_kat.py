"""Pins the oracle (numpy AND plain-C restatements) against every known answer the
reference's own test scripts hold for this path and against
each other.  CPU only."""
import numpy as np
import pytest

from conftest import rigid_case

PTS4 = np.array([[1, 5, 7], [4, 9, 3], [9, 3, 4], [1, 2, 4]], float)        # testTransformEstimation.m:2-5


def _impls(oracle_py, oracle_c):
    return [("numpy", oracle_py.estimateTransform), ("c", oracle_c.estimateTransform)]


def test_eul2rotm_conventions(oracle_py):
    R = oracle_py.eul2rotm([0.1, 0.2, 0.3])          # default 'ZYX'
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15) and np.isclose(np.linalg.det(R), 1.0)
    # ZYX: first angle about z.  A pure first angle rotates x towards y.
    Rz = oracle_py.eul2rotm([0.5, 0, 0])
    assert np.allclose(Rz @ [1, 0, 0], [np.cos(0.5), np.sin(0.5), 0])
    Rx = oracle_py.eul2rotm([0.5, 0, 0], "XYZ")
    assert np.allclose(Rx @ [0, 1, 0], [0, np.cos(0.5), np.sin(0.5)])


def test_testTransformEstimation_known_answer(oracle_py, oracle_c):
    """testTransformEstimation.m:2-17: estimateTransform(pts_tf, pts) == [R 0; t 1]."""
    R = oracle_py.eul2rotm([0.1, 0.2, 0.3]); t = np.array([1.0, 2.0, 3.0])
    pts_tf = PTS4 @ R + t
    Texp = np.eye(4); Texp[:3, :3] = R; Texp[3, :3] = t
    for name, est in _impls(oracle_py, oracle_c):
        assert np.abs(est(pts_tf, PTS4) - Texp).max() < 1e-13, name
        assert np.abs(est(pts_tf[:3], PTS4[:3]) - Texp).max() < 1e-13, name     # N == 3 branch (:18-37)
        assert np.abs(est(PTS4, pts_tf) - oracle_py.invertTF(Texp)).max() < 1e-13, name
        # the contract [pts2, 1] * T = [pts1, 1]   (estimateTransform.m:65, calcDists)
        assert np.abs(oracle_py.quickTF(PTS4, est(pts_tf, PTS4)) - pts_tf).max() < 1e-12


def test_estimate_transform_rank_rule(oracle_py, oracle_c):
    """estimateTransform.m:11-14: rank(pts1) < 3 || rank(pts2) < 2 -> []."""
    flat = PTS4.copy(); flat[:, 2] = 0.0
    line = np.outer(np.arange(1, 5), [1.0, 2.0, 3.0])
    for name, est in _impls(oracle_py, oracle_c):
        assert est(flat, PTS4) is None, name          # pts1 spans a plane through the origin
        assert est(PTS4, flat) is not None, name      # pts2 only needs rank 2
        assert est(PTS4, line) is None, name          # rank(pts2) = 1
        assert est(PTS4[:2], PTS4[:2]) is None, name
    assert oracle_c.rank_nx3(PTS4) == 3 and oracle_c.rank_nx3(flat) == 2 and oracle_c.rank_nx3(line) == 1
    assert oracle_py.matlab_rank(PTS4) == 3 and oracle_py.matlab_rank(flat) == 2


def test_matlab_round(oracle_py, oracle_c):
    for x, r in [(0.5, 1), (1.5, 2), (2.5, 3), (-0.5, -1), (2.4999, 2), (100 * 0.085, 9), (80.0, 80)]:
        assert oracle_py.matlab_round(x) == r
        assert oracle_c.lib().orc_matlab_round(__import__("ctypes").c_double(x)) == r


def test_testRANSAC_closed_forms(oracle_py, oracle_c):
    """testRANSAC.m:13-48: T_true2 / T_back closed forms, error2 = error3 = 0, and RANSAC on
    loc1M = pts + N(0,0.1^2), loc1S = pts*R+t with getInliersRANSAC.m:17-31 recovers T_back."""
    rng = np.random.default_rng(1)
    pts = rng.uniform([-3, -2, 0], [3, 2, 3], (1000, 3))       # teapot-sized stand-in (teapot.ply is MATLAB's)
    R = oracle_py.eul2rotm([1.5, -1.2, 0.8]); t = np.array([1.0, 2.0, 3.0])
    T_true2 = np.eye(4); T_true2[:3, :3] = R; T_true2[3, :3] = t
    T_back = np.eye(4); T_back[:3, :3] = R.T; T_back[3, :3] = -t @ R.T
    pts_tf = pts @ R + t
    assert np.abs(oracle_py.quickTF(pts, T_true2) - pts_tf).max() < 1e-13                 # error2
    assert np.abs(oracle_py.quickTF(pts_tf, T_back) - pts).max() < 1e-13                  # error3
    assert np.abs(oracle_py.invertTF(T_true2) - T_back).max() < 1e-15
    # estimateTransform(pts, pts_tf) maps pts_tf -> pts, i.e. equals T_back (the script's error1 is stale)
    assert np.abs(oracle_c.estimateTransform(pts, pts_tf) - T_back).max() < 1e-12
    loc1M = pts + np.random.default_rng(2).normal(0, 0.1, pts.shape)
    coeff = dict(oracle_py.GETINLIERS_COEFF, iterNum=500)
    for r in (oracle_py.ransac(loc1M, pts_tf, coeff, seed=3), oracle_c.ransac(loc1M, pts_tf, coeff, seed=3)):
        assert not r["failed"] and len(r["inlierIdx"]) == r["maxInliers"] >= 990
        assert np.linalg.norm(r["T"] - T_back) < 0.05


def test_debugRANSAC_exact_triple(oracle_py, oracle_c):
    """debugRANSAC.m:2-54: N = 3, every hypothesis is the exact transform."""
    rng = np.random.default_rng(11)
    pts = rng.normal(size=(3, 3))
    R = oracle_py.eul2rotm(rng.uniform(0, 2 * np.pi, 3), "XYZ"); t = rng.normal(size=3)
    T = np.eye(4); T[:3, :3] = R; T[3, :3] = t
    pts_tf = oracle_py.quickTF(pts, T)
    coef = dict(minPtNum=3, iterNum=200, thDist=0.1, thInlrRatio=0.5, REFINE=True)
    for r in (oracle_py.ransac(pts, pts_tf, coef, seed=0), oracle_c.ransac(pts, pts_tf, coef, seed=0)):
        assert np.linalg.norm(T @ r["T"] - np.eye(4)) < 1e-10                              # est_error (:38)
        assert r["numSuccess"] == 200 and r["maxInliers"] == 3 and list(r["inlierIdx"]) == [1, 2, 3]


@pytest.mark.parametrize("refine", [True, False])
def test_ransac_invariants_and_cross_check(refine, oracle_py, oracle_c):
    p1, p2, Ttrue = rigid_case(300, 7)
    coef = dict(minPtNum=3, iterNum=400, thDist=0.05, thInlrRatio=0.1, REFINE=refine)
    a = oracle_py.ransac(p1, p2, coef, seed=5)
    b = oracle_c.ransac(p1, p2, coef, seed=5)
    cc = a["inlrNum_refined"] if refine else a["inlrNum"]
    assert len(a["inlierIdx"]) == a["maxInliers"] == cc.max()                  # ransac.m:58,70,92
    assert a["numSuccess"] == int(np.sum(cc >= a["thInlr"]))                   # ransac.m:95
    assert a["winner"] == int(np.argmax(cc))                                   # first maximum
    np.testing.assert_array_equal(a["inlrNum"], b["inlrNum"])
    np.testing.assert_array_equal(a["inlrNum_refined"], b["inlrNum_refined"])
    np.testing.assert_array_equal(a["inlierIdx"], b["inlierIdx"])
    assert np.abs(a["T"] - b["T"]).max() < 1e-11
    # calcDists is the SQUARED distance (getInliersRANSAC.m:53)
    d = oracle_py.calcDists(a["T"], p1, p2)
    assert np.allclose(d, np.sum((p1 - oracle_py.quickTF(p2, a["T"])) ** 2, axis=1))
    assert np.allclose(oracle_c.calcDists(a["T"], p1, p2), d, rtol=1e-12, atol=1e-15)


def test_ransac_failure(oracle_py, oracle_c):
    rng = np.random.default_rng(5)
    p1 = rng.uniform(0, 100, (100, 3)); p2 = rng.uniform(0, 100, (100, 3))
    coef = dict(minPtNum=3, iterNum=100, thDist=1e-3, thInlrRatio=0.5, REFINE=True)
    for r in (oracle_py.ransac(p1, p2, coef, seed=1), oracle_c.ransac(p1, p2, coef, seed=1)):
        assert r["failed"] and r["T"] is None and r["numSuccess"] == 0 and r["maxInliers"] == 0 and len(r["inlierIdx"]) == 0


def test_sample_table_is_shared_and_valid(oracle_py, oracle_c):
    a = oracle_py.sample_table(50, 300, 3, 99)
    assert (a == oracle_c.sample_table(50, 300, 3, 99)).all()
    assert a.min() >= 1 and a.max() <= 50
    assert all(len(set(row)) == 3 for row in a)
    assert (oracle_py.sample_table(3, 20, 3, 1) >= 1).all()       # n == minPtNum: always a permutation of 1..3
    assert all(sorted(r) == [1, 2, 3] for r in oracle_py.sample_table(3, 20, 3, 1))


def test_match_features_semantics_small(oracle_py, oracle_c):
    """Hand-checkable matchFeatures behaviour (documented semantics; PARITY UNPINNED)."""
    f2 = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0]], float)
    f1 = np.array([[2, 0.1, 0], [0, 0.1, 3], [1, 1, 0.1]], float)
    par = dict(Metric="SSD", MatchThreshold=100.0, MaxRatio=1.0, Unique=False)
    pairs, met = oracle_py.matchFeatures(f1, f2, **par)
    assert pairs.tolist() == [[1, 1], [2, 3], [3, 4]] and pairs.dtype == np.uint32
    pc, mc = oracle_c.matchFeatures(f1, f2, par)
    assert pc.tolist() == pairs.tolist() and np.allclose(mc, met)
    # normalisation: scaling a row does not change anything
    p2, _ = oracle_py.matchFeatures(f1 * np.array([[3.0], [0.5], [10.0]]), f2, **par)
    assert p2.tolist() == pairs.tolist()
    # threshold in percent of the max unit-vector distance: SSD 4, SAD 2*sqrt(D)
    assert oracle_py.match_threshold(10, 981, "SAD") == pytest.approx(0.1 * 2 * np.sqrt(981))
    assert oracle_py.match_threshold(1, 64, "SSD") == pytest.approx(0.04)
    # ratio test drops ambiguous rows, Unique keeps only mutual best
    f1b = np.array([[1, 0, 0], [1, 0.01, 0]], float)
    pr, _ = oracle_py.matchFeatures(f1b, f2, Metric="SSD", MatchThreshold=100.0, MaxRatio=1.0, Unique=True)
    assert pr.tolist() == [[1, 1]]                     # both pick model 1; query 1 is its first-best
    # exact duplicates: second-best distance 0 < 1e-6 -> ratio forced to 1 -> rejected unless MaxRatio == 1
    f2d = np.vstack([f2, f2[:1]])
    pd_, _ = oracle_py.matchFeatures(f2[:1], f2d, Metric="SAD", MatchThreshold=100.0, MaxRatio=0.99)
    assert pd_.shape == (0, 2)


def test_get_matches_numpy_vs_c(oracle_py, oracle_c):
    rng = np.random.default_rng(2)
    dM = rng.poisson(3.0, (150, 30)).astype(float); dS = rng.poisson(3.0, (70, 30)).astype(float)
    dS[:30] = dM[:30] + rng.poisson(0.2, (30, 30))
    for metric in ("SAD", "SSD"):
        par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
                   MatchThreshold=10, MaxRatio=0.99, Metric=metric, Unique=True)
        a = oracle_py.getMatches(dS, dM, par); b = oracle_c.getMatches(dS, dM, par)
        np.testing.assert_array_equal(a, b)
        assert len(a) > 10 and (np.diff(a[:, 0].astype(int)) > 0).all()
    # the appended constant is norm_factor * mean row L1 norm over BOTH sets (getMatches.m:24-26)
    s, m = oracle_py.preprocess_descriptors(dS, dM, dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=False))
    avg = np.abs(np.vstack([dS, dM])).sum(axis=1).mean()
    assert s.shape[1] == 31 and np.allclose(s[:, -1], 2 * avg) and np.allclose(m[:, -1], 2 * avg)


def test_pca_convention_and_align_points(oracle_py, oracle_c):
    rng = np.random.default_rng(3)
    X = rng.normal(size=(500, 3)) * [3.0, 1.5, 0.4]
    coeff, score, latent = oracle_py.pca_eig(X)
    assert latent[0] >= latent[1] >= latent[2]
    assert all(coeff[np.argmax(np.abs(coeff[:, k])), k] > 0 for k in range(3))         # largest-|.| entry positive
    assert np.allclose(score, (X - X.mean(axis=0)) @ coeff)
    al, cu, c = oracle_py.AlignPoints_KNN(X + [10, 20, 30])
    assert np.isclose(abs(np.linalg.det(cu)), 1.0) and np.allclose(c, (X + [10, 20, 30]).mean(axis=0))
    assert np.allclose(al, (X + [10, 20, 30]) @ cu)                                     # un-centred pts (:59)
    # principal axis ~ x: the aligned cloud is widest along the first coordinate
    assert np.std(al[:, 0]) > np.std(al[:, 1]) > np.std(al[:, 2])
    for C1, C2 in [(False, False), (True, True)]:
        a1 = oracle_py.AlignPoints_KNN(X, C1, C2); a2 = oracle_c.AlignPoints_KNN(X, C1, C2)
        assert np.abs(a1[0] - a2[0]).max() < 1e-10 and np.abs(a1[1] - a2[1]).max() < 1e-12


def test_get_local_points(oracle_py):
    """getLocalPoints.m:8-35: strict box + strict radius, min <= n <= max, points relative to c."""
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 2.999], [3, 0, 0], [5, 5, 5]], float)
    rel, d = oracle_py.getLocalPoints(pts + 10, 3.0, [10, 10, 10], 1, 10)
    assert rel.shape == (4, 3) and np.allclose(d, [0, 1, 2, 2.999])                     # |p-c| == R excluded
    assert oracle_py.getLocalPoints(pts, 3.0, [0, 0, 0], 5, 10) == (None, None)
    assert oracle_py.getLocalPoints(pts, 3.0, [0, 0, 0], 1, 3) == (None, None)


def test_knn2_points_numpy_vs_c(oracle_py, oracle_c):
    rng = np.random.default_rng(4)
    q = rng.uniform(0, 10, (200, 3)).astype(np.float32); m = rng.uniform(0, 10, (1000, 3)).astype(np.float32)
    i1, d1 = oracle_py.knn2_points_f32(q, m); i2, d2 = oracle_c.knn2_points_f32(q, m)
    assert (i1 == i2).all() and (d1 == d2).all()
    np.testing.assert_array_equal(oracle_py.match_points_f32(q, m, 0.5, 0.8), oracle_c.match_points_f32(q, m, 0.5, 0.8))
    i3, d3 = oracle_c.knn2_points_f32(q, m[:1])
    assert (i3[:, 1] == -1).all() and np.isinf(d3[:, 1]).all()


def _strips(P, seed, ns=8):
    rng = np.random.default_rng(seed)
    per = P // ns
    out = []
    for s in range(ns):
        x = rng.uniform(0, 60, per); y = 6 * s + rng.uniform(-1.2, 1.2, per); z = 10 + 0.1 * x * np.sin(s) + rng.normal(0, 0.25, per)
        out.append(np.column_stack([x, y, z]))
    return np.vstack(out)


def test_histcounts_semantics(oracle_py):
    """histcn.m:108 -> histcounts: left-closed bins, last bin right-closed, outside / NaN -> 0."""
    e = np.array([0.0, 1.0, 2.0, 3.0])
    x = np.array([-0.1, 0.0, 0.5, 1.0, 2.999, 3.0, 3.0001, np.nan])
    assert oracle_py.histcounts_loc(x, e).tolist() == [0, 1, 1, 2, 3, 3, 0, 0]
    r, t, p = oracle_py.histogram_edges(3.5)
    assert len(r) == 11 and len(t) == 8 and len(p) == 15
    assert np.allclose(r ** 3, np.linspace(0, 3.5 ** 3, 11)) and np.isclose(t[-1], np.pi) and np.isclose(p[0], -np.pi)
    # phi = atan2(y, y) only ever takes three values (getSpacialHistogramDescriptors.m:152)
    assert oracle_py.histcounts_loc(np.array([np.pi / 4, -3 * np.pi / 4]), p).tolist() == [9, 2]


def test_descriptors_numpy_vs_c(oracle_py, oracle_c):
    pts = _strips(16000, 0)
    rng = np.random.default_rng(1)
    kp = np.column_stack([rng.uniform(5, 55, 60), 6 * rng.integers(0, 8, 60) + rng.uniform(-1.5, 1.5, 60), rng.uniform(9, 16, 60)])
    for align in (True, False):
        for k in (0.85, "all"):
            opt = dict(min_pts=60, max_pts=6000, R=3.5, thVar=[3, 1.5], k=k, ALIGN_POINTS=align)
            f1, d1 = oracle_py.getSpacialHistogramDescriptors(pts, kp, opt)
            f2, d2 = oracle_c.getSpacialHistogramDescriptors(pts, kp, opt)
            assert len(f1) > 5 and np.array_equal(f1, f2) and np.array_equal(d1, d2)
            # every local point lands in exactly one bin unless r == 0: row sums = support sizes
            n_local = [int(np.sum(np.linalg.norm(pts - c, axis=1) < 3.5)) for c in f1]
            assert d1.sum(axis=1).tolist() == n_local
            # only 2 of the 14 phi slabs can be populated (y > 0 -> pi/4, y < 0 -> -3pi/4)
            slabs = d1.reshape(len(d1), 14, 7, 10).sum(axis=(2, 3))
            assert (np.count_nonzero(slabs, axis=1) <= 2).all()


# ---- the one restatement of the match metric the reference itself holds ------------------------------------------
def getDescDist_literal(desc1, desc2):
    """visualizeGTMatches.m:390-410, line by line: append `unnorm*avg_desc_len` (2 * 1600), raise to 0.45,
    L2-normalise, L1 distance.  The order append -> power -> L2-normalise -> SAD is what getMatches.m:22-56 +
    matchFeatures('Metric','SAD') must reproduce."""
    change_metric, unnorm, avg_desc_len = 0.45, 2, 1600            # :392-394
    d1 = np.concatenate([desc1, [unnorm * avg_desc_len]])           # :397
    d2 = np.concatenate([desc2, [unnorm * avg_desc_len]])           # :398
    d1 = d1 ** change_metric                                        # :401
    d2 = d2 ** change_metric                                        # :402
    d1 = d1 / np.sqrt(np.sum(d1 * d1))                              # :405  norm(desc1, 2)
    d2 = d2 / np.sqrt(np.sum(d2 * d2))                              # :406
    return np.sum(np.abs(d1 - d2))                                  # :409  vecnorm(desc1 - desc2, 1)


def desc_rows_l1_1600(n, D, seed):
    """Non-negative integer count rows whose L1 norm is exactly 1600 each, so that getMatches.m:24's
    mean(vecnorm([descSurface; descModel], 1, 2)) is exactly getDescDist's avg_desc_len."""
    rng = np.random.default_rng(seed)
    rows = rng.multinomial(1600, rng.dirichlet(np.full(D, 0.3)), size=n).astype(np.float64)
    assert (rows.sum(axis=1) == 1600).all()
    return rows


def test_getDescDist_known_answer(oracle_py, oracle_c):
    """visualizeGTMatches.m:390-410 pins the metric getMatches hands to matchFeatures: for every matched pair the
    oracle's matchMetric must equal the reference's own scalar restatement."""
    D = 60
    dM = desc_rows_l1_1600(40, D, 1)
    dS = dM[np.random.default_rng(2).permutation(40)[:25]].copy()
    to = np.random.default_rng(3).integers(0, D, 25)
    for r in range(25):                                   # move three counts out of the fullest bin: L1 stays 1600
        dS[r, int(np.argmax(dS[r]))] -= 3; dS[r, to[r]] += 3
    assert (dS.sum(axis=1) == 1600).all()
    par = dict(Method="Exhaustive", Metric="SAD", MatchThreshold=100.0, MaxRatio=1.0, Unique=False, UNNORMALIZE=True,
               norm_factor=2, CHANGE_METRIC=True, metric_factor=0.45)
    # (a) the preprocessing alone: appended constant == unnorm * avg_desc_len == 3200, then the power
    pS, pM = oracle_py.preprocess_descriptors(dS, dM, par)
    assert pS.shape[1] == D + 1 and np.all(pS[:, -1] == 3200.0 ** 0.45) and np.all(pM[:, -1] == 3200.0 ** 0.45)
    # (b) the per-pair metric of the numpy oracle, all 25 x best-of-40
    pairs, met, allr = oracle_py.matchFeatures(pS, pM, Method="Exhaustive", MatchThreshold=100.0, MaxRatio=1.0,
                                               Metric="SAD", Unique=False, return_all=True)
    assert len(pairs) == 25
    for (i, j), d in zip(pairs.astype(int), met):
        assert abs(d - getDescDist_literal(dS[i - 1], dM[j - 1])) < 1e-15 * 8, (i, j)
        # and it is the minimum of the literal distance over the model rows (first index on ties)
        lit = np.array([getDescDist_literal(dS[i - 1], dM[k]) for k in range(40)])
        assert j - 1 == int(np.argmin(np.round(lit, 13))) or abs(lit[j - 1] - lit.min()) < 1e-14
    # (c) the C oracle: same pairs, same metric
    pc, mc = oracle_c.matchFeatures(pS, pM, dict(par, UNNORMALIZE=False, CHANGE_METRIC=False))
    assert np.array_equal(pc, pairs) and np.abs(mc - met).max() < 1e-14
    assert np.array_equal(oracle_c.getMatches(dS, dM, par), oracle_py.getMatches(dS, dM, par))
    # (d) the second-best distances too (what the ratio test divides by)
    for i in range(25):
        assert abs(allr["d2"][i] - getDescDist_literal(dS[i], dM[allr["i2"][i]])) < 1e-14
