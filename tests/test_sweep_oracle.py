"""CPU checks of the oracle's restatement of the sphere-sweep driver pieces
(completeExperimentFast.m:46-224, :383-391, :406-414, :435-439)."""
import numpy as np

import oracle.pcreg_oracle as o


def test_pcUniformSamples_is_matlab_meshgrid_order():
    pts = np.array([[0.0, 10.0, -1.0], [2.1, 13.0, 0.9]])          # limits: x 0..2.1, y 10..13, z -1..0.9
    s = o.pcUniformSamples(pts, 1.0)
    # x = 0:1:2.1 -> 0,1,2 ; y = 10:1:13 -> 10..13 ; z = -1:1:0.9 -> -1,0
    assert s.shape == (3 * 4 * 2, 3)
    np.testing.assert_array_equal(s[:5], [[0, 10, -1], [0, 11, -1], [0, 12, -1], [0, 13, -1], [1, 10, -1]])   # y fastest, then x
    np.testing.assert_array_equal(s[12], [0, 10, 0])                                                      # then z


def test_getDescriptorMask_is_strict():
    feat = np.array([[0, 0, 0], [3, 4, 0], [3, 4, 1e-9], [0, 0, 4.9999999]], dtype=float)
    np.testing.assert_array_equal(o.getDescriptorMask(feat, [0, 0, 0], 5.0), [True, False, False, True])
    np.testing.assert_array_equal(o.getDescriptorMask(feat, [0, 0, 0], 5.5, -0.5), [True, False, False, True])   # margin < 0


def test_refine_by_distance_recovers_the_refinement():
    rng = np.random.default_rng(0)
    p2 = rng.uniform(-10, 10, (400, 3))
    T = np.eye(4); T[:3, :3] = o.eul2rotm(np.array([0.02, -0.01, 0.015])).T; T[3, :3] = [0.1, -0.05, 0.02]
    p1 = o.quickTF(p2, T)
    p1[:100] += 50.0
    Tr, inl = o.refine_by_distance(p1, p2, 1.5)
    np.testing.assert_array_equal(inl, np.arange(100, 400))
    assert np.linalg.norm(Tr - T) < 1e-12                      # [pts2,1]*T = [pts1,1]
    np.testing.assert_allclose(o.quickTF(o.quickTF(p2, T), o.invertTF(T)), p2, atol=1e-12)
    none, inl0 = o.refine_by_distance(p1[:100], p2[:100], 1.5)
    assert none is None and len(inl0) == 0


def test_sphere_sweep_invariants_small():
    rng = np.random.default_rng(2)
    featM = rng.uniform(0, 20, (600, 3)); descM = rng.poisson(3.0, (600, 24)).astype(float)
    near = np.argsort(np.linalg.norm(featM - [10, 10, 10], axis=1))[:60]
    R = o.eul2rotm(np.array([0.2, 0.1, -0.1])); t = np.array([1.0, 2.0, -1.0])
    featS = featM[near] @ R.T + t; descS = descM[near] + rng.poisson(0.1, (60, 24))
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Exhaustive",
               MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)
    opt = dict(minPtNum=3, iterNum=200, thDist=0.3, thInlrRatio=0.08, REFINE=True)
    r = o.sphere_sweep(featM, descM, featS, descS, par, opt, R_desc=7.0, d_spheres=5.0, min_pts=60, putative_thresh=20)
    assert len(r["centres"]) > 0 and (r["num_desc"] >= 60).all()
    assert (r["num_putative"] <= 60).all() and len(r["trial"]) >= 1
    for k, i in enumerate(r["trial"]):
        assert r["statsPutative"][k] == r["num_putative"][i] > 20
        assert r["statsInliers"][k] <= r["statsPutative"][k]
    best = int(np.argmax(r["statsInliers"]))
    T = r["transforms"][best]
    # the winning sphere recovers the motion: [pts2,1]*T = [pts1,1] with pts1 = surface, pts2 = model
    Ttrue = np.eye(4); Ttrue[:3, :3] = R.T; Ttrue[3, :3] = t
    assert np.linalg.norm(T - Ttrue) < 1e-6
