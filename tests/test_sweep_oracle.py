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


def test_pcRandomUniformSamples_count_and_box():
    """completeExperimentFast.m:416-429: round(padded volume / d^3) keypoints inside the padded bounding box."""
    import oracle.pcreg_oracle as o
    pts = np.array([[0.0, 0, 0], [10, 4, 2], [5, 1, 1]])
    kp = o.pcRandomUniformSamples(pts, 0.5, 3.5, np.random.default_rng(1))
    assert len(kp) == round((17 * 11 * 9) / 0.125)
    assert (kp.min(axis=0) >= [-3.5, -3.5, -3.5]).all() and (kp.max(axis=0) <= [13.5, 7.5, 5.5]).all()


def test_final_stage_picks_the_cluster_with_most_close_matches():
    """completeExperimentFast.m:357-394 on stand-in descriptors / matches: the precision is the share of matches closer than
    maxDist, max() takes the FIRST maximum and skips NaN, T_refine moves the close matches onto each other."""
    import oracle.pcreg_oracle as o
    rng = np.random.default_rng(3)
    featM = rng.uniform(0, 20, (60, 3)); descM = np.arange(60, dtype=np.float64)[:, None] * np.ones((1, 4))
    surface = rng.uniform(0, 20, (500, 3))
    shift = np.eye(4); shift[3, :3] = [0.3, -0.2, 0.1]
    T_ok = np.eye(4)
    # stand-ins: "descriptors" = the keypoints that were asked for, with their model row number as the descriptor;
    # "getMatches" pairs row i with model row desc[i]
    def fake_desc(pts, kp, opt):
        assert opt["ALIGN_POINTS"] is False
        return kp[:, :3].copy(), kp[:, 3:4] * np.ones((1, 4))
    def fake_matches(dS, dM, par):
        rows = {float(v): j for j, v in enumerate(dM[:, 0])}
        return np.array([[i + 1, rows[float(v)] + 1] for i, v in enumerate(dS[:, 0]) if float(v) in rows], dtype=np.uint32).reshape(-1, 2)
    ids = np.arange(60.0)
    kp_good = np.column_stack([o.quickTF(featM, shift), ids])                 # every match 0.37 away: all close
    kp_half = np.column_stack([featM + np.where(np.arange(60)[:, None] % 2 == 0, 0.1, 5.0), ids])
    clusters = [(np.array([10.0, 10, 10]), T_ok), (np.array([10.0, 10, 10]), T_ok), (np.array([900.0, 0, 0]), T_ok), (np.array([10.0, 10, 10]), T_ok)]
    r = o.final_stage(surface, clusters, [kp_half, kp_good, kp_good, kp_good], featM, descM, 100.0, dict(R=3.5), dict(), 1.5,
                      get_descriptors=fake_desc, get_matches=fake_matches)
    np.testing.assert_allclose(r["precisions"][[0, 1, 3]], [50.0, 100.0, 100.0])
    assert np.isnan(r["precisions"][2]) and r["best"] == 1                     # first maximum, NaN skipped
    p1 = r["per_cluster"][1]["pts1"]; p2 = r["per_cluster"][1]["pts2"]
    np.testing.assert_allclose(o.quickTF(p2, r["T_refine"]), p1, atol=1e-9)    # [pts2, 1] * T = [pts1, 1] (estimateTransform.m:8-71)
    np.testing.assert_allclose(r["pts_final"], o.quickTF(surface, o.invertTF(r["T_refine"])), atol=1e-12)
