"""The resident sphere-sweep driver (pcreg_amd/sweep.py) against the oracle's CPU restatement of
completeExperimentFast.m:46-224, and the final-stage pieces (quickTF, invertTF, distance refine)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
           MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)
OPT = dict(minPtNum=3, iterNum=1500, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)


def _scene(seed=0, VM=6000, VS=260, D=96):
    """Synthetic keypoints + count-like descriptors: the surface is a rigidly moved patch of the model,
    its descriptors are noisy copies of the matching model rows."""
    import oracle.pcreg_oracle as o
    rng = np.random.default_rng(seed)
    featM = rng.uniform([0, 0, 0], [40, 30, 20], (VM, 3))
    descM = rng.poisson(3.0, (VM, D)).astype(np.float64)
    centre = np.array([22.0, 14.0, 9.0])
    near = np.argsort(np.linalg.norm(featM - centre, axis=1))[:VS]
    R = o.eul2rotm(np.array([0.3, -0.2, 0.1])); t = np.array([2.0, -1.0, 0.5])
    featS = featM[near] @ R.T + t + rng.normal(0, 0.02, (VS, 3))
    descS = descM[near] + rng.poisson(0.15, (VS, D))
    return featM, descM, featS, descS


def test_sphere_sweep_equals_oracle_driver(oracle_c, oracle_py):
    from pcreg_amd.sweep import SphereSweep
    featM, descM, featS, descS = _scene()
    kw = dict(R_desc=9.0, d_spheres=6.0, min_pts=500, putative_thresh=60, seed=3)
    ref = oracle_py.sphere_sweep(featM, descM, featS, descS, PAR, OPT, get_matches=oracle_c.getMatches,
                                 run_ransac=oracle_c.ransac, **kw)
    assert len(ref["centres"]) >= 4 and len(ref["trial"]) >= 1          # the scene exercises both stages
    sw = SphereSweep(featM, descM, featS, descS)
    got = sw.run(PAR, OPT, **kw)                                            # batched: two host syncs for the whole sweep
    ser = sw.run_serial(PAR, OPT, **kw)                                     # one sphere at a time
    for k in ("num_desc", "num_putative", "trial", "statsPutative", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(got[k], ser[k])
    for a, b in zip(got["transforms"], ser["transforms"]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    np.testing.assert_array_equal(got["centres"], ref["centres"])
    np.testing.assert_array_equal(got["num_desc"], ref["num_desc"])
    np.testing.assert_array_equal(got["num_putative"], ref["num_putative"])
    for a, b in zip(got["model_rows"], ref["model_rows"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(got["matches"], ref["matches"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(got["trial"], ref["trial"])
    for k in ("statsPutative", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(got[k], ref[k])
    np.testing.assert_allclose(got["statsRatio"], ref["statsRatio"], rtol=0, atol=1e-12)
    for a, b in zip(got["transforms"], ref["transforms"]):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.linalg.norm(a - b) < 1e-9


def test_sphere_counts_select_and_gather(oracle_py):
    import torch
    from pcreg_amd.sweep import SphereSweep
    featM, descM, featS, descS = _scene(seed=4, VM=3000, VS=40, D=24)
    sw = SphereSweep(featM, descM, featS, descS)
    centres = sw.sphere_centres(7.0)
    np.testing.assert_array_equal(centres, oracle_py.pcUniformSamples(featM, 7.0))
    valid, n = sw.valid_spheres(centres, 8.0, min_pts=100)
    ref_n = np.array([oracle_py.getDescriptorMask(featM, c, 8.0).sum() for c in centres])
    np.testing.assert_array_equal(n, ref_n)
    np.testing.assert_array_equal(valid, ref_n >= 100)
    # a centre ON a keypoint and a radius equal to a distance: strict '<' at the boundary
    c = featM[17]
    R = float(np.sort(np.linalg.norm(featM - c, axis=1))[50])
    m = sw.match_sphere(c, R, PAR)
    mask = oracle_py.getDescriptorMask(featM, c, R)
    np.testing.assert_array_equal(m["rows"].cpu().numpy(), np.nonzero(mask)[0])
    np.testing.assert_array_equal(m["featCur"][:m["num_desc"]].cpu().numpy(), featM[mask])
    # an empty sphere
    e = sw.match_sphere([1e6, 0, 0], 1.0, PAR)
    assert e["num_desc"] == 0 and e["num_putative"] == 0


def test_quicktf_inverttf_and_distance_refine(oracle_py):
    import torch
    import pcreg_amd as pc
    from pcreg_amd.device import soa
    from pcreg_amd.sweep import quickTF_dev, refine_by_distance_dev
    rng = np.random.default_rng(1)
    dev = torch.device("cuda", 0)
    pts = rng.uniform(-20, 20, (5000, 3))
    T = np.eye(4); T[:3, :3] = oracle_py.eul2rotm(np.array([0.4, 0.1, -0.3])).T; T[3, :3] = [1.0, -2.0, 3.0]
    p_dev = soa(torch.from_numpy(pts).to(dev))
    moved = quickTF_dev(p_dev, T)
    np.testing.assert_allclose(moved.t().cpu().numpy(), oracle_py.quickTF(pts, T), rtol=0, atol=1e-12)
    back = quickTF_dev(moved, pc.invertTF(T))                                   # quickTF(quickTF(p,T), invertTF(T)) == p
    np.testing.assert_allclose(back.t().cpu().numpy(), pts, rtol=0, atol=1e-11)
    # distance refine: pts1 = pts2 moved by a small transform + noise, 30 % gross outliers
    T2 = np.eye(4); T2[:3, :3] = oracle_py.eul2rotm(np.array([0.01, -0.02, 0.015])).T; T2[3, :3] = [0.05, -0.02, 0.03]
    p2 = pts[:1200]
    p1 = oracle_py.quickTF(p2, T2) + rng.normal(0, 0.01, p2.shape)
    p1[:360] += rng.uniform(3, 6, (360, 3))
    Tref, inl = oracle_py.refine_by_distance(p1, p2, 1.5)
    Tg, cnt = refine_by_distance_dev(soa(torch.from_numpy(p1).to(dev)), soa(torch.from_numpy(p2).to(dev)), len(p1), 1.5)
    assert cnt == len(inl) == 840
    assert np.linalg.norm(Tg - Tref) < 1e-10 and np.linalg.norm(Tg - T2) < 0.01
    # exactly three inliers -> estimateTransform's N == 3 branch; fewer -> []
    q1 = p1.copy(); q1[3:] += 100.0
    Tref3, inl3 = oracle_py.refine_by_distance(q1, p2, 1.5)
    assert len(inl3) <= 3
    Tg3, cnt3 = refine_by_distance_dev(soa(torch.from_numpy(q1).to(dev)), soa(torch.from_numpy(p2).to(dev)), len(q1), 1.5)
    assert cnt3 == len(inl3)
    assert (Tg3 is None) == (Tref3 is None)
    if Tg3 is not None:
        assert np.linalg.norm(Tg3 - Tref3) < 1e-9


def test_batched_sweep_counts_its_host_syncs(monkeypatch):
    """The batched driver's contract: no device-to-host read between the first sphere's select and the batched
    ransac launch.  torch.Tensor.cpu / .item are the only ways sweep.py reads the device; they are counted per phase."""
    import torch
    from pcreg_amd import sweep as sw_mod
    featM, descM, featS, descS = _scene(seed=2, VM=5000, VS=200, D=64)
    sw = sw_mod.SphereSweep(featM, descM, featS, descS)
    calls = []
    real_plan = sw_mod.lib().pcreg_dev_sweep_plan
    orig_cpu, orig_item = torch.Tensor.cpu, torch.Tensor.item
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: (calls.append("cpu"), orig_cpu(self, *a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "item", lambda self, *a, **k: (calls.append("item"), orig_item(self, *a, **k))[1])
    out = sw.run(PAR, OPT, R_desc=9.0, d_spheres=6.0, min_pts=400, putative_thresh=50, seed=1)
    S = len(out["centres"])
    assert S >= 4
    # sync 1 = featM (once, cached) + the counts; sync 2 = the final block of reads; nothing scales with S
    assert len(calls) <= 10, calls
