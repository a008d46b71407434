"""completeExperimentFast.m:280-394 chained on the device tier (pcreg_amd.sweep.FinalStage) against the oracle's CPU restatement
(oracle.final_stage): per cluster the moved surface, the no-LRF descriptors, the matches inside the cluster's sphere and the share
of close matches; then the best cluster, T_refine and the final surface."""
import numpy as np
import pytest

from test_gpu_descriptors import OPT, keypoints, strips

pytestmark = pytest.mark.gpu

PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
           MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)


def _scene(seed):
    import oracle.pcreg_oracle as o
    model = strips(40000, seed)
    rng = np.random.default_rng(seed + 1)
    # the stereo surface: a crop of the model seen in another frame
    R = o.eul2rotm(np.array([0.4, -0.25, 0.15])); t = np.array([3.0, -2.0, 1.5])
    T_true = np.eye(4); T_true[:3, :3] = R; T_true[3, :3] = t            # row-vector convention: [p, 1] * T
    sel = (model[:, 0] > 8) & (model[:, 0] < 42)
    surface = o.quickTF(model[sel], T_true) + rng.normal(0, 0.01, (sel.sum(), 3))
    kpM = keypoints(1200, seed + 2)                                         # the model's no-LRF keypoints (an input of the stage)
    return model, surface, kpM, T_true


def _perturbed(T, rng, ang=0.01, sh=0.08):
    import oracle.pcreg_oracle as o
    dT = np.eye(4); dT[:3, :3] = o.eul2rotm(rng.normal(0, ang, 3)); dT[3, :3] = rng.normal(0, sh, 3)
    return T @ dT


def test_final_stage_equals_oracle(oracle_c, oracle_py, monkeypatch):
    import torch
    from pcreg_amd.device import soa
    from pcreg_amd.sweep import FinalStage
    model, surface, kpM, T_true = _scene(5)
    opt = dict(OPT, ALIGN_POINTS=False)
    featM, descM = oracle_c.getSpacialHistogramDescriptors(model, kpM, opt)            # featModel0.3_noLRF / descModel0.3_noLRF (:313-316)
    assert len(featM) > 400
    rng = np.random.default_rng(9)
    # cluster 0: the right place with RANSAC's (slightly off) transform; 1: a wrong transform; 2: a sphere without model keypoints;
    # 3: the right place again with a coarser transform
    T_wrong = np.eye(4); T_wrong[:3, :3] = oracle_py.eul2rotm(np.array([1.2, 0.4, -0.7])); T_wrong[3, :3] = [10.0, 5.0, -4.0]
    clusters = [(np.array([25.0, 18.0, 12.0]), _perturbed(T_true, rng, 0.004, 0.03)), (np.array([12.0, 30.0, 11.0]), T_wrong),
                (np.array([500.0, 500.0, 500.0]), _perturbed(T_true, rng)), (np.array([24.0, 19.0, 12.0]), _perturbed(T_true, rng, 0.012, 0.1))]
    kps = []
    near = kpM[(kpM[:, 0] > 10) & (kpM[:, 0] < 40)]
    for loc, T in clusters:                                 # :297: the script draws them at random (dense, d = 0.5); an input here:
        moved = oracle_py.quickTF(surface, oracle_py.invertTF(T))          # some at random in the padded box, some beside model keypoints
        kps.append(np.vstack([oracle_py.pcRandomUniformSamples(moved, 3.0, 3.5, rng)[:300], near + rng.normal(0, 0.05, near.shape)]))
    R_desc = 14.0
    ref = oracle_py.final_stage(surface, clusters, kps, featM, descM, R_desc, OPT, PAR, maxDist=1.5,
                                get_descriptors=oracle_c.getSpacialHistogramDescriptors, get_matches=oracle_c.getMatches)
    assert ref["best"] in (0, 3) and ref["T_refine"] is not None and len(ref["per_cluster"][ref["best"]]["inliers"]) >= 10
    assert np.isnan(ref["precisions"][2])                                                           # no model descriptors there: 0 / 0

    dev = torch.device("cuda", 0)
    calls = []
    orig_cpu, orig_item = torch.Tensor.cpu, torch.Tensor.item
    fs = FinalStage(soa(torch.from_numpy(surface).to(dev)), featM, descM, device=dev)
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: (calls.append("cpu"), orig_cpu(self, *a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "item", lambda self, *a, **k: (calls.append("item"), orig_item(self, *a, **k))[1])
    got = fs.run(clusters, kps, OPT, PAR, R_desc, maxDist=1.5, return_matches=False)
    assert len(calls) == 2, calls                                    # sizes, results: two reads for the whole stage, not per cluster
    monkeypatch.undo()
    got = fs.run(clusters, kps, OPT, PAR, R_desc, maxDist=1.5)
    for i, c in enumerate(ref["per_cluster"]):
        assert got["num_keypoints"][i] == len(c["feat"]) and got["num_desc"][i] == len(c["featCur"])
        np.testing.assert_array_equal(got["matches"][i], c["matches"], err_msg=f"cluster {i}")
        assert got["num_close"][i] == len(c["inliers"])
        np.testing.assert_allclose(got["pts_tform"][i].cpu().numpy().T, c["pts_tform"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(np.isnan(got["precisions"]), np.isnan(ref["precisions"]))
    np.testing.assert_allclose(got["precisions"][~np.isnan(ref["precisions"])], ref["precisions"][~np.isnan(ref["precisions"])], rtol=0, atol=1e-12)
    assert got["best"] == ref["best"]
    assert np.linalg.norm(got["T_refine"] - ref["T_refine"]) < 1e-9
    np.testing.assert_allclose(got["pts_final"].cpu().numpy().T, ref["pts_final"], rtol=0, atol=1e-9)
    # the refinement is a refinement: the final surface lies on the model crop it came from
    sel = (model[:, 0] > 8) & (model[:, 0] < 42)
    assert np.abs(ref["pts_final"] - model[sel]).max() < 0.3


def test_final_stage_without_any_match(oracle_py):
    """Every cluster empty: precisions all NaN, MATLAB's max returns index 1, no refinement, the moved surface is the result."""
    import torch
    from pcreg_amd.device import soa
    from pcreg_amd.sweep import FinalStage
    model, surface, kpM, T_true = _scene(6)
    dev = torch.device("cuda", 0)
    featM = np.random.default_rng(0).uniform(0, 50, (50, 3)); descM = np.random.default_rng(1).poisson(3.0, (50, 980)).astype(np.float64)
    fs = FinalStage(soa(torch.from_numpy(surface).to(dev)), featM, descM, device=dev)
    clusters = [(np.array([900.0, 0, 0]), T_true), (np.array([-900.0, 0, 0]), T_true)]
    kp = keypoints(40, 3)
    got = fs.run(clusters, [kp, kp], OPT, PAR, 5.0)
    assert np.all(np.isnan(got["precisions"])) and got["best"] == 0 and got["T_refine"] is None
    np.testing.assert_allclose(got["pts_final"].cpu().numpy().T, oracle_py.quickTF(surface, oracle_py.invertTF(T_true)), rtol=0, atol=1e-12)
