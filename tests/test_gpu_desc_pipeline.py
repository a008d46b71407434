"""The resident descriptor chain (pcreg_dev_spatial_histogram_descriptors -> pcreg_dev_get_matches ->
pcreg_dev_gather_matched_rows -> pcreg_dev_ransac) against the oracle running the same chain of
completeExperimentFast.m:131-213 on the CPU: keypoints, counts, pairs bit-exact; T within 1e-9."""
import numpy as np
import pytest

from test_gpu_descriptors import OPT, keypoints, strips

pytestmark = pytest.mark.gpu

PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
           MatchThreshold=10, MaxRatio=0.8, Metric="SAD", Unique=True, VERBOSE=0)
COEF = dict(minPtNum=3, iterNum=2000, thDist=0.5, thInlrRatio=0.1, REFINE=True, VERBOSE=0)


def _scene(seed):
    import oracle.pcreg_oracle as o
    model = strips(40000, seed)
    rng = np.random.default_rng(seed + 1)
    R = o.eul2rotm(np.array([0.05, -0.02, 0.03])); t = np.array([0.4, -0.3, 0.2])
    sel = (model[:, 0] > 5) & (model[:, 0] < 45)
    surface = model[sel] @ R.T + t + rng.normal(0, 0.01, (sel.sum(), 3))
    kpM = keypoints(260, seed + 2)
    kpS = kpM[(kpM[:, 0] > 8) & (kpM[:, 0] < 42)] @ R.T + t
    return model, surface, kpM, kpS


@pytest.mark.parametrize("metric", ["SAD", "SSD"])
def test_resident_descriptor_chain_equals_oracle_chain(metric, oracle_c):
    import torch
    import pcreg_amd as pc
    from pcreg_amd.device import DescriptorPipeline, soa
    model, surface, kpM, kpS = _scene(11)
    par = dict(PAR, Metric=metric, MatchThreshold=10 if metric == "SAD" else 1.0)
    # --- oracle chain
    fM, dM = oracle_c.getSpacialHistogramDescriptors(model, kpM, OPT)
    fS, dS = oracle_c.getSpacialHistogramDescriptors(surface, kpS, OPT)
    assert len(fM) > 40 and len(fS) > 20
    ref_pairs = oracle_c.getMatches(dS, dM, par)
    assert len(ref_pairs) >= 3
    p1, p2 = fS[ref_pairs[:, 0] - 1], fM[ref_pairs[:, 1] - 1]
    ref = oracle_c.ransac(p1, p2, COEF, seed=5)
    assert not ref["failed"]
    # --- resident chain
    dev = torch.device("cuda", 0)
    pipe = DescriptorPipeline(dev)
    t = lambda a: soa(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    featM, descM, VM = pipe.describe(t(model), t(kpM), OPT)
    featS, descS, VS = pipe.describe(t(surface), t(kpS), OPT)
    assert (VM, VS) == (len(fM), len(fS))
    np.testing.assert_array_equal(featM[:VM].cpu().numpy(), fM)
    np.testing.assert_array_equal(descS[:VS].cpu().numpy(), dS)
    pairs, n_pairs = pipe.match(descS, VS, descM, VM, par)
    n = int(n_pairs.item())
    np.testing.assert_array_equal(pairs[:n].cpu().numpy().astype(np.uint32), ref_pairs)
    pipe.ransac(pairs, featS, featM, COEF, seed=5)
    r = pipe.fetch_result()
    assert r["n"] == len(ref_pairs) and not r["failed"]
    assert r["maxInliers"] == ref["maxInliers"] and r["numSuccess"] == ref["numSuccess"]
    np.testing.assert_array_equal(r["inlierIdx"].astype(np.int64), ref["inlierIdx"])
    assert np.linalg.norm(r["T"] - ref["T"]) < 1e-9
    # --- and the host tier agrees with both
    np.testing.assert_array_equal(pc.getMatches(dS, dM, par), ref_pairs)


def test_dev_get_matches_feature_major_layout_and_empty(oracle_c):
    """MATLAB-shaped (column-major) device input, metric output, caller buffers untouched, Q = 0."""
    import ctypes as C
    import torch
    from pcreg_amd import _lib
    from pcreg_amd.api import _match_opts
    from pcreg_amd.device import _p, _stream
    rng = np.random.default_rng(2)
    dM = rng.poisson(3.0, (900, 64)).astype(np.float64)
    dS = dM[rng.choice(900, 300, replace=False)] + rng.poisson(0.3, (300, 64))
    par = dict(PAR, MatchThreshold=20, MaxRatio=0.9)
    ref = oracle_c.getMatches(dS, dM, par)
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    tS = torch.from_numpy(np.asfortranarray(dS).T.copy()).to(dev)        # [D, Q] == column-major Q x D
    tM = torch.from_numpy(np.asfortranarray(dM).T.copy()).to(dev)
    keepS = tS.clone()
    pairs = torch.zeros((300, 2), dtype=torch.int32, device=dev)
    metric = torch.zeros(300, dtype=torch.float64, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(L.pcreg_dev_get_matches_workspace(300, 900, 64), dtype=torch.uint8, device=dev)
    o = _match_opts(par)
    _lib.check(L.pcreg_dev_get_matches(_p(tS), 300, 300, _p(tM), 900, 900, 64, _lib.LAYOUT_FEATURE_MAJOR, C.byref(o),
                                       _p(pairs), _p(metric), _p(n), _p(ws), C.c_size_t(ws.numel()), _stream()))
    k = int(n.item())
    np.testing.assert_array_equal(pairs[:k].cpu().numpy().astype(np.uint32), ref)
    assert torch.equal(tS, keepS)
    assert (metric[:k].cpu().numpy() > 0).all()
    _lib.check(L.pcreg_dev_get_matches(_p(tS), 0, 300, _p(tM), 900, 900, 64, _lib.LAYOUT_FEATURE_MAJOR, C.byref(o),
                                       _p(pairs), None, _p(n), _p(ws), C.c_size_t(ws.numel()), _stream()))
    assert int(n.item()) == 0
    # a too-small workspace is refused, not overrun
    rc = L.pcreg_dev_get_matches(_p(tS), 300, 300, _p(tM), 900, 900, 64, _lib.LAYOUT_FEATURE_MAJOR, C.byref(o),
                                 _p(pairs), None, _p(n), _p(ws), C.c_size_t(1024), _stream())
    assert rc == _lib.PCREG_E_WORKSPACE


def test_compact_u16_descriptors_give_the_same_chain(oracle_c):
    """pcreg_dev_spatial_histogram_descriptors_rows_u16 -> pcreg_dev_get_matches_rows_u16: the counts as uint16 rows in
    keypoint order + the survivor list (a quarter of the bytes, written once) equal the double rows value for value, and the
    pairs are those of the double chain and the oracle."""
    import torch
    from pcreg_amd.device import DescriptorPipeline, soa
    from pcreg_amd._lib import PcregError
    model, surface, kpM, kpS = _scene(23)
    fM, dM = oracle_c.getSpacialHistogramDescriptors(model, kpM, OPT)
    fS, dS = oracle_c.getSpacialHistogramDescriptors(surface, kpS, OPT)
    ref_pairs = oracle_c.getMatches(dS, dM, PAR)
    dev = torch.device("cuda", 0)
    pipe = DescriptorPipeline(dev)
    t = lambda a: soa(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    featM, descM, VM = pipe.describe(t(model), t(kpM), OPT, compact=True)
    featS, descS, VS = pipe.describe(t(surface), t(kpS), OPT, compact=True)
    assert descM.rows.dtype == torch.uint16 and (VM, VS) == (len(fM), len(fS))
    assert (np.diff(descM.index[:VM].cpu().numpy()) > 0).all()                   # ascending keypoint numbers
    np.testing.assert_array_equal(descM.compact(VM).cpu().numpy().astype(np.float64), dM)
    np.testing.assert_array_equal(descS.compact(VS).cpu().numpy().astype(np.float64), dS)
    np.testing.assert_array_equal(featS[:VS].cpu().numpy(), fS)
    pairs, n_pairs = pipe.match(descS, VS, descM, VM, PAR)
    n = int(n_pairs.item())
    assert n == len(ref_pairs) >= 3
    np.testing.assert_array_equal(pairs[:n].cpu().numpy().astype(np.uint32), ref_pairs)
    # max_pts beyond 65535 is fine: a count cannot exceed the support an LDS-resident list can hold (8191)
    f2, d2, V2 = pipe.describe(t(model), t(kpM), dict(OPT, max_pts=70000), compact=True)
    assert V2 >= VM
