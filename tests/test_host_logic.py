"""Host-side logic that needs no GPU: argument checks, the generic function-handle loop,
MATLAB-compat helpers."""
import numpy as np
import pytest


def test_quickTF_invertTF_roundtrip(oracle_py):
    import pcreg_amd as pc
    rng = np.random.default_rng(0)
    R = oracle_py.eul2rotm(rng.uniform(-3, 3, 3)); t = rng.normal(size=3)
    T = np.eye(4); T[:3, :3] = R; T[3, :3] = t
    p = rng.normal(size=(50, 3))
    assert np.abs(pc.quickTF(pc.quickTF(p, T), pc.invertTF(T)) - p).max() < 1e-13       # invertTF.m / quickTF.m
    assert np.abs(pc.invertTF(T) - np.linalg.inv(T)).max() < 1e-13
    assert np.abs(pc.quickTF(p, T) - oracle_py.quickTF(p, T)).max() == 0


def test_generic_handle_protocol_runs_on_host():
    """ransac.m:14-19: f = funcFindTransf(x1,y1); d = funcDist(f,x,y) -- any handles."""
    import pcreg_amd as pc
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 10, (200, 1)); y = 2 * x + 1 + rng.normal(0, 0.01, x.shape)
    y[::5] += rng.uniform(-5, 5, y[::5].shape)
    calls = {"fit": 0}

    def fit(a, b):
        calls["fit"] += 1
        return np.linalg.lstsq(np.hstack([a, np.ones_like(a)]), b, rcond=None)[0]

    def dist(f, a, b):
        return np.abs(np.hstack([a, np.ones_like(a)]) @ f - b)[:, 0]

    coef = dict(minPtNum=2, iterNum=40, thDist=0.05, thInlrRatio=0.5, REFINE=True, VERBOSE=0)
    f, inl, ns, mi, ratio = pc.ransac(x, y, coef, fit, dist, seed=1)
    assert abs(f[0, 0] - 2) < 0.01 and mi >= 150 and len(inl) == mi and calls["fit"] >= 40
    assert ratio == pytest.approx(100.0 * mi / 200)
    # failure path: T = [], zeros, message (ransac.m:77-89)
    coef_bad = dict(coef, thDist=1e-9, thInlrRatio=0.9)
    f, inl, ns, mi, ratio = pc.ransac(x, y, coef_bad, fit, dist, seed=1)
    assert f.size == 0 and inl.size == 0 and (ns, mi, ratio) == (0, 0, 0.0)


def test_argument_validation():
    import pcreg_amd as pc
    p = np.zeros((5, 3))
    with pytest.raises(KeyError):
        pc.ransac(p, p, dict(minPtNum=3, iterNum=10))                 # missing fields (ransac.m:23-29)
    with pytest.raises(ValueError):
        pc.ransac(p, np.zeros((4, 3)), dict(minPtNum=3, iterNum=10, thDist=1, thInlrRatio=0.1, REFINE=True))
    with pytest.raises(ValueError):
        pc.ransac(p, p, dict(minPtNum=3, iterNum=10, thDist=1, thInlrRatio=0.1, REFINE=True), sample_idx=np.ones((3, 3)))
    with pytest.raises(ValueError):
        pc.estimateTransform(np.zeros((5, 2)), np.zeros((5, 2)))
    with pytest.raises(ValueError):
        pc.calcDists(np.zeros((0, 0)), p, p)                          # the reference errors on [] too
    with pytest.raises(KeyError):
        pc.getMatches(np.zeros((2, 4)), np.zeros((3, 4)), dict(Metric="SAD"))
    with pytest.raises(ValueError):
        pc.getMatches(np.zeros((2, 4)), np.zeros((3, 5)), dict(UNNORMALIZE=False, CHANGE_METRIC=False, Method="Exhaustive",
                                                            MatchThreshold=10, MaxRatio=0.6, Metric="SAD", Unique=False))
    with pytest.raises(ValueError):
        pc.getMatches(np.zeros((2, 4)), np.zeros((3, 4)), dict(UNNORMALIZE=False, CHANGE_METRIC=False, Method="Exhaustive",
                                                            MatchThreshold=10, MaxRatio=0.6, Metric="cosine", Unique=False))
