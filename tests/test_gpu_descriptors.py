"""getSpacialHistogramDescriptors on the GPU vs the oracle: the surviving keypoint set and
every one of the 980 integer counts must match exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def strips(P, seed, ns=8):
    """Ridge-like strips: supports are elongated, so they pass the eigenvalue-ratio test (:118-121)."""
    rng = np.random.default_rng(seed)
    per = P // ns
    out = []
    for s in range(ns):
        x = rng.uniform(0, 60, per); y = 6 * s + rng.uniform(-1.2, 1.2, per); z = 10 + 0.1 * x * np.sin(s) + rng.normal(0, 0.25, per)
        out.append(np.column_stack([x, y, z]))
    return np.vstack(out)


def keypoints(S, seed, ns=8):
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(-2, 62, S), 6 * rng.integers(0, ns, S) + rng.uniform(-1.5, 1.5, S), rng.uniform(8, 17, S)])


OPT = dict(min_pts=150, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("k", [0.85, "all", 0.95, 0.5])
def test_descriptors_match_oracle(align, k, oracle_c):
    import pcreg_amd as pc
    pts, kp = strips(40000, 0), keypoints(300, 1)
    opt = dict(OPT, ALIGN_POINTS=align, k=k)
    feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt)
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp, opt)
    if k != 0.5:                       # half the support is too round for thVar = [3, 1.5]: nothing survives
        assert len(rfeat) > 50
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)
    assert desc.shape[1] == 980 and (desc.sum(axis=1) >= opt["min_pts"] - 1).all()


def test_descriptors_vs_numpy_oracle_and_limits(oracle_py, capsys):
    import pcreg_amd as pc
    pts, kp = strips(16000, 3), keypoints(60, 4)
    opt = dict(OPT, min_pts=60, max_pts=400, VERBOSE=1)          # max_pts rejects the dense supports
    feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt)
    assert "Calculated descriptors in" in capsys.readouterr().out      # getSpacialHistogramDescriptors.m:181
    rfeat, rdesc = oracle_py.getSpacialHistogramDescriptors(pts, kp, opt)
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)
    # nothing survives an impossible minimum; empty inputs give empty outputs
    f0, d0 = pc.getSpacialHistogramDescriptors(pts, kp, dict(OPT, min_pts=10**6))
    assert f0.shape == (0, 3) and d0.shape == (0, 980)
    f1, d1 = pc.getSpacialHistogramDescriptors(pts, kp[:0], OPT)
    assert f1.shape == (0, 3) and d1.shape == (0, 980)


def test_descriptors_duplicate_points_and_far_keypoints(oracle_c):
    """Duplicated cloud points create exact distance ties at the K-th boundary (stable-sort
    order decides); keypoints far outside the cloud must simply vanish."""
    import pcreg_amd as pc
    base = strips(12000, 5)
    pts = np.vstack([base, base[::3]])                         # every third point twice
    kp = np.vstack([keypoints(120, 6), np.array([[500.0, 500.0, 500.0], [-300.0, 0.0, 10.0]])])
    opt = dict(OPT, min_pts=100)
    feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt)
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp, opt)
    assert len(rfeat) > 20
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)
