"""`single` data through getLocalPoints (getLocalPoints.m:8-31): when either input is single, MATLAB evaluates the open box
test, pts_cube - c, sqrt(x^2 + y^2 + z^2) and dists < R in single.  Those are element-wise, hence reproducible: both oracles
restate them (single_mode 1: keypoints single; 2: cloud single, keypoints double), and must agree with each other bit for
bit -- on clouds with points planted within an ulp(single) of the sphere and of the box, where single and double arithmetic
decide differently.  CPU only."""
import numpy as np
import pytest

OPT = dict(min_pts=40, max_pts=6000, R=3.5, thVar=[1.0, 1.0], k=0.85, ALIGN_POINTS=True, VERBOSE=0)


def planted_scene(seed, n_fill=1500, n_plant=600, S=6, offset=(40.0, 25.0, 50.0)):
    """A cloud around S keypoints: a filling of the spheres, plus points planted ON the sphere |p - c| = R and on the box faces
    c +- R, then nudged by -2 .. +2 ulp(single) along the radius / the axis."""
    rng = np.random.default_rng(seed)
    R = np.float32(OPT["R"])
    kp = (rng.uniform(-8, 8, (S, 3)) + np.array(offset)).astype(np.float32)
    pts = []
    for c in kp:
        fill = rng.normal(size=(n_fill, 3)) * [2.2, 1.4, 0.6] + c
        u = rng.normal(size=(n_plant, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        on = (c.astype(np.float64) + np.float64(R) * u).astype(np.float32)
        for _ in range(int(rng.integers(0, 3))):
            on = np.nextafter(on, (on + np.sign(u).astype(np.float32) * np.float32(1e3)).astype(np.float32))
        face = rng.uniform(-3, 3, (n_plant // 4, 3)).astype(np.float32) + c
        ax = rng.integers(0, 3, len(face)); sg = rng.choice([-1.0, 1.0], len(face)).astype(np.float32)
        face[np.arange(len(face)), ax] = c[ax] + sg * R
        pts += [fill.astype(np.float32), on, face]
    return np.vstack(pts).astype(np.float32), kp


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_both_oracles_agree_in_single_arithmetic(mode, seed, oracle_py, oracle_c):
    pts32, kp32 = planted_scene(seed)
    kp = kp32.astype(np.float64) if mode == 1 else kp32.astype(np.float64) + 1e-9       # mode 2: genuinely double keypoints
    f_py, d_py = oracle_py.getSpacialHistogramDescriptors(pts32.astype(np.float64), kp, OPT, single_mode=mode)
    f_c, d_c = oracle_c.getSpacialHistogramDescriptors(pts32.astype(np.float64), kp, OPT, single_mode=mode)
    assert len(f_py) == len(f_c) > 0
    np.testing.assert_array_equal(f_py, f_c)
    # the supports are the same SETS of single pts_rel values, so the row sums (support sizes inside the histogram) agree exactly;
    # the counts themselves agree wherever no point sits within rounding of a bin edge of the two different eigen-solvers' frames
    np.testing.assert_array_equal(d_py.sum(axis=1), d_c.sum(axis=1))
    assert (d_py != d_c).sum() <= 4 * len(d_py)


def test_single_and_double_arithmetic_decide_differently_on_planted_points():
    """The test data is meaningful: for the planted points, `dists < R` evaluated in single (as MATLAB does for single data)
    and in double (what round 2 did after widening) disagree for a visible share of them."""
    pts32, kp32 = planted_scene(3)
    R = OPT["R"]
    disagree = near = 0
    for c in kp32:
        rel32 = pts32 - c
        d32 = np.sqrt((rel32[:, 0] * rel32[:, 0] + rel32[:, 1] * rel32[:, 1]) + rel32[:, 2] * rel32[:, 2])
        rel64 = pts32.astype(np.float64) - c.astype(np.float64)
        d64 = np.sqrt((rel64 * rel64).sum(axis=1))
        near += (np.abs(d64 - R) < 1e-5).sum()
        disagree += ((d32 < np.float32(R)) != (d64 < R)).sum()
    assert near > 1000 and disagree >= 10


def test_single_support_values_are_matlabs_single_pts_rel(oracle_py):
    pts32, kp32 = planted_scene(4, n_fill=300, n_plant=50, S=1)
    rel, d = oracle_py.getLocalPoints(pts32.astype(np.float64), OPT["R"], kp32[0].astype(np.float64), 1, 10**9, single_mode=1)
    assert np.array_equal(rel, rel.astype(np.float32).astype(np.float64))               # every coordinate is a single value
    keep = np.isin((pts32 - kp32[0]).astype(np.float64).view([("", np.float64)] * 3), rel.view([("", np.float64)] * 3)).ravel()
    assert keep.sum() >= len(rel)                                                        # ... namely single(p) - single(c)
    assert (d < np.float32(OPT["R"])).all() and np.array_equal(d, d.astype(np.float32).astype(np.float64))
