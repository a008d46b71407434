"""getLocalPoints.m:8-35 as a function of its own (pcreg_get_local_points) against the oracle: double and MATLAB's single arithmetic
(keypoint single / only the cloud single), both gates, points planted within one ulp of the box faces and of R, the cloud's order."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(seed, n=20000):
    rng = np.random.default_rng(seed)
    return rng.uniform([0, 0, 0], [40, 30, 20], (n, 3))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("seed", [0, 1])
def test_get_local_points_equals_oracle(mode, seed, oracle_py):
    import pcreg_amd as pc
    pts = _cloud(seed)
    c = np.array([21.3, 14.1, 9.7])
    R = 3.5
    # plant points on the sphere and on the box faces, one ulp to either side (in the arithmetic of the mode)
    f = np.float32 if mode else np.float64
    cf = c.astype(f)
    extra = []
    for axis in range(3):
        for sgn in (-1, 1):
            face = f(cf[axis] + f(sgn * R)) if mode != 2 else f(c[axis] + sgn * R)
            for v in (np.nextafter(face, f(-np.inf)), face, np.nextafter(face, f(np.inf))):
                p = cf.copy(); p[axis] = v
                extra.append(p.astype(np.float64))
            q = cf.astype(np.float64).copy(); q[axis] += sgn * R * (1 - 1e-7 if mode else 1 - 1e-15)
            extra.append(q)
    pts = np.vstack([pts, np.array(extra)])
    if mode:
        pts = pts.astype(np.float32).astype(np.float64) if mode == 1 else pts
    P = pts.astype(np.float32) if mode == 2 else pts
    cc = c.astype(np.float32) if mode == 1 else c
    if mode == 1:
        P = pts.astype(np.float32)                     # a double cloud meets a single centre: converted in the comparison anyway
    ref_p, ref_d = oracle_py.getLocalPoints(P, R, cc, 10, np.inf, single_mode=mode)
    got_p, got_d = pc.getLocalPoints(P, R, cc, 10, np.inf)
    assert ref_p is not None and len(ref_p) > 100
    np.testing.assert_array_equal(got_p.astype(np.float64), ref_p)
    np.testing.assert_array_equal(got_d.astype(np.float64), ref_d)
    assert got_p.dtype == (np.float32 if mode else np.float64)
    # gates: too few in the ball, too many in the ball, nothing at all
    n = len(ref_p)
    for lo, hi in ((n + 1, np.inf), (0, n - 1), (0, np.inf)):
        g, _ = pc.getLocalPoints(P, R, cc, lo, hi)
        r, _ = oracle_py.getLocalPoints(P, R, cc, lo, hi, single_mode=mode)
        assert (len(g) == 0) == (r is None)
    g, gd = pc.getLocalPoints(P, 1e-3, np.array([500.0, 500, 500]).astype(cc.dtype), 0, np.inf)
    assert g.shape == (0, 3) and gd.shape == (0,)
