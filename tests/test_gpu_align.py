"""AlignPoints_KNN on the GPU vs the oracle (floating point: tolerance 1e-9 absolute on
coordinates of O(100), stated here; sign decisions must agree exactly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _support(n, seed):
    rng = np.random.default_rng(seed)
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    return (rng.normal(size=(n, 3)) * np.array([3.0, 1.5, 0.4])) @ A + rng.uniform(-50, 50, 3)


@pytest.mark.parametrize("n", [2, 3, 20, 500, 777, 6000])
@pytest.mark.parametrize("C1,C2", [(False, False), (True, False), (False, True), (True, True)])
def test_align_points_knn(n, C1, C2, oracle_c):
    import pcreg_amd as pc
    X = _support(n, n)
    al, co, c = pc.AlignPoints_KNN(X, C1, C2)
    ral, rco, rc = oracle_c.AlignPoints_KNN(X, C1, C2)
    assert np.abs(c.ravel() - rc).max() < TOL
    if n >= 20:
        assert np.abs(co - rco).max() < TOL
        assert np.abs(al - ral).max() < TOL
        return
    # n = 2, 3: the covariance of the K kept points is singular.  The principal axis is defined (and must agree);
    # the basis of the null space (n = 2) and the sign of the plane's normal (n = 3: the vote of :45-50 counts
    # projections that are zero up to rounding) are whatever the eigen-solver's rounding makes them -- in MATLAB's
    # LAPACK as in the oracle's Jacobi -- so compare what the input determines
    assert np.abs(co[:, 0] - rco[:, 0]).max() < TOL and np.abs(al[:, 0] - ral[:, 0]).max() < TOL
    assert np.abs(co.T @ co - np.eye(3)).max() < TOL and np.abs(al - X @ co).max() < TOL
    if n == 3:
        assert np.abs(np.abs(co) - np.abs(rco)).max() < TOL


def test_align_points_knn_ties_and_varargin(oracle_c):
    import pcreg_amd as pc
    rng = np.random.default_rng(1)
    X = rng.integers(-3, 4, (400, 3)).astype(float)         # many equal distances at the K-th boundary
    al, co, c = pc.AlignPoints_KNN(X)
    ral, rco, rc = oracle_c.AlignPoints_KNN(X)
    assert np.abs(co - rco).max() < TOL and np.abs(al - ral).max() < TOL
    # a single flag is ignored, like AlignPoints_KNN.m:8-14
    al1, co1, _ = pc.AlignPoints_KNN(X, True)
    assert np.array_equal(co1, co)


def test_align_points_knn_batched(oracle_c):
    import pcreg_amd as pc
    sups = [_support(n, 50 + n) for n in (500, 1, 1400, 6000, 33)]
    al, co, c, status = pc.AlignPoints_KNN_batched(sups)
    assert list(status) == [0, 1, 0, 0, 0]
    for b, X in enumerate(sups):
        if status[b]:
            continue
        ral, rco, rc = oracle_c.AlignPoints_KNN(X)
        assert np.abs(co[b] - rco).max() < TOL and np.abs(al[b] - ral).max() < TOL and np.abs(c[b] - rc).max() < TOL


@pytest.mark.parametrize("n", [1024, 1025, 2048, 2049, 3000, 3072, 3073, 4096, 4097, 8192])
def test_align_points_knn_every_kernel_variant(n, oracle_c):
    """The register-resident kernel is instantiated per support size (points per thread x threads): both sides of
    every switch point, up to the largest support it takes."""
    import pcreg_amd as pc
    X = _support(n, 9000 + n)
    al, co, c = pc.AlignPoints_KNN(X)
    ral, rco, rc = oracle_c.AlignPoints_KNN(X)
    assert np.abs(c.ravel() - rc).max() < TOL and np.abs(co - rco).max() < TOL and np.abs(al - ral).max() < TOL


def test_align_points_knn_support_too_large_fails_loudly():
    import pcreg_amd as pc
    from pcreg_amd._lib import PcregError
    with pytest.raises(PcregError):
        pc.AlignPoints_KNN(_support(8193, 1))


def test_align_points_knn_ties_in_large_support(oracle_c):
    """Equal distances at the K-th boundary in a support that spans several rows of points per thread (the tie
    ranking walks rows, then threads)."""
    import pcreg_amd as pc
    rng = np.random.default_rng(5)
    X = rng.integers(-6, 7, (3000, 3)).astype(float)
    al, co, c = pc.AlignPoints_KNN(X)
    ral, rco, rc = oracle_c.AlignPoints_KNN(X)
    assert np.abs(co - rco).max() < TOL and np.abs(al - ral).max() < TOL
