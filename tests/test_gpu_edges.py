"""Edge cases through the C ABI: empty and tiny inputs, leading dimensions larger than the
row count (MATLAB sub-matrices), sizes that straddle every tile boundary."""
import ctypes as C

import numpy as np
import pytest

from conftest import rigid_case

pytestmark = pytest.mark.gpu


def test_empty_and_tiny_searches(oracle_c):
    import pcreg_amd as pc
    rng = np.random.default_rng(0)
    m = rng.uniform(0, 10, (50, 3)).astype(np.float32)
    q = rng.uniform(0, 10, (7, 3)).astype(np.float32)
    idx, dist = pc.knn2_points(q[:0], m)
    assert idx.shape == (0, 2)
    assert pc.match_points(q, m[:0], 1.0, 0.9).shape == (0, 2)
    assert pc.match_points(q[:0], m, 1.0, 0.9).shape == (0, 2)
    for M in (1, 2, 3, 4, 5, 16, 17):
        idx, dist = pc.knn2_points(q, m[:M])
        ridx, rdist = oracle_c.knn2_points_f32(q, m[:M])
        np.testing.assert_array_equal(idx, ridx)
        np.testing.assert_array_equal(dist, rdist)


def test_ransac_too_few_points_fails_cleanly():
    import pcreg_amd as pc
    coef = dict(minPtNum=3, iterNum=50, thDist=0.5, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    for n in (0, 1, 2):
        p = np.random.default_rng(n).normal(size=(n, 3))
        T, inl, ns, mi, ratio = pc.ransac(p, p, coef)
        assert T.size == 0 and inl.size == 0 and ns == 0 and mi == 0


def test_leading_dimension_larger_than_rows(oracle_c):
    """A MATLAB sub-matrix view: ld = rows of the parent array."""
    from pcreg_amd._lib import RansacOpts, check, lib
    n, ld = 500, 777
    p1, p2, _ = rigid_case(n, 31)
    big1 = np.full((ld, 3), np.nan, order="F"); big2 = np.full((ld, 3), np.nan, order="F")
    big1[:n] = p1; big2[:n] = p2
    o = RansacOpts(3, 400, 0.05, 0.1, 1, 0, 5)
    T = np.zeros(16); inl = np.zeros(n, np.int32)
    ni, ns, mi, fl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    check(lib().pcreg_ransac(dp(big1), dp(big2), n, ld, C.byref(o), None, dp(T), inl.ctypes.data_as(C.POINTER(C.c_int32)),
                             C.byref(ni), C.byref(ns), C.byref(mi), C.byref(fl), None, None))
    ref = oracle_c.ransac(p1, p2, dict(minPtNum=3, iterNum=400, thDist=0.05, thInlrRatio=0.1, REFINE=True), seed=5)
    np.testing.assert_array_equal(inl[:ni.value], ref["inlierIdx"])
    assert ns.value == ref["numSuccess"] and np.linalg.norm(T.reshape(4, 4, order="F") - ref["T"]) < 1e-5
    # estimateTransform through the same strided view
    T2 = np.zeros(16); empty = C.c_int()
    check(lib().pcreg_estimate_transform(dp(big1), dp(big2), n, ld, dp(T2), C.byref(empty)))
    assert not empty.value and np.abs(T2.reshape(4, 4, order="F") - oracle_c.estimateTransform(p1, p2)).max() < 1e-10
    # bad ld is an argument error, not a crash
    assert lib().pcreg_estimate_transform(dp(big1), dp(big2), n, n - 1, dp(T2), C.byref(empty)) == 1


@pytest.mark.parametrize("n", [1364, 1365, 1366, 2047, 2048, 2049, 4097])
def test_ransac_around_the_lds_and_tile_boundaries(n, oracle_c):
    """n = 1365 is the last size with LDS-resident correspondences; 1024 / 2048 are tile edges."""
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(n, n)
    coef = dict(minPtNum=3, iterNum=300, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=1)
    T, inl, ns, mi, _, it1, it2 = pc.ransac(p1, p2, coef, seed=1, return_iter_counts=True)
    np.testing.assert_array_equal(it1, ref["inlrNum"]); np.testing.assert_array_equal(it2, ref["inlrNum_refined"])
    np.testing.assert_array_equal(inl.astype(np.int64), ref["inlierIdx"])
    assert ns == ref["numSuccess"] and np.linalg.norm(T - ref["T"]) < 1e-5


def test_ransac_low_inlier_ratio_and_iter_counts_not_multiple_of_anything(oracle_c):
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(3000, 9, outlier_frac=0.9)
    coef = dict(minPtNum=3, iterNum=1237, thDist=0.05, thInlrRatio=0.05, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=3)
    T, inl, ns, mi, _, it1, it2 = pc.ransac(p1, p2, coef, seed=3, return_iter_counts=True)
    np.testing.assert_array_equal(it1, ref["inlrNum"]); np.testing.assert_array_equal(it2, ref["inlrNum_refined"])
    np.testing.assert_array_equal(inl.astype(np.int64), ref["inlierIdx"])
    assert ns == ref["numSuccess"] == int(np.sum(ref["inlrNum_refined"] >= round(0.05 * 3000)))
