"""Committed golden fixtures (tests/golden/*.npz, made by oracle/gen_golden.py): the oracle
must reproduce them on CPU; the HIP path must reproduce them on the GPU."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
           MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)


def _coef(z):
    c = z["coef"]
    return dict(minPtNum=int(c[0]), iterNum=int(c[1]), thDist=float(c[2]), thInlrRatio=float(c[3]), REFINE=bool(c[4]), VERBOSE=0)


@pytest.mark.parametrize("name", ["ransac_a", "ransac_b"])
def test_oracle_reproduces_ransac(name, oracle_c, oracle_py):
    z = np.load(os.path.join(G, name + ".npz"))
    assert (oracle_py.sample_table(len(z["pts1"]), int(z["coef"][1]), 3, int(z["seed"])) == z["sample_idx"]).all()
    r = oracle_c.ransac(z["pts1"], z["pts2"], _coef(z), sample_idx=z["sample_idx"])
    np.testing.assert_array_equal(r["inlierIdx"], z["inlierIdx"])
    np.testing.assert_array_equal(r["inlrNum"], z["inlrNum"])
    np.testing.assert_array_equal(r["inlrNum_refined"], z["inlrNum_refined"])
    assert r["numSuccess"] == int(z["numSuccess"]) and r["maxInliers"] == int(z["maxInliers"])
    assert np.abs(r["T"] - z["T"]).max() < 1e-10


def test_oracle_reproduces_the_rest(oracle_c):
    z = np.load(os.path.join(G, "estimate_transform.npz"))
    for i in range(3):
        assert np.abs(oracle_c.estimateTransform(z[f"p1_{i}"], z[f"p2_{i}"]) - z[f"T_{i}"]).max() < 1e-10
    z = np.load(os.path.join(G, "match_points.npz"))
    idx, dist = oracle_c.knn2_points_f32(z["surf"], z["model"])
    np.testing.assert_array_equal(idx, z["idx"])
    np.testing.assert_array_equal(oracle_c.match_points_f32(z["surf"], z["model"], float(z["thr"]), float(z["ratio"]), True), z["pairs"])
    z = np.load(os.path.join(G, "get_matches.npz"))
    np.testing.assert_array_equal(oracle_c.getMatches(z["descSurface"], z["descModel"], PAR), z["matches_sad"])
    np.testing.assert_array_equal(oracle_c.getMatches(z["descSurface"], z["descModel"], dict(PAR, Metric="SSD")), z["matches_ssd"])
    z = np.load(os.path.join(G, "align_points_knn.npz"))
    for C1 in (0, 1):
        for C2 in (0, 1):
            al, co, c = oracle_c.AlignPoints_KNN(z["pts"], bool(C1), bool(C2))
            assert np.abs(al - z[f"aligned_{C1}{C2}"]).max() < 1e-9 and np.abs(co - z[f"coeff_{C1}{C2}"]).max() < 1e-11


DOPT = dict(min_pts=60, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)


def test_oracle_reproduces_descriptors(oracle_c):
    z = np.load(os.path.join(G, "descriptors.npz"))
    f, d = oracle_c.getSpacialHistogramDescriptors(z["pts"], z["sample_pts"], DOPT)
    np.testing.assert_array_equal(f, z["feat"]); np.testing.assert_array_equal(d, z["desc"].astype(np.float64))
    f, d = oracle_c.getSpacialHistogramDescriptors(z["pts"], z["sample_pts"], dict(DOPT, ALIGN_POINTS=False))
    np.testing.assert_array_equal(f, z["feat_noalign"]); np.testing.assert_array_equal(d, z["desc_noalign"].astype(np.float64))


@pytest.mark.gpu
def test_hip_reproduces_descriptors():
    import pcreg_amd as pc
    z = np.load(os.path.join(G, "descriptors.npz"))
    f, d = pc.getSpacialHistogramDescriptors(z["pts"], z["sample_pts"], DOPT)
    np.testing.assert_array_equal(f, z["feat"]); np.testing.assert_array_equal(d, z["desc"].astype(np.float64))
    f, d = pc.getSpacialHistogramDescriptors(z["pts"], z["sample_pts"], dict(DOPT, ALIGN_POINTS=False))
    np.testing.assert_array_equal(f, z["feat_noalign"]); np.testing.assert_array_equal(d, z["desc_noalign"].astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ransac_a", "ransac_b"])
def test_hip_reproduces_ransac(name):
    import pcreg_amd as pc
    z = np.load(os.path.join(G, name + ".npz"))
    for kw in (dict(sample_idx=z["sample_idx"]), dict(seed=int(z["seed"]))):     # table and built-in sampler agree
        T, inl, ns, mi, _, it1, it2 = pc.ransac(z["pts1"], z["pts2"], _coef(z), return_iter_counts=True, **kw)
        np.testing.assert_array_equal(inl.astype(np.int64), z["inlierIdx"])
        np.testing.assert_array_equal(it1, z["inlrNum"])
        np.testing.assert_array_equal(it2, z["inlrNum_refined"])
        assert ns == int(z["numSuccess"]) and mi == int(z["maxInliers"])
        assert np.linalg.norm(T - z["T"]) < 1e-5


@pytest.mark.gpu
def test_hip_reproduces_the_rest():
    import pcreg_amd as pc
    z = np.load(os.path.join(G, "estimate_transform.npz"))
    for i in range(3):
        assert np.abs(pc.estimateTransform(z[f"p1_{i}"], z[f"p2_{i}"]) - z[f"T_{i}"]).max() < 1e-9
    z = np.load(os.path.join(G, "match_points.npz"))
    idx, dist = pc.knn2_points(z["surf"], z["model"])
    np.testing.assert_array_equal(idx, z["idx"])
    np.testing.assert_array_equal(pc.match_points(z["surf"], z["model"], float(z["thr"]), float(z["ratio"]), True), z["pairs"])
    z = np.load(os.path.join(G, "get_matches.npz"))
    np.testing.assert_array_equal(pc.getMatches(z["descSurface"], z["descModel"], PAR), z["matches_sad"])
    np.testing.assert_array_equal(pc.getMatches(z["descSurface"], z["descModel"], dict(PAR, Metric="SSD")), z["matches_ssd"])
    z = np.load(os.path.join(G, "align_points_knn.npz"))
    for C1 in (0, 1):
        for C2 in (0, 1):
            al, co, c = pc.AlignPoints_KNN(z["pts"], bool(C1), bool(C2))
            assert np.abs(al - z[f"aligned_{C1}{C2}"]).max() < 1e-9 and np.abs(co - z[f"coeff_{C1}{C2}"]).max() < 1e-9
