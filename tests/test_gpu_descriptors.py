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


# ---- `single` data: MATLAB's single arithmetic in getLocalPoints (which keypoints survive, which points form a support) ----
@pytest.mark.parametrize("mode", ["both single", "single cloud, double keypoints", "double cloud, single keypoints"])
@pytest.mark.parametrize("seed", [0, 1])
def test_single_inputs_follow_matlabs_single_arithmetic(mode, seed, oracle_c):
    """Points planted within an ulp(single) of the sphere and of the box faces: the HIP path must take the decisions of the
    single-arithmetic oracle (tests/test_oracle_single.py shows they differ from the double ones), return DOUBLE feat / desc
    (getSpacialHistogramDescriptors.m:61-62) and exactly the oracle's counts."""
    import pcreg_amd as pc
    from test_oracle_single import OPT as SOPT, planted_scene
    pts32, kp32 = planted_scene(10 + seed)
    if mode == "both single":
        pts, kp, sm = pts32, kp32, 1
    elif mode == "single cloud, double keypoints":
        pts, kp, sm = pts32, kp32.astype(np.float64) + 1e-9, 2
    else:
        pts, kp, sm = pts32.astype(np.float64) + 1e-9, kp32, 1
    feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, SOPT)
    assert feat.dtype == np.float64 and desc.dtype == np.float64
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(np.asarray(pts, np.float64), np.asarray(kp, np.float64), SOPT, single_mode=sm)
    assert len(rfeat) >= 3
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)
    # and they are NOT the double-arithmetic supports: the row sums (support sizes) differ for some keypoint
    dfeat, ddesc = oracle_c.getSpacialHistogramDescriptors(np.asarray(pts, np.float64), np.asarray(kp, np.float64), SOPT)
    if len(dfeat) == len(rfeat):
        assert (ddesc.sum(axis=1) != rdesc.sum(axis=1)).any()


def test_device_tier_single_mode_rows(oracle_c):
    """pcreg_dev_spatial_histogram_descriptors_rows_u16 with single_mode: uint16 rows in keypoint order + survivor list."""
    import torch
    from pcreg_amd.device import DescriptorPipeline, soa
    from test_oracle_single import OPT as SOPT, planted_scene
    pts32, kp32 = planted_scene(21)
    dev = torch.device("cuda", 0)
    t = lambda a: soa(torch.from_numpy(np.ascontiguousarray(a.astype(np.float64))).to(dev))
    feat, rows, V = DescriptorPipeline(dev).describe(t(pts32), t(kp32), SOPT, compact=True, single_mode=1)
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts32.astype(np.float64), kp32.astype(np.float64), SOPT, single_mode=1)
    assert V == len(rfeat) > 0
    np.testing.assert_array_equal(feat[:V].cpu().numpy(), rfeat)
    np.testing.assert_array_equal(rows.compact(V).cpu().numpy().astype(np.float64), rdesc)


def test_descriptors_on_a_georeferenced_cloud(oracle_c):
    """Coordinates around 4e6 (ADVICE, round 2): the slab-edge margins of the candidate rows must scale with the coordinate
    magnitude, and the fp32 screens work relative to the grid's origin -- counts still equal the oracle's."""
    import pcreg_amd as pc
    off = np.array([4.2e6, -3.9e6, 5.1e5])
    pts, kp = strips(30000, 7) + off, keypoints(150, 8) + off
    feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, OPT)
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp, OPT)
    assert len(rfeat) > 30
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)


def test_k_nearest_boundary_where_square_roots_collide(oracle_c):
    """The selection sorts SQUARED distances; the reference sorts their square roots, which collide for neighbouring doubles
    (sqrt halves the relative spacing): points whose squared distances differ by one ulp can be TIED in the reference's order
    and then go by original index.  A lattice cloud (massively equal and near-equal distances) exercises exactly that."""
    import pcreg_amd as pc
    g = np.arange(-8, 9) * 0.4
    X, Y, Z = np.meshgrid(g, g * 0.6, g * 0.2, indexing="ij")
    lattice = np.column_stack([X.ravel(), Y.ravel(), Z.ravel()])
    rng = np.random.default_rng(12)
    pts = np.vstack([lattice + [20.0, 10.0, 5.0], lattice[rng.permutation(len(lattice))[:1500]] + [20.0, 10.0, 5.0]])
    pts = pts[rng.permutation(len(pts))]
    kp = np.array([[20.0, 10.0, 5.0], [20.2, 10.0, 5.0], [20.3, 10.1, 5.05], [19.9, 9.95, 5.0]])
    for k in (0.85, 0.5, 0.31):
        opt = dict(OPT, min_pts=100, max_pts=8000, thVar=[1.0, 1.0], k=k)
        feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt)
        rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp, opt)
        assert len(rfeat) == 4
        np.testing.assert_array_equal(feat, rfeat)
        np.testing.assert_array_equal(desc, rdesc)


def test_more_candidate_chunks_than_the_kept_ballots(oracle_c):
    """Two far outliers stretch the bounding box, so the grid (at most 65 536 cells) is coarse and a keypoint's rows hold the
    whole dense strip: more than 4 x 192 chunks of 64 candidates per keypoint -- the chunks past the ballots a wave keeps are
    re-tested by the list pass -- and several flushes of the ballot registers before that."""
    import pcreg_amd as pc
    rng = np.random.default_rng(77)
    n = 100000
    x = rng.uniform(0, 150, n); y = rng.uniform(-1.2, 1.2, n); z = 10 + 0.05 * x + rng.normal(0, 0.2, n)
    pts = np.vstack([np.column_stack([x, y, z]), [[30000.0, 20000.0, 15000.0], [-30000.0, -20000.0, -15000.0]]])
    kx = rng.uniform(5, 145, 12)
    kp = np.column_stack([kx, rng.uniform(-0.8, 0.8, 12), 10 + 0.05 * kx])
    opt = dict(OPT, min_pts=100, max_pts=8000)
    feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt)
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp, opt)
    assert len(rfeat) >= 6 and rdesc.sum(axis=1).max() < 8000          # supports fit the list; the CANDIDATES are the 100 000
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)


def test_speedy_descriptors_one_call_equals_the_region_loop(oracle_c, oracle_py):
    """pcreg_amd.speedyDescriptors (the reference's region walk and host-side sampling,
    speedyDescriptors.m:17-101, then ONE device call for all regions on the whole cloud) returns the keypoints, locations and
    rows of the oracle's restatement of the reference's loop (one getSpacialHistogramDescriptors call per region on the
    region's CROP): same order, all 980 counts."""
    import pcreg_amd as pc
    rng = np.random.default_rng(17)
    # an elongated, gently curved sheet bundle: several regions per axis at max_region_size 12, supports of R = 2 well filled
    n = 120_000
    x = rng.uniform(0, 40, n); y = rng.uniform(0, 25, n); s_ = rng.integers(0, 3, n)
    pts = np.column_stack([x, y, 4.0 * s_ + 0.8 * np.sin(0.4 * x) + 0.5 * np.cos(0.5 * y) + rng.normal(0, 0.05, n)])
    opt = dict(min_pts=60, max_pts=6000, R=2.0, thVar=[1.05, 1.05], k=0.85, ALIGN_POINTS=True, VERBOSE=0, max_region_size=12.0)
    sopt = dict(d=1.6)
    c_desc = lambda p, k, o: oracle_c.getSpacialHistogramDescriptors(p, k, o)
    rfeat, rdesc, rkp = oracle_py.speedyDescriptors(pts, sopt, opt, rng=np.random.default_rng(5), get_descriptors=c_desc)
    feat, desc, kp = pc.speedyDescriptors(pts, sopt, opt, rng=np.random.default_rng(5))
    bounds, nreg = oracle_py.speedy_regions(pts, opt["max_region_size"])
    assert int(np.prod(nreg)) >= 8 and len(rkp) > 500 and len(rfeat) > 100
    np.testing.assert_array_equal(kp, rkp)                        # the same keypoints in the same (region) order
    np.testing.assert_array_equal(feat, rfeat)
    np.testing.assert_array_equal(desc, rdesc)
