"""HIP RANSAC path vs the oracle, through the C ABI (pcreg_amd.api -> libpcreg_hip.so).

Bar: inlier index sets, per-iteration inlier counts, numSuccess and maxInliers
bit-exact; T within 1e-5 Frobenius (BASELINE.json north_star) -- in practice ~1e-12.
"""
import numpy as np
import pytest

from conftest import rigid_case

pytestmark = pytest.mark.gpu

T_TOL = 1e-5      # Frobenius tolerance stated by BASELINE.json:north_star


def _cmp(res_gpu, ref, n):
    T, inl, ns, mi, ratio, it1, it2 = res_gpu
    assert not ref["failed"]
    np.testing.assert_array_equal(it1, ref["inlrNum"])
    np.testing.assert_array_equal(it2, ref["inlrNum_refined"])
    assert ns == ref["numSuccess"] and mi == ref["maxInliers"]
    np.testing.assert_array_equal(inl.astype(np.int64), ref["inlierIdx"])
    assert np.linalg.norm(T - ref["T"]) < T_TOL
    assert len(inl) == mi                                  # ransac.m invariant (:58,70,92)
    assert ratio == pytest.approx(100.0 * mi / n)


def test_estimate_transform_kat():
    """testTransformEstimation.m:2-14 -- known answer [R 0; t 1]."""
    import pcreg_amd as pc
    from oracle import pcreg_oracle as o
    pts = np.array([[1, 5, 7], [4, 9, 3], [9, 3, 4], [1, 2, 4]], float)
    R = o.eul2rotm([0.1, 0.2, 0.3]); t = np.array([1.0, 2.0, 3.0])
    pts_tf = pts @ R + t
    Texp = np.eye(4); Texp[:3, :3] = R; Texp[3, :3] = t
    assert np.abs(pc.estimateTransform(pts_tf, pts) - Texp).max() < 1e-12
    assert np.abs(pc.estimateTransform(pts_tf[:3], pts[:3]) - Texp).max() < 1e-12      # 4th-point path
    assert np.abs(pc.estimateTransform(pts, pts_tf) - pc.invertTF(Texp)).max() < 1e-12
    # rank-deficient -> [] (estimateTransform.m:11-14)
    flat = pts.copy(); flat[:, 2] = 0.0
    assert pc.estimateTransform(flat, pts).size == 0
    assert pc.estimateTransform(pts[:2], pts[:2]).size == 0


@pytest.mark.parametrize("n", [3, 4, 7, 64, 65, 500])
def test_estimate_transform_vs_oracle(n, oracle_c):
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(n, 100 + n, noise=0.5, outlier_frac=0.0)
    T = pc.estimateTransform(p1, p2)
    Tref = oracle_c.estimateTransform(p1, p2)
    assert np.abs(T - Tref).max() < 1e-10


def test_calc_dists_bit_exact(oracle_c):
    import pcreg_amd as pc
    p1, p2, T = rigid_case(1000, 5)
    np.testing.assert_array_equal(pc.calcDists(T, p1, p2), oracle_c.calcDists(T, p1, p2))


@pytest.mark.parametrize("n,iters,refine", [(1000, 2000, True), (1000, 2000, False), (200, 3000, True),
                                            (1500, 1000, True), (5000, 600, True), (37, 500, True)])
def test_ransac_builtin_sampler(n, iters, refine, oracle_c):
    import pcreg_amd as pc
    p1, p2, Ttrue = rigid_case(n, n + iters)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=refine, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=9)
    res = pc.ransac(p1, p2, coef, pc.estimateTransform, pc.calcDists, seed=9, return_iter_counts=True)
    _cmp(res, ref, n)
    if refine:      # a 3-point fit alone is noise-limited
        assert np.linalg.norm(res[0] - Ttrue) < 0.1


def test_ransac_sample_table(oracle_c):
    """Host-provided index table (the MATLAB randperm stand-in, ransac.m:42-43)."""
    import pcreg_amd as pc
    n, iters = 800, 1500
    p1, p2, _ = rigid_case(n, 77)
    rng = np.random.default_rng(3)
    table = np.stack([rng.permutation(n)[:3] + 1 for _ in range(iters)]).astype(np.int32)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, sample_idx=table)
    res = pc.ransac(p1, p2, coef, sample_idx=table, return_iter_counts=True)
    _cmp(res, ref, n)


def test_ransac_min_pt_num_4(oracle_c):
    import pcreg_amd as pc
    n, iters = 600, 800
    p1, p2, _ = rigid_case(n, 78)
    rng = np.random.default_rng(4)
    table = np.stack([rng.permutation(n)[:4] + 1 for _ in range(iters)]).astype(np.int32)
    coef = dict(minPtNum=4, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, sample_idx=table)
    res = pc.ransac(p1, p2, coef, sample_idx=table, return_iter_counts=True)
    _cmp(res, ref, n)


def test_testRANSAC_script(oracle_c, capsys):
    """testRANSAC.m:13-58 + getInliersRANSAC.m:17-42 on a teapot-sized stand-in cloud
    (teapot.ply ships with MATLAB, not with the reference): T must come back as T_back."""
    import pcreg_amd as pc
    from oracle import pcreg_oracle as o
    rng = np.random.default_rng(1)
    pts = rng.uniform([-3, -2, 0], [3, 2, 3], (1000, 3))
    R = o.eul2rotm([1.5, -1.2, 0.8]); t = np.array([1.0, 2.0, 3.0])            # :13-17
    T_true2 = np.eye(4); T_true2[:3, :3] = R; T_true2[3, :3] = t               # :27-29
    T_back = np.eye(4); T_back[:3, :3] = R.T; T_back[3, :3] = -t @ R.T         # :40-42
    pts_tf2 = pc.quickTF(pts, T_true2)
    assert np.abs(pc.quickTF(pts_tf2, T_back) - pts).max() < 1e-12             # error3 ~ 0 (:45-48)
    loc1M = pts + np.random.default_rng(2).normal(0, 0.1, pts.shape)           # :52-57
    loc1S = pts_tf2                                                            # :58
    ws = pc.getInliersRANSAC(loc1M, loc1S, seed=3)
    out = capsys.readouterr().out
    assert "RANSAC succeeded" in out and "Inliers" in out                      # ransac.m:100 format
    ref = oracle_c.ransac(loc1M, loc1S, dict(o.GETINLIERS_COEFF), seed=3)
    np.testing.assert_array_equal(ws["inlierPtIdx"].astype(np.int64), ref["inlierIdx"])
    assert np.linalg.norm(ws["T"] - ref["T"]) < T_TOL
    assert np.linalg.norm(ws["T"] - T_back) < 0.05                             # noise-limited
    assert len(ws["inlierPtIdx"]) >= 990


def test_debugRANSAC_script():
    """debugRANSAC.m:2-54 -- N = 3 exact correspondences: every hypothesis is the exact transform."""
    import pcreg_amd as pc
    from oracle import pcreg_oracle as o
    rng = np.random.default_rng(11)
    pts = rng.normal(size=(3, 3))
    R = o.eul2rotm(rng.uniform(0, 2 * np.pi, 3), "XYZ"); t = rng.normal(size=3)
    T = np.eye(4); T[:3, :3] = R; T[3, :3] = t
    pts_tf = pc.quickTF(pts, T)
    coef = dict(minPtNum=3, iterNum=1000, thDist=0.1, thInlrRatio=0.5, REFINE=True, VERBOSE=0)
    T_est, inl, ns, mi, ratio = pc.ransac(pts, pts_tf, coef, pc.estimateTransform, pc.calcDists)
    # REFINE calls estimateTransform on the 3 inliers again -> same answer
    assert np.linalg.norm(T @ T_est - np.eye(4)) < 1e-10                       # est_error (:38)
    assert ns == 1000 and mi == 3 and list(inl) == [1, 2, 3]


def test_ransac_failure_is_empty(capsys):
    """ransac.m:77-89 -- nothing found: T = [], inlierIdx = [], zeros; the message is printed."""
    import pcreg_amd as pc
    rng = np.random.default_rng(5)
    p1 = rng.uniform(0, 100, (300, 3)); p2 = rng.uniform(0, 100, (300, 3))
    coef = dict(minPtNum=3, iterNum=500, thDist=1e-3, thInlrRatio=0.5, REFINE=True, VERBOSE=1)
    T, inl, ns, mi, ratio = pc.ransac(p1, p2, coef)
    assert T.size == 0 and inl.size == 0 and ns == 0 and mi == 0 and ratio == 0.0
    assert "RANSAC could not find an appropriate transformation" in capsys.readouterr().out


def test_ransac_degenerate_inputs(oracle_c):
    """Duplicate rows / collinear samples: rank-deficient hypotheses score 0 on both sides."""
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(300, 21, outlier_frac=0.2)
    p1[10:40] = p1[10]                 # 30 identical surface points
    p2[50:60] = p2[50] + np.arange(10)[:, None] * np.array([1.0, 2.0, 3.0])   # collinear model points
    coef = dict(minPtNum=3, iterNum=4000, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=2)
    res = pc.ransac(p1, p2, coef, seed=2, return_iter_counts=True)
    _cmp(res, ref, 300)


def test_ransac_batched(oracle_c):
    import pcreg_amd as pc
    sizes = [170, 333, 1000, 64, 2000]
    cases = [rigid_case(n, 300 + i) for i, n in enumerate(sizes)]
    coef = dict(minPtNum=3, iterNum=1000, thDist=0.05, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    out = pc.ransac_batched([c[0] for c in cases], [c[1] for c in cases], coef, seed=40)
    for b, (p1, p2, _) in enumerate(cases):
        ref = oracle_c.ransac(p1, p2, coef, seed=40 + b)
        T, inl, ns, mi, _ = out[b]
        np.testing.assert_array_equal(inl.astype(np.int64), ref["inlierIdx"])
        assert ns == ref["numSuccess"] and mi == ref["maxInliers"]
        assert np.linalg.norm(T - ref["T"]) < T_TOL


def test_generic_handles_run_on_host():
    """Any other handle pair takes the reference's generic loop (ransac.m:14-19 protocol)."""
    import pcreg_amd as pc
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 10, (200, 1)); y = 2 * x + 1 + rng.normal(0, 0.01, x.shape)
    y[::5] += rng.uniform(-5, 5, y[::5].shape)

    def fit(a, b):
        A = np.hstack([a, np.ones_like(a)])
        return np.linalg.lstsq(A, b, rcond=None)[0]

    def dist(f, a, b):
        return np.abs(np.hstack([a, np.ones_like(a)]) @ f - b)[:, 0]

    coef = dict(minPtNum=2, iterNum=50, thDist=0.05, thInlrRatio=0.5, REFINE=True, VERBOSE=0)
    f, inl, ns, mi, _ = pc.ransac(x, y, coef, fit, dist, seed=1)
    assert abs(f[0, 0] - 2) < 0.01 and abs(f[1, 0] - 1) < 0.05 and mi >= 150


@pytest.mark.parametrize("case", ["clean", "outliers", "norefine", "min4", "three_inliers", "planar_pts1", "planar_pts2"])
def test_staged_pipeline_equals_fused_kernel_and_oracle(case, oracle_c, debug_set):
    """n >= 4096 runs the staged chain (rs_fit1 / rs_score / rs_moments / rs_fit2); pcreg_debug_set("ransac_fused", 1)
    selects the fused tiled kernel.  Both must reproduce the oracle's counts and inlier set."""
    import pcreg_amd as pc
    n, iters = 6000, 700
    p1, p2, _ = rigid_case(n, 4242, noise=0.02, outlier_frac=0.6 if case in ("outliers", "three_inliers") else 0.05)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=case != "norefine", VERBOSE=0)
    table = None
    if case == "min4":
        rng = np.random.default_rng(8)
        table = np.stack([rng.permutation(n)[:4] + 1 for _ in range(iters)]).astype(np.int32)
        coef["minPtNum"] = 4
    if case == "three_inliers":
        # every triple (3g, 3g+1, 3g+2) has its OWN rigid motion and the table samples exactly those triples:
        # each hypothesis has its three sample points as its only inliers, thInlr = round(0.0005 n) = 3, so
        # the refit is estimateTransform's N == 3 branch (estimateTransform.m:18-37) every time
        from oracle import pcreg_oracle as o
        rng = np.random.default_rng(9)
        p2 = rng.uniform(-50, 50, (n, 3)); p1 = np.empty_like(p2)
        for g in range(n // 3):
            R = o.eul2rotm(rng.uniform(-1, 1, 3)); t = rng.uniform(-20, 20, 3)
            p1[3 * g:3 * g + 3] = p2[3 * g:3 * g + 3] @ R + t
        table = (np.arange(iters)[:, None] * 3 + np.arange(3)[None, :] + 1).astype(np.int32)
        coef.update(thDist=1e-6, thInlrRatio=0.0005)
    if case == "planar_pts1":            # rank(pts1) = 2 for every subset: estimateTransform returns [] (:11-14), nothing is found
        p1 = p1.copy(); p1[:, 2] = 0.0
    if case == "planar_pts2":            # rank(pts2) = 2 is still allowed (needs >= 2): fits exist, few inliers
        p2 = p2.copy(); p2[:, 2] = 0.0
    ref = oracle_c.ransac(p1, p2, coef, sample_idx=table, seed=21)
    out = {}
    for mode in ("staged", "fused"):
        if mode == "fused":
            debug_set("ransac_fused")
        out[mode] = pc.ransac(p1, p2, coef, sample_idx=table, seed=21, return_iter_counts=True)
    for mode, res in out.items():
        np.testing.assert_array_equal(res[5], ref["inlrNum"], err_msg=mode)
        np.testing.assert_array_equal(res[6], ref["inlrNum_refined"], err_msg=mode)
        assert res[2] == ref["numSuccess"] and res[3] == ref["maxInliers"], mode
        np.testing.assert_array_equal(np.asarray(res[1]).astype(np.int64), ref["inlierIdx"], err_msg=mode)
        if not ref["failed"]:
            assert np.linalg.norm(res[0] - ref["T"]) < T_TOL, mode
    a, b = out["staged"], out["fused"]
    # the staged chain sums the refit moments per lane in index order, the fused kernel per wave in a tree:
    # same counts and inlier sets (checked above), transforms equal to rounding
    assert a[0].shape == b[0].shape and (a[0].size == 0 or np.abs(a[0] - b[0]).max() < 1e-11)
    if case == "planar_pts1":
        assert ref["failed"] and a[0].size == 0
    if case == "three_inliers":
        assert (ref["inlrNum"] == 3).sum() > 100          # the N == 3 refit branch really ran


@pytest.mark.parametrize("n,iters,frac", [(4097, 300, 0.3), (8191, 257, 0.0), (140001, 130, 0.2), (5000, 3, 0.0), (4500, 33, 0.3)])
def test_lane_refit_sizes_and_switch(n, iters, frac, oracle_c, debug_set):
    """rs_moments_mfma_kernel (masks kept by the first scoring pass, records added under them on the int8 matrix cores) at ragged sizes,
    past the 64 x 2048 point-block limit of one scoring launch, and against the sweep it replaces
    (pcreg_debug_set("ransac_nolane", 1)): same counts, same inlier set, transforms equal to rounding."""
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(n, 31 + n, noise=0.02, outlier_frac=frac)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=3)
    res = pc.ransac(p1, p2, coef, seed=3, return_iter_counts=True)
    debug_set("ransac_nolane")
    res_sweep = pc.ransac(p1, p2, coef, seed=3, return_iter_counts=True)
    assert not ref["failed"] and ref["numSuccess"] > 0
    for r in (res, res_sweep):
        np.testing.assert_array_equal(r[5], ref["inlrNum"])
        np.testing.assert_array_equal(r[6], ref["inlrNum_refined"])
        assert r[2] == ref["numSuccess"] and r[3] == ref["maxInliers"]
        np.testing.assert_array_equal(np.asarray(r[1]).astype(np.int64), ref["inlierIdx"])
        assert np.linalg.norm(r[0] - ref["T"]) < T_TOL
    assert np.abs(res[0] - res_sweep[0]).max() < 1e-10


@pytest.mark.parametrize("big_offset", [False, True])
def test_fp32_screen_band_is_decided_in_fp64(big_offset, oracle_c, debug_set):
    """rs_score32_kernel screens distances in fp32 and re-scores a hypothesis in fp64 when any distance falls
    within its certified band around thDist.  Here half of the correspondences sit 1e-7 .. 1e-4 (relative) off the
    threshold (offset by a vector of squared length thDist (1 +- delta), no noise): deep inside the fp32 band, far
    outside fp64 rounding, so nearly every hypothesis goes through the band and fp64 must decide it; with
    big_offset the coordinates are ~4000 and the band is wide.  Counts, inlier sets and the fp64-only run
    (pcreg_debug_set("ransac_f64score", 1)) must agree with the oracle exactly."""
    import pcreg_amd as pc
    n, iters, th = 6000, 400, 0.05
    p1, p2, _ = rigid_case(n, 1234, noise=0.0, outlier_frac=0.0)
    rng = np.random.default_rng(99)
    delta = rng.choice([-1.0, 1.0], n // 2) * 10.0 ** rng.uniform(-7, -4, n // 2)
    v = rng.normal(size=(n // 2, 3)); v *= (np.sqrt(th * (1.0 + delta)) / np.linalg.norm(v, axis=1))[:, None]
    p1 = p1.copy(); p1[::2] += v
    if big_offset:
        p1 = p1 + 4000.0; p2 = p2 + np.array([3900.0, -4100.0, 4050.0])
    coef = dict(minPtNum=3, iterNum=iters, thDist=th, thInlrRatio=0.2, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=11)
    res = pc.ransac(p1, p2, coef, seed=11, return_iter_counts=True)
    debug_set("ransac_f64score")
    res64 = pc.ransac(p1, p2, coef, seed=11, return_iter_counts=True)
    assert not ref["failed"]
    near = np.abs(np.asarray(ref["inlrNum"]) - n // 2) < n // 4          # hypotheses that split the cloud at the boundary
    assert near.sum() > 10
    for r in (res, res64):
        np.testing.assert_array_equal(r[5], ref["inlrNum"])
        np.testing.assert_array_equal(r[6], ref["inlrNum_refined"])
        assert r[2] == ref["numSuccess"] and r[3] == ref["maxInliers"]
        np.testing.assert_array_equal(np.asarray(r[1]).astype(np.int64), ref["inlierIdx"])
        assert np.linalg.norm(r[0] - ref["T"]) < T_TOL


@pytest.mark.parametrize("n,iters,world", [(900, 1000, 3), (6000, 701, 2), (6000, 64, 8)])
def test_hypotheses_split_in_shares_equal_the_single_run(n, iters, world):
    """pcreg_dev_ransac_partial on every share + the MAX/SUM combine + pcreg_dev_ransac_finish
    == pcreg_dev_ransac, field for field (what the ranks of a multi-GPU job do, in one process)."""
    import ctypes as C
    import torch
    from pcreg_amd import _lib
    from pcreg_amd._lib import DevRansacResult, RansacOpts
    from pcreg_amd.device import _p, _stream
    from pcreg_amd.sharded import hypothesis_share
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    p1, p2, _ = rigid_case(n, 600 + n, noise=0.02, outlier_frac=0.5)
    t1 = torch.from_numpy(np.ascontiguousarray(p1.T)).to(dev); t2 = torch.from_numpy(np.ascontiguousarray(p2.T)).to(dev)
    nd = torch.tensor([n], dtype=torch.int32, device=dev)
    o = RansacOpts(3, iters, 0.05, 0.1, 1, 0, 77)
    ws = torch.empty(L.pcreg_dev_ransac_workspace(n, iters), dtype=torch.uint8, device=dev)
    def fetch(res, inl):
        r = DevRansacResult.from_buffer_copy(res.cpu().numpy().tobytes())
        return (np.array(r.T[:]), r.n_inliers, r.num_success, r.max_inliers, r.failed, r.n, r.winner, inl[:r.n_inliers].cpu().numpy())
    res = torch.zeros(C.sizeof(DevRansacResult), dtype=torch.uint8, device=dev); inl = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.check(L.pcreg_dev_ransac(_p(t1), _p(t2), _p(nd), n, n, C.byref(o), None, _p(res), _p(inl), _p(ws), C.c_size_t(ws.numel()), _stream()))
    single = fetch(res, inl)
    parts = []
    for r in range(world):
        begin, count = hypothesis_share(iters, r, world)
        part = torch.zeros(14, dtype=torch.int64, device=dev)
        _lib.check(L.pcreg_dev_ransac_partial(_p(t1), _p(t2), _p(nd), n, n, C.byref(o), None, begin, count, _p(part), _p(ws),
                                              C.c_size_t(ws.numel()), _stream()))
        parts.append(part.cpu())
    keys = [int(p[0]) for p in parts]
    win = int(np.argmax(keys))
    comb = parts[win].clone()
    ns_has = comb[1:2].view(torch.int32)
    ns_has[0] = sum(int(p[1:2].view(torch.int32)[0]) for p in parts)
    comb = comb.to(dev)
    res2 = torch.zeros_like(res); inl2 = torch.zeros_like(inl)
    _lib.check(L.pcreg_dev_ransac_finish(_p(t1), _p(t2), _p(nd), n, n, C.byref(o), _p(comb), _p(res2), _p(inl2), _stream()))
    split = fetch(res2, inl2)
    assert not single[4]
    for a, b in zip(single, split):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("ratio", [0.0, 1e-13, 1e-11, 1e-9, 1e-6])
@pytest.mark.parametrize("n", [5, 50, 5000])
def test_estimate_transform_rank_decision_near_planar(n, ratio):
    """estimateTransform.m:11 -- rank(pts1) < 3 -> [].  A rotated plane through the origin with thickness
    ratio * extent: MATLAB's tolerance is N * eps(sigma_max), far below what a Gram matrix can resolve
    (sigma^2 drowns at ~1e-8); the public entry decides it from the points themselves."""
    import pcreg_amd as pc
    from oracle import pcreg_oracle as o
    rng = np.random.default_rng(n)
    flat = rng.uniform(-40, 40, size=(n, 3)); flat[:, 2] = rng.standard_normal(n) * ratio * 40.0
    R = o.eul2rotm([0.7, -0.4, 1.1])
    p1 = flat @ R
    Rt = o.eul2rotm([0.3, 0.2, -0.5]); p2 = p1 @ Rt + np.array([3.0, -2.0, 5.0])
    want = o.estimateTransform(p1, p2)                      # None stands for MATLAB's []
    assert (want is None) == (o.matlab_rank(p1) < 3)
    got = pc.estimateTransform(p1, p2)
    assert (got.size == 0) == (want is None), (o.matlab_rank(p1), np.linalg.svd(p1, compute_uv=False))
    if want is not None:
        # estimateTransform.m:62 has no reflection fix: the third singular pair of H (~ratio^2 of the first) picks rotation
        # or mirror image, and below ~1e-8 rounding noise picks it -- in LAPACK as much as here.  Both map the plane onto
        # itself, so compare T where the data decide and the residual everywhere.
        if ratio >= 1e-6:
            assert np.abs(got - want).max() < 1e-5
        res = np.hstack([p2, np.ones((n, 1))]) @ got
        assert np.abs(res[:, :3] - p1).max() < 1e-5 + 200.0 * ratio and np.abs(got[:3, :3] @ got[:3, :3].T - np.eye(3)).max() < 1e-12
    # and on the model side: rank(pts2) < 2 (a line through the origin) -> []
    line = np.outer(rng.uniform(-40, 40, n), R[0]) + np.outer(rng.standard_normal(n) * ratio * 40.0, R[1])
    want = o.estimateTransform(rng.uniform(-40, 40, size=(n, 3)), line)
    got = pc.estimateTransform(rng.uniform(-40, 40, size=(n, 3)), line)
    if want is None or ratio >= 1e-11:      # below ~1e-12 H has ONE usable singular pair: MATLAB returns a T made of rounding
        assert (got.size == 0) == (want is None)    # noise there, this library []


# ---- round 4: the LDS-resident kernel with the fp32 screen (ransac_hyp32_kernel), the reference's real sizes ----------------

@pytest.mark.parametrize("n", [1000, 1365])
@pytest.mark.parametrize("big_offset", [False, True])
def test_fp32_screen_band_is_decided_in_fp64_resident(n, big_offset, oracle_c, debug_set):
    """Resident twin of test_fp32_screen_band_is_decided_in_fp64: n = 1000 (first launch class, two workgroups to a CU) and
    n = 1365 (second class).  Half of the correspondences sit 1e-7 .. 1e-4 (relative) off thDist -- inside the fp32 band,
    far outside fp64 rounding -- so nearly every hypothesis of ransac_hyp32_kernel goes through its fp64 slot re-score; the
    per-iteration counts (both passes), numSuccess, maxInliers and the inlier set must be the oracle's bits."""
    import pcreg_amd as pc
    iters, th = 600, 0.05
    p1, p2, _ = rigid_case(n, 4321 + n, noise=0.0, outlier_frac=0.0)
    rng = np.random.default_rng(98)
    delta = rng.choice([-1.0, 1.0], n // 2) * 10.0 ** rng.uniform(-7, -4, n // 2)
    v = rng.normal(size=(n // 2, 3)); v *= (np.sqrt(th * (1.0 + delta)) / np.linalg.norm(v, axis=1))[:, None]
    p1 = p1.copy(); p1[:2 * (n // 2):2] += v
    if big_offset:
        p1 = p1 + 4000.0; p2 = p2 + np.array([3900.0, -4100.0, 4050.0])
    coef = dict(minPtNum=3, iterNum=iters, thDist=th, thInlrRatio=0.2, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=11)
    res = pc.ransac(p1, p2, coef, seed=11, return_iter_counts=True)
    near = np.abs(np.asarray(ref["inlrNum"]) - n // 2) < n // 4          # hypotheses that split the cloud at the boundary
    assert near.sum() > 10
    _cmp(res, ref, n)
    debug_set("ransac_resident_f64")                                      # the round-3 kernel: fp64 only, raw coordinates in LDS
    res64 = pc.ransac(p1, p2, coef, seed=11, return_iter_counts=True)
    _cmp(res64, ref, n)
    assert np.abs(res[0] - res64[0]).max() < 1e-10


@pytest.mark.parametrize("n", [3, 4, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2047, 2048, 2049])
@pytest.mark.parametrize("refine", [True, False])
def test_resident_kernel_slot_and_class_edges(n, refine, oracle_c):
    """Slot (64), slot-pair (128) and launch-class (1024 / 2048) edges of ransac_hyp32_kernel; 2049 is the first size on the tiled kernel."""
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(n, 7000 + n, outlier_frac=0.3 if n > 20 else 0.0)
    coef = dict(minPtNum=3, iterNum=257, thDist=0.05, thInlrRatio=0.1, REFINE=refine, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=5)
    if ref["failed"]:
        T = pc.ransac(p1, p2, coef, seed=5)[0]
        assert T is None or np.size(T) == 0
        return
    _cmp(pc.ransac(p1, p2, coef, seed=5, return_iter_counts=True), ref, n)


@pytest.mark.parametrize("n,noise", [(600, 0.0), (1500, 1e-4), (700, 0.02)])
def test_resident_kernel_hypotheses_with_identical_inlier_sets_share_their_refit(n, noise, oracle_c):
    """ransac_hyp32_kernel sums a group's hypotheses with EQUAL inlier bit lists once and hands the sums (and the second-pass count)
    to the copies.  Noise-free or nearly noise-free matches under a generous threshold make every good sample select the same
    points -- groups of two, three and four copies, mixed with failing samples -- and every per-hypothesis count (first and
    refined), numSuccess, the winner and its inliers must still be the oracle's; with noise 0.02 the sets differ again."""
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(n, 9100 + n, noise=noise, outlier_frac=0.35)
    coef = dict(minPtNum=3, iterNum=1024, thDist=0.3, thInlrRatio=0.3, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=17)
    assert not ref["failed"] and ref["numSuccess"] > 50
    _cmp(pc.ransac(p1, p2, coef, seed=17, return_iter_counts=True), ref, n)
    # the batched entry point runs the same kernel per registration: two copies of the problem and a shuffled one
    perm = np.random.default_rng(1).permutation(n)
    res = pc.ransac_batched([p1, p1[perm], p1], [p2, p2[perm], p2], coef, seed=17)
    for b, r in enumerate(res):
        refb = oracle_c.ransac(p1[perm] if b == 1 else p1, p2[perm] if b == 1 else p2, coef, seed=17 + b)
        assert (r[2], r[3]) == (refb["numSuccess"], refb["maxInliers"]), b
        np.testing.assert_array_equal(np.asarray(r[1]).ravel(), refb["inlierIdx"])


@pytest.mark.parametrize("case", ["three_inliers", "min4", "planar_pts1", "planar_pts2", "loose_sample"])
def test_resident_kernel_uncertified_refits(case, oracle_c):
    """The refits ransac_hyp32_kernel cannot run from the fifteen masked sums -- exactly three inliers (estimateTransform's
    N == 3 branch), minPtNum = 4 (no rank certificate), planar point sets, and samples whose own points are NOT inliers of
    their fit (thDist far below the noise) -- take the 27-sum pass on the raw coordinates; same bits as the oracle."""
    import pcreg_amd as pc
    n, iters = 900, 500
    p1, p2, _ = rigid_case(n, 4243, noise=0.02, outlier_frac=0.6 if case == "three_inliers" else 0.05)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    table = None
    if case == "min4":
        rng = np.random.default_rng(8)
        table = np.stack([rng.permutation(n)[:4] + 1 for _ in range(iters)]).astype(np.int32)
        coef["minPtNum"] = 4
    if case == "three_inliers":
        from oracle import pcreg_oracle as o
        rng = np.random.default_rng(9)
        iters = n // 3
        p2 = rng.uniform(-50, 50, (n, 3)); p1 = np.empty_like(p2)
        for g in range(n // 3):
            R = o.eul2rotm(rng.uniform(-1, 1, 3)); t = rng.uniform(-20, 20, 3)
            p1[3 * g:3 * g + 3] = p2[3 * g:3 * g + 3] @ R + t
        table = (np.arange(iters)[:, None] * 3 + np.arange(3)[None, :] + 1).astype(np.int32)
        coef.update(thDist=1e-6, thInlrRatio=0.003, iterNum=iters)
    if case == "planar_pts1":
        p1 = p1.copy(); p1[:, 2] = 0.0
    if case == "planar_pts2":
        p2 = p2.copy(); p2[:, 2] = 0.0
    if case == "loose_sample":            # noise 0.02 -> squared residuals ~1e-3; thDist 2e-4 keeps ~10 % and the sample points mostly out
        coef.update(thDist=2e-4, thInlrRatio=0.02)
    ref = oracle_c.ransac(p1, p2, coef, sample_idx=table, seed=21)
    res = pc.ransac(p1, p2, coef, sample_idx=table, seed=21, return_iter_counts=True)
    np.testing.assert_array_equal(res[5], ref["inlrNum"])
    np.testing.assert_array_equal(res[6], ref["inlrNum_refined"])
    assert res[2] == ref["numSuccess"] and res[3] == ref["maxInliers"]
    if ref["failed"]:
        assert res[0] is None or np.size(res[0]) == 0
    else:
        np.testing.assert_array_equal(np.asarray(res[1]).astype(np.int64), ref["inlierIdx"])
        assert np.linalg.norm(res[0] - ref["T"]) < T_TOL
    if case == "three_inliers":
        assert (ref["inlrNum"] == 3).sum() > 100
    if case == "loose_sample":
        assert ref["numSuccess"] > 0


def test_ransac_batched_across_launch_classes(oracle_c):
    """One batch whose registrations fall into both launch classes of ransac_hyp32_kernel (<= 1024, <= 2048) and above them
    (the fp64 kernel on the raw coordinates from L2), plus the degenerate sizes 0-2."""
    import pcreg_amd as pc
    sizes = [2, 170, 1024, 1025, 2048, 2500, 640, 3000, 33]
    cases = [rigid_case(max(n, 3), 900 + i)[:2] for i, n in enumerate(sizes)]
    cases = [(c[0][:n], c[1][:n]) for c, n in zip(cases, sizes)]
    coef = dict(minPtNum=3, iterNum=300, thDist=0.05, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    out = pc.ransac_batched([c[0] for c in cases], [c[1] for c in cases], coef, seed=70)
    for b, (p1, p2) in enumerate(cases):
        T, inl, ns, mi, _ = out[b]
        if sizes[b] < 3:
            assert T is None or np.size(T) == 0
            continue
        ref = oracle_c.ransac(p1, p2, coef, seed=70 + b)
        np.testing.assert_array_equal(inl.astype(np.int64), ref["inlierIdx"])
        assert ns == ref["numSuccess"] and mi == ref["maxInliers"]
        assert np.linalg.norm(T - ref["T"]) < T_TOL


@pytest.mark.parametrize("n,iters", [(3000, 7000), (2049, 9800), (4095, 4900)])
def test_staged_chain_between_2049_and_4095_when_the_work_pays(n, iters, oracle_c, debug_set):
    """Round 4: one registration of 2049 <= n < 4096 correspondences runs the staged chain (fp32-screened scoring) when
    n x iterNum >= 2 x 10^7, the fp64 tiled kernel otherwise (the chain's fixed cost); both give the oracle's bits."""
    import pcreg_amd as pc
    p1, p2, _ = rigid_case(n, 77 + n, outlier_frac=0.4)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)
    ref = oracle_c.ransac(p1, p2, coef, seed=9)
    _cmp(pc.ransac(p1, p2, coef, seed=9, return_iter_counts=True), ref, n)
    debug_set("ransac_fused")                                             # the tiled kernel on the same problem
    _cmp(pc.ransac(p1, p2, coef, seed=9, return_iter_counts=True), ref, n)
