"""BASELINE.json's configs at their REAL sizes on the GPU, with oracle spot checks (the pattern of
test_search_at_full_size_spot_check: the oracle cannot redo 10^10 pairs, but every query / keypoint / crop is
independent, so a random sample checked exhaustively on the host cores pins the whole run).

  cfg 2  getMatches, D = 981 (980 counts + the UNNORMALIZE column), 50 k x 200 k, SAD, Unique
  cfg 4  getSpacialHistogramDescriptors, 1 M keypoints on a 1 M-point cloud
  cfg 5  64 crops x 50 k surface points vs one 1 M-point model, match + RANSAC per crop
(cfg 3's 2 M-point model on one GPU is test_gpu_pipeline.py::test_search_at_full_size_spot_check; its 8-GPU form
needs eight GPUs and is bench.py --gpus 8.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CORES = min(len(os.sched_getaffinity(0)), 16)


def test_cfg2_get_matches_50k_x_200k_d981(oracle_c, oracle_py):
    import pcreg_amd as pc
    Q, M, D = 50_000, 200_000, 980
    rng = np.random.default_rng(0)
    dM = rng.poisson(3.0, (M, D)).astype(np.float64)
    dS = rng.poisson(3.0, (Q, D)).astype(np.float64)
    k = Q // 2                                                   # half of the surface rows are noisy copies of model rows
    src = rng.choice(M, k, replace=False)
    dS[:k] = dM[src] + rng.poisson(0.2, (k, D))
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
               MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)
    pairs = pc.getMatches(dS, dM, par)                           # P x 2 uint32, 1-based, ascending in column 1
    assert pairs.dtype == np.uint32 and pairs.shape[1] == 2 and len(pairs) > k // 2
    assert np.all(np.diff(pairs[:, 0].astype(np.int64)) > 0)
    assert len(np.unique(pairs[:, 1])) == len(pairs)             # Unique: a model row is matched at most once
    # ---- oracle, 200 random surface rows against ALL model rows ----------------------------------------------
    pS, pM = oracle_py.preprocess_descriptors(dS, dM, par)       # getMatches.m:22-37 on the full arrays (the constant is global)
    del dS, dM
    nS, nM = oracle_py._normalize_rows(pS), oracle_py._normalize_rows(pM)
    del pS, pM
    sel = np.sort(rng.choice(Q, 200, replace=False))
    par_nu = dict(Metric="SAD", MatchThreshold=10, MaxRatio=0.99, Unique=False, Prenormalized=True)
    cand, _ = oracle_c.matchFeatures(nS[sel], nM, par_nu, nthreads=CORES)          # forward matches of the sample
    cand_of = {int(sel[i - 1]): int(j) - 1 for i, j in cand}
    got_of = {int(i) - 1: int(j) - 1 for i, j in pairs if (int(i) - 1) in set(sel.tolist())}
    n_unique_drop = 0
    for i in sel.tolist():
        if i in got_of:
            assert cand_of.get(i) == got_of[i], f"surface row {i}: GPU matched {got_of[i]}, oracle forward match {cand_of.get(i)}"
        if i in cand_of:                                          # the Unique rule for this candidate, over ALL 50 k queries
            col = np.abs(nS - nM[cand_of[i]]).sum(axis=1)
            best = int(np.argmin(col))
            if best == i:
                assert i in got_of, f"surface row {i} is the best query of model row {cand_of[i]} but the GPU dropped the pair"
            else:
                assert i not in got_of, f"surface row {i} is not the best query of model row {cand_of[i]} (row {best} is)"
                n_unique_drop += 1
        else:
            assert i not in got_of
    assert len(cand_of) >= 80                                     # the sample really exercises matches (half the rows are copies)


def test_cfg4_descriptors_1m_keypoints(oracle_c):
    from bench import _ridge_cloud
    from pcreg_amd.device import DescriptorPipeline
    P = S = 1_000_000
    pts, kp = _ridge_cloud(P, S)
    opt = dict(min_pts=500, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)
    dev = torch.device("cuda", 0)
    dp = DescriptorPipeline(dev)
    tp = torch.from_numpy(np.ascontiguousarray(pts.T)).to(dev); tk = torch.from_numpy(np.ascontiguousarray(kp.T)).to(dev)
    feat, desc, V = dp.describe(tp, tk, opt)
    assert V > 0.9 * S
    sums = desc[:V].sum(dim=1)
    assert float(sums.min()) >= opt["min_pts"] and float(sums.max()) <= opt["max_pts"]     # every count row is a whole support
    assert bool((desc[:V] >= 0).all()) and bool((desc[:V] == desc[:V].round()).all())
    # ---- oracle on 200 random keypoints (two brute-force scans of the 1 M-point cloud each) -----------------------
    sel = np.sort(np.random.default_rng(4).choice(S, 200, replace=False))
    rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp[sel], opt, nthreads=CORES)
    featc = feat[:V].cpu().numpy()
    # surviving keypoints keep the input order: row v of the GPU output is the v-th survivor; find the sample's rows by
    # their coordinates (exact copies of the input keypoints)
    key = {tuple(r): v for v, r in enumerate(map(tuple, featc.tolist()))} if V < S else None
    rows = []
    for r in rfeat:
        v = key[tuple(r.tolist())] if key is not None else None
        rows.append(v)
    if key is None:
        assert len(rfeat) == len(sel)
        rows = sel.tolist()
    got = desc[torch.tensor(rows, device=dev)].cpu().numpy()
    np.testing.assert_array_equal(featc[rows], rfeat)
    np.testing.assert_array_equal(got, rdesc)                     # all 980 counts of all sampled keypoints


def test_cfg5_batch_of_64_crops_vs_1m_model(oracle_c):
    from bench import BBOX, MATCH_RATIO, MATCH_THR_ABS, RANSAC_COEF, make_crop
    from pcreg_amd.batch import BatchRegistration
    from pcreg_amd.device import soa
    M, Q, n_crops = 1_000_000, 50_000, 64
    rng = np.random.default_rng(10)
    model = rng.random((M, 3), dtype=np.float32) * BBOX.astype(np.float32)
    dev = torch.device("cuda", 0)
    ms = soa(torch.from_numpy(model).to(dev))
    crops = [make_crop(model, Q, c) for c in range(n_crops)]
    qs = [soa(torch.from_numpy(c).to(dev)) for c in crops]
    br = BatchRegistration(ms, Q, n_streams=2, device=dev)
    res = br.run(qs, MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF, seed=7)
    assert [r["crop"] for r in res] == list(range(n_crops))
    assert sum(r["failed"] for r in res) == 0
    assert min(r["n_inliers"] for r in res) > 0.5 * Q and all(r["n_inliers"] == r["maxInliers"] for r in res)
    for c in (0, 41):                                             # two crops end to end against the oracle
        ref_pairs = oracle_c.match_points_f32(crops[c], model, MATCH_THR_ABS, MATCH_RATIO, True, nthreads=CORES)
        assert res[c]["n_pairs"] == len(ref_pairs)
        rp1 = crops[c][ref_pairs[:, 0] - 1].astype(np.float64); rp2 = model[ref_pairs[:, 1] - 1].astype(np.float64)
        coef = dict(RANSAC_COEF, iterNum=RANSAC_COEF["iterNum"] if c == 0 else 1500)
        if c != 0:                                                # a second crop at a shorter hypothesis count (oracle time): rerun it alone
            r1 = BatchRegistration(ms, Q, n_streams=1, device=dev).run([qs[c]], MATCH_THR_ABS, MATCH_RATIO, coef, seed=7)[0]
        else:
            r1 = res[c]
        ref = oracle_c.ransac(rp1, rp2, coef, seed=7)
        assert not r1["failed"] and r1["numSuccess"] == ref["numSuccess"] and r1["maxInliers"] == ref["maxInliers"]
        assert r1["n_inliers"] == len(ref["inlierIdx"])
        assert np.linalg.norm(r1["T"] - ref["T"]) < 1e-5


def test_f16_split_rounding_midpoints_regression(oracle_c):
    """cfg 5's crop 0 holds six queries whose true 2nd neighbour has |m~|^2 exactly on an f16 rounding midpoint (216.9375,
    328.375, 171.4375, ...).  Round 1's prep kernel rounded the stored high part and the residual's high part differently
    there (v_fma_mixlo_f16 fold), the point's score came out one f16 ulp too large, it fell out of the candidates and the
    certificate still passed.  The whole crop must now equal the exhaustive oracle, indices and distance bits."""
    import pcreg_amd as pc
    from bench import BBOX, make_crop
    M, Q = 1_000_000, 50_000
    model = np.random.default_rng(10).random((M, 3), dtype=np.float32) * BBOX.astype(np.float32)
    for c in (0, 17, 41):                         # crop 0 holds the six known cases; two more crops, every query checked
        crop = make_crop(model, Q, c)
        gi, gd = pc.knn2_points(crop, model)
        oi, od = oracle_c.knn2_points_f32(crop, model, nthreads=CORES)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd, od)


def _midpoint_points(rng, want=400):
    """Scaled model points (x, y, z) whose fp32 value w = fma(z, z, fma(y, y, x * x)) is EXACTLY an f16 rounding
    midpoint while the exact product-sum is not: the case in which `(_Float16)fma(..)` rounded once (v_fma_mixlo_f16)
    and rounded twice (through fp32) give different f16 values."""
    f32 = np.float32
    out = []
    while len(out) < want:
        h = np.float16(rng.uniform(70, 800))
        ulp = float(np.spacing(h))
        W = float(h) + (ulp / 2 if rng.random() < 0.5 else -ulp / 2)                 # exactly representable in fp32
        xs, ys = f32(rng.integers(1, 4 * 6) / 4.0), f32(rng.integers(1, 4 * 6) / 4.0)
        s2 = float(ys) * float(ys) + float(xs) * float(xs)                           # exact, and exact in fp32
        if s2 >= W - 1:
            continue
        z0 = f32(np.sqrt(W - s2))
        for d in (0, 1, -1, 2, -2):
            z = np.nextafter(z0, f32(np.inf if d > 0 else -np.inf)) if d else z0
            if abs(d) == 2:
                z = np.nextafter(z, f32(np.inf if d > 0 else -np.inf))
            exact = float(z) * float(z) + s2                                          # float64: exact enough (48 + a few bits)
            if f32(exact) == f32(W) and exact != W:
                out.append((float(xs), float(ys), float(z)))
                break
    return np.array(out, dtype=np.float32)


def test_f16_split_midpoints_by_construction(oracle_c):
    """Model points whose sigma^2 |m - c|^2, AS COMPUTED IN fp32, is exactly an f16 rounding midpoint while the exact
    product-sum is not (see _midpoint_points): the box is pinned to [-64, 64]^3 by two corner points (centre 0,
    sigma = 1/2, both exact).  Queries sit next to such points so that they are first or second neighbours.  With round
    1's prep kernel the stored high part of |m~|^2 and the residual's high part were rounded differently there, the
    point's score came out one f16 ulp off, second neighbours went missing and the certificate still passed (verified:
    this test fails on that build); the search must equal the oracle exactly."""
    import pcreg_amd as pc
    rng = np.random.default_rng(12)
    ties = _midpoint_points(rng) * np.float32(2.0)                                     # unscaled: sigma = 1/2, exact doubling
    ties = (ties * rng.choice([-1.0, 1.0], ties.shape).astype(np.float32))[:, rng.permutation(3)]
    assert np.abs(ties).max() < 64
    filler = rng.uniform(-60, 60, (30000, 3)).astype(np.float32)
    corners = np.array([[-64, -64, -64], [64, 64, 64]], dtype=np.float32)
    model = np.vstack([corners, ties, filler]).astype(np.float32)
    model = model[rng.permutation(len(model))]
    q = np.vstack([ties + rng.normal(0, 0.3, ties.shape).astype(np.float32), rng.uniform(-60, 60, (2000, 3)).astype(np.float32)]).astype(np.float32)
    gi, gd = pc.knn2_points(q, model)
    oi, od = oracle_c.knn2_points_f32(q, model, nthreads=CORES)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd, od)


def test_sphere_sweep_at_the_reference_shape(oracle_c, oracle_py):
    """completeExperimentFast.m:46-224 at the reference's own shape (a 60 k-keypoint model, ~330 valid spheres of ~1500 descriptors,
    a 2000-keypoint surface, D = 980): the segmented launch chain against one getMatches chain per sphere -- every sphere's pairs,
    counts, trial list and transforms identical -- and against the oracle's getDescriptorMask + getMatches on spheres spread over
    the sweep (first, middle, last, the first trial sphere)."""
    from pcreg_amd.sweep import SphereSweep
    VM, VS, D = 60000, 2000, 980
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    featM = rng.uniform([0, 0, 0], [60, 50, 40], (VM, 3))
    g = torch.Generator(device=dev); g.manual_seed(1)
    descM = torch.poisson(torch.full((VM, D), 3.0, device=dev), generator=g).to(torch.float64)
    near = np.argsort(np.linalg.norm(featM - np.array([31.0, 24.0, 19.0]), axis=1))[:VS]
    c, s = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    featS = featM[near] @ R.T + np.array([2.0, -1.0, 0.5]) + rng.normal(0, 0.02, (VS, 3))
    descS = (descM[torch.from_numpy(near).to(dev)] + torch.poisson(torch.full((VS, D), 0.15, device=dev), generator=g).to(torch.float64)).contiguous()
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate", MatchThreshold=10, MaxRatio=0.99,
               Metric="SAD", Unique=True, VERBOSE=0)
    opt = dict(minPtNum=3, iterNum=2000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
    sw = SphereSweep(featM, descM, featS, descS, device=dev)
    out = sw.run(par, opt, **kw)
    ref = sw.run_streams(par, opt, n_streams=8, **kw)
    S = len(out["centres"])
    assert S >= 300 and len(out["trial"]) >= 50
    for k in ("num_desc", "num_putative", "trial", "statsPutative", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(out[k], ref[k])
    for a, b in zip(out["matches"], ref["matches"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(out["transforms"], ref["transforms"]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    dm, ds = descM.cpu().numpy(), descS.cpu().numpy()
    for i in sorted({0, S // 2, S - 1, int(out["trial"][0])}):
        idx = np.nonzero(oracle_py.getDescriptorMask(featM, out["centres"][i], kw["R_desc"], 0.0))[0]
        np.testing.assert_array_equal(idx, out["model_rows"][i])
        np.testing.assert_array_equal(oracle_c.getMatches(ds, dm[idx], par, nthreads=0), out["matches"][i])


def test_chain_cfg4_to_cfg2_on_rows_the_descriptor_kernel_produces(oracle_c, oracle_py):
    """getMatches at 20 k x 200 k and the sphere sweep at the reference's shape, on rows that the repo's own
    descriptor kernel wrote for a synthetic scene (a model cloud and a moved, noisy crop of it: sparse, spatially correlated
    histograms of overlapping supports -- not i.i.d. Poisson rows).  Oracle: 120 random surface rows against ALL model rows,
    incl. the Unique rule over all surface rows; three spheres of the sweep through the oracle's getMatches."""
    from bench import MATCH_PAR, desc_chain_data
    from pcreg_amd.sweep import SphereSweep
    dev = torch.device("cuda", 0)
    d = desc_chain_data(dev, 200_000, 20_000)
    dp, VM, VS = d["dp"], d["VM"], d["VS"]
    assert VM > 100_000 and VS > 10_000
    pairs_t, n_pairs = dp.match(d["descS"], VS, d["descM"], VM, MATCH_PAR)
    P = int(n_pairs.item())
    pairs = pairs_t[:P].cpu().numpy().astype(np.int64)
    assert P > VS // 10 and np.all(np.diff(pairs[:, 0]) > 0) and len(np.unique(pairs[:, 1])) == P
    iS = d["descS"].index[:VS].cpu().numpy(); iM = d["descM"].index[:VM].cpu().numpy()
    right = np.sum(d["near"][iS[pairs[:, 0] - 1]] == iM[pairs[:, 1] - 1])
    assert right > 0.5 * P                                        # the matches are the crop's own keypoints, mostly
    dS = d["descS"].compact(VS).cpu().numpy().astype(np.float64); dM = d["descM"].compact(VM).cpu().numpy().astype(np.float64)
    pS, pM = oracle_py.preprocess_descriptors(dS, dM, MATCH_PAR)
    del dS, dM
    nS, nM = oracle_py._normalize_rows(pS), oracle_py._normalize_rows(pM)
    del pS, pM
    rng = np.random.default_rng(12)
    sel = np.sort(rng.choice(VS, 120, replace=False))
    par_nu = dict(Metric="SAD", MatchThreshold=10, MaxRatio=0.99, Unique=False, Prenormalized=True)
    cand, _ = oracle_c.matchFeatures(nS[sel], nM, par_nu, nthreads=CORES)
    cand_of = {int(sel[i - 1]): int(j) - 1 for i, j in cand}
    sel_set = set(sel.tolist())
    got_of = {int(i) - 1: int(j) - 1 for i, j in pairs if (int(i) - 1) in sel_set}
    for i in sel.tolist():
        if i in got_of:
            assert cand_of.get(i) == got_of[i], f"surface row {i}: GPU matched {got_of[i]}, oracle forward match {cand_of.get(i)}"
        if i in cand_of:
            col = np.abs(nS - nM[cand_of[i]]).sum(axis=1)
            best = int(np.argmin(col))
            assert (i in got_of) == (best == i), f"surface row {i}: Unique rule (best query of model row {cand_of[i]} is {best})"
        else:
            assert i not in got_of
    assert len(cand_of) >= 30
    del nS, nM, d, dp
    torch.cuda.empty_cache()
    # ---- the sweep on described rows
    d = desc_chain_data(dev, 60_000, 2_000, compact=False)
    VM, VS = d["VM"], d["VS"]
    sw = SphereSweep(d["featM"][:VM], d["descM"][:VM], d["featS"][:VS], d["descS"][:VS], device=dev)
    opt = dict(minPtNum=3, iterNum=2000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    out = sw.run(MATCH_PAR, opt, R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
    S = len(out["centres"])
    assert S > 20
    fm, dm, ds = d["featM"][:VM].cpu().numpy(), d["descM"][:VM].cpu().numpy(), d["descS"][:VS].cpu().numpy()
    best = int(np.argmax(out["num_putative"]))
    for i in sorted({0, S // 2, best}):
        idx = np.nonzero(oracle_py.getDescriptorMask(fm, out["centres"][i], 9.0, 0.0))[0]
        np.testing.assert_array_equal(idx, out["model_rows"][i])
        np.testing.assert_array_equal(oracle_c.getMatches(ds, dm[idx], MATCH_PAR, nthreads=CORES), out["matches"][i])
    assert len(out["trial"]) >= 1 and any(t is not None for t in out["transforms"])
