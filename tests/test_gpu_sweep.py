"""The resident sphere-sweep driver (pcreg_amd/sweep.py) against the oracle's CPU restatement of
completeExperimentFast.m:46-224, and the final-stage pieces (quickTF, invertTF, distance refine)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
           MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)
OPT = dict(minPtNum=3, iterNum=1500, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)


def _scene(seed=0, VM=6000, VS=260, D=96):
    """Synthetic keypoints + count-like descriptors: the surface is a rigidly moved patch of the model,
    its descriptors are noisy copies of the matching model rows."""
    import oracle.pcreg_oracle as o
    rng = np.random.default_rng(seed)
    featM = rng.uniform([0, 0, 0], [40, 30, 20], (VM, 3))
    descM = rng.poisson(3.0, (VM, D)).astype(np.float64)
    centre = np.array([22.0, 14.0, 9.0])
    near = np.argsort(np.linalg.norm(featM - centre, axis=1))[:VS]
    R = o.eul2rotm(np.array([0.3, -0.2, 0.1])); t = np.array([2.0, -1.0, 0.5])
    featS = featM[near] @ R.T + t + rng.normal(0, 0.02, (VS, 3))
    descS = descM[near] + rng.poisson(0.15, (VS, D))
    return featM, descM, featS, descS


def test_sphere_sweep_equals_oracle_driver(oracle_c, oracle_py):
    from pcreg_amd.sweep import SphereSweep
    featM, descM, featS, descS = _scene()
    kw = dict(R_desc=9.0, d_spheres=6.0, min_pts=500, putative_thresh=60, seed=3)
    ref = oracle_py.sphere_sweep(featM, descM, featS, descS, PAR, OPT, get_matches=oracle_c.getMatches,
                                 run_ransac=oracle_c.ransac, **kw)
    assert len(ref["centres"]) >= 4 and len(ref["trial"]) >= 1          # the scene exercises both stages
    sw = SphereSweep(featM, descM, featS, descS)
    got = sw.run(PAR, OPT, **kw)                                            # batched: two host syncs for the whole sweep
    ser = sw.run_serial(PAR, OPT, **kw)                                     # one sphere at a time
    strm = sw.run_streams(PAR, OPT, n_streams=3, **kw)                       # one getMatches chain per sphere, on streams
    for other in (ser, strm):
        for k in ("num_desc", "num_putative", "trial", "statsPutative", "statsSuccess", "statsInliers"):
            np.testing.assert_array_equal(got[k], other[k])
        for a, b in zip(got["matches"], other["matches"]):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(got["transforms"], other["transforms"]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    np.testing.assert_array_equal(got["centres"], ref["centres"])
    np.testing.assert_array_equal(got["num_desc"], ref["num_desc"])
    np.testing.assert_array_equal(got["num_putative"], ref["num_putative"])
    for a, b in zip(got["model_rows"], ref["model_rows"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(got["matches"], ref["matches"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(got["trial"], ref["trial"])
    for k in ("statsPutative", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(got[k], ref[k])
    np.testing.assert_allclose(got["statsRatio"], ref["statsRatio"], rtol=0, atol=1e-12)
    for a, b in zip(got["transforms"], ref["transforms"]):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.linalg.norm(a - b) < 1e-9


def test_sphere_model_handle_edges(oracle_py):
    """The sphere-model handle at its edges: no spheres at all (an empty sweep, like sphereSweep's), counts that are not the spheres'
    (refused with the sphere's number), a handle used after close() (refused, not a crash), a surface of another descriptor length."""
    import pcreg_amd as pc
    from pcreg_amd._lib import PcregError
    featM, descM, featS, descS = _scene()
    R = 9.0
    centres = oracle_py.pcUniformSamples(featM, 6.0)
    counts = pc.sphereCounts(featM, centres, R)
    keep = counts >= 500
    with pc.DescSet(descS) as hS, pc.DescSet(descM) as hM:
        with pc.SphereModel(hM, featM, np.zeros((0, 3)), np.zeros(0, dtype=np.int32), R) as sm0:
            out = pc.sphereSweepOnModel(sm0, hS, featS, PAR, 60, OPT, seed=1)
            assert len(out["trial"]) == 0 and len(out["matches"]) == 0 and len(out["num_putative"]) == 0
        wrong = counts[keep].copy(); wrong[1] += 1
        with pytest.raises(PcregError, match="holds"):
            pc.SphereModel(hM, featM, centres[keep], wrong, R)
        sm = pc.SphereModel(hM, featM, centres[keep], counts[keep], R)
        ref = pc.sphereSweepOnModel(sm, hS, featS, PAR, 60, OPT, seed=1)
        assert len(ref["trial"]) >= 1
        with pc.DescSet(descS[:, :-1].copy()) as hShort:
            with pytest.raises(PcregError):
                pc.sphereSweepOnModel(sm, hShort, featS, PAR, 60, OPT, seed=1)
        again = pc.sphereSweepOnModel(sm, hS, featS, PAR, 60, OPT, seed=1)                   # a refused call leaves the handle usable
        assert np.array_equal(again["trial"], ref["trial"]) and all(np.array_equal(a, b) for a, b in zip(again["matches"], ref["matches"]))
        sm.close(); sm.close()
        with pytest.raises(PcregError):
            pc.sphereSweepOnModel(sm, hS, featS, PAR, 60, OPT, seed=1)


def test_host_tier_sphere_sweep_equals_the_device_driver(oracle_py):
    """pcreg_sphere_counts + pcreg_sphere_sweep (what MATLAB reaches through pcreg_mex: host arrays, descriptor sets resident) ==
    SphereSweep.run field for field -- counts, row lists, matches, trial spheres, transforms (the same kernels, the same seeds)."""
    import pcreg_amd as pc
    from pcreg_amd.sweep import SphereSweep
    featM, descM, featS, descS = _scene()
    kw = dict(R_desc=9.0, d_spheres=6.0, min_pts=500, putative_thresh=60, seed=3)
    sw = SphereSweep(featM, descM, featS, descS)
    want = sw.run(PAR, OPT, **kw)
    assert len(want["centres"]) >= 4 and len(want["trial"]) >= 1
    centres = oracle_py.pcUniformSamples(featM, kw["d_spheres"])
    counts = pc.sphereCounts(featM, centres, kw["R_desc"])
    np.testing.assert_array_equal(counts, [oracle_py.getDescriptorMask(featM, c, kw["R_desc"]).sum() for c in centres])
    keep = counts >= kw["min_pts"]
    with pc.DescSet(descS) as hS, pc.DescSet(descM) as hM:
        got = pc.sphereSweep(hS, hM, featS, featM, centres[keep], counts[keep], kw["R_desc"], PAR, kw["putative_thresh"], OPT, seed=kw["seed"])
        again = pc.sphereSweep(hS, hM, featS, featM, centres[keep], counts[keep], kw["R_desc"], PAR, kw["putative_thresh"], OPT, seed=kw["seed"])
        with pytest.raises(Exception):                                       # counts that are not the spheres' counts: refused, not a fault
            pc.sphereSweep(hS, hM, featS, featM, centres[keep], counts[keep] - 1, kw["R_desc"], PAR, kw["putative_thresh"], OPT)
        empty = pc.sphereSweep(hS, hM, featS, featM, centres[:0], counts[:0], kw["R_desc"], PAR, kw["putative_thresh"], OPT)
        # one model, many surfaces: the model side in a handle (pcreg_sphere_model_create), the model SET destroyed before the sweeps
        sm = pc.SphereModel(hM, featM, centres[keep], counts[keep], kw["R_desc"])
    with pc.DescSet(descS) as hS2:
        on_model = pc.sphereSweepOnModel(sm, hS2, featS, PAR, kw["putative_thresh"], OPT, seed=kw["seed"])
        on_model2 = pc.sphereSweepOnModel(sm, hS2, featS, dict(PAR, metric_factor=0.8), kw["putative_thresh"], OPT, seed=kw["seed"])
        on_model3 = pc.sphereSweepOnModel(sm, hS2, featS, PAR, kw["putative_thresh"], OPT, seed=kw["seed"])
    with pc.DescSet(descS) as hS3, pc.DescSet(descM) as hM3:
        other_par = pc.sphereSweep(hS3, hM3, featS, featM, centres[keep], counts[keep], kw["R_desc"], dict(PAR, metric_factor=0.8), kw["putative_thresh"], OPT, seed=kw["seed"])
    sm.close()
    for k in ("num_putative", "trial", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(on_model2[k], other_par[k], err_msg=k)
    assert len(empty["trial"]) == 0 and empty["matches"] == []
    for g in (got, again, on_model, on_model3):
        for k in ("centres", "num_desc", "num_putative", "trial", "statsPutative", "statsSuccess", "statsInliers", "statsRatio"):
            np.testing.assert_array_equal(g[k], want[k], err_msg=k)
        for k in ("matches", "model_rows"):
            assert len(g[k]) == len(want[k])
            for a, b in zip(g[k], want[k]):
                np.testing.assert_array_equal(a, b)
        for a, b in zip(g["transforms"], want["transforms"]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))


def test_sphere_counts_select_and_gather(oracle_py):
    import torch
    from pcreg_amd.sweep import SphereSweep
    featM, descM, featS, descS = _scene(seed=4, VM=3000, VS=40, D=24)
    sw = SphereSweep(featM, descM, featS, descS)
    centres = sw.sphere_centres(7.0)
    np.testing.assert_array_equal(centres, oracle_py.pcUniformSamples(featM, 7.0))
    valid, n = sw.valid_spheres(centres, 8.0, min_pts=100)
    ref_n = np.array([oracle_py.getDescriptorMask(featM, c, 8.0).sum() for c in centres])
    np.testing.assert_array_equal(n, ref_n)
    np.testing.assert_array_equal(valid, ref_n >= 100)
    # a centre ON a keypoint and a radius equal to a distance: strict '<' at the boundary
    c = featM[17]
    R = float(np.sort(np.linalg.norm(featM - c, axis=1))[50])
    m = sw.match_sphere(c, R, PAR)
    mask = oracle_py.getDescriptorMask(featM, c, R)
    np.testing.assert_array_equal(m["rows"].cpu().numpy(), np.nonzero(mask)[0])
    np.testing.assert_array_equal(m["featCur"][:m["num_desc"]].cpu().numpy(), featM[mask])
    # an empty sphere
    e = sw.match_sphere([1e6, 0, 0], 1.0, PAR)
    assert e["num_desc"] == 0 and e["num_putative"] == 0


def test_sphere_masks_decide_like_the_square_root_at_the_boundary(oracle_py):
    """The sphere kernels compare the square root's ARGUMENT with the smallest double whose rounded root reaches R
    (sphere_sqrt_threshold) instead of taking an fp64 square root per (sphere, keypoint).  Keypoints parked within a few ulps of
    the sphere's surface -- for radii whose square is not a double, and whose square root is not either -- must get the
    decision of `vecnorm(feat - c, 2, 2) < R` (getDescriptorMask, completeExperimentFast.m:435-439), counts and row lists."""
    import torch
    from pcreg_amd.sweep import SphereSweep
    rng = np.random.default_rng(8)
    for R in (9.0, 9.1, 0.3, float(np.sqrt(7.0)), 1e-3, 123.456):
        c = rng.uniform(-5, 5, 3)
        u = rng.standard_normal((4000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        k = rng.integers(-4, 5, (4000, 1))
        feat = c + u * (R * (1.0 + k * 2.220446049250313e-16))                  # distances within a few ulps of R, either side
        feat = np.vstack([feat, c + u[:500] * R * rng.uniform(0.2, 1.8, (500, 1))])
        d = feat - c
        want = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) < R
        assert 100 < want[:4000].sum() < 3900                                       # the boundary really splits the parked points
        np.testing.assert_array_equal(want, oracle_py.getDescriptorMask(feat, c, R))
        sw = SphereSweep(feat, np.zeros((feat.shape[0], 4)), feat[:8], np.zeros((8, 4)))
        _, n = sw.valid_spheres(np.array([c]), R, min_pts=1)
        assert int(n[0]) == int(want.sum()), R
        m = sw.match_sphere(c, R, PAR)
        np.testing.assert_array_equal(m["rows"].cpu().numpy(), np.nonzero(want)[0])


def test_quicktf_inverttf_and_distance_refine(oracle_py):
    import torch
    import pcreg_amd as pc
    from pcreg_amd.device import soa
    from pcreg_amd.sweep import quickTF_dev, refine_by_distance_dev
    rng = np.random.default_rng(1)
    dev = torch.device("cuda", 0)
    pts = rng.uniform(-20, 20, (5000, 3))
    T = np.eye(4); T[:3, :3] = oracle_py.eul2rotm(np.array([0.4, 0.1, -0.3])).T; T[3, :3] = [1.0, -2.0, 3.0]
    p_dev = soa(torch.from_numpy(pts).to(dev))
    moved = quickTF_dev(p_dev, T)
    np.testing.assert_allclose(moved.t().cpu().numpy(), oracle_py.quickTF(pts, T), rtol=0, atol=1e-12)
    back = quickTF_dev(moved, pc.invertTF(T))                                   # quickTF(quickTF(p,T), invertTF(T)) == p
    np.testing.assert_allclose(back.t().cpu().numpy(), pts, rtol=0, atol=1e-11)
    # distance refine: pts1 = pts2 moved by a small transform + noise, 30 % gross outliers
    T2 = np.eye(4); T2[:3, :3] = oracle_py.eul2rotm(np.array([0.01, -0.02, 0.015])).T; T2[3, :3] = [0.05, -0.02, 0.03]
    p2 = pts[:1200]
    p1 = oracle_py.quickTF(p2, T2) + rng.normal(0, 0.01, p2.shape)
    p1[:360] += rng.uniform(3, 6, (360, 3))
    Tref, inl = oracle_py.refine_by_distance(p1, p2, 1.5)
    Tg, cnt = refine_by_distance_dev(soa(torch.from_numpy(p1).to(dev)), soa(torch.from_numpy(p2).to(dev)), len(p1), 1.5)
    assert cnt == len(inl) == 840
    assert np.linalg.norm(Tg - Tref) < 1e-10 and np.linalg.norm(Tg - T2) < 0.01
    # exactly three inliers -> estimateTransform's N == 3 branch; fewer -> []
    q1 = p1.copy(); q1[3:] += 100.0
    Tref3, inl3 = oracle_py.refine_by_distance(q1, p2, 1.5)
    assert len(inl3) <= 3
    Tg3, cnt3 = refine_by_distance_dev(soa(torch.from_numpy(q1).to(dev)), soa(torch.from_numpy(p2).to(dev)), len(q1), 1.5)
    assert cnt3 == len(inl3)
    assert (Tg3 is None) == (Tref3 is None)
    if Tg3 is not None:
        assert np.linalg.norm(Tg3 - Tref3) < 1e-9


def test_batched_sweep_counts_its_host_syncs(monkeypatch):
    """The batched driver's contract: no device-to-host read between the first sphere's select and the batched
    ransac launch.  torch.Tensor.cpu / .item are the only ways sweep.py reads the device; they are counted per phase."""
    import torch
    from pcreg_amd import sweep as sw_mod
    featM, descM, featS, descS = _scene(seed=2, VM=5000, VS=200, D=64)
    sw = sw_mod.SphereSweep(featM, descM, featS, descS)
    calls = []
    real_plan = sw_mod.lib().pcreg_dev_sweep_plan
    orig_cpu, orig_item = torch.Tensor.cpu, torch.Tensor.item
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: (calls.append("cpu"), orig_cpu(self, *a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "item", lambda self, *a, **k: (calls.append("item"), orig_item(self, *a, **k))[1])
    out = sw.run(PAR, OPT, R_desc=9.0, d_spheres=6.0, min_pts=400, putative_thresh=50, seed=1)
    S = len(out["centres"])
    assert S >= 4
    # sync 1 = featM (once, cached) + the counts + the spheres' row lists; sync 2 = the final block of reads; nothing scales with S
    assert len(calls) <= 10, calls
    # the spheres belong to the model: a second sweep (another surface would do) reads the device ONCE, through its pinned
    # transfers -- no Tensor.cpu / .item at all -- and returns the same results
    calls.clear()
    again = sw.run(PAR, OPT, R_desc=9.0, d_spheres=6.0, min_pts=400, putative_thresh=50, seed=1)
    assert calls == [], calls
    for k in ("num_desc", "num_putative", "trial", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(out[k], again[k])
    for a, b in zip(out["matches"] + out["model_rows"], again["matches"] + again["model_rows"]):
        np.testing.assert_array_equal(a, b)
    other = sw.run(PAR, OPT, R_desc=8.0, d_spheres=6.0, min_pts=300, putative_thresh=50, seed=1)          # other spheres: made anew
    assert len(other["centres"]) != S or not np.array_equal(other["num_desc"], out["num_desc"])
    # another surface against the same model: the model's spheres and powered rows are reused, the results are those of a fresh object
    featM2, descM2, featS2, descS2 = _scene(seed=9, VM=5000, VS=170, D=64)
    sw.set_surface(featS2, descS2)
    calls.clear()
    got2 = sw.run(PAR, OPT, R_desc=9.0, d_spheres=6.0, min_pts=400, putative_thresh=50, seed=1)
    assert calls == [], calls
    fresh = sw_mod.SphereSweep(featM, descM, featS2, descS2).run(PAR, OPT, R_desc=9.0, d_spheres=6.0, min_pts=400, putative_thresh=50, seed=1)
    for k in ("num_desc", "num_putative", "trial", "statsSuccess", "statsInliers"):
        np.testing.assert_array_equal(got2[k], fresh[k])
    for a, b in zip(got2["matches"], fresh["matches"]):
        np.testing.assert_array_equal(a, b)
    # the kept spheres hold the restricted model set: only the two most recent parameter sets stay, and a dropped one is made anew
    sw.run(PAR, OPT, R_desc=8.5, d_spheres=6.0, min_pts=300, putative_thresh=50, seed=1)
    assert len(sw._spheres) == 2 and (9.0, 6.0, 400) not in sw._spheres
    back = sw.run(PAR, OPT, R_desc=9.0, d_spheres=6.0, min_pts=400, putative_thresh=50, seed=1)
    for a, b in zip(back["matches"], fresh["matches"]):
        np.testing.assert_array_equal(a, b)


def _segments_direct(descS, descM, rows_list, par, metric=False):
    """pcreg_dev_get_matches_segmented on explicit row lists -> per-segment (pairs, metric)."""
    import ctypes as C
    import torch
    from pcreg_amd._lib import check, lib
    from pcreg_amd.api import _match_opts
    dev = torch.device("cuda", 0)
    L = lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    Q, D = descS.shape
    VM = descM.shape[0]
    S = len(rows_list)
    off = np.zeros(S + 1, dtype=np.int32); off[1:] = np.cumsum([len(r) for r in rows_list])
    tot, n_max = int(off[-1]), max([len(r) for r in rows_list] + [0])
    dS = torch.from_numpy(np.ascontiguousarray(descS, dtype=np.float64)).to(dev)
    dM = torch.from_numpy(np.ascontiguousarray(descM, dtype=np.float64)).to(dev)
    rows = torch.from_numpy(np.concatenate([np.asarray(r, dtype=np.int32) for r in rows_list] + [np.zeros(0, np.int32)])).to(dev)
    if rows.numel() == 0:
        rows = torch.zeros(1, dtype=torch.int32, device=dev)
    seg_off = torch.from_numpy(off).to(dev)
    pairs = torch.zeros((S, Q, 2), dtype=torch.int32, device=dev)
    met = torch.zeros((S, Q), dtype=torch.float64, device=dev)
    n_pairs = torch.full((S,), -1, dtype=torch.int32, device=dev)
    o = _match_opts(par)
    wsb = L.pcreg_dev_get_matches_segmented_workspace(Q, VM, D, S, tot, n_max)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    check(L.pcreg_dev_get_matches_segmented(p(dS), Q, p(dM), VM, D, p(rows), p(seg_off), S, tot, n_max, C.byref(o), p(pairs),
                                            p(met) if metric else None, p(n_pairs), p(ws), C.c_size_t(ws.numel()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    n = n_pairs.cpu().numpy()
    ph, mh = pairs.cpu().numpy().astype(np.uint32), met.cpu().numpy()
    return [(ph[z, :n[z]], mh[z, :n[z]]) for z in range(S)]


@pytest.mark.parametrize("par_over", [dict(), dict(Unique=False), dict(UNNORMALIZE=False), dict(CHANGE_METRIC=False),
                                      dict(MatchThreshold=2.0, MaxRatio=0.8)])
def test_segmented_get_matches_equals_one_call_per_segment(oracle_c, par_over, debug_set):
    """Ragged segments (long, short, one row, empty, overlapping rows, the whole model) against pcreg.getMatches per segment and the
    oracle; with the certificate forced to fail every query takes the refinement (and, at level 2, the exhaustive exact kernel) and the
    answer must not change."""
    import os
    import pcreg_amd as pc
    rng = np.random.default_rng(11)
    VM, Q, D = 1500, 300, 120
    descM = rng.poisson(3.0, (VM, D)).astype(np.float64)
    pick = rng.choice(VM, Q, replace=False)
    descS = descM[pick] + rng.poisson(0.2, (Q, D))
    descS[5] = descS[4]                                     # duplicate surface rows: Unique ties
    descM[7] = 0.0                                          # an all-zero model row
    rows_list = [np.sort(rng.choice(VM, 700, replace=False)), np.sort(rng.choice(VM, 130, replace=False)), np.array([7]), np.zeros(0, np.int64),
                 np.arange(VM), np.sort(rng.choice(VM, 2, replace=False)), np.sort(pick[:200])]
    par = dict(PAR, **par_over)
    got = _segments_direct(descS, descM, rows_list, par, metric=True)
    for z, r in enumerate(rows_list):
        if len(r) == 0:
            assert got[z][0].shape[0] == 0
            continue
        want = pc.getMatches(descS, descM[r], par)
        np.testing.assert_array_equal(got[z][0], want, err_msg=f"segment {z}")
        np.testing.assert_array_equal(want, oracle_c.getMatches(descS, descM[r], par))
    debug_set("seg_batched")          # the bounded-workspace form: batches of consecutive segments on gathered sub-models (several
    batched = _segments_direct(descS, descM, rows_list, par, metric=True)      # batches here: the workspace is the one-chain size)
    debug_set("seg_batched", 0)
    for (a, ma), (b, mb) in zip(got, batched):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ma, mb)
    debug_set("seg_wave_finalize")    # the forward re-rank as one wave per query (the form of rounds 2-3) instead of pick / pairs / decide
    waves = _segments_direct(descS, descM, rows_list, par, metric=True)
    debug_set("seg_wave_finalize", 0)
    for (a, ma), (b, mb) in zip(got, waves):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ma, mb)
    for level in (1, 2):              # 1: every query through the score-filtered refinement; 2: and on to the exhaustive exact kernel
        debug_set("match_force_fallback", level)
        forced = _segments_direct(descS, descM, rows_list, par, metric=True)
        for (a, ma), (b, mb) in zip(got, forced):
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(ma, mb)                # the same exact fp64 distances either way


def _oracle_get_matches_with_metric(descS, descM, par):
    """orc_get_matches' pairs AND its matchMetric (the python wrapper of the C oracle drops the second)."""
    import ctypes as C
    from oracle import c_oracle as co
    a, b = co._f(descS), co._f(descM)
    Q, D = a.shape
    M = b.shape[0]
    o = co._mopts(par)
    pairs = np.zeros((max(Q, 1), 2), dtype=np.uint32)
    met = np.zeros(max(Q, 1))
    P = co.lib().orc_get_matches(co._p(a), C.c_int(Q), C.c_int(Q), co._p(b), C.c_int(M), C.c_int(M), C.c_int(D), C.byref(o),
                                 co._p(pairs, C.c_uint32), co._p(met), C.c_int(0))
    return pairs[:P].copy(), met[:P].copy()


@pytest.mark.parametrize("scale", [1.0, 3.7e-6, 3.7e88, 2.9e125, "counts", "counts_pow"])
@pytest.mark.parametrize("par_over", [dict(), dict(UNNORMALIZE=False, Unique=False)])
def test_segmented_metric_is_the_oracles_division_bit_for_bit(oracle_c, scale, par_over, debug_set):
    """The segmented chain's exact re-rank divides by the row norm through the norm's reciprocal and one correction step
    (seg_div): the matchMetric it returns must be the oracle's correctly rounded quotients summed in the oracle's order, bit for
    bit -- on dense mantissas (real-valued rows, a scale that is not a power of two), inside the range in which that division is
    proven (|x|, norm in [2^-400, 2^400]) and outside it (2.9e125, and the rows given a 1e-130 or a denormal entry: flagged, they take
    the plain division).
    Where bits cannot be promised the metric is held to 1e-12 relative + 1e-14 (a sum of D differences of values near 1; the pairs are the oracle's everywhere):
    * the .^0.6 step ("counts_pow"): the device's pow and the host libm's are two different sub-ulp functions, neither correctly
      rounded (glibc's own result depends on whether the host has FMA), so powered values may differ in the last bit;
    * UNNORMALIZE on real-valued rows: the appended constant is a mean of row sums, added sequentially by the oracle and as a tree
      by the device -- exact, and so identical, on the reference's data (integer counts: "counts"), an ulp apart on real values."""
    rng = np.random.default_rng(5)
    VM, Q, D = 900, 200, 75
    pick = rng.choice(VM, Q, replace=False)
    counts = isinstance(scale, str)
    if counts:
        descM = rng.poisson(3.0, (VM, D)).astype(np.float64)
        descS = descM[pick] + rng.poisson(0.2, (Q, D))
        par = dict(PAR, MatchThreshold=100.0, MaxRatio=0.95, CHANGE_METRIC=scale == "counts_pow", **par_over)
    else:
        descM = rng.random((VM, D)) * rng.integers(1, 40, (VM, 1))
        descS = descM[pick] * (1.0 + 0.05 * rng.standard_normal((Q, D)))
        descM[3] = 0.0
        descM, descS = descM * scale, np.abs(descS) * scale
        descM[10:40, 7] = 1.3e-130; descM[40:60, 9] = 4.1e-310; descS[0:20, 3] = 2.7e-131; descS[20:30, 5] = 1e-312
        par = dict(PAR, MatchThreshold=100.0, MaxRatio=0.95, CHANGE_METRIC=False, **par_over)
    exact = scale != "counts_pow" and (counts or not par["UNNORMALIZE"])
    rows_list = [np.sort(rng.choice(VM, 500, replace=False)), np.arange(VM), np.sort(rng.choice(VM, 40, replace=False))]
    for level in (0, 2):
        debug_set("match_force_fallback", level)
        got = _segments_direct(descS, descM, rows_list, par, metric=True)
        for z, r in enumerate(rows_list):
            pairs, met = _oracle_get_matches_with_metric(descS, descM[r], par)
            assert pairs.shape[0] > 5
            np.testing.assert_array_equal(got[z][0], pairs, err_msg=f"segment {z}, level {level}")
            if exact:
                bad = np.flatnonzero(got[z][1] != met)
                assert bad.size == 0, f"segment {z}, level {level}: {bad.size} of {met.size} metrics differ, e.g. {got[z][1][bad[:3]]} vs {met[bad[:3]]}"
            else:
                np.testing.assert_allclose(got[z][1], met, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("copies", [(3, 2), (20, 12)])
def test_segmented_get_matches_with_many_tied_candidates(oracle_c, copies, debug_set):
    """Model rows in identical copies and surface rows in identical copies: every query has `copies[0]` candidates at the same
    reference score and the same exact distance (with 20, more than the two groups of the dense re-rank hold: the remaining ones
    are summed by the wave path of segp_decide_more_kernel), every back-check item `copies[1]` surface rows tied with its own
    (with 12, more than two groups: summed inside segp_back_pick_kernel).  Ties go by index, as in the oracle."""
    import pcreg_amd as pc
    cm, cs = copies
    rng = np.random.default_rng(33)
    D = 90
    base = rng.poisson(3.0, (40, D)).astype(np.float64)
    descM = np.repeat(base, cm, axis=0)[rng.permutation(40 * cm)]
    descS = np.repeat(base[:25] + rng.poisson(0.1, (25, D)), cs, axis=0)[rng.permutation(25 * cs)]
    VM = descM.shape[0]
    par = dict(PAR, MatchThreshold=50.0, MaxRatio=1.0)
    rows_list = [np.arange(VM), np.sort(rng.choice(VM, VM // 2, replace=False)), np.sort(rng.choice(VM, 5, replace=False))]
    got = _segments_direct(descS, descM, rows_list, par, metric=True)
    debug_set("seg_wave_finalize")
    waves = _segments_direct(descS, descM, rows_list, par, metric=True)
    debug_set("seg_wave_finalize", 0)
    for z, r in enumerate(rows_list):
        pairs, met = _oracle_get_matches_with_metric(descS, descM[r], par)
        np.testing.assert_array_equal(got[z][0], pairs, err_msg=f"segment {z}")
        np.testing.assert_array_equal(got[z][0], waves[z][0])
        np.testing.assert_array_equal(got[z][1], waves[z][1])
        np.testing.assert_allclose(got[z][1], met, rtol=1e-12, atol=1e-14)
    assert sum(g[0].shape[0] for g in got) > 0


def test_segmented_get_matches_on_a_prepared_model():
    """pcreg_dev_segmented_model_prepare + pcreg_dev_get_matches_segmented_prepared == pcreg_dev_get_matches_segmented, also when the
    call's options are not the ones the model was prepared with (the call then makes its own powered rows)."""
    import ctypes as C
    import torch
    from pcreg_amd._lib import check, lib
    from pcreg_amd.api import _match_opts
    rng = np.random.default_rng(17)
    VM, Q, D = 1100, 240, 96
    descM = rng.poisson(3.0, (VM, D)).astype(np.float64)
    descS = descM[rng.choice(VM, Q, replace=False)] + rng.poisson(0.2, (Q, D))
    rows_list = [np.sort(rng.choice(VM, 500, replace=False)), np.arange(VM), np.sort(rng.choice(VM, 33, replace=False))]
    dev = torch.device("cuda", 0)
    L = lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    dS, dM = torch.from_numpy(descS).to(dev), torch.from_numpy(descM).to(dev)
    S = len(rows_list)
    off = np.zeros(S + 1, dtype=np.int32); off[1:] = np.cumsum([len(r) for r in rows_list])
    tot, n_max = int(off[-1]), max(len(r) for r in rows_list)
    rows = torch.from_numpy(np.concatenate(rows_list).astype(np.int32)).to(dev); seg_off = torch.from_numpy(off).to(dev)
    prep_par = dict(PAR)
    o_prep = _match_opts(prep_par)
    nb = L.pcreg_dev_segmented_model_bytes(VM, D)
    prep = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(L.pcreg_dev_segmented_model_prepare(p(dM), VM, D, C.byref(o_prep), p(prep), C.c_size_t(nb), st))
    ws = torch.empty(max(L.pcreg_dev_get_matches_segmented_workspace(Q, VM, D, S, tot, n_max), 256), dtype=torch.uint8, device=dev)
    for par in (prep_par, dict(PAR, metric_factor=0.8), dict(PAR, CHANGE_METRIC=False)):
        want = _segments_direct(descS, descM, rows_list, par, metric=True)
        o = _match_opts(par)
        pairs = torch.zeros((S, Q, 2), dtype=torch.int32, device=dev); met = torch.zeros((S, Q), dtype=torch.float64, device=dev)
        n_pairs = torch.full((S,), -1, dtype=torch.int32, device=dev)
        check(L.pcreg_dev_get_matches_segmented_prepared(p(dS), Q, p(dM), VM, D, p(prep), int(o_prep.change_metric), C.c_double(o_prep.metric_factor), p(rows),
                                                         p(seg_off), S, tot, n_max, C.byref(o), p(pairs), p(met), p(n_pairs), p(ws), C.c_size_t(ws.numel()), st))
        n = n_pairs.cpu().numpy(); ph = pairs.cpu().numpy().astype(np.uint32); mh = met.cpu().numpy()
        for z in range(S):
            np.testing.assert_array_equal(ph[z, :n[z]], want[z][0])
            np.testing.assert_array_equal(mh[z, :n[z]], want[z][1])


def test_segmented_get_matches_refuses_ssd():
    from pcreg_amd._lib import PcregError
    rng = np.random.default_rng(0)
    with pytest.raises(PcregError):
        _segments_direct(rng.random((8, 6)), rng.random((20, 6)), [np.arange(10)], dict(PAR, Metric="SSD"))


@pytest.mark.parametrize("case", ["long_surface", "prenormalized", "tiny_d"])
def test_segmented_get_matches_other_shapes(oracle_c, case):
    """long_surface: more than 2048 surface rows, so a chunk of the candidate lists spans several 64-row tiles (the back
    selection's per-lane lists and the wave merge see more than one entry per lane); prenormalized: matchFeatures' flag (no row
    normalisation, the appended constant still differs per segment); tiny_d: fewer features than one quantisation slab."""
    import pcreg_amd as pc
    rng = np.random.default_rng(21)
    if case == "long_surface":
        VM, Q, D = 2600, 2300, 24
    elif case == "tiny_d":
        VM, Q, D = 900, 150, 5
    else:
        VM, Q, D = 1200, 260, 64
    descM = rng.poisson(3.0, (VM, D)).astype(np.float64)
    pick = rng.choice(VM, Q, replace=Q > VM)
    descS = descM[pick] + rng.poisson(0.2, (Q, D))
    par = dict(PAR, Prenormalized=True) if case == "prenormalized" else dict(PAR)
    rows_list = [np.sort(rng.choice(VM, VM // 2, replace=False)), np.arange(VM), np.sort(rng.choice(VM, 70, replace=False))]
    got = _segments_direct(descS, descM, rows_list, par, metric=True)
    for z, r in enumerate(rows_list):
        want = pc.getMatches(descS, descM[r], par)
        np.testing.assert_array_equal(got[z][0], want, err_msg=f"{case}, segment {z}")
        np.testing.assert_array_equal(want, oracle_c.getMatches(descS, descM[r], par))


def test_segmented_get_matches_with_a_bounded_workspace_at_400k_x_6k_x_2000_spheres():
    """VERDICT r3 item 7 / ADVICE: the one-chain form would need tens of GB here (a 400 k x 6 k score matrix, 2000 x 6 k x 32 list
    entries); pcreg_dev_get_matches_segmented runs it in batches of consecutive spheres on gathered sub-models inside a
    workspace of <= 4 GB + O(VM + rows).  Sampled spheres must equal the per-sphere call (SphereSweep.match_sphere ->
    pcreg_dev_get_matches on the gathered rows)."""
    import torch
    from pcreg_amd._lib import lib
    from pcreg_amd.sweep import SphereSweep
    dev = torch.device("cuda", 0)
    VM, VS, D = 400_000, 6_000, 980
    rng = np.random.default_rng(5)
    box = np.array([120.0, 100.0, 66.0])
    featM = rng.uniform(0, 1, (VM, 3)) * box
    g = torch.Generator(device=dev); g.manual_seed(3)
    descM = torch.poisson(torch.full((VM, D), 3.0, device=dev), generator=g).to(torch.float64)
    near = np.argsort(np.linalg.norm(featM - box / 2, axis=1))[:VS]
    featS = featM[near] + rng.normal(0, 0.02, (VS, 3))
    descS = (descM[torch.from_numpy(near).to(dev)] + torch.poisson(torch.full((VS, D), 0.15, device=dev), generator=g).to(torch.float64)).contiguous()
    sw = SphereSweep(featM, descM, featS, descS, device=dev)
    kw = dict(R_desc=9.0, d_spheres=7.0, min_pts=700)
    centres = sw.sphere_centres(kw["d_spheres"])
    valid, counts = sw.valid_spheres(centres, kw["R_desc"], kw["min_pts"])
    S, tot, n_max = int(valid.sum()), int(counts[valid].sum()), int(counts[valid].max())
    assert S > 1500
    wsb = lib().pcreg_dev_get_matches_segmented_workspace(VS, VM, D, S, tot, n_max)
    assert wsb < (4 << 30) + 16 * (VM + tot + S) + (1 << 20), wsb                  # the bound INTEGRATION.md states
    opt = dict(minPtNum=3, iterNum=300, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    out = sw.run(PAR, opt, putative_thresh=170, seed=0, **kw)
    assert len(out["centres"]) == S
    hit = np.argsort(-out["num_putative"])[:3].tolist()                            # the spheres that hold the surface's originals
    for i in sorted(set(hit + [0, S // 3, S - 1])):
        m = sw.match_sphere(out["centres"][i], kw["R_desc"], PAR)
        np.testing.assert_array_equal(m["rows"].cpu().numpy().astype(np.int64), out["model_rows"][i])
        np.testing.assert_array_equal(m["pairs"][:m["num_putative"]].cpu().numpy().astype(np.uint32), out["matches"][i])
    assert out["num_putative"].max() > 170 and any(t is not None for t in out["transforms"])
