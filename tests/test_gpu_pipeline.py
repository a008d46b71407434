"""Device tier (pcreg_dev_* through pcreg_amd.device) vs the oracle: resident
match -> RANSAC pipeline, and a two-shard search merged like the multi-GPU path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(M=30000, Q=4000, seed=0):
    from oracle.pcreg_oracle import eul2rotm
    rng = np.random.default_rng(seed)
    model = (rng.random((M, 3)) * [100, 56, 99]).astype(np.float32)
    pick = np.sort(rng.choice(M, Q, replace=False))
    R = eul2rotm([0.010, -0.008, 0.012]); t = np.array([0.15, -0.10, 0.20])
    c = model[pick].astype(np.float64); ctr = c.mean(axis=0)
    surf = ((c - ctr) @ R + ctr + t + rng.normal(0, 0.05, c.shape)).astype(np.float32)
    return model, surf


def test_resident_pipeline_matches_oracle(oracle_c):
    from pcreg_amd.device import RegistrationPipeline, soa
    model, surf = _data()
    dev = torch.device("cuda", 0)
    ms, qs = soa(torch.from_numpy(model).to(dev)), soa(torch.from_numpy(surf).to(dev))
    pipe = RegistrationPipeline(len(surf), len(model), device=dev)
    pairs, p1, p2, n = pipe.match(qs, ms, 0.25, 0.8, unique=True)
    coef = dict(minPtNum=3, iterNum=1500, thDist=0.3, thInlrRatio=0.08, REFINE=True)
    pipe.ransac(coef, seed=7)
    res = pipe.fetch_result()
    n = int(n.item())
    ref_pairs = oracle_c.match_points_f32(surf, model, 0.25, 0.8, True)
    np.testing.assert_array_equal(pairs[:n].cpu().numpy().astype(np.uint32), ref_pairs)
    rp1 = surf[ref_pairs[:, 0] - 1].astype(np.float64); rp2 = model[ref_pairs[:, 1] - 1].astype(np.float64)
    np.testing.assert_array_equal(p1[:, :n].cpu().numpy().T, rp1)
    np.testing.assert_array_equal(p2[:, :n].cpu().numpy().T, rp2)
    ref = oracle_c.ransac(rp1, rp2, coef, seed=7)
    assert res["n"] == n and not res["failed"]
    np.testing.assert_array_equal(res["inlierIdx"].astype(np.int64), ref["inlierIdx"])
    assert res["numSuccess"] == ref["numSuccess"] and res["maxInliers"] == ref["maxInliers"]
    assert np.linalg.norm(res["T"] - ref["T"]) < 1e-5


def test_two_shards_merge_like_two_gpus(oracle_c):
    """Shard the model rows in two, search each shard with its idx_base, merge the stacked
    lists with the same kernel the all_gather feeds: must equal the unsharded search."""
    from pcreg_amd.device import HipOps, soa
    model, surf = _data(M=20001, Q=3000, seed=3)
    dev = torch.device("cuda", 0)
    qs = soa(torch.from_numpy(surf).to(dev))
    cut = 9000
    outs = []
    for lo, hi in ((0, cut), (cut, len(model))):
        ops = HipOps(len(surf), hi - lo, dev)
        i, d = ops.local_top2(qs, soa(torch.from_numpy(model[lo:hi]).to(dev)), lo)
        outs.append((i.clone(), d.clone()))
    ops = HipOps(len(surf), len(model), dev)
    idx, dist = ops.merge_top2(torch.stack([o[0] for o in outs]).contiguous(), torch.stack([o[1] for o in outs]).contiguous())
    ridx, rdist = oracle_c.knn2_points_f32(surf, model)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(dist.cpu().numpy(), rdist)


def test_fast_path_equals_exact_kernel_on_hard_inputs(oracle_c):
    """Large coordinate offsets blow up the rounding bound: every query fails the
    certificate and is redone by the exact kernels -- the answer must not change."""
    import pcreg_amd as pc
    rng = np.random.default_rng(9)
    m = (rng.random((5000, 3)) * 5 + 4000.0).astype(np.float32)        # |x| ~ 4000, spacing ~ 0.3
    q = (m[rng.choice(5000, 2500)] + rng.normal(0, 0.05, (2500, 3))).astype(np.float32)
    idx, dist = pc.knn2_points(q, m)
    ridx, rdist = oracle_c.knn2_points_f32(q, m)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)
    # a far-away outlier point inflates R_m for everybody
    m2 = np.vstack([m, np.array([[9e5, -9e5, 9e5]], np.float32)])
    idx, dist = pc.knn2_points(q, m2)
    ridx, rdist = oracle_c.knn2_points_f32(q, m2)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)


def test_search_at_full_size_spot_check(oracle_c):
    """BASELINE's shape and beyond (Q = 100 k, M = 2 M): 1000 random queries against the exhaustive C oracle,
    indices and fp32 distances bit for bit (the size-independent property: every query is independent)."""
    import torch
    from bench import synth
    from pcreg_amd.device import RegistrationPipeline, soa
    Q, M = 100_000, 2_000_000
    model, surf, _ = synth(M, Q)
    dev = torch.device("cuda", 0)
    pipe = RegistrationPipeline(Q, M, device=dev)
    pipe.search_local(soa(torch.from_numpy(surf).to(dev)), soa(torch.from_numpy(model).to(dev)))
    idx, dist = pipe._local
    sel = np.random.default_rng(5).choice(Q, 1000, replace=False)
    ri, rd = oracle_c.knn2_points_f32(surf[sel], model)
    np.testing.assert_array_equal(idx.cpu().numpy()[sel], ri)
    np.testing.assert_array_equal(dist.cpu().numpy()[sel], rd)


def test_few_hundred_unproven_queries_take_the_sliced_fallback(oracle_c):
    """700 queries that all fail the certificate (a model point 1e6 away inflates the rounding bound): fewer than
    1024, so the sliced exact fallback runs, every workgroup looping over several flagged queries."""
    import pcreg_amd as pc
    rng = np.random.default_rng(19)
    m = (rng.random((40000, 3)) * 5 + 4000.0).astype(np.float32)
    q = (m[rng.choice(40000, 700)] + rng.normal(0, 0.05, (700, 3))).astype(np.float32)
    m = np.vstack([m, np.array([[9e5, -9e5, 9e5]], np.float32)])
    idx, dist = pc.knn2_points(q, m)
    ridx, rdist = oracle_c.knn2_points_f32(q, m)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)
