"""Prepared models (pcreg_dev_model_* / pcreg_model_*): one model, many surfaces -- the reference's shape
(completeExperimentFast.m:131-149).  Every search against a handle must return the oracle's bits, whatever the surface."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(M, seed):
    rng = np.random.default_rng(seed)
    return (rng.random((M, 3)) * [100, 56, 99]).astype(np.float32)


def _surfaces(model, Q, seed):
    """surfaces of very different kinds against one model: a noisy crop, a cloud far outside the model's box, points
    so far away that they are not scored on the matrix cores at all, and a mix with exact duplicates of model points"""
    rng = np.random.default_rng(seed)
    M = len(model)
    crop = (model[rng.choice(M, Q, replace=False)] + rng.normal(0, 0.05, (Q, 3))).astype(np.float32)
    outside = (rng.random((Q, 3)) * [30, 30, 30] + [400, -300, 250]).astype(np.float32)
    far = crop.copy(); far[::3] += np.float32(3e6); far[1::7] = np.float32(-8e8)
    dup = crop.copy(); dup[: Q // 4] = model[rng.choice(M, Q // 4, replace=False)]
    return dict(crop=crop, outside=outside, far=far, dup=dup)


@pytest.mark.parametrize("M,Q", [(60000, 5000), (9000, 700), (300, 40)])
def test_one_prepared_model_many_surfaces(M, Q, oracle_c):
    from pcreg_amd.device import PreparedModel, RegistrationPipeline, soa
    dev = torch.device("cuda", 0)
    model = _model(M, 7 + M)
    pm = PreparedModel(soa(torch.from_numpy(model).to(dev)))
    pipe = RegistrationPipeline(Q, M, device=dev)
    for kind, surf in _surfaces(model, Q, M + Q).items():
        qs = soa(torch.from_numpy(surf).to(dev))
        pipe.search_local(qs, pm)
        idx, dist = pipe._local
        ridx, rdist = oracle_c.knn2_points_f32(surf, model)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx, err_msg=kind)
        np.testing.assert_array_equal(dist.cpu().numpy(), rdist, err_msg=kind)
        for unique in (True, False):
            pairs, p1, p2, n = pipe.match_after_search(qs, pm, 0.25 if kind != "outside" else 1e30, 0.8 if kind != "outside" else 1.0, unique)
            n = int(n.item())
            ref = oracle_c.match_points_f32(surf, model, 0.25 if kind != "outside" else 1e30, 0.8 if kind != "outside" else 1.0, unique)
            np.testing.assert_array_equal(pairs[:n].cpu().numpy().astype(np.uint32), ref, err_msg=f"{kind} unique={unique}")
            np.testing.assert_array_equal(p1[:, :n].cpu().numpy().T, surf[ref[:, 0] - 1].astype(np.float64))
            np.testing.assert_array_equal(p2[:, :n].cpu().numpy().T, model[ref[:, 1] - 1].astype(np.float64))
    pm.close()


def test_match_stage_reruns_on_one_search(oracle_c):
    """The match launch leaves the search workspace's counters as it found them: other thresholds on the same search."""
    from pcreg_amd.device import RegistrationPipeline, soa
    dev = torch.device("cuda", 0)
    model = _model(40000, 1); surf = _surfaces(model, 6000, 2)["dup"]
    ms, qs = soa(torch.from_numpy(model).to(dev)), soa(torch.from_numpy(surf).to(dev))
    pipe = RegistrationPipeline(len(surf), len(model), device=dev)
    pipe.search_local(qs, ms)
    for thr, ratio, unique in ((0.25, 0.8, True), (0.05, 0.6, True), (0.25, 0.8, False), (0.25, 0.8, True)):
        pairs, _, _, n = pipe.match_after_search(qs, ms, thr, ratio, unique)
        np.testing.assert_array_equal(pairs[:int(n.item())].cpu().numpy().astype(np.uint32), oracle_c.match_points_f32(surf, model, thr, ratio, unique))


def test_a_new_tensor_is_prepared_again(oracle_c):
    """The pipeline keys its handle on the tensor OBJECT and its version: another model, or the same tensor written in
    place, is prepared again instead of being searched through a stale handle."""
    from pcreg_amd.device import RegistrationPipeline, soa
    dev = torch.device("cuda", 0)
    a, b = _model(20000, 3), _model(20000, 4)
    surf = (a[:3000] + 0.01).astype(np.float32)
    qs = soa(torch.from_numpy(surf).to(dev))
    pipe = RegistrationPipeline(3000, 20000, device=dev)
    t = soa(torch.from_numpy(a).to(dev))
    pipe.search_local(qs, t)
    np.testing.assert_array_equal(pipe._local[0].cpu().numpy(), oracle_c.knn2_points_f32(surf, a)[0])
    t.copy_(soa(torch.from_numpy(b).to(dev)))                       # same object, new contents
    pipe.search_local(qs, t)
    np.testing.assert_array_equal(pipe._local[0].cpu().numpy(), oracle_c.knn2_points_f32(surf, b)[0])


def test_host_tier_model_handle(oracle_c):
    """pcreg_model_create / pcreg_model_match_points_f32: the handle MATLAB keeps between getMatches-like calls."""
    import pcreg_amd as pc
    model = _model(30000, 11)
    with pc.Model(model) as h:
        for k, surf in enumerate(_surfaces(model, 2500, 5).values()):
            got = h.match_points(surf, 0.25, 0.8, True)
            np.testing.assert_array_equal(got, oracle_c.match_points_f32(surf, model, 0.25, 0.8, True))
            np.testing.assert_array_equal(got, pc.match_points(surf, model, 0.25, 0.8, True))
    with pytest.raises(Exception):
        h.match_points(surf, 0.25, 0.8, True)                        # closed


def test_all_queries_unproven_take_the_tiled_tail(oracle_c):
    """More than 1024 unproven queries: the tail kernel's tiled form (tiles of 1024 listed queries x model chunks, the
    last workgroup of a tile merges)."""
    import pcreg_amd as pc
    rng = np.random.default_rng(23)
    m = (rng.random((30000, 3)) * 5 + 4000.0).astype(np.float32)
    q = (m[rng.choice(30000, 2600)] + rng.normal(0, 0.05, (2600, 3))).astype(np.float32)
    m = np.vstack([m, np.array([[9e5, -9e5, 9e5]], np.float32)])
    idx, dist = pc.knn2_points(q, m)
    ridx, rdist = oracle_c.knn2_points_f32(q, m)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)


def test_match_stage_with_several_chunks_per_workgroup(oracle_c):
    """More than 2048 x 32 queries: a workgroup of the match launch owns several chunks of 32 queries (ordered compaction
    across chunks, workgroups and the ticket order)."""
    import pcreg_amd as pc
    rng = np.random.default_rng(77)
    model = (rng.random((20000, 3)) * [40, 30, 35]).astype(np.float32)
    surf = (model[rng.integers(0, 20000, 70001)] + rng.normal(0, 0.05, (70001, 3))).astype(np.float32)
    for unique in (True, False):
        got = pc.match_points(surf, model, 0.25, 0.9, unique)
        np.testing.assert_array_equal(got, oracle_c.match_points_f32(surf, model, 0.25, 0.9, unique))
    assert len(got) > 20000
