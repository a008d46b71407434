"""Crop-parallel batch (BASELINE cfg 5, pcreg_amd/batch.py): every crop's result equals the single-pipeline
result and the oracle's, whatever the number of streams."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _crops(M=24000, Q=2500, n=5):
    from oracle.pcreg_oracle import eul2rotm
    rng = np.random.default_rng(77)
    model = (rng.random((M, 3)) * [100, 56, 99]).astype(np.float32)
    out = []
    for c in range(n):
        r = np.random.default_rng(100 + c)
        ctr = np.array([100, 56, 99]) * r.uniform(0.3, 0.7, 3)
        pick = np.sort(np.argpartition(((model - ctr) ** 2).sum(1), Q - 1)[:Q])
        R = eul2rotm(r.uniform(-0.012, 0.012, 3)); t = r.uniform(-0.2, 0.2, 3)
        p = model[pick].astype(np.float64)
        out.append(((p - ctr) @ R + ctr + t + r.normal(0, 0.05, p.shape)).astype(np.float32))
    return model, out


@pytest.mark.parametrize("streams", [1, 3])
def test_batch_equals_oracle_per_crop(streams, oracle_c):
    from pcreg_amd.batch import BatchRegistration
    from pcreg_amd.device import soa
    model, crops = _crops()
    dev = torch.device("cuda", 0)
    ms = soa(torch.from_numpy(model).to(dev))
    qs = [soa(torch.from_numpy(c).to(dev)) for c in crops]
    coef = dict(minPtNum=3, iterNum=800, thDist=0.3, thInlrRatio=0.08, REFINE=True)
    br = BatchRegistration(ms, crops[0].shape[0], n_streams=streams, device=dev)
    res = br.run(qs, 0.25, 0.8, coef, seed=7)
    assert [r["crop"] for r in res] == list(range(len(crops)))
    for c, r in enumerate(res):
        ref_pairs = oracle_c.match_points_f32(crops[c], model, 0.25, 0.8, True)
        assert r["n_pairs"] == len(ref_pairs)
        rp1 = crops[c][ref_pairs[:, 0] - 1].astype(np.float64); rp2 = model[ref_pairs[:, 1] - 1].astype(np.float64)
        ref = oracle_c.ransac(rp1, rp2, coef, seed=7)
        assert not r["failed"] and r["numSuccess"] == ref["numSuccess"] and r["maxInliers"] == ref["maxInliers"]
        assert r["n_inliers"] == len(ref["inlierIdx"])
        assert np.linalg.norm(r["T"] - ref["T"]) < 1e-5
    # a second run on the same object reuses every workspace and must reproduce the rows bit for bit
    rows = br.rows.copy()
    br.run(qs, 0.25, 0.8, coef, seed=7)
    np.testing.assert_array_equal(rows, br.rows)
