#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the numpy oracle (oracle/pcreg_oracle.py).

The reference is MATLAB and cannot be run or imported here, so these vectors are
outputs of THIS repo's restatement (inputs + expected outputs), pinned separately
against the reference's own known answers in tests/test_oracle_kat.py.
Run:  python oracle/gen_golden.py     (deterministic; rewrites the fixtures)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pcreg_oracle as o  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rigid_case(n, seed, noise=0.02, outlier_frac=0.3):
    rng = np.random.default_rng(seed)
    box = np.array([30.0, 20.0, 25.0])
    P = rng.uniform(-1, 1, (n, 3)) * box + np.array([40.0, 25.0, 50.0])
    R = o.eul2rotm(rng.uniform(-np.pi, np.pi, 3)); t = rng.uniform(-10, 10, 3)
    pts1 = P @ R + t + rng.normal(0, noise, P.shape)
    k = int(outlier_frac * n)
    bad = rng.choice(n, k, replace=False)
    pts1[bad] = rng.uniform(-1, 1, (k, 3)) * box * 1.5 + np.array([40.0, 25.0, 50.0])
    return pts1, P


def main():
    os.makedirs(OUT, exist_ok=True)
    # --- ransac: built-in sampler and an explicit (MATLAB-randperm-like) table
    for name, n, iters, refine, seed in [("ransac_a", 400, 600, True, 5), ("ransac_b", 150, 400, False, 6)]:
        p1, p2 = rigid_case(n, seed)
        coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.1, REFINE=refine)
        table = o.sample_table(n, iters, 3, seed=123)
        r = o.ransac(p1, p2, coef, sample_idx=table)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), pts1=p1, pts2=p2, sample_idx=table,
                            coef=np.array([3, iters, 0.05, 0.1, float(refine)]), T=r["T"], inlierIdx=r["inlierIdx"],
                            numSuccess=r["numSuccess"], maxInliers=r["maxInliers"], inlrNum=r["inlrNum"],
                            inlrNum_refined=r["inlrNum_refined"], seed=123)
    # --- estimateTransform on 3, 4 and many points
    rng = np.random.default_rng(1)
    cases = []
    for n in (3, 4, 50):
        p1, p2 = rigid_case(n, 30 + n, noise=0.3, outlier_frac=0.0)
        cases.append((p1, p2, o.estimateTransform(p1, p2)))
    np.savez_compressed(os.path.join(OUT, "estimate_transform.npz"),
                        **{f"p1_{i}": c[0] for i, c in enumerate(cases)}, **{f"p2_{i}": c[1] for i, c in enumerate(cases)},
                        **{f"T_{i}": c[2] for i, c in enumerate(cases)})
    # --- fp32 point matching
    model = (rng.random((6000, 3)) * [100, 56, 99]).astype(np.float32)
    pick = rng.choice(6000, 1500, replace=False)
    surf = (model[pick] + rng.normal(0, 0.3, (1500, 3))).astype(np.float32)
    idx, dist = o.knn2_points_f32(surf, model)
    pairs = o.match_points_f32(surf, model, 0.5, 0.8, True)
    np.savez_compressed(os.path.join(OUT, "match_points.npz"), surf=surf, model=model, idx=idx.astype(np.int32),
                        dist=dist, pairs=pairs, thr=0.5, ratio=0.8)
    # --- getMatches on count descriptors (completeExperimentFast.m:75-87 settings)
    D = 60
    dM = rng.poisson(3.0, (300, D)).astype(np.float64)
    dS = rng.poisson(3.0, (120, D)).astype(np.float64)
    dS[:60] = dM[rng.choice(300, 60, replace=False)] + rng.poisson(0.2, (60, D))
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
               MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True)
    m_sad = o.getMatches(dS, dM, par)
    m_ssd = o.getMatches(dS, dM, dict(par, Metric="SSD"))
    np.savez_compressed(os.path.join(OUT, "get_matches.npz"), descSurface=dS, descModel=dM, matches_sad=m_sad,
                        matches_ssd=m_ssd)
    # --- AlignPoints_KNN
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    X = (rng.normal(size=(700, 3)) * [3.0, 1.5, 0.4]) @ A + [5.0, -20.0, 33.0]
    out = {}
    for C1 in (0, 1):
        for C2 in (0, 1):
            al, co, c = o.AlignPoints_KNN(X, bool(C1), bool(C2))
            out[f"aligned_{C1}{C2}"] = al; out[f"coeff_{C1}{C2}"] = co; out[f"c_{C1}{C2}"] = c
    np.savez_compressed(os.path.join(OUT, "align_points_knn.npz"), pts=X, **out)
    # --- getSpacialHistogramDescriptors (ridge-like strips so supports pass the variance test)
    per = 2000
    strips = []
    for s in range(8):
        x = rng.uniform(0, 60, per); y = 6 * s + rng.uniform(-1.2, 1.2, per); z = 10 + 0.1 * x * np.sin(s) + rng.normal(0, 0.25, per)
        strips.append(np.column_stack([x, y, z]))
    cloud = np.vstack(strips)
    kp = np.column_stack([rng.uniform(5, 55, 40), 6 * rng.integers(0, 8, 40) + rng.uniform(-1.5, 1.5, 40), rng.uniform(9, 16, 40)])
    opt = dict(min_pts=60, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True)
    f, d = o.getSpacialHistogramDescriptors(cloud, kp, opt)
    f2, d2 = o.getSpacialHistogramDescriptors(cloud, kp, dict(opt, ALIGN_POINTS=False))
    np.savez_compressed(os.path.join(OUT, "descriptors.npz"), pts=cloud, sample_pts=kp, feat=f, desc=d.astype(np.uint16),
                        feat_noalign=f2, desc_noalign=d2.astype(np.uint16))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
