"""Randomised parity sweeps of the certified paths against the exhaustive C oracle: whatever the
geometry does to the seeding grid, the f16 split or the certificate, the answer must be the oracle's
bits (unproven queries fall back to the exact kernels)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(rng, n, kind):
    if kind == "uniform":
        return rng.uniform([0, 0, 0], [80, 50, 60], (n, 3))
    if kind == "blobs":
        c = rng.uniform(-100, 100, (rng.integers(1, 6), 3))
        return c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.01, 3.0), (n, 3))
    if kind == "plane":                                   # zero extent along z: degenerate grid / scale
        p = rng.uniform(0, 30, (n, 3)); p[:, 2] = 7.25; return p
    if kind == "line":
        t = rng.uniform(0, 1, n); return np.outer(t, [40.0, 0.0, 0.0]) + [1.0, 2.0, 3.0]
    if kind == "offset":                                  # large coordinates, small spacing: the error bound grows
        return rng.uniform(0, 5, (n, 3)) + [4000.0, -2500.0, 900.0]
    if kind == "tiny":
        return rng.uniform(0, 1e-3, (n, 3))
    if kind == "dupes":
        p = rng.uniform(0, 20, (max(n // 3, 1), 3)); return p[rng.integers(0, len(p), n)]
    raise ValueError(kind)


KINDS = ["uniform", "blobs", "plane", "line", "offset", "tiny", "dupes"]


# PCREG_FUZZ_SCALE=n runs n times as many seeds of every fuzzer (a soak after kernel changes)
_SCALE = max(1, int(os.environ.get("PCREG_FUZZ_SCALE", "1")))


@pytest.mark.parametrize("seed", range(6 * _SCALE))
def test_knn2_fuzz(seed, oracle_c):
    import pcreg_amd as pc
    rng = np.random.default_rng(1000 + seed)
    for kind in KINDS:
        M = int(rng.choice([1, 2, 3, 5, 63, 700, 16383, 16384, 20000, 70000]))
        Q = int(rng.choice([1, 7, 64, 513, 2500]))
        m = _cloud(rng, M, kind).astype(np.float32)
        src = rng.choice(["same", "near", "far"])
        if src == "same":
            q = _cloud(rng, Q, kind).astype(np.float32)
        elif src == "near":                               # noisy copies of model points (ties / tiny distances)
            q = (m[rng.integers(0, M, Q)] + rng.normal(0, 1e-3, (Q, 3))).astype(np.float32)
        else:                                             # queries far outside the model's box
            q = (_cloud(rng, Q, kind) * 3.0 + 500.0).astype(np.float32)
        idx, dist = pc.knn2_points(q, m)
        ridx, rdist = oracle_c.knn2_points_f32(q, m)
        np.testing.assert_array_equal(idx, ridx, err_msg=f"{kind} {src} Q={Q} M={M}")
        np.testing.assert_array_equal(dist, rdist, err_msg=f"{kind} {src} Q={Q} M={M}")


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_match_features_sad_fuzz(seed, oracle_c):
    """Descriptor matching: counts, constants, duplicates, heavy ties, tiny D, Q > M and Q < M."""
    import pcreg_amd as pc
    rng = np.random.default_rng(2000 + seed)
    for _ in range(6):
        Q, M, D = int(rng.choice([1, 3, 130, 700])), int(rng.choice([1, 2, 65, 900, 2600])), int(rng.choice([1, 2, 17, 64, 301]))
        lam = float(rng.choice([0.05, 1.0, 5.0]))
        dM = rng.poisson(lam, (M, D)).astype(np.float64)
        dS = rng.poisson(lam, (Q, D)).astype(np.float64)
        k = min(Q, M) // 2
        if k:
            dS[:k] = dM[rng.choice(M, k, replace=False)] + (rng.poisson(0.2, (k, D)) if rng.random() < 0.7 else 0)
        if rng.random() < 0.3:
            dM[rng.integers(0, M)] = dM[0]                 # duplicate model rows
        for unique in (True, False):
            kw = dict(Metric="SAD", MatchThreshold=float(rng.choice([5.0, 30.0, 100.0])), MaxRatio=float(rng.choice([0.6, 0.99, 1.0])), Unique=unique)
            pairs, met = pc.matchFeatures(dS, dM, **kw)
            rp, rm = oracle_c.matchFeatures(dS, dM, kw)
            np.testing.assert_array_equal(pairs, rp, err_msg=f"Q={Q} M={M} D={D} {kw}")
            np.testing.assert_array_equal(met, rm)


@pytest.mark.parametrize("seed", range(3 * _SCALE))
def test_align_points_knn_fuzz(seed, oracle_c):
    """Supports with heavy ties at the K-th distance (lattices, duplicates, shells), tiny and large sizes:
    the selection (histogram + exact rank, crowded-bin bisection, tie ranks) must pick the oracle's rows."""
    import pcreg_amd as pc
    rng = np.random.default_rng(3000 + seed)
    sups = []
    for _ in range(40):
        n = int(rng.choice([2, 3, 5, 64, 257, 1000, 3000, 7000]))
        kind = rng.choice(["gauss", "lattice", "dupes", "shell", "allsame"])
        if kind == "gauss":
            X = rng.normal(size=(n, 3)) * rng.uniform(0.1, 5, 3) + rng.uniform(-100, 100, 3)
        elif kind == "lattice":
            X = rng.integers(-2, 3, (n, 3)).astype(float)
        elif kind == "dupes":
            base = rng.normal(size=(max(n // 4, 1), 3)); X = base[rng.integers(0, len(base), n)]
        elif kind == "shell":                     # every point (nearly) equidistant from the centroid: one crowded bin
            v = rng.normal(size=(n, 3)); X = v / np.linalg.norm(v, axis=1, keepdims=True) * 2.0
        else:
            X = np.tile(rng.normal(size=(1, 3)), (n, 1))
        sups.append(X)
    al, co, c, status = pc.AlignPoints_KNN_batched(sups)
    for b, X in enumerate(sups):
        if status[b]:
            continue
        ral, rco, rc = oracle_c.AlignPoints_KNN(X)
        if not np.isfinite(rco).all():
            continue
        assert np.abs(c[b] - rc).max() < 1e-9
        # degenerate covariances (lattice / identical points) make eigenvectors non-unique: compare what is determined.
        # The covariance that matters is that of the K rows nearest to the centroid (AlignPoints_KNN.m:20-34): four
        # coplanar lattice points out of five leave the normal's sign to rounding noise in any eigen-solver.
        cc = X.mean(0); K = int(np.floor(len(X) * 0.85 + 0.5))
        sub = X[np.argsort(np.linalg.norm(X - cc, axis=1), kind="stable")[:K]]
        ev = np.linalg.eigvalsh(np.cov((sub - sub.mean(0)).T)) if K > 1 else np.zeros(3)
        # pca's sign convention (largest-magnitude entry of a column positive) is decided by rounding noise when a column's
        # two largest entries are equal in magnitude (e.g. (1, 0, 1) / sqrt 2 on a symmetric 5-point support)
        top2 = np.sort(np.abs(rco), axis=0)[-2:]
        if np.min(top2[1] - top2[0]) < 1e-9:
            continue
        if ev.min() > 1e-9 and np.min(np.diff(np.sort(ev))) > 1e-6 * ev.max():
            assert np.abs(co[b] - rco).max() < 1e-7, (b, len(X))
            assert np.abs(al[b] - ral).max() < 1e-6 * (1 + np.abs(X).max()), (b, len(X))


@pytest.mark.parametrize("seed", range(2 * _SCALE))
def test_descriptors_fuzz(seed, oracle_c):
    """Descriptor counts vs the oracle on clouds with duplicated points (ties at the K-th distance) and
    different densities / option sets."""
    import pcreg_amd as pc
    from test_gpu_descriptors import OPT, keypoints, strips
    rng = np.random.default_rng(4000 + seed)
    for _ in range(3):
        base = strips(int(rng.choice([12000, 30000])), int(rng.integers(0, 100)))
        pts = np.vstack([base, base[::int(rng.choice([2, 3, 5]))]]) if rng.random() < 0.6 else base
        kp = keypoints(int(rng.choice([40, 150])), int(rng.integers(0, 100)))
        opt = dict(OPT, min_pts=int(rng.choice([50, 150])), k=float(rng.choice([0.85, 0.6, 0.95])), ALIGN_POINTS=bool(rng.integers(0, 2)),
                   R=float(rng.choice([2.5, 3.5])))
        feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt)
        rfeat, rdesc = oracle_c.getSpacialHistogramDescriptors(pts, kp, opt)
        np.testing.assert_array_equal(feat, rfeat)
        np.testing.assert_array_equal(desc, rdesc)


@pytest.mark.parametrize("seed", range(3 * _SCALE))
def test_ransac_fuzz(seed, oracle_c):
    """Random sizes, outlier shares, iteration counts and thresholds: per-iteration counts, numSuccess, maxInliers and the inlier set
    are the oracle's bits whatever kernel the size selects (LDS-resident, tiled or staged) and however many hypotheses refit."""
    import pcreg_amd as pc
    from conftest import rigid_case
    from test_gpu_ransac import _cmp
    rng = np.random.default_rng(5000 + seed)
    for _ in range(4):
        n = int(rng.choice([12, 40, 150, 600, 1300, 1400, 2500, 4200, 6000]))
        iters = int(rng.integers(60, 900))
        frac = float(rng.choice([0.0, 0.2, 0.5, 0.8]))
        refine = bool(rng.random() < 0.8)
        p1, p2, _ = rigid_case(n, int(rng.integers(0, 10 ** 6)), noise=float(rng.choice([0.005, 0.02, 0.05])), outlier_frac=frac)
        coef = dict(minPtNum=3, iterNum=iters, thDist=float(rng.choice([0.01, 0.05, 0.3])), thInlrRatio=float(rng.choice([0.05, 0.1, 0.3])),
                    REFINE=refine, VERBOSE=0)
        ref = oracle_c.ransac(p1, p2, coef, seed=seed)
        if ref["failed"]:
            T = pc.ransac(p1, p2, coef, pc.estimateTransform, pc.calcDists, seed=seed)[0]
            assert T is None or np.size(T) == 0
            continue
        res = pc.ransac(p1, p2, coef, pc.estimateTransform, pc.calcDists, seed=seed, return_iter_counts=True)
        _cmp(res, ref, n)


@pytest.mark.parametrize("seed", range(2 * _SCALE))
def test_segmented_get_matches_fuzz(seed):
    """Random segment structures and matchFeatures settings: the segmented chain (shared scores, per-segment certificates, the
    queries skipped on the strength of their score bounds, the Unique back-check's shortcuts) gives the pairs of one getMatches
    call per segment."""
    import pcreg_amd as pc
    from test_gpu_sweep import PAR, _segments_direct
    rng = np.random.default_rng(6000 + seed)
    for _ in range(3):
        VM, Q, D = int(rng.integers(200, 2500)), int(rng.integers(40, 500)), int(rng.choice([6, 24, 64, 160]))
        lam = float(rng.choice([0.5, 3.0, 12.0]))
        descM = rng.poisson(lam, (VM, D)).astype(np.float64)
        pick = rng.choice(VM, Q, replace=Q > VM)
        descS = descM[pick] + rng.poisson(float(rng.choice([0.05, 0.3, 1.0])), (Q, D))
        if rng.random() < 0.5:
            descS[rng.integers(0, Q)] = descS[rng.integers(0, Q)]          # duplicate surface rows: Unique ties
        rows_list = []
        for _s in range(int(rng.integers(2, 7))):
            k = int(rng.choice([0, 1, 2, 5, VM // 7, VM // 2, VM]))
            rows_list.append(np.sort(rng.choice(VM, k, replace=False)) if k else np.zeros(0, np.int64))
        par = dict(PAR, Unique=bool(rng.random() < 0.7), MaxRatio=float(rng.choice([0.6, 0.9, 0.99, 1.0])),
                   MatchThreshold=float(rng.choice([1.0, 5.0, 10.0, 40.0, 100.0])), UNNORMALIZE=bool(rng.random() < 0.8),
                   CHANGE_METRIC=bool(rng.random() < 0.8))
        got = _segments_direct(descS, descM, rows_list, par, metric=True)
        for z, r in enumerate(rows_list):
            if len(r) == 0:
                assert got[z][0].shape[0] == 0
                continue
            np.testing.assert_array_equal(got[z][0], pc.getMatches(descS, descM[r], par), err_msg=f"seed {seed}, segment {z}, {par}")
