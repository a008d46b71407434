"""One-off extended fuzz (not part of the suite): the parametrised fuzz tests of tests/test_gpu_fuzz.py and the
RANSAC / descriptor-chain tests at many more seeds.  usage: python scripts/extended_fuzz.py [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import c_oracle
c_oracle.build(); c_oracle.lib()
import test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.time()
for seed in range(100, 100 + n):
    F.test_knn2_fuzz(seed, c_oracle)
    F.test_align_points_knn_fuzz(seed, c_oracle)
    F.test_match_features_sad_fuzz(seed, c_oracle)
    if seed % 3 == 0:
        F.test_descriptors_fuzz(seed, c_oracle)
    print(f"seed {seed} ok ({time.time() - t0:.0f} s)", flush=True)
print("extended fuzz: all equal")

# RANSAC: ragged sizes on both sides of every kernel switch (LDS-resident / tiled / staged with the fp32 screen and the
# one-ahead row fetch), random outlier shares, thresholds and iteration counts (odd ones too: the rows are fetched in pairs)
import numpy as np
import pcreg_amd as pc
from conftest import rigid_case
import test_gpu_ransac as R
rng = np.random.default_rng(77)
for k in range(n * 2):
    nn = int(rng.choice([37, 500, 1365, 1366, 4095, 4096, 4097, 6001, 9999, 20000, 33333]))
    iters = int(rng.choice([1, 2, 31, 32, 33, 63, 65, 257, 700, 1001]))
    frac = float(rng.choice([0.0, 0.2, 0.5, 0.8]))
    refine = bool(rng.integers(0, 2))
    coef = dict(minPtNum=3, iterNum=iters, thDist=float(rng.choice([0.01, 0.05, 0.3])), thInlrRatio=float(rng.choice([0.05, 0.1, 0.5])), REFINE=refine, VERBOSE=0)
    p1, p2, _ = rigid_case(nn, 900 + k, noise=float(rng.choice([0.0, 0.02, 0.1])), outlier_frac=frac)
    ref = c_oracle.ransac(p1, p2, coef, seed=k)
    res = pc.ransac(p1, p2, coef, seed=k, return_iter_counts=True)
    np.testing.assert_array_equal(res[5], ref["inlrNum"]); np.testing.assert_array_equal(res[6], ref["inlrNum_refined"])
    assert res[2] == ref["numSuccess"]
    if ref["failed"]:
        assert np.asarray(res[0]).size == 0
    else:
        R._cmp(res, ref, nn)
    print(f"ransac case {k}: n={nn} iters={iters} frac={frac} refine={refine} ok", flush=True)
print("extended ransac fuzz: all equal")
