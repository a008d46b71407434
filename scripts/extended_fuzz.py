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
