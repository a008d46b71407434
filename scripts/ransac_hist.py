"""Distribution of the first-pass inlier counts of the bench step's RANSAC (how sparse are the refit masks?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcreg_amd as pc
from bench import synth, MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF
model, surf, _ = synth(1_000_000, 50_000)
pairs = pc.match_points(surf, model, MATCH_THR_ABS, MATCH_RATIO, True)
p1 = surf[pairs[:, 0] - 1].astype(np.float64); p2 = model[pairs[:, 1] - 1].astype(np.float64)
res = pc.ransac(p1, p2, dict(RANSAC_COEF, VERBOSE=0), pc.estimateTransform, pc.calcDists, seed=7, return_iter_counts=True)
n = len(pairs); c1 = np.asarray(res[5]); c2 = np.asarray(res[6])
z = n - c1
print("n", n, "hyps", len(c1), "numSuccess", res[2], "maxInliers", res[3])
print("zeros of the first-pass masks: percentiles 1/10/50/90/99/100:", np.percentile(z, [1, 10, 50, 90, 99, 100]).astype(int))
for t in (64, 256, 1024, 2048, 4096, n // 4, n // 2):
    print(f"  hypotheses with min(c, n - c) <= {t}: {(np.minimum(c1, z) <= t).mean():.3f}")
print("sum of minority sizes / (hyps * n): %.4f" % (np.minimum(c1, z).sum() / (len(c1) * n)))
