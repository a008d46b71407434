"""One segmented sphere sweep at the reference's shape (for rocprofv3; scripts/sweep_bench.py has the timing)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcreg_amd.sweep import SphereSweep
VM, VS, D = 60000, 2000, 980
rng = np.random.default_rng(0)
dev = torch.device("cuda", 0)
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
opt = dict(minPtNum=3, iterNum=10000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
sw = SphereSweep(featM, descM, featS, descS, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = sw.run(par, opt, R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
torch.cuda.synchronize()
print(len(out["centres"]), len(out["trial"]))
