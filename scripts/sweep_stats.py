"""The certified matcher's counters for one segmented sweep at the reference's shape (Poisson rows): how many (sphere, query)
items reach the exact re-rank, how many candidates they carry inside the slack, how many stay unproven."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcreg_amd._lib import lib, check
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
L = lib()
check(L.pcreg_debug_set(b"match_stats", 1))
out = sw.run(par, opt, R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
st = (C.c_longlong * 8)()
check(L.pcreg_debug_match_stats(st, 1))
S = len(out["centres"])
print(dict(spheres=S, items=S * VS, reranked=st[0], inside_slack_per_reranked=round(st[1] / max(st[0], 1), 2), unproven=st[2], to_exhaustive=st[3],
           back_items=st[4], back_to_exhaustive=st[5], trials=len(out["trial"]),
           pairs_per_trial=[int(x) for x in np.sort(out["num_putative"][out["trial"]])[::-1][:12]],
           median_pairs=float(np.median(out["num_putative"][out["trial"]]))))
