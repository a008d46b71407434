"""The sphere sweep of completeExperimentFast.m:46-224 at the reference's shape (surface ~2000 keypoints, >= 1400 model
descriptors per sphere, a few hundred spheres, D = 980): batched SphereSweep.run against the one-sphere-at-a-time
run_serial, spheres per second.  Prints one JSON line."""
import json, os, sys, time
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
kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
sw = SphereSweep(featM, descM, featS, descS, device=dev)
NS = int(os.environ.get("SWEEP_STREAMS", "8"))
def timed(fn, reps=3):
    fn(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return out, min(ts)
out, tb = timed(lambda: sw.run(par, opt, **kw))                                   # segmented: one launch chain for all spheres
strm, tst = timed(lambda: sw.run_streams(par, opt, n_streams=NS, **kw), reps=2)    # round 2: one chain per sphere on NS streams
ser, ts = timed(lambda: sw.run_serial(par, opt, **kw), reps=1)
same = all(np.array_equal(out[k], o2[k]) for o2 in (strm, ser) for k in ("num_putative", "trial", "statsSuccess", "statsInliers")) and \
       all(np.array_equal(a, b) for o2 in (strm, ser) for a, b in zip(out["matches"], o2["matches"]))
S = len(out["centres"])
print(json.dumps({"workload": f"sphere sweep: {S} valid spheres (of a {VM}-keypoint model, {int(out['num_desc'].mean())} descriptors per sphere on average), "
                              f"surface {VS} keypoints, D {D}, {len(out['trial'])} trial spheres x RANSAC(3,1e4,0.3,0.08,REFINE)",
                  "segmented_ms": round(tb * 1e3, 1), "segmented_spheres_per_s": round(S / tb, 1),
                  "streams_ms": round(tst * 1e3, 1), "streams_spheres_per_s": round(S / tst, 1), "streams": NS,
                  "serial_ms": round(ts * 1e3, 1), "serial_spheres_per_s": round(S / ts, 1), "same_results": bool(same), "host_syncs": 2}), flush=True)
