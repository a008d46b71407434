"""Throughput of the sphere sweep with K surfaces in flight against ONE model: K threads, each with its own SphereSweep (own HIP
stream, own workspaces; the model's tensors are shared), every thread sweeping its surface N times.  Prints ms per sweep."""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcreg_amd.sweep import SphereSweep
VM, VS, D = 60000, 2000, 980
rng = np.random.default_rng(0)
dev = torch.device("cuda", 0)
featM = rng.uniform([0, 0, 0], [60, 50, 40], (VM, 3))
g = torch.Generator(device=dev); g.manual_seed(1)
descM = torch.poisson(torch.full((VM, D), 3.0, device=dev), generator=g).to(torch.float64)
par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate", MatchThreshold=10, MaxRatio=0.99,
           Metric="SAD", Unique=True, VERBOSE=0)
opt = dict(minPtNum=3, iterNum=10000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
def surface(k):
    near = np.argsort(np.linalg.norm(featM - np.array([31.0 - 3 * k, 24.0 + 2 * k, 19.0]), axis=1))[:VS]
    c, s = np.cos(0.3 + 0.1 * k), np.sin(0.3 + 0.1 * k)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    fS = featM[near] @ R.T + np.array([2.0, -1.0, 0.5]) + rng.normal(0, 0.02, (VS, 3))
    dS = (descM[torch.from_numpy(near).to(dev)] + torch.poisson(torch.full((VS, D), 0.15, device=dev), generator=g).to(torch.float64)).contiguous()
    return fS, dS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for K in (1, 2, 3):
    sws, streams, outs = [], [], [None] * K
    for k in range(K):
        fS, dS = surface(k)
        sws.append(SphereSweep(featM, descM, fS, dS, device=dev)); streams.append(torch.cuda.Stream(device=dev))
    def work(k, n):
        with torch.cuda.stream(streams[k]):
            for _ in range(n):
                outs[k] = sws[k].run(par, opt, **kw)
    for k in range(K): work(k, 2)                    # warm-up: spheres, powered rows, allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, N)) for k in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{K} in flight: {dt / (K * N) * 1e3:.2f} ms per sweep ({[len(o['trial']) for o in outs]} trials)", flush=True)
