"""Time pcreg_dev_ransac alone for a few (n, iterNum, REFINE, inlier-fraction) points."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcreg_amd.device import RegistrationPipeline
from oracle.pcreg_oracle import eul2rotm
dev = torch.device("cuda", 0)
def case(n, frac_out, seed=0):
    rng = np.random.default_rng(seed)
    p2 = rng.uniform(0, 40, (n, 3)); R = eul2rotm([0.01, -0.008, 0.012]); t = np.array([0.15, -0.1, 0.2])
    p1 = p2 @ R + t + rng.normal(0, 0.05, p2.shape)
    k = int(frac_out * n); p1[:k] = rng.uniform(0, 40, (k, 3))
    return p1, p2
CASES = [(32558, 10000, True, 0.0), (32558, 10000, False, 0.0), (32558, 10000, True, 0.97), (8000, 10000, True, 0.0), (2000, 10000, True, 0.0), (1000, 20000, True, 0.0), (1000, 20000, True, 0.95)]
if len(sys.argv) > 1:
    CASES = [tuple(float(x) if '.' in x else int(x) for x in a.split(',')) for a in sys.argv[1:]]
    CASES = [(int(n), int(it), bool(int(r)), float(fo)) for n, it, r, fo in CASES]
for n, iters, refine, fo in CASES:
    p1, p2 = case(n, fo)
    pipe = RegistrationPipeline(n, 16, device=dev)
    t1 = torch.from_numpy(np.ascontiguousarray(p1.T)).to(dev); t2 = torch.from_numpy(np.ascontiguousarray(p2.T)).to(dev)
    nd = torch.tensor([n], dtype=torch.int32, device=dev)
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.3, thInlrRatio=0.08, REFINE=refine)
    for _ in range(2): pipe.ransac(coef, seed=7, n_dev=nd, pts1=t1, pts2=t2)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pipe.ransac(coef, seed=7, n_dev=nd, pts1=t1, pts2=t2); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    r = pipe.fetch_result()
    print(f"n={n} iters={iters} refine={refine} outliers={fo}: {np.median(ts):.3f} ms  (numSuccess {r['numSuccess']}, maxInl {r['maxInliers']})", flush=True)
