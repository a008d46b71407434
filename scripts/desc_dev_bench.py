"""cfg 4 on the device tier: getSpacialHistogramDescriptors (uint16 rows written once) on the ridge cloud; wrap in rocprofv3
(scripts/prof_stats.sh) for kernel times, or in scripts/pmc_*.sh for counters.  args: P S [stop]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _ridge_cloud
from pcreg_amd.device import DescriptorPipeline
P, S = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (1000000, 100000)))
SM = 1 if (len(sys.argv) > 3 and sys.argv[3] == "single") else 0           # single: the data as pcread delivers it (single), MATLAB's arithmetic
pts, kp = _ridge_cloud(P, S)
if SM:
    pts, kp = pts.astype(np.float32).astype(np.float64), kp.astype(np.float32).astype(np.float64)
dev = torch.device("cuda", 0)
opt = dict(min_pts=500, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)
tp = torch.from_numpy(np.ascontiguousarray(pts.T)).to(dev); tk = torch.from_numpy(np.ascontiguousarray(kp.T)).to(dev)
dp = DescriptorPipeline(dev)
dp.describe(tp[:, :50000].contiguous(), tk[:, :1000].contiguous(), opt, compact=True, single_mode=SM)
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); feat, rows, V = dp.describe(tp, tk, opt, compact=True, single_mode=SM); b.record(); torch.cuda.synchronize()
    print(f"P={P} S={S} single_mode={SM}: {V} descriptors, {a.elapsed_time(b):.2f} ms", flush=True)
