"""The sphere sweep on rows the library's own descriptor kernel produces (bench.py's desc_chain shape), for rocprofv3."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcreg_amd.sweep import SphereSweep
dev = torch.device("cuda", 0)
d = bench.desc_chain_data(dev, 60_000, 2_000, compact=False)
VM, VS = int(d["VM"]), int(d["VS"])
sw = SphereSweep(d["featM"][:VM], d["descM"][:VM], d["featS"][:VS], d["descS"][:VS], device=dev)
opt = dict(minPtNum=3, iterNum=10000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = sw.run(bench.MATCH_PAR, opt, **kw)
torch.cuda.synchronize()
print(len(out["centres"]), len(out["trial"]))
