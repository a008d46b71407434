"""ms per registration against the number of registrations in flight (lanes) at the headline shape and at one rank's share of N = 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pcreg_amd._lib import check, lib
dev = torch.device("cuda", 0)
check(lib().pcreg_set_device(0))
ctx = bench.Ctx(0, 1, dev, False)
for M, coef in ((1_000_000, bench.RANSAC_COEF), (125_000, dict(bench.RANSAC_COEF, iterNum=1250))):
    row = []
    for k in (1, 2, 3, 4):
        r = bench.run_registration(ctx, M, 50_000, 48, 4, time_kernel=False, in_flight=k, coef=coef)
        row.append(f"{k}: {r['ms_per_step']:.4f}")
    print(f"M={M} iterNum={coef['iterNum']}  ms per registration by lanes  " + "  ".join(row), flush=True)
