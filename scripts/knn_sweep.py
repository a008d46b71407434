"""Time pcreg_dev_knn2_points_f32 alone (interleaved rounds in one process are not possible
across env-selected variants, so each variant runs in its own process; compare medians)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth
from pcreg_amd.device import RegistrationPipeline, soa
Q, M = 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
model, surf, _ = synth(M, Q)
dev = torch.device("cuda", 0)
ms, qs = soa(torch.from_numpy(model).to(dev)), soa(torch.from_numpy(surf).to(dev))
pipe = RegistrationPipeline(Q, M, device=dev)
for _ in range(3): pipe.search_local(qs, ms)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pipe.search_local(qs, ms); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(os.environ.get("PCREG_KNN_VARIANT", "0"), os.environ.get("PCREG_KNN_BLOCKS", "2048"), "median ms %.3f min %.3f" % (np.median(ts), min(ts)), flush=True)
