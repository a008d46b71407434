"""Time the match stage (one launch: filters + Unique + ordered compaction) on the bench's search result."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth, MATCH_THR_ABS, MATCH_RATIO
from pcreg_amd.device import RegistrationPipeline, soa
Q, M = 50000, 1000000
model, surf, _ = synth(M, Q)
dev = torch.device("cuda", 0)
ms, qs = soa(torch.from_numpy(model).to(dev)), soa(torch.from_numpy(surf).to(dev))
pipe = RegistrationPipeline(Q, M, device=dev)
pipe.search_local(qs, ms)
for unique in (True, False):
    for _ in range(3): pipe.match_after_search(qs, ms, MATCH_THR_ABS, MATCH_RATIO, unique)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pipe.match_after_search(qs, ms, MATCH_THR_ABS, MATCH_RATIO, unique); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print(f"unique={unique}: median {np.median(ts):.1f} us, min {min(ts):.1f} us, pairs {int(pipe.n_pairs.item())}", flush=True)
