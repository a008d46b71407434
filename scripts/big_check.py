import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import synth
from pcreg_amd.device import RegistrationPipeline, soa
import oracle.c_oracle as oc
oc.build()
dev = torch.device("cuda", 0)
for Q, M in [(200000, 4000000), (1000, 8000000), (300000, 300000)]:
    model, surf, _ = synth(M, Q)
    ms, qs = soa(torch.from_numpy(model).to(dev)), soa(torch.from_numpy(surf).to(dev))
    pipe = RegistrationPipeline(Q, M, device=dev)
    pipe.search_local(qs, ms); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pipe.search_local(qs, ms); b.record(); torch.cuda.synchronize()
    idx, dist = pipe._local
    # spot-check 2000 random queries against the C oracle (exhaustive, fp32 chain)
    rng = np.random.default_rng(0); sel = rng.choice(Q, min(Q, 2000), replace=False)
    ri, rd = oc.knn2_points_f32(surf[sel], model)
    gi, gd = idx.cpu().numpy()[sel], dist.cpu().numpy()[sel]
    ok = np.array_equal(gi, ri) and np.array_equal(gd, rd)
    print(f"Q={Q} M={M}: {a.elapsed_time(b):.2f} ms, {Q*M/a.elapsed_time(b)/1e6:.0f} Gpairs/s, spot-check bit-exact: {ok}", flush=True)
