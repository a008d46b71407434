"""Does capturing one registration step in a HIP graph (torch.cuda.CUDAGraph) shorten it?  Eager vs replay."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import synth, MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF
from pcreg_amd.device import RegistrationPipeline, soa
dev = torch.device("cuda", 0)
model, surf, _ = synth(1_000_000, 50_000)
ms, qs = soa(torch.from_numpy(model).to(dev)), soa(torch.from_numpy(surf).to(dev))
pipe = RegistrationPipeline(50_000, 1_000_000, device=dev)
def step():
    pipe.match(qs, ms, MATCH_THR_ABS, MATCH_RATIO, True); pipe.ransac(RANSAC_COEF, seed=7)
for _ in range(3): step()
torch.cuda.synchronize()
ref = pipe.fetch_result()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()                      # warm the side stream's scratch
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 20
res = pipe.fetch_result()
same = res["maxInliers"] == ref["maxInliers"] and res["numSuccess"] == ref["numSuccess"] and np.array_equal(res["T"], ref["T"])
print(f"eager {eager*1e3:.4f} ms/step, graph replay {graph*1e3:.4f} ms/step, same result: {same}", flush=True)
