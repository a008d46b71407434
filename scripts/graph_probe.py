"""Does capturing one registration step in a HIP graph (torch.cuda.CUDAGraph) shorten it, and do two lanes overlap?
usage: graph_probe.py [model_points [iterNum]]   -- eager, graph replay, two eager lanes, two graphs on two streams."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import synth, MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF
from pcreg_amd.device import PreparedModel, RegistrationPipeline, soa
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
coef = dict(RANSAC_COEF, iterNum=int(sys.argv[2]) if len(sys.argv) > 2 else RANSAC_COEF["iterNum"])
Q, K = 50_000, 40
dev = torch.device("cuda", 0)
model, surf, _ = synth(M, Q)
ms, qs = PreparedModel(soa(torch.from_numpy(model).to(dev))), soa(torch.from_numpy(surf).to(dev))
pipes = [RegistrationPipeline(Q, M, device=dev) for _ in range(2)]
def step(p):
    p.match(qs, ms, MATCH_THR_ABS, MATCH_RATIO, True); p.ransac(coef, seed=7)
def timed(fn, reps=K):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
eager = timed(lambda: step(pipes[0]))
ref = pipes[0].fetch_result()
# host cost of enqueueing one step (no synchronisation inside the loop; the queue is deep enough for 10 steps)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step(pipes[0])
host = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
graphs = []
for p, s in zip(pipes, streams):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step(p)                 # warm the side stream's scratch
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step(p)
    graphs.append(g)
torch.cuda.synchronize()
graph = timed(lambda: graphs[0].replay())
res = pipes[0].fetch_result()
same = res["maxInliers"] == ref["maxInliers"] and res["numSuccess"] == ref["numSuccess"] and np.array_equal(res["T"], ref["T"])
def two_eager():
    for p, s in zip(pipes, streams):
        with torch.cuda.stream(s): step(p)
def two_graphs():
    for g, s in zip(graphs, streams):
        with torch.cuda.stream(s): g.replay()
e2 = timed(two_eager, K // 2) / 2
g2 = timed(two_graphs, K // 2) / 2
print(f"M={M} iterNum={coef['iterNum']}: eager {eager:.4f} ms/step (host enqueue {host:.4f}), graph replay {graph:.4f}, two eager lanes {e2:.4f}, "
      f"two graphs on two streams {g2:.4f} ms per registration; same result: {same}", flush=True)
