"""Where the host spends the sphere sweep's wall time: cProfile of SphereSweep.run (GPU waits show up under .cpu() / .item())."""
import cProfile, os, pstats, runpy, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("sb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_prof.py"))
src = open(spec.origin).read().split("for _ in range(")[0]          # the data set-up of sweep_prof.py
ns = {"__name__": "setup", "__file__": spec.origin}; exec(compile(src, spec.origin, "exec"), ns)
sw, par, opt = ns["sw"], ns["par"], ns["opt"]
kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
sw.run(par, opt, **kw); sw.run(par, opt, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter(); sw.run(par, opt, **kw); torch.cuda.synchronize(); print("run: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); sw.run(par, opt, **kw); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
