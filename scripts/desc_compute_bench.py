"""cfg 4: getSpacialHistogramDescriptors on a synthetic ridge cloud; wrap in rocprofv3 for kernel times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcreg_amd as pc
P, S = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (1000000, 100000)))
rng = np.random.default_rng(0)
ns = 16; per = P // ns
pts = np.vstack([np.column_stack([rng.uniform(0, 100, per), 6 * s + rng.uniform(-1.2, 1.2, per), 10 + 0.1 * rng.uniform(0, 100, per) * np.sin(s) * 0 + 5 * np.sin(s) + rng.normal(0, 0.25, per)]) for s in range(ns)])
kp = np.column_stack([rng.uniform(2, 98, S), 6 * rng.integers(0, ns, S) + rng.uniform(-1.0, 1.0, S), 10 + 5 * np.sin(rng.integers(0, ns, S)) + rng.uniform(-0.5, 0.5, S)])
kp[:, 2] = 10 + 5 * np.sin(np.round(kp[:, 1] / 6).clip(0, ns - 1)) + rng.uniform(-0.5, 0.5, S)
opt = dict(min_pts=500, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)
pc.getSpacialHistogramDescriptors(pts[:20000], kp[:100], opt)
t0 = time.perf_counter(); feat, desc = pc.getSpacialHistogramDescriptors(pts, kp, opt); dt = time.perf_counter() - t0
print(f"P={P} S={S}: {len(feat)} descriptors, host tier {dt*1e3:.0f} ms ({S/dt:.0f} keypoints/s), mean support {desc.sum(axis=1).mean() if len(desc) else 0:.0f} pts", flush=True)
