"""getMatches (D = 980 count descriptors, SAD / SSD) timing through the host tier (includes
H2D/D2H of the descriptor matrices) and kernel-only timing from rocprof if wrapped."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcreg_amd as pc
Q, M, D = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (5000, 20000, 980)))
rng = np.random.default_rng(0)
dM = rng.poisson(3.0, (M, D)).astype(np.float64)
dS = rng.poisson(3.0, (Q, D)).astype(np.float64)
k = min(Q, M) // 2
dS[:k] = dM[rng.choice(M, k, replace=False)] + rng.poisson(0.2, (k, D))
for metric in (sys.argv[4:] or ("SAD", "SSD")):
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
               MatchThreshold=10, MaxRatio=0.99, Metric=metric, Unique=True, VERBOSE=0)
    pc.getMatches(dS[:64], dM[:64], par)
    t0 = time.perf_counter(); m = pc.getMatches(dS, dM, par); dt = time.perf_counter() - t0
    flop = (3 * (D + 1) - 1) * Q * M
    print(f"{metric}: Q={Q} M={M} D={D}: {dt*1e3:.1f} ms host-tier ({len(m)} matches), {Q*M/dt/1e9:.2f} Gpairs/s, {flop/dt/1e12:.2f} TFLOP/s-equivalent (3D-1 flop/pair, forward pass only)", flush=True)
