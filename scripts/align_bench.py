"""AlignPoints_KNN batched: B supports of n points each through the host tier; wrap in rocprofv3 for the kernel time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcreg_amd as pc
B, n = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (4096, 3000)))
rng = np.random.default_rng(0)
sups = [rng.normal(size=(n, 3)) * [3.0, 1.5, 0.4] + rng.uniform(-50, 50, 3) for _ in range(B)]
pc.AlignPoints_KNN_batched(sups[:8])
t0 = time.perf_counter(); al, co, c, st = pc.AlignPoints_KNN_batched(sups); dt = time.perf_counter() - t0
print(f"B={B} n={n}: host tier {dt*1e3:.1f} ms, {B/dt:.0f} supports/s; algorithmic bytes {B*n*(5*24+24)/1e6:.0f} MB (5 reads + 1 write of 24 B/pt)", flush=True)
