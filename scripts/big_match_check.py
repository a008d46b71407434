"""match_points (search + threshold + ratio + Unique on the query grid + pair gather) at a size well beyond the
unit tests, against the C oracle's exhaustive version."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pcreg_amd as pc
import oracle.c_oracle as oc
from bench import synth
oc.build()
for Q, M in [(100000, 400000), (60000, 1000000)]:
    model, surf, _ = synth(M, Q)
    t0 = time.perf_counter(); got = pc.match_points(surf, model, 0.25, 0.8, True); t1 = time.perf_counter()
    ref = oc.match_points_f32(surf, model, 0.25, 0.8, True)
    print(f"Q={Q} M={M}: {len(got)} pairs, host tier {1e3*(t1-t0):.1f} ms, equal to the oracle: {np.array_equal(got, ref)}", flush=True)
