"""cfg 2's descriptor variant (Q = 50 k, M = 200 k, D = 981, SAD): the certified fast path against the
exhaustive fp64 kernel of the same library (pcreg_debug_set("match_exact", 1)), pair lists bit for bit, with timings."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
Q, M, D = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (50000, 200000, 980)))
if len(sys.argv) > 4 and sys.argv[4] == "child":
    import pcreg_amd as pc
    from pcreg_amd._lib import check, lib
    exact = len(sys.argv) > 6 and sys.argv[6] == "exact"
    check(lib().pcreg_debug_set(b"match_exact", 1 if exact else 0))
    rng = np.random.default_rng(0)
    dM = rng.poisson(3.0, (M, D)).astype(np.float64)
    dS = rng.poisson(3.0, (Q, D)).astype(np.float64)
    k = Q // 2
    dS[:k] = dM[rng.choice(M, k, replace=False)] + rng.poisson(0.2, (k, D))
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
               MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)
    pc.getMatches(dS[:256], dM[:512], par)
    t0 = time.perf_counter(); m = pc.getMatches(dS, dM, par); dt = time.perf_counter() - t0
    np.save(sys.argv[5], m)
    print(f"{'exact' if exact else 'fast '}: {len(m)} matches, host tier {dt*1e3:.0f} ms", flush=True)
else:
    outs = []
    for mode in ("fast", "exact"):
        out = f"/tmp/desc_big_{mode}.npy"; outs.append(out)
        subprocess.run([sys.executable, __file__, str(Q), str(M), str(D), "child", out, mode], check=True)
    a, b = np.load(outs[0]), np.load(outs[1])
    print("pair lists identical:", a.shape == b.shape and bool((a == b).all()), a.shape)
