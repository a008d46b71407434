#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: KNN Gpairs/s + RANSAC registrations/s, 50k-point surface vs a
1M-point model.

One "step" = one pass of the hot path over one synthetic registration problem, all inputs resident in HBM:
    top-2 search of every surface point over the model shard   (knn_candidates_f16_pipe_kernel + exact re-rank)
    [N > 1: one all_gather of the per-rank top-2 lists + merge] (RCCL over xGMI)
    threshold + ratio test + Unique back-check (query grid) + pair gather
    [N > 1: one integer all_reduce of the candidate table]
    RANSAC (minPtNum 3, iterNum 1e4, thDist 0.3, thInlrRatio 0.08, REFINE; completeExperimentFast.m:169-173)
    [N > 1: hypotheses split over the ranks, one all_gather of the partial results]
`value` = surface points x model points (all ranks) x steps / wall time: end-to-end Gpairs/s including the filters
and RANSAC; 1000 / ms_per_step is registrations/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model-points M_TOTAL] [--surface-points Q]

N > 1 (launched by torch.distributed.run, one rank per GPU) is STRONG scaling: the model has --model-points rows in
TOTAL (default 1M, the metric's) and is sharded by rows, M_TOTAL / N per GPU.  The same line then also carries, as
extra objects, cfg 3's fixed 2M-point model sharded N ways (`cfg3_model_2M`) and the weak-scaling run with 1M rows
per GPU (`weak_1M_per_gpu`), and cfg 5's crop batch dealt over the ranks (`cfg5_batch`).  At N = 1 the line also
carries the other BASELINE configs at their real sizes, each with its own ms / roofline / cpu_baseline
(`extras`: getMatches D = 981 at cfg 2, descriptors at cfg 4, AlignPoints_KNN batched, RANSAC alone at cfg 1).
Nothing here imports oracle/ on the GPU path: the oracle is only the `cpu_baseline` leg.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BBOX = np.array([101.0, 56.0, 99.0])          # CT crop extent in mm
RANSAC_COEF = dict(minPtNum=3, iterNum=10000, thDist=0.3, thInlrRatio=0.08, REFINE=True)
MATCH_THR_ABS, MATCH_RATIO = 0.25, 0.8         # squared-distance threshold (0.5 mm), ratio test
FLOP_PER_PAIR = 8                              # SURVEY 8d's algorithmic figure: 3 sub + 3 mul + 2 add
FLOP_PER_PAIR_EXECUTED = 32                    # what one v_mfma_f32_32x32x16_f16 spends per pair: 16 k-slots x 2
PEAK_FP32_TFLOPS = 157.3                       # fp32 vector peak == fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_FP64_TFLOPS = 78.6                        # fp64 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0                  # dense f16/bf16 matrix-core peak
PEAK_HBM_GBPS = 8000.0


def eul2rotm_zyx(e) -> np.ndarray:
    """Rotation of ZYX Euler angles (the convention of testRANSAC.m:17); data generation only."""
    cz, sz, cy, sy, cx, sx = np.cos(e[0]), np.sin(e[0]), np.cos(e[1]), np.sin(e[1]), np.cos(e[2]), np.sin(e[2])
    return np.array([[cy * cz, sy * sx * cz - sz * cx, sy * cx * cz + sz * sx],
                     [cy * sz, sy * sx * sz + cz * cx, sy * cx * sz - cz * sx],
                     [-sy, cy * sx, cy * cx]])


def synth(M_total: int, Q: int, seed: int = 10):
    """Model ~U(bbox) fp32; surface = the Q model points nearest to a random centre,
    moved by a small rigid motion (ICP-like misalignment) + N(0, 0.05^2) noise."""
    rng = np.random.default_rng(seed)
    model = (rng.random((M_total, 3), dtype=np.float32) * BBOX.astype(np.float32))
    centre = (BBOX * np.array([0.45, 0.55, 0.5])).astype(np.float32)
    d2 = ((model - centre) ** 2).sum(axis=1)
    crop = np.argpartition(d2, Q - 1)[:Q]
    crop.sort()
    R = eul2rotm_zyx([0.010, -0.008, 0.012]).astype(np.float64)
    t = np.array([0.15, -0.10, 0.20])
    rng2 = np.random.default_rng(seed + 1)
    c = model[crop].astype(np.float64)
    surf = ((c - centre) @ R + centre + t + rng2.normal(0, 0.05, c.shape)).astype(np.float32)
    return model, surf, crop


def make_crop(model: np.ndarray, Q: int, c: int):
    """cfg 5 crop c: the Q model points nearest to a random centre (seed 100 + c), small rigid motion, noise."""
    rng = np.random.default_rng(100 + c)
    centre = (BBOX * rng.uniform(0.3, 0.7, 3)).astype(np.float32)
    d2 = ((model - centre) ** 2).sum(axis=1)
    crop = np.sort(np.argpartition(d2, Q - 1)[:Q])
    R = eul2rotm_zyx(rng.uniform(-0.012, 0.012, 3)); t = rng.uniform(-0.2, 0.2, 3)
    pts = model[crop].astype(np.float64)
    return ((pts - centre) @ R + centre + t + rng.normal(0, 0.05, pts.shape)).astype(np.float32)


# ---- CPU baseline (the oracle's C restatement; rank 0, N = 1 only) -------------------------------------------------

def kernel_source_hash(files) -> str:
    """sha256 over the named kernel sources (paths relative to the repo root), in the given order."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores() -> int:
    """Threads this job may really use: the affinity mask, cut to the cgroup CPU quota when there is one (the 1-GPU box
    shows 256 CPUs and grants a 16-CPU share; 256 OpenMP threads on that share run slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    env = os.environ.get("PCREG_BENCH_CORES")
    return max(1, min(n, int(env))) if env else min(n, 64)


def cpu_baseline(model: np.ndarray, surf: np.ndarray) -> dict:
    """the restatement of the MATLAB path on the host cores -- KNN on 1 thread and on all
    cores, RANSAC at the bench's own iterNum on all cores -- each on a bounded sample."""
    from oracle import c_oracle
    c_oracle.build()
    cores = host_cores()
    M = model.shape[0]

    def knn_rate(threads: int, budget_s: float):
        t0 = time.perf_counter()
        c_oracle.knn2_points_f32(surf[:256 * threads], model, nthreads=threads)
        rate = 256 * threads * M / max(time.perf_counter() - t0, 1e-6)
        qs = int(min(surf.shape[0], max(256 * threads, rate * budget_s / M)))
        t0 = time.perf_counter()
        c_oracle.knn2_points_f32(surf[:qs], model, nthreads=threads)
        dt = time.perf_counter() - t0
        return qs * M / dt / 1e9, qs, dt

    g1, q1, t1 = knn_rate(1, 4.0)
    ga, qa, ta = knn_rate(cores, 5.0)
    # RANSAC: the bench's problem (n = 32 k pairs, iterNum = 1e4, REFINE), hypotheses dealt to `cores` threads
    rng = np.random.default_rng(0)
    n = 32000
    p2 = rng.uniform(0, 40, (n, 3)); p1 = p2 + rng.normal(0, 0.05, p2.shape)
    share = (RANSAC_COEF["iterNum"] + cores - 1) // cores
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:          # ctypes releases the GIL inside the C call
        list(ex.map(lambda k: c_oracle.ransac(p1, p2, dict(RANSAC_COEF, iterNum=share), seed=1 + k), range(cores)))
    dr = time.perf_counter() - t0
    return {"value": round(ga, 3), "unit": "Gpairs/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle/pcreg_oracle.c knn2 (OpenMP, {cores} threads): first {qa} surface points vs all {M} model points, {ta:.1f} s",
            "knn_1thread_gpairs_per_s": round(g1, 3),
            "knn_1thread_sample": f"first {q1} surface points vs all {M} model points, {t1:.1f} s",
            "ransac_registrations_per_s_all_cores": round(1.0 / dr, 3),
            "ransac_sample": f"n={n}, iterNum={RANSAC_COEF['iterNum']} (REFINE) as {cores} shares of {share} hypotheses on {cores} threads, {dr:.2f} s "
                             f"(the reference itself ran 4 parfor workers, completeExperiment.m:135)"}


# ---- one timed workload -------------------------------------------------------------------------------------------

class Ctx:
    def __init__(self, rank, world, dev, collective):
        self.rank, self.world, self.dev, self.collective = rank, world, dev, collective


def run_registration(ctx: Ctx, M_total: int, Q: int, steps: int, warmup: int, time_kernel: bool, in_flight: int = 1,
                     coef: dict | None = None) -> dict:
    """The step described in the module docstring on a model of M_total rows sharded over ctx.world ranks.  in_flight > 1: that many
    registrations in flight per rank (pcreg_amd/pipelined.py: one lane = one HIP stream + its own buffers, surfaces dealt round-robin;
    every lane issues its collectives in host program order, the same on every rank)."""
    import torch
    import torch.distributed as dist
    from pcreg_amd.device import PreparedModel, RegistrationPipeline, soa
    from pcreg_amd.pipelined import PipelinedRegistration
    from pcreg_amd._lib import lib
    coef = coef or RANSAC_COEF
    model, surf, _ = synth(M_total, Q)
    per = (M_total + ctx.world - 1) // ctx.world
    m_lo = ctx.rank * per
    shard = model[m_lo:m_lo + per]
    model_t = soa(torch.from_numpy(shard).to(ctx.dev))
    q_soa = soa(torch.from_numpy(surf).to(ctx.dev))
    # the model shard is prepared ONCE (box, matrix-core operand tiles, seeding grid) -- one model, many surfaces is the
    # reference's shape (completeExperimentFast.m:131-149); its cost is reported beside the step, not inside it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model_soa = PreparedModel(model_t)
    torch.cuda.synchronize()
    prepare_ms = (time.perf_counter() - t0) * 1e3
    piped = in_flight > 1
    if piped:
        pr = PipelinedRegistration(Q, shard.shape[0], lanes=in_flight, m_lo=m_lo, M_total=M_total, device=ctx.dev)
        pipe = pr.lanes[0]
    else:
        pipe = RegistrationPipeline(Q, shard.shape[0], m_lo=m_lo, M_total=M_total, device=ctx.dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def step(k):
        if piped:
            pr.submit(q_soa, model_soa, MATCH_THR_ABS, MATCH_RATIO, coef, unique=True, seed=7, inputs_ready=True)   # resident before the loop
            return
        if k is not None:
            ev[k][0].record()
        pipe.search_local(q_soa, model_soa)
        if k is not None:
            ev[k][1].record()
        pipe.match_after_search(q_soa, model_soa, MATCH_THR_ABS, MATCH_RATIO, unique=True)
        pipe.ransac_sharded(coef, seed=7)               # one rank: plain ransac(); N ranks: hypotheses split N ways

    for _ in range(max(warmup, in_flight if piped else 0)):
        step(None)
    if time_kernel:
        lib().pcreg_dev_search_kernel_timing(1)       # HIP events around the dominant kernel, on its launch stream
    if ctx.collective:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    if ctx.collective:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if ctx.collective:
        tt = torch.tensor([elapsed], device=ctx.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    search_ms = None if piped else float(np.mean([a.elapsed_time(b) for a, b in ev]))           # the whole search call
    kernel_ms, launches = None, 0
    if time_kernel:
        kms, kn = C.c_float(0.0), C.c_int(0)
        lib().pcreg_dev_search_kernel_ms(C.byref(kms), C.byref(kn))
        lib().pcreg_dev_search_kernel_timing(0)
        if kn.value != steps:                                    # never price the roofline from another duration
            raise RuntimeError(f"the dominant kernel was timed {kn.value} times in {steps} steps: its HIP events are missing")
        kernel_ms, launches = float(kms.value), int(kn.value)
    if piped:
        allres = pr.results()
        res = allres[-1]
        n_pairs = res["n_pairs"]
    else:
        res = pipe.fetch_result()
        n_pairs = int(pipe.n_pairs.item())
    out = {"ms_per_step": elapsed / steps * 1e3, "value": float(Q) * float(M_total) * steps / elapsed / 1e9,
           "search_call_ms": search_ms, "kernel_ms": kernel_ms, "launches_timed": launches, "rows_per_gpu": int(shard.shape[0]),
           "model_prepare_ms": prepare_ms, "in_flight": in_flight,
           "ransac": {"n_pairs": n_pairs, "max_inliers": res["maxInliers"], "num_success": res["numSuccess"],
                      "failed": res["failed"]}, "_model": model, "_surf": surf}
    if piped:
        del pr
    del pipe, model_soa, model_t, q_soa
    torch.cuda.empty_cache()
    return out


def extra_two_in_flight(ctx: Ctx, Q: int, steps: int) -> dict:
    """throughput with TWO registrations in flight per GPU (two HIP streams) against one, (a) at the headline's
    shape and (b) at the shape one rank of an 8-GPU run sees: a 125 k-row shard and 1/8 of the hypotheses -- its collectives (three
    latency-sized exchanges) excluded.  The launch-sized kernels of a step do not shrink with the shard; a second lane fills them."""
    res = {}
    for key, M, coef in (("model_1M", 1_000_000, RANSAC_COEF), ("rank_of_8_emulated_125k", 125_000, dict(RANSAC_COEF, iterNum=RANSAC_COEF["iterNum"] // 8))):
        one = run_registration(ctx, M, Q, steps, 2, time_kernel=True, in_flight=1, coef=coef)
        two = run_registration(ctx, M, Q, steps, 2, time_kernel=False, in_flight=2, coef=coef)
        same = one["ransac"] == two["ransac"]
        res[key] = {"one_in_flight_ms": round(one["ms_per_step"], 4), "two_in_flight_ms": round(two["ms_per_step"], 4),
                    "speedup": round(one["ms_per_step"] / two["ms_per_step"], 3), "same_result": bool(same), "iterNum": coef["iterNum"],
                    "knn_kernel_ms_one_in_flight": round(one["kernel_ms"], 4)}
    res["note"] = "125k: one rank of N = 8 (its shard, 1/8 of the hypotheses), collectives excluded"
    return res


def run_batch_cfg5(ctx: Ctx, n_crops: int, M: int, Q: int) -> dict:
    """cfg 5: n_crops surface crops against ONE resident model, crops dealt to the ranks (replicas, no data-path
    collective), two HIP streams per GPU; end-to-end registrations/s over all ranks."""
    import torch
    import torch.distributed as dist
    from pcreg_amd.batch import BatchRegistration, crops_of_rank
    from pcreg_amd.device import soa
    rng = np.random.default_rng(10)
    model = rng.random((M, 3), dtype=np.float32) * BBOX.astype(np.float32)
    model_soa = soa(torch.from_numpy(model).to(ctx.dev))
    surfaces = [None] * n_crops
    for c in crops_of_rank(n_crops, ctx.rank, ctx.world):
        surfaces[c] = soa(torch.from_numpy(make_crop(model, Q, c)).to(ctx.dev))
    br = BatchRegistration(model_soa, Q, n_streams=2, device=ctx.dev)
    br.run(surfaces[:min(n_crops, 2 * ctx.world)], MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF, seed=7, gather=False)   # warm-up
    best, res = None, None
    for _ in range(2):
        if ctx.collective:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = br.run(surfaces, MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF, seed=7, gather=True)
        torch.cuda.synchronize()
        if ctx.collective:
            dist.barrier()
        dt = time.perf_counter() - t0
        if ctx.collective:
            tt = torch.tensor([dt], device=ctx.dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        best = dt if best is None else min(best, dt)
    out = {"workload": f"{n_crops} crops x {Q} surface pts vs one {M}-pt model, crops dealt to {ctx.world} GPU(s), 2 streams each, "
                       f"match + RANSAC(3,1e4,0.3,0.08,REFINE) per crop",
           "ms": round(best * 1e3, 3), "registrations_per_s": round(n_crops / best, 2), "failed": int(sum(r["failed"] for r in res)),
           "min_inliers": int(min(r["n_inliers"] for r in res)), "scaling": "strong (fixed batch)", "n_gpus": ctx.world,
           "roofline": {"note": "composite of the headline step (its dominant kernel is the headline's)"}}
    del br, model_soa, surfaces
    torch.cuda.empty_cache()
    return out


# ---- the other BASELINE configs at their real sizes (N = 1, after the headline, outside its timed region) ----------

def _ev_ms(fn, reps: int = 3, warm: int = 1) -> float:
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(min(ts))


def extra_get_matches(dev, with_cpu: bool) -> dict:
    """cfg 2's descriptor variant: getMatches, D = 981 (980 counts + the UNNORMALIZE column), Q = 50 k x M = 200 k, SAD,
    MatchThreshold 10, MaxRatio 0.99, Unique (completeExperimentFast.m:78-85), device tier."""
    import torch
    from pcreg_amd.device import DescriptorPipeline
    Q, M, D = 50_000, 200_000, 980
    g = torch.Generator(device=dev); g.manual_seed(20)
    lam = torch.full((1, D), 3000.0 / D, device=dev, dtype=torch.float32)
    dM = torch.poisson(lam.expand(M, D), generator=g).to(torch.float64)
    pick = torch.randperm(M, device=dev, generator=g)[:Q]
    dS = (dM[pick] + torch.poisson(torch.full((Q, D), 0.15, device=dev), generator=g).to(torch.float64)).contiguous()
    par = dict(Method="Approximate", Metric="SAD", MatchThreshold=10, MaxRatio=0.99, Unique=True, UNNORMALIZE=True, norm_factor=2.0,
               CHANGE_METRIC=True, metric_factor=0.6, VERBOSE=0)
    dp = DescriptorPipeline(dev)
    ms = _ev_ms(lambda: dp.match(dS, Q, dM, M, par), reps=2)
    n_pairs = int(dp.n_pairs.item())
    flops = (3.0 * (D + 1) - 1.0) * Q * M                       # (3D - 1) per pair, D = 981
    out = {"workload": f"getMatches SAD, {Q} x {M} descriptors, D = {D + 1} (UNNORMALIZE column), power 0.6, threshold 10 %, ratio 0.99, Unique",
           "ms": round(ms, 2), "gpairs_per_s": round(Q * M / ms / 1e6, 1), "pairs_found": n_pairs,
           "roofline": {"bound": "valu", "achieved": round(flops / ms / 1e9, 1), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flops / ms / 1e9 / PEAK_FP32_TFLOPS, 4), "traffic": None,
                        "note": "(3D-1) flop/pair vs fp32 vector peak; v_sad_u16 + exact fp64 re-rank"}}
    if with_cpu:
        from oracle import c_oracle
        cores = host_cores()
        qs, msub = 1024, 50_000
        a, b = dS[:qs].cpu().numpy(), dM[:msub].cpu().numpy()
        t0 = time.perf_counter(); c_oracle.getMatches(a, b, par, nthreads=cores); dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(qs * msub / dt / 1e9, 4), "unit": "Gpairs/s", "cores": cores, "kind": "port",
                               "sample": f"oracle getMatches (OpenMP, {cores} threads) on a {qs} x {msub} sub-problem, {dt:.1f} s"}
    del dS, dM, dp
    torch.cuda.empty_cache()
    return out


def _ridge_cloud(P: int, S: int, seed: int = 0):
    """cfg 4's cloud: 16 noisy sheets through the box so that R = 3.5 supports hold 500-6000 points; keypoints on the sheets."""
    rng = np.random.default_rng(seed)
    ns = 16; per = P // ns
    pts = np.vstack([np.column_stack([rng.uniform(0, 100, per), 6 * s + rng.uniform(-1.2, 1.2, per),
                                      10 + 5 * np.sin(s) + rng.normal(0, 0.25, per)]) for s in range(ns)])
    sheet = rng.integers(0, ns, S)
    kp = np.column_stack([rng.uniform(2, 98, S), 6 * sheet + rng.uniform(-1.0, 1.0, S), 10 + 5 * np.sin(sheet) + rng.uniform(-0.5, 0.5, S)])
    return pts, kp


def extra_descriptors(dev, with_cpu: bool) -> dict:
    """cfg 4: getSpacialHistogramDescriptors on 1 M keypoints of a 1 M-point cloud (completeExperimentFast.m:299-304 options)."""
    import torch
    from pcreg_amd.device import DescriptorPipeline
    P = S = 1_000_000
    pts, kp = _ridge_cloud(P, S)
    opt = dict(min_pts=500, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)
    tp = torch.from_numpy(np.ascontiguousarray(pts.T)).to(dev); tk = torch.from_numpy(np.ascontiguousarray(kp.T)).to(dev)
    dp = DescriptorPipeline(dev)
    dp.describe(tp[:, :50_000].contiguous(), tk[:, :1000].contiguous(), opt, compact=True)
    V = [0]
    def run():
        V[0] = dp.describe(tp, tk, opt, compact=True)[2]
    ms = _ev_ms(run, reps=2, warm=0)
    # what a resident pipeline keeps: uint16 rows written once in keypoint order + the survivor list (device tier); the
    # MATLAB-shaped double rows (8 x 980 B per keypoint) only exist at the host boundary
    alg_bytes = 24.0 * (P + S) + (2.0 * 980 + 24 + 4) * V[0]
    alg_bytes_f64 = 24.0 * (P + S) + 8.0 * 983 * V[0]
    out = {"workload": f"getSpacialHistogramDescriptors, {S} keypoints on a {P}-point cloud, R 3.5, min/max 500/6000, k 0.85, ALIGN_POINTS; "
                       f"uint16 rows written once + survivor list (pcreg_dev_spatial_histogram_descriptors_rows_u16)",
           "ms": round(ms, 2), "keypoints_per_s": round(S / ms * 1e3, 0), "descriptors": int(V[0]),
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / ms / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(alg_bytes / ms / 1e6 / PEAK_HBM_GBPS, 4), "traffic": None,
                        "algorithmic_bytes": int(alg_bytes), "algorithmic_bytes_matlab_double_rows": int(alg_bytes_f64),
                        "note": "cloud + keypoints read, V x (980 u16 + feat + index) written; not HBM-bound"}}
    if with_cpu:
        from oracle import c_oracle
        cores = host_cores()
        ks = 512 * cores
        t0 = time.perf_counter(); c_oracle.getSpacialHistogramDescriptors(pts, kp[:ks], opt, nthreads=cores); dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ks / dt, 2), "unit": "keypoints/s", "cores": cores, "kind": "port",
                               "sample": f"oracle getSpacialHistogramDescriptors (two brute-force scans per keypoint, OpenMP {cores} threads): {ks} keypoints on the {P}-point cloud, {dt:.1f} s"}
    del tp, tk, dp
    torch.cuda.empty_cache()
    return out


def extra_align(dev, with_cpu: bool) -> dict:
    """AlignPoints_KNN batched: 4096 supports of 3000 points (the R = 3.5 support size of completeExperimentFast.m:300-302)."""
    import torch
    from pcreg_amd._lib import check, lib
    B, n = 4096, 3000
    rng = np.random.default_rng(3)
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    sup = (rng.normal(size=(B * n, 3)) * np.array([3.0, 1.5, 0.4])) @ A + rng.uniform(-50, 50, 3)
    pts = torch.from_numpy(np.ascontiguousarray(sup.T)).to(dev)
    off = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    al = torch.empty_like(pts); co = torch.empty(9 * B, dtype=torch.float64, device=dev); c = torch.empty(3 * B, dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    L = lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    def run():
        check(L.pcreg_dev_align_points_knn_batched(p(pts), B * n, B * n, p(off), B, n, 0, 0, p(al), p(co), p(c), p(st),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ms = _ev_ms(run, reps=5)
    alg_bytes = 48.0 * B * n
    out = {"workload": f"AlignPoints_KNN, {B} supports x {n} points, one launch", "ms": round(ms, 4), "supports_per_s": round(B / ms * 1e3, 0),
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / ms / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(alg_bytes / ms / 1e6 / PEAK_HBM_GBPS, 4), "traffic": None,
                        "note": "algorithmic bytes = 24 B read + 24 B written per point (SURVEY 8d)"}}
    if with_cpu:
        from oracle import c_oracle
        k = 4096
        t0 = time.perf_counter()
        for b in range(k):
            c_oracle.AlignPoints_KNN(sup[b * n:(b + 1) * n])
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(k / dt, 1), "unit": "supports/s", "cores": 1, "kind": "port",
                               "sample": f"oracle AlignPoints_KNN (full stable sort + 3x3 eig), {k} supports of {n} points, 1 thread, {dt:.2f} s"}
    return out


def extra_ransac_cfg1(dev, with_cpu: bool) -> dict:
    """cfg 1 (testRANSAC.m): n = 1000 correspondences, getInliersRANSAC.m:17-31 options (3, 2e4, 0.5, 0.1, REFINE)."""
    import torch
    from pcreg_amd.device import RegistrationPipeline
    rng = np.random.default_rng(1)
    n = 1000
    pts = rng.uniform([-3, -2, 0], [3, 2, 3], (n, 3))
    R = eul2rotm_zyx([1.5, -1.2, 0.8]); t = np.array([1.0, 2.0, 3.0])
    loc1S = pts @ R + t
    loc1M = pts + np.random.default_rng(2).normal(0, 0.1, pts.shape)
    coef = dict(minPtNum=3, iterNum=20000, thInlrRatio=0.5, thDist=0.1, REFINE=True)
    pipe = RegistrationPipeline(n, 16, device=dev)
    p1 = torch.from_numpy(np.ascontiguousarray(loc1M.T)).to(dev); p2 = torch.from_numpy(np.ascontiguousarray(loc1S.T)).to(dev)
    nd = torch.tensor([n], dtype=torch.int32, device=dev)
    ms = _ev_ms(lambda: pipe.ransac(coef, seed=3, n_dev=nd, pts1=p1, pts2=p2), reps=5)
    res = pipe.fetch_result()
    flops = coef["iterNum"] * n * 76.0                          # score + refit-accumulate + rescore
    out = {"workload": f"ransac, n = {n}, iterNum = {coef['iterNum']}, thDist 0.1, thInlrRatio 0.5, REFINE (testRANSAC.m / getInliersRANSAC.m:17-31)",
           "ms": round(ms, 4), "registrations_per_s": round(1e3 / ms, 1), "max_inliers": res["maxInliers"], "failed": res["failed"],
           "roofline": {"bound": "valu-fp64", "achieved": round(flops / ms / 1e9, 2), "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flops / ms / 1e9 / PEAK_FP64_TFLOPS, 4), "traffic": None,
                        "note": "one small registration is latency-bound; see ransac_cfg1_batched"}}
    if with_cpu:
        from oracle import c_oracle
        t0 = time.perf_counter()
        for k in range(8):
            c_oracle.ransac(loc1M, loc1S, coef, seed=3 + k)
        dt = (time.perf_counter() - t0) / 8
        out["cpu_baseline"] = {"value": round(1.0 / dt, 3), "unit": "registrations/s", "cores": 1, "kind": "port",
                               "sample": f"oracle ransac, n = {n}, iterNum = {coef['iterNum']}, 8 registrations on 1 thread, {dt:.2f} s each"}
    return out


def extra_ransac_cfg1_batched(dev, with_cpu: bool) -> dict:
    """cfg 1 as a batch: 256 independent registrations of n = 1000 in ONE launch (pcreg_dev_ransac_batched, the parfor of
    completeExperimentFast.m:201-216); every registration has its own noise and its own sampler seed."""
    import torch
    from pcreg_amd._lib import DevRansacResult, RansacOpts, check, lib
    B, n, it = 256, 1000, 20000
    rng = np.random.default_rng(1)
    pts = rng.uniform([-3, -2, 0], [3, 2, 3], (n, 3))
    R = eul2rotm_zyx([1.5, -1.2, 0.8]); t = np.array([1.0, 2.0, 3.0])
    loc1S = pts @ R + t
    p1h = np.concatenate([pts + np.random.default_rng(100 + b).normal(0, 0.1, pts.shape) for b in range(B)])      # [B*n, 3]
    p2h = np.tile(loc1S, (B, 1))
    ld = B * n
    p1 = torch.from_numpy(np.ascontiguousarray(p1h.T)).to(dev); p2 = torch.from_numpy(np.ascontiguousarray(p2h.T)).to(dev)
    off = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    L = lib()
    o = RansacOpts(3, it, 0.1, 0.5, 1, 0, 3)
    rs = C.sizeof(DevRansacResult)
    res = torch.zeros((B, rs), dtype=torch.uint8, device=dev); inl = torch.empty(ld, dtype=torch.int32, device=dev)
    ws = torch.empty(max(L.pcreg_dev_ransac_batched_workspace(n, it, B), 256), dtype=torch.uint8, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    def run():
        check(L.pcreg_dev_ransac_batched(p(p1), p(p2), ld, p(off), B, n, C.byref(o), p(res), p(inl), p(ws), C.c_size_t(ws.numel()),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ms = _ev_ms(run, reps=3)
    raw = res.cpu().numpy()
    rr = [DevRansacResult.from_buffer_copy(raw[b].tobytes()) for b in range(B)]
    flops = float(B) * it * n * 76.0
    # What the launch executes per (hypothesis, correspondence) now (ransac_hyp32_kernel, every hypothesis refits on this data): two
    # screened scoring passes = 2 x 15 packed-fp32 operations (78.6 T lane-op/s) + 2 x 2 compares (39.3 T/s), and 15 fp64 operations
    # of the refit sums (39.3 T/s); the time those take at the issue peaks is the floor the kernel is measured against.
    pairs = float(B) * it * n
    floor_ms = pairs * (30.0 / 78.6e12 + 4.0 / 39.3e12 + 15.0 / 39.3e12) * 1e3
    return {"workload": f"{B} registrations x (n = {n}, iterNum = {it}, thDist 0.1, thInlrRatio 0.5, REFINE) in one launch (pcreg_dev_ransac_batched)",
            "ms": round(ms, 3), "registrations_per_s": round(B / ms * 1e3, 1), "failed": int(sum(r.failed for r in rr)),
            "max_inliers_min": int(min(r.max_inliers for r in rr)),
            "roofline": {"bound": "valu", "achieved": round(flops / ms / 1e9, 2), "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / ms / 1e9 / PEAK_FP64_TFLOPS, 4), "traffic": None,
                         "executed": {"per_pair": "30 packed-fp32 + 4 compares + 15 fp64 (two screened passes, masked refit sums)",
                                      "floor_ms_at_issue_peaks": round(floor_ms, 3), "frac_of_floor": round(floor_ms / ms, 4)},
                         "note": "76 flop per (hypothesis, point) of SURVEY 8d vs fp64 peak; `executed`: issue floor"}}


def extra_sweep(dev, with_cpu: bool) -> dict:
    """The sphere sweep of completeExperimentFast.m:46-224 at the reference's shape: a 60 k-keypoint model, ~329 valid spheres of
    ~1533 descriptors (R_desc 9, spacing 5, >= 1400), a 2000-keypoint surface, D = 980, getMatches per sphere, RANSAC on the
    spheres with > 170 putative matches.  SphereSweep.run: one segmented launch chain, two host synchronisations."""
    import torch
    from pcreg_amd.sweep import SphereSweep
    VM, VS, D = 60000, 2000, 980
    rng = np.random.default_rng(0)
    featM = rng.uniform([0, 0, 0], [60, 50, 40], (VM, 3))
    g = torch.Generator(device=dev); g.manual_seed(1)
    descM = torch.poisson(torch.full((VM, D), 3.0, device=dev), generator=g).to(torch.float64)
    near = np.argsort(np.linalg.norm(featM - np.array([31.0, 24.0, 19.0]), axis=1))[:VS]
    c, s_ = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
    featS = featM[near] @ R.T + np.array([2.0, -1.0, 0.5]) + rng.normal(0, 0.02, (VS, 3))
    descS = (descM[torch.from_numpy(near).to(dev)] + torch.poisson(torch.full((VS, D), 0.15, device=dev), generator=g).to(torch.float64)).contiguous()
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate", MatchThreshold=10, MaxRatio=0.99,
               Metric="SAD", Unique=True, VERBOSE=0)
    opt = dict(minPtNum=3, iterNum=10000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
    sw = SphereSweep(featM, descM, featS, descS, device=dev)
    out = sw.run(par, opt, **kw)                                     # warm-up (allocations)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = sw.run(par, opt, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ms = min(ts) * 1e3
    S = len(out["centres"])
    union = int(len(np.unique(np.concatenate(out["model_rows"])))) if S else 0
    pairs_dedup = float(VS) * float(union)                       # every (surface row, model row that lies in SOME sphere) pair once
    pairs_per_sphere_calls = float(VS) * float(out["num_desc"].sum())
    flops = (3.0 * (D + 1) - 1.0) * pairs_dedup
    res = {"workload": f"sphere sweep: {S} valid spheres of a {VM}-keypoint model ({int(out['num_desc'].mean())} descriptors per sphere on average), surface {VS} "
                       f"keypoints, D {D}, getMatches per sphere (SAD, power 0.6, 10 %, 0.99, Unique), {len(out['trial'])} trial spheres x RANSAC(3,1e4,0.3,0.08,REFINE)",
           "ms": round(ms, 2), "spheres_per_s": round(S / ms * 1e3, 1), "host_syncs": 1, "trial_spheres": int(len(out["trial"])),
           "model_prepared_once": "per model (and options / sphere parameters), like the search's prepared model: the spheres (centres, row lists: the script's "
                                  "first synchronisation) and the model set's powered rows (0.35 ms); a timed sweep matches and registers one surface",
           "registered": int(sum(t is not None for t in out["transforms"])),
           "roofline": {"bound": "valu", "achieved": round(flops / ms / 1e9, 1), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flops / ms / 1e9 / PEAK_FP32_TFLOPS, 4), "traffic": None, "dedup_model_rows": union,
                        "sphere_overlap": round(pairs_per_sphere_calls / max(pairs_dedup, 1.0), 2),
                        "note": "(3D-1) flop per de-duplicated pair; kernel split: docs/BENCH_NOTES.md"}}
    try:
        # the same sweep through the HOST tier (pcreg_sphere_counts + pcreg_sphere_sweep: what matlab/sphereSweep.m calls), keypoints
        # as pageable host arrays, the two descriptor sets resident (pcreg_desc_set_create, not timed: once per model / surface)
        import pcreg_amd as pc
        centres_all = sw.sphere_centres(kw["d_spheres"])
        with pc.DescSet(np.asfortranarray(descS.cpu().numpy())) as hS, pc.DescSet(np.asfortranarray(descM.cpu().numpy())) as hM:
            fMf, fSf = np.asfortranarray(featM), np.asfortranarray(featS)
            def host_sweep():
                cnt = pc.sphereCounts(fMf, centres_all, kw["R_desc"])
                keep = cnt >= kw["min_pts"]
                return pc.sphereSweep(hS, hM, fSf, fMf, centres_all[keep], cnt[keep], kw["R_desc"], par, kw["putative_thresh"], opt, seed=kw["seed"])
            h = host_sweep()
            th = []
            for _ in range(3):
                t0 = time.perf_counter(); h = host_sweep(); th.append(time.perf_counter() - t0)
            # the model side in a handle (pcreg_sphere_model_create: once per model), then one call per surface
            cnt = pc.sphereCounts(fMf, centres_all, kw["R_desc"]); keep = cnt >= kw["min_pts"]
            t0 = time.perf_counter(); sm = pc.SphereModel(hM, fMf, centres_all[keep], cnt[keep], kw["R_desc"]); t_model = time.perf_counter() - t0
            try:
                hm = pc.sphereSweepOnModel(sm, hS, fSf, par, kw["putative_thresh"], opt, seed=kw["seed"])
                tm = []
                for _ in range(3):
                    t0 = time.perf_counter(); hm = pc.sphereSweepOnModel(sm, hS, fSf, par, kw["putative_thresh"], opt, seed=kw["seed"]); tm.append(time.perf_counter() - t0)
            finally:
                sm.close()
        same_m = (np.array_equal(hm["trial"], out["trial"]) and np.array_equal(hm["num_putative"], out["num_putative"]) and
                  all(np.array_equal(a, b) for a, b in zip(hm["matches"], out["matches"])) and
                  all((a is None) == (b is None) and (a is None or np.array_equal(a, b)) for a, b in zip(hm["transforms"], out["transforms"])))
        same = (np.array_equal(h["trial"], out["trial"]) and np.array_equal(h["num_putative"], out["num_putative"]) and
                all(np.array_equal(a, b) for a, b in zip(h["matches"], out["matches"])) and
                all((a is None) == (b is None) and (a is None or np.array_equal(a, b)) for a, b in zip(h["transforms"], out["transforms"])))
        res["host_tier"] = {"ms": round(min(th) * 1e3, 2), "same_as_device_driver": bool(same),
                            "note": "pcreg_sphere_counts + pcreg_sphere_sweep, descriptor sets resident, keypoints + results over PCIe",
                            "on_model_ms": round(min(tm) * 1e3, 2), "model_create_ms": round(t_model * 1e3, 2), "on_model_same_as_device_driver": bool(same_m),
                            "on_model_note": "pcreg_sphere_model_create once per model, then pcreg_sphere_sweep_on_model per surface (matlab/sphereSweepModel.m + sphereSweepOn.m)"}
    except Exception as e:
        res["host_tier"] = {"error": f"{type(e).__name__}: {e}"}
    if with_cpu:
        from oracle import c_oracle
        import oracle.pcreg_oracle as opy
        cores = host_cores()
        fm, dm, ds = featM, descM.cpu().numpy(), descS.cpu().numpy()
        pick = sorted({0, S // 2, S - 1, int(out["trial"][0]) if len(out["trial"]) else 0})
        t0 = time.perf_counter(); same = True; n_r = 0
        for i in pick:
            idx = np.nonzero(opy.getDescriptorMask(fm, out["centres"][i], kw["R_desc"], 0.0))[0]
            m = c_oracle.getMatches(ds, dm[idx], par, nthreads=cores)
            same = same and np.array_equal(m, out["matches"][i]) and np.array_equal(idx, out["model_rows"][i])
            if len(m) > kw["putative_thresh"]:
                c_oracle.ransac(featS[m[:, 0].astype(np.int64) - 1], fm[idx][m[:, 1].astype(np.int64) - 1], opt, seed=0); n_r += 1
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(len(pick) / dt, 3), "unit": "spheres/s", "cores": cores, "kind": "port",
                               "sample": f"oracle: getDescriptorMask + getMatches (OpenMP, {cores} threads) on {len(pick)} of the {S} spheres, ransac (1 thread) on the "
                                         f"{n_r} of them above the putative threshold, {dt:.1f} s; their matches equal the GPU's: {bool(same)}"}
        res["same_as_oracle_on_sample"] = bool(same)
    del sw, descM, descS
    torch.cuda.empty_cache()
    return res


def _ridge_volume(P: int, S: int, seed: int = 5):
    """A 60 x 50 x 40 mm block of 48 undulating ridges (strips 2.4 mm wide, 6 mm apart in y, 6.5 mm in z): R = 3.5 supports hold
    ~1000-2000 points and are elongated, so keypoints pass getSpacialHistogramDescriptors' eigenvalue-ratio test (thVar [3, 1.5]);
    a sphere of R_desc = 9 takes in ~9 ridges -- the shape of the reference's CT surfaces as far as the path is concerned."""
    rng = np.random.default_rng(seed)
    ny, nz = 8, 6

    def on_ridges(n, half_w, dz):
        r = rng.integers(0, ny * nz, n); iy, iz = r % ny, r // ny
        x = rng.uniform(0, 60, n)
        return np.column_stack([x, 4 + 6 * iy + rng.uniform(-half_w, half_w, n), 3.5 + 6.5 * iz + 0.8 * np.sin(0.5 * x + 0.7 * iy + 1.3 * iz) + dz(n)])
    pts = on_ridges(P, 1.2, lambda n: rng.normal(0, 0.25, n))
    kp = on_ridges(S, 1.0, lambda n: rng.uniform(-0.4, 0.4, n))
    return pts, kp


DESC_OPT = dict(min_pts=500, max_pts=6000, R=3.5, thVar=[3, 1.5], k=0.85, ALIGN_POINTS=True, VERBOSE=0)
MATCH_PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate", MatchThreshold=10, MaxRatio=0.99,
                 Metric="SAD", Unique=True, VERBOSE=0)


def desc_chain_data(dev, n_model_kp: int, n_surface_kp: int, P: int = 600_000, compact: bool = True):
    """cfg 4 -> cfg 2 on rows the descriptor kernel produces: a model cloud with n_model_kp keypoints, and a SURFACE that is a
    moved, noisy crop of it (the n_surface_kp model keypoints nearest to a centre, jittered; the cloud around them with its own
    noise), both described by pcreg_dev_spatial_histogram_descriptors[_rows_u16].  -> dict with the device tensors and timings."""
    import torch
    from pcreg_amd.device import DescriptorPipeline
    cloud, kpM = _ridge_volume(P, n_model_kp)
    c0 = np.array([30.0, 25.0, 20.0])
    rng = np.random.default_rng(77)
    dk = np.linalg.norm(kpM - c0, axis=1)
    near = np.sort(np.argpartition(dk, n_surface_kp - 1)[:n_surface_kp])
    r_s = float(dk[near].max())
    crop = cloud[np.linalg.norm(cloud - c0, axis=1) < r_s + 4.5]
    c, s_ = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]]); t = np.array([2.0, -1.0, 0.5])
    move = lambda x: (x - c0) @ R.T + c0 + t
    surf = move(crop + rng.normal(0, 0.02, crop.shape))
    kpS = move(kpM[near] + rng.normal(0, 0.05, (n_surface_kp, 3)))
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
    dp = DescriptorPipeline(dev)
    tm, tkm, ts, tks = tt(cloud), tt(kpM), tt(surf), tt(kpS)
    dp.describe(tm[:, :50_000].contiguous(), tkm[:, :1000].contiguous(), DESC_OPT, compact=compact)           # warm-up (allocations)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    featM, descM, VM = dp.describe(tm, tkm, DESC_OPT, compact=compact)
    torch.cuda.synchronize(); ms_m = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    featS, descS, VS = dp.describe(ts, tks, DESC_OPT, compact=compact)
    torch.cuda.synchronize(); ms_s = (time.perf_counter() - t0) * 1e3
    return dict(dp=dp, cloud=cloud, kpM=kpM, surf=surf, kpS=kpS, near=near, featM=featM, descM=descM, VM=VM, featS=featS, descS=descS, VS=VS,
                describe_model_ms=ms_m, describe_surface_ms=ms_s, R=R, t=t, c0=c0)


def _match_stats(reset: bool = True) -> dict:
    from pcreg_amd._lib import check, lib
    out = (C.c_longlong * 8)()
    check(lib().pcreg_debug_match_stats(out, 1 if reset else 0))
    q = max(int(out[0]), 1)
    return {"queries": int(out[0]), "rescored_per_query": round(out[1] / q, 2), "unproven": int(out[2]), "to_exhaustive": int(out[3]),
            "back_items": int(out[4]), "back_to_exhaustive": int(out[5]), "calls": int(out[6])}


def extra_desc_chain(dev, with_cpu: bool) -> dict:
    """getMatches and the sphere sweep on rows that the repo's own descriptor kernel produces (sparse, spatially
    correlated spherical histograms of overlapping supports, getSpacialHistogramDescriptors.m:150-171) instead of i.i.d. Poisson
    rows.  (a) 200 k model keypoints, 20 k surface keypoints: describe -> pcreg_dev_get_matches_rows_u16; (b) the reference's
    sweep shape, 60 k / 2 k: describe -> SphereSweep.run.  Each with the certified matcher's counters."""
    import torch
    from pcreg_amd._lib import check, lib
    from pcreg_amd.sweep import SphereSweep
    L = lib()
    res = {}
    # ---- (a) cfg 4 -> cfg 2
    d = desc_chain_data(dev, 200_000, 20_000)
    dp, VM, VS = d["dp"], d["VM"], d["VS"]
    ms = _ev_ms(lambda: dp.match(d["descS"], VS, d["descM"], VM, MATCH_PAR), reps=2)
    check(L.pcreg_debug_set(b"match_stats", 1)); _match_stats()
    pairs, n_pairs = dp.match(d["descS"], VS, d["descM"], VM, MATCH_PAR)
    st = _match_stats(); check(L.pcreg_debug_set(b"match_stats", 0))
    P_ = int(n_pairs.item())
    pr = pairs[:P_].cpu().numpy().astype(np.int64)
    # a pair is right when the surface keypoint's model original is the matched model keypoint (both sets keep keypoint order)
    iS = d["descS"].index[:VS].cpu().numpy(); iM = d["descM"].index[:VM].cpu().numpy()
    right = int(np.sum(d["near"][iS[pr[:, 0] - 1]] == iM[pr[:, 1] - 1])) if P_ else 0
    flops = (3.0 * 981 - 1.0) * VS * VM
    res["get_matches"] = {"workload": f"describe {d['kpM'].shape[0]} + {d['kpS'].shape[0]} keypoints -> getMatches rows_u16 {VS} x {VM}, D 981",
                          "describe_model_ms": round(d["describe_model_ms"], 2), "describe_surface_ms": round(d["describe_surface_ms"], 2),
                          "ms": round(ms, 2), "gpairs_per_s": round(VS * VM / ms / 1e6, 1), "pairs_found": P_, "pairs_right": right, "stats": st,
                          "roofline": {"bound": "valu", "achieved": round(flops / ms / 1e9, 1), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(flops / ms / 1e9 / PEAK_FP32_TFLOPS, 4), "traffic": None}}
    del d, dp, pairs
    torch.cuda.empty_cache()
    # ---- (b) the sweep at the reference's shape on described rows
    d = desc_chain_data(dev, 60_000, 2_000, compact=False)
    VM, VS = d["VM"], d["VS"]
    sw = SphereSweep(d["featM"][:VM], d["descM"][:VM], d["featS"][:VS], d["descS"][:VS], device=dev)
    opt = dict(minPtNum=3, iterNum=10000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    kw = dict(R_desc=9.0, d_spheres=5.0, min_pts=1400, putative_thresh=170, seed=0)
    out = sw.run(MATCH_PAR, opt, **kw)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = sw.run(MATCH_PAR, opt, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    check(L.pcreg_debug_set(b"match_stats", 1)); _match_stats()
    sw.run(MATCH_PAR, opt, **kw)
    st = _match_stats(); check(L.pcreg_debug_set(b"match_stats", 0))
    ms = min(ts) * 1e3
    S = len(out["centres"])
    # a registration is right when its T carries the model keypoints onto their moved surface copies ([pts2, 1] * T = [pts1, 1],
    # pts1 = surface, pts2 = model: estimateTransform.m's actual contract)
    good = 0
    for T in out["transforms"]:
        if T is not None:
            fwd = (np.c_[d["kpM"][d["near"][:50]], np.ones(50)] @ T)[:, :3]
            good += bool(np.abs(fwd - d["kpS"][:50]).max() < 0.5)
    union = int(len(np.unique(np.concatenate(out["model_rows"])))) if S else 0
    flops = (3.0 * 981 - 1.0) * VS * union                       # each (surface row, model row IN SOME sphere) pair once
    res["sweep"] = {"workload": f"describe 60000 + 2000 keypoints -> sweep: {S} spheres of {VM} described model keypoints, {VS} surface rows",
                    "describe_model_ms": round(d["describe_model_ms"], 2), "ms": round(ms, 2), "spheres_per_s": round(S / ms * 1e3, 1) if ms > 0 else 0,
                    "trial_spheres": int(len(out["trial"])), "registered": int(sum(t is not None for t in out["transforms"])), "registered_right": int(good),
                    "stats": st,
                    "roofline": {"bound": "valu", "achieved": round(flops / ms / 1e9, 1), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(flops / ms / 1e9 / PEAK_FP32_TFLOPS, 4), "traffic": None, "dedup_model_rows": union}}
    del sw, d
    torch.cuda.empty_cache()
    return res


def extra_host_tier(dev, with_cpu: bool) -> dict:
    """what a MATLAB caller gets -- the HOST tier of the C ABI (pageable host arrays in, host arrays out), wall
    clock, beside the device-tier figures above.  The arrays are COLUMN-major like MATLAB's (what mxGetPr hands the gateway): a
    row-major numpy array would first be transposed by numpy on the host, 0.55 s for cfg 2's 1.96 GB, which is not the library's
    time.  Steady state = the second call (the library's scratch is grow-only); the first call's allocations are reported too."""
    import pcreg_amd as pc
    from pcreg_amd._lib import check, lib
    from pcreg_amd.api import _desc_opts
    res = {}
    F = np.asfortranarray
    def _t(fn):
        t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3
    wall = lambda fn, reps=3: min(_t(fn) for _ in range(reps))
    # ransac, cfg 1's data
    rng = np.random.default_rng(1)
    n = 1000
    pts = rng.uniform([-3, -2, 0], [3, 2, 3], (n, 3))
    R = eul2rotm_zyx([1.5, -1.2, 0.8]); t = np.array([1.0, 2.0, 3.0])
    loc1S = F(pts @ R + t); loc1M = F(pts + np.random.default_rng(2).normal(0, 0.1, pts.shape))
    coef = dict(minPtNum=3, iterNum=20000, thInlrRatio=0.5, thDist=0.1, REFINE=True, VERBOSE=0)
    pc.ransac(loc1M, loc1S, coef, pc.estimateTransform, pc.calcDists, seed=3)
    res["ransac_n1000_ms"] = round(wall(lambda: pc.ransac(loc1M, loc1S, coef, pc.estimateTransform, pc.calcDists, seed=3), 5), 3)
    # getMatches at the sweep's per-sphere shape and at cfg 2's
    par = dict(MATCH_PAR)
    dM = rng.poisson(3.0, (1500, 980)).astype(np.float64)
    dS = F(np.vstack([dM[rng.choice(1500, 1500)] + rng.poisson(0.15, (1500, 980)), rng.poisson(3.0, (500, 980))]).astype(np.float64)); dM = F(dM)
    pc.getMatches(dS, dM, par)
    res["getMatches_2000x1500_ms"] = round(wall(lambda: pc.getMatches(dS, dM, par)), 2)
    with pc.DescSet(dS) as hS, pc.DescSet(dM) as hM:                     # both sets resident: the sphere loop's form
        rows = np.arange(0, 1500, 1)
        pc.getMatchesOnSet(hS, hM, rows, par)
        res["getMatchesOnSet_2000x1500_ms"] = round(wall(lambda: pc.getMatchesOnSet(hS, hM, rows, par)), 2)
    # all spheres of the sweep in one call: 329 row subsets (~1533 rows each) of a 60 000 x 980 model set, 2000 surface rows --
    # with the 470 MB model set uploaded by every call, and with both sets resident
    VMs, Ss = 60_000, 329
    dMs = rng.poisson(3.0, (VMs, 980)).astype(np.float64)
    dSs = F((dMs[rng.choice(VMs, 2000, replace=False)] + rng.poisson(0.15, (2000, 980))).astype(np.float64)); dMs = F(dMs)
    rows_list = [np.sort(rng.choice(VMs, 1533, replace=False)) for _ in range(Ss)]
    pc.getMatchesSegmented(dSs, dMs, rows_list, par)
    res["getMatchesSegmented_329x1533_ms"] = round(wall(lambda: pc.getMatchesSegmented(dSs, dMs, rows_list, par), 2), 1)
    with pc.DescSet(dSs) as hS, pc.DescSet(dMs) as hM:
        pc.getMatchesSegmentedOnSet(hS, hM, rows_list, par)
        res["getMatchesSegmentedOnSet_329x1533_ms"] = round(wall(lambda: pc.getMatchesSegmentedOnSet(hS, hM, rows_list, par), 2), 1)
    del dSs, dMs
    Q, M = 50_000, 200_000
    dM = rng.poisson(3.0, (M, 980)).astype(np.float64)
    dS = F((dM[rng.choice(M, Q, replace=False)] + rng.poisson(0.15, (Q, 980))).astype(np.float64)); dM = F(dM)
    res["getMatches_50kx200k_first_call_ms"] = round(_t(lambda: pc.getMatches(dS, dM, par)), 1)
    res["getMatches_50kx200k_ms"] = round(wall(lambda: pc.getMatches(dS, dM, par), 2), 1)
    res["getMatches_50kx200k_upload_GB"] = round((Q + M) * 980 * 8 / 1e9, 2)
    del dS, dM
    # descriptors, 100 k keypoints: 784 MB of doubles come back (the C ABI call itself, outputs preallocated as a MEX gateway does)
    ptsc, kp = _ridge_cloud(1_000_000, 100_000)
    ptsc, kp = F(ptsc), F(kp)
    P_, S_ = ptsc.shape[0], kp.shape[0]
    o = _desc_opts(DESC_OPT)
    feat = np.empty((S_, 3)); desc = np.empty((S_, 980)); V = C.c_int(0)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    call = lambda: check(lib().pcreg_spatial_histogram_descriptors(pd(ptsc), P_, P_, pd(kp), S_, S_, C.byref(o), pd(feat), pd(desc), C.byref(V)))
    res["descriptors_100k_first_call_ms"] = round(_t(call), 1)
    res["descriptors_100k_ms"] = round(wall(call, 2), 1)
    res["descriptors_100k_rows_out"] = int(V.value)
    res["note"] = "column-major pageable arrays in/out (what the MEX gateway passes); steady state + first call"
    return res


def summary_of(out: dict) -> dict:
    """Every config's headline number in < 600 characters, as the last object of the line."""
    ex = out.get("extras", {})
    g = lambda d, *ks: (lambda v: None if v is None else v)(_dig(d, ks))
    sm = {"step_ms": out.get("ms_per_step"), "step_frac": g(out, "roofline", "frac"), "step_frac_alone": g(out, "roofline", "alone", "frac"),
          "step_ms_with_prepare": out.get("ms_per_step_with_model_prepare"),
          "cfg2_ms": g(ex, "getMatches_cfg2", "ms"), "cfg2_frac": g(ex, "getMatches_cfg2", "roofline", "frac"),
          "cfg3_ms": g(out, "cfg3_model_2M", "ms_per_step"),
          "cfg4_ms": g(ex, "descriptors_cfg4", "ms"), "cfg4_frac": g(ex, "descriptors_cfg4", "roofline", "frac"),
          "align_ms": g(ex, "align_points_knn_batched", "ms"), "align_frac": g(ex, "align_points_knn_batched", "roofline", "frac"),
          "cfg1_ms": g(ex, "ransac_cfg1", "ms"), "cfg1b_ms": g(ex, "ransac_cfg1_batched", "ms"), "cfg1b_frac": g(ex, "ransac_cfg1_batched", "roofline", "frac"),
          "sweep_ms": g(ex, "sweep", "ms"), "sweep_host_ms": g(ex, "sweep", "host_tier", "ms"), "sweep_host_on_model_ms": g(ex, "sweep", "host_tier", "on_model_ms"), "sweep_frac": g(ex, "sweep", "roofline", "frac"),
          "cfg5_regs_per_s": g(out, "cfg5_batch", "registrations_per_s"),
          "step_serial_ms": g(out, "two_in_flight", "model_1M", "one_in_flight_ms"),
          "two_1M_x": g(out, "two_in_flight", "model_1M", "speedup"), "two_125k_x": g(out, "two_in_flight", "rank_of_8_emulated_125k", "speedup"),
          "two_125k_ms": g(out, "two_in_flight", "rank_of_8_emulated_125k", "two_in_flight_ms"),
          "chain_match_ms": g(ex, "desc_chain", "get_matches", "ms"), "chain_match_unproven": g(ex, "desc_chain", "get_matches", "stats", "unproven"),
          "chain_sweep_ms": g(ex, "desc_chain", "sweep", "ms"), "chain_sweep_right": g(ex, "desc_chain", "sweep", "registered_right"),
          "host_ransac_ms": g(ex, "host_tier", "ransac_n1000_ms"), "host_cfg2_ms": g(ex, "host_tier", "getMatches_50kx200k_ms"),
          "host_sphere_ms": g(ex, "host_tier", "getMatches_2000x1500_ms"), "host_sphere_on_sets_ms": g(ex, "host_tier", "getMatchesOnSet_2000x1500_ms"),
          "host_sweep_ms": g(ex, "host_tier", "getMatchesSegmented_329x1533_ms"), "host_sweep_on_sets_ms": g(ex, "host_tier", "getMatchesSegmentedOnSet_329x1533_ms"),
          "host_desc100k_ms": g(ex, "host_tier", "descriptors_100k_ms")}
    return {k: v for k, v in sm.items() if v is not None}


def _dig(d, ks):
    for k in ks:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


# ---- main ---------------------------------------------------------------------------------------------------------

def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model-points", type=int, default=1_000_000, help="model points in TOTAL (sharded over the GPUs)")
    ap.add_argument("--surface-points", type=int, default=50_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only")
    ap.add_argument("--crops", type=int, default=64, help="crops of the cfg 5 batch")
    ap.add_argument("--in-flight", type=int, default=2, help="registrations in flight per rank (one HIP stream + buffer set each); 1 = the serial step of rounds 1-3")
    ap.add_argument("--skip", default="", help="comma-separated extras to leave out (e.g. host_tier,desc_chain)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # PCREG_BENCH_SHARE_GPU=1 (tests only): every rank on cuda:0 with gloo carrying the collectives -- RCCL refuses two ranks
    # on one device; this rehearses the whole N > 1 control flow of this file on the one-GPU box.  Its timings mean nothing.
    share = os.environ.get("PCREG_BENCH_SHARE_GPU") == "1"
    ordinal = 0 if share else local_rank
    torch.cuda.set_device(ordinal)
    dev = torch.device("cuda", ordinal)
    force = os.environ.get("PCREG_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ     # one-rank RCCL rehearsal
    collective = world > 1 or force
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    from pcreg_amd._lib import lib, check
    check(lib().pcreg_set_device(ordinal))
    torch.cuda.set_device(ordinal)
    ctx = Ctx(rank, world, dev, collective)

    Q, M_total = args.surface_points, args.model_points
    head = run_registration(ctx, M_total, Q, args.steps, args.warmup, time_kernel=True, in_flight=args.in_flight)
    model, surf = head.pop("_model"), head.pop("_surf")
    extra_steps, extra_warm = max(3, min(args.steps, 10)), min(args.warmup, 2)
    more = {}
    if not args.no_extras:
        if world > 1:
            for key, mt in (("cfg3_model_2M", 2_000_000), ("weak_1M_per_gpu", 1_000_000 * world)):
                r = run_registration(ctx, mt, Q, extra_steps, extra_warm, time_kernel=False, in_flight=args.in_flight)
                r.pop("_model"); r.pop("_surf")
                more[key] = {"workload": f"{Q} surface pts vs {mt} model pts, {r['rows_per_gpu']} rows per GPU", "scaling": "strong" if key.startswith("cfg3") else "weak",
                             "ms_per_step": round(r["ms_per_step"], 4), "value": round(r["value"], 2), "unit": "Gpairs/s", "in_flight": args.in_flight,
                             "ransac": r["ransac"], "steps": extra_steps}
        else:
            r = run_registration(ctx, 2_000_000, Q, extra_steps, extra_warm, time_kernel=False, in_flight=args.in_flight)
            r.pop("_model"); r.pop("_surf")
            more["cfg3_model_2M"] = {"workload": f"{Q} surface pts vs 2000000 model pts on one GPU (the 1-GPU point of cfg 3's strong-scaling curve)",
                                     "scaling": "strong", "ms_per_step": round(r["ms_per_step"], 4), "value": round(r["value"], 2), "unit": "Gpairs/s",
                                     "in_flight": args.in_flight, "ransac": r["ransac"], "steps": extra_steps}
        more["cfg5_batch"] = run_batch_cfg5(ctx, args.crops, 1_000_000, Q)
        if world == 1:
            try:
                more["two_in_flight"] = extra_two_in_flight(ctx, Q, 40)
            except Exception as e:          # an extra must never take the headline down with it
                more["two_in_flight"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        rows = head["rows_per_gpu"]
        kernel_ms, search_ms = head["kernel_ms"], head["search_call_ms"]
        alg_tflops = FLOP_PER_PAIR * float(Q) * rows / (kernel_ms * 1e-3) / 1e12
        exe_tflops = FLOP_PER_PAIR_EXECUTED * float(Q) * rows / (kernel_ms * 1e-3) / 1e12
        alg_bytes = 4.0 * 3 * (Q + rows) + 16.0 * Q
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is the one
        # scripts/pmc_traffic.sh measured -- valid only for the kernel sources it was taken from.  profiles/traffic.json records
        # the content hash of those sources; a build from different sources reports null and says why.
        traffic, traffic_source = None, "not measured for this shape"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(f"knn_search:Q{Q}:M{rows}")
                if isinstance(ent, dict):
                    now = kernel_source_hash(ent.get("sources", []))
                    if now == ent.get("sources_sha256"):
                        traffic, traffic_source = ent["bytes"], f"profiles/traffic.json ({ent.get('measured', '?')}; sources sha256 {now[:12]} = this build's)"
                    else:
                        traffic_source = f"profiles/traffic.json is stale: taken from sources {str(ent.get('sources_sha256'))[:12]}, this build is {now[:12]}"
            except Exception as e:
                traffic_source = f"profiles/traffic.json unreadable: {e}"
        out = {
            "metric": "KNN Gpairs/s end-to-end (search + filters + RANSAC per step; registrations/s = 1000/ms_per_step)",
            "value": round(head["value"], 2), "unit": "Gpairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4),
            "ms_per_step_with_model_prepare": round(head["ms_per_step"] + head["model_prepare_ms"], 4),
            "metric_definition": "ms_per_step = wall time of `steps` registrations / steps, model PREPARED once (since round 3), `in_flight` "
                                 "registrations in flight per rank (since round 4: 2; the serial step is two_in_flight.model_1M.one_in_flight_ms "
                                 "or --in-flight 1); rounds 1-2 prepared the model inside every step (ms_per_step_with_model_prepare)",
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 (search: f16-split matrix-core candidates, exact f32 re-rank) / f64 (RANSAC)", "data": "synthetic",
            "config": {"workload": f"{Q} surface pts vs a FIXED {M_total}-pt model ({rows} rows per GPU, row-sharded over {world} GPU(s)), "
                                   f"top-2 + threshold/ratio/Unique + RANSAC(3,1e4,0.3,0.08,REFINE)",
                       "surface_points": Q, "model_points_total": M_total, "parallelism": f"model-shard x{world}"},
            "registrations_per_s": round(1e3 / head["ms_per_step"], 2), "in_flight": head["in_flight"],
            "knn_kernel": {"name": "knn_candidates_f16_pipe_kernel", "ms": round(kernel_ms, 4), "search_call_ms": None if search_ms is None else round(search_ms, 4),
                           "launches_timed": head["launches_timed"], "gpairs_per_s_per_gpu": round(Q * rows / ((search_ms or kernel_ms) * 1e-3) / 1e9, 1),
                           "model_prepare_ms_once": round(head["model_prepare_ms"], 3),
                           "note": "model prepared once per model; step = 4 search + 1 match launch + RANSAC chain"},
            "ransac": head["ransac"],
            "roofline": {"bound": "mfma", "achieved": round(alg_tflops, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(alg_tflops / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_flop_per_pair": FLOP_PER_PAIR,
                         "vs_fp32_vector_peak": round(alg_tflops / PEAK_FP32_TFLOPS, 3),
                         "matrix_pipe": {"executed_flop_per_pair": FLOP_PER_PAIR_EXECUTED, "tflops": round(exe_tflops, 1),
                                         "occupancy_of_dense_f16_peak": round(exe_tflops / PEAK_F16_MFMA_TFLOPS, 4)},
                         "hbm_algorithmic_gbps": round(alg_bytes / ((search_ms or kernel_ms) * 1e-3) / 1e9, 2),
                         "note": "HIP events per launch; VALU-issue-bound, not HBM: docs/BENCH_NOTES.md"},
        }
        out.update(more)
        k1 = _dig(more, ("two_in_flight", "model_1M", "knn_kernel_ms_one_in_flight"))
        if head["in_flight"] > 1:
            out["roofline"]["in_flight"] = head["in_flight"]
            out["roofline"]["contended"] = "the kernel shares the chip with the other lane's RANSAC chain inside the timed region"
            if k1 and M_total == 1_000_000 and Q == 50_000:
                t1 = FLOP_PER_PAIR * float(Q) * rows / (k1 * 1e-3) / 1e12
                out["roofline"]["alone"] = {"ms": k1, "achieved": round(t1, 2), "frac": round(t1 / PEAK_F16_MFMA_TFLOPS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, surf)
    del model, surf
    if world == 1 and not args.no_extras:
        ex = {}
        skip = set(x for x in args.skip.split(",") if x)
        with_cpu = not args.no_cpu_baseline
        for name, fn in (("getMatches_cfg2", extra_get_matches), ("descriptors_cfg4", extra_descriptors),
                         ("align_points_knn_batched", extra_align), ("ransac_cfg1", extra_ransac_cfg1),
                         ("ransac_cfg1_batched", extra_ransac_cfg1_batched), ("sweep", extra_sweep),
                         ("desc_chain", extra_desc_chain), ("host_tier", extra_host_tier)):
            if name in skip:
                continue
            try:
                ex[name] = fn(dev, with_cpu)
            except Exception as e:          # an extra must never take the headline down with it
                ex[name] = {"error": f"{type(e).__name__}: {e}"}
        out["extras"] = ex
    if rank == 0:
        out["summary"] = summary_of(out)          # LAST key: the tail of stdout carries every config's number
        print(json.dumps(out), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
