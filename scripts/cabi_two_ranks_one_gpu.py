"""World-2 run of the C-ABI multi-GPU entry points (comm.hip) with both ranks on cuda:0: the host-staged communicator
(pcreg_comm_init_host_staged) carries the three exchanges that RCCL carries on an N-GPU node.  Each rank passes ITS half of
the model; both must return the single-GPU pairs / registration, checked against the oracle.  Ragged cases: a rank with
M_local == 0, uneven hypothesis shares, a second call on the same communicator (persistent scratch).
usage: cabi_two_ranks_one_gpu.py RANK WORLD SHM_NAME"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
from pcreg_amd._lib import RansacOpts, check, lib        # noqa: E402
from oracle import c_oracle                              # noqa: E402  (checker)

L = lib()
check(L.pcreg_set_device(0))
check(L.pcreg_comm_init_host_staged(rank, world, name.encode()))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def sharded_match(surf, model, cuts, thr, ratio, unique):
    lo, hi = cuts[rank], cuts[rank + 1]
    qf = np.asfortranarray(surf); mf = np.asfortranarray(model[lo:hi]) if hi > lo else np.zeros((1, 3), np.float32, order="F")
    pairs = np.zeros((len(surf), 2), dtype=np.uint32); P = C.c_int(0)
    check(L.pcreg_match_points_sharded_f32(_p(qf, C.c_float), len(surf), len(surf), _p(mf, C.c_float), hi - lo, max(hi - lo, 1), lo, len(model),
                                           C.c_float(thr), C.c_float(ratio), unique, _p(pairs, C.c_uint32), C.byref(P)))
    return pairs[:P.value]


ok = True
rng = np.random.default_rng(3)
model = (rng.random((50001, 3)) * [100, 56, 99]).astype(np.float32)
surf = (model[rng.choice(50001, 6000, replace=False)] + rng.normal(0, 0.03, (6000, 3))).astype(np.float32)
surf[:64] = surf[64:128]                                  # duplicated queries: Unique ties, the lower index wins
for cuts, unique in (([0, 20000, 50001], 1), ([0, 20000, 50001], 0), ([0, 50001, 50001], 1), ([0, 0, 50001], 1)):   # incl. empty shards
    got = sharded_match(surf, model, cuts, 0.25, 0.8, unique)
    ref = c_oracle.match_points_f32(surf, model, 0.25, 0.8, bool(unique))
    ok = ok and np.array_equal(got, ref) and len(ref) > 1000
# a smaller second problem on the same communicator (buffers are reused, counts differ)
got = sharded_match(surf[:900], model[:7000], [0, 3000, 7000], 1.0, 0.9, 1)
ok = ok and np.array_equal(got, c_oracle.match_points_f32(surf[:900], model[:7000], 1.0, 0.9, True))

ref = c_oracle.match_points_f32(surf, model, 0.25, 0.8, True)
p1 = np.asfortranarray(surf[ref[:, 0] - 1].astype(np.float64)); p2 = np.asfortranarray(model[ref[:, 1] - 1].astype(np.float64))
n = len(ref)
for iters in (1001, 1):                                   # uneven shares; a rank with no hypotheses at all
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    o = RansacOpts(3, iters, 0.3, 0.08, 1, 0, 11)
    T = np.zeros(16); inl = np.zeros(n, dtype=np.int32); ni, ns, mi, fl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(L.pcreg_ransac_sharded(_p(p1, C.c_double), _p(p2, C.c_double), n, n, C.byref(o), _p(T, C.c_double), _p(inl, C.c_int32),
                                 C.byref(ni), C.byref(ns), C.byref(mi), C.byref(fl)))
    rr = c_oracle.ransac(p1, p2, coef, seed=11)
    ok = ok and bool(fl.value) == bool(rr["failed"]) and ns.value == rr["numSuccess"] and mi.value == rr["maxInliers"]
    ok = ok and np.array_equal(inl[:ni.value].astype(np.int64), rr["inlierIdx"])
    if not rr["failed"]:
        ok = ok and np.linalg.norm(T.reshape(4, 4, order="F") - rr["T"]) < 1e-9
check(L.pcreg_comm_destroy())
print(f"rank {rank}: cabi_world2_ok={ok}", flush=True)
sys.exit(0 if ok else 1)
