"""Multi-GPU behind the C ABI (pcreg_comm_* / *_sharded, comm.hip): a one-rank RCCL communicator rehearses every
collective of the N > 1 protocol INSIDE the library (dlopen'ed librccl, ncclAllGather x2, ncclAllReduce) and must
give exactly the single-GPU results.  (RCCL refuses two ranks on one device, and this pool gives one GPU per box:
the world-2 protocol itself is covered by tests/test_sharded_cpu.py over gloo and by the two-ranks-one-GPU run;
the 8-GPU run is bench.py --gpus 8.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def test_one_rank_communicator_equals_single_gpu(oracle_c):
    import pcreg_amd as pc
    from pcreg_amd._lib import RansacOpts, check, lib
    L = lib()
    cid = CommId()
    check(L.pcreg_comm_get_unique_id(C.byref(cid)))
    check(L.pcreg_comm_init(0, 1, C.byref(cid)))
    try:
        r, w = C.c_int(-1), C.c_int(-1)
        check(L.pcreg_comm_rank(C.byref(r), C.byref(w)))
        assert (r.value, w.value) == (0, 1)
        assert L.pcreg_comm_init(0, 1, C.byref(cid)) != 0           # a second communicator is refused, not leaked
        rng = np.random.default_rng(3)
        model = (rng.random((60000, 3)) * [100, 56, 99]).astype(np.float32)
        pick = rng.choice(60000, 7000, replace=False)
        surf = (model[pick] + rng.normal(0, 0.03, (7000, 3))).astype(np.float32)
        qf, mf = np.asfortranarray(surf), np.asfortranarray(model)
        for unique in (1, 0):
            pairs = np.zeros((7000, 2), dtype=np.uint32); P = C.c_int(0)
            check(L.pcreg_match_points_sharded_f32(_p(qf, C.c_float), 7000, 7000, _p(mf, C.c_float), 60000, 60000, 0, 60000,
                                                   C.c_float(0.25), C.c_float(0.8), unique, _p(pairs, C.c_uint32), C.byref(P)))
            want = pc.match_points(surf, model, 0.25, 0.8, bool(unique))
            np.testing.assert_array_equal(pairs[:P.value], want)
            np.testing.assert_array_equal(want, oracle_c.match_points_f32(surf, model, 0.25, 0.8, bool(unique)))
        # ransac with the hypotheses "split" over the one rank == pcreg_ransac with the built-in sampler
        want = pc.match_points(surf, model, 0.25, 0.8, True)
        p1 = np.asfortranarray(surf[want[:, 0] - 1].astype(np.float64)); p2 = np.asfortranarray(model[want[:, 1] - 1].astype(np.float64))
        n = len(want)
        coef = dict(minPtNum=3, iterNum=3000, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
        o = RansacOpts(3, 3000, 0.3, 0.08, 1, 0, 11)
        T = np.zeros(16); inl = np.zeros(n, dtype=np.int32); ni, ns, mi, fl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(L.pcreg_ransac_sharded(_p(p1, C.c_double), _p(p2, C.c_double), n, n, C.byref(o), _p(T, C.c_double), _p(inl, C.c_int32),
                                     C.byref(ni), C.byref(ns), C.byref(mi), C.byref(fl)))
        Tr, inr, nsr, mir, _ = pc.ransac(p1, p2, coef, pc.estimateTransform, pc.calcDists, seed=11)
        assert fl.value == 0 and (ns.value, mi.value) == (nsr, mir)
        np.testing.assert_array_equal(inl[:ni.value], np.asarray(inr).ravel().astype(np.int32))
        assert np.linalg.norm(T.reshape(4, 4, order="F") - Tr) < 1e-12
    finally:
        check(L.pcreg_comm_destroy())
    assert L.pcreg_comm_rank(C.byref(r), C.byref(w)) != 0             # closed


def _run_two_ranks(name, stagger=0.0):
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = {}
    for r in (1, 0):                        # rank 1 first: with `stagger` it meets whatever lies under the name before rank 0 replaces it
        procs[r] = subprocess.Popen([sys.executable, os.path.join(root, "scripts", "cabi_two_ranks_one_gpu.py"), str(r), "2", name],
                                    stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=root)
        if r == 1 and stagger:
            time.sleep(stagger)
    try:
        for r in (0, 1):
            o = procs[r].communicate(timeout=600)[0]
            assert procs[r].returncode == 0 and f"rank {r}: cabi_world2_ok=True" in o, o[-3000:]
    finally:
        for pr in procs.values():
            if pr.poll() is None:
                pr.kill()
        try:
            os.unlink("/dev/shm" + name)    # a failed run leaks its segment (the last rank to leave unlinks it otherwise)
        except FileNotFoundError:
            pass
    assert not os.path.exists("/dev/shm" + name)


def test_two_ranks_through_the_c_abi_on_one_gpu():
    """The N > 1 path of comm.hip with two PROCESSES on cuda:0 (host-staged communicator): rank strides of the gathered
    top-2, one contributor per table column, empty shards, uneven / empty hypothesis shares, reuse of the scratch."""
    import os
    _run_two_ranks(f"/pcreg_test_{os.getpid()}")


@pytest.mark.parametrize("debris", ["complete_run", "crashed_mid_attach", "garbage"])
def test_host_staged_communicator_ignores_a_dirty_segment(debris):
    """a segment left under the same name by a crashed run -- every rank attached, sense = 1, a
    half-counted barrier; or a run that died while attaching; or garbage -- must not be trusted.  Rank 0 unlinks and
    creates exclusively, rank 1 (started first, so it opens the debris) is turned away or notices the name moving on.
    The world-2 protocol then has to give the oracle's results as usual."""
    import os
    import struct
    name = f"/pcreg_dirty_{os.getpid()}_{debris}"
    path = "/dev/shm" + name
    fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
    try:
        os.ftruncate(fd, 4096 + 2 * (64 << 20))
        magic = 0x70637265675f6873
        if debris == "complete_run":       # magic, world, attached, ready, detached, count, sense
            os.pwrite(fd, struct.pack("<Qiiiiii", magic, 2, 2, 1, 0, 1, 1), 0)
        elif debris == "crashed_mid_attach":
            os.pwrite(fd, struct.pack("<Qiiiiii", magic, 2, 1, 0, 0, 0, 0), 0)
        else:
            os.pwrite(fd, bytes(range(256)) * 16, 0)
        os.pwrite(fd, b"\xff" * 4096, 4096)                       # stale slot contents
    finally:
        os.close(fd)
    _run_two_ranks(name, stagger=1.5)
