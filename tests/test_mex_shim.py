"""mex/pcreg_mex.cpp through a compiler and through a driver that plays MATLAB.

MATLAB's mex.h exists on no box, so the shim is built against tests/mexstub/mex.h (a minimal
mxArray: test infrastructure, not a MATLAB-compatibility claim) with -Wall -Wextra -Werror and
linked to libpcreg_hip.so; tests/mexstub/mex_driver.cpp builds the mxArrays matlab/*.m would pass
and calls mexFunction.  CPU: the gateway compiles, links, dispatches, and raises pcreg:hip /
pcreg:usage through mexErrMsgIdAndTxt.  GPU: one round trip per command equals the ctypes path
(the same library underneath: any difference is the shim's transposes, struct unpacking or nlhs
handling)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "mexstub", "libmexdrv.so")


@pytest.fixture(scope="module")
def drv():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "pcreg_amd", "libpcreg_hip.so")):
        g.build()
    srcs = [os.path.join(ROOT, "mex", "pcreg_mex.cpp"), os.path.join(ROOT, "tests", "mexstub", "mex_driver.cpp")]
    newest = max(os.path.getmtime(p) for p in srcs + [os.path.join(ROOT, "include", "pcreg.h"), os.path.join(ROOT, "tests", "mexstub", "mex.h")])
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        # the shim itself must be warning-free; it is the only consumer of mex.h
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "mexstub"),
                               "-I" + os.path.join(ROOT, "include"), srcs[0]])
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "mexstub"),
                               "-I" + os.path.join(ROOT, "include"), *srcs, "-o", OUT, "-L" + os.path.join(ROOT, "pcreg_amd"), "-lpcreg_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "pcreg_amd")])
    return C.CDLL(OUT)


def _d(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _err():
    return C.create_string_buffer(1024)


def test_shim_compiles_and_raises_usage(drv):
    e = _err()
    assert drv.drv_bad_command(e, 1024) == 1
    assert e.value.decode().startswith("pcreg:usage")
    assert drv.drv_live_arrays() == 0


def test_shim_reports_nodevice_through_mexerr(drv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    p = _d(np.eye(4, 3))
    T = np.zeros(16); empty = C.c_int(0); e = _err()
    assert drv.drv_estimate_transform(_p(p), _p(p), 4, _p(T), C.byref(empty), e, 1024) == 1
    assert e.value.decode().startswith("pcreg:hip") and "no CPU fallback" in e.value.decode()


@pytest.mark.gpu
def test_shim_round_trips_equal_the_ctypes_path(drv):
    import pcreg_amd as pc
    from conftest import rigid_case
    e = _err()
    # estimateTransform + calcDists
    p1, p2, _ = rigid_case(40, 3, outlier_frac=0.0)
    T = np.zeros(16); empty = C.c_int(-1)
    assert drv.drv_estimate_transform(_p(_d(p1)), _p(_d(p2)), 40, _p(T), C.byref(empty), e, 1024) == 0, e.value
    Tm = T.reshape(4, 4, order="F")
    assert empty.value == 0 and np.array_equal(Tm, pc.estimateTransform(p1, p2))
    d = np.zeros(40)
    assert drv.drv_calc_dists(_p(_d(Tm)), _p(_d(p1)), _p(_d(p2)), 40, _p(d), e, 1024) == 0, e.value
    assert np.array_equal(d, pc.calcDists(Tm, p1, p2).ravel())
    # rank-deficient input -> []
    z = np.zeros((5, 3))
    assert drv.drv_estimate_transform(_p(_d(z)), _p(_d(z)), 5, _p(T), C.byref(empty), e, 1024) == 0 and empty.value == 1
    # ransac with a MATLAB-shaped sample table (minPtNum x iterNum int32, one column per hypothesis)
    p1, p2, _ = rigid_case(300, 5)
    rng = np.random.default_rng(0)
    iters = 200
    tab = np.stack([rng.permutation(300)[:3] + 1 for _ in range(iters)]).astype(np.int32)     # [iterNum][3]
    coef = dict(minPtNum=3, iterNum=iters, thDist=0.05, thInlrRatio=0.3, REFINE=True, VERBOSE=0)
    ref = pc.ransac(p1, p2, coef, pc.estimateTransform, pc.calcDists, sample_idx=tab)
    coef5 = np.array([3, iters, 0.05, 0.3, 1], dtype=np.float64)
    inl = np.zeros(300); ni, ns, mi, fl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    tabF = np.asfortranarray(tab.T)                                                           # 3 x iterNum column-major
    assert drv.drv_ransac(_p(_d(p1)), _p(_d(p2)), 300, _p(coef5), _p(tabF, C.c_int32), C.c_double(0), _p(T), _p(inl),
                          C.byref(ni), C.byref(ns), C.byref(mi), C.byref(fl), e, 1024) == 0, e.value
    assert fl.value == 0 and np.array_equal(T.reshape(4, 4, order="F"), ref[0])
    assert np.array_equal(inl[:ni.value], np.asarray(ref[1], dtype=np.float64).ravel()) and (ns.value, mi.value) == (ref[2], ref[3])
    # ransacBatched: three registrations of different sizes (one of them hopeless: pure noise) in ONE gateway call, sample tables
    # registration after registration -- every field equals the per-registration ransac calls
    sizes = [300, 120, 40]
    cases = [rigid_case(sizes[0], 5), rigid_case(sizes[1], 6), (rng.normal(size=(40, 3)), rng.normal(size=(40, 3)) * 7 + 3, None)]
    tabs = [np.stack([rng.permutation(nn)[:3] + 1 for _ in range(iters)]).astype(np.int32) for nn in sizes]
    refs = [pc.ransac(c[0], c[1], coef, pc.estimateTransform, pc.calcDists, sample_idx=tb) for c, tb in zip(cases, tabs)]
    P1 = np.vstack([c[0] for c in cases]); P2 = np.vstack([c[1] for c in cases])
    offb = np.zeros(4, dtype=np.int32); offb[1:] = np.cumsum(sizes)
    tabB = np.asfortranarray(np.vstack(tabs).T)                                               # 3 x (iterNum B) column-major
    TB = np.zeros(3 * 16); inlB = np.zeros(sum(sizes)); nit = C.c_int()
    nB, nsB, miB, flB = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
    assert drv.drv_ransac_batched(_p(_d(P1)), _p(_d(P2)), sum(sizes), _p(offb, C.c_int32), 3, _p(coef5), _p(tabB, C.c_int32), C.c_double(0), _p(TB), _p(inlB),
                                  C.byref(nit), _p(nB), _p(nsB), _p(miB), _p(flB), e, 1024) == 0, e.value
    k = 0
    for b, ref_b in enumerate(refs):
        failed_b = ref_b[0] is None or np.size(ref_b[0]) == 0
        assert bool(flB[b]) == failed_b, b
        if failed_b:
            assert nB[b] == 0 and not TB[16 * b:16 * b + 16].any()
            continue
        assert np.array_equal(TB[16 * b:16 * b + 16].reshape(4, 4, order="F"), ref_b[0]), b
        want_inl = np.asarray(ref_b[1], dtype=np.float64).ravel()
        assert nB[b] == len(want_inl) and np.array_equal(inlB[k:k + len(want_inl)], want_inl), b
        assert (nsB[b], miB[b]) == (ref_b[2], ref_b[3]), b
        k += len(want_inl)
    assert k == nit.value and flB[2] == 1 and flB[0] == 0
    # getMatches (par struct with strings; P x 2 uint32 column-major out)
    dS = rng.poisson(3.0, (60, 980)).astype(np.float64); dM = rng.poisson(3.0, (90, 980)).astype(np.float64)
    dM[:40] = dS[:40] + (rng.random((40, 980)) < 0.02)
    par = dict(Method="Approximate", Metric="SAD", MatchThreshold=10, MaxRatio=0.99, Unique=True, UNNORMALIZE=True, norm_factor=2.0,
               CHANGE_METRIC=True, metric_factor=0.6, VERBOSE=0)
    want = pc.getMatches(dS, dM, par)
    par7 = np.array([10, 0.99, 1, 1, 2.0, 1, 0.6], dtype=np.float64)
    pairs = np.zeros(60 * 2, dtype=np.uint32); P = C.c_int()
    assert drv.drv_get_matches(_p(_d(dS)), 60, _p(_d(dM)), 90, 980, b"SAD", _p(par7), _p(pairs, C.c_uint32), C.byref(P), e, 1024) == 0, e.value
    got = pairs[:2 * P.value].reshape(P.value, 2, order="F")
    assert P.value == len(want) and np.array_equal(got, want)
    # getMatchesSegmented: three row subsets of the model set in one gateway call == three getMatches calls
    rows_list = [np.arange(0, 90, 2), np.arange(90), np.array([3, 7, 8, 50])]
    off = np.zeros(4, dtype=np.int32); off[1:] = np.cumsum([len(r) for r in rows_list])
    rows1 = (np.concatenate(rows_list) + 1).astype(np.int32)
    pairs3 = np.zeros(3 * 60 * 2, dtype=np.uint32); npairs = np.zeros(3, dtype=np.int32); Pt = C.c_int()
    assert drv.drv_get_matches_segmented(_p(_d(dS)), 60, _p(_d(dM)), 90, 980, _p(par7), _p(rows1, C.c_int32), len(rows1), _p(off, C.c_int32), 3,
                                         _p(pairs3, C.c_uint32), C.byref(Pt), _p(npairs, C.c_int32), e, 1024) == 0, e.value
    # descCreate / getMatchesOnSet / descDestroy: the same three subsets + the whole set ([]) on RESIDENT sets, bit for bit getMatches'
    subs = rows_list + [np.zeros(0, np.int64)]
    offs = np.zeros(len(subs) + 1, dtype=np.int32); offs[1:] = np.cumsum([len(r) for r in subs])
    rows1s = (np.concatenate(subs) + 1).astype(np.int32)
    pairs4 = np.zeros(len(subs) * 60 * 2, dtype=np.uint32); P4 = (C.c_int * len(subs))()
    assert drv.drv_desc_set_round_trip(_p(_d(dS)), 60, _p(_d(dM)), 90, 980, _p(par7), _p(rows1s, C.c_int32), _p(offs, C.c_int32), len(subs),
                                       _p(pairs4, C.c_uint32), P4, e, 1024) == 0, e.value
    for z, r in enumerate(subs):
        wz = pc.getMatches(dS, dM[r] if len(r) else dM, par)
        gz = pairs4[z * 120:z * 120 + 2 * P4[z]].reshape(P4[z], 2, order="F")
        assert P4[z] == len(wz) and np.array_equal(gz, wz), z
    allp = pairs3[:2 * Pt.value].reshape(Pt.value, 2, order="F")
    seg = pc.getMatchesSegmented(dS, dM, rows_list, par)
    k = 0
    for z, r in enumerate(rows_list):
        wz = pc.getMatches(dS, dM[r], par)
        assert npairs[z] == len(wz) and np.array_equal(allp[k:k + npairs[z]], wz) and np.array_equal(seg[z], wz)
        k += npairs[z]
    assert k == Pt.value
    # getMatchesSegmentedOnSet: the three subsets in ONE gateway call on resident sets == getMatchesSegmented, and the Python mirror
    off3 = np.zeros(len(rows_list) + 1, dtype=np.int32); off3[1:] = np.cumsum([len(r) for r in rows_list])
    rows13 = (np.concatenate(rows_list) + 1).astype(np.int32)
    pairs5 = np.zeros(len(rows_list) * 60 * 2, dtype=np.uint32); P5 = (C.c_int * len(rows_list))(); Pt5 = C.c_int()
    assert drv.drv_desc_set_segmented(_p(_d(dS)), 60, _p(_d(dM)), 90, 980, _p(par7), _p(rows13, C.c_int32), len(rows13), _p(off3, C.c_int32),
                                      len(rows_list), _p(pairs5, C.c_uint32), P5, C.byref(Pt5), e, 1024) == 0, e.value
    assert Pt5.value == Pt.value and list(P5) == [int(x) for x in npairs[:len(rows_list)]]
    assert np.array_equal(pairs5[:2 * Pt5.value].reshape(Pt5.value, 2, order="F"), allp)
    with pc.DescSet(dS) as hS, pc.DescSet(dM) as hM:
        on = pc.getMatchesSegmentedOnSet(hS, hM, rows_list, par)
        on2 = pc.getMatchesSegmentedOnSet(hS, hM, rows_list[::-1], par)          # the sets' row-major copies are reused
        one = pc.getMatchesOnSet(hS, hM, rows_list[0], par)                        # and the column-major ones still serve the per-sphere call
    for z in range(len(rows_list)):
        assert np.array_equal(on[z], seg[z]) and np.array_equal(on2[len(rows_list) - 1 - z], seg[z])
    assert np.array_equal(one, seg[0])
    # AlignPoints_KNN
    X = rng.normal(size=(500, 3)) * [3.0, 1.5, 0.4] + 20
    al = np.zeros((500, 3), order="F"); co = np.zeros(9); c3 = np.zeros(3)
    assert drv.drv_align_points_knn(_p(_d(X)), 500, 0, 0, _p(al), _p(co), _p(c3), e, 1024) == 0, e.value
    ral, rco, rc = pc.AlignPoints_KNN(X)
    assert np.array_equal(al, ral) and np.array_equal(co.reshape(3, 3, order="F"), rco) and np.array_equal(c3, np.asarray(rc).ravel())
    # getSpacialHistogramDescriptors (options struct with thVar vector and k = 'all')
    cloud = rng.uniform(0, 12, (6000, 3)); kp = rng.uniform(3, 9, (12, 3))
    opts = dict(min_pts=50, max_pts=6000, R=3.5, thVar=[1.0, 1.0], k="all", ALIGN_POINTS=True, VERBOSE=0)
    rfeat, rdesc = pc.getSpacialHistogramDescriptors(cloud, kp, opts)
    feat = np.zeros((12, 3), order="F"); desc = np.zeros((12, 980), order="F"); V = C.c_int()
    o6 = np.array([50, 6000, 3.5, 1.0, 1.0, 1], dtype=np.float64)
    assert drv.drv_descriptors(_p(_d(cloud)), 6000, _p(_d(kp)), 12, _p(o6), _p(feat), _p(desc), C.byref(V), e, 1024) == 0, e.value
    v = V.value
    assert v == len(rfeat) and v > 0
    assert np.array_equal(feat.ravel(order="F")[:3 * v].reshape(v, 3, order="F"), rfeat)
    assert np.array_equal(desc.ravel(order="F")[:980 * v].reshape(v, 980, order="F"), rdesc)
    assert drv.drv_live_arrays() == 0
    # `single` inputs keep their class (pcread clouds: completeExperimentFast.m:309) and equal the widened double run, rounded once
    Xs = X.astype(np.float32)
    al32 = np.zeros((500, 3), np.float32, order="F"); co32 = np.zeros(9, np.float32); c32 = np.zeros(3, np.float32)
    assert drv.drv_align_points_knn_f32(_p(np.asfortranarray(Xs), C.c_float), 500, _p(al32, C.c_float), _p(co32, C.c_float), _p(c32, C.c_float), e, 1024) == 0, e.value
    wal, wco, wc = pc.AlignPoints_KNN(Xs.astype(np.float64))
    assert np.array_equal(al32, wal.astype(np.float32)) and np.array_equal(co32.reshape(3, 3, order="F"), wco.astype(np.float32))
    pal, pco, pcc = pc.AlignPoints_KNN(Xs)                                      # the Python mirror dispatches on the class too
    assert pal.dtype == np.float32 and np.array_equal(pal, al32) and np.array_equal(pcc.ravel(), c32)
    # getSpacialHistogramDescriptors with single inputs: feat / desc come back DOUBLE whatever the input classes (the reference
    # preallocates them with nan(...): getSpacialHistogramDescriptors.m:61-62), every class combination goes through in its own class
    cl32, kp32 = cloud.astype(np.float32), kp.astype(np.float32)
    for ps, ss in ((1, 1), (1, 0), (0, 1)):
        pa = np.asfortranarray(cl32 if ps else cloud); ka = np.asfortranarray(kp32 if ss else kp)
        fo = np.zeros((12, 3), order="F"); do = np.zeros((12, 980), order="F")
        assert drv.drv_descriptors_classes(pa.ctypes.data_as(C.c_void_p), ps, 6000, ka.ctypes.data_as(C.c_void_p), ss, 12, _p(o6), _p(fo), _p(do),
                                           C.byref(V), e, 1024) == 0, e.value
        v = V.value
        pf, pd = pc.getSpacialHistogramDescriptors(cl32 if ps else cloud, kp32 if ss else kp, opts)      # the Python mirror: same dispatch
        assert pf.dtype == np.float64 and pd.dtype == np.float64 and v == len(pf) and v > 0
        assert np.array_equal(fo.ravel(order="F")[:3 * v].reshape(v, 3, order="F"), pf)
        assert np.array_equal(do.ravel(order="F")[:980 * v].reshape(v, 980, order="F"), pd)
    # the prepared-model commands: one handle, three surfaces
    rngm = np.random.default_rng(5)
    model = (rngm.random((9000, 3)) * [40, 30, 35]).astype(np.float32)
    surfs = [(model[rngm.choice(9000, 800, replace=False)] + rngm.normal(0, 0.03, (800, 3))).astype(np.float32) for _ in range(3)]
    flat = np.concatenate([np.asfortranarray(s_).ravel(order="F") for s_ in surfs]).astype(np.float32)
    pairs = np.zeros(3 * 800 * 2, dtype=np.uint32); Pn = (C.c_int * 3)()
    assert drv.drv_model_round_trip(_p(np.asfortranarray(model), C.c_float), 9000, _p(flat, C.c_float), 800, 3, C.c_float(0.25), C.c_float(0.8), 1,
                                    _p(pairs, C.c_uint32), Pn, e, 1024) == 0, e.value
    for k in range(3):
        got = pairs[k * 1600:k * 1600 + 2 * Pn[k]].reshape(Pn[k], 2, order="F")
        assert np.array_equal(got, pc.match_points(surfs[k], model, 0.25, 0.8, True)) and Pn[k] > 100
    assert drv.drv_live_arrays() == 0


@pytest.mark.gpu
def test_shim_multi_gpu_commands_one_worker(drv):
    """The spmd block of INTEGRATION.md section 3 with one worker, through the gateway: setDevice, commId (1 x 128 uint8),
    commInit, matchPointsSharded, ransacSharded, commDestroy; results equal the single-GPU entry points."""
    import pcreg_amd as pc
    rng = np.random.default_rng(8)
    model = (rng.random((30000, 3)) * [100, 56, 99]).astype(np.float32)
    pick = rng.choice(30000, 4000, replace=False)
    surf = (model[pick] + rng.normal(0, 0.03, (4000, 3))).astype(np.float32)
    want = pc.match_points(surf, model, 0.25, 0.8, True)
    p1 = surf[want[:, 0] - 1].astype(np.float64); p2 = model[want[:, 1] - 1].astype(np.float64)
    n = len(want)
    coef5 = np.array([3, 1500, 0.3, 0.08, 1], dtype=np.float64)
    pairs = np.zeros(4000 * 2, dtype=np.uint32); P = C.c_int()
    T = np.zeros(16); inl = np.zeros(n); ni, ns, mi, fl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    e = _err()
    rc = drv.drv_comm_round_trip(_p(np.asfortranarray(surf), C.c_float), 4000, _p(np.asfortranarray(model), C.c_float), 30000, C.c_float(0.25), C.c_float(0.8),
                                 _p(pairs, C.c_uint32), C.byref(P), _p(coef5), C.c_double(5), _p(T), _p(inl), C.byref(ni), C.byref(ns), C.byref(mi), C.byref(fl),
                                 _p(_d(p1)), _p(_d(p2)), n, e, 1024)
    assert rc == 0, e.value
    assert np.array_equal(pairs[:2 * P.value].reshape(P.value, 2, order="F"), want)
    coef = dict(minPtNum=3, iterNum=1500, thDist=0.3, thInlrRatio=0.08, REFINE=True, VERBOSE=0)
    Tr, inr, nsr, mir, _ = pc.ransac(p1, p2, coef, pc.estimateTransform, pc.calcDists, seed=5)
    assert fl.value == 0 and (ns.value, mi.value) == (nsr, mir) and np.array_equal(inl[:ni.value], np.asarray(inr, dtype=np.float64).ravel())
    assert np.linalg.norm(T.reshape(4, 4, order="F") - Tr) < 1e-12
    assert drv.drv_live_arrays() == 0


@pytest.mark.gpu
def test_shim_sphere_sweep(drv, oracle_py):
    """sphereCounts + descCreate + sphereSweep + descDestroy through the gateway (what matlab/sphereSweep.m calls) == the Python mirror
    of the same C entry points, output for output (row lists and trial numbers 1-based at the MATLAB boundary)."""
    import pcreg_amd as pc
    from test_gpu_sweep import _scene, PAR, OPT
    featM, descM, featS, descS = _scene()
    R, d_sph, min_pts, thresh, seed = 9.0, 6.0, 500, 60, 3
    centres = oracle_py.pcUniformSamples(featM, d_sph)
    counts = pc.sphereCounts(featM, centres, R)
    keep = counts >= min_pts
    with pc.DescSet(descS) as hS, pc.DescSet(descM) as hM:
        want = pc.sphereSweep(hS, hM, featS, featM, centres[keep], counts[keep], R, PAR, thresh, OPT, seed=seed)
    S, Q, VM, D, n_c = int(keep.sum()), featS.shape[0], featM.shape[0], descM.shape[1], len(centres)
    assert S >= 4 and len(want["trial"]) >= 1
    par7 = np.array([PAR["MatchThreshold"], PAR["MaxRatio"], PAR["Unique"], PAR["UNNORMALIZE"], PAR["norm_factor"], PAR["CHANGE_METRIC"], PAR["metric_factor"]], dtype=np.float64)
    coef5 = np.array([OPT["minPtNum"], OPT["iterNum"], OPT["thDist"], OPT["thInlrRatio"], OPT["REFINE"]], dtype=np.float64)
    cnt = np.zeros(n_c); rows1 = np.zeros(int(counts[keep].sum())); nrows = C.c_int()
    pairs = np.zeros(S * Q * 2, dtype=np.uint32); Pt = C.c_int(); npairs = np.zeros(S); S_out = C.c_int()
    trial1 = np.zeros(S); nt = C.c_int(); T = np.zeros(S * 16); ns = np.zeros(S); mi = np.zeros(S); fl = np.zeros(S)
    e = _err()
    rc = drv.drv_sphere_sweep(_p(_d(descS)), Q, _p(_d(descM)), VM, D, _p(_d(featS)), _p(_d(featM)), _p(_d(centres)), n_c, C.c_double(R), C.c_double(min_pts),
                              _p(par7), C.c_double(thresh), _p(coef5), C.c_double(seed), _p(cnt), _p(rows1), C.byref(nrows), _p(pairs, C.c_uint32), C.byref(Pt),
                              _p(npairs), C.byref(S_out), _p(trial1), C.byref(nt), _p(T), _p(ns), _p(mi), _p(fl), e, 1024)
    assert rc == 0, e.value
    assert np.array_equal(cnt, counts) and S_out.value == S
    assert np.array_equal(npairs, want["num_putative"]) and nrows.value == len(rows1)
    assert np.array_equal(rows1, np.concatenate(want["model_rows"]) + 1)
    allp = pairs[:2 * Pt.value].reshape(Pt.value, 2, order="F")
    assert np.array_equal(allp, np.vstack([m for m in want["matches"]]))
    n = nt.value
    assert np.array_equal(trial1[:n], want["trial"] + 1) and np.array_equal(ns[:n], want["statsSuccess"]) and np.array_equal(mi[:n], want["statsInliers"])
    for t in range(n):
        Tt = want["transforms"][t]
        assert bool(fl[t]) == (Tt is None)
        if Tt is not None:
            assert np.array_equal(T[16 * t:16 * t + 16].reshape(4, 4, order="F"), Tt)
    assert drv.drv_live_arrays() == 0


@pytest.mark.gpu
def test_shim_sphere_sweep_model_handle(drv, oracle_py):
    """sphereModelCreate (then descDestroy of the model set) + sphereSweepOnModel for two surfaces + sphereModelDestroy through the
    gateway (matlab/sphereSweepModel.m, sphereSweepOn.m) == sphereSweep on fresh sets, surface for surface."""
    import pcreg_amd as pc
    from test_gpu_sweep import _scene, PAR, OPT
    featM, descM, featS, descS = _scene()
    R, d_sph, min_pts, thresh, seed = 9.0, 6.0, 500, 60, 5
    centres = oracle_py.pcUniformSamples(featM, d_sph)
    counts = pc.sphereCounts(featM, centres, R)
    keep = counts >= min_pts
    rng = np.random.default_rng(8)
    descS2 = np.ascontiguousarray(descS[::-1]) * (1.0 + 0.02 * rng.random(descS.shape))
    featS2 = np.ascontiguousarray(featS[::-1])
    surfaces = [(featS, descS), (featS2, descS2)]
    want = []
    for f, d in surfaces:
        with pc.DescSet(d) as hS, pc.DescSet(descM) as hM:
            want.append(pc.sphereSweep(hS, hM, f, featM, centres[keep], counts[keep], R, PAR, thresh, OPT, seed=seed))
    S, Q, VM, D = int(keep.sum()), featS.shape[0], featM.shape[0], descM.shape[1]
    par7 = np.array([PAR["MatchThreshold"], PAR["MaxRatio"], PAR["Unique"], PAR["UNNORMALIZE"], PAR["norm_factor"], PAR["CHANGE_METRIC"], PAR["metric_factor"]], dtype=np.float64)
    coef5 = np.array([OPT["minPtNum"], OPT["iterNum"], OPT["thDist"], OPT["thInlrRatio"], OPT["REFINE"]], dtype=np.float64)
    U = len(surfaces)
    dS_all = np.concatenate([_d(d).ravel(order="K") for _, d in surfaces]); fS_all = np.concatenate([_d(f).ravel(order="K") for f, _ in surfaces])
    rows1 = np.zeros(int(counts[keep].sum())); nrows = C.c_int()
    pairs = np.zeros(U * S * Q * 2, dtype=np.uint32); Pt = (C.c_int * U)(); npairs = np.zeros(U * S)
    trial1 = np.zeros(U * S); nt = (C.c_int * U)(); T = np.zeros(U * S * 16); ns = np.zeros(U * S); mi = np.zeros(U * S); fl = np.zeros(U * S)
    nd = np.ascontiguousarray(counts[keep], dtype=np.int32)
    e = _err()
    rc = drv.drv_sphere_sweep_model(_p(dS_all), _p(fS_all), U, Q, _p(_d(descM)), VM, D, _p(_d(featM)), _p(_d(centres[keep])), S, _p(nd, C.c_int32), C.c_double(R),
                                    _p(par7), C.c_double(thresh), _p(coef5), C.c_double(seed), _p(rows1), C.byref(nrows), _p(pairs, C.c_uint32), Pt,
                                    _p(npairs), _p(trial1), nt, _p(T), _p(ns), _p(mi), _p(fl), e, 1024)
    assert rc == 0, e.value
    assert nrows.value == len(rows1) and np.array_equal(rows1, np.concatenate(want[0]["model_rows"]) + 1)
    for u in range(U):
        w = want[u]
        assert len(w["trial"]) >= 1
        assert np.array_equal(npairs[u * S:(u + 1) * S], w["num_putative"])
        allp = pairs[u * S * Q * 2:][:2 * Pt[u]].reshape(Pt[u], 2, order="F")
        assert np.array_equal(allp, np.vstack([m for m in w["matches"]]))
        n, o = nt[u], u * S
        assert np.array_equal(trial1[o:o + n], w["trial"] + 1) and np.array_equal(ns[o:o + n], w["statsSuccess"]) and np.array_equal(mi[o:o + n], w["statsInliers"])
        for t in range(n):
            Tt = w["transforms"][t]
            assert bool(fl[o + t]) == (Tt is None)
            if Tt is not None:
                assert np.array_equal(T[16 * (o + t):16 * (o + t) + 16].reshape(4, 4, order="F"), Tt)
    assert drv.drv_live_arrays() == 0


@pytest.mark.gpu
def test_shim_get_local_points_keeps_matlabs_classes(drv, oracle_py):
    """getLocalPoints through the gateway: a single cloud with a double centre is evaluated in single arithmetic and comes back
    single (getLocalPoints.m:8-25); [] when a gate fails."""
    rng = np.random.default_rng(3)
    pts = rng.uniform([0, 0, 0], [40, 30, 20], (5000, 3)).astype(np.float32)
    c = np.array([20.0, 15.0, 10.0])
    e = _err()
    out = np.zeros(5000 * 3); dd = np.zeros(5000); n = C.c_int(); cls = C.c_int()
    pf = np.asfortranarray(pts)
    args = (pf.ctypes.data_as(C.c_void_p), 1, 5000, C.c_double(4.0), c.ctypes.data_as(C.c_void_p), 0)
    tail = (_p(out), _p(dd), C.byref(n), C.byref(cls), e, 1024)
    assert drv.drv_get_local_points(*args, C.c_double(5), C.c_double(np.inf), *tail) == 0, e.value
    rp, rd = oracle_py.getLocalPoints(pts, 4.0, c, 5, np.inf, single_mode=2)
    assert cls.value == 1 and n.value == len(rp) > 50
    np.testing.assert_array_equal(out[:3 * n.value].reshape(n.value, 3, order="F"), rp)
    np.testing.assert_array_equal(dd[:n.value], rd)
    assert drv.drv_get_local_points(*args, C.c_double(len(rp) + 1), C.c_double(np.inf), *tail) == 0 and n.value == 0      # [] (:31)
    assert drv.drv_live_arrays() == 0
