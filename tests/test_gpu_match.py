"""HIP correspondence search vs the oracle through the C ABI: index sets bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _clouds(Q, M, seed, box=(100.0, 56.0, 99.0)):
    rng = np.random.default_rng(seed)
    model = (rng.uniform(0, 1, (M, 3)) * np.array(box)).astype(np.float32)
    pick = rng.choice(M, min(Q, M), replace=Q > M)
    surf = (model[pick] + rng.normal(0, 0.05, (len(pick), 3))).astype(np.float32)
    if Q > len(pick):
        surf = np.vstack([surf, (rng.uniform(0, 1, (Q - len(pick), 3)) * np.array(box)).astype(np.float32)])
    return surf, model


@pytest.mark.parametrize("Q,M", [(1, 1), (5, 2), (100, 1), (1000, 3000), (1025, 4097), (4096, 20000), (3000, 70001)])
def test_knn2_points_bit_exact(Q, M, oracle_c):
    import pcreg_amd as pc
    q, m = _clouds(Q, M, Q + M)
    idx, dist = pc.knn2_points(q, m)
    ridx, rdist = oracle_c.knn2_points_f32(q, m)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)        # same bits (fmaf chain)


def test_knn2_points_ties_lowest_index(oracle_c):
    import pcreg_amd as pc
    rng = np.random.default_rng(0)
    m = rng.integers(0, 4, (5000, 3)).astype(np.float32)     # massive duplication -> ties everywhere
    q = rng.integers(0, 4, (300, 3)).astype(np.float32)
    idx, dist = pc.knn2_points(q, m)
    ridx, rdist = oracle_c.knn2_points_f32(q, m)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)


@pytest.mark.parametrize("Q,M,unique", [(2000, 9000, True), (2000, 9000, False), (5000, 1500, True)])
def test_match_points_pairs(Q, M, unique, oracle_c):
    import pcreg_amd as pc
    q, m = _clouds(Q, M, 17 + Q)
    pairs = pc.match_points(q, m, 0.5, 0.8, unique)
    ref = oracle_c.match_points_f32(q, m, 0.5, 0.8, unique)
    np.testing.assert_array_equal(pairs, ref)
    assert pairs.dtype == np.uint32 and (np.diff(pairs[:, 0].astype(np.int64)) > 0).all()


def _descs(Q, M, D, seed):
    rng = np.random.default_rng(seed)
    dM = rng.poisson(3.0, (M, D)).astype(np.float64)
    dS = rng.poisson(3.0, (Q, D)).astype(np.float64)
    k = min(Q, M) // 2
    dS[:k] = dM[rng.choice(M, k, replace=False)] + rng.poisson(0.2, (k, D))
    return dS, dM


PAR = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate",
           MatchThreshold=10, MaxRatio=0.99, Metric="SAD", Unique=True, VERBOSE=0)
"""completeExperimentFast.m:75-87."""


@pytest.mark.parametrize("Q,M,D", [(60, 200, 40), (300, 1000, 980), (130, 70, 17), (129, 65, 16), (1, 5, 8)])
@pytest.mark.parametrize("metric", ["SAD", "SSD"])
def test_get_matches(Q, M, D, metric, oracle_c):
    import pcreg_amd as pc
    dS, dM = _descs(Q, M, D, Q * 7 + D)
    par = dict(PAR, Metric=metric)
    got = pc.getMatches(dS, dM, par)
    ref = oracle_c.getMatches(dS, dM, par)
    np.testing.assert_array_equal(got, ref)
    assert got.dtype == np.uint32


def test_match_features_metric_and_options(oracle_c):
    import pcreg_amd as pc
    dS, dM = _descs(200, 500, 64, 3)
    for kw in (dict(Metric="SSD", MatchThreshold=1.0, MaxRatio=0.6, Unique=False),
               dict(Metric="SAD", MatchThreshold=10.0, MaxRatio=0.9, Unique=True),
               dict(Metric="SSD", MatchThreshold=100.0, MaxRatio=1.0, Unique=False, Prenormalized=True)):
        pairs, met = pc.matchFeatures(dS, dM, **kw)
        rp, rm = oracle_c.matchFeatures(dS, dM, kw)
        np.testing.assert_array_equal(pairs, rp)
        np.testing.assert_allclose(met, rm, rtol=0, atol=1e-14)


def test_get_matches_empty_and_verbose(capsys):
    import pcreg_amd as pc
    dS, dM = _descs(10, 20, 12, 1)
    out = pc.getMatches(dS[:0], dM, dict(PAR, VERBOSE=1))
    assert out.shape == (0, 2)
    assert "Calculated matches in" in capsys.readouterr().out                 # getMatches.m:58


@pytest.mark.parametrize("mode", ["fast", "exact", "fallback"])
@pytest.mark.parametrize("Q,M,D", [(700, 2500, 981), (257, 130, 33), (3, 2, 5)])
def test_sad_fast_path_is_the_exact_search(Q, M, D, mode, oracle_c, debug_set):
    """The certified u16 SAD path (match_sad16.hip), the exhaustive fp64 kernel and the
    all-queries-unproven fallback return the same pairs AND the same fp64 metric values."""
    import pcreg_amd as pc
    if mode == "exact":
        debug_set("match_exact")
    if mode == "fallback":
        debug_set("match_force_fallback")
    dS, dM = _descs(Q, M, D, Q + M + D)
    dM[M // 2] = dM[0]                                  # duplicate model rows: ties go to the lowest index
    if M > 100:
        dM[M - 1] = dM[7]; dS[5] = dM[7]
    for kw in (dict(Metric="SAD", MatchThreshold=10.0, MaxRatio=0.95, Unique=True),
               dict(Metric="SAD", MatchThreshold=100.0, MaxRatio=1.0, Unique=False)):
        pairs, met = pc.matchFeatures(dS, dM, **kw)
        rp, rm = oracle_c.matchFeatures(dS, dM, kw)
        np.testing.assert_array_equal(pairs, rp)
        np.testing.assert_array_equal(met, rm)


def test_sad_fast_path_constant_descriptors(oracle_c):
    """Zero range (all values equal after normalisation): every distance ties at 0."""
    import pcreg_amd as pc
    dS = np.full((40, 16), 2.0); dM = np.full((90, 16), 2.0)
    kw = dict(Metric="SAD", MatchThreshold=100.0, MaxRatio=1.0, Unique=False)
    pairs, met = pc.matchFeatures(dS, dM, **kw)
    rp, rm = oracle_c.matchFeatures(dS, dM, kw)
    np.testing.assert_array_equal(pairs, rp)
    np.testing.assert_array_equal(met, rm)


# ---------------------------------------------------------------- Unique back-check on the query grid (Q >= 4096)
def _unique_case(kind, Q, M, seed):
    rng = np.random.default_rng(seed)
    if kind == "volume":
        model = (rng.random((M, 3)) * [40, 30, 35]).astype(np.float32)
        surf = (model[rng.choice(M, Q, replace=False)] + rng.normal(0, 0.05, (Q, 3))).astype(np.float32)
    elif kind == "planar":                       # a flat sheet: one grid layer in z
        model = (rng.random((M, 3)) * [60, 60, 0]).astype(np.float32)
        surf = (model[rng.choice(M, Q, replace=False)] + rng.normal(0, 0.04, (Q, 3)) * [1, 1, 0]).astype(np.float32)
    elif kind == "duplicates":                   # repeated queries: equal distances, the lower index must win
        model = (rng.random((M, 3)) * [40, 30, 35]).astype(np.float32)
        base = (model[rng.choice(M, Q // 2, replace=False)] + rng.normal(0, 0.05, (Q // 2, 3))).astype(np.float32)
        surf = np.concatenate([base, base])[rng.permutation(2 * (Q // 2))]
    elif kind == "clusters":                     # tight clumps of queries: cells overflow, the exhaustive scan decides
        model = (rng.random((M, 3)) * [40, 30, 35]).astype(np.float32)
        ctr = model[rng.choice(M, Q // 32, replace=False)]
        surf = (np.repeat(ctr, 32, axis=0) + rng.normal(0, 0.01, (Q // 32 * 32, 3))).astype(np.float32)
    else:                                        # "far": sparse queries far from the model, loose threshold: big balls
        model = (rng.random((M, 3)) * [40, 30, 35]).astype(np.float32)
        surf = (rng.random((Q, 3)) * [40, 30, 35] + [0, 0, 3.0]).astype(np.float32)
    return surf, model


@pytest.mark.parametrize("kind,Q,M,thr,ratio", [
    ("volume", 6000, 30000, 0.25, 0.8), ("volume", 4096, 9000, 1e30, 1.0), ("planar", 5000, 20000, 0.25, 0.9),
    ("duplicates", 8000, 20000, 0.25, 1.0), ("clusters", 6400, 20000, 1.0, 1.0), ("far", 4500, 5000, 1e30, 1.0)])
def test_unique_grid_check_equals_oracle(kind, Q, M, thr, ratio, oracle_c):
    """launch_unique_points_f32's grid path: same pairs as the oracle's column-minimum Unique, ties included."""
    import pcreg_amd as pc
    surf, model = _unique_case(kind, Q, M, 5 + Q)
    ref = oracle_c.match_points_f32(surf, model, thr, ratio, True)
    loose = oracle_c.match_points_f32(surf, model, thr, ratio, False)
    got = pc.match_points(surf, model, thr, ratio, True)
    np.testing.assert_array_equal(got, ref)
    assert len(ref) > 0
    if kind in ("duplicates", "clusters"):
        assert len(ref) < len(loose)             # Unique really removed something


def test_getDescDist_known_answer_on_the_hip_path(oracle_py):
    """GPU twin of tests/test_oracle_kat.py::test_getDescDist_known_answer: pcreg_match_features' matchMetric output, on rows
    prepared with visualizeGTMatches.m:390-410's constant, equals the reference's own scalar restatement of the metric;
    pcreg_get_matches (which computes the constant itself, getMatches.m:24) returns the same pairs."""
    import pcreg_amd as pc
    from test_oracle_kat import desc_rows_l1_1600, getDescDist_literal
    D = 980
    dM = desc_rows_l1_1600(300, D, 11)
    dS = dM[np.random.default_rng(12).permutation(300)[:120]].copy()
    to = np.random.default_rng(13).integers(0, D, 120)
    for r in range(120):
        dS[r, int(np.argmax(dS[r]))] -= 3; dS[r, to[r]] += 3
    par = dict(Method="Exhaustive", Metric="SAD", MatchThreshold=100.0, MaxRatio=1.0, Unique=False, UNNORMALIZE=True,
               norm_factor=2, CHANGE_METRIC=True, metric_factor=0.45, VERBOSE=0)
    pS, pM = oracle_py.preprocess_descriptors(dS, dM, par)         # appended 3200, power 0.45
    pairs, met = pc.matchFeatures(pS, pM, Method="Exhaustive", MatchThreshold=100.0, MaxRatio=1.0, Metric="SAD", Unique=False)
    assert len(pairs) == 120
    for (i, j), d in zip(pairs.astype(int), met):
        assert abs(d - getDescDist_literal(dS[i - 1], dM[j - 1])) < 1e-13, (i, j)
    np.testing.assert_array_equal(pc.getMatches(dS, dM, par), pairs)


@pytest.mark.parametrize("par_over", [dict(), dict(Unique=False, UNNORMALIZE=False), dict(Metric="SSD", MatchThreshold=1.0, MaxRatio=0.8)])
def test_get_matches_on_resident_sets_equals_get_matches(par_over, oracle_c):
    """pcreg_desc_set_create + pcreg_get_matches_on_sets == pcreg_get_matches on the same rows, bit for bit
    (pairs AND the order), for row subsets, the whole set, an empty subset; the handles survive many calls."""
    import pcreg_amd as pc
    rng = np.random.default_rng(31)
    Q, M, D = 300, 1200, 980
    dM = rng.poisson(3.0, (M, D)).astype(np.float64)
    dS = dM[rng.choice(M, Q, replace=False)] + rng.poisson(0.15, (Q, D))
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate", MatchThreshold=10, MaxRatio=0.99,
               Metric="SAD", Unique=True, VERBOSE=0)
    par.update(par_over)
    with pc.DescSet(dS) as hS, pc.DescSet(dM) as hM:
        assert (hS.n, hS.D, hM.n) == (Q, D, M)
        for rows in (np.sort(rng.choice(M, 700, replace=False)), np.arange(0, M, 3), np.array([5]), None, np.zeros(0, np.int64)):
            got = pc.getMatchesOnSet(hS, hM, rows, par)
            if rows is not None and len(rows) == 0:
                assert got.shape == (0, 2)
                continue
            sub = dM if rows is None else dM[rows]
            want = pc.getMatches(dS, sub, par)
            np.testing.assert_array_equal(got, want)
            if par["Metric"] == "SAD":
                np.testing.assert_array_equal(want, oracle_c.getMatches(dS, sub, par))
        with pytest.raises(Exception):
            pc.getMatchesOnSet(hS, hM, np.array([M]), par)                 # out of range: an argument error, not a fault
        if par["Metric"] == "SAD":
            # all subsets in ONE call on the resident sets (pcreg_get_matches_segmented_on_sets) == the per-subset calls == the host form
            rows_list = [np.sort(rng.choice(M, 700, replace=False)), np.arange(0, M, 3), np.array([5]), np.zeros(0, np.int64), np.arange(M)]
            on = pc.getMatchesSegmentedOnSet(hS, hM, rows_list, par)
            host = pc.getMatchesSegmented(dS, dM, rows_list, par)
            for z, r in enumerate(rows_list):
                want = pc.getMatchesOnSet(hS, hM, r, par) if len(r) else np.zeros((0, 2), np.uint32)
                np.testing.assert_array_equal(on[z], want, err_msg=f"segment {z}")
                np.testing.assert_array_equal(host[z], want, err_msg=f"segment {z}")
            # the model set caches its powered rows for the options of the last call: other options must not see them
            for other in (dict(par, metric_factor=0.8, CHANGE_METRIC=True), dict(par, CHANGE_METRIC=False), par):
                on2 = pc.getMatchesSegmentedOnSet(hS, hM, rows_list[:2], other)
                host2 = pc.getMatchesSegmented(dS, dM, rows_list[:2], other)
                for a, b in zip(on2, host2):
                    np.testing.assert_array_equal(a, b)
            assert pc.getMatchesSegmentedOnSet(hS, hM, [], par) == []
            with pytest.raises(Exception):
                pc.getMatchesSegmentedOnSet(hS, hM, [np.array([M])], par)
        else:
            with pytest.raises(Exception):
                pc.getMatchesSegmentedOnSet(hS, hM, [np.arange(10)], par)  # SSD: refused, as by the host form


def test_match_stats_hook_counts_what_the_certificate_did(debug_set):
    """pcreg_debug_match_stats: with "match_stats" on, a getMatches call reports its queries (forward + the Unique back-search),
    the candidates it re-scored exactly, what stayed unproven and what went to the exhaustive kernel; forcing the fallback makes
    every query unproven; with the switch off the counters do not move."""
    import ctypes as C
    import pcreg_amd as pc
    from pcreg_amd._lib import check, lib
    rng = np.random.default_rng(3)
    Q, M, D = 400, 3000, 980
    dM = rng.poisson(3.0, (M, D)).astype(np.float64)
    dS = dM[rng.choice(M, Q, replace=False)] + rng.poisson(0.15, (Q, D))
    par = dict(UNNORMALIZE=True, norm_factor=2, CHANGE_METRIC=True, metric_factor=0.6, Method="Approximate", MatchThreshold=10, MaxRatio=0.99,
               Metric="SAD", Unique=True, VERBOSE=0)
    out = (C.c_longlong * 8)()
    def stats(reset=1):
        check(lib().pcreg_debug_match_stats(out, reset))
        return [int(v) for v in out]
    debug_set("match_stats")
    stats()
    want = pc.getMatches(dS, dM, par)
    s = stats()
    assert s[6] == 2 and s[0] >= Q and s[1] >= s[0] and s[2] <= s[0] // 10 and s[3] == s[2]        # forward + back call; few unproven
    debug_set("match_force_fallback")
    got = pc.getMatches(dS, dM, par)
    s2 = stats()
    np.testing.assert_array_equal(got, want)
    assert s2[2] == s2[0] and s2[3] == s2[0]                                                          # every query handed on
    debug_set("match_force_fallback", 0); debug_set("match_stats", 0)
    pc.getMatches(dS, dM, par)
    assert stats() == [0] * 8
