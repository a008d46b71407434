"""The N > 1 path on CPU: world_size-2 gloo run of pcreg_amd.sharded.ShardedMatcher with
the ORACLE standing in for the kernels (tests may use the oracle; the product may not).
Checks that the all_gather + merge and the SUM-assembled [4, Q] table (Unique verdicts + matched
coordinates, column = query) give exactly what one process gets on the unsharded model."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleOps:
    """ShardedMatcher `ops` implemented with the oracle on CPU tensors (test double)."""

    def __init__(self, Q):
        self.Q = Q
        self.device = torch.device("cpu")

    @staticmethod
    def _np(t):
        return t.numpy()

    def local_top2(self, q, model, m_lo):
        from oracle import c_oracle
        idx, d = c_oracle.knn2_points_f32(q.numpy().T, model.numpy().T)
        idx = np.where(idx >= 0, idx + m_lo, -1).astype(np.int32)
        return torch.from_numpy(idx), torch.from_numpy(d)

    def merge_top2(self, idx_all, dist_all):
        ia, da = idx_all.numpy(), dist_all.numpy()
        R, Q, _ = ia.shape
        oi = np.full((Q, 2), -1, np.int32); od = np.full((Q, 2), np.inf, np.float32)
        for qi in range(Q):
            c = [(float(da[r, qi, k]), int(ia[r, qi, k])) for r in range(R) for k in range(2) if ia[r, qi, k] >= 0]
            c.sort()
            for k, (d, i) in enumerate(c[:2]):
                oi[qi, k], od[qi, k] = i, d
        return torch.from_numpy(oi), torch.from_numpy(od)

    @staticmethod
    def _filter(idx, d, M_total, thr, ratio):
        """matchFeatures' threshold + ratio test on a merged top-2 (fp32 arithmetic, like the kernel)."""
        i, dd = idx.numpy(), d.numpy()
        keep = (i[:, 0] >= 0) & (dd[:, 0] <= np.float32(thr))
        if M_total > 1:
            z = dd[:, 1] < np.float32(1e-6)
            t1 = np.where(z, np.float32(1), dd[:, 0]); t2 = np.where(z, np.float32(1), dd[:, 1])
            keep &= (t1 / t2).astype(np.float32) <= np.float32(ratio)
        return keep

    def _compact(self, q, cq, cm, p2rows):
        pairs = np.zeros((self.Q, 2), np.int32); p1 = np.zeros((3, self.Q)); p2 = np.zeros((3, self.Q))
        pairs[:len(cq), 0] = cq + 1; pairs[:len(cq), 1] = cm + 1
        p1[:, :len(cq)] = q.numpy()[:, cq].astype(np.float64)
        p2[:, :len(cq)] = p2rows.astype(np.float64)
        return torch.from_numpy(pairs), torch.from_numpy(p1), torch.from_numpy(p2), torch.tensor([len(cq)], dtype=torch.int32)

    def match_table(self, q, model, m_lo, M_total, idx, d, thr, ratio, unique):
        """This rank's [4, Q] contribution, column = query: coordinate bits of the query's nearest model point + its
        Unique verdict, only for candidates whose nearest point lives in this shard; zero elsewhere."""
        from oracle import c_oracle
        cand = self._filter(idx, d, M_total, thr, ratio)
        j = idx.numpy()[:, 0]
        M = model.shape[1]
        mine = cand & (j >= m_lo) & (j < m_lo + M)
        table = np.zeros((4, self.Q), np.int32)
        loc = np.nonzero(mine)[0]
        if len(loc):
            rows = model.numpy().T[j[loc] - m_lo]
            table[0:3, loc] = np.ascontiguousarray(rows.T).view(np.int32)
            verdict = np.ones(len(loc), np.int32)
            if unique:
                back, _ = c_oracle.knn2_points_f32(rows, q.numpy().T)       # first-best query of each matched model point
                verdict = (back[:, 0] == loc).astype(np.int32)
            table[3, loc] = verdict
        return torch.from_numpy(table)

    def match_from_table(self, q, M_total, idx, d, thr, ratio, table):
        cand = self._filter(idx, d, M_total, thr, ratio)
        t = table.numpy()
        cq = np.nonzero(cand & (t[3] != 0))[0]
        return self._compact(q, cq, idx.numpy()[cq, 0], np.ascontiguousarray(t[0:3, cq]).view(np.float32))

    def match_single(self, q, model, idx, d, thr, ratio, unique):
        M = model.shape[1]
        return self.match_from_table(q, M, idx, d, thr, ratio, self.match_table(q, model, 0, M, idx, d, thr, ratio, unique))


def _data():
    rng = np.random.default_rng(0)
    model = (rng.random((4000, 3)) * [100, 56, 99]).astype(np.float32)
    pick = rng.choice(4000, 900, replace=False)
    surf = (model[pick] + rng.normal(0, 0.4, (900, 3))).astype(np.float32)
    return model, surf


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcreg_amd.sharded import ShardedMatcher
    model, surf = _data()
    M_local = len(model) // world
    m_lo = rank * M_local
    shard = torch.from_numpy(np.ascontiguousarray(model[m_lo:m_lo + M_local].T))
    q = torch.from_numpy(np.ascontiguousarray(surf.T))
    sm = ShardedMatcher(OracleOps(len(surf)), len(surf), M_local, m_lo, len(model))
    pairs, p1, p2, n = sm.match(q, shard, 0.5, 0.8, unique=True)
    n = int(n)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), pairs=pairs.numpy()[:n], p1=p1.numpy()[:, :n], p2=p2.numpy()[:, :n],
             idx=sm.idx.numpy(), dist=sm.dist.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_match_equals_unsharded_world2(tmp_path):
    from oracle import c_oracle
    c_oracle.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    model, surf = _data()
    ref_pairs = c_oracle.match_points_f32(surf, model, 0.5, 0.8, True)
    ref_idx, ref_dist = c_oracle.knn2_points_f32(surf, model)
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_array_equal(z["idx"], ref_idx)                      # merged global top-2 on every rank
        np.testing.assert_array_equal(z["dist"], ref_dist)
        np.testing.assert_array_equal(z["pairs"].astype(np.uint32), ref_pairs)
        np.testing.assert_array_equal(z["p1"].T, surf[ref_pairs[:, 0] - 1].astype(np.float64))
        np.testing.assert_array_equal(z["p2"].T, model[ref_pairs[:, 1] - 1].astype(np.float64))
    assert len(ref_pairs) > 100


def test_world1_is_a_no_op_protocol():
    from pcreg_amd.sharded import ShardedMatcher
    from oracle import c_oracle
    model, surf = _data()
    sm = ShardedMatcher(OracleOps(len(surf)), len(surf), len(model), 0, len(model))
    pairs, p1, p2, n = sm.match(torch.from_numpy(np.ascontiguousarray(surf.T)), torch.from_numpy(np.ascontiguousarray(model.T)), 0.5, 0.8, True)
    np.testing.assert_array_equal(pairs.numpy()[:int(n)].astype(np.uint32), c_oracle.match_points_f32(surf, model, 0.5, 0.8, True))


# ---- one registration's hypotheses split over the ranks (pcreg_amd.sharded.gather_ransac_parts / combine_gathered_parts) ----
def _ransac_case():
    from conftest import rigid_case
    p1, p2, _ = rigid_case(400, 31, noise=0.02, outlier_frac=0.4)
    coef = dict(minPtNum=3, iterNum=501, thDist=0.05, thInlrRatio=0.1, REFINE=True, VERBOSE=0)     # 501: uneven shares
    return p1, p2, coef


def _ransac_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle, pcreg_oracle as o
    from pcreg_amd.sharded import PART_WORDS, combine_gathered_parts, gather_ransac_parts, hypothesis_share
    p1, p2, coef = _ransac_case()
    table = o.sample_table(len(p1), coef["iterNum"], 3, 17)                      # the global sampler stream
    begin, count = hypothesis_share(coef["iterNum"], rank, world)
    # this rank's share, by the oracle: counts of its hypotheses, local first maximum, its transform
    r = c_oracle.ransac(p1, p2, dict(coef, iterNum=count), sample_idx=table[begin:begin + count])
    counts = r["inlrNum_refined"]
    thInlr = o.matlab_round(coef["thInlrRatio"] * len(p1))
    w = int(np.argmax(counts))
    key = (int(counts[w]) << 32) | (0xFFFFFFFF - (begin + w))
    has = 0.0 if r["failed"] else 1.0
    T12 = np.zeros(12) if r["failed"] else np.concatenate([np.append(r["T"][:3, j], r["T"][3, j]) for j in range(3)])
    # this rank's pcreg_dev_ransac_part as 14 int64 words: key | (num_success, has) | T[12]
    part = np.zeros(PART_WORDS, dtype=np.int64)
    part[0:1].view(np.uint64)[0] = key
    part[1:2].view(np.int32)[:] = (int((counts >= thInlr).sum()), int(has))
    part[2:14].view(np.float64)[:] = T12
    allp = gather_ransac_parts(torch.from_numpy(part))
    k, ns, h, T = combine_gathered_parts(allp)
    np.savez(os.path.join(out_dir, f"ransac{rank}.npz"), key=np.array([k], dtype=np.uint64), ns=np.array([ns]), hT=np.concatenate([[float(h)], T]))
    dist.barrier()
    dist.destroy_process_group()


def test_ransac_hypotheses_split_over_two_ranks(tmp_path):
    from oracle import c_oracle, pcreg_oracle as o
    c_oracle.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_ransac_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p1, p2, coef = _ransac_case()
    ref = c_oracle.ransac(p1, p2, coef, sample_idx=o.sample_table(len(p1), coef["iterNum"], 3, 17))
    assert not ref["failed"]
    for r in range(2):
        z = np.load(tmp_path / f"ransac{r}.npz")
        key = int(z["key"][0])
        assert key >> 32 == ref["maxInliers"]
        assert 0xFFFFFFFF - (key & 0xFFFFFFFF) == int(np.argmax(ref["inlrNum_refined"]))     # FIRST maximum, global index
        assert int(z["ns"][0]) == ref["numSuccess"]
        assert z["hT"][0] == 1.0
        T = np.eye(4)
        for j in range(3):
            T[:3, j] = z["hT"][1 + 4 * j:4 + 4 * j]; T[3, j] = z["hT"][4 + 4 * j]
        np.testing.assert_array_equal(T, ref["T"])                                 # the winner's transform, exactly


def test_hypothesis_share_covers_everything():
    from pcreg_amd.sharded import hypothesis_share
    for iters in (1, 7, 8, 10000, 10001):
        for world in (1, 2, 3, 8):
            spans = [hypothesis_share(iters, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == iters
            pos = 0
            for b, c in spans:
                assert b == min(pos, iters) and c >= 0
                pos += c


# ---------------------------------------------------------------- crop-parallel batches (BASELINE cfg 5)
def _batch_worker(rank, world, port, out_dir, n_crops):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcreg_amd.batch import ROW, crops_of_rank, gather_rows
    mine = crops_of_rank(n_crops, rank, world)
    rows = torch.zeros((len(mine), ROW), dtype=torch.float64)
    for k, c in enumerate(mine):                      # a recognisable row per crop, stamped with the rank that made it
        rows[k] = torch.arange(ROW, dtype=torch.float64) * (c + 1)
        rows[k, 0] = c; rows[k, 1] = rank
    allr = gather_rows(rows, n_crops)
    np.save(os.path.join(out_dir, f"batch{rank}.npy"), allr)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_crops", [7, 8, 1])
def test_batch_rows_gather_in_crop_order_world2(tmp_path, n_crops):
    """Crops are dealt round-robin, every rank ends with every crop's row, in crop order (ragged tail included)."""
    from pcreg_amd.batch import ROW, crops_of_rank, rows_to_results
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_batch_worker, args=(2, port, str(tmp_path), n_crops), nprocs=2, join=True)
    a, b = np.load(tmp_path / "batch0.npy"), np.load(tmp_path / "batch1.npy")
    np.testing.assert_array_equal(a, b)
    assert a.shape == (n_crops, ROW)
    for c in range(n_crops):
        assert a[c, 0] == c and a[c, 1] == c % 2 and a[c, 5] == 5 * (c + 1)
    res = rows_to_results(a)
    assert [r["crop"] for r in res] == list(range(n_crops)) and res[0]["T"].shape == (4, 4)
    for world in (1, 2, 3, 8):
        dealt = sorted(c for r in range(world) for c in crops_of_rank(n_crops, r, world))
        assert dealt == list(range(n_crops))
