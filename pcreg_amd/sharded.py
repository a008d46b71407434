"""Model-sharded correspondence search: the multi-GPU protocol of the hot path.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The model
rows are split in contiguous shards, one per rank; the (small) query set is replicated.
Exchanges per match:

  1. all_gather of every rank's per-query top-2 (idx i32 + dist f32: Q*16 bytes per rank)
     followed by a local merge ordered by (dist, idx)  -> identical global top-2 everywhere;
  2. threshold / ratio filters run redundantly on every rank (deterministic);
  3. the Unique back-check of a candidate is owned by the rank that holds its model row
     (the queries are replicated, so the column minimum is shard-local); one
     all_reduce(MAX) over the keep flags publishes the verdicts;
  4. the matched model coordinates are assembled with one all_reduce(SUM) of a dense
     [3, Q] table in which exactly one rank contributes each column (exact: x + 0).

The arithmetic is delegated to an `ops` object so that the protocol itself can be
exercised on CPU (gloo, world_size 2) with the oracle standing in for the kernels --
tests/test_sharded_cpu.py -- while production passes pcreg_amd.device.HipOps.

`ops` interface (tensors live on ops.device):
    local_top2(q, model, m_lo)                     -> idx [Q,2] i32 (global rows), dist [Q,2] f32
    merge_top2(idx_all [R,Q,2], dist_all [R,Q,2])  -> idx [Q,2], dist [Q,2]
    filter_top2(idx, dist, M_total, thr, ratio)    -> cand_q [Q] i32, cand_m [Q] i32, n_cand [1] i32
    unique_local(q, model, m_lo, cand_q, cand_m, n_cand) -> keep [Q] i32 (1/0; 0 for rows of other shards)
    gather_pairs(q, table, table_is_dense, cand_q, cand_m, keep|None, n_cand)
                                                   -> pairs [Q,2] i32 (1-based), pts1 [3,Q] f64, pts2 [3,Q] f64, n_pairs [1] i32
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def hypothesis_share(iter_num: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, begin + count) of a registration's hypotheses that `rank` evaluates."""
    share = (iter_num + world - 1) // world
    begin = min(rank * share, iter_num)
    return begin, min(share, iter_num - begin)


def combine_ransac_parts(key: torch.Tensor, num_success: torch.Tensor, has_T: torch.Tensor, group=None):
    """Cross-rank winner of one registration whose hypotheses were split over the ranks:
    key [1] int64 = (inlier count << 32 | ~global hypothesis index) by MAX -- the first maximum of
    ransac.m:70-72 --, num_success [1] int64 by SUM, has_T [13] float64 = (has, T[12]) taken from the
    one rank whose key won (every other rank contributes zeros, so the SUM is exact).
    Three tiny collectives; returns (key, num_success, has_T) identical on every rank."""
    local_key = key.clone()
    dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(num_success, op=dist.ReduceOp.SUM, group=group)
    own = (local_key == key) & (key != 0)
    has_T = torch.where(own, has_T, torch.zeros_like(has_T))
    dist.all_reduce(has_T, op=dist.ReduceOp.SUM, group=group)
    return key, num_success, has_T


class ShardedMatcher:
    def __init__(self, ops, Q: int, M_local: int, m_lo: int, M_total: int, group=None, replica: bool = False):
        self.ops, self.Q, self.M_local, self.m_lo, self.M_total = ops, Q, M_local, m_lo, M_total
        self.group = group
        # replica: this rank holds the WHOLE model and works alone (crop-parallel batches, pcreg_amd/batch.py)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized() and not replica) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # PCREG_FORCE_COLLECTIVES=1: run the collective path even on a one-rank group (rehearses every RCCL call
        # of the N > 1 protocol -- dtypes, shapes, in-place views -- on a single GPU)
        self.collective = self.world > 1 or (os.environ.get("PCREG_FORCE_COLLECTIVES") == "1" and not replica
                                             and dist.is_available() and dist.is_initialized())
        self.idx = self.dist = None

    # -- step 1 -----------------------------------------------------------------------
    def search(self, q, model):
        idx_l, dist_l = self.ops.local_top2(q, model, self.m_lo)
        return self.merge_ranks(idx_l, dist_l)

    def merge_ranks(self, idx_l, dist_l):
        if not self.collective:
            self.idx, self.dist = idx_l, dist_l
            return idx_l, dist_l
        idx_all = torch.empty((self.world,) + tuple(idx_l.shape), dtype=idx_l.dtype, device=idx_l.device)
        dist_all = torch.empty((self.world,) + tuple(dist_l.shape), dtype=dist_l.dtype, device=dist_l.device)
        # flat views: the concatenated 1-D form is what every backend (RCCL, gloo) accepts
        dist.all_gather_into_tensor(idx_all.view(-1), idx_l.contiguous().view(-1), group=self.group)
        dist.all_gather_into_tensor(dist_all.view(-1), dist_l.contiguous().view(-1), group=self.group)
        self.idx, self.dist = self.ops.merge_top2(idx_all, dist_all)
        return self.idx, self.dist

    # -- steps 2-4 ----------------------------------------------------------------------
    def finish(self, q, model, thr_abs: float, max_ratio: float, unique: bool = True):
        ops = self.ops
        cand_q, cand_m, n_cand = ops.filter_top2(self.idx, self.dist, self.M_total, thr_abs, max_ratio)
        keep = None
        if unique:
            keep = ops.unique_local(q, model, self.m_lo, cand_q, cand_m, n_cand)
            if self.collective:
                dist.all_reduce(keep, op=dist.ReduceOp.MAX, group=self.group)
        if not self.collective:
            return ops.gather_pairs(q, model, False, cand_q, cand_m, keep, n_cand)
        # dense [3,Q] table of the candidates' model coordinates: column k is written by
        # the rank owning row cand_m[k], zero elsewhere
        Q = self.Q
        ar = torch.arange(Q, device=cand_m.device, dtype=torch.int32)
        local = (cand_m >= self.m_lo) & (cand_m < self.m_lo + self.M_local) & (ar < n_cand)
        j = torch.where(local, cand_m - self.m_lo, torch.zeros_like(cand_m)).long()
        table = torch.where(local.unsqueeze(0), model[:, j], torch.zeros((), dtype=model.dtype, device=model.device))
        table = table.contiguous()
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=self.group)
        return ops.gather_pairs(q, table, True, cand_q, cand_m, keep, n_cand)

    def match(self, q, model, thr_abs: float, max_ratio: float, unique: bool = True):
        self.search(q, model)
        return self.finish(q, model, thr_abs, max_ratio, unique)
