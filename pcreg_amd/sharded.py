"""Model-sharded correspondence search: the multi-GPU protocol of the hot path.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The model
rows are split in contiguous shards, one per rank; the (small) query set is replicated.
Exchanges per match:

  1. ONE all_gather of every rank's per-query top-2 (idx i32 and dist f32 in a single [2,Q,2] buffer: Q*16
     bytes per rank) followed by a local merge ordered by (dist, idx)  -> identical global top-2 everywhere;
  2. threshold / ratio filters run redundantly on every rank (deterministic); the Unique back-check of a query's
     nearest model point is owned by the rank that holds that point's row (the queries are replicated, so the
     column minimum is shard-local);
  3. ONE all_reduce(SUM) of a dense [4, Q] table of 4-byte words, column = QUERY: rows 0-2 = the coordinates of the
     query's nearest model point (float bit patterns), row 3 = the owner's Unique verdict (1 when Unique is off);
     a column is written by the owner and only for a query that passed the filters, zero elsewhere, so the integer
     sums are exact (x + 0) and both publish the verdicts and assemble the matched coordinates;
  4. filters again + ordered compaction from the summed table, on every rank.
  RANSAC with its hypotheses split over the ranks adds ONE all_gather of the 112-byte partial results
  (combine_gathered_parts / pcreg_dev_ransac_finish_parts).  Three collectives per registration in all.

The arithmetic is delegated to an `ops` object so that the protocol itself can be
exercised on CPU (gloo, world_size 2) with the oracle standing in for the kernels --
tests/test_sharded_cpu.py -- while production passes pcreg_amd.device.HipOps.

`ops` interface (tensors live on ops.device; `model` is whatever the ops object takes as this rank's shard):
    local_top2(q, model, m_lo)                     -> idx [Q,2] i32 (global rows), dist [Q,2] f32
    merge_top2(idx_all [R,Q,2], dist_all [R,Q,2])  -> idx [Q,2], dist [Q,2]
    match_single(q, model, idx, dist, thr, ratio, unique)                    (one rank: steps 2 and 4 in one)
    match_table(q, model, m_lo, M_total, idx, dist, thr, ratio, unique)      -> table [4,Q] i32 (this rank's contribution)
    match_from_table(q, M_total, idx, dist, thr, ratio, table)               (table after the SUM)
        the last three -> pairs [Q,2] i32 (1-based), pts1 [3,Q] f64, pts2 [3,Q] f64, n_pairs [1] i32
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


PART_WORDS = 14      # pcreg_dev_ransac_part as int64 words: key | (num_success, has) | T[12]


def gather_ransac_parts(part: torch.Tensor, group=None) -> torch.Tensor:
    """part [PART_WORDS] int64 (this rank's pcreg_dev_ransac_part) -> [world, PART_WORDS], identical on every rank.
    The one collective of a registration whose hypotheses were split over the ranks."""
    world = dist.get_world_size(group)
    allp = torch.empty((world, PART_WORDS), dtype=torch.int64, device=part.device)
    dist.all_gather_into_tensor(allp.view(-1), part.contiguous().view(-1), group=group)
    return allp


def combine_gathered_parts(allp: torch.Tensor):
    """What pcreg_dev_ransac_finish_parts does with the gathered parts, on the host (tests, non-HIP callers):
    key by MAX -- (inlier count << 32 | ~global hypothesis index), the first maximum of ransac.m:70-72 --,
    num_success by SUM, (has, T[12]) of the share whose key won.  Returns (key, num_success, has, T12)."""
    import numpy as np
    a = allp.cpu().numpy()
    keys = a[:, 0].astype(np.uint64)
    win = int(np.argmax(keys))
    ns_has = a[:, 1:2].copy().view(np.int32)                    # [R, 2]: num_success, has
    T12 = a[win, 2:14].copy().view(np.float64)
    key = int(keys[win])
    has = bool(ns_has[win, 1]) and key != 0
    return key, int(ns_has[:, 0].sum()), has, T12


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
        # one buffer [2, Q, 2] of 4-byte words per rank: indices, then the distances' bit patterns
        pack = getattr(self.ops, "top2_local", None)
        if pack is None:
            pack = torch.stack([idx_l.contiguous().view(torch.int32), dist_l.contiguous().view(torch.int32)])
        allp = torch.empty((self.world,) + tuple(pack.shape), dtype=torch.int32, device=pack.device)
        dist.all_gather_into_tensor(allp.view(-1), pack.view(-1), group=self.group)
        self.idx, self.dist = self.ops.merge_top2(allp[:, 0], allp[:, 1].view(torch.float32))     # rank stride 4 Q
        return self.idx, self.dist

    # -- steps 2-4 ----------------------------------------------------------------------
    def finish(self, q, model, thr_abs: float, max_ratio: float, unique: bool = True):
        ops = self.ops
        if not self.collective:
            return ops.match_single(q, model, self.idx, self.dist, thr_abs, max_ratio, unique)
        table = ops.match_table(q, model, self.m_lo, self.M_total, self.idx, self.dist, thr_abs, max_ratio, unique)
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=self.group)
        return ops.match_from_table(q, self.M_total, self.idx, self.dist, thr_abs, max_ratio, table)

    def match(self, q, model, thr_abs: float, max_ratio: float, unique: bool = True):
        self.search(q, model)
        return self.finish(q, model, thr_abs, max_ratio, unique)
