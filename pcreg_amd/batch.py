"""Crop-parallel batches: many surface crops registered against ONE model (BASELINE.json cfg 5, the shape of
completeExperimentFast.m's outer use: one CT model, a stack of stereo crops).

The unit of parallelism is the crop.  Every rank holds the whole model (12 MB for 1 M points: replicas, not
shards) and takes crops rank, rank + world, ...; there is NO data-path collective.  The only exchange is the
gather of the fixed-size result rows at the end (24 doubles per crop).

On one GPU the crops of a rank run round-robin on `n_streams` HIP streams, each with its own workspaces, so the
small latency-bound kernels of one crop's RANSAC overlap the chip-filling search of the next.

The kernels are the same C-ABI entry points bench.py times (pcreg_dev_knn2_points_f32 ... pcreg_dev_ransac);
without the HIP library this module raises on import of its pipelines: there is no CPU path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

ROW = 24      # crop, failed, n_surface, n_pairs, n_inliers, numSuccess, maxInliers, winner, T[16] (column-major 4x4)


def crops_of_rank(n_crops: int, rank: int, world: int) -> list[int]:
    """Crops `rank` registers: rank, rank + world, ... (interleaved, so ragged tails spread evenly)."""
    return list(range(rank, n_crops, world))


def gather_rows(local_rows: torch.Tensor, n_crops: int, group=None) -> np.ndarray:
    """All ranks' result rows -> [n_crops, ROW] float64 in crop order, identical on every rank.
    local_rows [k, ROW] float64 (row[0] = crop id).  One all_gather of ceil(n_crops / world) rows per rank."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    per = (n_crops + world - 1) // world
    pad = torch.full((per, ROW), -1.0, dtype=torch.float64, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    if world > 1:
        allr = torch.empty((world * per, ROW), dtype=torch.float64, device=local_rows.device)
        dist.all_gather_into_tensor(allr.view(-1), pad.view(-1), group=group)
    else:
        allr = pad
    allr = allr.cpu().numpy()
    out = np.full((n_crops, ROW), np.nan)
    for r in allr:
        if r[0] >= 0:
            out[int(r[0])] = r
    if np.isnan(out[:, 0]).any():
        raise RuntimeError("batch gather: crops missing from every rank: %s" % np.nonzero(np.isnan(out[:, 0]))[0][:8])
    return out


def rows_to_results(rows: np.ndarray) -> list[dict]:
    return [dict(crop=int(r[0]), failed=bool(r[1]), n_surface=int(r[2]), n_pairs=int(r[3]), n_inliers=int(r[4]),
                 numSuccess=int(r[5]), maxInliers=int(r[6]), winner=int(r[7]), T=r[8:24].reshape(4, 4, order="F").copy())
            for r in rows]


class BatchRegistration:
    """match + ransac of a list of surface crops against one resident model, this rank's share of them."""

    def __init__(self, model_soa: torch.Tensor, Q_cap: int, n_streams: int = 2, group=None,
                 device: torch.device | None = None):
        from .device import RegistrationPipeline, as_prepared
        self.dev = device or model_soa.device
        with torch.cuda.device(self.dev):
            self.model = as_prepared(model_soa)             # ONE prepared model for every crop and every stream
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.Q_cap = Q_cap
        M = self.model.M
        self.pipes = [RegistrationPipeline(Q_cap, M, device=self.dev, replica=True) for _ in range(max(1, n_streams))]
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in self.pipes]
        self.rows = None

    def run(self, surfaces: list[torch.Tensor], thr_abs: float, max_ratio: float, coef: dict, seed: int = 0,
            unique: bool = True, gather: bool = True):
        """surfaces[c]: [3, Q_cap] float32 SoA of crop c (every rank passes the same list; a rank only touches
        its own crops).  Returns the list of result dicts in crop order (all crops if `gather`, else this
        rank's).  Synchronises once, at the end."""
        import ctypes as C
        from ._lib import DevRansacResult
        mine = crops_of_rank(len(surfaces), self.rank, self.world)
        rs = C.sizeof(DevRansacResult)
        results = torch.zeros((max(len(mine), 1), rs), dtype=torch.uint8, device=self.dev)      # one struct per crop
        n_pairs = torch.zeros(max(len(mine), 1), dtype=torch.int32, device=self.dev)
        cur = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            s.wait_stream(cur)
        for k, c in enumerate(mine):
            pipe, st = self.pipes[k % len(self.pipes)], self.streams[k % len(self.pipes)]
            q = surfaces[c]
            if q.shape[1] != self.Q_cap:
                raise ValueError(f"crop {c}: {q.shape[1]} points, pipelines were sized for {self.Q_cap}")
            with torch.cuda.stream(st):
                pipe.result = results[k]                      # pcreg_dev_ransac writes the crop's struct in place
                pipe.match(q, self.model, thr_abs, max_ratio, unique)
                pipe.ransac(coef, seed=seed)
                n_pairs[k:k + 1].copy_(pipe.n_pairs)
        for s in self.streams:
            cur.wait_stream(s)
        raw = results.cpu().numpy()                             # the one synchronisation of the batch
        npr = n_pairs.cpu().numpy()
        rows = np.zeros((len(mine), ROW))
        for k, c in enumerate(mine):
            r = DevRansacResult.from_buffer_copy(raw[k].tobytes())
            rows[k, :8] = (c, r.failed, surfaces[c].shape[1], npr[k], r.n_inliers, r.num_success, r.max_inliers, r.winner)
            rows[k, 8:24] = r.T[:]
        self.rows = rows
        if gather and self.world > 1:
            return rows_to_results(gather_rows(torch.from_numpy(rows).to(self.dev), len(surfaces), self.group))
        return rows_to_results(rows)
