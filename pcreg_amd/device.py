"""Device tier: the resident correspondence-search + RANSAC pipeline on torch tensors.

torch is plumbing only (device memory, streams, torch.distributed over RCCL); every
operation below is one of the pcreg_dev_* entry points of include/pcreg.h running
hand-written HIP kernels on the current stream.

    RegistrationPipeline     single GPU or model-sharded over the ranks of a process group
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from ._lib import DevRansacResult, RansacOpts, check, lib


def _p(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def soa(points: torch.Tensor) -> torch.Tensor:
    """N x 3 row-major -> 3 x N contiguous, i.e. MATLAB's column-major N x 3 (ld = N)."""
    return points.t().contiguous()


@dataclass
class MatchResult:
    idx: torch.Tensor        # [Q,2] int32 global model rows (0-based)
    dist: torch.Tensor       # [Q,2] float32 squared distances
    pairs: torch.Tensor      # [Q,2] uint32-as-int32 capacity buffer, first n_pairs rows valid (1-based)
    pts1: torch.Tensor       # [3,Q] float64 (column-major Q x 3): surface points of the pairs
    pts2: torch.Tensor       # [3,Q] float64: model points of the pairs
    n_pairs: torch.Tensor    # [1] int32 (device)


class RegistrationPipeline:
    """match (top-2 search -> threshold -> ratio -> Unique) then RANSAC, all resident.

    Sharded mode: every rank holds `model_shard` = rows
    [m_lo, m_lo + M_local) of the model and the full query set; one all_gather of the
    per-rank top-2 lists (Q*2*8 bytes per rank) is the only data-path collective of the
    search, one all_reduce(MAX) of the keep flags closes the Unique back-check.
    """

    def __init__(self, Q: int, M_local: int, m_lo: int = 0, M_total: int | None = None,
                 group=None, device: torch.device | None = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None or (dist.is_available() and dist.is_initialized())) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.Q, self.M_local, self.m_lo = Q, M_local, m_lo
        self.M_total = M_total if M_total is not None else M_local
        L = lib()
        i32, f32, f64 = torch.int32, torch.float32, torch.float64
        kw = dict(device=self.dev)
        self.ws_knn = torch.empty(max(L.pcreg_dev_knn2_points_f32_workspace(Q, M_local), 256), dtype=torch.uint8, **kw)
        self.ws_unq = torch.empty(max(L.pcreg_dev_unique_points_f32_workspace(Q), 256), dtype=torch.uint8, **kw)
        self.idx_local = torch.empty((Q, 2), dtype=i32, **kw)
        self.dist_local = torch.empty((Q, 2), dtype=f32, **kw)
        if self.world > 1:
            self.idx_all = torch.empty((self.world, Q, 2), dtype=i32, **kw)
            self.dist_all = torch.empty((self.world, Q, 2), dtype=f32, **kw)
        self.idx = torch.empty((Q, 2), dtype=i32, **kw)
        self.distm = torch.empty((Q, 2), dtype=f32, **kw)
        self.cand_q = torch.empty(Q, dtype=i32, **kw)
        self.cand_m = torch.empty(Q, dtype=i32, **kw)
        self.keep = torch.zeros(Q, dtype=i32, **kw)
        self.n_cand = torch.zeros(1, dtype=i32, **kw)
        self.n_pairs = torch.zeros(1, dtype=i32, **kw)
        self.pairs = torch.empty((Q, 2), dtype=i32, **kw)
        self.pts1 = torch.empty((3, Q), dtype=f64, **kw)
        self.pts2 = torch.empty((3, Q), dtype=f64, **kw)
        self.match_q = torch.empty((3, Q), dtype=f32, **kw)      # model coords of the candidates (sharded gather)
        self.inliers = torch.empty(Q, dtype=i32, **kw)
        self.result = torch.zeros(C.sizeof(DevRansacResult), dtype=torch.uint8, **kw)
        self.ws_ransac = None

    # -- search --------------------------------------------------------------------
    def search_local(self, q_soa: torch.Tensor, model_soa: torch.Tensor):
        """Top-2 of every query over THIS rank's model shard (the dominant kernel)."""
        L = lib()
        Q, M = self.Q, self.M_local
        check(L.pcreg_dev_knn2_points_f32(_p(q_soa), Q, q_soa.shape[1], _p(model_soa), M, model_soa.shape[1],
                                          C.c_int32(self.m_lo), _p(self.idx_local), _p(self.dist_local),
                                          _p(self.ws_knn), C.c_size_t(self.ws_knn.numel()), _stream()))

    def merge_ranks(self):
        """The search's only collective: all_gather of the per-rank lists, then the merge kernel."""
        if self.world == 1:
            self.idx, self.distm = self.idx_local, self.dist_local
            return
        self.dist.all_gather_into_tensor(self.idx_all, self.idx_local, group=self.group)
        self.dist.all_gather_into_tensor(self.dist_all, self.dist_local, group=self.group)
        check(lib().pcreg_dev_merge_top2_f32(_p(self.idx_all), _p(self.dist_all), self.world, self.Q, _p(self.idx),
                                             _p(self.distm), _stream()))

    def search(self, q_soa: torch.Tensor, model_soa: torch.Tensor):
        """Top-2 of every query over the whole (possibly sharded) model -> self.idx / self.distm."""
        self.search_local(q_soa, model_soa)
        self.merge_ranks()

    def match(self, q_soa: torch.Tensor, model_soa: torch.Tensor, thr_abs: float, max_ratio: float,
              unique: bool = True) -> MatchResult:
        self.search_local(q_soa, model_soa)
        return self.match_after_search(q_soa, model_soa, thr_abs, max_ratio, unique)

    def match_after_search(self, q_soa: torch.Tensor, model_soa: torch.Tensor, thr_abs: float, max_ratio: float,
                           unique: bool = True) -> MatchResult:
        L = lib()
        Q, M = self.Q, self.M_local
        self.merge_ranks()
        check(L.pcreg_dev_filter_top2_f32(_p(self.idx), _p(self.distm), Q, self.M_total, C.c_float(thr_abs),
                                          C.c_float(max_ratio), _p(self.cand_q), _p(self.cand_m), _p(self.n_cand),
                                          _stream()))
        keep_ptr = None
        if unique:
            self.keep.zero_()
            check(L.pcreg_dev_unique_points_f32(_p(q_soa), Q, q_soa.shape[1], _p(model_soa), M, model_soa.shape[1],
                                                C.c_int32(self.m_lo), _p(self.cand_q), _p(self.cand_m), _p(self.n_cand),
                                                _p(self.keep), _p(self.ws_unq), C.c_size_t(self.ws_unq.numel()), _stream()))
            if self.world > 1:      # each candidate was judged by the rank owning its model row
                self.dist.all_reduce(self.keep, op=self.dist.ReduceOp.MAX, group=self.group)
            keep_ptr = _p(self.keep)
        if self.world == 1:
            check(L.pcreg_dev_gather_pairs_f32(_p(q_soa), Q, q_soa.shape[1], _p(model_soa), model_soa.shape[1],
                                               _p(self.cand_q), _p(self.cand_m), keep_ptr, _p(self.n_cand),
                                               _p(self.pairs), _p(self.pts1), _p(self.pts2), _p(self.n_pairs), _stream()))
        else:
            # model coordinates of the candidates live on their owning rank: gather them
            # into a dense [3,Q] table indexed by candidate (zeros elsewhere), SUM-reduce,
            # then run the same gather kernel against that table with identity indices.
            self._gather_sharded(q_soa, model_soa, keep_ptr)
        return MatchResult(self.idx, self.distm, self.pairs, self.pts1, self.pts2, self.n_pairs)

    def _gather_sharded(self, q_soa, model_soa, keep_ptr):
        L = lib()
        Q = self.Q
        local = (self.cand_m >= self.m_lo) & (self.cand_m < self.m_lo + self.M_local)
        local &= torch.arange(Q, device=self.dev, dtype=torch.int32) < self.n_cand
        j = torch.where(local, self.cand_m - self.m_lo, torch.zeros_like(self.cand_m)).long()
        self.match_q.copy_(torch.where(local.unsqueeze(0), model_soa[:, j], torch.zeros((), device=self.dev)))
        self.dist.all_reduce(self.match_q, op=self.dist.ReduceOp.SUM, group=self.group)   # exact: one non-zero term
        ident = torch.arange(Q, device=self.dev, dtype=torch.int32)
        # pairs carry the global model row; coordinates come from the dense table
        check(L.pcreg_dev_gather_pairs_f32(_p(q_soa), Q, q_soa.shape[1], _p(self.match_q), Q, _p(self.cand_q),
                                           _p(ident), keep_ptr, _p(self.n_cand), None, _p(self.pts1), _p(self.pts2),
                                           _p(self.n_pairs), _stream()))
        check(L.pcreg_dev_gather_pairs_f32(_p(q_soa), Q, q_soa.shape[1], _p(self.match_q), Q, _p(self.cand_q),
                                           _p(self.cand_m), keep_ptr, _p(self.n_cand), _p(self.pairs), None, None,
                                           _p(self.n_pairs), _stream()))

    # -- ransac --------------------------------------------------------------------
    def ransac(self, coef: dict, seed: int = 0, n_dev: torch.Tensor | None = None, pts1=None, pts2=None,
               sample_idx: torch.Tensor | None = None):
        """Device-resident ransac on the matched pairs (or on explicit [3,cap] float64 tensors)."""
        L = lib()
        o = RansacOpts(int(coef["minPtNum"]), int(coef["iterNum"]), float(coef["thDist"]), float(coef["thInlrRatio"]),
                       int(bool(coef["REFINE"])), 0, int(seed))
        pts1 = self.pts1 if pts1 is None else pts1
        pts2 = self.pts2 if pts2 is None else pts2
        n_dev = self.n_pairs if n_dev is None else n_dev
        cap = pts1.shape[1]
        need = L.pcreg_dev_ransac_workspace(cap, o.iterNum)
        if self.ws_ransac is None or self.ws_ransac.numel() < need:
            self.ws_ransac = torch.empty(need, dtype=torch.uint8, device=self.dev)
        if self.inliers.numel() < cap:
            self.inliers = torch.empty(cap, dtype=torch.int32, device=self.dev)
        check(L.pcreg_dev_ransac(_p(pts1), _p(pts2), _p(n_dev), cap, cap, C.byref(o),
                                 _p(sample_idx) if sample_idx is not None else None, _p(self.result), _p(self.inliers),
                                 _p(self.ws_ransac), C.c_size_t(self.ws_ransac.numel()), _stream()))

    def fetch_result(self) -> dict:
        """D2H copy of the last ransac result (synchronises)."""
        import numpy as np
        raw = self.result.cpu().numpy().tobytes()
        r = DevRansacResult.from_buffer_copy(raw)
        T = np.array(r.T[:]).reshape(4, 4, order="F")
        inl = self.inliers[:r.n_inliers].cpu().numpy()
        return dict(T=T, n_inliers=r.n_inliers, numSuccess=r.num_success, maxInliers=r.max_inliers,
                    failed=bool(r.failed), n=r.n, winner=r.winner, inlierIdx=inl)
