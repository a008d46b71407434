"""Device tier: the resident correspondence-search + RANSAC pipeline on torch tensors.

torch is plumbing only (device memory, streams, torch.distributed over RCCL); every
operation of HipOps is one pcreg_dev_* entry point of include/pcreg.h launching
hand-written HIP kernels on torch's current stream.  Layout: point sets are [3, N]
contiguous tensors, i.e. MATLAB's column-major N x 3 with ld = N.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import DevRansacResult, RansacOpts, check, lib
from . import _lib as _l
from .sharded import ShardedMatcher, gather_ransac_parts, hypothesis_share


def _p(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def soa(points: torch.Tensor) -> torch.Tensor:
    """N x 3 row-major -> [3, N] contiguous (column-major N x 3, ld = N)."""
    return points.t().contiguous()


class PreparedModel:
    """A model shard prepared ONCE for any number of searches (pcreg_dev_model_create: bounding box, matrix-core operand
    tiles, model-wide seeding grid) -- one model, many surfaces is the reference's shape (completeExperimentFast.m:131-149).
    Keeps the [3, M] float32 tensor alive; the handle is only valid while that tensor is unchanged."""

    def __init__(self, model_soa: torch.Tensor):
        if model_soa.dtype != torch.float32 or model_soa.dim() != 2 or model_soa.shape[0] != 3 or model_soa.stride(1) != 1:
            raise TypeError("a model is a [3, M] float32 tensor with contiguous rows (column-major M x 3)")
        self.tensor, self.version = model_soa, model_soa._version
        self.key = _model_key(model_soa)
        self.M, self.ld = int(model_soa.shape[1]), int(model_soa.stride(0)) if model_soa.shape[1] > 0 else 1
        self.handle = C.c_void_p()
        with torch.cuda.device(model_soa.device):
            check(lib().pcreg_dev_model_create(_p(model_soa), self.M, max(self.ld, 1), _stream(), C.byref(self.handle)))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            lib().pcreg_dev_model_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_dropped: list = []            # handles replaced by as_prepared(): freed later, not in the middle of a hot loop
_reprepared = 0


def release_dropped() -> None:
    """Free the prepared models that as_prepared() replaced (pcreg_dev_model_destroy synchronises the device: not something to
    do between two launches of a hot loop, which is where a replaced cache entry used to die)."""
    while _dropped:
        _dropped.pop().close()


def _model_key(t: torch.Tensor):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, t.device.index)


def as_prepared(model, cache: PreparedModel | None = None) -> PreparedModel:
    """model: a PreparedModel (used as is) or a [3, M] tensor -- prepared now, unless `cache` was prepared from the same memory
    (data pointer, shape, strides) and torch has seen no write to it since (`_version`).  A fresh VIEW of the same storage
    (`model[:, :M]`, `soa(x)` returning its input) therefore hits the cache.  Writes through raw pointers (this library's own
    ctypes calls, another stream) are invisible to `_version`: after such a write build a new PreparedModel yourself.
    A replaced entry is parked (release_dropped()) instead of being freed here; repeated re-preparation is reported once."""
    global _reprepared
    if isinstance(model, PreparedModel):
        return model
    if cache is not None and cache.handle.value and cache.key == _model_key(model):
        return cache
    if cache is not None:
        _dropped.append(cache)
        _reprepared += 1
        if _reprepared == 8:
            import warnings
            warnings.warn("pcreg_amd: a model tensor was re-prepared 8 times (box, operand tiles and seeding grid rebuilt per call); "
                          "pass a PreparedModel, or keep the model tensor unmodified between calls", RuntimeWarning, stacklevel=3)
        if len(_dropped) > 4:
            _dropped.pop(0).close()
    return PreparedModel(model)


class HipOps:
    """The `ops` backend of ShardedMatcher on an MI355X (preallocated, no host syncs).  Five launches per match on one
    rank: four for the search against the prepared model, one for filters + Unique + compaction."""

    def __init__(self, Q: int, M_local: int, device: torch.device):
        L = lib()
        self.device, self.Q, self.M_local = device, Q, M_local
        i32, f32, f64 = torch.int32, torch.float32, torch.float64
        kw = dict(device=device)
        self.ws = torch.empty(max(L.pcreg_dev_model_search_workspace(Q, M_local), 256), dtype=torch.uint8, **kw)
        self.top2_local = torch.empty((2, Q, 2), dtype=i32, **kw)        # one buffer: a single all_gather carries both
        self.idx_local = self.top2_local[0]
        self.dist_local = self.top2_local[1].view(f32)
        self.idx = torch.empty((Q, 2), dtype=i32, **kw)
        self.dist = torch.empty((Q, 2), dtype=f32, **kw)
        self.n_pairs = torch.zeros(1, dtype=i32, **kw)
        self.pairs = torch.empty((Q, 2), dtype=i32, **kw)
        self.pts1 = torch.empty((3, Q), dtype=f64, **kw)
        self.pts2 = torch.empty((3, Q), dtype=f64, **kw)
        self.table = torch.zeros((4, Q), dtype=i32, **kw)               # multi-GPU table, column = query (sharded.py step 3)
        self._prepared = None

    def prepared(self, model) -> PreparedModel:
        self._prepared = as_prepared(model, self._prepared)
        if self._prepared.M > self.M_local:
            raise ValueError(f"model of {self._prepared.M} rows, the pipeline was sized for {self.M_local}")
        return self._prepared

    def local_top2(self, q, model, m_lo):
        pm = self.prepared(model)
        check(lib().pcreg_dev_model_search_f32(pm.handle, _p(q), self.Q, q.stride(0), C.c_int32(m_lo), _p(self.idx_local), _p(self.dist_local),
                                               _p(self.ws), C.c_size_t(self.ws.numel()), _stream()))
        return self.idx_local, self.dist_local

    def merge_top2(self, idx_all, dist_all):
        check(lib().pcreg_dev_merge_top2_strided_f32(_p(idx_all), _p(dist_all), idx_all.shape[0], self.Q,
                                                     C.c_size_t(idx_all.stride(0)), _p(self.idx), _p(self.dist), _stream()))
        return self.idx, self.dist

    def match_single(self, q, model, idx, dist, thr, ratio, unique):
        pm = self.prepared(model)
        check(lib().pcreg_dev_model_match_f32(pm.handle, _p(q), self.Q, q.stride(0), _p(idx), _p(dist), C.c_float(thr), C.c_float(ratio),
                                              int(bool(unique)), _p(self.ws), C.c_size_t(self.ws.numel()), _p(self.pairs), _p(self.pts1),
                                              _p(self.pts2), _p(self.n_pairs), _stream()))
        return self.pairs, self.pts1, self.pts2, self.n_pairs

    def match_table(self, q, model, m_lo, M_total, idx, dist, thr, ratio, unique):
        pm = self.prepared(model)
        check(lib().pcreg_dev_model_match_table_f32(pm.handle, C.c_int32(m_lo), M_total, _p(q), self.Q, q.stride(0), _p(idx), _p(dist),
                                                    C.c_float(thr), C.c_float(ratio), int(bool(unique)), _p(self.ws),
                                                    C.c_size_t(self.ws.numel()), _p(self.table), _stream()))
        return self.table

    def match_from_table(self, q, M_total, idx, dist, thr, ratio, table):
        check(lib().pcreg_dev_match_from_table_f32(_p(q), self.Q, q.stride(0), M_total, _p(idx), _p(dist), C.c_float(thr), C.c_float(ratio),
                                                   _p(table), _p(self.ws), C.c_size_t(self.ws.numel()), _p(self.pairs), _p(self.pts1),
                                                   _p(self.pts2), _p(self.n_pairs), _stream()))
        return self.pairs, self.pts1, self.pts2, self.n_pairs


class RegistrationPipeline:
    """match (top-2 search -> threshold -> ratio -> Unique) then RANSAC, all resident in HBM."""

    def __init__(self, Q: int, M_local: int, m_lo: int = 0, M_total: int | None = None, group=None,
                 device: torch.device | None = None, replica: bool = False):
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.Q, self.M_local, self.m_lo = Q, M_local, m_lo
        self.M_total = M_total if M_total is not None else M_local
        self.ops = HipOps(Q, M_local, self.dev)
        self.matcher = ShardedMatcher(self.ops, Q, M_local, m_lo, self.M_total, group, replica=replica)
        self.world = self.matcher.world
        self.inliers = torch.empty(Q, dtype=torch.int32, device=self.dev)
        self.result = torch.zeros(C.sizeof(DevRansacResult), dtype=torch.uint8, device=self.dev)
        self.ws_ransac = None
        self.pairs = self.pts1 = self.pts2 = None
        self.n_pairs = self.ops.n_pairs
        self._local = None

    def search_local(self, q_soa, model_soa):
        """Top-2 of every query over THIS rank's model shard (the dominant kernel).  model_soa: the [3, M] tensor (prepared
        on first use and again whenever another tensor, or a modified one, is passed) or a PreparedModel."""
        self._local = self.ops.local_top2(q_soa, model_soa, self.m_lo)

    def match_after_search(self, q_soa, model_soa, thr_abs: float, max_ratio: float, unique: bool = True):
        self.matcher.merge_ranks(*self._local)
        self.pairs, self.pts1, self.pts2, self.n_pairs = self.matcher.finish(q_soa, model_soa, thr_abs, max_ratio, unique)
        return self.pairs, self.pts1, self.pts2, self.n_pairs

    def match(self, q_soa, model_soa, thr_abs: float, max_ratio: float, unique: bool = True):
        self.search_local(q_soa, model_soa)
        return self.match_after_search(q_soa, model_soa, thr_abs, max_ratio, unique)

    def ransac(self, coef: dict, seed: int = 0, n_dev=None, pts1=None, pts2=None, sample_idx=None):
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

    def ransac_sharded(self, coef: dict, seed: int = 0):
        """The same registration with its hypotheses split over the ranks of the group (each rank scores
        iterNum / world of them, one all_gather of the 112-byte partial results agrees the winner): identical result to
        ransac(), 1/world of its time.  Falls back to ransac() on one rank."""
        if not self.matcher.collective:
            return self.ransac(coef, seed)
        L = lib()
        o = RansacOpts(int(coef["minPtNum"]), int(coef["iterNum"]), float(coef["thDist"]), float(coef["thInlrRatio"]),
                       int(bool(coef["REFINE"])), 0, int(seed))
        cap = self.pts1.shape[1]
        begin, count = hypothesis_share(o.iterNum, self.matcher.rank, self.world)
        need = L.pcreg_dev_ransac_workspace(cap, max(count, 1))
        if self.ws_ransac is None or self.ws_ransac.numel() < need:
            self.ws_ransac = torch.empty(need, dtype=torch.uint8, device=self.dev)
        if self.inliers.numel() < cap:
            self.inliers = torch.empty(cap, dtype=torch.int32, device=self.dev)
        part = torch.zeros(14, dtype=torch.int64, device=self.dev)      # pcreg_dev_ransac_part: key | (ns, has) | T[12]
        check(L.pcreg_dev_ransac_partial(_p(self.pts1), _p(self.pts2), _p(self.n_pairs), cap, cap, C.byref(o), None,
                                         begin, count, _p(part), _p(self.ws_ransac), C.c_size_t(self.ws_ransac.numel()), _stream()))
        allp = gather_ransac_parts(part, self.matcher.group)             # the one collective: 112 bytes per rank
        check(L.pcreg_dev_ransac_finish_parts(_p(self.pts1), _p(self.pts2), _p(self.n_pairs), cap, cap, C.byref(o), _p(allp),
                                              allp.shape[0], _p(self.result), _p(self.inliers), _stream()))

    def fetch_result(self) -> dict:
        """D2H copy of the last ransac result (synchronises)."""
        import numpy as np
        r = DevRansacResult.from_buffer_copy(self.result.cpu().numpy().tobytes())
        T = np.array(r.T[:]).reshape(4, 4, order="F")
        return dict(T=T, n_inliers=r.n_inliers, numSuccess=r.num_success, maxInliers=r.max_inliers,
                    failed=bool(r.failed), n=r.n, winner=r.winner, inlierIdx=self.inliers[:r.n_inliers].cpu().numpy())


class U16Rows:
    """Descriptor rows as the kernel leaves them: uint16 counts [S, 980] in keypoint order + the ascending list of the
    surviving keypoints (pcreg_dev_spatial_histogram_descriptors_rows_u16)."""

    def __init__(self, rows: torch.Tensor, index: torch.Tensor):
        self.rows, self.index = rows, index

    def compact(self, V: int) -> torch.Tensor:
        """[V, 980] uint16: the survivors' rows in order (a copy; for inspection and tests)."""
        return self.rows.view(torch.int16)[self.index[:V].long()].view(torch.uint16)      # (torch cannot index uint16 tensors)


class DescriptorPipeline:
    """One sphere position of completeExperimentFast.m:131-213, resident in HBM:
    getSpacialHistogramDescriptors (surface + model) -> getMatches -> matched keypoints -> ransac.

    Point sets are [3, N] float64 tensors.  Descriptors stay on the device as row-major [V][980];
    the only host traffic is the keypoint counts (two int32 reads that size the match grid)."""

    ND = 980

    def __init__(self, device: torch.device | None = None):
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self._ws = {}
        self.result = torch.zeros(C.sizeof(DevRansacResult), dtype=torch.uint8, device=self.dev)
        self.n_pairs = torch.zeros(1, dtype=torch.int32, device=self.dev)

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = self._ws[key] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.dev)
        return t

    def describe(self, pts: torch.Tensor, sample_pts: torch.Tensor, options: dict, compact: bool = False, single_mode: int = 0):
        """describe_async + the read of the survivor count (one host synchronisation)."""
        feat, desc, counters = self.describe_async(pts, sample_pts, options, compact, single_mode)
        V, overflow = (int(v) for v in counters.cpu())
        if overflow:
            raise ValueError(f"a support holds {overflow} points: more than an LDS-resident support may have (lower max_pts)")
        return feat, desc, V

    def describe_async(self, pts: torch.Tensor, sample_pts: torch.Tensor, options: dict, compact: bool = False, single_mode: int = 0,
                       ws_key: str = "desc"):
        """-> (feat [S,3], desc, counters): enqueued, nothing read back.  counters (device int32 [2]) = the number V of surviving
        keypoints and the overflow flag; rows < V of feat are the survivors in sample order.  ws_key: callers that keep several
        descriptor stages in flight on different streams give each its own workspace.
        compact=False: desc [S,980] float64, rows < V compact (MATLAB's shape).
        compact=True: desc is a U16Rows -- the counts as uint16 rows in KEYPOINT order, written once by the kernel, plus the
        list of the V survivors (a quarter of the bytes and no staging copy; `match` takes it as it is).
        single_mode: 0 double data; 1 / 2 `single` arithmetic for the support (include/pcreg.h), compact form only."""
        from .api import _desc_opts
        L = lib()
        o = _desc_opts(options)
        P, S = pts.shape[1], sample_pts.shape[1]
        feat = torch.empty((S, 3), dtype=torch.float64, device=self.dev)
        counters = torch.zeros(2, dtype=torch.int32, device=self.dev)
        ws = self._workspace(ws_key, L.pcreg_dev_spatial_histogram_descriptors_workspace(P, S))
        if compact:
            rows = torch.empty((S, self.ND), dtype=torch.uint16, device=self.dev)
            index = torch.empty(S, dtype=torch.int32, device=self.dev)
            check(L.pcreg_dev_spatial_histogram_descriptors_rows_u16(_p(pts), P, pts.stride(0), _p(sample_pts), S, sample_pts.stride(0), C.byref(o),
                                                                     int(single_mode), _p(feat), _p(rows), _p(index), _p(counters), _p(ws),
                                                                     C.c_size_t(ws.numel()), _stream()))
            desc = U16Rows(rows, index)
        else:
            if single_mode:
                raise ValueError("single_mode is implemented for the compact (uint16 rows) form and for the host tier")
            desc = torch.empty((S, self.ND), dtype=torch.float64, device=self.dev)
            check(L.pcreg_dev_spatial_histogram_descriptors(_p(pts), P, pts.stride(0), _p(sample_pts), S, sample_pts.stride(0), C.byref(o), _p(feat),
                                                            _p(desc), _p(counters), _p(ws), C.c_size_t(ws.numel()), _stream()))
        return feat, desc, counters

    def match(self, descS, VS: int, descM, VM: int, par: dict, pairs_out: torch.Tensor | None = None,
              n_pairs_out: torch.Tensor | None = None, ws_cap: tuple | None = None):
        """-> (pairs [VS,2] int32 1-based, n_pairs device int32).  pairs_out / n_pairs_out: write into the caller's
        buffers (the batched sweep); ws_cap = (VS, VM) sizes the workspace once for a series of calls."""
        from .api import _match_opts
        L = lib()
        o = _match_opts(par)
        pairs = pairs_out if pairs_out is not None else torch.empty((max(VS, 1), 2), dtype=torch.int32, device=self.dev)
        n_pairs = n_pairs_out if n_pairs_out is not None else self.n_pairs
        cap = ws_cap or (VS, VM)
        if isinstance(descS, U16Rows) or isinstance(descM, U16Rows):        # uint16 rows + survivor lists on both sides
            if not (isinstance(descS, U16Rows) and isinstance(descM, U16Rows)):
                raise TypeError("uint16 rows on one side only")
            D = descS.rows.shape[1]
            ws = self._workspace("match", L.pcreg_dev_get_matches_workspace(cap[0], cap[1], D))
            check(L.pcreg_dev_get_matches_rows_u16(_p(descS.rows), _p(descS.index), VS, _p(descM.rows), _p(descM.index), VM, D, C.byref(o),
                                                   _p(pairs), None, _p(n_pairs), _p(ws), C.c_size_t(ws.numel()), _stream()))
        else:
            D = descS.shape[1]
            ws = self._workspace("match", L.pcreg_dev_get_matches_workspace(cap[0], cap[1], D))
            check(L.pcreg_dev_get_matches(_p(descS), VS, D, _p(descM), VM, D, D, _l.LAYOUT_ROW_MAJOR,
                                          C.byref(o), _p(pairs), None, _p(n_pairs), _p(ws), C.c_size_t(ws.numel()),
                                          _stream()))
        return pairs, n_pairs

    def ransac(self, pairs: torch.Tensor, featS: torch.Tensor, featM: torch.Tensor, coef: dict, seed: int = 0):
        """ransac on featS(pairs(:,1),:) / featM(pairs(:,2),:); result via fetch_result()."""
        L = lib()
        cap = pairs.shape[0]
        self.pts1 = torch.empty((3, cap), dtype=torch.float64, device=self.dev)
        self.pts2 = torch.empty((3, cap), dtype=torch.float64, device=self.dev)
        check(L.pcreg_dev_gather_matched_rows(_p(pairs), _p(self.n_pairs), cap, _p(featS), _p(featM), _p(self.pts1),
                                              _p(self.pts2), _stream()))
        o = RansacOpts(int(coef["minPtNum"]), int(coef["iterNum"]), float(coef["thDist"]), float(coef["thInlrRatio"]),
                       int(bool(coef["REFINE"])), 0, int(seed))
        ws = self._workspace("ransac", L.pcreg_dev_ransac_workspace(cap, o.iterNum))
        self.inliers = torch.empty(cap, dtype=torch.int32, device=self.dev)
        check(L.pcreg_dev_ransac(_p(self.pts1), _p(self.pts2), _p(self.n_pairs), cap, cap, C.byref(o), None,
                                 _p(self.result), _p(self.inliers), _p(ws), C.c_size_t(ws.numel()), _stream()))

    fetch_result = RegistrationPipeline.fetch_result
