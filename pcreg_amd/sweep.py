"""The sphere-sweep driver of completeExperimentFast.m:46-225 and its final stage (:291, :356-394)
as a device-resident pipeline.

Host logic (loops, thresholds, the stats arrays) mirrors the .m script line by line; everything that
touches descriptors or points runs through the device tier of include/pcreg.h on resident buffers:
the model's keypoints and descriptors are uploaded once, each sphere is an index list + a row gather
on the device, `getMatches` and `ransac` consume those buffers in place, and only counts and the
per-sphere results (pairs, 4x4 transforms) come back.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as _l
from ._lib import RansacOpts, check, lib
from .api import _match_opts, invertTF
from .device import DescriptorPipeline, _p, _stream


def pcUniformSamples(pts, d: float) -> np.ndarray:
    """completeExperimentFast.m:406-414: meshgrid over the cloud's limits, y fastest, then x, then z."""
    pts = np.asarray(pts, dtype=np.float64)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    ax = [lo[k] + d * np.arange(int(np.floor((hi[k] - lo[k]) / d + 1e-12)) + 1) for k in range(3)]
    X, Y, Z = np.meshgrid(ax[0], ax[1], ax[2])
    return np.column_stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")])


class SphereSweep:
    """featModel [VM,3] / descModel [VM,980] / featSurface [VS,3] / descSurface [VS,980]: numpy arrays or
    row-major CUDA tensors (e.g. straight from DescriptorPipeline.describe)."""

    def __init__(self, featModel, descModel, featSurface, descSurface, device: torch.device | None = None):
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        t = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))).to(self.dev).contiguous()
        self.featM, self.descM, self.featS, self.descS = t(featModel), t(descModel), t(featSurface), t(descSurface)
        self.VM, self.D = self.descM.shape
        self.VS = self.descS.shape[0]
        self.pipe = DescriptorPipeline(self.dev)
        L = lib()
        self._idx = torch.empty(max(self.VM, 1), dtype=torch.int32, device=self.dev)
        self._n = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._ws_sel = torch.empty(max(L.pcreg_dev_sphere_select_workspace(self.VM), 256), dtype=torch.uint8, device=self.dev)
        self._featM_host = None

    # -- :49-50
    def sphere_centres(self, d_spheres: float = 5.0) -> np.ndarray:
        if self._featM_host is None:
            self._featM_host = self.featM.cpu().numpy()
        return pcUniformSamples(self._featM_host, d_spheres)

    # -- :52-64
    def valid_spheres(self, centres: np.ndarray, R_desc: float, min_pts: int = 1400, max_pts: float = float("inf")):
        c = torch.from_numpy(np.ascontiguousarray(centres, dtype=np.float64)).to(self.dev)
        counts = torch.zeros(max(len(centres), 1), dtype=torch.int32, device=self.dev)
        check(lib().pcreg_dev_sphere_counts(_p(self.featM), self.VM, _p(c), len(centres), C.c_double(R_desc), _p(counts), _stream()))
        n = counts[:len(centres)].cpu().numpy().astype(np.int64)
        return (n >= min_pts) & (n <= max_pts), n

    def _select(self, centre, R: float) -> int:
        cc = (C.c_double * 3)(*[float(v) for v in centre])
        check(lib().pcreg_dev_sphere_select(_p(self.featM), self.VM, cc, C.c_double(R), _p(self._idx), _p(self._n), _p(self._ws_sel),
                                            C.c_size_t(self._ws_sel.numel()), _stream()))
        return int(self._n.item())

    # -- :109-149, one sphere
    def match_sphere(self, centre, R_desc: float, par: dict) -> dict:
        L = lib()
        n = self._select(centre, R_desc)                                   # getDescriptorMask, :118
        rows = self._idx[:n].clone()
        descCur = torch.empty((max(n, 1), self.D), dtype=torch.float64, device=self.dev)
        featCur = torch.empty((max(n, 1), 3), dtype=torch.float64, device=self.dev)
        if n > 0:
            check(L.pcreg_dev_gather_rows_f64(_p(self.descM), self.D, _p(rows), _p(self._n), n, _p(descCur), _stream()))   # :121-125
            check(L.pcreg_dev_gather_rows_f64(_p(self.featM), 3, _p(rows), _p(self._n), n, _p(featCur), _stream()))
        pairs, n_pairs = self.pipe.match(self.descS, self.VS, descCur, n, par) if n > 0 else (torch.zeros((1, 2), dtype=torch.int32, device=self.dev), None)
        P = int(n_pairs.item()) if n_pairs is not None else 0
        return dict(rows=rows, featCur=featCur, num_desc=n, pairs=pairs, num_putative=P)

    # -- :200-224, one promising sphere
    def ransac_sphere(self, m: dict, options: dict, seed: int = 0) -> dict:
        self.pipe.n_pairs.fill_(m["num_putative"])
        self.pipe.ransac(m["pairs"], self.featS, m["featCur"], options, seed=seed)
        return self.pipe.fetch_result()

    def run(self, par: dict, options: dict, R_desc: float, d_spheres: float = 5.0, min_pts: int = 1400,
            putative_thresh: int = 170, seed: int = 0) -> dict:
        """completeExperimentFast.m:46-224: centres, num_desc, num_putative, matches, model_rows, trial and the stats arrays."""
        centres = self.sphere_centres(d_spheres)
        valid, _ = self.valid_spheres(centres, R_desc, min_pts)
        centres = centres[valid]
        S = len(centres)
        per = [self.match_sphere(centres[i], R_desc, par) for i in range(S)]
        num_putative = np.array([m["num_putative"] for m in per], dtype=np.int64)
        num_desc = np.array([m["num_desc"] for m in per], dtype=np.int64)
        trial = np.nonzero(num_putative > putative_thresh)[0]                # :175
        sp, ss, si, sr, tf = [], [], [], [], []
        for t, i in enumerate(trial):
            r = self.ransac_sphere(per[i], options, seed=seed + t)
            sp.append(per[i]["num_putative"]); ss.append(r["numSuccess"]); si.append(r["maxInliers"])
            sr.append(100.0 * r["maxInliers"] / per[i]["num_putative"] if not r["failed"] else 0.0)
            tf.append(None if r["failed"] else r["T"])
        return dict(centres=centres, num_desc=num_desc, num_putative=num_putative,
                    matches=[m["pairs"][:m["num_putative"]].cpu().numpy().astype(np.uint32) for m in per],
                    model_rows=[m["rows"].cpu().numpy().astype(np.int64) for m in per], trial=trial,
                    statsPutative=np.array(sp, dtype=np.int64), statsSuccess=np.array(ss, dtype=np.int64),
                    statsInliers=np.array(si, dtype=np.int64), statsRatio=np.array(sr, dtype=np.float64), transforms=tf)


def quickTF_dev(pts_soa: torch.Tensor, TF: np.ndarray) -> torch.Tensor:
    """quickTF.m:5-7 on a resident [3, N] float64 cloud; TF 4x4 (numpy, row-vector convention)."""
    out = torch.empty_like(pts_soa)
    T = (C.c_double * 16)(*np.asarray(TF, dtype=np.float64).ravel(order="F"))
    check(lib().pcreg_dev_quick_tf(_p(pts_soa), pts_soa.shape[1], pts_soa.stride(0), T, _p(out), out.stride(0), _stream()))
    return out


def refine_by_distance_dev(pts1_soa: torch.Tensor, pts2_soa: torch.Tensor, n: int | torch.Tensor, maxDist: float):
    """completeExperimentFast.m:383-391 on resident [3, cap] float64 matched points: (T or None, #inliers)."""
    dev = pts1_soa.device
    cap = pts1_soa.shape[1]
    n_dev = n if isinstance(n, torch.Tensor) else torch.tensor([int(n)], dtype=torch.int32, device=dev)
    T16 = torch.zeros(16, dtype=torch.float64, device=dev)
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    check(lib().pcreg_dev_refine_by_distance(_p(pts1_soa), _p(pts2_soa), _p(n_dev), cap, pts1_soa.stride(0), C.c_double(maxDist),
                                             _p(T16), _p(info), _stream()))
    cnt, empty = (int(v) for v in info.cpu())
    return (None if empty else T16.cpu().numpy().reshape(4, 4, order="F")), cnt


__all__ = ["SphereSweep", "pcUniformSamples", "quickTF_dev", "refine_by_distance_dev", "invertTF"]
