"""The sphere-sweep driver of completeExperimentFast.m:46-225 and its final stage (:291, :356-394)
as a device-resident pipeline.

Everything that touches descriptors or points runs through the device tier of include/pcreg.h on resident
buffers: the model's keypoints and descriptors are uploaded once, each sphere is an index list + a row gather
on the device, `getMatches` and `ransac` consume those buffers in place.  `SphereSweep.run` is the batched form
(the reference's two parfor loops as one enqueue chain, TWO host synchronisations for the whole sweep);
`run_serial` walks the spheres one at a time like the script's inner loops and is kept as its cross-check.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as _l
from ._lib import RansacOpts, check, lib
from .api import _match_opts, invertTF
from .device import DescriptorPipeline, _p, _stream


def pcUniformSamples(pts, d: float, limits=None) -> np.ndarray:
    """completeExperimentFast.m:406-414: meshgrid over the cloud's limits, y fastest, then x, then z.  `limits` = (min, max) per
    axis if the caller already has them."""
    if limits is None:
        pts = np.asarray(pts, dtype=np.float64)
        limits = (pts.min(axis=0), pts.max(axis=0))
    lo, hi = limits
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
        self._limits = None

    def set_surface(self, featSurface, descSurface) -> None:
        """The next surface against the SAME model (completeExperimentFast.m runs once per surface crop): everything run() keeps
        for the model -- its spheres, the powered rows of its descriptor set, the workspaces -- stays."""
        t = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))).to(self.dev).contiguous()
        fS, dS = t(featSurface), t(descSurface)
        if dS.shape[1] != self.D or fS.shape[0] != dS.shape[0]:
            raise ValueError("surface descriptors must have the model's length and one row per surface keypoint")
        self.featS, self.descS = fS, dS
        self.VS = dS.shape[0]

    def release(self) -> None:
        """Drop the cached workspaces (the segmented chain's ~1.4 GB, the batched ransac's, the per-stream pipelines of
        run_streams()); the next run allocates them again.  The descriptor sets stay resident."""
        self._seg_ws = None
        self._rs_ws = None
        if hasattr(self, "_lanes"):
            del self._lanes
        self.pipe = DescriptorPipeline(self.dev)
        torch.cuda.empty_cache()

    # -- :49-50
    def sphere_centres(self, d_spheres: float = 5.0) -> np.ndarray:
        """The model is fixed for the life of this object (one model, many surfaces: completeExperimentFast.m's shape): the limits
        of its keypoints are taken once (numpy's column-wise min / max of 60 000 x 3 rows: ~2 ms of host time that a 19-ms sweep
        would otherwise spend with the GPU idle); the mesh itself is rebuilt per call."""
        if self._limits is None:
            if self._featM_host is None:
                self._featM_host = self.featM.cpu().numpy()
            self._limits = (self._featM_host.min(axis=0), self._featM_host.max(axis=0))
        return pcUniformSamples(None, d_spheres, self._limits)

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

    def run_serial(self, par: dict, options: dict, R_desc: float, d_spheres: float = 5.0, min_pts: int = 1400,
                   putative_thresh: int = 170, seed: int = 0) -> dict:
        """The script's loops one sphere at a time (two host reads per sphere): kept as the plain reading of
        completeExperimentFast.m:46-224 and as the cross-check of run()."""
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

    def run(self, par: dict, options: dict, R_desc: float, d_spheres: float = 5.0, min_pts: int = 1400,
            putative_thresh: int = 170, seed: int = 0) -> dict:
        """completeExperimentFast.m:46-224 with every per-sphere loop as ONE launch chain over all spheres, TWO host
        synchronisations for the first sweep with a set of sphere parameters and ONE for every later one (the spheres -- centres,
        row lists, gathered keypoints -- depend on the model only and are kept).

        (sync 1) the descriptor counts of all candidate centres fix the valid spheres and every buffer size (:52-64).
        Then, with nothing read back in between: getDescriptorMask of every valid sphere as one launch writing the row
        lists and featCur back to back (:109-125, pcreg_dev_sphere_select_batched); getMatches of the surface against every
        sphere's rows as ONE segmented chain of ~18 launches (:131-149, pcreg_dev_get_matches_segmented: the powered
        columns are computed once, the appended constant and the row norms per sphere); the putative threshold as a device
        list of trial spheres (:175, pcreg_dev_sweep_plan); the matched keypoints of all trials packed back to back
        (:205-206, pcreg_dev_sweep_gather); ONE batched ransac launch over the trial capacity (:201-216,
        pcreg_dev_ransac_batched, seed + trial ordinal).  (sync 2) counts, pairs, rows and result structs come back
        together.  Same outputs as run_streams() / run_serial(); SSD (never used by the reference's drivers) takes
        run_streams()."""
        from ._lib import DevRansacResult
        if str(par.get("Metric", "SSD")).upper() != "SAD":
            return self.run_streams(par, options, R_desc, d_spheres, min_pts, putative_thresh, seed)
        L = lib()
        dev = self.dev
        i32, f64 = torch.int32, torch.float64
        VS, sp = self.VS, _stream()
        # The spheres are a property of the MODEL (its keypoints, R_desc, the spacing, min_pts): their centres, row lists and gathered
        # keypoints are made at the first sweep with these parameters and kept -- one model, many surfaces.  That first sweep has the
        # script's two host synchronisations (the counts fix every size); the later ones have ONE, at the end.
        skey = (float(R_desc), float(d_spheres), int(min_pts))
        sph = self.__dict__.setdefault("_spheres", {}).get(skey)
        if sph is None:
            centres = self.sphere_centres(d_spheres)
            valid, counts = self.valid_spheres(centres, R_desc, min_pts)                     # ---- sync 1 (first sweep only)
            centres = centres[valid]
            num_desc = counts[valid].astype(np.int64)
            S = len(centres)
            sph = dict(centres=centres, num_desc=num_desc, S=S)
            if S:
                row_off = np.zeros(S + 1, dtype=np.int64); row_off[1:] = np.cumsum(num_desc)
                tot, n_max = int(row_off[-1]), int(num_desc.max())
                if tot >= 2**31:
                    raise ValueError("sphere sweep: more than 2^31 rows over all spheres")
                seg_off = torch.from_numpy(row_off.astype(np.int32)).to(dev)
                cen = torch.from_numpy(np.ascontiguousarray(centres, dtype=np.float64)).to(dev)
                roff_dev = torch.from_numpy(row_off[:S].copy()).to(dev)
                rows_all = torch.empty(tot, dtype=i32, device=dev)
                feat_all = torch.empty((tot, 3), dtype=f64, device=dev)
                n_sel = torch.zeros(S, dtype=i32, device=dev)
                check(L.pcreg_dev_sphere_select_batched(_p(self.featM), self.VM, _p(cen), S, C.c_double(R_desc), _p(seg_off), _p(rows_all), _p(feat_all),
                                                        _p(n_sel), sp))                                                     # :109-125
                assert np.array_equal(n_sel.cpu().numpy(), num_desc), "sphere_select disagrees with sphere_counts"
                rows_host = rows_all.cpu().numpy().astype(np.int64)
                # model rows that lie in NO sphere never meet the surface: the matcher gets the model set restricted to the union of the
                # spheres' rows (the lists renumbered into it; a pair's model index counts inside its sphere, so nothing else changes)
                union = np.unique(rows_host)
                if len(union) < 0.95 * self.VM:
                    desc_u = self.descM[torch.from_numpy(union).to(dev)].contiguous()
                    rows_u = torch.from_numpy(np.searchsorted(union, rows_host).astype(np.int32)).to(dev)
                else:
                    desc_u, rows_u = self.descM, rows_all
                sph.update(row_off=row_off, tot=tot, n_max=n_max, seg_off=seg_off, roff_dev=roff_dev, rows_all=rows_all, feat_all=feat_all,
                           model_rows=[rows_host[row_off[i]:row_off[i + 1]] for i in range(S)], desc_u=desc_u, rows_u=rows_u, n_union=int(len(union)))
            # an entry holds the restricted model set (hundreds of MB at the reference's shape): keep the two most recent parameter sets
            while len(self._spheres) >= 2:
                self._spheres.pop(next(iter(self._spheres)))
            self._spheres[skey] = sph
        centres, num_desc, S = sph["centres"], sph["num_desc"], sph["S"]
        if S == 0:
            return dict(centres=centres, num_desc=num_desc, num_putative=np.zeros(0, np.int64), matches=[], model_rows=[], trial=np.zeros(0, np.int64),
                        statsPutative=np.zeros(0, np.int64), statsSuccess=np.zeros(0, np.int64), statsInliers=np.zeros(0, np.int64),
                        statsRatio=np.zeros(0), transforms=[])
        row_off, tot, n_max, seg_off, roff_dev, rows_all, feat_all = (sph[k] for k in ("row_off", "tot", "n_max", "seg_off", "roff_dev", "rows_all", "feat_all"))
        pairs_all = torch.zeros((S, max(VS, 1), 2), dtype=i32, device=dev)
        n_pairs = torch.zeros(S, dtype=i32, device=dev)
        o = _match_opts(par)
        wsb = L.pcreg_dev_get_matches_segmented_workspace(VS, sph["desc_u"].shape[0], self.D, S, tot, n_max)
        if getattr(self, "_seg_ws", None) is None or self._seg_ws.numel() < wsb:
            self._seg_ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        # the model is fixed for the life of this object (one model, many surfaces): its powered rows and row scalars -- what the
        # segmented matcher makes of the model set alone -- are prepared once per set of options, not once per sweep
        desc_u, rows_u = sph["desc_u"], sph["rows_u"]
        VMu = desc_u.shape[0]
        key = (skey, int(o.change_metric), float(o.metric_factor) if o.change_metric else 0.0)
        if getattr(self, "_seg_prep_key", None) != key:
            nb = L.pcreg_dev_segmented_model_bytes(VMu, self.D)
            if getattr(self, "_seg_prep", None) is None or self._seg_prep.numel() < nb:
                self._seg_prep = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
            check(L.pcreg_dev_segmented_model_prepare(_p(desc_u), VMu, self.D, C.byref(o), _p(self._seg_prep), C.c_size_t(self._seg_prep.numel()), sp))
            self._seg_prep_key = key
        check(L.pcreg_dev_get_matches_segmented_prepared(_p(self.descS), VS, _p(desc_u), VMu, self.D, _p(self._seg_prep), key[1], C.c_double(o.metric_factor),
                                                         _p(rows_u), _p(seg_off), S, tot, n_max, C.byref(o), _p(pairs_all), None, _p(n_pairs), _p(self._seg_ws),
                                                         C.c_size_t(self._seg_ws.numel()), sp))                                  # :131-149
        return self._finish_sweep(centres, num_desc, row_off, rows_all, feat_all, None, pairs_all, n_pairs, options, putative_thresh, seed, roff_dev,
                                  model_rows=sph["model_rows"])

    def _pinned(self, key: str, n: int, dtype) -> "torch.Tensor":
        """A page-locked host buffer of at least n elements, kept across sweeps (the first n elements are returned)."""
        cache = self.__dict__.setdefault("_pin", {})
        t = cache.get(key)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
            cache[key] = t
        return t[:n]

    def _finish_sweep(self, centres, num_desc, row_off, rows_all, feat_all, n_sel, pairs_all, n_pairs, options, putative_thresh, seed, roff_dev=None,
                      model_rows=None) -> dict:
        """:166-224 on the current stream + the sweep's second (last) host synchronisation."""
        from ._lib import DevRansacResult
        L = lib()
        dev = self.dev
        S = len(centres)
        i32, f64 = torch.int32, torch.float64
        sp = _stream()
        trial_idx = torch.empty(S, dtype=i32, device=dev); offsets = torch.empty(S + 1, dtype=i32, device=dev)
        n_trials = torch.zeros(1, dtype=i32, device=dev)
        check(L.pcreg_dev_sweep_plan(_p(n_pairs), S, int(putative_thresh), _p(trial_idx), _p(offsets), _p(n_trials), sp))
        ld = S * max(self.VS, 1)                                    # capacity of the packed correspondences (Unique: <= VS pairs per sphere)
        p1 = torch.zeros((3, ld), dtype=f64, device=dev); p2 = torch.zeros((3, ld), dtype=f64, device=dev)
        if roff_dev is None:
            roff_dev = torch.from_numpy(row_off[:S].copy()).to(dev)
        check(L.pcreg_dev_sweep_gather(_p(pairs_all), self.VS, _p(n_pairs), _p(trial_idx), _p(offsets), _p(n_trials), S, _p(self.featS),
                                       _p(feat_all), _p(roff_dev), _p(p1), _p(p2), ld, sp))
        o = RansacOpts(int(options["minPtNum"]), int(options["iterNum"]), float(options["thDist"]), float(options["thInlrRatio"]),
                       int(bool(options["REFINE"])), 0, int(seed))
        rs = C.sizeof(DevRansacResult)
        results = torch.zeros((S, rs), dtype=torch.uint8, device=dev)
        inliers = torch.empty(ld, dtype=i32, device=dev)
        wsb = L.pcreg_dev_ransac_batched_workspace(self.VS, o.iterNum, S)
        if getattr(self, "_rs_ws", None) is None or self._rs_ws.numel() < wsb:
            self._rs_ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        check(L.pcreg_dev_ransac_batched(_p(p1), _p(p2), ld, _p(offsets), S, self.VS, C.byref(o), _p(results), _p(inliers), _p(self._rs_ws),
                                         C.c_size_t(self._rs_ws.numel()), sp))
        # ---- sync 2: everything comes back together -- the small results in ONE pinned transfer (seven separate .cpu() calls were
        # seven synchronisations, ~0.4 ms of the sweep), then the pair and row lists, which need the pair counts for their size
        check_sel = n_sel is not None
        if not check_sel:                                            # run(): the row lists were checked and fetched when the spheres were made
            n_sel = torch.zeros(S, dtype=i32, device=dev)            # (keeps the layout of the packed transfer below)
        small = torch.cat([n_pairs, n_trials, trial_idx, n_sel, results.view(torch.int32).view(-1)])
        hs = self._pinned("small", small.numel(), torch.int32)
        hs.copy_(small, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        h = hs.numpy()
        npr = h[:S].astype(np.int64)
        nt = int(h[S])
        trial = h[S + 1:S + 1 + nt].astype(np.int64)
        nsel = h[2 * S + 1:3 * S + 1].copy()
        raw = h[3 * S + 1:].view(np.uint8).reshape(S, rs)[:max(nt, 1)].copy()
        m_pairs = max(int(npr.max()) if S else 0, 1)                # only the columns that hold pairs cross the bus
        hp = self._pinned("pairs", S * m_pairs * 2, torch.int32).view(S, m_pairs, 2)
        hp.copy_(pairs_all[:, :m_pairs], non_blocking=True)
        if model_rows is None:
            hr = self._pinned("rows", max(rows_all.numel(), 1), torch.int32)[:rows_all.numel()]
            hr.copy_(rows_all, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        pairs_host = hp.numpy().astype(np.uint32)                   # one conversion (a copy: the pinned buffer is reused); the per-sphere lists are views of it
        if model_rows is None:
            rows_host = hr.numpy().astype(np.int64)
            model_rows = [rows_host[row_off[i]:row_off[i + 1]] for i in range(S)]
        assert not check_sel or np.array_equal(nsel, num_desc), "sphere_select disagrees with sphere_counts"
        sp_, ss, si, sr, tf = [], [], [], [], []
        for t in range(nt):
            r = DevRansacResult.from_buffer_copy(raw[t].tobytes())
            P = int(npr[trial[t]])
            sp_.append(P); ss.append(r.num_success); si.append(r.max_inliers)
            sr.append(100.0 * r.max_inliers / P if not r.failed else 0.0)
            tf.append(None if r.failed else np.array(r.T[:]).reshape(4, 4, order="F"))
        return dict(centres=centres, num_desc=num_desc, num_putative=npr,
                    matches=[pairs_host[i, :npr[i]] for i in range(S)],
                    model_rows=model_rows, trial=trial,
                    statsPutative=np.array(sp_, dtype=np.int64), statsSuccess=np.array(ss, dtype=np.int64),
                    statsInliers=np.array(si, dtype=np.int64), statsRatio=np.array(sr, dtype=np.float64), transforms=tf)

    def run_streams(self, par: dict, options: dict, R_desc: float, d_spheres: float = 5.0, min_pts: int = 1400,
                    putative_thresh: int = 170, seed: int = 0, n_streams: int = 8) -> dict:
        """Round 2's batched form, kept as the cross-check of run() and for SSD: one getMatches CHAIN per sphere,
        round-robin on `n_streams` HIP streams, then the same plan / gather / batched ransac.

        The reference runs the per-sphere getMatches under parfor (:131-149) and the per-trial ransac under a second
        parfor (:201-216).  Here: (sync 1) the descriptor counts of all candidate centres, which fix the valid spheres
        and every buffer size (:52-64); then, with nothing read back in between, for every valid sphere -- round-robin on
        `n_streams` HIP streams so that the latency-sized kernels of different spheres overlap -- getDescriptorMask as an
        index list, the row gathers and getMatches (:109-149); then on the main stream the putative threshold as a device
        list of trial spheres (:175, pcreg_dev_sweep_plan), the matched keypoints of all trials packed back to back
        (:205-206, pcreg_dev_sweep_gather) and ONE batched ransac launch over the trial capacity
        (pcreg_dev_ransac_batched, seed + trial ordinal); (sync 2) counts, pairs, rows and the result structs come back
        together.  Same outputs as run_serial()."""
        from ._lib import DevRansacResult
        L = lib()
        dev = self.dev
        centres = self.sphere_centres(d_spheres)
        valid, counts = self.valid_spheres(centres, R_desc, min_pts)                     # ---- sync 1
        centres = centres[valid]
        num_desc = counts[valid].astype(np.int64)
        S = len(centres)
        empty = dict(centres=centres, num_desc=num_desc, num_putative=np.zeros(0, np.int64), matches=[], model_rows=[], trial=np.zeros(0, np.int64),
                     statsPutative=np.zeros(0, np.int64), statsSuccess=np.zeros(0, np.int64), statsInliers=np.zeros(0, np.int64),
                     statsRatio=np.zeros(0), transforms=[])
        if S == 0:
            return empty
        i32, f64 = torch.int32, torch.float64
        row_off = np.zeros(S + 1, dtype=np.int64); row_off[1:] = np.cumsum(num_desc)
        tot = int(row_off[-1]); n_max = int(num_desc.max())
        rows_all = torch.empty(tot, dtype=i32, device=dev)
        feat_all = torch.empty((tot, 3), dtype=f64, device=dev)
        desc_all = torch.empty((tot, self.D), dtype=f64, device=dev)
        n_sel = torch.zeros(S, dtype=i32, device=dev)
        pairs_all = torch.zeros((S, max(self.VS, 1), 2), dtype=i32, device=dev)
        n_pairs = torch.zeros(S, dtype=i32, device=dev)
        ns = max(1, min(n_streams, S))
        if not hasattr(self, "_lanes") or len(self._lanes) < ns:
            self._lanes = [(torch.cuda.Stream(device=dev), DescriptorPipeline(dev),
                            torch.empty(max(L.pcreg_dev_sphere_select_workspace(self.VM), 256), dtype=torch.uint8, device=dev)) for _ in range(ns)]
        cur = torch.cuda.current_stream(dev)
        for st, _, _ in self._lanes[:ns]:
            st.wait_stream(cur)
        for i in range(S):
            st, pipe, ws_sel = self._lanes[i % ns]
            lo, n = int(row_off[i]), int(num_desc[i])
            cc = (C.c_double * 3)(*[float(v) for v in centres[i]])
            with torch.cuda.stream(st):
                sp = C.c_void_p(st.cuda_stream)
                rows = rows_all[lo:lo + n]
                check(L.pcreg_dev_sphere_select(_p(self.featM), self.VM, cc, C.c_double(R_desc), _p(rows), _p(n_sel[i:i + 1]), _p(ws_sel),
                                                C.c_size_t(ws_sel.numel()), sp))                                         # getDescriptorMask, :118
                check(L.pcreg_dev_gather_rows_f64(_p(self.descM), self.D, _p(rows), _p(n_sel[i:i + 1]), n, _p(desc_all[lo:lo + n]), sp))   # :121-125
                check(L.pcreg_dev_gather_rows_f64(_p(self.featM), 3, _p(rows), _p(n_sel[i:i + 1]), n, _p(feat_all[lo:lo + n]), sp))
                pipe.match(self.descS, self.VS, desc_all[lo:lo + n], n, par, pairs_out=pairs_all[i], n_pairs_out=n_pairs[i:i + 1],
                           ws_cap=(self.VS, n_max))                                                                       # getMatches, :141
        for st, _, _ in self._lanes[:ns]:
            cur.wait_stream(st)
        return self._finish_sweep(centres, num_desc, row_off, rows_all, feat_all, n_sel, pairs_all, n_pairs, options, putative_thresh, seed)


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


class FinalStage:
    """completeExperimentFast.m:280-394 on the device tier: for every cluster of promising spheres, move the surface by the
    cluster's RANSAC transform, describe it again WITHOUT local alignment, match against the model's no-LRF descriptors
    inside the cluster's sphere, keep the matches closer than maxDist; the cluster with the largest share of close matches
    gives T_refine and the final surface.

    TWO host synchronisations for the whole stage, whatever the number of clusters: (A) the sizes that fix the match
    grids -- every cluster's surviving keypoint count and descriptor count in its sphere (like the sweep's first read);
    (B) every cluster's match count, close-match count and refined transform, together.  The final surface
    quickTF(pts_tform, invertTF(T_refine)) is enqueued after (B) and stays on the device.

    ptsSurface [3, N] float64 on the device; featModel_noLRF [VM, 3] / descModel_noLRF [VM, 980] as for SphereSweep."""

    def __init__(self, ptsSurface: torch.Tensor, featModel_noLRF, descModel_noLRF, device: torch.device | None = None):
        self.dev = device or ptsSurface.device
        t = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))).to(self.dev).contiguous()
        self.pts = ptsSurface.to(self.dev)
        self.featM, self.descM = t(featModel_noLRF), t(descModel_noLRF)
        self.VM, self.D = self.descM.shape
        self.pipe = DescriptorPipeline(self.dev)

    def run(self, clusters, sample_pts, descOpt: dict, par: dict, R_desc: float, maxDist: float = 1.5, return_matches: bool = True) -> dict:
        """clusters = [(locCur [3], transCur [4,4])] (:283-288); sample_pts[i] = the keypoints drawn for cluster i (:297,
        pcRandomUniformSamples is the caller's: its random stream is not the library's business), [S_i, 3] numpy or [3, S_i]
        device tensors."""
        L = lib()
        dev, K = self.dev, len(clusters)
        i32, f64 = torch.int32, torch.float64
        if K == 0:
            raise ValueError("final stage: no clusters")
        opt = dict(descOpt, ALIGN_POINTS=False, VERBOSE=0)                                         # :300
        tf, feats, descs = [], [], []
        counters = torch.zeros((K, 2), dtype=i32, device=dev)
        for i, ((loc, T), kp) in enumerate(zip(clusters, sample_pts)):
            kp = kp if isinstance(kp, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(kp, dtype=np.float64).T)).to(dev)
            pt = quickTF_dev(self.pts, invertTF(np.asarray(T, dtype=np.float64)))                  # :291
            f, d, c = self.pipe.describe_async(pt, kp, opt)                                        # :309-310
            counters[i].copy_(c)
            tf.append(pt); feats.append(f); descs.append(d)
        locs = torch.from_numpy(np.ascontiguousarray([np.asarray(c[0], dtype=np.float64) for c in clusters])).to(dev)
        n_in = torch.zeros(K, dtype=i32, device=dev)
        check(L.pcreg_dev_sphere_counts(_p(self.featM), self.VM, _p(locs), K, C.c_double(R_desc), _p(n_in), _stream()))
        cnt = torch.cat([counters.view(-1), n_in]).cpu().numpy()                                   # ---- sync A: sizes only
        V = cnt[0:2 * K:2].astype(np.int64); over = cnt[1:2 * K:2]; nM = cnt[2 * K:].astype(np.int64)
        if over.any():
            raise ValueError(f"a support holds {int(over.max())} points: more than an LDS-resident support may have (lower max_pts)")
        seg_off_h = np.zeros(K + 1, dtype=np.int32); seg_off_h[1:] = np.cumsum(nM)
        tot = int(seg_off_h[-1])
        seg_off = torch.from_numpy(seg_off_h).to(dev)
        rows_all = torch.empty(max(tot, 1), dtype=i32, device=dev)
        featCur_all = torch.empty((max(tot, 1), 3), dtype=f64, device=dev)
        check(L.pcreg_dev_sphere_select_batched(_p(self.featM), self.VM, _p(locs), K, C.c_double(R_desc), _p(seg_off), _p(rows_all), _p(featCur_all),
                                                None, _stream()))                                  # :319-322, every cluster's sphere
        n_pairs = torch.zeros(K, dtype=i32, device=dev)
        info = torch.zeros((K, 2), dtype=i32, device=dev); info[:, 1] = 1
        T16 = torch.zeros((K, 16), dtype=f64, device=dev)
        pairs = []
        n_tot = torch.tensor([tot], dtype=i32, device=dev)
        for i in range(K):
            v, n, lo = int(V[i]), int(nM[i]), int(seg_off_h[i])
            pr = torch.zeros((max(v, 1), 2), dtype=i32, device=dev)
            pairs.append(pr)
            if v == 0 or n == 0:
                continue
            descCur = torch.empty((n, self.D), dtype=f64, device=dev)
            check(L.pcreg_dev_gather_rows_f64(_p(self.descM), self.D, _p(rows_all[lo:lo + n]), _p(n_tot), n, _p(descCur), _stream()))
            self.pipe.match(descs[i][:v], v, descCur, n, par, pairs_out=pr, n_pairs_out=n_pairs[i:i + 1])          # :344
            p1 = torch.empty((3, v), dtype=f64, device=dev); p2 = torch.empty((3, v), dtype=f64, device=dev)
            check(L.pcreg_dev_gather_matched_rows(_p(pr), _p(n_pairs[i:i + 1]), v, _p(feats[i]), _p(featCur_all[lo:lo + n]), _p(p1), _p(p2), _stream()))
            check(L.pcreg_dev_refine_by_distance(_p(p1), _p(p2), _p(n_pairs[i:i + 1]), v, v, C.c_double(maxDist), _p(T16[i]), _p(info[i]),
                                                 _stream()))                                       # :357-391 for every cluster
        back = torch.cat([n_pairs.to(f64), info.view(-1).to(f64), T16.view(-1)]).cpu().numpy()     # ---- sync B: the results
        npr = back[:K].astype(np.int64); inf = back[K:3 * K].reshape(K, 2).astype(np.int64); Ts = back[3 * K:].reshape(K, 16)
        with np.errstate(invalid="ignore", divide="ignore"):
            prec = np.where(npr > 0, inf[:, 0] / np.maximum(npr, 1) * 100.0, np.nan)              # :373
        best = 0 if np.all(np.isnan(prec)) else int(np.nanargmax(prec))                           # :381, MATLAB's max
        T_refine = None if inf[best, 1] else Ts[best].reshape(4, 4, order="F")
        pts_final = quickTF_dev(tf[best], invertTF(T_refine)) if T_refine is not None else tf[best]   # :394, stays on the device
        out = dict(num_keypoints=V, num_desc=nM, num_matches=npr, num_close=inf[:, 0], precisions=prec, best=best, T_refine=T_refine,
                   pts_final=pts_final, pts_tform=tf, host_syncs=2)
        if return_matches:                                                                        # a third, optional read (tests, plots)
            out["matches"] = [pairs[i][:npr[i]].cpu().numpy().astype(np.uint32) for i in range(K)]
            out["model_rows"] = [rows_all[seg_off_h[i]:seg_off_h[i + 1]].cpu().numpy().astype(np.int64) for i in range(K)]
        return out


__all__ = ["SphereSweep", "FinalStage", "pcUniformSamples", "quickTF_dev", "refine_by_distance_dev", "invertTF"]
