"""The drivers' on-disk formats through libpcreg_hip (host code, no GPU needed):
pcread / pcwrite for .pcd clouds (completeExperimentFast.m:12-13,30,403) and `load` of the
.mat descriptor caches (:21-24,312-313)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib


def pcread(path: str):
    """-> (Location N x 3 float32, Color N x 3 uint8 or None), like MATLAB's pointCloud."""
    L = lib()
    n, has = C.c_int(0), C.c_int(0)
    check(L.pcreg_pcd_info(path.encode(), C.byref(n), C.byref(has)))
    N = n.value
    xyz = np.zeros((max(N, 1), 3), dtype=np.float32, order="F")
    rgb = np.zeros(max(N, 1), dtype=np.uint32) if has.value else None
    check(L.pcreg_pcd_read(path.encode(), xyz.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(max(N, 1)),
                           rgb.ctypes.data_as(C.POINTER(C.c_uint32)) if rgb is not None else None, C.c_int(N)))
    color = None
    if rgb is not None:
        color = np.stack([(rgb[:N] >> 16) & 255, (rgb[:N] >> 8) & 255, rgb[:N] & 255], axis=1).astype(np.uint8)
    return np.ascontiguousarray(xyz[:N]), color


def pcwrite(path: str, location, color=None, encoding: str = "ascii") -> None:
    """pcwrite(ptCloud, filename, 'Encoding', 'ascii' | 'binary')."""
    if encoding not in ("ascii", "binary"):
        raise ValueError("encoding must be 'ascii' or 'binary'")
    xyz = np.asfortranarray(np.asarray(location, dtype=np.float32).reshape(-1, 3))
    N = xyz.shape[0]
    rgb = None
    if color is not None:
        c = np.asarray(color, dtype=np.uint32).reshape(-1, 3)
        rgb = np.ascontiguousarray((c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2], dtype=np.uint32)
    check(lib().pcreg_pcd_write(path.encode(), xyz.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(N), C.c_int(max(N, 1)),
                                rgb.ctypes.data_as(C.POINTER(C.c_uint32)) if rgb is not None else None,
                                C.c_int(encoding == "binary")))


def load_mat(path: str, name: str | None = None) -> np.ndarray:
    """One real numeric variable of a Level-5 MAT-file (v6 / v7) as a float64 matrix."""
    L = lib()
    r, c = C.c_int(0), C.c_int(0)
    nm = name.encode() if name else None
    check(L.pcreg_mat_read_double(path.encode(), nm, None, C.byref(r), C.byref(c)))
    out = np.zeros((r.value, c.value), dtype=np.float64, order="F")
    if out.size:
        check(L.pcreg_mat_read_double(path.encode(), nm, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(r), C.byref(c)))
    return out


__all__ = ["pcread", "pcwrite", "load_mat"]
