"""Loader of libpcreg_hip.so (the C ABI declared in include/pcreg.h).

There is no CPU fallback: if the shared library has not been built, or no gfx950
device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCREG_LIB") or os.path.join(_HERE, "libpcreg_hip.so")     # PCREG_LIB: another build of the same library (A/B runs)

PCREG_OK, PCREG_E_ARG, PCREG_E_HIP, PCREG_E_NODEVICE, PCREG_E_WORKSPACE = 0, 1, 2, 3, 4
METRIC_SAD, METRIC_SSD = 0, 1
LAYOUT_FEATURE_MAJOR, LAYOUT_ROW_MAJOR = 0, 1


class PcregError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpcreg_hip error {code}: {msg}")
        self.code = code


class RansacOpts(C.Structure):
    """pcreg_ransac_opts (include/pcreg.h) == ransacCoef of ransac.m:7-12."""
    _fields_ = [("minPtNum", C.c_int32), ("iterNum", C.c_int32), ("thDist", C.c_double),
                ("thInlrRatio", C.c_double), ("REFINE", C.c_int32), ("VERBOSE", C.c_int32),
                ("seed", C.c_uint64)]


class MatchOpts(C.Structure):
    """pcreg_match_opts (include/pcreg.h) == `par` of getMatches.m."""
    _fields_ = [("metric", C.c_int32), ("matchThreshold", C.c_double), ("maxRatio", C.c_double),
                ("unique", C.c_int32), ("prenormalized", C.c_int32), ("unnormalize", C.c_int32),
                ("norm_factor", C.c_double), ("change_metric", C.c_int32), ("metric_factor", C.c_double)]


class DescOpts(C.Structure):
    """pcreg_desc_opts (include/pcreg.h) == `options` of getSpacialHistogramDescriptors.m."""
    _fields_ = [("min_pts", C.c_int32), ("max_pts", C.c_int32), ("R", C.c_double), ("thVar", C.c_double * 2),
                ("k", C.c_double), ("ALIGN_POINTS", C.c_int32)]


class DevRansacResult(C.Structure):
    """pcreg_dev_ransac_result (include/pcreg.h)."""
    _fields_ = [("T", C.c_double * 16), ("n_inliers", C.c_int32), ("num_success", C.c_int32),
                ("max_inliers", C.c_int32), ("failed", C.c_int32), ("n", C.c_int32), ("winner", C.c_int32)]


# every symbol include/pcreg.h declares (tests/test_abi.py checks the two lists agree)
SYMBOLS = [
    "pcreg_last_error", "pcreg_version", "pcreg_device_count", "pcreg_set_device", "pcreg_device_name", "pcreg_debug_set", "pcreg_debug_match_stats",
    "pcreg_estimate_transform", "pcreg_calc_dists", "pcreg_ransac", "pcreg_ransac_batched",
    "pcreg_knn2_points_f32", "pcreg_match_points_f32", "pcreg_match_features", "pcreg_get_matches", "pcreg_desc_set_create", "pcreg_desc_set_destroy", "pcreg_desc_set_size", "pcreg_get_matches_on_sets", "pcreg_get_matches_segmented_on_sets", "pcreg_sphere_counts", "pcreg_sphere_sweep", "pcreg_sphere_model_create", "pcreg_sphere_model_destroy", "pcreg_sphere_sweep_on_model", "pcreg_get_matches_segmented", "pcreg_get_local_points",
    "pcreg_model_create", "pcreg_model_destroy", "pcreg_model_match_points_f32",
    "pcreg_dev_model_create", "pcreg_dev_model_destroy", "pcreg_dev_model_search_workspace", "pcreg_dev_model_search_f32",
    "pcreg_dev_model_match_f32", "pcreg_dev_model_match_table_f32", "pcreg_dev_match_from_table_f32",
    "pcreg_align_points_knn", "pcreg_align_points_knn_f32", "pcreg_align_points_knn_batched", "pcreg_spatial_histogram_descriptors",
    "pcreg_spatial_histogram_descriptors_f32", "pcreg_spatial_histogram_descriptors_mixed",
    "pcreg_dev_knn2_points_f32_workspace", "pcreg_dev_knn2_points_f32", "pcreg_dev_merge_top2_f32", "pcreg_dev_merge_top2_strided_f32",
    "pcreg_dev_ransac_workspace", "pcreg_dev_ransac",
    "pcreg_dev_ransac_partial", "pcreg_dev_ransac_finish", "pcreg_dev_ransac_finish_parts",
    "pcreg_dev_search_kernel_timing", "pcreg_dev_search_kernel_ms",
    "pcreg_dev_spatial_histogram_descriptors_workspace", "pcreg_dev_spatial_histogram_descriptors",
    "pcreg_dev_spatial_histogram_descriptors_rows_u16", "pcreg_dev_get_matches_rows_u16",
    "pcreg_dev_get_matches_workspace", "pcreg_dev_get_matches", "pcreg_dev_gather_matched_rows",
    "pcreg_dev_sphere_counts", "pcreg_dev_sphere_select_workspace", "pcreg_dev_sphere_select", "pcreg_dev_sphere_select_batched",
    "pcreg_dev_get_matches_segmented_workspace", "pcreg_dev_get_matches_segmented", "pcreg_dev_segmented_model_bytes", "pcreg_dev_segmented_model_prepare", "pcreg_dev_get_matches_segmented_prepared",
    "pcreg_dev_gather_rows_f64", "pcreg_dev_sweep_plan", "pcreg_dev_sweep_gather", "pcreg_dev_ransac_batched_workspace",
    "pcreg_dev_ransac_batched", "pcreg_dev_align_points_knn_batched", "pcreg_dev_quick_tf", "pcreg_dev_refine_by_distance",
    "pcreg_comm_get_unique_id", "pcreg_comm_init", "pcreg_comm_init_host_staged", "pcreg_comm_rank", "pcreg_comm_destroy",
    "pcreg_match_points_sharded_f32", "pcreg_ransac_sharded",
    "pcreg_pcd_info", "pcreg_pcd_read", "pcreg_pcd_write", "pcreg_mat_read_double",
]

_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises if it was never built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C pcreg_amd/csrc` "
                "(or __graft_entry__.build()).  pcreg_amd has no CPU fallback.")
        # Load torch (the process's device-memory / RCCL plumbing) BEFORE the library: the
        # torch wheel bundles its own HIP runtime, and on this image the runtime that is
        # loaded second only sees the GPU when torch's was loaded first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.pcreg_last_error.restype = C.c_char_p
        L.pcreg_version.restype = C.c_char_p
        for name in ("pcreg_dev_model_search_workspace", "pcreg_dev_knn2_points_f32_workspace",
                     "pcreg_dev_ransac_workspace", "pcreg_dev_spatial_histogram_descriptors_workspace",
                     "pcreg_dev_get_matches_workspace", "pcreg_dev_sphere_select_workspace", "pcreg_dev_ransac_batched_workspace",
                     "pcreg_dev_get_matches_segmented_workspace", "pcreg_dev_segmented_model_bytes"):
            getattr(L, name).restype = C.c_size_t
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != PCREG_OK:
        raise PcregError(rc, lib().pcreg_last_error().decode(errors="replace"))
