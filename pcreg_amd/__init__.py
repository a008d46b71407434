"""pcreg_amd -- MI355X (gfx950) implementation of PCReg's correspondence-search +
RANSAC rigid-alignment hot path, behind the reference's own function names.

The arithmetic lives in libpcreg_hip.so (hand-written HIP kernels, C ABI in
include/pcreg.h).  This package is the thin host-side mirror of the MATLAB interface;
it has no CPU fallback and raises if the library or a gfx950 device is missing.
"""
from .api import (EMPTY, GETINLIERS_COEFF, Model, AlignPoints_KNN, AlignPoints_KNN_batched, calcDists,  # noqa: F401
                  estimateTransform, getInliersRANSAC, getLocalPoints, getMatches, getMatchesOnSet, getMatchesSegmentedOnSet, sphereCounts, sphereSweep, SphereModel, sphereSweepOnModel, DescSet, getMatchesSegmented, getSpacialHistogramDescriptors, speedyDescriptors, invertTF, knn2_points, matchFeatures,
                  match_points, quickTF, ransac, ransac_batched)

__all__ = ["EMPTY", "GETINLIERS_COEFF", "Model", "AlignPoints_KNN", "AlignPoints_KNN_batched", "calcDists",
           "estimateTransform", "getInliersRANSAC", "getLocalPoints", "getMatches", "getMatchesOnSet", "getMatchesSegmentedOnSet", "sphereCounts", "sphereSweep", "SphereModel", "sphereSweepOnModel", "DescSet", "getMatchesSegmented", "getSpacialHistogramDescriptors", "speedyDescriptors", "invertTF", "knn2_points", "matchFeatures",
           "match_points", "quickTF", "ransac", "ransac_batched"]
