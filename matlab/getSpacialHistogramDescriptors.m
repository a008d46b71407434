function [feat, desc] = getSpacialHistogramDescriptors(pts, sample_pts, options)
%GETSPACIALHISTOGRAMDESCRIPTORS  Drop-in for the reference's function of the same name
%   (same options struct: min_pts, max_pts, R, thVar, k, ALIGN_POINTS[, VERBOSE]).
%   feat: V x 3 locations, desc: V x 980 spherical count histograms of the surviving keypoints -- DOUBLE whatever the
%   classes of the inputs, as in the reference (it preallocates both with nan(...): getSpacialHistogramDescriptors.m:61-62).
%   single inputs (clouds from pcread: completeExperimentFast.m:309) are passed in their own class: getLocalPoints'
%   element-wise single arithmetic (box test, pts_cube - c, vecnorm, dists < R) is reproduced, the rest runs in double
%   (INTEGRATION.md, "Differences a user can observe").
    if isfield(options, 'VERBOSE'), VERBOSE = options.VERBOSE; else, VERBOSE = 1; end
    if VERBOSE, tic; end
    if ~isa(pts, 'single'), pts = double(pts); end
    if ~isa(sample_pts, 'single'), sample_pts = double(sample_pts); end
    [feat, desc] = pcreg_mex('getSpacialHistogramDescriptors', pts, sample_pts, options);
    if VERBOSE, fprintf('Calculated descriptors in %0.1f seconds...\n', toc); end
end
