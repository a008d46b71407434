function [feat, desc] = getSpacialHistogramDescriptors(pts, sample_pts, options)
%GETSPACIALHISTOGRAMDESCRIPTORS  Drop-in for the reference's function of the same name
%   (same options struct: min_pts, max_pts, R, thVar, k, ALIGN_POINTS[, VERBOSE]).
%   feat: V x 3 locations, desc: V x 980 spherical count histograms of the surviving keypoints.
%   Two single inputs (clouds from pcread: completeExperimentFast.m:309) give single outputs, as in MATLAB;
%   the support test and the binning are evaluated in double (INTEGRATION.md, "Differences a user can observe").
    if isfield(options, 'VERBOSE'), VERBOSE = options.VERBOSE; else, VERBOSE = 1; end
    if VERBOSE, tic; end
    if ~(isa(pts, 'single') && isa(sample_pts, 'single')), pts = double(pts); sample_pts = double(sample_pts); end
    [feat, desc] = pcreg_mex('getSpacialHistogramDescriptors', pts, sample_pts, options);
    if VERBOSE, fprintf('Calculated descriptors in %0.1f seconds...\n', toc); end
end
