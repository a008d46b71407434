function [feat, desc] = getSpacialHistogramDescriptors(pts, sample_pts, options)
%GETSPACIALHISTOGRAMDESCRIPTORS  Drop-in for the reference's function of the same name
%   (same options struct: min_pts, max_pts, R, thVar, k, ALIGN_POINTS[, VERBOSE]).
%   feat: V x 3 locations, desc: V x 980 spherical count histograms of the surviving keypoints.
    if isfield(options, 'VERBOSE'), VERBOSE = options.VERBOSE; else, VERBOSE = 1; end
    if VERBOSE, tic; end
    [feat, desc] = pcreg_mex('getSpacialHistogramDescriptors', double(pts), double(sample_pts), options);
    if VERBOSE, fprintf('Calculated descriptors in %0.1f seconds...\n', toc); end
end
