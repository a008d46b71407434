function [TCells, inlierCells, numSuccess, maxInliers, ratio] = ransacBatched(pts1Cells, pts2Cells, ransacCoef)
%RANSACBATCHED  ransac(pts1Cells{i}, pts2Cells{i}, ransacCoef, @estimateTransform, @calcDists) for every i in ONE library call.
%   The second parfor of completeExperimentFast.m (:201-216) runs one ransac per sphere that passed the putative threshold;
%   through the drop-in ransac.m that is one MEX call, one upload and one download per sphere.  Here all of them are one launch:
%       for i = 1:numTrials, pts1Cells{i} = loc1M_i; pts2Cells{i} = loc1S_i; end        % what :203-209 assemble per sphere
%       [TCells, inlierCells, numSuccess, maxInliers, ratio] = ransacBatched(pts1Cells, pts2Cells, ransacCoef);
%   TCells{i} is [] where ransac found no transformation (as ransac returns), inlierCells{i} the inlier indices of registration i,
%   numSuccess / maxInliers / ratio column vectors (what :210-216 store in statsSuccess / statsInliers / statsRatio).
%   Sampling as in the drop-in ransac.m: by default the triples are drawn HERE with randperm, registration after registration,
%   exactly as a for-loop over ransac would consume MATLAB's stream; ransacCoef.SAMPLER = 'device' uses the library's counter-based
%   sampler instead (seed ransacCoef.seed + i - 1), which is what makes the call cheap for many registrations.
    B = numel(pts1Cells);
    if isfield(ransacCoef, 'VERBOSE'), VERBOSE = ransacCoef.VERBOSE; else, VERBOSE = 1; end
    lens = cellfun(@(p) size(p, 1), pts1Cells(:));
    pts1 = double(vertcat(pts1Cells{:})); pts2 = double(vertcat(pts2Cells{:}));
    if isempty(pts1), pts1 = zeros(0, 3); pts2 = zeros(0, 3); end
    offsets = int32([0; cumsum(lens)]);
    iterNum = ransacCoef.iterNum; minPtNum = ransacCoef.minPtNum;
    if isfield(ransacCoef, 'SAMPLER') && strcmp(ransacCoef.SAMPLER, 'device')
        sampleIdx = [];
        if isfield(ransacCoef, 'seed'), seed = ransacCoef.seed; else, seed = 0; end
    else
        sampleIdx = zeros(minPtNum, iterNum * B, 'int32');       % one COLUMN per hypothesis, registration after registration
        for b = 1:B
            for p = 1:iterNum
                r = randperm(lens(b));                           % same stream consumption as ransac.m:42 in a loop over the spheres
                sampleIdx(:, (b - 1) * iterNum + p) = r(1:minPtNum);
            end
        end
        seed = 0;
    end
    [T, inlierIdx, nInliers, numSuccess, maxInliers, failed] = pcreg_mex('ransacBatched', pts1, pts2, offsets, ransacCoef, sampleIdx, seed);
    TCells = cell(size(pts1Cells)); inlierCells = cell(size(pts1Cells));
    k = 0;
    for b = 1:B
        if failed(b)
            TCells{b} = []; inlierCells{b} = [];
        else
            TCells{b} = T(:, :, b);
            inlierCells{b} = inlierIdx(k + 1:k + nInliers(b));
            k = k + nInliers(b);
        end
    end
    ratio = 100 * maxInliers ./ max(lens, 1);
    if VERBOSE, fprintf('RANSAC: %d of %d registrations found a transformation\n', sum(~failed), B); end
end
