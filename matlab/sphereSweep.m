function out = sphereSweep(hSurface, hModel, featSurface, featModel, centres, R_desc, min_pts, max_pts, par, putativeThresh, ransacCoef, seed)
%SPHERESWEEP  The sphere loops of completeExperimentFast.m:52-224 as two library calls on resident descriptor sets.
%   Replaces, for a model / surface pair whose descriptor matrices were uploaded with pcreg_mex('descCreate', ...):
%       :52-64    num_desc(i) = nnz(getDescriptorMask(featModel, centres(i,:), R_desc)) and the min_pts / max_pts filter
%       :109-149  per kept sphere: mask, descCur = descModel(mask,:), matches = getMatches(descSurface, descCur, par)   (the parfor)
%       :166-184  the spheres with more than putativeThresh matches
%       :201-216  per such sphere: ransac(featSurface(matches(:,1),:), featCur(matches(:,2),:), ransacCoef, ...)          (the parfor)
%   out.valid (logical, one per candidate centre), out.centres (the kept ones), out.numDesc, out.numPutative,
%   out.modelRows{i} (= find(mask_i)), out.matches{i} (P_i x 2 uint32, model index inside modelRows{i}), out.trial (indices into the
%   kept spheres), out.TForms{t}, out.statsPutative / statsSuccess / statsInliers / statsRatio (what :210-216 record per trial).
%   The library's sampler draws the triples (seed + t - 1 for the t-th trial), as with ransacCoef.SAMPLER = 'device' in ransac.m.
    if nargin < 12, seed = 0; end
    counts = pcreg_mex('sphereCounts', double(featModel), double(centres), R_desc);
    valid = counts >= min_pts & counts <= max_pts;
    c = double(centres(valid, :)); nd = int32(counts(valid));
    [rows, pairs, nPairs, trial, T, numSuccess, maxInliers, failed] = pcreg_mex('sphereSweep', hSurface, hModel, double(featSurface), double(featModel), ...
                                                                                 c, nd(:), R_desc, par, putativeThresh, ransacCoef, seed);
    S = size(c, 1);
    out.valid = valid; out.centres = c; out.numDesc = double(nd(:)); out.numPutative = nPairs;
    if S == 0
        out.modelRows = cell(0, 1); out.matches = cell(0, 1);
    else
        out.modelRows = mat2cell(reshape(rows, [], 1), double(nd(:)), 1);
        out.matches = mat2cell(pairs, nPairs, 2);
    end
    out.trial = trial;
    out.TForms = cell(numel(trial), 1);
    for t = 1:numel(trial), if ~failed(t), out.TForms{t} = T(:, :, t); end, end
    out.statsPutative = nPairs(trial); out.statsSuccess = numSuccess; out.statsInliers = maxInliers;
    out.statsRatio = 100 * maxInliers ./ max(nPairs(trial), 1); out.statsRatio(logical(failed)) = 0;
    assert(numel(out.matches) == S);
end
