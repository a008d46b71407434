function matchesCells = getMatchesSegmented(descSurface, descModel, rowsCells, par)
%GETMATCHESSEGMENTED  getMatches(descSurface, descModel(rowsCells{i}, :), par) for every i in ONE library call.
%   The per-sphere loop of completeExperimentFast.m:131-149 (a parfor over getMatches) becomes
%       rowsCells{i} = find(getDescriptorMask(featModel, c_i, R_desc, 0));     % :118, for every sphere
%       matchesCells = getMatchesSegmented(descSurface, descModel, rowsCells, par);
%   matchesCells{i} is what getMatches returns for sphere i (P_i x 2 uint32, model indices counting inside rowsCells{i}).
%   The library computes the powered columns and the approximate scores once for all spheres (they overlap) and certifies per
%   sphere: same pairs, about ten times faster than the loop.  Metric must be 'SAD' (every driver of the reference uses it).
    S = numel(rowsCells);
    lens = cellfun(@numel, rowsCells(:));
    rows = int32(vertcat(rowsCells{:}));
    if isempty(rows), rows = zeros(0, 1, 'int32'); end
    segOff = int32([0; cumsum(lens)]);
    [pairs, nPairs] = pcreg_mex('getMatchesSegmented', double(descSurface), double(descModel), rows(:), segOff, par);
    matchesCells = mat2cell(pairs, double(nPairs), 2);
    matchesCells = reshape(matchesCells, size(rowsCells));
end
