function matchesCells = getMatchesSegmentedOnSet(hSurface, hModel, rowsCells, par)
%GETMATCHESSEGMENTEDONSET  getMatchesSegmented on descriptor sets that already sit on the GPU.
%   getMatchesSegmented uploads both descriptor matrices with every call -- for the sweep's 60 000 x 980 model set 470 MB,
%   which takes longer to cross the bus than all spheres take to match.  With ONE model and many surfaces
%   (completeExperimentFast.m runs once per surface crop), upload the model once:
%       hM = pcreg_mex('descCreate', double(descModel));                        % once per model
%       hS = pcreg_mex('descCreate', double(descSurface));                      % once per surface
%       rowsCells{i} = find(getDescriptorMask(featModel, c_i, R_desc, 0));      % :118, for every sphere
%       matchesCells = getMatchesSegmentedOnSet(hS, hM, rowsCells, par);        % instead of the parfor over getMatches (:131-149)
%       pcreg_mex('descDestroy', hS);  ...  pcreg_mex('descDestroy', hM);
%   Same pairs as getMatchesSegmented / as getMatches per sphere.  Metric must be 'SAD'.
    lens = cellfun(@numel, rowsCells(:));
    rows = int32(vertcat(rowsCells{:}));
    if isempty(rows), rows = zeros(0, 1, 'int32'); end
    segOff = int32([0; cumsum(lens)]);
    [pairs, nPairs] = pcreg_mex('getMatchesSegmentedOnSet', hSurface, hModel, rows(:), segOff, par);
    matchesCells = mat2cell(pairs, double(nPairs), 2);
    matchesCells = reshape(matchesCells, size(rowsCells));
end
