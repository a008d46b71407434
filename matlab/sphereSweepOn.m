function out = sphereSweepOn(sm, hSurface, featSurface, par, putativeThresh, ransacCoef, seed)
%SPHERESWEEPON  sphereSweep for one more surface against a model prepared by sphereSweepModel: the same fields, the same values.
%   hSurface = pcreg_mex('descCreate', double(descSurface)).  See sphereSweep.m for the fields of out.
    if nargin < 7, seed = 0; end
    S = size(sm.centres, 1);
    [pairs, nPairs, trial, T, numSuccess, maxInliers, failed] = pcreg_mex('sphereSweepOnModel', sm.handle, hSurface, double(featSurface), S, par, ...
                                                                          putativeThresh, ransacCoef, seed);
    out.valid = sm.valid; out.centres = sm.centres; out.numDesc = sm.numDesc; out.numPutative = nPairs; out.modelRows = sm.modelRows;
    if S == 0, out.matches = cell(0, 1); else, out.matches = mat2cell(pairs, nPairs, 2); end
    out.trial = trial;
    out.TForms = cell(numel(trial), 1);
    for t = 1:numel(trial), if ~failed(t), out.TForms{t} = T(:, :, t); end, end
    out.statsPutative = nPairs(trial); out.statsSuccess = numSuccess; out.statsInliers = maxInliers;
    out.statsRatio = 100 * maxInliers ./ max(nPairs(trial), 1); out.statsRatio(logical(failed)) = 0;
end
