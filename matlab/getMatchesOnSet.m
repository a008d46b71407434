function matches = getMatchesOnSet(hSurface, hModel, rows, par)
%GETMATCHESONSET  getMatches(descSurface, descModel(rows, :), par) on descriptor sets that already sit on the GPU.
%   The sphere loop of completeExperimentFast.m:101-150 matches ONE surface set against hundreds of row subsets of ONE
%   model set; through getMatches every pass uploads both (~28 MB of doubles per sphere).  Upload them once instead:
%       hS = pcreg_mex('descCreate', double(descSurface));     % before the loop (:101)
%       hM = pcreg_mex('descCreate', double(descModel));
%       ... in the loop (:121-149), instead of descCur = descModel(mask, :); matches = getMatches(descSurface, descCur, par):
%       matches = getMatchesOnSet(hS, hM, find(mask), par);
%       ... after the loop
%       pcreg_mex('descDestroy', hS); pcreg_mex('descDestroy', hM);
%   Same pairs as getMatches, bit for bit (the same kernels on the same values); rows = [] means the whole model set.
%   Under parfor every worker creates its own handles (a handle belongs to the process and GPU that made it).
    if isfield(par, 'VERBOSE'), VERBOSE = par.VERBOSE; else, VERBOSE = 1; end
    tic
    if isempty(rows), rows = []; else, rows = int32(rows(:)); end
    matches = pcreg_mex('getMatchesOnSet', hSurface, hModel, rows, par);
    if VERBOSE, fprintf('Calculated matches in %0.1f seconds...\n', toc); end
end
