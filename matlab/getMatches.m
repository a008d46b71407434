function matches = getMatches(descSurface, descModel, par)
%GETMATCHES  Drop-in for the reference's getMatches.m (same par struct, same output:
%   P x 2 uint32 [surfaceIdx, modelIdx], ascending in column 1).  'Approximate' is
%   answered with the exact search.
    if isfield(par, 'VERBOSE'), VERBOSE = par.VERBOSE; else, VERBOSE = 1; end
    tic
    matches = pcreg_mex('getMatches', double(descSurface), double(descModel), par);
    if VERBOSE, fprintf('Calculated matches in %0.1f seconds...\n', toc); end
end
