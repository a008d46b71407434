function [T, varargout] = ransac(pts1, pts2, ransacCoef, funcFindTransf, funcDist)
%RANSAC  Drop-in for the reference's ransac.m (same signature, same outputs).
%   With the handles every PCReg caller passes (@estimateTransform, @calcDists) the
%   whole loop runs on an MI355X through pcreg_mex / libpcreg_hip.so; any other pair
%   of handles runs the reference's generic loop below (ransac.m:21-116).
%
%   The sample indices the reference draws with randperm(ptNum) (ransac.m:42-43) are
%   drawn HERE, from MATLAB's own generator, and handed to the kernel, so a seeded
%   MATLAB session reproduces the reference run hypothesis for hypothesis.
    nout = max(nargout, 1);
    if isfield(ransacCoef, 'VERBOSE'), VERBOSE = ransacCoef.VERBOSE; else, VERBOSE = 1; end
    onGpu = strcmp(func2str(funcFindTransf), 'estimateTransform') && ...
            ~isempty(regexp(func2str(funcDist), 'calcDists$', 'once'));
    ptNum = size(pts1, 1);
    if onGpu
        iterNum = ransacCoef.iterNum; minPtNum = ransacCoef.minPtNum;
        if isfield(ransacCoef, 'SAMPLER') && strcmp(ransacCoef.SAMPLER, 'device')
            % the fast path: the library's counter-based sampler (reproducible by ransacCoef.seed, NOT MATLAB's stream).
            % Drawing iterNum permutations of ptNum below costs seconds at n = 32 k, iterNum = 1e4; the kernel takes 0.7 ms.
            sampleIdx = [];
            if isfield(ransacCoef, 'seed'), seed = ransacCoef.seed; else, seed = 0; end
        else
            sampleIdx = zeros(minPtNum, iterNum, 'int32');      % one COLUMN per hypothesis
            for p = 1:iterNum
                r = randperm(ptNum);                             % same stream consumption as ransac.m:42
                sampleIdx(:, p) = r(1:minPtNum);
            end
            seed = 0;
        end
        [T, inlierIdx, numSuccess, maxInliers, failed] = pcreg_mex('ransac', double(pts1), double(pts2), ...
                                                                   ransacCoef, sampleIdx, seed);
        if failed
            if VERBOSE, fprintf('RANSAC could not find an appropriate transformation\n'); end
        elseif VERBOSE
            fprintf('RANSAC succeeded %d times with a maximum of %d Inliers (%0.2f %%)\n', ...
                    numSuccess, maxInliers, 100*maxInliers/ptNum);
        end
    else
        [T, inlierIdx, numSuccess, maxInliers] = ransac_generic(pts1, pts2, ransacCoef, funcFindTransf, funcDist, VERBOSE);
    end
    if nout > 1, varargout{1} = inlierIdx; end
    if nout > 2, varargout{2} = numSuccess; end
    if nout > 3, varargout{3} = maxInliers; end
    if nout > 4, varargout{4} = 100*maxInliers/ptNum; end
end

function [T, inlierIdx, numSuccess, maxInliers] = ransac_generic(pts1, pts2, c, fit, dst, VERBOSE)
% the reference algorithm for arbitrary handles (ransac.m:36-102)
    ptNum = size(pts1, 1); thInlr = round(c.thInlrRatio*ptNum);
    inlrNum = zeros(1, c.iterNum); inlrNum_refined = zeros(1, c.iterNum); TForms = cell(1, c.iterNum);
    for p = 1:c.iterNum
        r = randperm(ptNum); s = r(1:c.minPtNum);
        f1 = fit(pts1(s, :), pts2(s, :));
        inl = find(dst(f1, pts1, pts2) < c.thDist); inlrNum(p) = length(inl);
        if length(inl) >= thInlr
            if c.REFINE
                f2 = fit(pts1(inl, :), pts2(inl, :));
                inlrNum_refined(p) = length(find(dst(f2, pts1, pts2) < c.thDist));
                if inlrNum_refined(p) >= thInlr, TForms{p} = f2; end
            else
                TForms{p} = f1;
            end
        end
    end
    if c.REFINE, [maxInliers, idx] = max(inlrNum_refined); else, [maxInliers, idx] = max(inlrNum); end
    T = TForms{idx};
    try
        d = dst(T, pts1, pts2);
        inlierIdx = find(d < c.thDist);
        if c.REFINE, numSuccess = sum(inlrNum_refined >= thInlr); else, numSuccess = sum(inlrNum >= thInlr); end
        if VERBOSE
            fprintf('RANSAC succeeded %d times with a maximum of %d Inliers (%0.2f %%)\n', numSuccess, maxInliers, 100*maxInliers/ptNum);
        end
    catch
        if VERBOSE, fprintf('RANSAC could not find an appropriate transformation\n'); end
        T = []; inlierIdx = []; numSuccess = 0; maxInliers = 0;
    end
end
