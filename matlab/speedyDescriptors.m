function [feat, desc] = speedyDescriptors(pts, sample_opts, options)
%SPEEDYDESCRIPTORS  Drop-in for the reference's speedyDescriptors.m: same signature, same keypoints, same rows.
%   The reference tiles the model into cuboid regions (speedyDescriptors.m:17-27) because its getLocalPoints scans the whole
%   cloud per keypoint; it samples keypoints region by region (:55, :86-101) and calls getSpacialHistogramDescriptors once
%   per region on the region's crop.  A keypoint lies at least R inside its crop's bounding box, so its support is the same
%   in the crop and in the whole cloud.  This file keeps the region walk and the host-side sampling -- the same rand calls
%   in the same order, hence the same keypoints for the same rng state -- and hands ALL keypoints to ONE
%   getSpacialHistogramDescriptors call on the whole cloud (the library's uniform grid does what the tiling was for):
%   feat / desc come back in the same order the region loop would have stacked them.
%   options: the descriptor options plus max_region_size and VERBOSE; sample_opts.d: sampling distance.
    VERBOSE = options.VERBOSE;
    options.VERBOSE = options.VERBOSE - 1;                                        % :12
    d = sample_opts.d;
    R = options.R;
    lim = [min(pts, [], 1); max(pts, [], 1)];                                     % pointCloud limits, :17-18
    ext = (lim(2, :) - lim(1, :))';
    nReg = ceil(ext / options.max_region_size);                                   % :20
    step = ext ./ nReg;                                                           % :21
    edges = cell(3, 1);
    for a = 1:3, edges{a} = lim(1, a):step(a):lim(2, a); end                      % :24-26
    assert(isequal(cellfun(@numel, edges) - 1, nReg));                            % :27
    if VERBOSE, fprintf('Divided model into %d regions\n', prod(nReg)); end       % :35-37
    tic
    draws = cell(prod(nReg), 1);
    k = 0;
    for ix = 1:nReg(1)                                                            % region order of :44-46
        inx = pts(:, 1) > edges{1}(ix) - R & pts(:, 1) < edges{1}(ix + 1) + R;
        for iy = 1:nReg(2)
            inxy = inx & pts(:, 2) > edges{2}(iy) - R & pts(:, 2) < edges{2}(iy + 1) + R;
            for iz = 1:nReg(3)
                k = k + 1;
                crop = pts(inxy & pts(:, 3) > edges{3}(iz) - R & pts(:, 3) < edges{3}(iz + 1) + R, :);    % :48-52
                if size(crop, 1) > 500                                            % :87
                    clo = min(crop, [], 1); span = max(crop, [], 1) - clo - 2 * R;                        % :89-91 (margin -R)
                    num = round(prod(span) / d^3);                                % :92
                    draws{k} = rand(num, 3) .* span + clo + R;                    % :95-101: the reference's rand call, same shape
                end
            end
        end
    end
    spts = vertcat(draws{:});
    if isempty(spts)
        feat = []; desc = [];
    else
        [feat, desc] = getSpacialHistogramDescriptors(pts, spts, options);        % ONE library call instead of :58-59 per region
    end
    if VERBOSE, fprintf('Calculated descriptors in %0.1f seconds...\n', toc); end % :77-79
end
