function sm = sphereSweepModel(hModel, featModel, centres, R_desc, min_pts, max_pts)
%SPHERESWEEPMODEL  The model side of sphereSweep, made once per model (completeExperimentFast.m runs once per surface crop).
%   sm = sphereSweepModel(hModel, featModel, centres, R_desc, min_pts, max_pts) counts the model keypoints in every candidate sphere
%   (:52-64), keeps those with min_pts <= count <= max_pts, and builds in the library what the sweep needs of the model alone: the
%   kept spheres' row lists and keypoints, the model descriptor set (hModel = pcreg_mex('descCreate', double(descModel))) restricted
%   to the rows that lie in some sphere, its powered rows.  Then, for every surface:
%       out = sphereSweepOn(sm, hSurface, featSurface, par, putativeThresh, ransacCoef, seed);
%   and at the end pcreg_mex('sphereModelDestroy', sm.handle).  Fields: handle, valid, centres, numDesc, modelRows (as sphereSweep's).
    counts = pcreg_mex('sphereCounts', double(featModel), double(centres), R_desc);
    valid = counts >= min_pts & counts <= max_pts;
    c = double(centres(valid, :)); nd = int32(counts(valid));
    [h, rows] = pcreg_mex('sphereModelCreate', hModel, double(featModel), c, nd(:), R_desc);
    sm.handle = h; sm.valid = valid; sm.centres = c; sm.numDesc = double(nd(:));
    if isempty(nd), sm.modelRows = cell(0, 1); else, sm.modelRows = mat2cell(reshape(rows, [], 1), double(nd(:)), 1); end
end
