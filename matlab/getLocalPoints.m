function [pts_sphere, dists] = getLocalPoints(pts, R, c, min_points, max_points)
%GETLOCALPOINTS  Drop-in for the reference's getLocalPoints.m: the points of pts strictly within R of c, RELATIVE to c
%   (and their distances), or [] when fewer than min_points lie in the box / the ball or more than max_points in the ball.
%   Classes as MATLAB's own function: a single cloud or centre is evaluated in single arithmetic and returns singles.
    if ~isa(pts, 'single'), pts = double(pts); end
    if ~isa(c, 'single'), c = double(c); end
    [pts_sphere, dists] = pcreg_mex('getLocalPoints', pts, double(R), c, double(min_points), double(max_points));
end
