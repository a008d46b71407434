function [pts_aligned, coeff_unambig, c] = AlignPoints_KNN(pts, varargin)
%ALIGNPOINTS_KNN  Drop-in for the reference's AlignPoints_KNN.m.
%   single in -> single out (the class MATLAB's own function returns); the arithmetic is double either way
%   (INTEGRATION.md, "Differences a user can observe").
    if length(varargin) == 2, C1 = varargin{1}; C2 = varargin{2}; else, C1 = false; C2 = false; end
    if ~isa(pts, 'single'), pts = double(pts); end
    [pts_aligned, coeff_unambig, c] = pcreg_mex('AlignPoints_KNN', pts, double(C1), double(C2));
end
