function [pts_aligned, coeff_unambig, c] = AlignPoints_KNN(pts, varargin)
%ALIGNPOINTS_KNN  Drop-in for the reference's AlignPoints_KNN.m.
    if length(varargin) == 2, C1 = varargin{1}; C2 = varargin{2}; else, C1 = false; C2 = false; end
    [pts_aligned, coeff_unambig, c] = pcreg_mex('AlignPoints_KNN', double(pts), double(C1), double(C2));
end
