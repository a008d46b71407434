function T = estimateTransform(pts1, pts2)
%ESTIMATETRANSFORM  Drop-in for the reference's estimateTransform.m:
%   [pts2, 1] * T = [pts1, 1]; returns [] when rank(pts1) < 3 or rank(pts2) < 2.
    T = pcreg_mex('estimateTransform', double(pts1), double(pts2));
end
