function h = sbtv_psf_of_handle(A, M, N)
% h = sbtv_psf_of_handle(A, M, N)  - the PSF taps behind a blur handle of the demos.
% The reference pads the kernel into the TOP-LEFT corner of the image (utils/resize.m:8-11), so A(delta) IS the padded
% kernel; it must be confined to a top-left square of at most 15 x 15 (the C-ABI's PSF size limit).
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
delta = zeros(M, N); delta(1,1) = 1;
hp = A(delta);
sig = abs(hp) > 1e-10 * max(abs(hp(:)));
[ri, ci] = find(sig);
t = max([ri; ci]);
if isempty(t) || t > 15 || t > M || t > N
    error('sbtv:psf', 'A(delta) is not confined to a top-left square of at most 15 x 15 (Mask does not fit)');
end
h = hp(1:t, 1:t);
end
