function [A, AT, invLS, dA] = sbtv_operators(h, mu, dh)
% [A, AT, invLS, dA] = sbtv_operators(h, mu, dh)
% GPU-backed versions of the demos' operator closures (run_Gaussian_demo.m:136-139,224-225):
%   A(x)     = real(ifft2(resize(h)  .* fft2(x)))        sbtv_A_wrapper mode 1
%   AT(x)    = real(ifft2(conj(...)) .* fft2(x)))        mode 2
%   invLS(x) = real(ifft2(fft2(x) ./ (abs(H).^2 + mu)))  mode 9
%   dA(x)    = operator of the derivative taps dh        mode 3
% h, dh: taille x taille taps (Gaussian_psf / psf_moffat / psf_laplace and the diff_* taps).
ctx = sbtv_load(0);
A     = @(x) apply(ctx, h, [], x, 1);
AT    = @(x) apply(ctx, h, [], x, 2);
invLS = @(x) apply(ctx, h, mu, x, 9);
if nargin > 2, dA = @(x) apply(ctx, dh, [], x, 3); else, dA = []; end
end

function out = apply(ctx, h, mu, x, mode)
[M, N] = size(x);
po = libpointer('doublePtr', zeros(M, N));
rc = calllib('libsbtv', 'sbtv_A_wrapper', ctx, h, int32(size(h,1)), mu, x, po, int32(M), int32(N), int32(1), ...
             int32(mode), int32(0));
if rc ~= 0, error('sbtv:A_wrapper', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
out = reshape(po.Value, M, N);
end
