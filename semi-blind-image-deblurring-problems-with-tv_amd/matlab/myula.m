function xMAP = myula(op, im)
% Replacement of SALSA/myula.m:1-22 (plain MYULA chain at fixed theta / PSF, last sample returned) through libsbtv.so
% (sbtv_myula), with the closures of SALSA/run_deblur_tv.m:126,131 built in: proxG = Chambolle prox with op.lambda*theta,
% gradF = AT(A x - y)/sigma2.  WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
%
% op: y, lambda, gamma, theta_op, samples as in the reference, plus what its closures captured and the C-ABI needs as data:
% op.psf (taps, or op.A to probe), op.sigma2 (the tau_op of gradF), op.chambolleit (default 25), op.seed (default 1).
% The normals come from the device Philox stream (MATLAB's randn stream cannot be reproduced).
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
[M, N] = size(op.y);
if isfield(op, 'psf'), h = op.psf; else, h = sbtv_psf_of_handle(op.A, M, N); end
K = 25; if isfield(op, 'chambolleit'), K = op.chambolleit; end
seed = 1; if isfield(op, 'seed'), seed = op.seed; end
px = libpointer('doublePtr', zeros(M,N));
rc = calllib('libsbtv', 'sbtv_myula', ctx, op.y, int32(M), int32(N), int32(1), h, int32(size(h,1)), op.lambda, op.gamma, ...
             op.theta_op, op.sigma2, int32(op.samples), int32(K), uint64(seed), int32(0), [], px, int32(0));
if rc ~= 0, error('sbtv:myula', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
xMAP = reshape(px.Value, M, N);
end
