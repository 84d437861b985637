function [x, objective, times, mses] = my_fista(b, A, AT, tau, L, Phi, Psi, stopcriterion, tolerance, maxiters, true, verbose)
% Replacement of SALSA/my_fista.m:5-56 for the TV prox of the demos (Psi = Chambolle prox with 25 cold iterations,
% Phi = TVnorm: run_moffat_demo.m:181-182), device-resident through libsbtv.so (sbtv_fista_tv).  Same signature.
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
%
% A / AT are the demos' blur handles: the taps are recovered by probing A(delta) (sbtv_psf_of_handle); Phi / Psi are not
% called (the GPU path has the TV pair built in) - pass the iteration count of the prox as the global SBTV_PROX_ITERS
% if it is not 25.
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
global SBTV_PROX_ITERS
prox_iters = 25; if ~isempty(SBTV_PROX_ITERS), prox_iters = SBTV_PROX_ITERS; end
if nargin < 12, verbose = 0; end
if (sum(stopcriterion == [1 2 3])==0), error('Invalid stopping criterion!'); end      % my_fista.m:45
[M, N] = size(b);
h = sbtv_psf_of_handle(A, M, N);
px = libpointer('doublePtr', zeros(M,N));
pobj = libpointer('doublePtr', zeros(1,maxiters)); pmse = libpointer('doublePtr', zeros(1,maxiters));
pn = libpointer('int32Ptr', int32(0));
t0 = tic;
rc = calllib('libsbtv', 'sbtv_fista_tv', ctx, b, int32(M), int32(N), int32(1), h, int32(size(h,1)), tau, L, ...
             int32(prox_iters), int32(stopcriterion), tolerance, int32(maxiters), int32(0), true, px, pobj, pmse, pn, int32(0));
if rc ~= 0, error('sbtv:my_fista', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
k = double(pn.Value);
x = reshape(px.Value, M, N); objective = pobj.Value(1:k); mses = pmse.Value(1:k);
times = linspace(0, toc(t0), k);          % the C-ABI returns no per-iteration clock for FISTA: evenly spaced wall time
if verbose, fprintf('iter = %d, obj = %3.3g\n', k, objective(end)); end
end
