function [x, numA, numAt, objective, distance1, distance2, criterion, times, mses] = csalsa(y, A, mu1, mu2, sigma, varargin)
% Replacement of SALSA/CSALSA_v2.m:160-561 (function csalsa) for its TV path ('TVINITIALIZATION', 1, P = PT = identity)
% through libsbtv.so (sbtv_CSALSA_v2).  Same signature and name/value options.
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
stopCriterion = 3; maxiter = 10000; init = 0; AT = 0; tolA = 0.001; isTV = 0; TViters = 5; verbose = 1; isinvLS = 0;
delta = 1; epsilon = 0; compute_mse = 0; true_x = []; xinit = []; h = []; definedP = 0; definedPT = 0; P = []; PT = [];
if (rem(length(varargin),2)==1), error('Optional parameters should always go by pairs'); end
for i = 1:2:(length(varargin)-1)
    switch upper(varargin{i})
        case 'PSF',                h = varargin{i+1};
        case {'PSI','PHI'}         % accepted and ignored on the TV path (CSALSA_v2.m:331-333: ignored with a warning)
        case 'P',                  definedP = 1; P = varargin{i+1};       % CSALSA_v2.m:208-210
        case 'PT',                 definedPT = 1; PT = varargin{i+1};     % CSALSA_v2.m:211-213
        case 'TVINITIALIZATION',   isTV = varargin{i+1};
        case 'TVITERS',            TViters = varargin{i+1};
        case 'STOPCRITERION',      stopCriterion = varargin{i+1};
        case 'TOLERANCEA',         tolA = varargin{i+1};
        case 'MAXITERA',           maxiter = varargin{i+1};
        case 'INITIALIZATION'
            if numel(varargin{i+1}) > 1, init = 33333; xinit = varargin{i+1}; else, init = varargin{i+1}; end
        case 'TRUE_X',             compute_mse = 1; true_x = varargin{i+1};
        case 'AT',                 AT = varargin{i+1};
        case 'LS',                 isinvLS = 1;
        case 'VERBOSE',            verbose = varargin{i+1};
        case 'CONTINUATIONFACTOR', delta = varargin{i+1};
        case 'EPSILON',            epsilon = varargin{i+1};
        otherwise, error(['Unrecognized option: ''' varargin{i} '''']);
    end
end
if (sum(stopCriterion == [1 2 3])==0), error('Unknown stopping criterion'); end
if isa(A, 'function_handle') && ~isa(AT,'function_handle'), error('The function handle for transpose of A is missing'); end
if ~isinvLS, error('(A^T A + \mu I)^(-1) must be specified as a function handle.\n'); end
if ~isTV, error('sbtv:csalsa', 'only ''TVINITIALIZATION'',1 runs on the GPU path'); end
[M, N] = size(y);
sbtv_check_identity('sbtv:csalsa', P, PT, definedP, definedPT, M, N, 'P', 'PT');     % PTx = PT(x) (CSALSA_v2.m:402,473)
if isempty(h), h = sbtv_psf_of_handle(A, M, N); end
o = libstruct('sbtv_salsa_opts');
calllib('libsbtv', 'sbtv_salsa_opts_default', o);
o.stopcriterion = stopCriterion; o.maxiter = maxiter; o.TViters = TViters; o.initialization = init;
o.compute_mse = compute_mse; o.tolA = tolA;
z = @() libpointer('doublePtr', zeros(1,maxiter));
px = libpointer('doublePtr', zeros(M,N)); pobj = z(); pd1 = z(); pd2 = z(); pcr = z(); ptim = z(); pmse = z();
pnA = libpointer('int32Ptr', int32(0)); pnAt = libpointer('int32Ptr', int32(0)); pn = libpointer('int32Ptr', int32(0));
rc = calllib('libsbtv', 'sbtv_CSALSA_v2', ctx, y, int32(M), int32(N), int32(1), h, int32(size(h,1)), mu1, mu2, sigma, ...
             epsilon, delta, o, true_x, xinit, px, pobj, pd1, pd2, pcr, ptim, pmse, pnA, pnAt, pn, int32(0));
if rc ~= 0, error('sbtv:csalsa', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
k = double(pn.Value);                      % traces are 1-based like the reference: entry 1 = the state before the loop
x = reshape(px.Value, M, N); numA = double(pnA.Value); numAt = double(pnAt.Value);
objective = pobj.Value(1:k); distance1 = pd1.Value(1:k); distance2 = pd2.Value(1:k); criterion = pcr.Value(1:k);
times = ptim.Value(1:k);
if compute_mse, mses = pmse.Value(1:k); else, mses = []; end
if verbose, fprintf('\niter = %d, obj = %3.3g, criterion = %3.3g\n', k, objective(end), criterion(end)); end
end
