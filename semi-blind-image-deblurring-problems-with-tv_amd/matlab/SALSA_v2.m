function [x, numA, numAt, objective, distance, times, mses] = SALSA_v2(y, A, tau, varargin)
% Replacement of SALSA/SALSA_v2.m for the TV path of the demos, running device-resident on the MI355X
% through libsbtv.so.  Same signature and name/value options, so the call at run_Gaussian_demo.m:229-242
% runs unchanged.  WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
%
% The C-ABI takes the operator as PSF taps, not as function handles.  They are recovered from the handle
% the caller already passes: the reference pads the kernel into the TOP-LEFT corner (utils/resize.m:8-11),
% so A(delta) is the padded kernel; it must be confined to a top-left square of at most 15 x 15.  The mu
% the 'LS' filter was built with is 1/sum(invLS(delta)) - sum(h)^2 (run_Gaussian_demo.m:222-225) and must
% equal 'MU' (the C-ABI has one mu).  The optional extra option 'PSF', h skips the probe of A.
% A / 'AT' / 'LS' are otherwise only checked for presence, like the reference checks them.
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
stopCriterion = 1; maxiter = 10000; init = 0; AT = 0; mu = 1e-3; tolA = 0.001;
isTVinitialization = 0; TViters = 5; verbose = 1; isinvLS = 0; invLS = []; compute_mse = 0; h = []; true_x = []; xinit = [];
definedP = 0; definedPT = 0; P = []; PT = [];
if (rem(length(varargin),2)==1)
    error('Optional parameters should always go by pairs');
end
for i = 1:2:(length(varargin)-1)
    switch upper(varargin{i})
        case 'PSF',              h = varargin{i+1};
        case {'PSI','PHI'}       % accepted and ignored: with 'TVINITIALIZATION' = 1 the reference ignores them too
                                 % and prints a warning (SALSA_v2.m:318-320,354-359, quirk Q7)
        case 'P',                definedP = 1; P = varargin{i+1};      % SALSA_v2.m:198-200
        case 'PT',               definedPT = 1; PT = varargin{i+1};    % SALSA_v2.m:201-203
        case 'TVINITIALIZATION', isTVinitialization = varargin{i+1};
        case 'TVITERS',          TViters = varargin{i+1};
        case 'MU',               mu = varargin{i+1};
        case 'STOPCRITERION',    stopCriterion = varargin{i+1};
        case 'TOLERANCEA',       tolA = varargin{i+1};
        case 'MAXITERA',         maxiter = varargin{i+1};
        case 'INITIALIZATION'
            if numel(varargin{i+1}) > 1, init = 33333; xinit = varargin{i+1}; else, init = varargin{i+1}; end
        case 'TRUE_X',           compute_mse = 1; true_x = varargin{i+1};
        case 'AT',               AT = varargin{i+1};
        case 'VERBOSE',          verbose = varargin{i+1};
        case 'LS',               isinvLS = 1; invLS = varargin{i+1};
        otherwise
            error(['Unrecognized option: ''' varargin{i} '''']);
    end
end
if (sum(stopCriterion == [1 2 3])==0), error('Unknown stopping criterion'); end
if isa(A, 'function_handle') && ~isa(AT,'function_handle')
    error('The function handle for transpose of A is missing');
end
if ~isinvLS, error('(A^T A + \mu I)^(-1) must be specified as a function handle.\n'); end
if ~isTVinitialization, error('sbtv:SALSA_v2', 'only ''TVINITIALIZATION'',1 runs on the GPU path'); end
[M, N] = size(y);
sbtv_check_identity('sbtv:SALSA_v2', P, PT, definedP, definedPT, M, N, 'P', 'PT');   % r = ATy + mu*P(u+bu) (SALSA_v2.m:434)
delta = zeros(M, N); delta(1,1) = 1;
if isempty(h)
    if ~isa(A, 'function_handle'), error('sbtv:SALSA_v2', 'A must be a function handle (or pass ''PSF'', h)'); end
    hp = A(delta);                                   % = kernel zero-padded into the top-left corner (resize.m:8-11)
    sig = abs(hp) > 1e-10 * max(abs(hp(:)));
    [ri, ci] = find(sig);
    t = max([ri; ci]);
    if isempty(t) || t > 15 || t > M || t > N
        error('sbtv:SALSA_v2', 'A(delta) is not confined to a top-left square of at most 15 x 15 (Mask does not fit)');
    end
    h = hp(1:t, 1:t);
end
if isa(invLS, 'function_handle')
    mu_ls = 1 / sum(sum(invLS(delta))) - sum(h(:))^2;   % DC gain of 1./(abs(H).^2 + mu)
    if abs(mu_ls - mu) > 1e-6 * abs(mu)
        error('sbtv:SALSA_v2', '''MU'' (%g) differs from the mu of the ''LS'' filter (%g): the GPU path has one mu', mu, mu_ls);
    end
end
o = libstruct('sbtv_salsa_opts');
calllib('libsbtv', 'sbtv_salsa_opts_default', o);
o.stopcriterion = stopCriterion; o.maxiter = maxiter; o.TViters = TViters; o.initialization = init;
o.compute_mse = compute_mse; o.tolA = tolA;
px = libpointer('doublePtr', zeros(M,N));
pobj = libpointer('doublePtr', zeros(1,maxiter+1)); pdist = libpointer('doublePtr', zeros(1,maxiter));
ptim = libpointer('doublePtr', zeros(1,maxiter+1)); pmse = libpointer('doublePtr', zeros(1,maxiter+1));
pnA = libpointer('int32Ptr', int32(0)); pnAt = libpointer('int32Ptr', int32(0)); pn = libpointer('int32Ptr', int32(0));
rc = calllib('libsbtv', 'sbtv_SALSA_v2', ctx, y, int32(M), int32(N), int32(1), h, int32(size(h,1)), tau, mu, o, ...
             true_x, xinit, px, pobj, pdist, ptim, pmse, pnA, pnAt, pn, int32(0));
if rc ~= 0, error('sbtv:SALSA_v2', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
k = double(pn.Value);
x = reshape(px.Value, M, N); numA = double(pnA.Value); numAt = double(pnAt.Value);
objective = pobj.Value(1:k+1); distance = pdist.Value(1:k); times = ptim.Value(1:k+1);
if compute_mse, mses = pmse.Value(1:k+1); else, mses = []; end
if verbose
    fprintf('\niter = %d, obj = %3.3g\n', k, objective(end));
end
end
