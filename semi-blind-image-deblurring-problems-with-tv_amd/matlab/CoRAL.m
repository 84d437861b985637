function [x, numA, numAt, objective, distance, times, mses] = CoRAL(y, A, tau1, tau2, varargin)
% Replacement of SALSA/CoRAL_v2.m:2-476 (function CoRAL) for two TV terms ('TVINITIALIZATION1' = 'TVINITIALIZATION2' = 1)
% through libsbtv.so (sbtv_CoRAL_v2).  Same signature and name/value options.
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
persistent ctx
if isempty(ctx), ctx = sbtv_load(0); end
stopCriterion = 1; maxiter = 10000; init = 0; AT = 0; tolA = 0.001; mu1 = 1e-3; mu2 = 1e-3; isTV1 = 0; isTV2 = 0;
TViters1 = 5; TViters2 = 5; verbose = 1; isinvLS = 0; invLS = []; compute_mse = 0; true_x = []; xinit = []; h = [];
dP = [0 0 0 0]; Pops = {[], [], [], []};      % P1, P1T, P2, P2T (CoRAL_v2.m:61-72)
if (rem(length(varargin),2)==1), error('Optional parameters should always go by pairs'); end
for i = 1:2:(length(varargin)-1)
    switch upper(varargin{i})
        case 'PSF',               h = varargin{i+1};
        case {'PSI1','PHI1','PSI2','PHI2'}    % accepted and ignored on the TV paths (CoRAL_v2.m:79-92,229-231,277-279)
        % options the reference parses but whose only uses are in commented-out code (CoRAL_v2.m:479-563): no effect
        % there, none here
        case {'W','WT','MASK','UNITARYTRANSFORMDOMAINMASK','CONVOLUTIONFILTER','INNERITERS'}    % CoRAL_v2.m:55-60,73-78,103-104
        case 'P1',                dP(1) = 1; Pops{1} = varargin{i+1};
        case 'P1T',               dP(2) = 1; Pops{2} = varargin{i+1};
        case 'P2',                dP(3) = 1; Pops{3} = varargin{i+1};
        case 'P2T',               dP(4) = 1; Pops{4} = varargin{i+1};
        case 'TVINITIALIZATION1', isTV1 = varargin{i+1};
        case 'TVINITIALIZATION2', isTV2 = varargin{i+1};
        case 'TVITERS1',          TViters1 = varargin{i+1};
        case 'TVITERS2',          TViters2 = varargin{i+1};
        case 'MU1',               mu1 = varargin{i+1};
        case 'MU2',               mu2 = varargin{i+1};
        case 'STOPCRITERION',     stopCriterion = varargin{i+1};
        case 'TOLERANCEA',        tolA = varargin{i+1};
        case 'MAXITERA',          maxiter = varargin{i+1};
        case 'INITIALIZATION'
            if numel(varargin{i+1}) > 1, init = 33333; xinit = varargin{i+1}; else, init = varargin{i+1}; end
        case 'TRUE_X',            compute_mse = 1; true_x = varargin{i+1};
        case 'AT',                AT = varargin{i+1};
        case 'LS',                isinvLS = 1; invLS = varargin{i+1};
        case 'VERBOSE',           verbose = varargin{i+1};
        otherwise, error(['Unrecognized option: ''' varargin{i} '''']);
    end
end
if (sum(stopCriterion == [1 2 3])==0), error('Unknown stopping criterion'); end
if isa(A, 'function_handle') && ~isa(AT,'function_handle'), error('The function handle for transpose of A is missing'); end
if ~isinvLS, error('(A^T A + \mu I)^(-1) must be specified as a function handle.\n'); end
if ~(isTV1 && isTV2), error('sbtv:CoRAL', 'only the TV + TV problem runs on the GPU path'); end
[M, N] = size(y);
% r = ATy + mu1*P1(u+bu) + mu2*P2(v+bv), P1Tx = P1T(x), P2Tx = P2T(x) (CoRAL_v2.m:349-350,411,417-418): identity only
sbtv_check_identity('sbtv:CoRAL', Pops{1}, Pops{2}, dP(1), dP(2), M, N, 'P1', 'P1T');
sbtv_check_identity('sbtv:CoRAL', Pops{3}, Pops{4}, dP(3), dP(4), M, N, 'P2', 'P2T');
if isempty(h), h = sbtv_psf_of_handle(A, M, N); end
mu_ls = mu1 + mu2;                          % CoRAL_v2.m:137; a different filter weight is read off the 'LS' handle
if isa(invLS, 'function_handle')
    delta = zeros(M, N); delta(1,1) = 1;
    mu_ls = 1 / sum(sum(invLS(delta))) - sum(h(:))^2;
end
o = libstruct('sbtv_salsa_opts');
calllib('libsbtv', 'sbtv_salsa_opts_default', o);
o.stopcriterion = stopCriterion; o.maxiter = maxiter; o.TViters = TViters1; o.initialization = init;
o.compute_mse = compute_mse; o.tolA = tolA;
px = libpointer('doublePtr', zeros(M,N));
pobj = libpointer('doublePtr', zeros(1,maxiter+1)); pdist = libpointer('doublePtr', zeros(2,maxiter));
ptim = libpointer('doublePtr', zeros(1,maxiter+1)); pmse = libpointer('doublePtr', zeros(1,maxiter+1));
pnA = libpointer('int32Ptr', int32(0)); pnAt = libpointer('int32Ptr', int32(0)); pn = libpointer('int32Ptr', int32(0));
rc = calllib('libsbtv', 'sbtv_CoRAL_v2', ctx, y, int32(M), int32(N), int32(1), h, int32(size(h,1)), tau1, tau2, mu1, mu2, ...
             mu_ls, int32(TViters2), o, true_x, xinit, px, pobj, pdist, ptim, pmse, pnA, pnAt, pn, int32(0));
if rc ~= 0, error('sbtv:CoRAL', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
k = double(pn.Value);
x = reshape(px.Value, M, N); numA = double(pnA.Value); numAt = double(pnAt.Value);
objective = pobj.Value(1:k+1); d = reshape(pdist.Value, 2, maxiter); distance = d(:, 1:k); times = ptim.Value(1:k+1);
if compute_mse, mses = pmse.Value(1:k+1); else, mses = []; end
if verbose, fprintf('\niter = %d, obj = %3.3g\n', k, objective(end)); end
end
