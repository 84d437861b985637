function [X, numA, numAt, objective, distance, times, mses, n_outer] = sbtv_salsa_batch(Y, H, tau, mu, varargin)
% [X, numA, numAt, objective, distance, times, mses, n_outer] = sbtv_salsa_batch(Y, H, tau, mu, ...)
% SALSA_v2 (SALSA/SALSA_v2.m:156-494, TV path) on a BATCH of independent observations in ONE call - no counterpart in the
% reference, whose SALSA_v2 takes one image; this is the `*_batch` entry of SURVEY.md section 8b for a MATLAB host.
%   Y    M x N x B observations (MATLAB's column-major layout IS the batch layout of the C-ABI)
%   H    t x t PSF (one for all images) or t x t x B (one per image); t <= 15, top-left convention of utils/resize.m
%   tau, mu   scalars or 1 x B
% name / value options: 'TRUE_X' (M x N x B), 'INITIALIZATION' (0, 2 or an M x N x B array), 'STOPCRITERION' (1),
%   'TOLERANCEA' (1e-3), 'MAXITERA' (10000), 'TVITERS' (5)  - as in SALSA_v2.m:196-241 - and
%   'GROUP', g   an sbtv_group from sbtv_load_group(devices): the images are dealt to its GPUs in contiguous blocks
%                (sbtv_SALSA_v2_sharded).  Without it the call runs on GPU 0, where a batch is dealt to the two lanes
%                (internal streams) of the context: image k comes out bit for bit as from a call with that image alone.
% Outputs: X M x N x B; numA, numAt, n_outer 1 x B; objective, times, mses (maxiter+1) x B and distance maxiter x B, column b
% valid up to n_outer(b) (+1).
% WRITTEN WITHOUT ACCESS TO MATLAB: never executed, see INTEGRATION.md.
persistent ctx
stopCriterion = 1; maxiter = 10000; init = 0; tolA = 0.001; TViters = 5; true_x = []; xinit = []; g = [];
if (rem(length(varargin),2)==1), error('Optional parameters should always go by pairs'); end
for i = 1:2:(length(varargin)-1)
    switch upper(varargin{i})
        case 'TRUE_X',         true_x = varargin{i+1};
        case 'INITIALIZATION'
            if numel(varargin{i+1}) > 1, init = 33333; xinit = varargin{i+1}; else, init = varargin{i+1}; end
        case 'STOPCRITERION',  stopCriterion = varargin{i+1};
        case 'TOLERANCEA',     tolA = varargin{i+1};
        case 'MAXITERA',       maxiter = varargin{i+1};
        case 'TVITERS',        TViters = varargin{i+1};
        case 'GROUP',          g = varargin{i+1};
        otherwise, error(['Unrecognized option: ''' varargin{i} '''']);
    end
end
if (sum(stopCriterion == [1 2 3])==0), error('Unknown stopping criterion'); end
[M, N, B] = size(Y);
t = size(H, 1);
if size(H, 3) == 1, H = repmat(H, [1 1 B]); end
if numel(tau) == 1, tau = repmat(tau, 1, B); end
if numel(mu) == 1, mu = repmat(mu, 1, B); end
if size(H, 3) ~= B || numel(tau) ~= B || numel(mu) ~= B, error('sbtv:batch', 'H, tau and mu must be given once or once per image'); end
o = libstruct('sbtv_salsa_opts');
calllib('libsbtv', 'sbtv_salsa_opts_default', o);
o.stopcriterion = stopCriterion; o.maxiter = maxiter; o.TViters = TViters; o.initialization = init;
o.compute_mse = ~isempty(true_x); o.tolA = tolA;
pX = libpointer('doublePtr', zeros(M, N, B));
pobj = libpointer('doublePtr', zeros(maxiter+1, B)); pdist = libpointer('doublePtr', zeros(maxiter, B));
ptim = libpointer('doublePtr', zeros(maxiter+1, B)); pmse = libpointer('doublePtr', zeros(maxiter+1, B));
pnA = libpointer('int32Ptr', zeros(1, B, 'int32')); pnAt = libpointer('int32Ptr', zeros(1, B, 'int32'));
pn = libpointer('int32Ptr', zeros(1, B, 'int32'));
if isempty(g)
    if isempty(ctx), ctx = sbtv_load(0); end
    rc = calllib('libsbtv', 'sbtv_SALSA_v2', ctx, Y, int32(M), int32(N), int32(B), H, int32(t), tau, mu, o, ...
                 true_x, xinit, pX, pobj, pdist, ptim, pmse, pnA, pnAt, pn, int32(0));
    if rc ~= 0, error('sbtv:batch', '%s', calllib('libsbtv', 'sbtv_last_error', ctx)); end
else
    rc = calllib('libsbtv', 'sbtv_SALSA_v2_sharded', g, Y, int32(M), int32(N), int32(B), H, int32(t), tau, mu, o, ...
                 true_x, xinit, pX, pobj, pdist, ptim, pmse, pnA, pnAt, pn);
    if rc ~= 0, error('sbtv:batch', '%s', calllib('libsbtv', 'sbtv_group_last_error', g)); end
end
X = reshape(pX.Value, M, N, B);
numA = double(pnA.Value); numAt = double(pnAt.Value); n_outer = double(pn.Value);
objective = reshape(pobj.Value, maxiter+1, B); distance = reshape(pdist.Value, maxiter, B);
times = reshape(ptim.Value, maxiter+1, B);
if ~isempty(true_x), mses = reshape(pmse.Value, maxiter+1, B); else, mses = []; end
end
